// mifsk_hostpipe.cpp -- the receive path for streams that start in HOST memory or in
// FILES (SURVEY 8 d "H2D-inclusive", 8 f3; reference: src/simpleaudio-sndfile.c:42-74,
// src/minimodem.c:1014-1032 -- `--rx --file x.wav`).
//
// The kernels take 0.5 ms for a batch that needs 35 ms to cross PCIe, so what a
// host caller sees is the copy.  This file makes the copy the only thing it sees:
//
//   chunk k+1:  stage (threads: memcpy / pread into pinned memory, or nothing when the
//               caller's memory is pinned) -> hipMemcpyAsync on the copy stream
//   chunk k  :  S16 -> float + --Xrxnoise (device, mifsk_ingest.hip) -> receive loop
//               (mifsk_demod_batch) on the compute stream
//   chunk k-1:  results -> host on the output stream
//
// Two slots of everything; events order the three streams; the host thread only
// waits for a slot when it is about to reuse it.  16-bit input crosses the bus as
// 16-bit and is converted on the device, exactly as libsndfile converts it
// (value / 32768).  Nothing here computes on the host: no device, no result.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "mifsk.h"
#include "mifsk_ctx.h"

namespace mifsk {

// wav header of a file of which only the first `have` bytes are in memory (mifsk_ingest.hip)
int wav_parse_sized( const void *file, size_t have, size_t file_size, mifsk_wav_info *info );

struct HostWork {
    std::mutex	lock;			// one host call at a time per context
    hipStream_t	s_in = nullptr, s_comp = nullptr, s_out = nullptr;
    hipEvent_t	ev_in[2] = { nullptr, nullptr }, ev_comp[2] = { nullptr, nullptr }, ev_out[2] = { nullptr, nullptr };
    void	*pin[2] = { nullptr, nullptr };	// staging for sources that are not page-locked
    size_t	pin_cap = 0;
    uint32_t	*pin_n[2] = { nullptr, nullptr };	// per-chunk stream lengths
    size_t	pin_n_cap = 0;
    bool	ready = false;
};

void host_work_destroy( HostWork *w )
{
    if ( !w )
	return;
    for ( int i = 0; i < 2; i++ ) {
	if ( w->ev_in[i] ) (void)hipEventDestroy(w->ev_in[i]);
	if ( w->ev_comp[i] ) (void)hipEventDestroy(w->ev_comp[i]);
	if ( w->ev_out[i] ) (void)hipEventDestroy(w->ev_out[i]);
	if ( w->pin[i] ) (void)hipHostFree(w->pin[i]);
	if ( w->pin_n[i] ) (void)hipHostFree(w->pin_n[i]);
    }
    if ( w->s_in ) (void)hipStreamDestroy(w->s_in);
    if ( w->s_comp ) (void)hipStreamDestroy(w->s_comp);
    if ( w->s_out ) (void)hipStreamDestroy(w->s_out);
    delete w;
}

namespace {

constexpr size_t kChunkBytes = 64u << 20;	// of input per chunk: ~1 ms of PCIe gen5 x16

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int host_work_get( mifsk_ctx *ctx, HostWork **out )
{
    std::lock_guard<std::mutex> g(ctx->lock);
    if ( !ctx->host )
	ctx->host = new (std::nothrow) HostWork();
    if ( !ctx->host )
	return -ENOMEM;
    *out = ctx->host;
    return 0;
}

// `overlapped`: the job has more than one chunk, i.e. something to overlap: the copy streams and
// the events that order the three are made then.  A job of one chunk -- a file, a few files: the
// reference's own use -- runs on s_comp alone (a stream costs ~10 ms to make: a third of what
// such a call spent inside the library).
int host_work_init( HostWork *w, bool overlapped )
{
    if ( w->ready || !overlapped )
	return 0;
    if ( !w->s_comp )
	HIP_OK(hipStreamCreateWithFlags(&w->s_comp, hipStreamNonBlocking));
    HIP_OK(hipStreamCreateWithFlags(&w->s_in, hipStreamNonBlocking));
    HIP_OK(hipStreamCreateWithFlags(&w->s_out, hipStreamNonBlocking));
    for ( int i = 0; i < 2; i++ ) {
	HIP_OK(hipEventCreateWithFlags(&w->ev_in[i], hipEventDisableTiming));
	HIP_OK(hipEventCreateWithFlags(&w->ev_comp[i], hipEventDisableTiming));
	HIP_OK(hipEventCreateWithFlags(&w->ev_out[i], hipEventDisableTiming));
    }
    w->ready = true;
    return 0;
}

int pin_reserve( HostWork *w, size_t bytes, size_t nrows )
{
    if ( bytes > w->pin_cap ) {
	for ( int i = 0; i < 2; i++ ) {
	    if ( w->pin[i] ) (void)hipHostFree(w->pin[i]);
	    w->pin[i] = nullptr;
	}
	w->pin_cap = 0;
	for ( int i = 0; i < 2; i++ )
	    if ( hipHostMalloc(&w->pin[i], bytes, hipHostMallocDefault) != hipSuccess )
		return -ENOMEM;
	w->pin_cap = bytes;
    }
    if ( nrows > w->pin_n_cap ) {
	for ( int i = 0; i < 2; i++ ) {
	    if ( w->pin_n[i] ) (void)hipHostFree(w->pin_n[i]);
	    w->pin_n[i] = nullptr;
	}
	w->pin_n_cap = 0;
	for ( int i = 0; i < 2; i++ )
	    if ( hipHostMalloc((void **)&w->pin_n[i], nrows * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess )
		return -ENOMEM;
	w->pin_n_cap = nrows;
    }
    return 0;
}

bool is_pinned( const void *p )
{
    if ( !p )
	return false;
    hipPointerAttribute_t a;
    std::memset(&a, 0, sizeof(a));
    if ( hipPointerGetAttributes(&a, p) != hipSuccess ) {
	(void)hipGetLastError();		// ordinary memory: not an error
	return false;
    }
    return a.type == hipMemoryTypeHost;
}

// fn(i) for i in [0, n) on up to `nthreads` threads (the caller's included)
template <typename F>
void parallel_for( size_t n, unsigned nthreads, F fn )
{
    if ( n == 0 )
	return;
    if ( nthreads > n ) nthreads = (unsigned)n;
    if ( nthreads <= 1 ) {
	for ( size_t i = 0; i < n; i++ ) fn(i);
	return;
    }
    // An exception on a worker thread would terminate the process, and one thrown while the
    // threads are being created would destroy joinable threads (the same): every item runs
    // inside a try, the first exception is kept, every thread that did start is joined, and
    // the exception is rethrown on the caller -- whose own catch turns it into an error code
    // at the C ABI.
    std::atomic<size_t> next(0);
    std::exception_ptr first;
    std::mutex first_lock;
    auto body = [&]() {
	for (;;) {
	    const size_t i = next.fetch_add(1);
	    if ( i >= n ) break;
	    try {
		fn(i);
	    } catch ( ... ) {
		std::lock_guard<std::mutex> g(first_lock);
		if ( !first )
		    first = std::current_exception();
		next.store(n);			// nothing more is started
	    }
	}
    };
    std::vector<std::thread> ts;
    try {
	ts.reserve(nthreads);
	for ( unsigned t = 1; t < nthreads; t++ )
	    ts.emplace_back(body);
    } catch ( ... ) {				// thread creation failed: go on with those there are
    }
    body();
    for ( std::thread &t : ts )
	t.join();
    if ( first )
	std::rethrow_exception(first);
}

unsigned staging_threads()
{
    unsigned hw = std::thread::hardware_concurrency();
    if ( hw == 0 ) hw = 4;
    unsigned t = hw / 2;
    if ( t < 2 ) t = 2;
    if ( t > 16 ) t = 16;
    return t;
}

// one stream of a job: `n` samples at `mem`, or in the file `path` from byte `off`
struct Row {
    const void	*mem;
    const char	*path;
    uint64_t	off;
    uint32_t	n;
    int		*err;		// file rows: where a read error is reported
};

struct Job {
    const mifsk_rx_config	*cfg;
    std::vector<Row>		rows;
    bool			s16;		// rows are int16_t (else float)
    float			rxnoise;
    unsigned			flags;		// MIFSK_IO_* for mifsk_demod_batch
    // rows are equally spaced in one host array (mem rows only): pitch in elements
    size_t			src_pitch;
    // host result arrays, [rows][cap] (any may be NULL)
    mifsk_demod_io		out;
    mifsk_host_stats		*stats;
};

struct Chunk {
    size_t	lo, hi;		// rows
    size_t	stride;		// elements per device row
};

struct Slot {
    void		*d_in = nullptr;	// the chunk as it crossed the bus
    float		*d_x = nullptr;		// S16 input: the converted samples
    uint32_t		*d_n = nullptr;
    uint8_t		*d_bytes = nullptr;
    uint64_t		*d_bits = nullptr;
    mifsk_frame		*d_frames = nullptr;
    mifsk_episode	*d_eps = nullptr;
    uint32_t		*d_nbytes = nullptr, *d_nframes = nullptr, *d_neps = nullptr, *d_status = nullptr;
    int32_t		*d_band = nullptr;
    uint64_t		*d_cnt = nullptr;
};

struct SlotGuard {
    Slot s[2];
    ~SlotGuard()
    {
	for ( Slot &x : s ) {
	    void *ps[] = { x.d_in, x.d_x, x.d_n, x.d_bytes, x.d_bits, x.d_frames, x.d_eps, x.d_nbytes,
			   x.d_nframes, x.d_neps, x.d_status, x.d_band, x.d_cnt };
	    for ( void *p : ps )
		if ( p ) (void)hipFree(p);
	}
    }
};

template <typename T>
int dev_alloc( T **p, size_t n )
{
    return hipMalloc((void **)p, ( n ? n : 1 ) * sizeof(T)) == hipSuccess ? 0 : -ENOMEM;
}

int read_fully( int fd, void *buf, size_t n, uint64_t off )
{
    unsigned char *p = (unsigned char *)buf;
    while ( n ) {
	const ssize_t r = pread(fd, p, n, (off_t)off);
	if ( r < 0 ) {
	    if ( errno == EINTR ) continue;
	    return -errno;
	}
	if ( r == 0 )
	    return -EIO;		// shorter than its header said a moment ago
	p += r; off += (uint64_t)r; n -= (size_t)r;
    }
    return 0;
}

int run_job( mifsk_ctx *ctx, Job &job )
{
    const size_t nrows = job.rows.size();
    if ( nrows == 0 )
	return 0;
    HIP_OK(hipSetDevice(ctx->device));
    HostWork *w = nullptr;
    int rc = host_work_get(ctx, &w);
    if ( rc )
	return rc;
    std::lock_guard<std::mutex> g(w->lock);
    // MIFSK_CLI_TIMING=1: the phases of a job on stderr (what a batch of ONE pays: INTEGRATION.md 1b)
    const bool timing = std::getenv("MIFSK_CLI_TIMING") != nullptr;
    const double t_enter = now_s();
    const size_t esz = job.s16 ? 2 : 4;

    // chunks of whole streams, ~kChunkBytes of input each
    std::vector<Chunk> chunks;
    size_t max_bytes = 0, max_rows = 0, max_floats = 0;
    bool uniform_n = true;
    for ( size_t lo = 0; lo < nrows; ) {
	size_t hi = lo, maxn = 0;
	while ( hi < nrows ) {
	    const size_t m = std::max(maxn, (size_t)job.rows[hi].n);
	    const size_t stride = ( m + 7 ) & ~(size_t)7;
	    if ( hi > lo && ( hi - lo + 1 ) * stride * esz > kChunkBytes )
		break;
	    maxn = m;
	    hi++;
	}
	Chunk c;
	c.lo = lo; c.hi = hi;
	c.stride = std::max<size_t>(( maxn + 7 ) & ~(size_t)7, 8);
	chunks.push_back(c);
	max_bytes = std::max(max_bytes, ( hi - lo ) * c.stride * esz);
	max_floats = std::max(max_floats, ( hi - lo ) * c.stride);
	max_rows = std::max(max_rows, hi - lo);
	lo = hi;
    }
    for ( size_t i = 1; i < nrows; i++ )
	uniform_n = uniform_n && job.rows[i].n == job.rows[0].n;
    const bool single = chunks.size() == 1;
    rc = host_work_init(w, !single);
    if ( rc )
	return rc;
    const double t_begin = now_s();
    // (one chunk: everything in order on the null stream, which the runtime has anyway)
    const hipStream_t st_comp = single ? nullptr : w->s_comp;
    const hipStream_t st_in = single ? st_comp : w->s_in, st_out = single ? st_comp : w->s_out;

    // page-locked source rows are copied from where they are
    const bool direct = job.rows[0].mem && job.src_pitch && is_pinned(job.rows[0].mem)
		     && is_pinned((const char *)job.rows[nrows - 1].mem + (size_t)job.rows[nrows - 1].n * esz - ( job.rows[nrows - 1].n ? 1 : 0 ));
    rc = pin_reserve(w, direct ? 0 : max_bytes, max_rows);
    if ( rc )
	return rc;

    const mifsk_demod_io &ho = job.out;
    const size_t fc = ho.frames_cap, ec = ho.episodes_cap;
    SlotGuard sg;
    for ( Slot &s : sg.s ) {
	if ( dev_alloc((unsigned char **)&s.d_in, max_bytes + 64) ) return -ENOMEM;
	if ( job.s16 && dev_alloc(&s.d_x, max_floats + 16) ) return -ENOMEM;
	if ( dev_alloc(&s.d_n, max_rows) ) return -ENOMEM;
	if ( ho.d_bytes && dev_alloc(&s.d_bytes, max_rows * fc) ) return -ENOMEM;
	if ( ho.d_bits && dev_alloc(&s.d_bits, max_rows * fc) ) return -ENOMEM;
	if ( ho.d_frames && dev_alloc(&s.d_frames, max_rows * fc) ) return -ENOMEM;
	if ( ho.d_episodes && dev_alloc(&s.d_eps, max_rows * ec) ) return -ENOMEM;
	if ( ho.d_nbytes && dev_alloc(&s.d_nbytes, max_rows) ) return -ENOMEM;
	if ( ho.d_nframes && dev_alloc(&s.d_nframes, max_rows) ) return -ENOMEM;
	if ( ho.d_nepisodes && dev_alloc(&s.d_neps, max_rows) ) return -ENOMEM;
	if ( ho.d_status && dev_alloc(&s.d_status, max_rows) ) return -ENOMEM;
	if ( ho.d_carrier_band && dev_alloc(&s.d_band, max_rows) ) return -ENOMEM;
	if ( ho.d_counters && dev_alloc(&s.d_cnt, max_rows * MIFSK_NCOUNTERS) ) return -ENOMEM;
    }

    const double t_alloc = now_s();
    const unsigned nthreads = staging_threads();
    // (a test hook like every other knob: honoured only with MIFSK_EXPERIMENT set, so that a stray
    // variable cannot turn production reads into rows of zeros)
    const char *fault_tag = experiment_env("MIFSK_TEST_FAULT_READ");
    if ( fault_tag && !*fault_tag )
	fault_tag = nullptr;
    double t_stage = 0.0;
    uint64_t bytes_in = 0, bytes_out = 0;

    auto copy_out = [&]( size_t ci ) -> int {		// results of chunk ci -> host, on s_out
	const Chunk &c = chunks[ci];
	const Slot &s = sg.s[ci & 1];
	const size_t r = c.hi - c.lo;
	if ( !single )
	    HIP_OK(hipStreamWaitEvent(st_out, w->ev_comp[ci & 1], 0));
#define MIFSK_OUT(HOSTP, DEVP, PER_ROW)										\
	if ( HOSTP ) {												\
	    const size_t nb = r * (PER_ROW) * sizeof(*(HOSTP));							\
	    HIP_OK(hipMemcpyAsync((HOSTP) + c.lo * (PER_ROW), DEVP, nb, hipMemcpyDeviceToHost, st_out));	\
	    bytes_out += nb;											\
	}
	MIFSK_OUT(ho.d_bytes, s.d_bytes, fc)
	MIFSK_OUT(ho.d_bits, s.d_bits, fc)
	MIFSK_OUT(ho.d_frames, s.d_frames, fc)
	MIFSK_OUT(ho.d_episodes, s.d_eps, ec)
	MIFSK_OUT(ho.d_nbytes, s.d_nbytes, 1)
	MIFSK_OUT(ho.d_nframes, s.d_nframes, 1)
	MIFSK_OUT(ho.d_nepisodes, s.d_neps, 1)
	MIFSK_OUT(ho.d_status, s.d_status, 1)
	MIFSK_OUT(ho.d_counters, s.d_cnt, MIFSK_NCOUNTERS)
	if ( ho.d_carrier_band && job.cfg->auto_carrier_threshold > 0.0f ) {
	    MIFSK_OUT(ho.d_carrier_band, s.d_band, 1)
	}
#undef MIFSK_OUT
	if ( !single )
	    HIP_OK(hipEventRecord(w->ev_out[ci & 1], st_out));
	return 0;
    };

    for ( size_t ci = 0; ci < chunks.size(); ci++ ) {
	const Chunk &c = chunks[ci];
	const int sl = (int)( ci & 1 );
	Slot &s = sg.s[sl];
	const size_t r = c.hi - c.lo;
	if ( ci >= 2 )
	    HIP_OK(hipEventSynchronize(w->ev_out[sl]));	// chunk ci - 2 has left this slot
	// ---- stage
	const void *src = nullptr;
	size_t src_pitch_bytes = 0;
	if ( direct ) {
	    src = job.rows[c.lo].mem;
	    src_pitch_bytes = job.src_pitch * esz;
	} else {
	    const double t0 = now_s();
	    unsigned char *dst = (unsigned char *)w->pin[sl];
	    parallel_for(r, nthreads, [&]( size_t i ) {
		const Row &row = job.rows[c.lo + i];
		unsigned char *d = dst + i * c.stride * esz;
		if ( row.mem ) {
		    std::memcpy(d, row.mem, (size_t)row.n * esz);
		} else if ( row.n ) {
		    const int fd = open(row.path, O_RDONLY | O_CLOEXEC);
		    int e = fd < 0 ? -errno : read_fully(fd, d, (size_t)row.n * esz, row.off);
		    // (fault injection for tests/test_gpu_files.py: a file that shrinks between
		    // its header and its samples cannot be staged without a race)
		    if ( fault_tag && std::strstr(row.path, fault_tag) )
			e = -EIO;
		    if ( fd >= 0 ) close(fd);
		    if ( e ) {
			// this file's error (a file that shrank after its header was read, a
			// read error): reported through its own row, the batch goes on with
			// a row of zeros in its place
			std::memset(d, 0, (size_t)row.n * esz);
			if ( row.err ) *row.err = e;
		    }
		}
	    });
	    t_stage += now_s() - t0;
	    src = dst;
	    src_pitch_bytes = c.stride * esz;
	}
	for ( size_t i = 0; i < r; i++ )
	    w->pin_n[sl][i] = job.rows[c.lo + i].n;
	// ---- host -> device
	size_t width = std::min(src_pitch_bytes, c.stride * esz);
	if ( direct && nrows == 1 ) {
	    // a lone row's stride means nothing (its length is not clipped to it either)
	    width = (size_t)job.rows[0].n * esz;
	    src_pitch_bytes = c.stride * esz;
	}
	if ( src_pitch_bytes == c.stride * esz )	// rows back to back on both sides: one linear copy
	    HIP_OK(hipMemcpyAsync(s.d_in, src, width * r, hipMemcpyHostToDevice, st_in));
	else
	    HIP_OK(hipMemcpy2DAsync(s.d_in, c.stride * esz, src, src_pitch_bytes, width, r,
				    hipMemcpyHostToDevice, st_in));
	HIP_OK(hipMemcpyAsync(s.d_n, w->pin_n[sl], r * sizeof(uint32_t), hipMemcpyHostToDevice, st_in));
	bytes_in += width * r;
	// ---- convert + receive loop
	if ( !single ) {
	    HIP_OK(hipEventRecord(w->ev_in[sl], st_in));
	    HIP_OK(hipStreamWaitEvent(st_comp, w->ev_in[sl], 0));
	}
	const float *d_x = (const float *)s.d_in;
	if ( job.s16 ) {
	    rc = mifsk_ingest_s16(ctx, (const int16_t *)s.d_in, c.stride, s.d_x, c.stride, s.d_n, 0,
				  (int)r, job.rxnoise, st_comp);
	    d_x = s.d_x;
	} else if ( job.rxnoise != 0.0f ) {
	    rc = mifsk_ingest_rxnoise_f32(ctx, (float *)s.d_in, c.stride, s.d_n, 0, (int)r, job.rxnoise, st_comp);
	}
	if ( rc )
	    return rc;
	mifsk_demod_io io;
	std::memset(&io, 0, sizeof(io));
	io.d_samples = d_x;
	io.stream_stride = c.stride;
	io.d_nsamples = uniform_n ? nullptr : s.d_n;
	io.nsamples = job.rows[c.lo].n;
	io.nstreams = (int)r;
	io.d_bytes = s.d_bytes;	io.d_nbytes = s.d_nbytes;
	io.d_bits = s.d_bits;	io.d_frames = s.d_frames;	io.d_nframes = s.d_nframes;
	io.frames_cap = fc;
	io.d_episodes = s.d_eps;	io.d_nepisodes = s.d_neps;	io.episodes_cap = ec;
	io.d_status = s.d_status;	io.d_counters = s.d_cnt;	io.d_carrier_band = s.d_band;
	io.flags = job.flags;
	rc = mifsk_demod_batch(ctx, job.cfg, &io, st_comp);
	if ( rc )
	    return rc;
	if ( !single )
	    HIP_OK(hipEventRecord(w->ev_comp[sl], st_comp));
	// ---- the chunk before this one: results -> host (after this chunk's work is queued,
	// so that a copy into pageable memory, which blocks this thread, hides behind it)
	if ( ci >= 1 ) {
	    rc = copy_out(ci - 1);
	    if ( rc )
		return rc;
	}
    }
    rc = copy_out(chunks.size() - 1);
    if ( rc )
	return rc;
    const double t_queued = now_s();
    if ( !single )
	HIP_OK(hipStreamSynchronize(w->s_out));
    HIP_OK(hipStreamSynchronize(st_comp));
    if ( !single )
	HIP_OK(hipStreamSynchronize(w->s_in));
    if ( timing )
	std::fprintf(stderr, "### TIMING job of %zu rows: streams + events %.1f ms, pinned + device buffers %.1f ms, "
			     "read + queue (incl. the first launch: code object, tables) %.1f ms, device until done %.1f ms\n",
		     nrows, 1e3 * ( t_begin - t_enter ), 1e3 * ( t_alloc - t_begin ), 1e3 * ( t_queued - t_alloc ),
		     1e3 * ( now_s() - t_queued ));
    if ( job.stats ) {
	mifsk_host_stats &st = *job.stats;
	st.seconds_total += now_s() - t_begin;
	st.seconds_staging += t_stage;
	st.bytes_h2d += bytes_in;
	st.bytes_d2h += bytes_out;
	st.chunks += (uint32_t)chunks.size();
	st.streams += (uint32_t)nrows;
	st.source_pinned = direct ? 1u : 0u;
    }
    return 0;		// (what a row could not read is in its own error slot)
}

} // namespace
} // namespace mifsk

using mifsk::Job;
using mifsk::Row;

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------

extern "C" void *mifsk_host_alloc( size_t bytes )
{
    void *p = nullptr;
    if ( hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess ) {
	(void)hipGetLastError();
	return nullptr;
    }
    return p;
}

extern "C" void mifsk_host_free( void *p )
{
    if ( p )
	(void)hipHostFree(p);
}

extern "C" int mifsk_demod_batch_host_ex( mifsk_ctx *ctx, const mifsk_rx_config *cfg,
	const mifsk_demod_io *hio, float rxnoise, mifsk_host_stats *stats )
{
    if ( !ctx || !hio || mifsk_check_cfg(cfg) || hio->nstreams < 0 )
	return -EINVAL;
    if ( stats )
	std::memset(stats, 0, sizeof(*stats));
    const size_t ns = (size_t)hio->nstreams;
    if ( ns == 0 )
	return 0;
    if ( !hio->d_samples )
	return -EINVAL;
    try {		// (no exception crosses the C ABI)
    Job job;
    job.cfg = cfg;
    job.s16 = ( hio->flags & MIFSK_IO_HOST_S16 ) != 0;
    job.rxnoise = rxnoise;
    job.flags = hio->flags & ~MIFSK_IO_HOST_S16;
    job.src_pitch = hio->stream_stride;
    job.out = *hio;
    job.stats = stats;
    const size_t esz = job.s16 ? 2 : 4;
    job.rows.resize(ns);
    for ( size_t i = 0; i < ns; i++ ) {
	Row &r = job.rows[i];
	r.mem = (const char *)hio->d_samples + i * hio->stream_stride * esz;
	r.path = nullptr;
	r.off = 0;
	uint32_t n = hio->d_nsamples ? hio->d_nsamples[i] : hio->nsamples;
	if ( ns > 1 && (size_t)n > hio->stream_stride )
	    n = (uint32_t)hio->stream_stride;		// never trust a length beyond the row
	r.n = n;
	r.err = nullptr;
    }
    return mifsk::run_job(ctx, job);
    } catch ( const std::bad_alloc & ) {
	return -ENOMEM;
    } catch ( ... ) {
	return -EIO;
    }
}

extern "C" int mifsk_demod_batch_host( mifsk_ctx *ctx, const mifsk_rx_config *cfg,
	const mifsk_demod_io *hio )
{
    return mifsk_demod_batch_host_ex(ctx, cfg, hio, 0.0f, nullptr);
}

// ---------------------------------------------------------------------------
// a list of files -> one batch (what `minimodem --rx --file` does for one)
// ---------------------------------------------------------------------------

struct mifsk_files {
    struct Group {				// files of one (sample rate, sample format)
	mifsk_rx_config			cfg;
	std::vector<int>		members;	// indices into `files`
	size_t				fcap = 0, ecap = 0;
	std::vector<uint8_t>		bytes;
	std::vector<uint64_t>		bits;
	std::vector<mifsk_frame>	frames;
	std::vector<mifsk_episode>	eps;
	std::vector<uint32_t>		nbytes, nframes, neps, status;
	std::vector<int32_t>		band;
    };
    std::vector<mifsk_file_result>	files;
    std::vector<Group>			groups;
    std::vector<std::string>		paths;
    mifsk_host_stats			stats;
};

extern "C" size_t mifsk_max_episodes( const mifsk_rx_config *cfg, size_t nsamples )
{
    if ( !cfg )
	return 0;
    // an episode is at least one frame followed by 21 searches without confidence,
    // each of which moves the cursor by the search range (minimodem.c:1292-1321,1407)
    const size_t adv = cfg->frame_nsamples > cfg->nsamples_overscan ? cfg->frame_nsamples - cfg->nsamples_overscan : 1;
    const size_t tm = std::min(cfg->try_max[0], cfg->try_max[1]);
    return nsamples / ( adv + 21 * ( tm ? tm : 1 ) ) + 2;
}

static int demod_files_impl( mifsk_ctx *ctx, const mifsk_modem_args *args,
	const char *const *paths, int nfiles, float rxnoise, unsigned flags, mifsk_files *F )
{
    std::memset(&F->stats, 0, sizeof(F->stats));
    F->files.resize((size_t)nfiles);
    F->paths.resize((size_t)nfiles);
    const double t0 = mifsk::now_s();
    // ---- headers (threads): format, rate, where the samples start, how many
    for ( int i = 0; i < nfiles; i++ ) {
	std::memset(&F->files[(size_t)i], 0, sizeof(mifsk_file_result));
	F->paths[(size_t)i] = paths[i] ? paths[i] : "";
	F->files[(size_t)i].carrier_band = -1;
    }
    mifsk::parallel_for((size_t)nfiles, mifsk::staging_threads(), [&]( size_t i ) {
	mifsk_file_result &fr = F->files[i];
	const int fd = open(F->paths[i].c_str(), O_RDONLY | O_CLOEXEC);
	if ( fd < 0 ) {
	    fr.error = -errno;
	    return;
	}
	struct stat st;
	if ( fstat(fd, &st) != 0 ) {
	    fr.error = -errno;
	    close(fd);
	    return;
	}
	// (the data chunk normally starts at byte 44; LIST / fact chunks in front of it are
	// rarely more than a few hundred bytes)
	std::vector<unsigned char> head(65536);
	const ssize_t got = pread(fd, head.data(), head.size(), 0);
	close(fd);
	if ( got < 0 ) {
	    fr.error = -errno;
	    return;
	}
	fr.error = mifsk::wav_parse_sized(head.data(), (size_t)got, (size_t)st.st_size, &fr.info);
	if ( !fr.error && fr.info.nframes > 0xFFFFFF00ull )
	    fr.error = -EFBIG;			// stream lengths are 32-bit on the device
    });
    // ---- one batch per (sample rate, sample format): the reference takes its sample rate
    // from the file and derives everything from it (minimodem.c:1021-1032)
    std::map<std::pair<unsigned, int>, size_t> index;
    for ( int i = 0; i < nfiles; i++ ) {
	const mifsk_file_result &fr = F->files[(size_t)i];
	if ( fr.error )
	    continue;
	const std::pair<unsigned, int> key(fr.info.sample_rate, fr.info.is_float);
	auto it = index.find(key);
	if ( it == index.end() ) {
	    mifsk_files::Group g;
	    mifsk_modem_args a = *args;
	    a.sample_rate = fr.info.sample_rate;
	    const int rc = mifsk_rx_config_init(&g.cfg, &a);
	    if ( rc ) {
		F->files[(size_t)i].error = rc;		// (e.g. tones above this file's Nyquist rate)
		continue;
	    }
	    F->groups.push_back(std::move(g));
	    it = index.emplace(key, F->groups.size() - 1).first;
	}
	F->groups[it->second].members.push_back(i);
    }
    // ---- length classes.  A batch's output arrays (host vectors, device slots, the copies
    // back) are sized by its LONGEST file, so one hour-long recording among ten thousand short
    // ones would cost every one of them the long one's capacity.  Each (rate, format) group is
    // therefore cut, by length, into classes whose longest file is at most twice the shortest:
    // a class is one batch with its own capacities, at most 2 x what its files need.
    {
	std::vector<mifsk_files::Group> classes;
	for ( mifsk_files::Group &g : F->groups ) {
	    std::stable_sort(g.members.begin(), g.members.end(), [&]( int a, int b ) {
		return F->files[(size_t)a].info.nframes < F->files[(size_t)b].info.nframes;
	    });
	    size_t lo = 0;
	    while ( lo < g.members.size() ) {
		const uint64_t shortest = std::max<uint64_t>(F->files[(size_t)g.members[lo]].info.nframes, 4096);
		size_t hi = lo;
		while ( hi < g.members.size() && F->files[(size_t)g.members[hi]].info.nframes <= 2 * shortest )
		    hi++;
		mifsk_files::Group c;
		c.cfg = g.cfg;
		c.members.assign(g.members.begin() + (long)lo, g.members.begin() + (long)hi);
		classes.push_back(std::move(c));
		lo = hi;
	    }
	}
	F->groups.swap(classes);
    }
    int rc_all = 0;
    for ( mifsk_files::Group &g : F->groups ) {
	const size_t n = g.members.size();
	size_t maxn = 0;
	for ( int i : g.members )
	    maxn = std::max(maxn, F->files[(size_t)i].info.nframes);
	g.fcap = mifsk_max_frames(&g.cfg, maxn);
	g.ecap = mifsk_max_episodes(&g.cfg, maxn);
	g.bits.assign(n * g.fcap, 0);
	g.bytes.assign(n * g.fcap, 0);
	if ( flags & MIFSK_FILES_WANT_FRAMES )
	    g.frames.resize(n * g.fcap);
	g.eps.resize(n * g.ecap);
	g.nbytes.assign(n, 0); g.nframes.assign(n, 0); g.neps.assign(n, 0); g.status.assign(n, 0);
	g.band.assign(n, -1);
	Job job;
	job.cfg = &g.cfg;
	job.s16 = !F->files[(size_t)g.members[0]].info.is_float;
	job.rxnoise = rxnoise;
	job.flags = flags & ( MIFSK_IO_RING_EXACT | MIFSK_IO_ENGINE_WAVE | MIFSK_IO_ENGINE_WORKGROUP );
	job.src_pitch = 0;
	job.stats = &F->stats;
	std::memset(&job.out, 0, sizeof(job.out));
	job.out.d_bytes = g.bytes.data();	job.out.d_nbytes = g.nbytes.data();
	job.out.d_bits = g.bits.data();
	job.out.d_frames = g.frames.empty() ? nullptr : g.frames.data();
	job.out.d_nframes = g.nframes.data();	job.out.frames_cap = g.fcap;
	job.out.d_episodes = g.eps.data();	job.out.d_nepisodes = g.neps.data();	job.out.episodes_cap = g.ecap;
	job.out.d_status = g.status.data();
	job.out.d_carrier_band = g.band.data();
	job.rows.resize(n);
	for ( size_t k = 0; k < n; k++ ) {
	    mifsk_file_result &fr = F->files[(size_t)g.members[k]];
	    Row &r = job.rows[k];
	    r.mem = nullptr;
	    r.path = F->paths[(size_t)g.members[k]].c_str();
	    r.off = fr.info.data_offset;
	    r.n = (uint32_t)fr.info.nframes;
	    r.err = &fr.error;
	}
	const int rc = mifsk::run_job(ctx, job);
	if ( rc && !rc_all )
	    rc_all = rc;
	for ( size_t k = 0; k < n; k++ ) {
	    mifsk_file_result &fr = F->files[(size_t)g.members[k]];
	    fr.cfg = &g.cfg;
	    if ( fr.error )
		continue;			// (could not be read after all: its row was zeros)
	    fr.nframes = g.nframes[k] < g.fcap ? g.nframes[k] : (uint32_t)g.fcap;
	    fr.nbytes = g.nbytes[k] < g.fcap ? g.nbytes[k] : (uint32_t)g.fcap;
	    fr.nepisodes = g.neps[k] < g.ecap ? g.neps[k] : (uint32_t)g.ecap;
	    fr.status = g.status[k];
	    fr.carrier_band = g.band[k];
	    fr.bits = g.bits.data() + k * g.fcap;
	    fr.bytes = g.bytes.data() + k * g.fcap;
	    fr.frames = g.frames.empty() ? nullptr : g.frames.data() + k * g.fcap;
	    fr.episodes = g.eps.data() + k * g.ecap;
	}
    }
    F->stats.seconds_total = mifsk::now_s() - t0;	// headers and grouping included
    // a file that could not be read is that file's error (mifsk_file_result.error), not the
    // batch's: run_job reports only what stopped the pipeline (HIP, allocation)
    return rc_all;
}

extern "C" int mifsk_demod_files( mifsk_ctx *ctx, const mifsk_modem_args *args,
	const char *const *paths, int nfiles, float rxnoise, unsigned flags, mifsk_files **out )
{
    if ( !ctx || !args || !out || nfiles < 0 || ( nfiles && !paths ) )
	return -EINVAL;
    *out = nullptr;
    mifsk_files *F = new (std::nothrow) mifsk_files();
    if ( !F )
	return -ENOMEM;
    int rc;
    try {		// (no exception crosses the C ABI: the vectors above can throw)
	rc = demod_files_impl(ctx, args, paths, nfiles, rxnoise, flags, F);
    } catch ( const std::bad_alloc & ) {
	delete F;
	return -ENOMEM;
    } catch ( ... ) {
	delete F;
	return -EIO;
    }
    *out = F;
    return rc;
}

extern "C" int mifsk_files_count( const mifsk_files *f ) { return f ? (int)f->files.size() : 0; }

extern "C" const mifsk_file_result *mifsk_files_get( const mifsk_files *f, int i )
{
    return ( f && i >= 0 && (size_t)i < f->files.size() ) ? &f->files[(size_t)i] : nullptr;
}

extern "C" const mifsk_host_stats *mifsk_files_stats( const mifsk_files *f ) { return f ? &f->stats : nullptr; }

extern "C" void mifsk_files_free( mifsk_files *f ) { delete f; }
