// mifsk_session.cpp -- a batch of streams that arrive in pieces, fed from host memory
// (include/mifsk.h "streams fed in pieces from host memory").
//
// The reference reads its one stream half a samplebuf at a time and never holds more
// (src/minimodem.c:1144-1174): a recording longer than memory, or live audio, is its normal case.
// mifsk_demod_slab is that for a batch on the device -- state in, state out -- but leaves its
// caller the bookkeeping: which samples the loop has not passed yet, where each row starts in its
// stream, output arrays of the right size, the copies.  A session owns all of that: feed() takes
// each stream's NEW samples (any amount, also none), puts them behind the stream's unconsumed
// tail, runs the loop as far as the data allows and hands back what that made.  Whatever the
// cuts, the concatenated results are those of one call over the whole streams, bit for bit
// (tests/test_gpu_session.py).
#include <hip/hip_runtime.h>

#include <cerrno>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "mifsk.h"
#include "mifsk_ctx.h"

namespace {

template <class T>
struct DevBuf {
    T		*p = nullptr;
    size_t	cap = 0;		// elements
    int fit( size_t n )
    {
	if ( n <= cap )
	    return 0;
	if ( p ) (void)hipFree(p);
	p = nullptr;
	cap = 0;
	const size_t want = n + n / 4;
	if ( hipMalloc((void **)&p, want * sizeof(T)) != hipSuccess )
	    return -ENOMEM;
	cap = want;
	return 0;
    }
    void drop() { if ( p ) (void)hipFree(p); p = nullptr; cap = 0; }
};

template <class T>
struct PinBuf {
    T		*p = nullptr;
    size_t	cap = 0;
    int fit( size_t n )
    {
	if ( n <= cap )
	    return 0;
	if ( p ) (void)hipHostFree(p);
	p = nullptr;
	cap = 0;
	const size_t want = n + n / 4;
	if ( hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault) != hipSuccess )
	    return -ENOMEM;
	cap = want;
	return 0;
    }
    void drop() { if ( p ) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

} // namespace

struct mifsk_session {
    mifsk_ctx		*ctx = nullptr;
    mifsk_rx_config	cfg;
    int			n = 0;
    unsigned		flags = 0;
    bool		want_frames = false, ring = false, finished = false;
    hipStream_t		stream = nullptr;
    // per stream: what the loop has not passed yet, and where that starts in the stream
    std::vector<std::vector<float>>	tail;
    std::vector<uint64_t>		origin;
    // device: loop state, RING cells, the rows of a feed and its outputs
    DevBuf<mifsk_stream_state>	d_state;
    DevBuf<float>		d_ring;
    DevBuf<float>		d_rows;
    DevBuf<uint32_t>		d_lens, d_counts;	// counts: nframes | nbytes | nepisodes | status | carrier_band
    DevBuf<uint64_t>		d_origin, d_bits;
    DevBuf<uint8_t>		d_bytes;
    DevBuf<mifsk_frame>		d_frames;
    DevBuf<mifsk_episode>	d_eps;
    // host (page-locked): staging of the rows, the results of the last feed
    PinBuf<float>		h_rows;
    PinBuf<uint32_t>		h_lens, h_counts;
    PinBuf<uint64_t>		h_origin, h_bits;
    PinBuf<uint8_t>		h_bytes;
    PinBuf<mifsk_frame>		h_frames;
    PinBuf<mifsk_episode>	h_eps;
    PinBuf<mifsk_stream_state>	h_state;
    size_t			fc = 0, ec = 0;		// capacities of the last feed's arrays
    std::vector<mifsk_session_result>	results;
};

extern "C" void mifsk_session_destroy( mifsk_session *s )
{
    if ( !s )
	return;
    if ( s->ctx )
	(void)hipSetDevice(s->ctx->device);
    if ( s->stream ) {
	(void)hipStreamSynchronize(s->stream);
	(void)hipStreamDestroy(s->stream);
    }
    s->d_state.drop(); s->d_ring.drop(); s->d_rows.drop(); s->d_lens.drop(); s->d_counts.drop();
    s->d_origin.drop(); s->d_bits.drop(); s->d_bytes.drop(); s->d_frames.drop(); s->d_eps.drop();
    s->h_rows.drop(); s->h_lens.drop(); s->h_counts.drop(); s->h_origin.drop(); s->h_bits.drop();
    s->h_bytes.drop(); s->h_frames.drop(); s->h_eps.drop(); s->h_state.drop();
    delete s;
}

extern "C" int mifsk_session_create( mifsk_session **out, mifsk_ctx *ctx, const mifsk_rx_config *cfg,
	int nstreams, unsigned flags )
{
    if ( !out )
	return -EINVAL;
    *out = nullptr;
    if ( !ctx || !cfg || nstreams <= 0 )
	return -EINVAL;
    if ( flags & ~( MIFSK_IO_RING_EXACT | MIFSK_IO_ENGINE_WAVE | MIFSK_IO_ENGINE_WORKGROUP | MIFSK_SESSION_WANT_FRAMES ) )
	return -EINVAL;
    if ( ( flags & MIFSK_IO_ENGINE_WAVE ) && ( flags & MIFSK_IO_ENGINE_WORKGROUP ) )
	return -EINVAL;
    if ( ( flags & MIFSK_IO_RING_EXACT ) && ( flags & MIFSK_IO_ENGINE_WORKGROUP ) )
	return -EINVAL;				// (RING addressing is the wavefront engine's)
    int rc = mifsk_check_cfg(cfg);
    if ( rc != 0 )
	return rc;
    mifsk_session *s = new (std::nothrow) mifsk_session;
    if ( !s )
	return -ENOMEM;
    s->ctx = ctx;
    s->cfg = *cfg;
    s->n = nstreams;
    s->flags = flags & ( MIFSK_IO_ENGINE_WAVE | MIFSK_IO_ENGINE_WORKGROUP );
    s->want_frames = ( flags & MIFSK_SESSION_WANT_FRAMES ) != 0;
    s->ring = ( flags & MIFSK_IO_RING_EXACT ) != 0;
    try {
	s->tail.resize((size_t)nstreams);
	s->origin.assign((size_t)nstreams, 0);
	s->results.resize((size_t)nstreams);
    } catch ( const std::bad_alloc & ) {
	delete s;
	return -ENOMEM;
    }
    rc = -EIO;
    if ( hipSetDevice(ctx->device) == hipSuccess
	    && hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) == hipSuccess ) {
	rc = s->d_state.fit((size_t)nstreams);
	if ( rc == 0 && hipMemsetAsync(s->d_state.p, 0, (size_t)nstreams * sizeof(mifsk_stream_state), s->stream) != hipSuccess )
	    rc = -EIO;				// (all zero: a new stream)
	if ( rc == 0 && s->ring ) {
	    const size_t nf = mifsk_ring_floats(cfg) * (size_t)nstreams;
	    rc = s->d_ring.fit(nf);
	    if ( rc == 0 && hipMemsetAsync(s->d_ring.p, 0, nf * sizeof(float), s->stream) != hipSuccess )
		rc = -EIO;
	}
    }
    if ( rc != 0 ) {
	mifsk_session_destroy(s);
	return rc;
    }
    *out = s;
    return 0;
}

extern "C" int mifsk_session_feed( mifsk_session *s, const float *const *samples, const uint32_t *nsamples, int final )
{
    if ( !s )
	return -EINVAL;
    if ( s->finished )
	return -EINVAL;				// the final piece has been fed
    const size_t n = (size_t)s->n;
    HIP_OK(hipSetDevice(s->ctx->device));
    size_t width = 4;
    try {
	for ( size_t i = 0; i < n; i++ ) {
	    const uint32_t k = nsamples ? nsamples[i] : 0u;
	    if ( k && ( !samples || !samples[i] ) )
		return -EINVAL;
	    if ( k )
		s->tail[i].insert(s->tail[i].end(), samples[i], samples[i] + k);
	    if ( s->tail[i].size() > 0xFFFFFFF0ull )
		return -EOVERFLOW;
	    width = s->tail[i].size() > width ? s->tail[i].size() : width;
	}
    } catch ( const std::bad_alloc & ) {
	return -ENOMEM;
    }
    width = ( width + 3 ) & ~(size_t)3;
    const size_t fc = mifsk_max_frames(&s->cfg, width), ec = mifsk_max_episodes(&s->cfg, width);
    int rc = 0;
    if ( ( rc = s->h_rows.fit(n * width) ) || ( rc = s->d_rows.fit(n * width) )
	    || ( rc = s->h_lens.fit(n) ) || ( rc = s->d_lens.fit(n) )
	    || ( rc = s->h_origin.fit(n) ) || ( rc = s->d_origin.fit(n) )
	    || ( rc = s->h_counts.fit(5 * n) ) || ( rc = s->d_counts.fit(5 * n) )
	    || ( rc = s->h_state.fit(n) )
	    || ( rc = s->d_bits.fit(n * fc) ) || ( rc = s->h_bits.fit(n * fc) )
	    || ( rc = s->d_bytes.fit(n * fc) ) || ( rc = s->h_bytes.fit(n * fc) )
	    || ( rc = s->d_eps.fit(n * ec) ) || ( rc = s->h_eps.fit(n * ec) ) )
	return rc;
    if ( s->want_frames && ( ( rc = s->d_frames.fit(n * fc) ) || ( rc = s->h_frames.fit(n * fc) ) ) )
	return rc;
    for ( size_t i = 0; i < n; i++ ) {
	float *row = s->h_rows.p + i * width;
	const size_t k = s->tail[i].size();
	if ( k )
	    std::memcpy(row, s->tail[i].data(), k * sizeof(float));
	std::memset(row + k, 0, ( width - k ) * sizeof(float));
	s->h_lens.p[i] = (uint32_t)k;
	s->h_origin.p[i] = s->origin[i];
    }
    hipStream_t st = s->stream;
    HIP_OK(hipMemcpyAsync(s->d_rows.p, s->h_rows.p, n * width * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(s->d_lens.p, s->h_lens.p, n * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(s->d_origin.p, s->h_origin.p, n * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    HIP_OK(hipMemsetAsync(s->d_counts.p, 0, 4 * n * sizeof(uint32_t), st));
    HIP_OK(hipMemsetAsync(s->d_counts.p + 4 * n, 0xFF, n * sizeof(uint32_t), st));	// carrier_band: -1

    mifsk_demod_io io;
    std::memset(&io, 0, sizeof(io));
    io.d_samples = s->d_rows.p;
    io.stream_stride = width;
    io.d_nsamples = s->d_lens.p;
    io.nsamples = (uint32_t)width;
    io.nstreams = s->n;
    io.d_bytes = s->d_bytes.p;
    io.d_bits = s->d_bits.p;
    io.d_frames = s->want_frames ? s->d_frames.p : nullptr;
    io.frames_cap = fc;
    io.d_episodes = s->d_eps.p;
    io.episodes_cap = ec;
    io.d_nframes = s->d_counts.p;
    io.d_nbytes = s->d_counts.p + n;
    io.d_nepisodes = s->d_counts.p + 2 * n;
    io.d_status = s->d_counts.p + 3 * n;
    io.d_carrier_band = reinterpret_cast<int32_t *>(s->d_counts.p + 4 * n);
    io.flags = s->flags | ( s->ring ? MIFSK_IO_RING_EXACT : 0u );
    rc = s->ring ? mifsk_demod_slab_ring(s->ctx, &s->cfg, &io, s->d_state.p, s->d_origin.p, s->d_ring.p, final ? 1 : 0, st)
		 : mifsk_demod_slab(s->ctx, &s->cfg, &io, s->d_state.p, s->d_origin.p, final ? 1 : 0, st);
    if ( rc != 0 )
	return rc;
    HIP_OK(hipMemcpyAsync(s->h_counts.p, s->d_counts.p, 5 * n * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(s->h_state.p, s->d_state.p, n * sizeof(mifsk_stream_state), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    // the columns that hold something (counts keep counting past the capacity)
    size_t mf = 0, mb = 0, me = 0;
    for ( size_t i = 0; i < n; i++ ) {
	const size_t nf = s->h_counts.p[i], nb = s->h_counts.p[n + i], ne = s->h_counts.p[2 * n + i];
	mf = nf > mf ? nf : mf;
	mb = nb > mb ? nb : mb;
	me = ne > me ? ne : me;
    }
    mf = mf < fc ? mf : fc;
    mb = mb < fc ? mb : fc;
    me = me < ec ? me : ec;
    if ( mf )
	HIP_OK(hipMemcpy2DAsync(s->h_bits.p, fc * sizeof(uint64_t), s->d_bits.p, fc * sizeof(uint64_t),
				mf * sizeof(uint64_t), n, hipMemcpyDeviceToHost, st));
    if ( mf && s->want_frames )
	HIP_OK(hipMemcpy2DAsync(s->h_frames.p, fc * sizeof(mifsk_frame), s->d_frames.p, fc * sizeof(mifsk_frame),
				mf * sizeof(mifsk_frame), n, hipMemcpyDeviceToHost, st));
    if ( mb )
	HIP_OK(hipMemcpy2DAsync(s->h_bytes.p, fc, s->d_bytes.p, fc, mb, n, hipMemcpyDeviceToHost, st));
    if ( me )
	HIP_OK(hipMemcpy2DAsync(s->h_eps.p, ec * sizeof(mifsk_episode), s->d_eps.p, ec * sizeof(mifsk_episode),
				me * sizeof(mifsk_episode), n, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    s->fc = fc;
    s->ec = ec;
    for ( size_t i = 0; i < n; i++ ) {
	mifsk_session_result &r = s->results[i];
	std::memset(&r, 0, sizeof(r));
	r.nframes = s->h_counts.p[i];
	r.nbytes = s->h_counts.p[n + i];
	r.nepisodes = s->h_counts.p[2 * n + i];
	r.status = s->h_counts.p[3 * n + i];
	r.carrier_band = (int32_t)s->h_counts.p[4 * n + i];
	if ( r.nframes > fc ) r.nframes = (uint32_t)fc;		// (what the arrays hold; status says it was cut)
	if ( r.nbytes > fc ) r.nbytes = (uint32_t)fc;
	if ( r.nepisodes > ec ) r.nepisodes = (uint32_t)ec;
	r.bits = s->h_bits.p + i * fc;
	r.bytes = s->h_bytes.p + i * fc;
	r.frames = s->want_frames ? s->h_frames.p + i * fc : nullptr;
	r.episodes = s->h_eps.p + i * ec;
	const mifsk_stream_state &ss = s->h_state.p[i];
	r.consumed = ss.base;
	r.finished = ( ss.flags & MIFSK_STATE_FINISHED ) ? 1u : 0u;
	// everything before the cursor has been passed for good
	if ( ss.base > s->origin[i] ) {
	    size_t drop = (size_t)( ss.base - s->origin[i] );
	    if ( drop > s->tail[i].size() )
		drop = s->tail[i].size();
	    s->tail[i].erase(s->tail[i].begin(), s->tail[i].begin() + (long)drop);
	    s->origin[i] += drop;
	}
    }
    if ( final )
	s->finished = true;
    return 0;
}

extern "C" const mifsk_session_result *mifsk_session_get( const mifsk_session *s, int stream )
{
    if ( !s || stream < 0 || stream >= s->n )
	return nullptr;
    return &s->results[(size_t)stream];
}

extern "C" size_t mifsk_session_pending( const mifsk_session *s, int stream )
{
    if ( !s || stream < 0 || stream >= s->n )
	return 0;
    return s->tail[(size_t)stream].size();
}
