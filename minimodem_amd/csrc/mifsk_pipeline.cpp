// mifsk_pipeline.cpp -- several batches in flight behind the C ABI (include/mifsk.h "passes in
// flight").
//
// A launch of mifsk_demod_batch ends well after its mean stream: the streams of a batch are
// serial chains of unequal length, and while the last ones finish most of the chip is idle
// (DESIGN.md section 6).  Nothing in the reference's call site -- one file after another,
// src/minimodem.c:1265,1373 through integration/minimodem-rx-batch.patch -- says batch i + 1 must
// wait for the last stream of batch i.  A pipeline owns what keeping P batches in flight needs:
// P lanes, each a device context (a context owns its launch scratch), a HIP stream and a
// completion event, and optionally P sets of output arrays; pass t runs on lane t mod P.
//
// HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless the
// variable says otherwise; read when the runtime starts) and the null stream takes one of them:
// lanes beyond that share a queue and run one after the other -- measured slower than fewer lanes
// (profiles/r05_history.md section 6) -- so the depth is clamped to what the queues carry, and
// the clamp is reported.
#include <hip/hip_runtime.h>

#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "mifsk.h"
#include "mifsk_ctx.h"

namespace {

struct OutSet {
    mifsk_demod_io	io;		// output fields only
    std::vector<void *>	blocks;		// what hipMalloc returned
};

struct Lane {
    mifsk_ctx	*ctx = nullptr;
    hipStream_t	stream = nullptr;
    hipEvent_t	done = nullptr;		// behind the last pass submitted on this lane
    hipEvent_t	after = nullptr;	// the producer's stream at submit time
    uint64_t	last = 0;		// ticket of that pass
    bool	used = false;
    OutSet	set;
};

void free_set( OutSet &s )
{
    for ( void *b : s.blocks )
	(void)hipFree(b);
    s.blocks.clear();
    std::memset(&s.io, 0, sizeof(s.io));
}

} // namespace

struct mifsk_pipeline {
    int			device = 0;
    uint32_t		requested = 0, depth = 0, hw_queues = 0;
    std::vector<Lane>	lanes;
    uint64_t		next = 0;
    bool		sets = false;
    std::mutex		lock;
};

extern "C" void mifsk_pipeline_destroy( mifsk_pipeline *p )
{
    if ( !p )
	return;
    (void)hipSetDevice(p->device);
    for ( Lane &l : p->lanes ) {
	if ( l.stream )
	    (void)hipStreamSynchronize(l.stream);
	free_set(l.set);
	if ( l.done ) (void)hipEventDestroy(l.done);
	if ( l.after ) (void)hipEventDestroy(l.after);
	if ( l.stream ) (void)hipStreamDestroy(l.stream);
	if ( l.ctx ) mifsk_ctx_destroy(l.ctx);
    }
    delete p;
}

extern "C" int mifsk_pipeline_create( mifsk_pipeline **out, int device, int depth )
{
    if ( !out )
	return -EINVAL;
    *out = nullptr;
    if ( depth < 1 )
	depth = 1;
    if ( depth > MIFSK_PIPELINE_MAX_DEPTH )
	depth = MIFSK_PIPELINE_MAX_DEPTH;
    mifsk_pipeline *p = new (std::nothrow) mifsk_pipeline;
    if ( !p )
	return -ENOMEM;
    p->requested = (uint32_t)depth;
    // what the runtime read when it started (a value set later has no effect on it)
    int hwq = 4;
    if ( const char *e = std::getenv("GPU_MAX_HW_QUEUES") ) {
	const int v = std::atoi(e);
	if ( v > 0 )
	    hwq = v;
    }
    p->hw_queues = (uint32_t)hwq;
    const int carry = hwq > 1 ? hwq - 1 : 1;	// the null stream takes a queue
    p->depth = (uint32_t)( depth < carry ? depth : carry );
    p->lanes.resize(p->depth);
    int rc = 0;
    for ( uint32_t i = 0; i < p->depth && rc == 0; i++ ) {
	Lane &l = p->lanes[i];
	rc = mifsk_ctx_create(&l.ctx, device);
	if ( rc != 0 )
	    break;
	if ( i == 0 )
	    p->device = mifsk::ctx_device(l.ctx);
	device = p->device;			// (device < 0: every lane on the one the first lane got)
	if ( hipSetDevice(p->device) != hipSuccess
		|| hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking) != hipSuccess
		|| hipEventCreateWithFlags(&l.done, hipEventDisableTiming) != hipSuccess
		|| hipEventCreateWithFlags(&l.after, hipEventDisableTiming) != hipSuccess )
	    rc = -EIO;
    }
    if ( rc != 0 ) {
	mifsk_pipeline_destroy(p);
	return rc;
    }
    *out = p;
    return 0;
}

extern "C" int mifsk_pipeline_info_get( const mifsk_pipeline *p, mifsk_pipeline_info *info )
{
    if ( !p || !info )
	return -EINVAL;
    info->depth_requested = p->requested;
    info->depth = p->depth;
    info->hw_queues = p->hw_queues;
    info->output_sets = p->sets ? p->depth : 0u;
    return 0;
}

extern "C" int mifsk_pipeline_outputs_alloc( mifsk_pipeline *p, int nstreams, size_t frames_cap,
	size_t episodes_cap, unsigned want )
{
    if ( !p || nstreams <= 0 || frames_cap == 0 )
	return -EINVAL;
    std::lock_guard<std::mutex> g(p->lock);
    if ( hipSetDevice(p->device) != hipSuccess )
	return -EIO;
    int rc = 0;
    for ( Lane &l : p->lanes ) {
	if ( l.stream )
	    (void)hipStreamSynchronize(l.stream);	// nothing may still write the old set
	free_set(l.set);
	auto take = [&]( size_t bytes ) -> void * {
	    void *d = nullptr;
	    if ( rc != 0 )
		return nullptr;
	    if ( hipMalloc(&d, bytes ? bytes : 1) != hipSuccess ) {
		rc = -ENOMEM;
		return nullptr;
	    }
	    l.set.blocks.push_back(d);
	    if ( hipMemsetAsync(d, 0, bytes, l.stream) != hipSuccess )
		rc = -EIO;
	    return d;
	};
	const size_t ns = (size_t)nstreams;
	mifsk_demod_io &o = l.set.io;
	o.frames_cap = frames_cap;
	o.episodes_cap = ( want & MIFSK_WANT_EPISODES ) ? ( episodes_cap ? episodes_cap : 1 ) : 0;
	o.d_nframes = (uint32_t *)take(ns * sizeof(uint32_t));
	o.d_nbytes = (uint32_t *)take(ns * sizeof(uint32_t));
	o.d_status = (uint32_t *)take(ns * sizeof(uint32_t));
	if ( want & MIFSK_WANT_BYTES )
	    o.d_bytes = (uint8_t *)take(ns * frames_cap);
	if ( want & MIFSK_WANT_BITS )
	    o.d_bits = (uint64_t *)take(ns * frames_cap * sizeof(uint64_t));
	if ( want & MIFSK_WANT_FRAMES )
	    o.d_frames = (mifsk_frame *)take(ns * frames_cap * sizeof(mifsk_frame));
	if ( want & MIFSK_WANT_EPISODES ) {
	    o.d_episodes = (mifsk_episode *)take(ns * o.episodes_cap * sizeof(mifsk_episode));
	    o.d_nepisodes = (uint32_t *)take(ns * sizeof(uint32_t));
	}
	o.nstreams = nstreams;
    }
    if ( rc != 0 ) {
	for ( Lane &l : p->lanes )
	    free_set(l.set);
	p->sets = false;
	return rc;
    }
    p->sets = true;
    return 0;
}

extern "C" int mifsk_pipeline_outputs_get( mifsk_pipeline *p, uint64_t ticket, mifsk_demod_io *io )
{
    if ( !p || !io || !p->sets )
	return -EINVAL;
    const mifsk_demod_io &o = p->lanes[ticket % p->depth].set.io;
    io->d_bytes = o.d_bytes;
    io->d_nbytes = o.d_nbytes;
    io->d_bits = o.d_bits;
    io->d_frames = o.d_frames;
    io->d_nframes = o.d_nframes;
    io->frames_cap = o.frames_cap;
    io->d_episodes = o.d_episodes;
    io->d_nepisodes = o.d_nepisodes;
    io->episodes_cap = o.episodes_cap;
    io->d_status = o.d_status;
    return 0;
}

extern "C" uint64_t mifsk_pipeline_next_ticket( const mifsk_pipeline *p )
{
    return p ? p->next : 0;
}

extern "C" int mifsk_pipeline_submit( mifsk_pipeline *p, const mifsk_rx_config *cfg,
	const mifsk_demod_io *io, void *after, uint64_t *ticket )
{
    if ( !p || !cfg || !io )
	return -EINVAL;
    std::lock_guard<std::mutex> g(p->lock);
    Lane &l = p->lanes[p->next % p->depth];
    if ( hipSetDevice(p->device) != hipSuccess )
	return -EIO;
    mifsk_demod_io use = *io;
    if ( p->sets && !io->d_bytes && !io->d_bits && !io->d_frames && !io->d_nframes && !io->d_nbytes ) {
	// no outputs given: the lane's own set (which must have been made for batches this size)
	if ( io->nstreams > l.set.io.nstreams )
	    return -EINVAL;
	const mifsk_demod_io &o = l.set.io;
	use.d_bytes = o.d_bytes;
	use.d_nbytes = o.d_nbytes;
	use.d_bits = o.d_bits;
	use.d_frames = o.d_frames;
	use.d_nframes = o.d_nframes;
	use.frames_cap = o.frames_cap;
	use.d_episodes = o.d_episodes;
	use.d_nepisodes = o.d_nepisodes;
	use.episodes_cap = o.episodes_cap;
	use.d_status = o.d_status;
    }
    if ( after != MIFSK_PIPELINE_NO_PRODUCER ) {
	// the batch was produced on the caller's stream: the lane waits for the point it has reached
	if ( hipEventRecord(l.after, (hipStream_t)after) != hipSuccess
		|| hipStreamWaitEvent(l.stream, l.after, 0) != hipSuccess )
	    return -EIO;
    }
    const int rc = mifsk_demod_batch(l.ctx, cfg, &use, l.stream);
    if ( rc != 0 )
	return rc;
    if ( hipEventRecord(l.done, l.stream) != hipSuccess )
	return -EIO;
    l.last = p->next;
    l.used = true;
    if ( ticket )
	*ticket = p->next;
    p->next++;
    return 0;
}

static Lane *lane_of( mifsk_pipeline *p, uint64_t ticket )
{
    if ( !p || ticket >= p->next )
	return nullptr;
    return &p->lanes[ticket % p->depth];
}

extern "C" int mifsk_pipeline_wait( mifsk_pipeline *p, uint64_t ticket )
{
    Lane *l = lane_of(p, ticket);
    if ( !l )
	return -EINVAL;
    // (the lane's event marks its LATEST pass: waiting for it covers every earlier one)
    return hipEventSynchronize(l->done) == hipSuccess ? 0 : -EIO;
}

extern "C" int mifsk_pipeline_join( mifsk_pipeline *p, uint64_t ticket, void *stream )
{
    Lane *l = lane_of(p, ticket);
    if ( !l )
	return -EINVAL;
    return hipStreamWaitEvent((hipStream_t)stream, l->done, 0) == hipSuccess ? 0 : -EIO;
}

extern "C" int mifsk_pipeline_drain( mifsk_pipeline *p )
{
    if ( !p )
	return -EINVAL;
    int rc = 0;
    for ( Lane &l : p->lanes )
	if ( l.used && hipStreamSynchronize(l.stream) != hipSuccess )
	    rc = -EIO;
    return rc;
}

extern "C" void *mifsk_pipeline_stream( mifsk_pipeline *p, uint64_t ticket )
{
    return p ? (void *)p->lanes[ticket % p->depth].stream : nullptr;
}

extern "C" mifsk_ctx *mifsk_pipeline_ctx( mifsk_pipeline *p, uint64_t ticket )
{
    return p ? p->lanes[ticket % p->depth].ctx : nullptr;
}
