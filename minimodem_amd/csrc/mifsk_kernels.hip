// mifsk_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X / CDNA4): the receive loop
// with ONE WORKGROUP PER STREAM (demod_kernel: a master wave and two or three worker waves),
// the legacy single-search kernel (find_frame_kernel) and the spectrum of fsk_detect_carrier.
//
// The reference's FSK receive path (src/fsk.c:107-538 driven by the loop in
// src/minimodem.c:1137-1463) as it is laid out here:
//
//  * The reference runs a full r2c FFT per bit and reads two bins.  Here a bit window is what
//    it mathematically is: two complex dot products
//        X[b] = sum_n x[n] * exp(-2 pi i b n / fftsize),  b in {mark, space}
//    One LANE owns one bit window.  The twiddle table is spread over the lanes of each 16-lane
//    row (three groups of 16 entries resident in VGPRs in the Bell-202 instantiation) and
//    `v_fmac_f64_dpp ... row_newbcast:J` broadcasts entry J and does the FMA in one
//    instruction (mifsk_devlib.h): a sample costs one convert and four of these.
//  * f64 accumulation in index order, exactly the operation sequence of the oracle, so results
//    are bit-identical to it (the reference's "-P" tests need the off-tone bin below
//    FLT_EPSILON, which f32 accumulation cannot guarantee).
//  * The receive loop is a serial, data-dependent cursor, so the stream's workgroup is a
//    two-stage pipeline.  While carrier is held the next frame is first looked for exactly
//    lock_advance samples after the last and that try is accepted whenever it reaches the
//    search limit: the positions that will be asked for lie on a LATTICE.  The WORKERS turn
//    audio into magnitudes for a batch of consecutive lattice frames -- each stages the span
//    of its 64 windows as 64 x 10 consecutive float4 (unaligned 16-byte global loads into
//    registers, the next round already in flight, then ds_write_b128 into a private, unskewed
//    LDS region) and correlates one window per lane -- while the MASTER scores the batch
//    before (one lane per frame), replays the reference's acceptance predicates and f32
//    state recurrences over it in frame order as a DPP lane scan, and writes the outputs.
//    One LDS-only barrier per batch connects them (a two-slot command block).
//  * Whatever leaves the lattice -- acquisition, a frame that fails a predicate -- goes through
//    SCAN: fsk_find_frame at one cursor, every candidate of the zig-zag evaluated by all
//    waves from a row-skewed LDS slab.  The carrier-held fine rescan of a frame the lattice had
//    already scored (minimodem.c:1357-1389) is run by the master alone (solo_fine): one pass
//    of two windows per lane from a coalesced sweep into the free magnitude buffer, scores and
//    the reference's selection (a DPP argmax) in registers, the workers undisturbed.
//  * Nothing ever depends on the speculation being right: a frame that fails any predicate is
//    recomputed from the samples.
//
// No MFMA: the path is bound by HBM reads (4 B per sample) and by the length of each stream's
// serial chain, not by a dense contraction.
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>

#include "mifsk_device.h"
#include "mifsk_devmath.h"
#include "mifsk_devlib.h"

// cycle timers for tools/counters.py; off in the production build because
// each s_memtime read costs the serial wave a round trip
#ifdef MIFSK_PROFILE
#define MIFSK_CLOCK() ((uint32_t)clock64())
#else
#define MIFSK_CLOCK() 0u
#endif

namespace mifsk {



// ---------------------------------------------------------------------------
// kernel 1: N independent fsk_find_frame problems (one 64-lane workgroup each)
// Samples are read straight from global memory with a bounds check; this is
// the compatibility / known-answer path, not the throughput path.
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(64)
void find_frame_kernel( const DevCfg *__restrict__ cfgp, const double *__restrict__ tw,
	const float *__restrict__ samples, const mifsk_search *__restrict__ problems,
	mifsk_search_result *__restrict__ results )
{
    __shared__ float2 s_mags[W_CAP];
    __shared__ float s_conf[P_CAP];
    __shared__ float s_ampl[P_CAP];
    __shared__ uint64_t s_bits[P_CAP];

    const DevCfg &cfg = *cfgp;
    const mifsk_search pr = problems[blockIdx.x];
    const float *x = samples + pr.sample_offset;
    const uint32_t navail = pr.navail;
    const uint32_t n_bits = cfg.n_bits;
    const uint32_t B = cfg.bit_nsamples;
    const uint32_t ek = pr.use_sync_string ? 1u : 0u;
    const uint32_t lane = threadIdx.x;

    const ZigZag zz(pr.try_first, pr.try_max, pr.try_step);
    uint32_t qmax = W_CAP / n_bits;
    if ( qmax > P_CAP ) qmax = P_CAP;

    float best_c = 0.0f, best_a = 0.0f;
    uint64_t best_bits = 0;
    uint32_t best_t = 0, n_tried = 0;
    bool done = false;

    for ( uint32_t c0 = 0; c0 < zz.J && !done; c0 += qmax ) {
	const uint32_t Q = zz.J - c0 < qmax ? zz.J - c0 : qmax;
	const uint32_t nwin = Q * n_bits;
	for ( uint32_t w0 = 0; w0 < nwin; w0 += 64 ) {
	    const uint32_t w = w0 + lane;
	    const bool active = w < nwin;
	    const uint32_t q = active ? w / n_bits : 0;
	    const uint32_t k = active ? w - q * n_bits : 0;
	    const uint32_t a = zz.at(c0 + q) + cfg.bit_offset[k];
	    double mr = 0.0, mi = 0.0, sr = 0.0, si = 0.0;
	    for ( uint32_t n = 0; n < B; n++ ) {
		const uint32_t idx = a + n;
		const float xv = ( active && idx < navail ) ? x[idx] : 0.0f;
		const double xd = (double)xv;
		const double *t = tw + 4 * (size_t)n;
		mr = fma(xd, t[0], mr);
		mi = fma(xd, t[1], mi);
		sr = fma(xd, t[2], sr);
		si = fma(xd, t[3], si);
	    }
	    if ( active )
		s_mags[w] = band_mag2(mr, mi, sr, si, cfg.magscalar);
	}
	__syncthreads();
	if ( lane < Q ) {
	    const FrameOut f = frame_confidence(&s_mags[lane * n_bits], cfg.req_mask[ek],
						cfg.req_val[ek], n_bits);
	    s_conf[lane] = f.conf;
	    s_ampl[lane] = f.ampl;
	    s_bits[lane] = f.bits;
	}
	__syncthreads();
	for ( uint32_t i = 0; i < Q; i++ ) {		// fsk.c:492-501
	    const float c = s_conf[i];
	    n_tried++;
	    if ( best_c < c ) {
		best_c = c;
		best_a = s_ampl[i];
		best_bits = s_bits[i];
		best_t = zz.at(c0 + i);
		if ( best_c >= pr.search_limit ) {
		    done = true;
		    break;
		}
	    }
	}
	__syncthreads();
    }
    if ( lane == 0 ) {
	mifsk_search_result r;
	r.bits = best_bits;
	r.confidence = best_c;
	r.amplitude = best_a;
	r.frame_start = best_t;
	r.n_positions = n_tried;
	results[blockIdx.x] = r;
    }
}

// ---------------------------------------------------------------------------
// kernel 2: the receive loop, one workgroup (4 waves) per stream
//
// Wave 0 is the MASTER: it owns the reference's loop state (minimodem.c:
// 1079-1133) and runs the serial decision logic.  Waves 1..3 are WORKERS: they
// turn audio into per-bit magnitudes.  Two protocols connect them, selected by
// a command block in LDS that the master publishes before a barrier:
//
//  LATTICE (steady state, pipelined).  Once carrier is held, frame k+1 is
//    searched first at exactly lock_advance samples after frame k (minimodem.c:
//    1263,1407) and that first try is accepted whenever its confidence reaches
//    the search limit (fsk.c:499) -- the normal case.  So the positions that
//    will be asked for lie on a lattice.  Workers evaluate a batch of P
//    consecutive lattice frames: each worker wave stages the audio its own 64
//    bit windows need into a private LDS region (coalesced 16-byte loads, no
//    cross-wave dependency) and correlates them, one lane per window.  One
//    barrier per batch: while the workers correlate batch k+1 the master
//    scores batch k (per-frame confidence, then the acceptance predicates and
//    f32 state recurrences replayed in frame order).
//  SCAN (acquisition, refinement, anything off the lattice).  The reference's
//    fsk_find_frame for one cursor: all four waves stage one slab and evaluate
//    every candidate position of the zig-zag scan, synchronously.
//
// Nothing ever depends on the speculation being right: a frame that fails any
// predicate is handed to the general path, which recomputes from the samples.
// ---------------------------------------------------------------------------

__device__ __forceinline__ void lds_barrier();
__device__ __forceinline__ float lane_bcast( float v, uint32_t src );

// A word of LDS that another wave of the workgroup may be writing.  Through a generic
// pointer hipcc makes a volatile access a system-scope FLAT instruction and waits for
// vmcnt(0) behind it -- i.e. for every global load the wave has in flight; as a DS access it
// is one ds_read_b32 / ds_write_b32 (LDS operations of a workgroup are coherent as they are).
typedef __attribute__((address_space(3))) volatile uint32_t lds_vu32_t;
__device__ __forceinline__ uint32_t lds_peek( const uint32_t *p ) { return *(lds_vu32_t *)p; }
__device__ __forceinline__ void lds_poke( uint32_t *p, uint32_t v ) { *(lds_vu32_t *)p = v; }

// A uniform value read ONCE and kept: the empty asm makes it opaque, so the compiler can
// neither re-load it from the configuration at every use (an s_load and a wait for the scalar
// cache in the middle of a serial chain) nor fold it -- it stays in a scalar register or in a
// lane of a spill VGPR (a v_readlane: one cycle).
template <typename T>
__device__ __forceinline__ T keep_scalar( T v )
{
    asm volatile("" : "+s"(v));
    return v;
}

// A stream's workgroup is a master wave and NW worker waves (blockDim.x = 64 (NW +
// 1)): three workers, or two in the Bell-202 instantiation -- configs[1] measured
// 0.448 ms with two (rounds of 12 frames, batches of 24, 142 VGPRs at three waves
// per SIMD: nothing spilled), 0.474-0.486 with three (19 / 38, 128 VGPRs, 12
// spilled), 0.52 with one; the long-window linear modes are 10-15 % faster with
// three (tools/gpu/eng50.py).

struct StreamLds {
    float2	mags[2][W_CAP];	// [buffer][window]: (mark, space) magnitudes
    uint64_t	c_bits[P_CAP];
    float	c_conf[P_CAP];
    float	c_ampl[P_CAP];
    uint32_t	c_pos[P_CAP];
    // 1 + number of the LATTICE command whose results are no longer wanted (the
    // master needs a SCAN): the workers poll it and go straight to the barrier
    uint32_t	abort;
    uint32_t	pad[3];		// (which entries are valid is the master's private state)
    // work counters (MIFSK_CNT_*), bumped by lane 0 of the master: kept out of the
    // scalar registers, which the loop state needs
    uint32_t	cnt[MIFSK_NCOUNTERS];
    // Two command slots used alternately: the one published before barrier
    // number n is slot n & 1, so a slot is rewritten only after every wave has
    // passed another barrier and is done reading it.
    struct Cmd {
	uint32_t	op;		// CMD_*
	uint32_t	nq;		// SCAN: candidates in c_pos[]
	uint32_t	stage;		// SCAN: samples to (re)stage at row_org first (0: none)
	uint32_t	row_org;	// SCAN: absolute sample index of slab row 0
	uint32_t	anchor;		// LATTICE: first-try position of frame 0
	uint32_t	frames;		// LATTICE: frames in the batch
	uint32_t	buf;		// LATTICE: mags[] buffer to fill
	uint32_t	pad;
    }		cmd[2];
    float	slab[1];	// really slab_floats long (dynamic LDS)
};

enum { CMD_EXIT = 0, CMD_SCAN = 1, CMD_LATTICE = 2, CMD_IDLE = 3 };




extern __shared__ __attribute__((aligned(16))) unsigned char mifsk_smem[];



// SCAN: stage samples [row_org, row_org + nstage) into the slab (all threads)
__device__ __forceinline__ void par_stage( const DevCfg &cfg, StreamLds *lds, const float *__restrict__ x,
	uint32_t N, uint32_t slab_cap, uint32_t row_org, uint32_t nstage )
{
    const uint32_t org4 = row_org & ~3u;
    const uint32_t head = row_org - org4;		// 0..3 samples before row 0: dropped
    const uint32_t nvec = ( nstage + head + 3 ) >> 2;
    for ( uint32_t v0 = 0; v0 < nvec; v0 += blockDim.x * STAGE_VEC ) {
	float4 buf[STAGE_VEC];
	// all loads of the round in flight before the first LDS write; vector
	// groups wholly past the end are neither loaded nor stored (uniform tests:
	// a refinement with the carrier held stages a few hundred samples only)
#pragma unroll
	for ( int i = 0; i < STAGE_VEC; i++ ) {
	    const uint32_t v = v0 + i * blockDim.x + threadIdx.x;
	    if ( v0 + i * blockDim.x < nvec )
		buf[i] = load4_raw(x, org4 + ( v << 2 ), N);
	}
#pragma unroll
	for ( int i = 0; i < STAGE_VEC; i++ ) {
	    const uint32_t v = v0 + i * blockDim.x + threadIdx.x;
	    if ( v0 + i * blockDim.x < nvec && v < nvec )
		store4_skewed(cfg, lds->slab, slab_cap, v << 2, head, buf[i], org4 + ( v << 2 ), N);
	}
    }
}

// SCAN: correlate every bit window of candidates c_pos[0..nq) into mags[0]
template <bool USE_SLAB>
__device__ __forceinline__ void par_correlate( const DevCfg &cfg, const double *__restrict__ tw, StreamLds *lds,
	const float *__restrict__ x, uint32_t N, uint32_t row_org, uint32_t nq )
{
    const uint32_t n_bits = cfg.n_bits;
    const uint32_t B = cfg.bit_nsamples;
    const uint32_t nwin = nq * n_bits;
    for ( uint32_t w0 = 0; w0 < nwin; w0 += blockDim.x ) {
	const uint32_t w = w0 + threadIdx.x;
	if ( w0 + ( threadIdx.x & ~63u ) >= nwin )
	    continue;			// nothing for this wave in this pass (a refinement has ~90 windows)
	const bool active = w < nwin;
	const uint32_t q = active ? udiv_magic(w, n_bits, cfg.nbits_magic) : 0;
	const uint32_t k = active ? w - q * n_bits : 0;
	const uint32_t a = lds->c_pos[q] + cfg.bit_offset[k];
	double acc[4] = { 0.0, 0.0, 0.0, 0.0 };
	if ( USE_SLAB ) {
	    corr_skewed_stream(cfg, tw, lds->slab, a - row_org, threadIdx.x & 63u, acc);
	} else {
	    for ( uint32_t n = 0; n < B; n++ ) {
		const uint32_t idx = a + n;
		const float xv = ( idx < N && idx >= a ) ? x[idx] : 0.0f;
		const double xd = (double)xv;
		const double *t = tw + 4 * (size_t)n;
		acc[0] = fma(xd, t[0], acc[0]);
		acc[1] = fma(xd, t[1], acc[1]);
		acc[2] = fma(xd, t[2], acc[2]);
		acc[3] = fma(xd, t[3], acc[3]);
	    }
	}
	if ( active )
	    lds->mags[0][w] = band_mag2(acc[0], acc[1], acc[2], acc[3], cfg.magscalar);
    }
}

// SCAN: what every wave does between "command published" and "correlated"
template <bool USE_SLAB>
__device__ __forceinline__ void scan_part( const DevCfg &cfg, const double *__restrict__ tw,
	StreamLds *lds, const StreamLds::Cmd *cmd, const float *__restrict__ x, uint32_t N,
	uint32_t slab_cap )
{
    const uint32_t nq = cmd->nq;
    const uint32_t row_org = cmd->row_org;
    if ( USE_SLAB && cmd->stage ) {
	par_stage(cfg, lds, x, N, slab_cap, row_org, cmd->stage);
	lds_barrier();
    }
    par_correlate<USE_SLAB>(cfg, tw, lds, x, N, row_org, nq);
    lds_barrier();
}

// Workgroup barrier that orders LDS only.  __syncthreads() also drains vmcnt,
// i.e. it would make every wave wait for its outstanding GLOBAL loads (the
// workers' register prefetch) at every batch.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}



template <bool USE_SLAB>
struct Master {
    const DevCfg	&cfg;
    const double	*tw;
    const float		*x;		// this stream's samples
    uint32_t		N;		// valid samples; reads beyond see 0.0
    StreamLds		*lds;
    uint32_t		slab_cap;	// SCAN: samples the whole slab can hold
    uint32_t		slab_lo, slab_hi;	// SCAN: absolute range currently staged
    uint32_t		lat_batch;	// LATTICE: frames per batch = per round x rounds (0 = off)
    uint32_t		conf_idx;	// LATTICE: where this lane's frame starts in mags[buf][]
    uint32_t		lat_first;	// LATTICE: frames of the first batch after a (re)start
    // How far to speculate.  `spec` = frames per batch: after a failure, as many
    // as were accepted from the lattice since the failure before it; doubled
    // every time a whole batch holds.  (A signal whose first tries keep falling
    // short of the search limit, or that is refined every few frames, gets
    // short batches; Bell-202 full ones.)  After `cold` batches in a row that
    // yielded nothing the lattice pauses.
    uint32_t		spec, run, cold, pause;
    uint32_t		spec_floor;	// never fewer frames than this per batch
    uint32_t		lane;
    // the lattice batch the workers are computing right now
    bool		inflight;
    uint32_t		inflight_anchor, inflight_frames, inflight_buf;
    uint32_t		seq;		// barriers that published a command so far
    // scored lattice frames held in lds->c_conf/c_ampl/c_bits[0 .. lat_n): entry e
    // is the frame whose first try sits at lat_anchor + e * lock_advance
    uint32_t		lat_n, lat_anchor;
    uint32_t		hit_base = 0xFFFFFFFFu;	// cursor of the last scan a lattice frame answered
    bool		cnt_on = false;		// work counters wanted (io.d_counters)
    uint32_t		cyc_par = 0, cyc_conf = 0, cyc_wait = 0, cyc_scan_wait = 0;
    uint32_t		cyc_sf_load = 0, cyc_sf_corr = 0, cyc_sf_score = 0, cyc_sf_sel = 0;	// solo_fine's phases (profile build)
    // what the steady state reads of the configuration at every batch, read once (keep_scalar)
    uint32_t		h_la, h_la_magic, h_nbits;
    uint64_t		h_req_mask, h_req_val;
    TwGroup		mtg[3];		// the table's first groups, resident (solo_fine)

    template <int NG>
    __device__ __forceinline__ void keep_groups()
    {
#pragma unroll
	for ( int gi = 0; gi < NG; gi++ )
	    mtg[gi] = tw_group_load(tw, (uint32_t)gi, lane);
    }

    __device__ __forceinline__ Master( const DevCfg &c, const double *t, const float *xs,
	    uint32_t n, StreamLds *l, uint32_t cap, uint32_t lf, uint32_t lat_round )
	: cfg(c), tw(t), x(xs), N(n), lds(l), slab_cap(cap), slab_lo(0), slab_hi(0),
	  lat_batch(lf), lane(threadIdx.x), inflight(false), inflight_anchor(0),
	  inflight_frames(0), inflight_buf(0), seq(0), lat_n(0), lat_anchor(0)
    {
	lat_first = lat_round;
	spec = lat_batch;
	run = cold = pause = 0;
	spec_floor = 2;
	h_la = keep_scalar(cfg.lock_advance);
	h_la_magic = keep_scalar(cfg.la_magic);
	h_nbits = keep_scalar(cfg.n_bits);
	h_req_mask = keep_scalar(cfg.req_mask[0]);
	h_req_val = keep_scalar(cfg.req_val[0]);
	// frame `lane` of a batch = frame (lane % lat_round) of round (lane / lat_round)
	conf_idx = lane * cfg.n_bits;
	if ( cfg.lat_grid && lat_round ) {
	    const uint32_t r = lane / lat_round, fr = lane - r * lat_round;
	    conf_idx = r * ( lat_round * ( cfg.n_bits - 1u ) + 1u ) + fr * ( cfg.n_bits - 1u );
	}
    }

    // (ds_add_u32 without return: fire and forget, the serial wave never waits for it)
    __device__ __forceinline__ void bump( uint32_t which, uint32_t by = 1u ) const
    {
	if ( cnt_on && lane == 0 )
	    (void)__hip_atomic_fetch_add(&lds->cnt[which], by, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }

    // index of the scored lattice frame whose first try is at p, or ~0u
    __device__ __forceinline__ uint32_t lattice_lookup( uint32_t p ) const
    {
	if ( !lat_n || p < lat_anchor )
	    return ~0u;
	const uint32_t d = p - lat_anchor;
	const uint32_t e = udiv_magic(d, h_la, h_la_magic);
	return ( e < lat_n && e * h_la == d ) ? e : ~0u;
    }

    // the slot for the command that the NEXT barrier publishes
    __device__ __forceinline__ StreamLds::Cmd *next_cmd() { return &lds->cmd[seq & 1u]; }

    // frames of a lattice batch anchored at `anchor` that still start inside the stream
    __device__ __forceinline__ uint32_t lattice_frames_at( uint32_t anchor ) const
    {
	if ( anchor >= N )
	    return 0;
	const uint32_t left = udiv_magic(N - anchor - 1u, h_la, h_la_magic) + 1u;
	return left < lat_batch ? left : lat_batch;
    }

    // publish a LATTICE (or IDLE) command; the caller then meets the barrier
    __device__ __forceinline__ void publish_lattice( uint32_t anchor, uint32_t frames, uint32_t buf )
    {
	if ( lane == 0 ) {
	    StreamLds::Cmd *c = next_cmd();
	    c->op = frames ? CMD_LATTICE : CMD_IDLE;
	    c->anchor = anchor;
	    c->frames = frames;
	    c->buf = buf;
	}
	inflight = frames != 0;
	inflight_anchor = anchor;
	inflight_frames = frames;
	inflight_buf = buf;
	slab_lo = slab_hi = 0;		// the regions overwrite whatever SCAN had staged
    }

    // Start the pipeline at `anchor` (nothing is in flight).  The first batch is
    // a single worker round: scoring starts one round earlier, the batches
    // after it are full ones.
    __device__ __forceinline__ void lattice_start( uint32_t anchor )
    {
	uint32_t frames = lattice_frames_at(anchor);
	if ( !frames )
	    return;
	frames = frames < lat_first ? frames : lat_first;
	frames = frames < spec ? frames : spec;
	publish_lattice(anchor, frames, 0);
	lds_barrier();			// workers pick the command up
	seq++;
    }

    // The batch in flight is the one the cursor has reached: start the workers
    // on the batch after it (speculatively) and score this one.
    __device__ __forceinline__ void lattice_advance()
    {
	const uint32_t anchor = inflight_anchor, frames = inflight_frames, buf = inflight_buf;
	const uint32_t next = anchor + frames * h_la;
	uint32_t nf = lattice_frames_at(next);
	nf = nf < spec ? nf : spec;
	publish_lattice(next, nf, buf ^ 1u);
	const uint32_t t_w = MIFSK_CLOCK();
	lds_barrier();			// batch `anchor` is complete in mags[buf]
	seq++;
	const uint32_t t_c = MIFSK_CLOCK();
	cyc_wait += t_c - t_w;
	bump(MIFSK_CNT_LATTICE_BATCHES);
	if ( lane < frames ) {
	    uint32_t fb = 0u;
	    const FrameOut fo = frame_confidence_any_staged(&lds->mags[buf][conf_idx], h_req_mask, h_req_val, h_nbits,
							    fb);
	    if ( cnt_on && fb )
		bump(MIFSK_CNT_CONF_FALLBACKS);
	    lds->c_conf[lane] = fo.conf;
	    lds->c_ampl[lane] = fo.ampl;
	    lds->c_bits[lane] = fo.bits;
	}
	lat_n = frames;
	lat_anchor = anchor;
	wave_lds_sync();
	cyc_conf += MIFSK_CLOCK() - t_c;
    }

    // Whatever the failing frame's fate, the batch in flight sits on a lattice the
    // cursor is about to leave: tell the workers to drop it now (their issue slots
    // are better spent by the other workgroups' waves).
    __device__ __forceinline__ void give_up()
    {
	if ( inflight && lane == 0 )	// command number seq - 1 is the batch in flight
	    lds_poke(&lds->abort, seq);
	inflight = false;
    }

    // SCAN: evaluate candidates c_pos[0..nq) (already in LDS) spanning [lo, hi)
    __device__ __forceinline__ void evaluate( uint32_t nq, uint32_t kind, uint32_t lo, uint32_t hi )
    {
	bool restage = false;
	if ( USE_SLAB && ( lo < slab_lo || hi > slab_hi ) ) {
	    restage = true;
	    slab_lo = lo;
	    // Without a carrier (and in modes that have no lattice) the searches
	    // that follow advance through the stream and reuse what is staged now:
	    // fill the slab.  With the lattice running, the next SCAN is many
	    // frames away and the regions overwrite the slab in between: stage
	    // what this search reads and no more.
	    const uint32_t need = ( hi - lo + 7u ) & ~3u;
	    slab_hi = lo + ( ( kind == 0u && lat_batch && need < slab_cap ) ? need : slab_cap );
	    bump(MIFSK_CNT_STAGES);
	}
	StreamLds::Cmd *c = next_cmd();
	if ( lane == 0 ) {
	    c->op = CMD_SCAN;
	    c->nq = nq;
	    c->stage = restage ? slab_hi - slab_lo : 0u;
	    c->row_org = slab_lo;
	}
	if ( inflight && lane == 0 )	// command number seq - 1 is the batch in flight
	    lds_poke(&lds->abort, seq);
	inflight = false;		// the barrier below also retires any batch in flight
	bump(MIFSK_CNT_BATCHES);
	bump(MIFSK_CNT_POSITIONS, nq);
	const uint32_t t_par = MIFSK_CLOCK();
	lds_barrier();			// command (and c_pos[]) published
	seq++;
	cyc_scan_wait += MIFSK_CLOCK() - t_par;	// = waiting for the batch in flight to retire
	scan_part<USE_SLAB>(cfg, tw, lds, c, x, N, slab_cap);
	const uint32_t t_conf = MIFSK_CLOCK();
	cyc_par += t_conf - t_par;
	if ( lane < nq ) {
	    uint32_t fb = 0u;
	    const FrameOut f = frame_confidence_any_staged(&lds->mags[0][lane * cfg.n_bits],
						cfg.req_mask[kind], cfg.req_val[kind], cfg.n_bits, fb);
	    if ( cnt_on && fb )
		bump(MIFSK_CNT_CONF_FALLBACKS);
	    lds->c_conf[lane] = f.conf;
	    lds->c_ampl[lane] = f.ampl;
	    lds->c_bits[lane] = f.bits;
	}
	wave_lds_sync();
	cyc_conf += MIFSK_CLOCK() - t_conf;
    }

    // The carrier-held fine rescan (minimodem.c:1357-1389) of a frame whose first
    // try -- candidate 0 of both scans -- is a scored lattice frame, run by this
    // wave ALONE: the other candidates' windows all lie within a few hundred
    // samples, which the wave fetches in one coalesced sweep into the magnitude
    // buffer no batch is using; each lane then reads its window from there
    // (tables in registers), scores and selects in registers.  No slab, no
    // barrier, nothing of the lattice state touched: the workers go on with the
    // batch in flight, which stays good when the rescan confirms the position.
    // Windows of 4 NQ samples; false when this does not apply (the caller then
    // takes the general route).
    template <int NQ>
    __device__ __forceinline__ bool solo_fine( uint32_t base, const ZigZag &zz, uint32_t first,
	    const ScanResult &c0, ScanResult &r )
    {
	static_assert(NQ >= 1 && NQ <= 12, "three table groups");
	constexpr uint32_t B = 4u * NQ;
	const uint32_t nb = cfg.n_bits;
	const uint32_t nwin = ( zz.J - 1u ) * nb;
	if ( zz.J < 2u || nwin > 128u || zz.J - 1u > 64u )
	    return false;
	const uint32_t t_sf0 = MIFSK_CLOCK();
	// what the candidates read
	const uint32_t lo = base + first - zz.D * zz.step + cfg.bit_offset[0];
	const uint32_t hi = base + first + ( zz.U - 1u ) * zz.step + cfg.bit_offset[( nb - 1u ) & 63u] + B;
	// The span is staged from `lo` itself (the 16-byte global loads need no alignment): with an
	// even search step every window then starts an EVEN number of samples into the buffer (bit
	// offsets are multiples of 4 here) and is read two samples per LDS instruction.
	const uint32_t org4 = lo;
	const bool pairs = ( zz.step & 1u ) == 0u;
	const uint32_t nvec = ( hi - org4 + 3u ) >> 2;
	if ( hi > N || hi < lo || org4 + 4u * nvec > N || nvec > 128u
		|| nvec * 16u + nwin * sizeof(float2) > sizeof(lds->mags[0]) )
	    return false;
	// the buffer the batch in flight does NOT fill has been scored already
	// (lattice_advance): samples first, then this search's magnitudes
	float *sbuf = reinterpret_cast<float *>(lds->mags[inflight ? ( inflight_buf ^ 1u ) : 0u]);
	float2 *sm = reinterpret_cast<float2 *>(sbuf + 4u * nvec);
	{
	    const bool h0 = lane < nvec, h1 = lane + 64u < nvec;
	    float4_u v0, v1;
	    if ( h0 ) v0 = *reinterpret_cast<const float4_u *>(x + org4 + 4u * lane);
	    if ( h1 ) v1 = *reinterpret_cast<const float4_u *>(x + org4 + 4u * ( lane + 64u ));
	    if ( h0 ) *reinterpret_cast<float4 *>(sbuf + 4u * lane) = make_float4(v0.x, v0.y, v0.z, v0.w);
	    if ( h1 ) *reinterpret_cast<float4 *>(sbuf + 4u * ( lane + 64u )) = make_float4(v1.x, v1.y, v1.z, v1.w);
	}
	const TwGroup (&tg)[3] = mtg;	// (resident: 24 registers against six LDS loads per rescan)
	wave_lds_sync();
#ifdef MIFSK_PROFILE
	asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
	const uint32_t t_sf1 = MIFSK_CLOCK();
	{
	    // ONE pass: lane l sums window l AND window l + 64 (slots without a window shadow a real
	    // one: every lane must be active for the broadcasts) -- eight independent accumulator
	    // chains instead of four, which is what a wave running alone needs to keep its f64
	    // pipe issuing (a dependent v_fmac_f64_dpp comes back after ~30 cycles), and one trip
	    // through the table, the magnitudes and the loop instead of two.  Same sums, same order.
	    const uint32_t wA = lane < nwin ? lane : 0u;
	    const uint32_t wB = lane + 64u < nwin ? lane + 64u : wA;
	    const uint32_t qA = udiv_magic(wA, nb, cfg.nbits_magic), qB = udiv_magic(wB, nb, cfg.nbits_magic);
	    // (on a grid of bit lengths bit k starts k B into the frame: no per-lane load from the
	    // offset table in the middle of the chain)
	    const bool grid = cfg.lat_grid != 0u;
	    const uint32_t kA = wA - qA * nb, kB = wB - qB * nb;
	    const uint32_t oA = grid ? kA * B : cfg.bit_offset[kA & 63u], oB = grid ? kB * B : cfg.bit_offset[kB & 63u];
	    const float *pA = sbuf + ( base + zz.at(1u + qA) + oA - org4 );
	    const float *pB = sbuf + ( base + zz.at(1u + qB) + oB - org4 );
	    double accA[4] = { 0.0, 0.0, 0.0, 0.0 }, accB[4] = { 0.0, 0.0, 0.0, 0.0 };
	    auto quad_of = [&]( const float *p, int q ) -> float4 {
		if ( pairs ) {
		    const float2 a = *reinterpret_cast<const float2 *>(p + 4 * q);
		    const float2 b = *reinterpret_cast<const float2 *>(p + 4 * q + 2);
		    return make_float4(a.x, a.y, b.x, b.y);
		}
		return make_float4(p[4 * q], p[4 * q + 1], p[4 * q + 2], p[4 * q + 3]);
	    };
	    float4 xa = quad_of(pA, 0), xb = quad_of(pB, 0);
	    dpp_settle();
#define MIFSK_SOLO2_SAMPLE(J, G, XA, XB) fma4_bcast<J>(accA, G, XA); fma4_bcast<J>(accB, G, XB);
#define MIFSK_SOLO2_QUAD(Q)									\
	    if ( (Q) < NQ ) {									\
		const float4 ca = xa, cb = xb;							\
		if ( (Q) + 1 < NQ ) {		/* the next quad's reads ahead of this quad's sums */	\
		    xa = quad_of(pA, (Q) + 1 < NQ ? (Q) + 1 : 0);					\
		    xb = quad_of(pB, (Q) + 1 < NQ ? (Q) + 1 : 0);					\
		}										\
		const TwGroup &G = tg[(Q) / 4 < 3 ? (Q) / 4 : 0];					\
		MIFSK_SOLO2_SAMPLE(4 * ( (Q) % 4 ) + 0, G, ca.x, cb.x)					\
		MIFSK_SOLO2_SAMPLE(4 * ( (Q) % 4 ) + 1, G, ca.y, cb.y)					\
		MIFSK_SOLO2_SAMPLE(4 * ( (Q) % 4 ) + 2, G, ca.z, cb.z)					\
		MIFSK_SOLO2_SAMPLE(4 * ( (Q) % 4 ) + 3, G, ca.w, cb.w)					\
	    }
	    MIFSK_SOLO2_QUAD(0) MIFSK_SOLO2_QUAD(1) MIFSK_SOLO2_QUAD(2) MIFSK_SOLO2_QUAD(3)
	    MIFSK_SOLO2_QUAD(4) MIFSK_SOLO2_QUAD(5) MIFSK_SOLO2_QUAD(6) MIFSK_SOLO2_QUAD(7)
	    MIFSK_SOLO2_QUAD(8) MIFSK_SOLO2_QUAD(9) MIFSK_SOLO2_QUAD(10) MIFSK_SOLO2_QUAD(11)
#undef MIFSK_SOLO2_QUAD
#undef MIFSK_SOLO2_SAMPLE
	    if ( lane < nwin )
		sm[lane] = band_mag2(accA[0], accA[1], accA[2], accA[3], cfg.magscalar);
	    if ( lane + 64u < nwin )
		sm[lane + 64u] = band_mag2(accB[0], accB[1], accB[2], accB[3], cfg.magscalar);
	}
	wave_lds_sync();
	const uint32_t t_sf2 = MIFSK_CLOCK();
	FrameOut f;
	f.conf = 0.0f; f.ampl = 0.0f; f.bits = 0;
	if ( lane < zz.J - 1u ) {
	    uint32_t fb = 0u;
	    f = frame_confidence_any_staged(&sm[lane * nb], cfg.req_mask[0], cfg.req_val[0], nb, fb);
	    if ( cnt_on && fb )
		bump(MIFSK_CNT_CONF_FALLBACKS);
	}
	bump(MIFSK_CNT_POSITIONS, zz.J - 1u);
	const uint32_t t_sf3 = MIFSK_CLOCK();
	// fsk.c:492-501 in scan order, candidate 0 first; the limit of this scan
	// is INFINITY (minimodem.c:1367)
	// -- i.e. the FIRST candidate, in scan order, that attains the maximum of the candidates'
	// confidences wins if that maximum exceeds candidate 0's (strict >, a NaN never wins, and
	// after an infinite confidence nothing can): lane i - 1 holds candidate i
	r = c0;
	uint32_t win = 0;
	{
	    const bool cand = lane < zz.J - 1u && f.conf == f.conf;
	    const float best = wave_max_f32(cand ? f.conf : -INFINITY);
	    if ( r.conf < best ) {
		const unsigned long long at = __ballot(cand && f.conf == best);
		win = (uint32_t)__ffsll((long long)at);		// 1 + the first such lane = its candidate index
		r.conf = best;
	    }
	}
	if ( win ) {
	    r.ampl = lane_bcast(f.ampl, win - 1u);
	    const uint32_t blo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)f.bits, (int)( win - 1u ));
	    const uint32_t bhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)( f.bits >> 32 ), (int)( win - 1u ));
	    r.bits = ( (uint64_t)bhi << 32 ) | blo;
	    r.start = zz.at(win);
	}
	cyc_sf_load += t_sf1 - t_sf0;
	cyc_sf_corr += t_sf2 - t_sf1;
	cyc_sf_score += t_sf3 - t_sf2;
	cyc_sf_sel += MIFSK_CLOCK() - t_sf3;
	return true;
    }

    // fsk_find_frame at cursor `base` (absolute)
    __device__ __forceinline__ ScanResult scan( uint32_t base, const ZigZag &zz, uint32_t first,
	    float limit, uint32_t kind )
    {
	ScanResult r;
	r.conf = 0.0f; r.ampl = 0.0f; r.bits = 0; r.start = 0;
	hit_base = 0xFFFFFFFFu;
	if ( zz.J == 0 )
	    return r;
	const uint32_t p0 = base + first;

	// the first candidate may be a lattice frame that is already scored
	// (those are scored against the data string)
	{
	    const uint32_t hit = kind == 0u ? lattice_lookup(p0) : ~0u;
	    if ( hit != ~0u ) {
		const float c = lds->c_conf[hit];
		if ( c > 0.0f && c >= limit ) {		// fsk.c:492,499
		    r.conf = c;
		    r.ampl = lds->c_ampl[hit];
		    r.bits = lds->c_bits[hit];
		    r.start = first;
		    bump(MIFSK_CNT_CACHE_HITS);
		    hit_base = base;		// (candidate 0 of a rescan at this cursor is this frame)
		    return r;
		}
	    }
	}

	uint32_t qmax = W_CAP / cfg.n_bits;
	if ( qmax > P_CAP ) qmax = P_CAP;
	bool done = false;
	for ( uint32_t c0 = 0; c0 < zz.J && !done; c0 += qmax ) {
	    const uint32_t Q = zz.J - c0 < qmax ? zz.J - c0 : qmax;
	    // Extent of this chunk's candidates, in closed form: within the zig-zag
	    // order the up-steps grow and the down-steps shrink with the index, so
	    // the extremes are the LAST up / down step inside [c0, c0 + Q) -- or
	    // the chunk's first candidate when it has none of that kind.
	    const uint32_t iend = c0 + Q - 1u;
	    const uint32_t lu = iend > 2u * zz.D ? iend : ( ( iend & 1u ) ? iend : iend - 1u );
	    const uint32_t ld = ( iend < 2u * zz.D ? iend : 2u * zz.D ) & ~1u;
	    const bool has_up = lu >= ( c0 > 1u ? c0 : 1u ) && lu <= iend && iend >= 1u;
	    const bool has_down = ld >= 2u && ld >= c0;
	    const uint32_t thi = has_up ? zz.at(lu) : zz.at(c0);
	    const uint32_t tlo = has_down ? zz.at(ld) : zz.at(c0);
	    if ( lane < Q )
		lds->c_pos[lane] = base + zz.at(c0 + lane);
	    lat_n = 0;			// the SCAN's scores overwrite the lattice frames'
	    evaluate(Q, kind, base + tlo, base + thi + cfg.last_reach);
	    // fsk.c:492-501 over the chunk: one LDS read, then the reference's
	    // selection in scan order on lane values (strict >, first wins ties,
	    // stop at the limit); the winner's other fields are fetched once
	    const float cl = lane < Q ? lds->c_conf[lane] : 0.0f;
	    uint32_t win = ~0u;
	    for ( uint32_t i = 0; i < Q; i++ ) {
		const float c = lane_bcast(cl, i);
		if ( r.conf < c ) {
		    r.conf = c;
		    win = i;
		    if ( r.conf >= limit ) {
			done = true;
			break;
		    }
		}
	    }
	    if ( win != ~0u ) {
		r.ampl = lds->c_ampl[win];
		r.bits = lds->c_bits[win];
		r.start = zz.at(c0 + win);
	    }
	}
	return r;
    }
};


// demod_kernel's arguments as they lie in the kernarg segment (each parameter at its natural
// alignment, in order: exactly this struct's layout), for the cold end of the loop to re-read
// them there (KernArgs in mifsk_devlib.h).  The kernel itself takes them as separate
// parameters: only a `__restrict__` POINTER PARAMETER tells the compiler that nothing the
// kernel stores can change *cfgp, and without that every configuration read inside the loop
// is a vector load plus v_readfirstlane instead of a scalar load (measured with a struct
// parameter: configs[1] 0.45 -> 0.54 ms, 12000 baud 1.29 -> 1.89 ms).
// streams that arrive in pieces (mifsk_demod_slab) and chained launches: what the resumable
// instantiation demod_kernel<., ., ., true> needs beyond the batch itself (DESIGN.md 4.10,
// 4.11; the wavefront engine's WaveAuto carries the same five)
struct WgResume {
    mifsk_stream_state	*d_state;	// [nstreams] or null (then the kernel behaves as the plain one)
    const uint64_t	*d_origin;	// [nstreams] index of each row's first sample in its stream, or null
    uint32_t		final;		// no more samples will follow these rows
    uint32_t		limit;		// chained launch: this call sees the first `limit` samples of a row (0: all)
    uint32_t		append;		// chained launch: outputs continue behind the chunk before
    uint32_t		bufsize;	// the reference's samplebuf_size (minimodem.c:1056-1069)
};

struct DemodArgs {
    const DevCfg	*cfgp;
    const double	*tw;
    mifsk_demod_io	io;
    uint32_t		slab_cap, lat_frames, lat_rounds, region_floats, region_cap, lat_mode;
    WgResume		rs;
};

// where stream s writes its results.  Made once: the serial loop is latency-bound,
// and a scalar kept in (or spilled to a VGPR lane from) a register costs a cycle where a
// reload from the kernarg segment costs a scalar-cache round trip per block of frames
// (measured: 0.45 -> 0.52 ms on configs[1] with the pointers re-made at every use)
struct StreamOut {
    uint8_t		*bytes;
    uint64_t		*bits;
    mifsk_frame		*frames;
    mifsk_episode	*eps;
    size_t		fcap, ecap;
};


// The reference's receive loop (minimodem.c:1137-1463); executed by wave 0 only.
template <bool USE_SLAB, int NQ, bool ST>
__device__ __forceinline__ void master_loop( const DevCfg &cfg, const double *__restrict__ tw,
	const mifsk_demod_io &io, uint32_t slab_cap, uint32_t lat_frames, uint32_t lat_round,
	uint32_t lat_mode, uint32_t base0, StreamLds *lds, const WgResume &rs )
{
    const uint32_t s = blockIdx.x;
    const float *x = io.d_samples + (size_t)s * io.stream_stride;
    uint32_t N = io.d_nsamples ? io.d_nsamples[s] : io.nsamples;
    if ( io.nstreams > 1 && (size_t)N > io.stream_stride )
	N = (uint32_t)io.stream_stride;		// never trust a length beyond the row
    // chained launches: this call takes the stream up to rs.limit only
    bool cut = false;
    if constexpr ( ST ) {
	if ( rs.d_state && rs.limit != 0u && rs.limit < N ) {
	    N = rs.limit;
	    cut = true;
	}
    }

    StreamOut o;
    o.fcap = io.frames_cap;
    o.ecap = io.episodes_cap;
    o.bytes = io.d_bytes ? io.d_bytes + (size_t)s * o.fcap : nullptr;
    o.bits = io.d_bits ? io.d_bits + (size_t)s * o.fcap : nullptr;
    o.frames = io.d_frames ? io.d_frames + (size_t)s * o.fcap : nullptr;
    o.eps = io.d_episodes ? io.d_episodes + (size_t)s * o.ecap : nullptr;
    const uint32_t lane = threadIdx.x;
    const bool t0 = lane == 0;

    Master<USE_SLAB> ctx(cfg, tw, x, N, lds, slab_cap, lat_frames, lat_round);
    if constexpr ( NQ > 0 )
	ctx.template keep_groups<( NQ + 3 ) / 4>();
    // What the bulk path below reads of the configuration for every batch of lattice frames,
    // read once (keep_scalar): left as cfg.x the compiler re-loads each at every use -- a dozen
    // s_load + s_waitcnt round trips through the scalar cache per batch, in the one wave whose
    // serial chain paces the workgroup.
    const uint32_t h_first1 = keep_scalar(cfg.try_first[1]);
    const uint32_t h_expect = keep_scalar(cfg.expect_nsamples);
    const uint32_t h_fn = keep_scalar(cfg.frame_nsamples);
    const uint32_t h_overscan = keep_scalar(cfg.overscan);
    const float h_limit = keep_scalar(cfg.search_limit);
    const float h_thr = keep_scalar(cfg.conf_threshold);
    // data_bits_of() (minimodem.c:1415-1428) as one shift and one mask
    const uint32_t h_dshift = keep_scalar(( cfg.has_stopbits ? 1u : 0u ) + cfg.nstartbits);
    const uint64_t h_dmask = keep_scalar(cfg.n_data_bits >= 64u ? ~0ULL : ( 1ULL << cfg.n_data_bits ) - 1ULL);
    const bool h_msb = cfg.msb_first != 0u;
    const bool h_rx_sync = cfg.do_rx_sync != 0u;
    const uint64_t h_sync_byte = keep_scalar(cfg.sync_byte);
    // ... and what an iteration of the general path reads (one per refinement: hundreds per
    // stream on a noisy signal)
    const uint32_t h_try_max0 = keep_scalar(cfg.try_max[0]), h_try_max1 = keep_scalar(cfg.try_max[1]);
    const uint32_t h_try_step0 = keep_scalar(cfg.try_step[0]), h_try_step1 = keep_scalar(cfg.try_step[1]);
    const uint32_t h_first0 = keep_scalar(cfg.try_first[0]);
    // LINEAR rounds are cheap (coalesced staging, short windows): never speculate
    // less than one round; DIRECT rounds stream whole windows per lane
    if ( lat_mode == LAT_LINEAR )
	ctx.spec_floor = lat_round;
    ctx.cnt_on = io.d_counters != nullptr;

    // reference loop state (minimodem.c:1079-1088,1132-1133), uniform in the wave
    bool carrier = false;
    float confidence_total = 0.0f, amplitude_total = 0.0f;
    uint32_t nframes_decoded = 0;
    uint64_t carrier_nsamples = 0;
    uint32_t noconfidence = 0;
    uint32_t advance = 0;
    float track_amplitude = 0.0f, peak_confidence = 0.0f;

    uint32_t base = base0 < N ? base0 : N;	// absolute index of samplebuf[0]
    uint32_t n_out_frames = 0, n_out_bytes = 0, n_out_eps = 0, ep_first = 0;
    uint32_t ep_b_mark = 0;			// (carried in the state record only)
    uint32_t status = 0;

    // mifsk_demod_slab / chained launches (ST): this row is the stream from index `origin`
    // on, the loop resumes from the state the call before left (minimodem.c:1079-1088,
    // 1132-1133,1144-1174).  Same rules, same state record as the wavefront engine
    // (demod_wave_kernel<., ., true>): positions inside the kernel are relative to the row;
    // what leaves it (frame starts, episode frame indices, the saved state) counts from the
    // start of the stream.  `rp` is the reference's file position: base + samples_nvalid.
    const bool stateful = ST && rs.d_state != nullptr;
    const bool last_slab = !stateful || rs.final != 0u || ( rs.limit != 0u && !cut );
    uint64_t origin = 0;
    uint32_t frame_base = 0;			// frames emitted by the calls before
    bool resumable = true;
    uint32_t rp = base;
    if constexpr ( ST ) {
	if ( stateful ) {
	    if ( rs.d_origin )
		origin = rs.d_origin[s];
	    const mifsk_stream_state st = rs.d_state[s];
	    if ( st.flags & MIFSK_STATE_FINISHED ) {
		resumable = false;			// nothing more to do for this stream
		if ( rs.append == 0u )
		    status |= st.status & MIFSK_STREAM_ABORTED;
	    } else if ( st.flags & MIFSK_STATE_STARTED ) {
		if ( st.base < origin || st.base - origin > (uint64_t)N || st.rp < st.base ) {
		    status |= MIFSK_STREAM_ABORTED;	// the caller dropped samples the loop still needs
		    resumable = false;
		} else {
		    // (the record was read with vector loads: every field goes through
		    // v_readfirstlane so that the loop state stays in scalar registers)
		    auto u32 = []( uint32_t v ) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
		    auto f32 = [&]( float v ) { return __uint_as_float(u32(__float_as_uint(v))); };
		    base = u32((uint32_t)( st.base - origin ));
		    rp = base + u32((uint32_t)( st.rp - st.base ));
		    advance = u32(st.advance);
		    carrier = ( u32(st.flags) & MIFSK_STATE_CARRIER ) != 0u;
		    carrier_nsamples = ( (uint64_t)u32((uint32_t)( st.carrier_nsamples >> 32 )) << 32 )
				     | u32((uint32_t)st.carrier_nsamples);
		    confidence_total = f32(st.confidence_total);
		    amplitude_total = f32(st.amplitude_total);
		    nframes_decoded = u32(st.nframes_decoded);
		    noconfidence = u32(st.noconfidence);
		    track_amplitude = f32(st.track_amplitude);
		    peak_confidence = f32(st.peak_confidence);
		    ep_first = u32(st.ep_first);
		    ep_b_mark = u32(st.ep_b_mark);
		    frame_base = u32((uint32_t)st.nframes_total);
		    if ( rs.append ) {
			// the outputs continue where the call before stopped instead of at index 0
			n_out_frames = frame_base;
			n_out_bytes = u32(st.nbytes_total);
			n_out_eps = u32(st.nepisodes_total);
			status = u32(st.status);
			frame_base = 0u;
		    }
		}
	    }
	}
    }
    // With more of the stream to come, a pass of the loop is run only when a whole
    // samplebuf beyond its cursor is in the row: then every refill is a full half buffer
    // and every sample a search can read is the stream's, exactly as in a single call.
    // Lattice frames are accepted up to that horizon (N_lat), the general path stops at it.
    const uint32_t bufsize = ST ? rs.bufsize : 0u, half = bufsize / 2u;
    const uint32_t N_lat = last_slab ? N : ( N > bufsize ? N - bufsize : 0u );
    bool paused = false;

    uint32_t cyc_bulk = 0, cyc_general = 0, cyc_restart = 0, cyc_s1 = 0, cyc_s2 = 0, cyc_dpp = 0;
    const uint32_t t_start = MIFSK_CLOCK();
#ifdef MIFSK_PROFILE
    const uint32_t t_wall0 = (uint32_t)wall_clock64();
#endif

    const ZigZag zc0(cfg, 0u), zc1(cfg, 1u), zf0(cfg, 2u), zf1(cfg, 3u);

    for (;;) {
	if ( ST && !resumable )
	    break;
	// A lattice frame that failed the replay below and goes through the general path: its
	// coarse search is known -- the first try is that scored frame, and a first try at or above
	// the limit ends the search (fsk.c:492,499) -- so the iteration is entered behind it.
	bool fwd = false;
	ScanResult fwd_sr;
	fwd_sr.conf = 0.0f; fwd_sr.ampl = 0.0f; fwd_sr.bits = 0; fwd_sr.start = 0;
	// ------------------------------------------------------------------
	// Bulk acceptance of lattice frames.  While carrier is held and the
	// cursor lands on the lattice, the reference's iteration for frame k
	// reduces to: first try wins the coarse scan (c >= limit), no refine
	// (c >= 0.75 peak), no squelch (a >= 0.25 track, c > threshold).  Those
	// predicates and the f32 state recurrences are replayed here in frame
	// order from the scored (confidence, amplitude) pairs; the first frame
	// that fails any of them falls through to the general path below.
	// ------------------------------------------------------------------
	if ( carrier && advance && ( ST ? ( base <= N_lat && advance <= N_lat - base ) : advance <= N - base ) ) {
	    const uint32_t t_bulk = MIFSK_CLOCK();
	    const uint32_t first = h_first1;
	    const uint32_t nb = base + advance;		// cursor of the next iteration
	    const uint32_t p = nb + first;
	    const uint32_t e0 = ctx.lattice_lookup(p);
	    bool progressed = false, broke = false;
	    if ( e0 != ~0u ) {
		uint32_t K = ctx.lat_n - e0;
		// frame k sits at cursor nb + k*lock_advance and needs expect_nsamples from there
		const uint32_t fn = h_fn;
		const uint32_t la = ctx.h_la;
		const uint32_t room = N_lat - nb >= h_expect
				    ? udiv_magic(N_lat - nb - h_expect, la, ctx.h_la_magic) + 1u : 0u;
		K = K < room ? K : room;
		// this lane's candidate (lane k <-> entry e0 + k)
		const bool have = lane < K;
		const float cv = have ? lds->c_conf[e0 + lane] : 0.0f;
		const float av = have ? lds->c_ampl[e0 + lane] : 0.0f;
		// Replay the f32 state recurrences over all K candidates without
		// branching, as a lane scan.  Lane k computes the state AFTER frame k,
		//   A_k = step(A_{k-1}, c_k, a_k),   A_{-1} = the current state,
		// by applying step() to its lower neighbour's state (DPP wave_shr:1).
		// Lane 0 has no lower neighbour: with bound_ctrl off the DPP
		// instructions leave its destination alone, so it keeps the A_0 it
		// is seeded with.  After j steps lanes 0..j are final; K - 1 steps
		// settle lanes 0..K-1 (further steps change nothing).  Same f32
		// operations in the same order as the scalar loop.
		const uint32_t t_dpp = MIFSK_CLOCK();
		float xt = ( track_amplitude + av ) / 2.0f;		// minimodem.c:1391
		float xpk = peak_confidence < cv ? cv : peak_confidence;	// :1392-1393
		float xsc = confidence_total + cv;			// :1397-1398
		float xsa = amplitude_total + av;
		// ... and the state before frame `lane`: the lower neighbour's "after";
		// lane 0 keeps what these are seeded with, the current state
		float my_t = track_amplitude, my_pk = peak_confidence;
		float my_sc = confidence_total, my_sa = amplitude_total;
		replay_scan_asm(xt, xpk, xsc, xsa, my_t, my_pk, my_sc, my_sa, cv, av, K, o.eps != nullptr);
		cyc_dpp += MIFSK_CLOCK() - t_dpp;
		const float t = xt, pk = xpk, sc = xsc, sa = xsa;	// state after frame `lane`
		const bool ok = have
		    && cv > 0.0f && cv >= h_limit		// fsk.c:492,499: first try ends the scan
		    && !( cv < my_pk * 0.75f )			// minimodem.c:1278
		    && !( av < my_t * 0.25f )			// minimodem.c:1286
		    && !( cv <= h_thr );			// minimodem.c:1292
		const unsigned long long bad = __ballot(have && !ok);
		const uint32_t n = bad ? (uint32_t)__ffsll((long long)bad) - 1u : K;
		float track, peak, ctot, atot;
		if ( n == K ) {
		    track = lane_bcast(t, K - 1u);
		    peak = lane_bcast(pk, K - 1u);
		    ctot = lane_bcast(sc, K - 1u);
		    atot = lane_bcast(sa, K - 1u);
		} else {
		    track = lane_bcast(my_t, n);
		    peak = lane_bcast(my_pk, n);
		    ctot = lane_bcast(my_sc, n);
		    atot = lane_bcast(my_sa, n);
		}
		ctx.run += n;
		// (bad != 0 <=> n < K.)  With a search step of one sample a "refine"
		// is a flag and no search (minimodem.c:1357): the cursor stays on the
		// lattice and the batch in flight stays good -- 12000 baud lives there
		// (where this wave can run the rescan alone the batch is dropped
		// later, and only if the rescan moves the cursor off the lattice)
		if ( NQ == 0 && bad != 0ULL && cfg.try_step[1] > 1u )
		    ctx.give_up();
		if ( n < K ) {			// the lattice broke here: remember how long it held
		    ctx.spec = ctx.run < ctx.spec_floor ? ctx.spec_floor
			     : ( ctx.run < ctx.lat_batch ? ctx.run : ctx.lat_batch );
		    ctx.cold = ctx.run ? 0u : ctx.cold + 1u;
		    ctx.run = 0;
		} else if ( e0 + K == ctx.lat_n ) {	// a whole batch held: speculate further
		    ctx.spec = 2u * ctx.spec < ctx.lat_batch ? 2u * ctx.spec : ctx.lat_batch;
		}
		if ( n ) {
		    // outputs of frames 0..n-1, one lane each
		    const bool mine = lane < n;
		    uint64_t db = 0;
		    bool suppressed = false;
		    if ( mine ) {
			db = ( lds->c_bits[e0 + lane] >> h_dshift ) & h_dmask;	// data_bits_of()
			if ( h_msb )
			    db = bit_reverse(db, cfg.n_data_bits);
			suppressed = h_rx_sync && db == h_sync_byte;
		    }
		    const unsigned long long keep = __ballot(mine && !suppressed);

		    if ( mine ) {
			const uint32_t fi = n_out_frames + lane;
			if ( fi < o.fcap ) {
			    if ( o.bits )
				o.bits[fi] = db;
			    if ( o.frames ) {
				mifsk_frame f;
				f.bits = db;
				f.start = origin + (uint64_t)nb + (uint64_t)lane * la + first;
				f.confidence = cv;
				f.amplitude = av;
				f.flags = suppressed ? MIFSK_FRAME_SYNC : 0u;
				f.reserved = 0;
				o.frames[fi] = f;
			    }
			}
			if ( !suppressed && o.bytes ) {
			    const uint32_t bi = n_out_bytes
				+ (uint32_t)__popcll(keep & ( ( 1ULL << lane ) - 1ULL ));
			    if ( bi < o.fcap )
				o.bytes[bi] = (uint8_t)( db & 0xFFu );
			}
		    }
		    n_out_frames += n;
		    n_out_bytes += (uint32_t)__popcll(keep);
		    // state after n trivially accepted frames: each advanced the
		    // cursor by lock_advance = first + fn - overscan and added
		    // fn + first - overscan to carrier_nsamples (minimodem.c:1324-1330,1407)
		    track_amplitude = track;
		    peak_confidence = peak;
		    confidence_total = ctot;
		    amplitude_total = atot;
		    nframes_decoded += n;
		    noconfidence = 0;
		    carrier_nsamples += (uint64_t)n * ( fn + first - h_overscan );
		    base = nb + ( n - 1u ) * la;
		    advance = la;
		    if constexpr ( ST ) {
			// (one refill per iteration whenever fewer than half a buffer is valid:
			// after n iterations the file position is the first rp + m * half that
			// leaves at least half a buffer beyond the cursor)
			// -- in closed form (a loop here leaves hipcc with a loop-carried
			// value it will not keep in a scalar register)
			if ( rp < base + half && rp < N ) {
			    const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane(
				(int)( ( base + half - rp + half - 1u ) / half ));
			    const uint64_t r2 = (uint64_t)rp + (uint64_t)m * half;
			    rp = r2 > (uint64_t)N ? N : (uint32_t)r2;
			}
		    }
		    ctx.bump(MIFSK_CNT_BULK_FRAMES, n);
		    progressed = true;
		}
		if ( n < K ) {
		    // Frame n is the next iteration's, and it is not a trivial one: straight to the
		    // general path (a second replay would only find n = 0 again), with the coarse
		    // search's answer where the first try decides it.
		    broke = true;
		    const float cn = lane_bcast(cv, n);
		    if ( cn > 0.0f && cn >= h_limit ) {
			fwd = true;
			fwd_sr.conf = cn;
			fwd_sr.ampl = lane_bcast(av, n);
			fwd_sr.bits = lds->c_bits[e0 + n];
			fwd_sr.start = first;
		    }
		}
	    } else if ( ctx.inflight && ctx.inflight_anchor == p ) {
		// the cursor has walked onto the batch the workers are finishing
		ctx.lattice_advance();
		progressed = true;
	    }
	    cyc_bulk += MIFSK_CLOCK() - t_bulk;
	    if ( progressed && !broke )
		continue;
	}

	if constexpr ( ST ) {
	    // the reference's buffer arithmetic in full (the state carries the file position):
	    // minimodem.c:1146-1176,1229, statement for statement what the wavefront engine runs
	    if ( !last_slab && (uint64_t)base + advance + bufsize > (uint64_t)N ) {
		paused = true;			// it could read beyond the row: the next slab resumes here
		break;
	    }
	    if ( advance == bufsize ) {		// :1146-1149: samples_nvalid = 0
		base += advance;
		rp = base;
		advance = 0;
	    }
	    if ( base > rp )
		break;
	    if ( advance ) {			// :1150-1156
		if ( advance > rp - base )
		    break;
		base += advance;
		advance = 0;
	    }
	    if ( rp - base < half )		// :1158-1174
		rp += N - rp < half ? N - rp : half;
	    const uint32_t nvalid = rp - base;
	    if ( nvalid == 0 || nvalid < cfg.expect_nsamples )	// :1176,1229
		break;
	} else {
	    // minimodem.c:1150-1156,1176,1229 under flat addressing
	    if ( advance ) {
		if ( advance > N - base )
		    break;
		base += advance;
		advance = 0;
	    }
	    const uint32_t avail = N - base;
	    if ( avail == 0 || avail < h_expect )
		break;
	}
	ctx.bump(MIFSK_CNT_ITERATIONS);
	const uint32_t t_gen = MIFSK_CLOCK();

	const uint32_t ci = carrier ? 1u : 0u;
	const uint32_t try_max = carrier ? h_try_max1 : h_try_max0;
	const uint32_t try_step = carrier ? h_try_step1 : h_try_step0;
	const uint32_t try_first = carrier ? h_first1 : h_first0;

	const uint32_t t_s1 = MIFSK_CLOCK();
	ScanResult sr;						// minimodem.c:1265-1274
	if ( fwd ) {
	    sr = fwd_sr;		// (what scan() returns for a scored first try at or above the limit)
	    ctx.hit_base = base;
	    ctx.bump(MIFSK_CNT_CACHE_HITS);
	} else {
	    sr = ctx.scan(base, carrier ? zc1 : zc0, try_first, h_limit, carrier ? 0u : 1u);
	}
	cyc_s1 += MIFSK_CLOCK() - t_s1;
	float confidence = sr.conf;
	float amplitude = sr.ampl;
	uint64_t bits = sr.bits;
	uint32_t frame_start = sr.start;

	bool refine = false;
	if ( confidence < peak_confidence * 0.75f ) {		// minimodem.c:1278-1282
	    refine = true;
	    peak_confidence = 0.0f;
	}
	if ( amplitude < track_amplitude * 0.25f )		// minimodem.c:1286-1288
	    confidence = 0.0f;

	if ( confidence <= h_thr ) {				// minimodem.c:1292-1321
	    if ( ++noconfidence > 20u ) {
		    if ( carrier ) {

			if ( t0 && o.eps && n_out_eps < o.ecap ) {
			    mifsk_episode e;
			e.carrier_nsamples = carrier_nsamples;
			e.first_frame = ep_first;
			e.nframes = nframes_decoded;
			e.confidence_total = confidence_total;
			e.amplitude_total = amplitude_total;
			e.end_reason = 1;
			e.b_mark = cfg.b_mark;
			o.eps[n_out_eps] = e;
		    }
		    n_out_eps++;
		    carrier = false;
		    carrier_nsamples = 0;
		    confidence_total = 0.0f;
		    amplitude_total = 0.0f;
		    nframes_decoded = 0;
		    track_amplitude = 0.0f;
		    if ( cfg.rx_one )
			break;
		}
	    }
	    advance = try_max;
	    continue;
	}

	carrier_nsamples += h_fn;				// minimodem.c:1324
	uint32_t flags = 0;
	if ( carrier ) {
	    carrier_nsamples += frame_start;			// minimodem.c:1329-1330
	    carrier_nsamples -= h_overscan;
	} else {
	    carrier = true;					// minimodem.c:1350-1353
	    refine = true;
	    flags |= MIFSK_FRAME_ACQUIRE;
	    ep_first = frame_base + n_out_frames;
	    ep_b_mark = cfg.b_mark;
	}

	if ( refine && confidence < INFINITY && try_step > 1u ) {	// minimodem.c:1357-1389
	    // `carrier` is already set: an acquiring frame is re-searched with
	    // the data string over the no-carrier range (minimodem.c:1378)
	    const uint32_t t_s2 = MIFSK_CLOCK();
	    ScanResult s2;
	    bool alone = false;
	    if constexpr ( NQ > 0 ) {
		if ( ci && ctx.hit_base == base )
		    alone = ctx.template solo_fine<NQ>(base, zf1, try_first, sr, s2);
	    }
	    if ( !alone )
		s2 = ctx.scan(base, ci ? zf1 : zf0, try_first, INFINITY, 0u);
	    cyc_s2 += MIFSK_CLOCK() - t_s2;
	    flags |= MIFSK_FRAME_REFINED;
	    ctx.bump(MIFSK_CNT_REFINES);
	    if ( s2.conf > confidence ) {
		bits = s2.bits;
		amplitude = s2.ampl;
		frame_start = s2.start;
	    }
	}

	advance = frame_start + h_fn - h_overscan;			// minimodem.c:1407

	// (first of all, so that the workers' first round on a moved lattice is under way while
	// this wave does the frame's bookkeeping and outputs)
	// carrier is held: (re)start the lattice pipeline where the next
	// iteration will search first, unless a matching batch is already in
	// flight or already scored
	if ( ctx.lat_batch && advance <= N - base ) {
	    const uint32_t p = base + advance + h_first1;
	    const bool cached = ctx.lattice_lookup(p) != ~0u;
	    const uint32_t t_ls = MIFSK_CLOCK();
	    if ( ctx.pause ) {
		ctx.pause--;			// the lattice kept missing: plain searches for a while
	    } else if ( !cached && !( ctx.inflight && ctx.inflight_anchor == p ) ) {
		ctx.give_up();			// (a batch left running by a rescan that moved the cursor)
		if ( ctx.cold >= 4u ) {
		    ctx.cold = 0;
		    ctx.pause = 32;
		} else {
		    ctx.lattice_start(p);
		}
	    }
	    cyc_restart += MIFSK_CLOCK() - t_ls;
	}

	track_amplitude = ( track_amplitude + amplitude ) / 2.0f;	// minimodem.c:1391-1400
	if ( peak_confidence < confidence )
	    peak_confidence = confidence;
	confidence_total += confidence;
	amplitude_total += amplitude;
	nframes_decoded++;
	noconfidence = 0;

	bits = ( bits >> h_dshift ) & h_dmask;				// data_bits_of(), minimodem.c:1415-1428
	if ( h_msb )
	    bits = bit_reverse(bits, cfg.n_data_bits);
	const bool suppressed = h_rx_sync && bits == h_sync_byte;	// minimodem.c:1436-1439
	if ( suppressed )
	    flags |= MIFSK_FRAME_SYNC;

	if ( t0 ) {

	    if ( n_out_frames < o.fcap ) {
		if ( o.bits )
		    o.bits[n_out_frames] = bits;
		if ( o.frames ) {
		    mifsk_frame f;
		    f.bits = bits;
		    f.start = origin + (uint64_t)base + frame_start;
		    f.confidence = confidence;
		    f.amplitude = amplitude;
		    f.flags = flags;
		    f.reserved = 0;
		    o.frames[n_out_frames] = f;
		}
	    }
	    if ( !suppressed && o.bytes && n_out_bytes < o.fcap )
		o.bytes[n_out_bytes] = (uint8_t)( bits & 0xFFu );
	}
	n_out_frames++;
	if ( !suppressed )
	    n_out_bytes++;

	cyc_general += MIFSK_CLOCK() - t_gen;
    }


    if constexpr ( ST ) {
	if ( stateful && t0 && !( status & MIFSK_STREAM_ABORTED ) && resumable ) {
	    mifsk_stream_state st;
	    st.base = origin + base;
	    st.rp = origin + rp;
	    st.carrier_nsamples = carrier_nsamples;
	    st.nframes_total = (uint64_t)frame_base + n_out_frames;
	    st.advance = advance;
	    st.flags = MIFSK_STATE_STARTED | ( carrier ? MIFSK_STATE_CARRIER : 0u )
		     | ( paused ? 0u : MIFSK_STATE_FINISHED );
	    st.confidence_total = confidence_total;
	    st.amplitude_total = amplitude_total;
	    st.nframes_decoded = nframes_decoded;
	    st.noconfidence = noconfidence;
	    st.track_amplitude = track_amplitude;
	    st.peak_confidence = peak_confidence;
	    // (--auto-carrier runs on the wavefront engine: a record that engine started keeps what
	    // it holds there, a record started here says "no band held")
	    const bool was_started = ( rs.d_state[s].flags & MIFSK_STATE_STARTED ) != 0u;
	    st.carrier_band = was_started ? rs.d_state[s].carrier_band : -1;
	    st.first_band = was_started ? rs.d_state[s].first_band : -1;
	    st.b_mark = cfg.b_mark;
	    st.ep_b_mark = ep_b_mark;
	    st.ep_first = ep_first;
	    const uint32_t b0 = rs.append ? 0u : rs.d_state[s].nbytes_total, e0 = rs.append ? 0u : rs.d_state[s].nepisodes_total;
	    st.nbytes_total = b0 + n_out_bytes;
	    st.nepisodes_total = e0 + n_out_eps;
	    st.status = rs.d_state[s].status | status;
	    rs.d_state[s] = st;
	} else if ( stateful && t0 && ( status & MIFSK_STREAM_ABORTED ) ) {
	    mifsk_stream_state st = rs.d_state[s];
	    st.flags |= MIFSK_STATE_STARTED | MIFSK_STATE_FINISHED;
	    st.status |= MIFSK_STREAM_ABORTED;
	    rs.d_state[s] = st;
	}
    }
    if ( carrier && !( ST && ( paused || !resumable ) ) ) {	// minimodem.c:1469-1474
	if ( t0 && o.eps && n_out_eps < o.ecap ) {
	    mifsk_episode e;
	    e.carrier_nsamples = carrier_nsamples;
	    e.first_frame = ep_first;
	    e.nframes = nframes_decoded;
	    e.confidence_total = confidence_total;
	    e.amplitude_total = amplitude_total;
	    e.end_reason = 2;
	    e.b_mark = cfg.b_mark;
	    o.eps[n_out_eps] = e;
	}
	n_out_eps++;
    }
    // (append: a stream the call before finished keeps the outputs that call wrote)
    if ( t0 && !( ST && rs.append != 0u && !resumable && status == 0u ) ) {
	if ( n_out_frames > o.fcap && ( o.bits || o.frames || o.bytes ) )
	    status |= MIFSK_STREAM_FRAMES_TRUNCATED;
	if ( n_out_eps > o.ecap && o.eps )
	    status |= MIFSK_STREAM_EPISODES_TRUNCATED;
	const KernArgs<DemodArgs>::ptr a = KernArgs<DemodArgs>::here();
	if ( a->io.d_nframes ) a->io.d_nframes[s] = n_out_frames;
	if ( a->io.d_nbytes ) a->io.d_nbytes[s] = n_out_bytes;
	if ( a->io.d_nepisodes ) a->io.d_nepisodes[s] = n_out_eps;
	if ( a->io.d_status ) a->io.d_status[s] = status;
	if ( a->io.d_counters ) {
	    uint64_t *c = a->io.d_counters + (size_t)s * MIFSK_NCOUNTERS;
	    for ( int i = 0; i < MIFSK_NCOUNTERS; i++ )
		c[i] = lds->cnt[i];		// (event counts; the cycle totals below are profile-build only)
	    c[MIFSK_CNT_CYC_TOTAL] = MIFSK_CLOCK() - t_start;
	    c[MIFSK_CNT_CYC_PARALLEL] = ctx.cyc_par;
	    c[MIFSK_CNT_CYC_WAIT] = ctx.cyc_wait;
	    c[MIFSK_CNT_CYC_CONFIDENCE] = ctx.cyc_conf;
	    c[MIFSK_CNT_CYC_BULK] = cyc_bulk;
	    c[16] = cyc_general;	// general iterations that produced a frame (profile build)
	    // (profile build: solo_fine's phases ride in the high words)
	    c[17] = cyc_restart | ( (uint64_t)ctx.cyc_sf_load << 32 );
	    c[18] = cyc_s1 | ( (uint64_t)ctx.cyc_sf_corr << 32 );
	    c[19] = cyc_s2 | ( (uint64_t)ctx.cyc_sf_score << 32 );
	    c[20] = cyc_dpp;
	    c[21] = ctx.cyc_scan_wait | ( (uint64_t)ctx.cyc_sf_sel << 32 );
#ifdef MIFSK_PROFILE
	    // when this stream started and ended on the chip-wide 100 MHz clock
	    c[23] = ( wall_clock64() & 0xFFFFFFFFull ) | ( (uint64_t)t_wall0 << 32 );
	    // HW_REG_XCC_ID (gfx940+)
	    c[22] = (uint64_t)( __builtin_amdgcn_s_getreg(( 3 << 11 ) | 20) & 15 ) << 32;
#endif
	}
    }
    if ( t0 )
	ctx.next_cmd()->op = CMD_EXIT;		// (whatever was or was not written above)
    lds_barrier();				// releases the workers with "exit"
}


// what a worker reads of the configuration in every round, read once per kernel (keep_scalar),
// and of the command, read once per command: each used to be an s_load / ds_read round trip
// with its wait at the top of every round
struct WorkerHot {
    uint32_t	n_bits, B, la;
    bool	grid;
    float	magscalar;
    uint32_t	buf, total, anchor0;	// the LATTICE command being worked on
};

template <int NQ>
__device__ __forceinline__ void worker_lattice_linear( const WorkerHot &h, const double *__restrict__ tw,
	StreamLds *lds, const float *__restrict__ x, uint32_t N,
	uint32_t region_floats, uint32_t lat_frames, uint32_t wkr, uint32_t done, uint32_t win_base,
	uint32_t rel_lane, uint32_t safe_limit,
	float4 (&pbuf)[STAGE_VEC], uint32_t &pref_org4, uint32_t (&wcyc)[6], const TwGroup (&tgr)[3] )
{
    uint32_t lane = threadIdx.x & 63u;
    asm volatile("" : "+v"(lane));	// per-round values derived from it are recomputed, not spilled
    const uint32_t n_bits = h.n_bits, B = h.B;
    if ( h.grid )
	rel_lane = ( wkr * 64u + lane ) * B;	// (one multiply: cheaper than keeping it live)
    const uint32_t buf = h.buf;
    float *region = lds->slab + (size_t)wkr * region_floats;
    // A batch is scored by the master in one go but correlated here in ROUNDS
    // of lat_frames frames (what the regions hold); the lattice simply
    // continues from one round into the next, and so does the prefetch.
    const uint32_t total = h.total;
    {
	const uint32_t t_in = MIFSK_CLOCK();
	const uint32_t frames = total - done < lat_frames ? total - done : lat_frames;
	const uint32_t anchor = h.anchor0 + done * h.la;
	// distinct windows of the round (cfg.lat_grid: consecutive frames share one)
	const uint32_t nwin = h.grid ? frames * ( n_bits - 1u ) + 1u : frames * n_bits;
	const uint32_t w = wkr * 64u + lane;
	if ( wkr * 64u >= nwin ) {
	    pref_org4 = 0xFFFFFFFFu;
	    return;				// nothing for this wave this round (uniform)
	}
	const bool active = w < nwin;
	// window start: rel_lane = f * lock_advance + bit_offset[k] for window
	// w = f * n_bits + k is the same in every round (worker_main); idle
	// lanes shadow the first window
	const uint32_t last = nwin - wkr * 64u > 64u ? 63u : nwin - wkr * 64u - 1u;	// uniform
	const uint32_t a1 = anchor + rel_lane;
	const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)a1);
	const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)a1, (int)last) + B;
	const uint32_t a = active ? a1 : lo;
	const uint32_t nvec = ( hi - lo + 3 ) >> 2;

	// Raw loads of one round: 64 * STAGE_VEC consecutive float4 from sample
	// `from`, whatever the round really needs -- no per-lane bounds logic, the
	// addresses differ by immediates.  `safe_limit` (>= N) is how far this
	// row may be over-read without leaving the batch's allocation; a round
	// that would cross it is fetched from sample 0 instead and its data never
	// used (it reaches the end of the stream, so it is re-read by element).
	constexpr uint32_t kRoundFloats = 64u * STAGE_VEC * 4u;
	if ( pref_org4 != lo ) {
	    // nothing usable in flight: fetch this round now
	    const bool ok = lo <= safe_limit - kRoundFloats;	// (the master guarantees safe_limit >= kRoundFloats)
	    const float *pb = x + ( ok ? lo : 0u ) + ( lane << 2 );
#pragma unroll
	    for ( int i = 0; i < STAGE_VEC; i++ ) {
		const float4_u sv = *reinterpret_cast<const float4_u *>(pb + i * 256);
		pbuf[i] = make_float4(sv.x, sv.y, sv.z, sv.w);
	    }
	}
#ifdef MIFSK_PROFILE
	// (profile build: the staging phase split into its wait for the round's loads,
	// the LDS writes, and the issue of the next round's loads)
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
	const uint32_t t_a = MIFSK_CLOCK();
	// Registers -> LDS.  The store address of vector i is a constant 1 KiB
	// further on than that of vector i - 1 (an immediate offset, no address
	// registers).  Vectors beyond nvec are simply not stored.  Only a round
	// that reaches the end of the stream (uniform test) re-reads by element:
	// load4_unaligned() fetched from a clamped address there.
	float *lane_base = region + ( lane << 2 );
	const uint32_t nvec_lane = nvec > lane ? nvec - lane : 0u;	// vector i stored iff 64 i < nvec_lane
	const uint32_t elast = lo + kRoundFloats;
	if ( elast <= N && elast >= lo && region_floats >= kRoundFloats ) {
	    // the region holds a whole round of vectors: store them all, needed or
	    // not (no per-vector predicates: ten plain ds_write_b128)
#pragma unroll
	    for ( int i = 0; i < STAGE_VEC; i++ )
		*reinterpret_cast<float4 *>(lane_base + i * 256) = pbuf[i];
	} else if ( elast <= N && elast >= lo ) {
#pragma unroll
	    for ( int i = 0; i < STAGE_VEC; i++ )
		if ( (uint32_t)( i * 64 ) < nvec_lane )
		    *reinterpret_cast<float4 *>(lane_base + i * 256) = pbuf[i];
	} else {
#pragma unroll
	    for ( int i = 0; i < STAGE_VEC; i++ ) {
		const uint32_t e = lo + ( ( i * 64 + lane ) << 2 );
		float4 sv;
		sv.x = ( e < N ) ? x[e] : 0.0f;
		sv.y = ( e + 1 < N && e + 1 > e ) ? x[e + 1] : 0.0f;
		sv.z = ( e + 2 < N && e + 2 > e ) ? x[e + 2] : 0.0f;
		sv.w = ( e + 3 < N && e + 3 > e ) ? x[e + 3] : 0.0f;
		if ( (uint32_t)( i * 64 ) < nvec_lane )
		    *reinterpret_cast<float4 *>(lane_base + i * 256) = sv;
	    }
	}
	const uint32_t t_b = MIFSK_CLOCK();
	// The same share of the next round (of this batch or the next), assuming
	// the lattice goes on; issued unconditionally and with no control flow
	// after it (see worker_lattice).  (A second, alternating register buffer
	// -- two rounds of look-ahead -- does not fit: measured again in round 4 on the
	// Bell-202 instantiation at its 168-VGPR budget, 33 VGPRs spill into this
	// loop and configs[1] goes 0.465 -> 0.555 ms.  Staging by LDS-DMA instead
	// needs the landing zone in LDS, which four workgroups per CU do not leave;
	// and with one round of look-ahead the memory system already delivers
	// 6.1 TB/s to this access pattern: profiles/r04_history.md.)
	{
	    const uint32_t nlo = lo + lat_frames * h.la;
	    const bool ok = nlo >= lo && nlo <= safe_limit - kRoundFloats;
	    const float *pb = x + ( ok ? nlo : 0u ) + ( lane << 2 );
#pragma unroll
	    for ( int i = 0; i < STAGE_VEC; i++ ) {
		const float4_u sv = *reinterpret_cast<const float4_u *>(pb + i * 256);
		pbuf[i] = make_float4(sv.x, sv.y, sv.z, sv.w);
	    }
	    pref_org4 = ok ? nlo : 0xFFFFFFFFu;
	}
	const uint32_t t_c = MIFSK_CLOCK();
	wave_lds_sync();
	const uint32_t t_mid = MIFSK_CLOCK();
	wcyc[3] += t_a - t_in;		// waiting for the round's samples
	wcyc[4] += t_b - t_a;		// registers -> LDS
	wcyc[5] += t_c - t_b;		// issuing the next round's loads

	double mr = 0.0, mi = 0.0, sr = 0.0, si = 0.0;
	if constexpr ( NQ > 0 ) {
	    // an instantiation for this bit length (B == 4 NQ): the table is
	    // resident in vector registers (three groups of 16 entries) and
	    // broadcast with DPP -- no scalar loads, no SGPR block for the compiler
	    // to spill around
	    double acc[4] = { 0.0, 0.0, 0.0, 0.0 };
	    corr_lds_fixed_halves<NQ>(tgr, region + ( a - lo ), acc);
	    mr = acc[0]; mi = acc[1]; sr = acc[2]; si = acc[3];
	} else {
	    // any bit length (B % 4 == 0): one table group per 16 samples through the
	    // vector cache, broadcast with DPP like the resident table (the last group
	    // is loaded whole -- the region has the slack -- and accumulated up to B)
	    double acc[4] = { 0.0, 0.0, 0.0, 0.0 };
	    corr_lds_stream_lean(tw, region + ( a - lo ), B >> 2, lane, acc);
	    mr = acc[0]; mi = acc[1]; sr = acc[2]; si = acc[3];
	}
	if ( active )
	    lds->mags[buf][win_base + w] = band_mag2(mr, mi, sr, si, h.magscalar);
	wave_lds_sync();			// the region is rewritten by the next round
	const uint32_t t_out = MIFSK_CLOCK();
	wcyc[0] += t_mid - t_in;
	wcyc[1] += t_out - t_mid;
    }
}

// LATTICE, direct variant: any bit length, no LDS.  One lane per bit window as
// everywhere; the lane streams ITS OWN window straight from global memory, 64
// bytes (a group of 16 samples) per step with the next group already in flight,
// twiddles broadcast with DPP from one table group per step (corr_global_stream).
// Nothing has to fit anywhere: what the workgroup engine runs when it is forced
// onto modes whose windows do not fit the linear variant's regions (RTTY, SAME,
// Bell-103) -- the library itself sends those to the wavefront engine.  Same sums
// in the same order as every other correlator.

__device__ __forceinline__ void worker_lattice_direct( const DevCfg &cfg, const double *__restrict__ tw,
	StreamLds *lds, const StreamLds::Cmd *cmd, const float *__restrict__ x, uint32_t N,
	uint32_t lat_frames, uint32_t wkr, uint32_t done, uint32_t win_base, uint32_t (&wcyc)[6] )
{
    const uint32_t t_in = MIFSK_CLOCK();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_bits = cfg.n_bits, B = cfg.bit_nsamples;
    const uint32_t total = cmd->frames;
    const uint32_t frames = total - done < lat_frames ? total - done : lat_frames;
    const uint32_t anchor = cmd->anchor + done * cfg.lock_advance;
    const uint32_t nwin = cfg.lat_grid ? frames * ( n_bits - 1u ) + 1u : frames * n_bits;
    const uint32_t w = wkr * 64u + lane;
    if ( wkr * 64u >= nwin )
	return;					// nothing for this wave this round (uniform)
    const bool active = w < nwin;
    const uint32_t wc = active ? w : wkr * 64u;	// idle lanes shadow the wave's first window
    uint32_t a;
    if ( cfg.lat_grid ) {
	a = anchor + wc * B;
    } else {
	const uint32_t f = udiv_magic(wc, n_bits, cfg.nbits_magic);
	a = anchor + f * cfg.lock_advance + cfg.bit_offset[( wc - f * n_bits ) & 63u];
    }
    const uint32_t Bpad = ( B + 15u ) & ~15u;	// whole groups of 16 are loaded
    double acc[4] = { 0.0, 0.0, 0.0, 0.0 };
    if ( __all(a + Bpad <= N && a + Bpad >= a) ) {
	// every window of the wave (padded to whole groups) lies inside the stream
	corr_global_stream(tw, x + a, B, lane, acc);
    } else {
	// a window reaches the end of the stream: per-sample guarded reads
	// (samples at or beyond N are 0.0), a handful of rounds per stream
	for ( uint32_t n = 0; n < B; n++ ) {
	    const uint32_t idx = a + n;
	    const double xd = (double)( ( idx < N && idx >= a ) ? x[idx] : 0.0f );
	    const double *t = tw + 4 * (size_t)n;
	    acc[0] = fma(xd, t[0], acc[0]);
	    acc[1] = fma(xd, t[1], acc[1]);
	    acc[2] = fma(xd, t[2], acc[2]);
	    acc[3] = fma(xd, t[3], acc[3]);
	}
    }
    if ( active )
	lds->mags[cmd->buf][win_base + w] = band_mag2(acc[0], acc[1], acc[2], acc[3], cfg.magscalar);
    wcyc[1] += MIFSK_CLOCK() - t_in;
}

// The worker waves' whole life.  A real (non-inlined) function on purpose: it
// gets its own register allocation, so the 64 SGPRs of twiddles that the
// correlator wants in flight do not compete with the master's scalar state.
template <bool USE_SLAB, int NQ>
__device__ __forceinline__ void worker_main( const DevCfg *__restrict__ cfgp,
	const double *__restrict__ tw, StreamLds *lds, const float *__restrict__ x, uint32_t N,
	uint32_t slab_cap, uint32_t lat_frames, uint32_t region_floats, uint32_t region_cap,
	uint32_t lat_mode, uint32_t safe_limit, uint64_t *counters )
{
    const DevCfg &cfg = *cfgp;
    const uint32_t wkr = ( threadIdx.x >> 6 ) - 1u;
    // linear LATTICE: start of this lane's window relative to the round's anchor
    uint32_t rel_lane = 0;
    if ( USE_SLAB && lat_mode == LAT_LINEAR ) {
	const uint32_t w = wkr * 64u + ( threadIdx.x & 63u );
	const uint32_t f = udiv_magic(w, cfg.n_bits, cfg.nbits_magic);
	rel_lane = cfg.lat_grid ? w * cfg.bit_nsamples
				: f * cfg.lock_advance + cfg.bit_offset[( w - f * cfg.n_bits ) & 63u];
    }
    // windows a full round writes into mags[]: rounds of one batch sit back to back
    const uint32_t wins_per_round = cfg.lat_grid ? lat_frames * ( cfg.n_bits - 1u ) + 1u
						 : lat_frames * cfg.n_bits;
    float4 pbuf[STAGE_VEC];
#pragma unroll
    for ( int i = 0; i < STAGE_VEC; i++ )
	pbuf[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    // groups 0..2 of the twiddle table, one entry per lane of each 16-lane row
    // (the table is padded to at least 48 entries)
    TwGroup tgr[3];
    if constexpr ( NQ > 0 ) {
#pragma unroll
	for ( int gi = 0; gi < ( NQ + 3 ) / 4; gi++ )
	    tgr[gi] = tw_group_load(tw, (uint32_t)gi, threadIdx.x & 63u);
    }
    uint32_t pref_org4 = 0xFFFFFFFFu;
    uint32_t wcyc[6] = { 0, 0, 0, 0, 0, 0 };
    WorkerHot hot;
    hot.n_bits = keep_scalar(cfg.n_bits);
    hot.B = keep_scalar(cfg.bit_nsamples);
    hot.la = keep_scalar(cfg.lock_advance);
    hot.grid = cfg.lat_grid != 0u;
    hot.magscalar = keep_scalar(cfg.magscalar);
    hot.buf = hot.total = hot.anchor0 = 0u;
    for ( uint32_t seq = 0; ; seq++ ) {
	const uint32_t t_b = MIFSK_CLOCK();
	lds_barrier();			// command number `seq` has been published
	wcyc[2] += MIFSK_CLOCK() - t_b;
	const StreamLds::Cmd *cmd = &lds->cmd[seq & 1u];
	const uint32_t op = cmd->op;
	if ( op == CMD_EXIT )
	    break;
	if ( op == CMD_SCAN ) {
	    scan_part<USE_SLAB>(cfg, tw, lds, cmd, x, N, slab_cap);
	    // nothing is in flight after a SCAN: end the prefetch registers' live
	    // ranges here so that they do not add to the pressure inside it
	    pref_org4 = 0xFFFFFFFFu;
#pragma unroll
	    for ( int i = 0; i < STAGE_VEC; i++ )
		pbuf[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	} else if ( USE_SLAB && op == CMD_LATTICE ) {
	    if ( lat_mode == LAT_LINEAR ) {
		// rounds of lat_frames frames (what the regions hold)
		const uint32_t total = cmd->frames;
		hot.total = keep_scalar((uint32_t)__builtin_amdgcn_readfirstlane((int)total));
		hot.buf = keep_scalar((uint32_t)__builtin_amdgcn_readfirstlane((int)cmd->buf));
		hot.anchor0 = keep_scalar((uint32_t)__builtin_amdgcn_readfirstlane((int)cmd->anchor));
		uint32_t win_base = 0;
		for ( uint32_t done = 0; done < total; done += lat_frames, win_base += wins_per_round ) {
		    // given up by the master (the lattice broke in the batch before):
		    // the issue slots are better spent by the other workgroups' waves
		    if ( lds_peek(&lds->abort) == seq + 1u )
			break;
		    worker_lattice_linear<NQ>(hot, tw, lds, x, N, region_floats, lat_frames,
					  wkr, done, win_base, rel_lane, safe_limit, pbuf, pref_org4, wcyc, tgr);
		}
	    } else if constexpr ( NQ == 0 ) {	// (an instantiation for one bit length is linear by construction)
		const uint32_t total = cmd->frames;
		uint32_t win_base = 0;
		for ( uint32_t done = 0; done < total; done += lat_frames, win_base += wins_per_round ) {
		    if ( lds_peek(&lds->abort) == seq + 1u )
			break;
		    worker_lattice_direct(cfg, tw, lds, cmd, x, N, lat_frames, wkr, done, win_base, wcyc);
		}
	    }
	}
    }
#ifdef MIFSK_PROFILE
    if ( threadIdx.x == 64 && counters ) {
	// (high words: the staging phase's parts -- tools/counters.py splits them)
	counters[13] = wcyc[0] | ( (uint64_t)wcyc[3] << 32 );
	counters[14] = wcyc[1] | ( (uint64_t)wcyc[4] << 32 );
	counters[15] = wcyc[2] | ( (uint64_t)wcyc[5] << 32 );
    }
#else
    (void)counters;
#endif
}

template <bool USE_SLAB, int NQ, int NW, bool ST = false>
__global__ __launch_bounds__(64 * ( NW + 1 ), 3)
void demod_kernel( const DevCfg *__restrict__ cfgp, const double *__restrict__ tw,
	mifsk_demod_io io, uint32_t slab_cap, uint32_t lat_frames, uint32_t lat_rounds,
	uint32_t region_floats, uint32_t region_cap, uint32_t lat_mode, WgResume rs )
{
    StreamLds *lds = reinterpret_cast<StreamLds *>(mifsk_smem);
    const uint32_t base0 = 0u;
    // the configuration lives in device memory (uniform -> scalar loads); it is
    // NOT a by-value kernel argument so that the worker body below can be a real
    // function with its own register allocation
    const DevCfg &cfg = *cfgp;

    if ( threadIdx.x == 0 )
	lds->abort = 0;
    if ( threadIdx.x < MIFSK_NCOUNTERS )
	lds->cnt[threadIdx.x] = 0;
    lds_barrier();

    // How far this stream's row may be over-read (in samples from its start)
    // without leaving the batch: the rows after it, or for the last row its own
    // length.  The linear LATTICE fetches whole rounds of 64 * STAGE_VEC float4
    // with no per-lane bounds logic and needs at least one round of room.
    uint32_t n_own = io.d_nsamples ? io.d_nsamples[blockIdx.x] : io.nsamples;
    if ( io.nstreams > 1 && (size_t)n_own > io.stream_stride )
	n_own = (uint32_t)io.stream_stride;
    const uint32_t n_row = n_own;
    if constexpr ( ST ) {
	// a chained launch sees the first rs.limit samples of a row, master and workers alike
	if ( rs.d_state && rs.limit != 0u && rs.limit < n_own )
	    n_own = rs.limit;
    }
    const uint64_t rows_after = (uint64_t)( io.nstreams - 1 - (int)blockIdx.x ) * io.stream_stride;
    const uint32_t safe_limit = rows_after == 0 ? n_row
			      : rows_after > 0xFFFF0000ull ? 0xFFFF0000u : (uint32_t)rows_after;
    if ( lat_mode == LAT_LINEAR && safe_limit < 64u * STAGE_VEC * 4u )
	lat_frames = 0;

    if ( threadIdx.x < 64 ) {
	// the serial chain is the critical path of the workgroup: let it win
	// issue arbitration against the (throughput-bound) worker waves
	__builtin_amdgcn_s_setprio(3);
	master_loop<USE_SLAB, NQ, ST>(cfg, tw, io, slab_cap, lat_frames * lat_rounds, lat_frames, lat_mode, base0, lds, rs);
    } else {
	worker_main<USE_SLAB, NQ>(cfgp, tw, lds, io.d_samples + (size_t)blockIdx.x * io.stream_stride,
			      n_own, slab_cap, lat_frames, region_floats, region_cap, lat_mode, safe_limit,
			      io.d_counters ? io.d_counters + (size_t)blockIdx.x * MIFSK_NCOUNTERS : nullptr);
    }
}

// ---------------------------------------------------------------------------
// kernel 3: full-spectrum magnitudes for fsk_detect_carrier (fsk.c:543-581)
// one thread per band; window of <= fftsize samples
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256)
void spectrum_kernel( const float *__restrict__ x, uint32_t nsamples,
	const double *__restrict__ cs, uint32_t fftsize, uint32_t nbands,
	float *__restrict__ mags )
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if ( b >= nbands )
	return;
    double re = 0.0, im = 0.0;
    uint32_t k = 0;				// (b * n) mod fftsize, incrementally
    for ( uint32_t n = 0; n < nsamples; n++ ) {
	const double xd = (double)x[n];
	re = fma(xd, cs[2 * (size_t)k], re);
	im = fma(xd, cs[2 * (size_t)k + 1], im);
	k += b;
	if ( k >= fftsize )
	    k -= fftsize;
    }
    const float magscalar = 1.0f / ( (float)nsamples / 2.0f );	// fsk.c:553
    mags[b] = band_mag(re, im, magscalar);
}

// ---------------------------------------------------------------------------
// kernel 4: self-test of band_mag2()'s short square root against the exact sequence
// (mifsk_selftest_sqrt; tests/test_gpu_math.py).  Thread t of `total` evaluates `per_thread`
// sums of squares: pseudo-random pairs of floats (every exponent alike, as hypotf_check.c), and
// -- every other value -- doubles placed within a few hundred units in the last place of a
// float rounding boundary, where the two paths could part.  out[0]: values whose short path
// was taken as safe and differs from the exact one (must be 0); out[1]: values the guard sent
// to the exact sequence; out[2]: values whose UNGUARDED short result differs (what the guard is
// for); out[3]: values evaluated.
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256)
void selftest_sqrt_kernel( uint64_t seed, uint32_t per_thread, unsigned long long *out )
{
    uint64_t st = seed + 0x9E3779B97F4A7C15ull * ( (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1u );
    auto next = [&]() -> uint64_t {
	uint64_t z = ( st += 0x9E3779B97F4A7C15ull );
	z = ( z ^ ( z >> 30 ) ) * 0xBF58476D1CE4E5B9ull;
	z = ( z ^ ( z >> 27 ) ) * 0x94D049BB133111EBull;
	return z ^ ( z >> 31 );
    };
    uint32_t bad = 0, guarded = 0, raw_bad = 0;
    for ( uint32_t i = 0; i < per_thread; i++ ) {
	const uint64_t r = next();
	double s;
	if ( i & 1u ) {
	    // the square of a point within +-600 units (of the double's last place) of the midpoint
	    // between float f and its upper neighbour, rounded to double: sqrt lands about there
	    uint32_t fb = (uint32_t)r & 0x7FFFFFFFu;
	    if ( ( fb & 0x7F800000u ) == 0x7F800000u ) fb ^= 0x00800000u;
	    const float f = __uint_as_float(fb);
	    const float fn = __uint_as_float(fb + 1u);
	    const double mid = 0.5 * ( (double)f + (double)fn );
	    const long long off = (long long)( ( r >> 32 ) % 1201u ) - 600;
	    const double m = __longlong_as_double(__double_as_longlong(mid) + off);
	    s = m * m;
	} else {
	    uint32_t a = (uint32_t)r & 0x7FFFFFFFu, b = (uint32_t)( r >> 32 ) & 0x7FFFFFFFu;
	    // (non-finite inputs now and then: the range test must send them to the exact path)
	    const float fa = __uint_as_float(a), fbv = __uint_as_float(b);
	    s = __builtin_fma((double)fa, (double)fa, (double)fbv * (double)fbv);
	}
	bool unsafe;
	const float quick = (float)sqrt_newton1(s, unsafe);
	const float exact = (float)sqrt_sumsq(s);
	const bool differ = __float_as_uint(quick) != __float_as_uint(exact) && !( quick != quick && exact != exact );
	if ( unsafe )
	    guarded++;
	else if ( differ )
	    bad++;
	if ( differ )
	    raw_bad++;
    }
    atomicAdd(&out[0], (unsigned long long)bad);
    atomicAdd(&out[1], (unsigned long long)guarded);
    atomicAdd(&out[2], (unsigned long long)raw_bad);
    atomicAdd(&out[3], (unsigned long long)per_thread);
}

int launch_selftest_sqrt( uint64_t seed, uint32_t blocks, uint32_t per_thread, unsigned long long *d_out, void *stream )
{
    hipLaunchKernelGGL(selftest_sqrt_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, seed, per_thread, d_out);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------

static inline int hip_rc( hipError_t e )
{
    return e == hipSuccess ? 0 : -5 /* -EIO */;
}

int launch_find_frame_batch( const DevCfg &cfg, const DevCfg *d_cfg, const double *d_tw,
	const float *d_samples, const mifsk_search *d_problems,
	mifsk_search_result *d_results, int nproblems, void *stream )
{
    if ( nproblems <= 0 )
	return 0;
    hipLaunchKernelGGL(find_frame_kernel, dim3((unsigned)nproblems), dim3(64), 0,
		       (hipStream_t)stream, d_cfg, d_tw, d_samples, d_problems, d_results);
    return hip_rc(hipGetLastError());
}

static constexpr size_t kLdsHeader = offsetof(StreamLds, slab);
static constexpr size_t kLdsPerCu = 160 * 1024;

// `nworkers` = 2 asks for the Bell-202 instantiation; kNotBell202 if the
// configuration does not end up there (the caller then plans with three)
static constexpr int kNotBell202 = -100000;

static int launch_with_workers( const DevCfg &cfg, const DevCfg *d_cfg, const double *d_tw,
	const mifsk_demod_io &io, void *stream, LaunchInfo *plan_only, const uint32_t nworkers,
	const WgHostArgs *wh )
{
    const uint32_t B = cfg.bit_nsamples;
    const uint32_t lat_lanes = nworkers * 64u;	// bit windows per lattice round
    const uint32_t block = 64u * ( nworkers + 1u );
    // samples one search must see at once
    const uint32_t reach = ( cfg.try_max[0] > cfg.try_max[1] ? cfg.try_max[0] : cfg.try_max[1] )
			 + cfg.last_reach + 8;
    auto floats_for = [&]( uint32_t nsamp ) -> size_t {
	return ( (size_t)nsamp + (size_t)( nsamp / B + 2 ) * cfg.skew + 8 + 3 ) & ~(size_t)3;
    };
    // LDS budget: 4 workgroups per CU when the stream count can use them
    const size_t budget_small = kLdsPerCu / 4 - 64;

    // LATTICE geometry.  A round is as many frames as fill the worker lanes (64 per worker)
    // with distinct bit windows.  LINEAR workers stage their 64 windows' span in
    // a private LDS region (fastest; needs bit length, offsets and frame step
    // in multiples of 4 samples and the span to fit ten 16-byte loads per
    // lane); otherwise DIRECT workers stream each window from global memory.
    const uint32_t frames_max = cfg.lat_grid ? ( lat_lanes - 1u ) / ( cfg.n_bits - 1u )
					     : lat_lanes / cfg.n_bits;
    uint32_t lat_frames = frames_max > P_CAP ? P_CAP : frames_max;
    auto wins_in = [&]( uint32_t frames ) -> uint32_t {
	return cfg.lat_grid ? frames * ( cfg.n_bits - 1u ) + 1u : frames * cfg.n_bits;
    };
    // window starts must not decrease in window order (the workers take the
    // wave's span from its first and last lane)
    bool ordered = true;
    for ( uint32_t w = 1; w < lat_frames * cfg.n_bits; w++ ) {
	const uint32_t a0 = ( ( w - 1 ) / cfg.n_bits ) * cfg.lock_advance + cfg.bit_offset[( w - 1 ) % cfg.n_bits];
	const uint32_t a1 = ( w / cfg.n_bits ) * cfg.lock_advance + cfg.bit_offset[w % cfg.n_bits];
	ordered = ordered && a1 >= a0;
    }
    uint32_t lat_mode = lat_frames ? LAT_DIRECT : LAT_NONE;
    uint32_t region_cap = 0;
    size_t region_floats = 0;
    if ( lat_frames && cfg.lat_linear && ordered ) {
	// span of the widest wave: 64 windows (or all of them), plus what the
	// group-wise correlator (corr_lds_stream: whole groups of 16 samples) loads
	// beyond the last window -- nothing in the Bell-202 instantiation, whose
	// resident-table correlator reads the window and no more
	const uint32_t over = ( nworkers == 2u && B == 40u ) ? 0u : ( 16u - B % 16u ) % 16u;
	uint32_t span = 0;
	if ( cfg.lat_grid ) {
	    const uint32_t nwin = wins_in(lat_frames);
	    span = ( nwin < 64u ? nwin : 64u ) * B + over;
	} else {
	    const uint32_t nwin = lat_frames * cfg.n_bits;
	    for ( uint32_t w0 = 0; w0 < nwin; w0 += 64 ) {
		const uint32_t wl = w0 + 63 < nwin ? w0 + 63 : nwin - 1;
		const uint32_t lo = ( w0 / cfg.n_bits ) * cfg.lock_advance + cfg.bit_offset[w0 % cfg.n_bits];
		const uint32_t hi = ( wl / cfg.n_bits ) * cfg.lock_advance + cfg.bit_offset[wl % cfg.n_bits] + B;
		span = hi - lo > span ? hi - lo : span;
	    }
	    span += over;
	}
	region_cap = ( span + 3 ) & ~3u;
	region_floats = floats_for(region_cap);
	if ( region_cap <= 64u * STAGE_VEC * 4u
		&& kLdsHeader + nworkers * region_floats * 4 <= budget_small
		&& nworkers * region_floats >= floats_for(reach + 4) )
	    lat_mode = LAT_LINEAR;
    }

    // the master scores `lat_rounds` rounds at once: halves the per-frame cost
    // of everything that is paid per batch -- the confidence pass, the barrier,
    // the command hand-off
    uint32_t lat_rounds = 1;
    if ( lat_frames ) {
	lat_rounds = 2;
	if ( const char *e = experiment_env("MIFSK_LAT_ROUNDS") )	// experiments only
	    lat_rounds = (uint32_t)std::atoi(e) < 1u ? 1u : (uint32_t)std::atoi(e);
	while ( lat_rounds > 1 && ( lat_frames * lat_rounds > P_CAP
				    || wins_in(lat_frames) * lat_rounds > W_CAP ) )
	    lat_rounds--;
    }

    uint32_t slab_cap = 0;
    size_t slab_floats = 0;
    bool use_slab = true;
    if ( lat_mode == LAT_LINEAR ) {
	slab_floats = nworkers * region_floats;
	// samples the whole slab holds in SCAN mode
	size_t ns = slab_floats * B / ( B + cfg.skew );
	ns = ns > 16 ? ns - 16 : 0;
	slab_cap = (uint32_t)( ns & ~(size_t)3 );
    } else {
	// no regions: the slab serves SCAN only; take what one search needs
	region_cap = 0;
	region_floats = 0;
	const size_t need = kLdsHeader + floats_for(reach + 4) * 4;
	if ( need <= kLdsPerCu - 1024 ) {
	    slab_cap = ( reach + 4 + 3 ) & ~3u;
	    slab_floats = floats_for(slab_cap);
	} else {
	    use_slab = false;	// e.g. 0.5 baud: windows of 96000 samples
	    lat_mode = LAT_NONE;
	    lat_frames = 0;
	}
    }

    hipStream_t st = (hipStream_t)stream;
    const bool bell202 = nworkers == 2u && use_slab && lat_mode == LAT_LINEAR && B == 40u;
    if ( nworkers == 2u && !bell202 )
	return kNotBell202;
    const size_t lds_all = use_slab ? kLdsHeader + slab_floats * 4 : kLdsHeader + 16;
    // Chained launches (as in launch_demod_wave): the batch cut into G groups of
    // streams x K time chunks, each (group, chunk) its own grid of the RESUMABLE instantiation on
    // the group's stream.  The mechanism is the wavefront engine's and gives the single launch's
    // results bit for bit (tests/test_gpu_chain.py) -- but this engine's library default is ONE
    // launch at every batch size: its streams are short chains (0.45 ms for 10 s of Bell-202), the
    // dispatcher refills a finished workgroup's slot with the next stream anyway, and every chunk
    // restarts with a search and a pipeline fill.  Measured (tools/gpu/wg_chain_sizes.py,
    // profiles/r04_history.md): 1536 / 3000 / 5000 streams 0.825 / 1.43 / 2.16 ms in one launch,
    // 0.84-0.90 / 1.43-1.52 / 2.24-2.38 chained (2x2 ... 3x3).  MIFSK_CHAIN forces a cut
    // (experiments and tests).
    uint32_t chain_g = 0, chain_k = 0;
    if ( wh ) {
	const bool allowed = ( plan_only ? wh->chain_ok : wh->chain != nullptr ) && !wh->d_state
			  && !io.d_counters && io.nstreams > 0;
	if ( const char *e = experiment_env("MIFSK_CHAIN") ) {	// experiments and tests only: "G,K", any batch
	    int a = 0, b = 0;
	    if ( allowed && std::sscanf(e, "%d,%d", &a, &b) == 2 ) {
		chain_g = (uint32_t)( a < 0 ? 0 : a );
		chain_k = (uint32_t)( b < 0 ? 0 : b );
	    }
	}
	if ( chain_g > (uint32_t)WaveChain::kMaxGroups ) chain_g = (uint32_t)WaveChain::kMaxGroups;
	if ( chain_g > (uint32_t)io.nstreams ) chain_g = (uint32_t)io.nstreams;
	// (the cut is made by io.nsamples: with per-stream lengths only -- io.nsamples == 0 -- a
	// limit of 0 would mean "all samples" to every chunk but the last; such a batch is not cut)
	if ( chain_g < 1u || chain_k < 2u || io.nsamples == 0u )
	    chain_g = chain_k = 0u;
    }
    const bool resumable = ( wh && wh->d_state ) || chain_g;
    if ( plan_only ) {
	plan_only->kernel = !use_slab ? ( resumable ? "mifsk::demod_kernel<false, 0, 3, true>" : "mifsk::demod_kernel<false, 0, 3>" )
			  : bell202 ? ( resumable ? "mifsk::demod_kernel<true, 10, 2, true>" : "mifsk::demod_kernel<true, 10, 2>" )
				    : ( resumable ? "mifsk::demod_kernel<true, 0, 3, true>" : "mifsk::demod_kernel<true, 0, 3>" );
	plan_only->workgroup_size = block;
	plan_only->lds_bytes = (uint32_t)lds_all;
	plan_only->lattice_mode = lat_mode;
	plan_only->frames_per_block = lat_frames * lat_rounds;
	plan_only->waves_per_simd = bell202 ? 3 : 4;
	plan_only->chain_groups = chain_g;
	plan_only->chain_chunks = chain_k;
	return 0;
    }
    WgResume rs;
    std::memset(&rs, 0, sizeof(rs));
    if ( resumable ) {
	// mifsk_demod_slab and the chained launches: the instantiations with the state code
	rs.d_state = wh->d_state;
	rs.d_origin = wh->d_origin;
	rs.final = wh->final ? 1u : 0u;
	rs.bufsize = wh->samplebuf_size;
	const void *fn = !use_slab ? reinterpret_cast<const void *>(&demod_kernel<false, 0, 3, true>)
		       : bell202 ? reinterpret_cast<const void *>(&demod_kernel<true, 10, 2, true>)
				 : reinterpret_cast<const void *>(&demod_kernel<true, 0, 3, true>);
	if ( hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_all) != hipSuccess )
	    return -5;
	uint32_t a_slab_cap = use_slab ? slab_cap : 0u, a_lat_frames = use_slab ? lat_frames : 0u;
	uint32_t a_lat_rounds = use_slab ? lat_rounds : 1u, a_region_floats = use_slab ? (uint32_t)region_floats : 0u;
	uint32_t a_region_cap = use_slab ? region_cap : 0u, a_lat_mode = use_slab ? lat_mode : (uint32_t)LAT_NONE;
	if ( !chain_g ) {
	    mifsk_demod_io o = io;
	    void *kargs[] = { (void *)&d_cfg, (void *)&d_tw, (void *)&o, (void *)&a_slab_cap, (void *)&a_lat_frames,
			      (void *)&a_lat_rounds, (void *)&a_region_floats, (void *)&a_region_cap, (void *)&a_lat_mode, (void *)&rs };
	    (void)hipLaunchKernel(fn, dim3((unsigned)io.nstreams), dim3(block), kargs, lds_all, st);
	    return hip_rc(hipGetLastError());
	}
	const WaveChain &ch = *wh->chain;
	if ( (size_t)io.nstreams > ch.state_cap )
	    return -12;
	hipEvent_t fork = (hipEvent_t)ch.ev_fork;
	if ( hipEventRecord(fork, st) != hipSuccess )
	    return -5;
	mifsk_demod_io gio[WaveChain::kMaxGroups];
	uint32_t glo[WaveChain::kMaxGroups];
	bool prepared = true;
	for ( uint32_t gi = 0; gi < chain_g; gi++ ) {
	    hipStream_t gs = (hipStream_t)ch.streams[gi];
	    // behind the caller's stream, and behind whatever the call before left on ANY group's
	    // stream (its groups were other ranges of the state array)
	    (void)hipStreamWaitEvent(gs, fork, 0);
	    for ( uint32_t h = 0; h < (uint32_t)WaveChain::kMaxGroups; h++ )
		if ( h != gi )
		    (void)hipStreamWaitEvent(gs, (hipEvent_t)ch.ev_done[h], 0);
	    const uint32_t lo = (uint32_t)( (uint64_t)io.nstreams * gi / chain_g );
	    const uint32_t hi = (uint32_t)( (uint64_t)io.nstreams * ( gi + 1u ) / chain_g );
	    glo[gi] = lo;
	    mifsk_demod_io &o = gio[gi];
	    o = io;
	    o.nstreams = (int)( hi - lo );
	    o.d_samples = io.d_samples + (size_t)lo * io.stream_stride;
	    if ( io.d_nsamples ) o.d_nsamples = io.d_nsamples + lo;
	    if ( io.d_bytes ) o.d_bytes = io.d_bytes + (size_t)lo * io.frames_cap;
	    if ( io.d_nbytes ) o.d_nbytes = io.d_nbytes + lo;
	    if ( io.d_bits ) o.d_bits = io.d_bits + (size_t)lo * io.frames_cap;
	    if ( io.d_frames ) o.d_frames = io.d_frames + (size_t)lo * io.frames_cap;
	    if ( io.d_nframes ) o.d_nframes = io.d_nframes + lo;
	    if ( io.d_episodes ) o.d_episodes = io.d_episodes + (size_t)lo * io.episodes_cap;
	    if ( io.d_nepisodes ) o.d_nepisodes = io.d_nepisodes + lo;
	    if ( io.d_status ) o.d_status = io.d_status + lo;
	    if ( o.nstreams > 0
		    && hipMemsetAsync(ch.d_state + lo, 0, (size_t)o.nstreams * sizeof(mifsk_stream_state), gs) != hipSuccess )
		prepared = false;		// (no early return: the caller's stream is joined below either way)
	}
	const uint32_t chunk = ( io.nsamples + chain_k - 1u ) / chain_k;
	rs.append = 1u;
	rs.d_origin = nullptr;
	for ( uint32_t k = 0; k < chain_k && prepared; k++ ) {
	    const bool last = k + 1u == chain_k;
	    rs.final = last ? 1u : 0u;
	    const uint64_t lim = (uint64_t)( k + 1u ) * chunk;
	    rs.limit = last ? 0u : ( lim > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)lim );
	    for ( uint32_t gi = 0; gi < chain_g; gi++ ) {
		if ( gio[gi].nstreams <= 0 )
		    continue;
		rs.d_state = ch.d_state + glo[gi];
		void *kargs[] = { (void *)&d_cfg, (void *)&d_tw, (void *)&gio[gi], (void *)&a_slab_cap, (void *)&a_lat_frames,
				  (void *)&a_lat_rounds, (void *)&a_region_floats, (void *)&a_region_cap, (void *)&a_lat_mode, (void *)&rs };
		(void)hipLaunchKernel(fn, dim3((unsigned)gio[gi].nstreams), dim3(block), kargs, lds_all,
				      (hipStream_t)ch.streams[gi]);
	    }
	}
	const bool launched = hipGetLastError() == hipSuccess && prepared;
	for ( uint32_t gi = 0; gi < chain_g; gi++ ) {
	    (void)hipEventRecord((hipEvent_t)ch.ev_done[gi], (hipStream_t)ch.streams[gi]);
	    (void)hipStreamWaitEvent(st, (hipEvent_t)ch.ev_done[gi], 0);
	}
	return launched ? 0 : -5;
    }
    if ( use_slab ) {
	const size_t lds_bytes = kLdsHeader + slab_floats * 4;
	hipError_t e = hipFuncSetAttribute(
		bell202 ? reinterpret_cast<const void *>(&demod_kernel<true, 10, 2>)
			: reinterpret_cast<const void *>(&demod_kernel<true, 0, 3>),
		hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
	if ( e != hipSuccess )
	    return hip_rc(e);
	if ( bell202 )
	    hipLaunchKernelGGL((demod_kernel<true, 10, 2>), dim3((unsigned)io.nstreams), dim3(block),
			       lds_bytes, st, d_cfg, d_tw, io, slab_cap, lat_frames, lat_rounds,
			       (uint32_t)region_floats, region_cap, lat_mode, rs);
	else
	    hipLaunchKernelGGL((demod_kernel<true, 0, 3>), dim3((unsigned)io.nstreams), dim3(block),
			       lds_bytes, st, d_cfg, d_tw, io, slab_cap, lat_frames, lat_rounds,
			       (uint32_t)region_floats, region_cap, lat_mode, rs);
    } else {
	hipLaunchKernelGGL((demod_kernel<false, 0, 3>), dim3((unsigned)io.nstreams), dim3(block),
			   kLdsHeader + 16, st, d_cfg, d_tw, io, 0u, 0u, 1u, 0u, 0u, (uint32_t)LAT_NONE, rs);
    }
    return hip_rc(hipGetLastError());
}

int launch_demod_batch( const DevCfg &cfg, const DevCfg *d_cfg, const double *d_tw,
	const mifsk_demod_io &io, void *stream, LaunchInfo *plan_only, const WgHostArgs *wh )
{
    if ( io.nstreams <= 0 && !plan_only )
	return 0;
    if ( cfg.lat_linear && cfg.bit_nsamples == 40u ) {
	const int rc = launch_with_workers(cfg, d_cfg, d_tw, io, stream, plan_only, 2u, wh);
	if ( rc != kNotBell202 )
	    return rc;
    }
    return launch_with_workers(cfg, d_cfg, d_tw, io, stream, plan_only, 3u, wh);
}

int launch_detect_carrier( const float *d_samples, unsigned nsamples,
	const double *d_cs, unsigned fftsize, unsigned nbands, float *d_mags, void *stream )
{
    const unsigned blocks = ( nbands + 255 ) / 256;
    hipLaunchKernelGGL(spectrum_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
		       d_samples, nsamples, d_cs, fftsize, nbands, d_mags);
    return hip_rc(hipGetLastError());
}

} // namespace mifsk
