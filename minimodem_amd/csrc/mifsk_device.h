// mifsk_device.h -- types shared by the host glue and the HIP kernels.
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstdlib>

#include "mifsk.h"

namespace mifsk {

// ---------------------------------------------------------------------------
// Shared-segment plan of one zig-zag scan with long bit windows (SCAN in the tiled
// instantiation of the wavefront engine; DESIGN.md "shared segments").
//
// All windows of one fsk_find_frame -- every candidate x every bit (fsk.c:199-254,
// 477-502) -- cover one contiguous span of the stream, and windows of neighbouring
// candidates overlap by most of their length.  The span is cut at every window
// edge (and long pieces once more, to balance the lanes) into SEGMENTS; each
// segment's two-bin partial DFT, phase origin at its own start, is computed ONCE,
// one lane per segment in at most two passes of lanes; a window is then the sum of
// its segments' partials, each rotated by the table entry of its offset inside the
// window.  Indices in POSITION order: a window's segments are seg[first .. first+count).
// Made on the host (fill_devcfg), relative to the search cursor.
// ---------------------------------------------------------------------------
constexpr int SEG_MAX = 128;		// segments per plan (two passes of lanes)
constexpr int SEGW_MAX = 128;		// windows per plan

struct SegPlan {
    uint32_t	valid;			// 0: this scan correlates every window by itself
    uint32_t	nseg, npass, nwin;	// nwin = candidates x bits, w = candidate (scan order) * n_bits + bit
    uint32_t	span_hi;		// every segment ends at or before cursor + span_hi
    uint32_t	pass_len[2];		// longest segment of the pass (its lock-step length)
    uint32_t	pass_min[2];		// shortest segment of the pass (groups below it need no mask)
    float	bound_c;		// |assembled - index order| <= bound_c * 2^-53 * sum |x| over the window
    uint32_t	seg_rel[SEG_MAX];	// segment start, relative to the cursor
    uint16_t	seg_len[SEG_MAX];
    uint16_t	slot_seg[SEG_MAX];	// pass * 64 + lane -> segment, 0xFFFF: idle lane
    uint16_t	win_first[SEGW_MAX];	// window -> its first segment ...
    uint16_t	win_count[SEGW_MAX];	// ... and how many
    // the same, packed as the kernel reads it (one word per lane and pass / per window):
    uint32_t	p_slot[SEG_MAX];	// pass * 64 + lane -> seg_rel (20 bits) | seg_len << 20
    uint32_t	p_win[SEGW_MAX];	// window -> first | count << 8 | (window start rel. to the cursor) << 16
    uint8_t	p_slot_seg[SEG_MAX];	// pass * 64 + lane -> segment, 0xFF: idle lane
};

// Everything a kernel needs, as one POD passed by value in the kernarg
// segment (so it lands in SGPRs / the scalar cache, uniform for the launch).
struct DevCfg {
    uint32_t	n_bits;			// expect_n_bits (<= 64)
    uint32_t	bit_nsamples;		// samples per bit window   (fsk.c:183)
    uint32_t	last_reach;		// bit_offset[n_bits-1] + bit_nsamples
    float	magscalar;		// 2.0f / bit_nsamples      (fsk.c:132)
    uint32_t	frame_nsamples;
    uint32_t	expect_nsamples;
    uint32_t	overscan;
    uint32_t	try_first[2], try_max[2], try_step[2], try_step_fine[2];
    float	conf_threshold;
    float	search_limit;
    uint32_t	n_data_bits;
    uint32_t	nstartbits;
    uint32_t	has_stopbits;
    uint32_t	msb_first;
    uint32_t	do_rx_sync;
    uint32_t	rx_one;
    uint64_t	sync_byte;
    uint32_t	skew;			// LDS slab row padding (see mifsk_kernels.hip)
    uint32_t	div_magic;		// floor(2^32 / bit_nsamples)
    uint32_t	lock_advance;		// cursor step of a frame locked at its first try
    uint32_t	la_magic;		// floor(2^32 / lock_advance)
    uint32_t	nbits_magic;		// floor(2^32 / n_bits)
    uint32_t	lock_back;		// slab row 0 sits this far before a locked frame's first try
    uint32_t	lat_linear;		// lattice windows all start on 16-byte boundaries of their region
    // the windows of consecutive locked frames tile one grid of bit lengths
    // (bit_offset[k] = k B, lock_advance = (n_bits - 1) B): the last window of a
    // frame IS the first window of the next, and a round of F frames has only
    // F (n_bits - 1) + 1 distinct windows
    uint32_t	lat_grid;
    uint32_t	b_mark;			// the plan's mark band (episodes report it)
    uint32_t	b_space, fftsize;	// (with b_mark: what the shared segments' rotation tables are made for)
    // closed form of the four zig-zag scans of the receive loop (fsk.c:477-484; ZigZag in
    // mifsk_devlib.h): up / down candidate counts of [0] coarse without carrier, [1] coarse with
    // carrier, [2] fine without, [3] fine with (minimodem.c:1236-1263,1366)
    uint32_t	zz_up[4], zz_down[4];
    uint32_t	bit_offset[MIFSK_MAX_FRAME_BITS];	// fsk.c:204
    // expect strings as bit masks, [0]=data [1]=sync: bit k of req_mask is set
    // when bit k of the frame is required ('0'/'1'), req_val holds its value
    uint64_t	req_mask[2];
    uint64_t	req_val[2];
    // shared-segment plans of the four scans (index as zz_up / zz_down); valid only for
    // the long-window modes the tiled instantiation runs.  [4]: the carrier-held coarse scan
    // and the fine scan that may follow it at the same cursor (minimodem.c:1265,1373) as ONE
    // plan -- windows 0 .. nwin(1)-1 are the coarse scan's, the fine scan's follow: the span is
    // read and summed once, the fine scan's windows are assembled from sums already there
    SegPlan	seg[5];
    uint32_t	seg_union_first_fine;	// [4]: index of the fine scan's first window
};

// twiddles: tw[4*n + {0,1,2,3}] = cos_mark, -sin_mark, cos_space, -sin_space
// of angle 2*pi*((b*n) mod fftsize)/fftsize, in double.

void fill_devcfg( DevCfg &d, const mifsk_rx_config &c );

// entries (samples) of a twiddle table for bit windows of B samples: whole
// groups of 16 plus one group of look-ahead, never fewer than three groups
inline size_t tw_entries( unsigned B )
{
    const size_t n = ( ( (size_t)B + 15 ) & ~(size_t)15 ) + 16;
    return n < 48 ? 48 : n;
}

// launchers (mifsk_kernels.hip); `stream` is a hipStream_t
int launch_find_frame_batch( const DevCfg &cfg, const DevCfg *d_cfg, const double *d_tw,
	const float *d_samples, const mifsk_search *d_problems,
	mifsk_search_result *d_results, int nproblems, void *stream );

// what a launcher decided (mifsk_demod_plan): filled instead of launching when
// the pointer is given
struct LaunchInfo {
    const char	*kernel;
    uint32_t	workgroup_size;
    uint32_t	lds_bytes;		// dynamic LDS per workgroup
    uint32_t	lattice_mode;		// LAT_*
    uint32_t	frames_per_block;	// LATTICE frames scored at once, at most
    uint32_t	waves_per_simd;		// what the instantiation is compiled for (its VGPR budget)
    uint32_t	chain_groups, chain_chunks;	// chained launches (WaveChain): groups of streams x time chunks; 0 = one launch
};

struct WaveChain;
// what the host glue hands the workgroup engine's launcher besides cfg / io: the loop state of
// streams that arrive in pieces (mifsk_demod_slab) and what a chained launch needs (the
// wavefront engine's WaveHostArgs carries the same)
struct WgHostArgs {
    int		ncu;		// compute units of the device
    uint32_t	samplebuf_size;
    mifsk_stream_state *d_state;	// mifsk_demod_slab: state in / out (nullptr: one call = whole streams)
    const uint64_t *d_origin;
    bool	final;
    // non-NULL: the launcher may chain (it decides by the batch's shape); the caller holds
    // whatever serialises the chain's users.  chain_ok: what a plan-only call assumes
    const WaveChain *chain;
    bool	chain_ok;
};

int launch_demod_batch( const DevCfg &cfg, const DevCfg *d_cfg, const double *d_tw,
	const mifsk_demod_io &io, void *stream, LaunchInfo *plan_only = nullptr,
	const WgHostArgs *wh = nullptr );

// ---- one wavefront per stream (mifsk_wave.hip) ---------------------------

enum { LAT_NONE = 0u, LAT_LINEAR = 1u, LAT_DIRECT = 2u };	// how a LATTICE block gets at the samples

// launch geometry and run-time options of demod_wave_kernel (by value: SGPRs)
struct WaveGeom {
    uint32_t	mags_cap;	// LDS: (mark, space) magnitude slots
    uint32_t	slab_floats;	// LDS: floats in the sample slab
    uint32_t	slab_cap;	// samples a skewed SCAN slab holds (0: SCAN streams from global memory)
    uint32_t	tiled;		// the slab is a TILE_FLOATS tile: long windows come from global memory through it
    uint32_t	lat_mode;	// LAT_*
    uint32_t	lat_fmax;	// frames per LATTICE block, at most (<= 64)
    uint32_t	lat_fmin;	// ... and at least (one pass of lanes)
    uint32_t	round_wins;	// LINEAR: bit windows per staging round
    uint32_t	bufsize;	// the reference's samplebuf_size (minimodem.c:1063-1070)
    uint32_t	ring_exact;	// RING addressing (stale-cell semantics)
    uint32_t	ring_stride;	// floats per stream in the device-resident samplebuf
    // --auto-carrier (minimodem.c:1179-1220)
    uint32_t	autodetect;
    float	auto_threshold;
    float	nps;		// nsamples_per_scan = min(nsamples_per_bit, fftsize)
    int32_t	b_shift;	// space band = mark band + b_shift
    uint32_t	fftsize, nbands;
    uint32_t	tw_entries;	// samples per per-stream twiddle table
};

struct WaveAuto {
    const double	*d_cs;		// [fftsize][2]: cos, -sin of 2 pi k / fftsize
    double		*d_tw_scratch;	// [nstreams][tw_entries][4]
    float		*d_ring;	// [nstreams][ring_stride], zero-initialised
    // mifsk_demod_slab: the loop's state between calls (NULL: one call = one whole stream)
    mifsk_stream_state	*d_state;
    const uint64_t	*d_origin;	// stream index of each row's first sample (NULL: 0)
    uint32_t		final;		// the rows end where the streams end
    uint32_t		limit;		// chained launches: this call sees the first `limit` samples of
					// every row (a row that ends before is complete); 0 = all
    // shared segments: the rotation factor of segment i of window w of scan `kind`, laid out
    // [i][w] so that the lanes of the assembly (lane = window) read consecutive entries:
    // d_rot[kind][(i * rot_stride[kind] + w) * 4 .. + 3] = table entry of the segment's offset
    // inside the window (NULL: gathered from the stream's own table -- --auto-carrier)
    const double	*d_rot[5];
    uint32_t		rot_stride[5];
    uint32_t		append;		// outputs continue behind the call before (state: n*_total)
};

// Chained launches.  A batch of more streams than the chip holds runs in rounds of serial
// chains of unequal length: while a round's stragglers finish, the slots the others left stay
// empty.  Cut in time instead -- G groups of streams x K chunks of every stream, each (group,
// chunk) one launch of the resumable kernel, a group's chunks in order on the group's own HIP
// stream -- the dispatcher fills those slots with the other groups' next chunk, and the idle
// tail shrinks to that of one chunk.  The context owns what this needs (mifsk_capi.cpp).
struct WaveChain {
    enum { kMaxGroups = 3 };
    void		*streams[kMaxGroups];	// hipStream_t, non-blocking
    void		*ev_fork;		// hipEvent_t: the caller's stream at the call
    void		*ev_done[kMaxGroups];	// ... and each group's last chunk
    mifsk_stream_state	*d_state;		// [state_cap] the loop's state between chunks
    size_t		state_cap;
};

// what the host glue hands the launcher besides cfg / io
struct WaveHostArgs {
    int		ncu;		// compute units of the device
    uint32_t	samplebuf_size;
    bool	ring_exact;
    uint32_t	ring_stride;
    float	*d_ring;
    bool	autodetect;
    float	auto_threshold;
    float	nps;
    int32_t	b_shift;
    uint32_t	fftsize, nbands;
    uint32_t	tw_entries;
    const double *d_cs;
    double	*d_tw_scratch;
    mifsk_stream_state *d_state;
    const uint64_t *d_origin;
    bool	final;
    const double *d_rot[5];
    uint32_t	rot_stride[5];
    // non-NULL: the launcher may chain (it decides by the batch's shape); the caller holds
    // whatever serialises the chain's users.  chain_ok: what a plan-only call assumes
    const WaveChain *chain;
    bool	chain_ok;
};

int launch_demod_wave( const DevCfg &cfg, const DevCfg *d_cfg, const double *d_tw,
	const mifsk_demod_io &io, const WaveHostArgs &ha, void *stream, LaunchInfo *plan_only = nullptr );

int launch_detect_carrier( const float *d_samples, unsigned nsamples,
	const double *d_cs /* [fftsize][2] */, unsigned fftsize, unsigned nbands,
	float *d_mags /* [nbands] */, void *stream );

// self-test of the short square root of band_mag2() (mifsk_kernels.hip); d_out: four counters
int launch_selftest_sqrt( uint64_t seed, uint32_t blocks, uint32_t per_thread, unsigned long long *d_out, void *stream );

// Tuning overrides for experiments (MIFSK_ENGINE, MIFSK_WAVES_PER_CU, MIFSK_SV,
// MIFSK_LDS_PAD, MIFSK_LAT_ROUNDS, MIFSK_CHAIN): honoured only when MIFSK_EXPERIMENT is set in the
// environment, so that a stray variable cannot change what production launches.
inline const char *experiment_env( const char *name )
{
    return std::getenv("MIFSK_EXPERIMENT") != nullptr ? std::getenv(name) : nullptr;
}

// the HIP device a context is bound to (mifsk_capi.cpp)
int ctx_device( const mifsk_ctx *ctx );

} // namespace mifsk
