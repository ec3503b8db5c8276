// mifsk_wave.hip -- the receive loop, ONE WAVEFRONT PER STREAM (gfx950 / CDNA4).
//
// The reference's loop (src/minimodem.c:1137-1463) is a serial, data-dependent
// cursor per stream; streams are independent.  MI355X has 1024 SIMDs and every
// BASELINE workload has at least 1024 streams, so the natural mapping is one
// 64-lane wavefront per stream: a workgroup IS a wave.  Nothing is ever
// exchanged between waves -- no s_barrier, no command block, no idle master --
// and the SIMDs are kept busy by however many independent streams are resident
// (1 to 4 waves per SIMD, decided by the LDS each needs).
//
// Inside the wave the 64 lanes are used three ways:
//   correlate   one lane = one bit window: two complex dot products against the
//               mark / space twiddles, f64 fma in index order (fsk.c:117-174 is a
//               zero-padded FFT read at two bins; mifsk_devlib.h).  The twiddle
//               index is uniform, so twiddles come through the scalar cache.
//   score       one lane = one candidate frame: fsk_frame_analyze's confidence
//               (fsk.c:271-342) over that frame's magnitudes.
//   replay      one lane = one frame of a LATTICE block: the loop's f32 state
//               recurrences replayed in frame order as a DPP lane scan.
//
// LATTICE blocks: while carrier is held the next frame is first looked for
// exactly lock_advance samples after the last (minimodem.c:1263,1407) and that
// first try ends the search whenever it reaches the search limit (fsk.c:499).
// So the wave evaluates a block of up to 64 consecutive lattice frames at once
// -- audio staged through LDS in coalesced 16-byte loads with the next round's
// loads already in flight, or streamed per lane for long windows -- scores them,
// and replays the reference's decisions over them; the first frame that fails a
// predicate goes through the general path (fsk_find_frame as a SCAN of all
// candidates followed by the reference's selection in scan order), which
// recomputes from the samples.  Results never depend on the speculation.
//
// Buffer arithmetic: (base, rp) = absolute index of samplebuf[0] and the file
// position; samples_nvalid = rp - base evolves exactly as minimodem.c:1144-1174
// makes it.  FLAT addressing (default): a search that reads past samples_nvalid
// sees the stream itself, 0.0 beyond its end.  RING addressing
// (MIFSK_IO_RING_EXACT): the reference's samplebuf is kept cell for cell in
// device memory (memmove and half-buffer refills included), so such reads see
// the stale cells the reference sees.
//
// --auto-carrier (minimodem.c:1179-1220,1297): fsk_detect_carrier runs inside
// the loop whenever no band is held -- also again after 21 searches without
// confidence -- and the stream's twiddle table is rebuilt on the device for the
// (mark, space) pair found.
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "mifsk_device.h"
#include "mifsk_devmath.h"
#include "mifsk_devlib.h"

#ifdef MIFSK_PROFILE
#define MIFSK_WCLOCK() ((uint32_t)clock64())
#else
#define MIFSK_WCLOCK() 0u
#endif

namespace mifsk {

// A staging round of the linear LATTICE loads SV float4 per lane (SV KiB per
// wave).  Two instantiations: 10 where a wave can have 10 KiB of LDS for it
// (1024 streams on 256 CUs leave each wave a quarter of a CU's LDS), 4 where
// sixteen waves share a CU.

// demod_wave_kernel's arguments as they lie in the kernarg segment (see DemodArgs in
// mifsk_kernels.hip: a view for the cold end of the loop; the kernel takes them as separate
// parameters because only a __restrict__ pointer parameter keeps the configuration reads
// scalar)
struct WaveArgs {
    const DevCfg	*cfgp;
    const double	*tw_default;
    mifsk_demod_io	io;
    WaveGeom		g;
    WaveAuto		au;
};

// where stream s writes its results.  Made once (see StreamOut in mifsk_kernels.hip: the
// serial loop is latency-bound; re-making these from the kernarg segment at every use cost
// 12000 baud 1.29 -> 1.79 ms)
struct WaveOut {
    uint8_t		*bytes;
    uint64_t		*bits;
    mifsk_frame		*frames;
    mifsk_episode	*eps;
    uint32_t		fcap, ecap;
};

// ---------------------------------------------------------------------------
// fsk_detect_carrier over one window (fsk.c:543-581): the band (>= 1) with the
// largest magnitude among those not below the threshold, the lowest such band
// on a tie (the reference keeps the first strict maximum), or -1.  One band per
// lane per pass; X[b] = sum_n x[n] e^{-2 pi i (b n mod N)/N} in f64 fma, n
// ascending -- the oracle's sums in the oracle's order.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int wave_detect_window( const float *__restrict__ w, uint32_t n_win,
	const double *__restrict__ cs, uint32_t fftsize, uint32_t nbands, float threshold )
{
    const uint32_t lane = threadIdx.x;
    const float magscalar = 1.0f / ( (float)n_win / 2.0f );		// fsk.c:553
    float best = 0.0f;
    int best_band = -1;
    for ( uint32_t b0 = 1u; b0 < nbands; b0 += 64u ) {
	const uint32_t b = b0 + lane;
	const bool active = b < nbands;
	const uint32_t bb = active ? b : 1u;
	double re = 0.0, im = 0.0;
	uint32_t k = 0;						// (b * n) mod fftsize
	for ( uint32_t n = 0; n < n_win; n++ ) {
	    const double x = (double)w[n];			// one address for the whole wave
	    re = fma(x, cs[2 * (size_t)k], re);
	    im = fma(x, cs[2 * (size_t)k + 1], im);
	    k += bb;
	    if ( k >= fftsize )
		k -= fftsize;
	}
	const float mag = band_mag(re, im, magscalar);
	if ( active && !( mag < threshold ) && best < mag ) {	// fsk.c:570-575 (ascending bands per lane)
	    best = mag;
	    best_band = (int)b;
	}
    }
    // first strict maximum over all bands = largest magnitude, lowest band on ties
#pragma unroll
    for ( int o = 32; o > 0; o >>= 1 ) {
	const float m2 = __shfl_xor(best, o);
	const int b2 = __shfl_xor(best_band, o);
	const bool take = b2 >= 0 && ( best_band < 0 || m2 > best || ( m2 == best && b2 < best_band ) );
	if ( take ) {
	    best = m2;
	    best_band = b2;
	}
    }
    return __builtin_amdgcn_readfirstlane(best_band);
}

// fsk_set_tones_by_bandshift (fsk.c:584-598) for one stream of an --auto-carrier batch: its
// own table tw[4 n + {0,1,2,3}] = cos, -sin of 2 pi ((b n) mod N) / N for mark and space, made
// from the cos / -sin table of the spectrum (the same values the host builds)
__device__ __forceinline__ void wave_build_table( double *tw_own, const double *__restrict__ cs,
	uint32_t band, uint32_t b_space, uint32_t bit_nsamples, uint32_t fftsize, uint32_t tw_entries )
{
    for ( uint32_t n = threadIdx.x; n < tw_entries; n += 64u ) {
	double w0 = 0.0, w1 = 0.0, w2 = 0.0, w3 = 0.0;
	if ( n < bit_nsamples ) {
	    const uint32_t km = ( band * n ) % fftsize;
	    const uint32_t ks = ( b_space * n ) % fftsize;
	    w0 = cs[2 * (size_t)km];
	    w1 = cs[2 * (size_t)km + 1];
	    w2 = cs[2 * (size_t)ks];
	    w3 = cs[2 * (size_t)ks + 1];
	}
	double *t = tw_own + 4 * (size_t)n;
	t[0] = w0; t[1] = w1; t[2] = w2; t[3] = w3;
    }
    // the table is read back through the scalar cache (and by vector loads in the
    // tail paths): complete the stores, then drop stale lines
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    __builtin_amdgcn_s_dcache_inv();
}

// everything the kernel keeps per stream beyond the loop's scalars
// NQ: > 0 = the resident-table correlator for windows of 4 NQ samples (linear
// LATTICE); 0 = any bit length, linear LATTICE; kDirect = any bit length, the lattice's
// windows streamed per lane from global memory (or no lattice): no staging rounds, no round
// prefetch registers; kTiled = the instantiation for long windows read from global memory
// through the LDS tile (no linear LATTICE in it either)
constexpr int kTiled = -1;
constexpr int kDirect = -2;

template <int SV, int NQ>
struct Wave {
    static constexpr uint32_t kRoundFloats = 64u * SV * 4u;	// samples one staging round loads
    static constexpr bool kLinear = NQ >= 0;			// the lattice of this instantiation is the linear one
    const DevCfg	&cfg;
    const WaveGeom	&g;
    const double	*tw;
    const float		*x;		// this stream's samples
    uint32_t		N;		// its length
    float2		*mags;		// LDS: (mark, space) magnitude per bit window
    float		*slab;		// LDS: staged samples
    float		*ring;		// RING addressing: the reference's samplebuf in device memory
    uint32_t		lane;
    uint32_t		safe_limit;	// how far this row may be over-read (whole-round loads)
    // SCAN slab: absolute range currently staged (skewed layout)
    uint32_t		slab_lo, slab_hi;
    // LATTICE block held in registers: lane f = the frame whose first try sits at
    // lat_anchor + f * lock_advance, for f < lat_n
    float		l_conf, l_ampl;
    uint64_t		l_bits;
    uint32_t		lat_n, lat_anchor;
    // how far to speculate: frames per block
    uint32_t		spec, run, cold, pause;
    // score of candidate 0 (position cursor + try_first, data string) of the scan
    // just made, reusable by a rescan at the same cursor (scan())
    FrameOut		k0;
    bool		k0_valid;
    // shared segments: the coarse + fine plan's partial sums for the search at this cursor
    // are on the tile (nothing has used the tile since)
    uint32_t		un_base = 0;
    bool		un_valid = false;
    // register prefetch of the next LINEAR round
    float4		pbuf[SV];
    uint32_t		pref_lo;
    // groups 0..2 of the twiddle table (entries 0..47), resident (NQ > 0): what the
    // linear LATTICE correlator needs for its bit windows of 4 NQ <= 48 samples
    TwGroup		tgr[3];
    // work counters (MIFSK_CNT_*): a block of LDS words bumped by lane 0 -- kept out of
    // the scalar registers, which the loop state needs
    uint32_t		*cnt;
    bool		cnt_on;		// (the caller asked for them: io.d_counters)
    uint32_t		cyc_block, cyc_scan, cyc_stage, cyc_corr, cyc_conf;
    uint32_t		cyc_s_stage, cyc_s_corr, cyc_s_conf;
    uint32_t		cyc_g_pass = 0, cyc_g_asm = 0, cyc_g_redo = 0;	// shared-segment SCAN: passes, assembly, index-order repeats



    __device__ __forceinline__ Wave( const DevCfg &c, const WaveGeom &gg, const double *t,
	    const float *xs, uint32_t n, float2 *m, float *s, float *r,uint32_t safe, uint32_t *counters, bool counting )
	: cfg(c), g(gg), tw(t), x(xs), N(n), mags(m), slab(s), ring(r), lane(threadIdx.x),
	  safe_limit(safe), slab_lo(0), slab_hi(0), l_conf(0.0f), l_ampl(0.0f), l_bits(0),
	  lat_n(0), lat_anchor(0), spec(gg.lat_fmin), run(0), cold(0), pause(0),
	  k0_valid(false), pref_lo(0xFFFFFFFFu),cnt(counters), cnt_on(counting), cyc_block(0), cyc_scan(0), cyc_stage(0), cyc_corr(0), cyc_conf(0),
	  cyc_s_stage(0), cyc_s_corr(0), cyc_s_conf(0)
    {
	if constexpr ( kLinear ) {
#pragma unroll
	    for ( int i = 0; i < SV; i++ )
		pbuf[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	}
	load_resident_twiddles();
    }

    // (ds_add_u32 without return: fire and forget, the wave never waits for it)
    __device__ __forceinline__ void bump( uint32_t which, uint32_t by = 1u ) const
    {
	if ( cnt_on && lane == 0 )
	    (void)__hip_atomic_fetch_add(&cnt[which], by, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }

    // (again after --auto-carrier has rebuilt the stream's table)
    __device__ __forceinline__ void load_resident_twiddles()
    {
	if constexpr ( NQ > 0 ) {
#pragma unroll
	    for ( int gi = 0; gi < ( NQ + 3 ) / 4; gi++ )
		tgr[gi] = tw_group_load(tw, (uint32_t)gi, lane);
	}
    }

    // nothing prefetched is wanted any more: end the registers' live ranges
    __device__ __forceinline__ void drop_prefetch()
    {
	if constexpr ( kLinear ) {
	    pref_lo = 0xFFFFFFFFu;
#pragma unroll
	    for ( int i = 0; i < SV; i++ )
		pbuf[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	}
    }

    // index of the scored lattice frame whose first try is at p, or ~0u
    __device__ __forceinline__ uint32_t lattice_lookup( uint32_t p ) const
    {
	if ( !lat_n || p < lat_anchor )
	    return ~0u;
	const uint32_t d = p - lat_anchor;
	const uint32_t e = udiv_magic(d, cfg.lock_advance, cfg.la_magic);
	return ( e < lat_n && e * cfg.lock_advance == d ) ? e : ~0u;
    }

    // window w of a block: start relative to the block's anchor
    __device__ __forceinline__ uint32_t win_rel( uint32_t w ) const
    {
	if ( cfg.lat_grid )
	    return NQ > 0 ? w * (uint32_t)( 4 * ( NQ > 0 ? NQ : 1 ) )	// (this instantiation's bit length: a shift)
			  : w * cfg.bit_nsamples;
	const uint32_t f = udiv_magic(w, cfg.n_bits, cfg.nbits_magic);
	return f * cfg.lock_advance + cfg.bit_offset[( w - f * cfg.n_bits ) & 63u];
    }

    // ------------------------------------------------------------------
    // LATTICE, linear variant: bit length, bit offsets and frame step are
    // multiples of 4 samples, so every window of a round starts a multiple of
    // 16 bytes after the round's first sample.  A round = the windows whose
    // span fits one staging pass of 64 x STAGE_VEC float4: one (possibly
    // unaligned) 16-byte global load and one ds_write_b128 per lane per KiB,
    // then one lane per window reading ds_read_b128.  The next round's loads
    // are issued before this round is correlated and consumed a round later.
    // ------------------------------------------------------------------
    __device__ __forceinline__ void round_linear( uint32_t A, uint32_t w0, uint32_t nw, uint32_t next_lo )
    {
	const uint32_t B = cfg.bit_nsamples;
	const uint32_t t_in = MIFSK_WCLOCK();
	const uint32_t lo = A + win_rel(w0);			// uniform
	// Raw loads of one round: 64 * SV consecutive float4 from sample
	// `from`, whatever the round really needs -- no per-lane bounds logic.
	// `safe_limit` (>= N) is how far this row may be over-read without leaving
	// the batch's allocation; a round that would cross it is fetched from
	// sample 0 instead and its data never used (it reaches the end of the
	// stream, so it is re-read by element below).
	if ( pref_lo != lo ) {
	    const bool ok = lo <= safe_limit - kRoundFloats;
	    const float *pb = x + ( ok ? lo : 0u ) + ( lane << 2 );
#pragma unroll
	    for ( int i = 0; i < SV; i++ ) {
		const float4_u sv = *reinterpret_cast<const float4_u *>(pb + i * 256);
		pbuf[i] = make_float4(sv.x, sv.y, sv.z, sv.w);
	    }
	}
	float *lane_base = slab + ( lane << 2 );
	const uint32_t elast = lo + kRoundFloats;
	if ( elast <= N && elast >= lo ) {
#pragma unroll
	    for ( int i = 0; i < SV; i++ )
		*reinterpret_cast<float4 *>(lane_base + i * 256) = pbuf[i];
	} else {
	    // the round reaches the end of the stream: by element, 0.0 beyond it
#pragma unroll
	    for ( int i = 0; i < SV; i++ ) {
		const uint32_t e = lo + ( ( i * 64 + lane ) << 2 );
		float4 sv;
		sv.x = ( e < N ) ? x[e] : 0.0f;
		sv.y = ( e + 1 < N && e + 1 > e ) ? x[e + 1] : 0.0f;
		sv.z = ( e + 2 < N && e + 2 > e ) ? x[e + 2] : 0.0f;
		sv.w = ( e + 3 < N && e + 3 > e ) ? x[e + 3] : 0.0f;
		*reinterpret_cast<float4 *>(lane_base + i * 256) = sv;
	    }
	}
	// the next round (of this block or, speculatively, of the block after it);
	// issued unconditionally, no control flow after it that joins before the
	// correlator (a join makes hipcc drain vmcnt)
	{
	    const bool ok = next_lo >= lo && next_lo <= safe_limit - kRoundFloats;
	    const float *pb = x + ( ok ? next_lo : 0u ) + ( lane << 2 );
#pragma unroll
	    for ( int i = 0; i < SV; i++ ) {
		const float4_u sv = *reinterpret_cast<const float4_u *>(pb + i * 256);
		pbuf[i] = make_float4(sv.x, sv.y, sv.z, sv.w);
	    }
	    pref_lo = ok ? next_lo : 0xFFFFFFFFu;
	}
	wave_lds_sync();
	const uint32_t t_mid = MIFSK_WCLOCK();
	cyc_stage += t_mid - t_in;
	const uint32_t nq = B >> 2;				// (linear: B % 4 == 0; == NQ when NQ > 0)
	for ( uint32_t s0 = 0; s0 < nw; s0 += 64u ) {
	    const uint32_t w = w0 + s0 + lane;
	    const bool active = s0 + lane < nw;
	    // (idle lanes shadow the round's first window)
	    uint32_t rel;
	    if ( NQ > 0 && cfg.lat_grid )	// window w starts w * B past the anchor: (w - w0) * B past `lo`
		rel = ( active ? s0 + lane : 0u ) * (uint32_t)( 4 * ( NQ > 0 ? NQ : 1 ) );
	    else
		rel = A + win_rel(active ? w : w0) - lo;
	    double acc[4] = { 0.0, 0.0, 0.0, 0.0 };
	    if constexpr ( NQ > 0 )
		corr_lds_fixed<NQ>(tgr, slab + rel, acc);	// the instantiation for this bit length
	    else
		corr_lds_stream(tw, slab + rel, nq, lane, acc);
	    if ( active )
		mags[w] = band_mag2(acc[0], acc[1], acc[2], acc[3], cfg.magscalar);
	}
	wave_lds_sync();			// the slab is rewritten by the next round
	cyc_corr += MIFSK_WCLOCK() - t_mid;
    }

    // LATTICE, direct variant: any bit length, no LDS staging.  One lane per bit
    // window; the lane streams ITS OWN window straight from global memory, 32
    // bytes per step with the next chunk already in flight; twiddles through the
    // scalar cache as always.  What runs SAME (92-sample windows on a 92.16
    // grid), Bell-103 (160) and RTTY (1056).
    __device__ __forceinline__ void pass_direct( uint32_t A, uint32_t w0, uint32_t nw )
    {
	const uint32_t B = cfg.bit_nsamples;
	const uint32_t w = w0 + lane;
	const bool active = lane < nw;
	const uint32_t a = A + win_rel(active ? w : w0);
	// whole groups of 16 (tile: steps of 64) are loaded
	const uint32_t Bpad = NQ == kTiled ? ( ( B + 63u ) & ~63u ) : ( ( B + 15u ) & ~15u );
	double acc[4] = { 0.0, 0.0, 0.0, 0.0 };
	if ( __all(a + Bpad <= N && a + Bpad >= a) ) {
	    if constexpr ( NQ == kTiled )
		corr_global_tiled(tw, x, a, nw < 64u ? nw : 64u, B, lane, slab, acc);
	    else
		corr_global_stream(tw, x + a, B, lane, acc);
	} else {
	    // a window reaches the end of the stream: per-sample guarded reads
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
	    mags[w] = band_mag2_exact(acc[0], acc[1], acc[2], acc[3], cfg.magscalar);
    }

    // Evaluate F lattice frames anchored at A (first-try position of frame 0).
    __device__ __forceinline__ void lattice_block( uint32_t A, uint32_t F )
    {
	const uint32_t t0 = MIFSK_WCLOCK();
	const uint32_t nb = cfg.n_bits;
	const uint32_t W = cfg.lat_grid ? F * ( nb - 1u ) + 1u : F * nb;
	if constexpr ( kLinear ) {
	    const uint32_t rw = g.round_wins;
	    for ( uint32_t w0 = 0; w0 < W; w0 += rw ) {
		const uint32_t nw = W - w0 < rw ? W - w0 : rw;
		// what comes after this round if the lattice goes on
		const uint32_t next_lo = w0 + rw < W ? A + win_rel(w0 + rw)
						     : A + F * cfg.lock_advance;
		round_linear(A, w0, nw, next_lo);
	    }
	} else {
	    for ( uint32_t w0 = 0; w0 < W; w0 += 64u )
		pass_direct(A, w0, W - w0 < 64u ? W - w0 : 64u);
	    wave_lds_sync();
	}
	// score: lane f = frame f (fsk.c:178-446 after the magnitudes)
	const uint32_t t_cf = MIFSK_WCLOCK();
	FrameOut fo;
	fo.conf = 0.0f; fo.ampl = 0.0f; fo.bits = 0;
	if ( lane < F ) {
	    const uint32_t ci = cfg.lat_grid ? lane * ( nb - 1u ) : lane * nb;
	    uint32_t fb = 0u;
	    fo = frame_confidence_any(&mags[ci], cfg.req_mask[0], cfg.req_val[0], nb, fb);
	    if ( cnt_on && fb )
		bump(MIFSK_CNT_CONF_FALLBACKS);
	}
	un_valid = false;			// (a block's windows may have gone through the tile)
	l_conf = fo.conf;
	l_ampl = fo.ampl;
	l_bits = fo.bits;
	lat_n = F;
	lat_anchor = A;
	slab_lo = slab_hi = 0;			// the rounds overwrote whatever SCAN had staged
	wave_lds_sync();
	bump(MIFSK_CNT_LATTICE_BATCHES);
	cyc_conf += MIFSK_WCLOCK() - t_cf;
	cyc_block += MIFSK_WCLOCK() - t0;
    }

    // ------------------------------------------------------------------
    // SCAN: fsk_find_frame at cursor `base` (absolute).  Every candidate of the
    // zig-zag scan and every bit of it is evaluated (one lane per bit window),
    // then the reference's selection (strict >, first tried wins ties, early
    // exit at the limit, fsk.c:492-501) is replayed in scan order.
    // ------------------------------------------------------------------

    // sample at absolute index i as a search at `base` sees it
    __device__ __forceinline__ float sample_at( uint32_t base, uint32_t i ) const
    {
	if ( ring )
	    return ring[i - base];		// the reference's buffer cell (always allocated)
	return i < N ? x[i] : 0.0f;
    }

    // stage [lo, lo + n) into the skewed slab whose row 0 is sample lo
    __device__ __forceinline__ void stage_slab( uint32_t base, uint32_t lo, uint32_t n )
    {

	const uint32_t org4 = lo & ~3u;
	const uint32_t head = lo - org4;
	const uint32_t nvec = ( n + head + 3 ) >> 2;
	if ( ring ) {
	    // RING addressing: cells of the device-resident samplebuf, element-wise
	    for ( uint32_t v0 = 0; v0 < nvec; v0 += 64u ) {
		const uint32_t v = v0 + lane;
		if ( v < nvec ) {
		    const uint32_t a = org4 + ( v << 2 );
		    float4 s;
		    s.x = a >= base ? ring[a - base] : 0.0f;
		    s.y = a + 1 >= base ? ring[a + 1 - base] : 0.0f;
		    s.z = a + 2 >= base ? ring[a + 2 - base] : 0.0f;
		    s.w = a + 3 >= base ? ring[a + 3 - base] : 0.0f;
		    store4_skewed(cfg, slab, g.slab_cap, v << 2, head, s, a, 0xFFFFFFF0u);
		}
	    }
	    return;
	}
	// (instantiations for one short bit length -- 12000 baud's runs at 128 VGPRs with eight
	// spilled and searches four times per stream -- are left as they were)
	if ( NQ <= 0 && cfg.skew == 0u ) {
	    // A slab without pad words (SAME: the host found that padding spreads no banks) is the
	    // stream itself: row 0 is put at the float4 boundary below `lo` -- the caller sets slab_lo
	    // to it (plain_slab_origin) -- and a vector is ONE aligned ds_write_b128, samples at or
	    // beyond N zeroed; the per-sample row arithmetic of store4_skewed (a divide by the bit
	    // length and four scalar stores per vector: ~75 VALU instructions per float4, 300 per
	    // fine scan of SAME, whose kernel is bound by VALU issue) is for padded slabs only.
	    for ( uint32_t v0 = 0; v0 < nvec; v0 += 64u * SV ) {
		float4 buf[SV];
#pragma unroll
		for ( int i = 0; i < SV; i++ ) {
		    const uint32_t v = v0 + i * 64 + lane;
		    if ( v0 + i * 64 < nvec )
			buf[i] = load4_raw(x, org4 + ( v << 2 ), N);
		}
#pragma unroll
		for ( int i = 0; i < SV; i++ ) {
		    const uint32_t v = v0 + i * 64 + lane;
		    if ( v0 + i * 64 < nvec && v < nvec ) {
			const uint32_t a = org4 + ( v << 2 );
			const float4 sv = ( a + 3u < N && a + 3u >= a ) ? buf[i] : mask4(buf[i], a, N);
			*reinterpret_cast<float4 *>(slab + ( v << 2 )) = sv;
		    }
		}
	    }
	    return;
	}
	for ( uint32_t v0 = 0; v0 < nvec; v0 += 64u * SV ) {
	    float4 buf[SV];
#pragma unroll
	    for ( int i = 0; i < SV; i++ ) {
		const uint32_t v = v0 + i * 64 + lane;
		if ( v0 + i * 64 < nvec )
		    buf[i] = load4_raw(x, org4 + ( v << 2 ), N);
	    }
#pragma unroll
	    for ( int i = 0; i < SV; i++ ) {
		const uint32_t v = v0 + i * 64 + lane;
		if ( v0 + i * 64 < nvec && v < nvec )
		    store4_skewed(cfg, slab, g.slab_cap, v << 2, head, buf[i], org4 + ( v << 2 ), N);
	    }
	}
    }

    // where row 0 of the slab lies after stage_slab(base, lo, n)
    __device__ __forceinline__ uint32_t plain_slab_origin( uint32_t lo ) const
    {
	if ( NQ <= 0 && cfg.skew == 0u && !ring )
	    return lo & ~3u;
	return lo;
    }

    // correlate candidates c0 .. c0+Q of `zz` at cursor base into mags[q * n_bits + k]
    __device__ __forceinline__ void scan_correlate( uint32_t base, const ZigZag &zz, uint32_t c0,
	    uint32_t Q, bool use_slab )
    {

	const uint32_t nb = cfg.n_bits, B = cfg.bit_nsamples;
	const uint32_t nwin = Q * nb;
	for ( uint32_t w0 = 0; w0 < nwin; w0 += 64u ) {
	    const uint32_t w = w0 + lane;
	    const bool active = w < nwin;
	    const uint32_t q = active ? udiv_magic(w, nb, cfg.nbits_magic) : 0u;
	    const uint32_t k = active ? w - q * nb : 0u;
	    const uint32_t a = base + zz.at(c0 + q) + cfg.bit_offset[k & 63u];
	    double acc[4] = { 0.0, 0.0, 0.0, 0.0 };
	    if ( use_slab ) {
		corr_skewed_stream(cfg, tw, slab, a - slab_lo, lane, acc);
	    } else if ( !ring && __all(a + ( NQ == kTiled ? ( ( B + 63u ) & ~63u ) : ( ( B + 15u ) & ~15u ) ) <= N
				       && a + B + 64u >= a) ) {
		// long windows (or no slab at this occupancy): from global memory,
		// through the LDS tile where there is one, else 64 bytes per lane per
		// 16 samples with the next group in flight
		if constexpr ( NQ == kTiled )
		    corr_global_tiled(tw, x, a, nwin - w0 < 64u ? nwin - w0 : 64u, B, lane, slab, acc);
		else
		    corr_global_stream(tw, x + a, B, lane, acc);
	    } else {
		for ( uint32_t n = 0; n < B; n++ ) {
		    const uint32_t idx = a + n;
		    const double xd = (double)( idx >= a ? sample_at(base, idx) : 0.0f );
		    const double *t = tw + 4 * (size_t)n;
		    acc[0] = fma(xd, t[0], acc[0]);
		    acc[1] = fma(xd, t[1], acc[1]);
		    acc[2] = fma(xd, t[2], acc[2]);
		    acc[3] = fma(xd, t[3], acc[3]);
		}
	    }
	    if ( active )
		mags[w] = band_mag2_exact(acc[0], acc[1], acc[2], acc[3], cfg.magscalar);
	}
    }

    // ------------------------------------------------------------------
    // SCAN with SHARED SEGMENTS (long windows, tiled instantiation; SegPlan in
    // mifsk_device.h): every sample of the span the scan covers is read and
    // summed once -- one lane per segment, phase origin at the segment's start -- and a
    // window is assembled from its segments' partial sums, each rotated by the table
    // entry of its offset inside the window:
    //     X_w = sum_s  S_s * w[p_s - a_w]
    // That is the index-order sum in another order, so it differs from it by rounding
    // only, and by no more than  delta = bound_c * 2^-53 * sum |x|  over the window
    // (derivation in DESIGN.md).  The reference rounds X to float (the FFT's output
    // type): wherever  (float)(X - delta) == (float)(X + delta)  that float is the one
    // the index-order sum gives, bit for bit, and everything downstream with it.  The few
    // windows where it is not are summed again in index order (corr_global_tiled).
    // Fills mags[(j - c0) * n_bits + k] for candidates j = c0 .. J-1 of `zz`.
    // ------------------------------------------------------------------
    // LDS of the tiled instantiation: [tile | plan words of the two carrier-held scans | list].
    // The partial sums live ON the tile between the passes and the assembly.
    static constexpr uint32_t kPlanWords = 2u * SEG_MAX + SEG_MAX / 4u;	// p_slot, p_win, p_slot_seg per kind
    __device__ __forceinline__ uint32_t *plan_cache() const
    {
	return reinterpret_cast<uint32_t *>(slab + TILE_FLOATS);
    }
    __device__ __forceinline__ void plan_cache_fill()
    {
	uint32_t *pc = plan_cache();
	for ( uint32_t kc = 0; kc < 2u; kc++ ) {
	    // slot 0: what the carrier-held coarse scan runs (with the fine scan's windows in
	    // it where there is such a plan), slot 1: the fine scan on its own
	    const SegPlan &sp = cfg.seg[kc ? 3u : ( cfg.seg[4].valid ? 4u : 1u )];
	    uint32_t *q = pc + kc * kPlanWords;
	    for ( uint32_t i = lane; i < (uint32_t)SEG_MAX; i += 64u ) {
		q[i] = sp.p_slot[i];
		q[SEG_MAX + i] = sp.p_win[i];
	    }
	    for ( uint32_t i = lane; i < (uint32_t)SEG_MAX / 4u; i += 64u )
		q[2u * SEG_MAX + i] = reinterpret_cast<const uint32_t *>(sp.p_slot_seg)[i];
	}
	wave_lds_sync();
    }

    // pi: the plan (0..3 = the scan's own, 4 = coarse + fine in one); w_off: the plan's index of
    // this scan's first window; with sum_it false the partial sums of the scan before (same
    // plan, same cursor) are still on the tile and only the assembly runs.  Returns whether
    // the partial sums are still there afterwards (no window had to be summed again).
    __device__ __forceinline__ bool seg_correlate( uint32_t base, const ZigZag &zz, uint32_t c0,
	    uint32_t pi, uint32_t w_off, bool sum_it )
    {
	const uint32_t kind = pi;
	// (read where it is used, from the kernarg segment: KernArgs in mifsk_devlib.h)
	const KernArgs<WaveArgs>::ptr ka = KernArgs<WaveArgs>::here();
	const double *rot = ka->au.d_rot[kind];		// (uniform) NULL: gather from the stream's table
	const uint32_t rstride = ka->au.rot_stride[kind];
	const SegPlan &sp = cfg.seg[kind];
	const uint32_t nwin_scan = zz.J * cfg.n_bits;	// this scan's windows: plan windows w_off ...
	const uint32_t nb = cfg.n_bits, B = cfg.bit_nsamples;
	// (one entry more than there can be segments: an all-zero partial sum, what the lanes
	// of the assembly read once their own window's segments are used up)
	constexpr uint32_t kParts = SEG_MAX + 1u;
	static_assert(kParts * ( 4 * sizeof(double) + 2 * sizeof(float) ) <= TILE_FLOATS * sizeof(float),
		      "the partial sums are kept on the tile");
	double *partD = reinterpret_cast<double *>(slab);			// [kParts][4]
	float *partA = reinterpret_cast<float *>(partD + 4 * kParts);		// [kParts]
	uint32_t *partRel = reinterpret_cast<uint32_t *>(partA + kParts);	// [kParts]
	const bool both = cfg.seg[1].valid && cfg.seg[3].valid;
	const bool cached = both && ( kind == 3u || kind == ( cfg.seg[4].valid ? 4u : 1u ) );
	const uint32_t *pc = plan_cache() + ( kind == 3u ? 1u : 0u ) * kPlanWords;
	const uint32_t tg0 = MIFSK_WCLOCK();
	const uint32_t np = sp.npass;
	if ( sum_it ) {
	// this lane's segment in each pass: start | length << 20, position index (0xFF: none)
	uint32_t sw0, sw1, si0, si1;
	if ( cached ) {
	    sw0 = pc[lane];
	    sw1 = pc[64u + lane];
	    si0 = reinterpret_cast<const uint8_t *>(pc + 2u * SEG_MAX)[lane];
	    si1 = reinterpret_cast<const uint8_t *>(pc + 2u * SEG_MAX)[64u + lane];
	} else {
	    sw0 = sp.p_slot[lane];
	    sw1 = sp.p_slot[64u + lane];
	    si0 = sp.p_slot_seg[lane];
	    si1 = sp.p_slot_seg[64u + lane];
	}
	const bool have0 = si0 != 0xFFu, have1 = np > 1u && si1 != 0xFFu;
	// idle lanes shadow lane 0's segment (lane 0 always has one)
	sw0 = have0 ? sw0 : (uint32_t)__builtin_amdgcn_readfirstlane((int)sw0);
	sw1 = have1 ? sw1 : ( np > 1u ? (uint32_t)__builtin_amdgcn_readfirstlane((int)sw1) : sw0 );
	const uint32_t rel0 = sw0 & 0xFFFFFu, len0 = sw0 >> 20, rel1 = sw1 & 0xFFFFFu, len1 = sw1 >> 20;
	const uint32_t ng0 = ( sp.pass_len[0] + 15u ) >> 4, ng1 = np > 1u ? ( sp.pass_len[1] + 15u ) >> 4 : 0u;
	const uint32_t ns0 = ( ng0 + (uint32_t)TILE_GPS - 1u ) / (uint32_t)TILE_GPS;
	const uint32_t ns1 = ( ng1 + (uint32_t)TILE_GPS - 1u ) / (uint32_t)TILE_GPS;
	const uint32_t min0 = sp.pass_min[0], min1 = sp.pass_min[1];
	// the first pass's sums wait in registers while the second one uses the tile
	double hold[4] = { 0.0, 0.0, 0.0, 0.0 };
	float hold_a = 0.0f;
	double acc[4] = { 0.0, 0.0, 0.0, 0.0 };
	float2v esum = { 0.0f, 0.0f };			// sum of x^2 over the segment (two running halves)
	{
	    // Both passes in ONE run of tile steps: while the last step of the first pass is
	    // summed, the first step of the second is already on its way (corr_global_tiled's
	    // scheme -- TILE_LPW lanes fetch the TILE_K contiguous samples of one segment, every
	    // lane reads its own row back -- with a per-lane length, see seg_group)
	    constexpr int NLD = (int)( 64u / TILE_WPL );
	    const uint32_t sub = lane % TILE_LPW, grp = lane / TILE_LPW;
	    const uint32_t a0 = base + rel0, a1 = base + rel1;
	    uint32_t off0[NLD], off1[NLD];
#pragma unroll
	    for ( int i = 0; i < NLD; i++ ) {
		off0[i] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)( ( TILE_WPL * (uint32_t)i + grp ) << 2 ), (int)a0) + 4u * sub;
		off1[i] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)( ( TILE_WPL * (uint32_t)i + grp ) << 2 ), (int)a1) + 4u * sub;
	    }
	    float *wr = slab + grp * TILE_ROW + 4u * sub;
	    const float *rd = slab + lane * TILE_ROW;
	    float4 L[NLD];
#define MIFSK_SEGP_FETCH(USE1, S)								\
	    _Pragma("unroll")									\
	    for ( int i = 0; i < NLD; i++ ) {							\
		const uint32_t o_ = ( USE1 ) ? off1[i] : off0[i];				\
		const float4_u v = *reinterpret_cast<const float4_u *>(x + o_ + TILE_K * (S));	\
		L[i] = make_float4(v.x, v.y, v.z, v.w);						\
	    }
	    MIFSK_SEGP_FETCH(false, 0u)
	    TwGroup G = tw_group_load(tw, 0, lane);
	    uint32_t ps = 0, ls = 0;
	    const uint32_t total = ns0 + ( np > 1u ? ns1 : 0u );
	    for ( uint32_t t = 0; t < total; t++ ) {
#pragma unroll
		for ( int i = 0; i < NLD; i++ )
		    *reinterpret_cast<float4 *>(wr + TILE_WPL * (uint32_t)i * TILE_ROW) = L[i];
		const bool last = ls + 1u == ( ps ? ns1 : ns0 );
		const bool turn = last && ps + 1u < np;			// the next step opens the second pass
		const bool use1 = turn || ps == 1u;
		const uint32_t sn = turn ? 0u : ( last ? ls : ls + 1u );	// (the very last step is fetched twice)
		MIFSK_SEGP_FETCH(use1, sn)
		const uint32_t g0 = (uint32_t)TILE_GPS * ls;
		const uint32_t ng = ps ? ng1 : ng0, len = ps ? len1 : len0, lmin = ps ? min1 : min0;
		float4 xs[TILE_GPS][4];
#pragma unroll
		for ( int h = 0; h < TILE_GPS; h++ )
#pragma unroll
		    for ( int j = 0; j < 4; j++ )
			xs[h][j] = *reinterpret_cast<const float4 *>(rd + 16 * h + 4 * j);
#pragma unroll
		for ( int h = 0; h < TILE_GPS; h++ ) {
		    const uint32_t gi = g0 + (uint32_t)h;
		    // (the table has the group after the pass's last one: see plan_segments)
		    const TwGroup Gn = tw_group_load(tw, ( last && h == TILE_GPS - 1 ) ? 0u : gi + 1u, lane);
		    if ( gi < ng )
			seg_group(acc, esum, G, xs[h][0], xs[h][1], xs[h][2], xs[h][3], 16u * gi, len,
				  16u * gi + 16u <= lmin);
		    G = Gn;
		}
		if ( last ) {
		    if ( ps == 0u && np > 1u ) {
			hold[0] = acc[0]; hold[1] = acc[1]; hold[2] = acc[2]; hold[3] = acc[3];
			hold_a = esum.x + esum.y;
			acc[0] = acc[1] = acc[2] = acc[3] = 0.0;
			esum = float2v{0.0f, 0.0f};
		    }
		    ps++;
		    ls = 0;
		} else {
		    ls++;
		}
	    }
#undef MIFSK_SEGP_FETCH
	}
	// the tile has served: the partial sums take its place, by segment position
	wave_lds_sync();
	if ( lane == 0 ) {
	    *reinterpret_cast<double2_a16 *>(partD + 4u * SEG_MAX) = double2_a16{0.0, 0.0};
	    *reinterpret_cast<double2_a16 *>(partD + 4u * SEG_MAX + 2u) = double2_a16{0.0, 0.0};
	    partA[SEG_MAX] = 0.0f;
	}
	if ( np > 1u ) {
	    if ( have0 ) {
		*reinterpret_cast<double2_a16 *>(partD + 4u * si0) = double2_a16{hold[0], hold[1]};
		*reinterpret_cast<double2_a16 *>(partD + 4u * si0 + 2u) = double2_a16{hold[2], hold[3]};
		partA[si0] = hold_a;
		partRel[si0] = rel0;
	    }
	    if ( have1 ) {
		*reinterpret_cast<double2_a16 *>(partD + 4u * si1) = double2_a16{acc[0], acc[1]};
		*reinterpret_cast<double2_a16 *>(partD + 4u * si1 + 2u) = double2_a16{acc[2], acc[3]};
		partA[si1] = esum.x + esum.y;
		partRel[si1] = rel1;
	    }
	} else if ( have0 ) {
	    *reinterpret_cast<double2_a16 *>(partD + 4u * si0) = double2_a16{acc[0], acc[1]};
	    *reinterpret_cast<double2_a16 *>(partD + 4u * si0 + 2u) = double2_a16{acc[2], acc[3]};
	    partA[si0] = esum.x + esum.y;
	    partRel[si0] = rel0;
	}
	wave_lds_sync();
	bump(20);
	}	// sum_it
	const uint32_t tg1 = MIFSK_WCLOCK();
	cyc_g_pass += tg1 - tg0;
	// assemble: lane = window
	// delta = bound_c * 2^-53 * sum |x|, and sum |x| <= sqrt(B * sum x^2) (the sums of squares
	// are float sums: a little slack)
	const double dscale = (double)sp.bound_c * 1.0001 * 1.1102230246251565e-16 * sqrt((double)B);
	unsigned long long redo0 = 0ull, redo1 = 0ull;
	// A scan of at most 32 windows (the carrier-held coarse scan: 3 candidates) would leave half
	// the lanes idle: each window's segments are then shared by TWO lanes -- lane w takes the
	// even ones, lane 32 + w the odd ones -- and the two halves are added at the end (another
	// order of the same sum: two more roundings, counted in the plan's bound_c).
	const bool split = rot != nullptr && nwin_scan <= 32u;
	// (the sums of squares are float sums: a square below 2^-126 is rounded to a multiple of
	// 2^-149, one below 2^-150 to zero -- audio at 1e-23 of full scale would switch the guard off.
	// 3e-45 per sample covers what those roundings can lose; sqrt(a + b) <= sqrt(a) + sqrt(b): the
	// bound is delta = dscale sqrt(aw) + dfloor, the multiplication an fma.)
	const double dfloor = dscale * sqrt((double)B * 3.0e-45);
	for ( uint32_t g0 = 0; g0 < nwin_scan; g0 += 64u ) {
	    const uint32_t part = split ? lane >> 5 : 0u;
	    const uint32_t w = split ? ( lane & 31u ) : g0 + lane;
	    const bool active = w < nwin_scan;
	    const uint32_t wl = active ? w : 0u;		// the scan's window ...
	    const uint32_t ww = w_off + wl;			// ... is this one of the plan
	    const uint32_t j = udiv_magic(wl, nb, cfg.nbits_magic), k = wl - j * nb;
	    const uint32_t pw = cached ? pc[SEG_MAX + ww] : sp.p_win[ww];
	    const uint32_t first = pw & 0xFFu, cnt = ( pw >> 8 ) & 0xFFu, a_rel = pw >> 16;
	    const uint32_t cmax = wave_max_u32(cnt);
	    double mr = 0.0, mi = 0.0, sr = 0.0, si = 0.0;
	    float aw = 0.0f;
	    constexpr int AU = 4;			// segments per turn, every load issued before the first use
	    if ( rot ) {
		// rotation factors from the scan's own table, laid out for this loop: row i holds
		// segment i of every window, lanes read side by side, rows beyond a window's own
		// segments (and up to three turns beyond the longest window's) hold zeros
		// (mifsk_capi.cpp) -- a running pointer, no row arithmetic per segment
		const uint32_t stepi = split ? 2u : 1u;
		const double *tp = rot + 4 * ( (size_t)ww + (size_t)part * rstride );
		const size_t tstep = 4 * (size_t)rstride * stepi;
		const uint32_t turns = split ? ( cmax + 1u ) >> 1 : cmax;
		for ( uint32_t i0 = 0; i0 < turns; i0 += (uint32_t)AU ) {
		    double2_a16 p0[AU], p1[AU], t0[AU], t1[AU];
		    float as[AU];
#pragma unroll
		    for ( int u = 0; u < AU; u++ ) {
			const uint32_t i = ( i0 + (uint32_t)u ) * stepi + part;
			const uint32_t s = i < cnt ? first + i : (uint32_t)SEG_MAX;	// (beyond its own: the zero entry)
			p0[u] = *reinterpret_cast<const double2_a16 *>(partD + 4u * s);
			p1[u] = *reinterpret_cast<const double2_a16 *>(partD + 4u * s + 2u);
			as[u] = partA[s];
			t0[u] = *reinterpret_cast<const double2_a16 *>(tp);
			t1[u] = *reinterpret_cast<const double2_a16 *>(tp + 2);
			tp += tstep;
		    }
#pragma unroll
		    for ( int u = 0; u < AU; u++ ) {
			const double Sr = p0[u].x, Si = p0[u].y, Tr = p1[u].x, Ti = p1[u].y;
			// (Sr + i Si) (c + i m),  (c, m) = (cos, -sin) of the offset
			mr = fma(Sr, t0[u].x, mr);  mr = fma(-Si, t0[u].y, mr);
			mi = fma(Sr, t0[u].y, mi);  mi = fma(Si, t0[u].x, mi);
			sr = fma(Tr, t1[u].x, sr);  sr = fma(-Ti, t1[u].y, sr);
			si = fma(Tr, t1[u].y, si);  si = fma(Ti, t1[u].x, si);
			aw += as[u];
		    }
		}
		if ( split ) {
		    mr += __shfl_xor(mr, 32);
		    mi += __shfl_xor(mi, 32);
		    sr += __shfl_xor(sr, 32);
		    si += __shfl_xor(si, 32);
		    aw += __shfl_xor(aw, 32);
		}
	    } else
	    for ( uint32_t i0 = 0; i0 < cmax; i0 += (uint32_t)AU ) {
		double2_a16 p0[AU], p1[AU], t0[AU], t1[AU];
		float as[AU];
#pragma unroll
		for ( int u = 0; u < AU; u++ ) {
		    const uint32_t i = i0 + (uint32_t)u;
		    const bool in = i < cnt;
		    const uint32_t s = in ? first + i : (uint32_t)SEG_MAX;	// (beyond its own: the zero entry)
		    p0[u] = *reinterpret_cast<const double2_a16 *>(partD + 4u * s);
		    p1[u] = *reinterpret_cast<const double2_a16 *>(partD + 4u * s + 2u);
		    as[u] = partA[s];
		    // entry d of the stream's own table (--auto-carrier: the tones are the stream's);
		    // beyond its own segments a lane multiplies the zero entry by a factor that exists
		    const uint32_t d = in ? partRel[s] - a_rel : 0u;	// offset of the segment inside the window
		    const double *t = tw + 4 * (size_t)d;
		    t0[u] = *reinterpret_cast<const double2_a16 *>(t);
		    t1[u] = *reinterpret_cast<const double2_a16 *>(t + 2);
		}
#pragma unroll
		for ( int u = 0; u < AU; u++ ) {
		    const double Sr = p0[u].x, Si = p0[u].y, Tr = p1[u].x, Ti = p1[u].y;
		    mr = fma(Sr, t0[u].x, mr);  mr = fma(-Si, t0[u].y, mr);
		    mi = fma(Sr, t0[u].y, mi);  mi = fma(Si, t0[u].x, mi);
		    sr = fma(Tr, t1[u].x, sr);  sr = fma(-Ti, t1[u].y, sr);
		    si = fma(Tr, t1[u].y, si);  si = fma(Ti, t1[u].x, si);
		    aw += as[u];
		}
	    }
	    const double delta = fma(dscale, sqrt((double)aw), dfloor);
	    const bool stable = (float)( mr - delta ) == (float)( mr + delta )
			     && (float)( mi - delta ) == (float)( mi + delta )
			     && (float)( sr - delta ) == (float)( sr + delta )
			     && (float)( si - delta ) == (float)( si + delta );
	    const bool wanted = active && j >= c0 && part == 0u;
	    if ( wanted && stable )
		mags[( j - c0 ) * nb + k] = band_mag2_exact(mr, mi, sr, si, cfg.magscalar);
	    const unsigned long long again = __ballot(wanted && !stable);
	    if ( g0 == 0u )
		redo0 = again;
	    else
		redo1 = again;
	}
	const uint32_t tg2 = MIFSK_WCLOCK();
	cyc_g_asm += tg2 - tg1;
	// A second look at the windows the bound does not settle (round 6).  delta above prices the
	// oracle's own index-order chain at its worst case, (B - 1) u sum |x|.  What that chain can
	// really be off by is u times the sum of its partial sums' magnitudes (acc_k = (acc_{k-1} +
	// x_k w_k)(1 + d_k), |d_k| <= u: the error is sum d_k acc_k), and inside segment s -- L_s
	// samples, A_s = sum |x| <= sqrt(L_s E_s) -- no partial sum exceeds |P_s| + A_s, P_s the sum
	// over the segments before it: the assembled partial sum to within this assembly's own error
	// (u c' sum A, c' = bound_c - B, added to every |P_s|).  So
	//     delta2 = u [ c' sum A_s + sum_s L_s ( |P_s| + A_s + u c' sum A ) ]
	// per component -- a third of delta where the band is silent (its partial sums stay small)
	// and on the carrying band alike.  One lane per unsettled window walks its segments once
	// more, in order; what settles is written, the rest goes on to the index-order sum.  (A
	// window with a NaN or an infinity settles nowhere.)
	for ( uint32_t h = 0; h < 2u; h++ ) {
	    const unsigned long long m = h ? redo1 : redo0;
	    if ( !m )
		continue;
	    bool settled = false;
	    if ( ( m >> lane ) & 1ull ) {
		const uint32_t wl = 64u * h + lane;
		const uint32_t ww = w_off + wl;
		const uint32_t j = udiv_magic(wl, nb, cfg.nbits_magic), k = wl - j * nb;
		const uint32_t pw = sp.p_win[ww];
		const uint32_t first = pw & 0xFFu, cnt = ( pw >> 8 ) & 0xFFu, a_rel = pw >> 16;
		const double cown = (double)sp.bound_c - (double)B;
		double X[4] = { 0.0, 0.0, 0.0, 0.0 }, run[4] = { 0.0, 0.0, 0.0, 0.0 };
		double asum = 0.0, lsum = 0.0;
		for ( uint32_t i = 0; i < cnt; i++ ) {
		    const uint32_t sg = first + i;
		    const double2_a16 q0 = *reinterpret_cast<const double2_a16 *>(partD + 4u * sg);
		    const double2_a16 q1 = *reinterpret_cast<const double2_a16 *>(partD + 4u * sg + 2u);
		    const double L = (double)sp.seg_len[sg];
		    // (partA is a float sum of float squares: a little slack, and what underflow can lose)
		    const double A = sqrt(L * ( (double)partA[sg] + L * 3.0e-45 )) * 1.0001;
		    const double *t = rot ? rot + 4 * ( (size_t)i * rstride + ww )
					  : tw + 4 * (size_t)( partRel[sg] - a_rel );
		    const double2_a16 r0 = *reinterpret_cast<const double2_a16 *>(t);
		    const double2_a16 r1 = *reinterpret_cast<const double2_a16 *>(t + 2);
#pragma unroll
		    for ( int c = 0; c < 4; c++ )
			run[c] = fma(L, fabs(X[c]) + A, run[c]);
		    X[0] = fma(q0.x, r0.x, X[0]);  X[0] = fma(-q0.y, r0.y, X[0]);
		    X[1] = fma(q0.x, r0.y, X[1]);  X[1] = fma(q0.y, r0.x, X[1]);
		    X[2] = fma(q1.x, r1.x, X[2]);  X[2] = fma(-q1.y, r1.y, X[2]);
		    X[3] = fma(q1.x, r1.y, X[3]);  X[3] = fma(q1.y, r1.x, X[3]);
		    asum += A;
		    lsum += L;
		}
		const double u = 1.1102230246251565e-16;
		const double own = cown * asum;				// this assembly against the exact sum, in units of u
		bool ok = true;
#pragma unroll
		for ( int c = 0; c < 4; c++ ) {
		    const double d2 = 1.0001 * u * ( own + run[c] + lsum * ( u * own ) );
		    ok = ok && (float)( X[c] - d2 ) == (float)( X[c] + d2 );
		}
		if ( ok ) {
		    mags[( j - c0 ) * nb + k] = band_mag2_exact(X[0], X[1], X[2], X[3], cfg.magscalar);
		    settled = true;
		}
	    }
	    const unsigned long long still = m & ~__ballot(settled);
	    if ( cnt_on && still != m )
		bump(MIFSK_CNT_SEG_SECOND_LOOKS);
	    if ( h )
		redo1 = still;
	    else
		redo0 = still;
	}
	// ... and what is left: again, in index order
	for ( uint32_t h = 0; h < 2u; h++ ) {
	    const unsigned long long m = h ? redo1 : redo0;
	    if ( !m )
		continue;
	    bump(21);
	    wave_lds_sync();				// (the partials are not needed any more: the tile
	    uint32_t *list = plan_cache() + 2u * kPlanWords;	//  is a tile again; the windows' list sits behind the plan words)
	    const uint32_t nredo = (uint32_t)__popcll(m);
	    if ( ( m >> lane ) & 1ull )
		list[__popcll(m & ( ( 1ull << lane ) - 1ull ))] = 64u * h + lane;
	    wave_lds_sync();
	    const uint32_t w = list[lane < nredo ? lane : 0u];
	    const uint32_t j = udiv_magic(w, nb, cfg.nbits_magic), k = w - j * nb;
	    const uint32_t a = base + zz.at(j) + cfg.bit_offset[k & 63u];
	    double acc[4] = { 0.0, 0.0, 0.0, 0.0 };
	    corr_global_tiled(tw, x, a, nredo, B, lane, slab, acc);
	    if ( lane < nredo )
		mags[( j - c0 ) * nb + k] = band_mag2_exact(acc[0], acc[1], acc[2], acc[3], cfg.magscalar);
	    wave_lds_sync();
	}
	cyc_g_redo += MIFSK_WCLOCK() - tg2;
	return ( redo0 | redo1 ) == 0ull;
    }

    // `reuse0`: candidate 0 of this scan is the candidate 0 of the scan just made at
    // the same cursor with the same expect string (the fine rescan after a
    // carrier-held coarse scan, minimodem.c:1373 after :1265: same try_first, same
    // data string) -- its score is known and is not computed again.  The
    // reference computes it twice from the same samples; the result is the same.
    __device__ __forceinline__ ScanResult scan( uint32_t base, const ZigZag &zz, uint32_t first,
	    float limit, uint32_t kind, bool carrier_held, bool reuse0 = false )
    {
	ScanResult r;
	r.conf = 0.0f; r.ampl = 0.0f; r.bits = 0; r.start = 0;
	if ( !reuse0 )
	    k0_valid = false;
	if ( zz.J == 0 )
	    return r;
	const uint32_t t0 = MIFSK_WCLOCK();
	// the first candidate may be a lattice frame that is already scored
	// (those are scored against the data string)
	{
	    const uint32_t hit = kind == 0u ? lattice_lookup(base + first) : ~0u;
	    if ( hit != ~0u && !k0_valid ) {
		k0.conf = lane_bcast(l_conf, hit);
		k0.ampl = lane_bcast(l_ampl, hit);
		const uint32_t blo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)l_bits, (int)hit);
		const uint32_t bhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)( l_bits >> 32 ), (int)hit);
		k0.bits = ( (uint64_t)bhi << 32 ) | blo;
		k0_valid = true;
		bump(MIFSK_CNT_CACHE_HITS);
	    }
	}
	bool done = false;
	uint32_t c_begin = 0;
	if ( k0_valid ) {
	    // candidate 0 first, as in the scan order (fsk.c:492-501)
	    if ( r.conf < k0.conf ) {
		r.conf = k0.conf;
		r.ampl = k0.ampl;
		r.bits = k0.bits;
		r.start = first;
		if ( r.conf >= limit ) {
		    cyc_scan += MIFSK_WCLOCK() - t0;
		    return r;
		}
	    }
	    c_begin = 1;
	}
	const uint32_t nb = cfg.n_bits;
	uint32_t qmax = g.mags_cap / nb;
	if ( qmax > 64u ) qmax = 64u;
	for ( uint32_t c0 = c_begin; c0 < zz.J && !done; c0 += qmax ) {
	    const uint32_t Q = zz.J - c0 < qmax ? zz.J - c0 : qmax;
	    // extent of this chunk's candidates, in closed form (see ZigZag)
	    const uint32_t iend = c0 + Q - 1u;
	    const uint32_t lu = iend > 2u * zz.D ? iend : ( ( iend & 1u ) ? iend : iend - 1u );
	    const uint32_t ld = ( iend < 2u * zz.D ? iend : 2u * zz.D ) & ~1u;
	    const bool has_up = lu >= ( c0 > 1u ? c0 : 1u ) && lu <= iend && iend >= 1u;
	    const bool has_down = ld >= 2u && ld >= c0;
	    const uint32_t thi = has_up ? zz.at(lu) : zz.at(c0);
	    const uint32_t tlo = has_down ? zz.at(ld) : zz.at(c0);
	    const uint32_t lo = base + tlo, hi = base + thi + cfg.last_reach;
	    bool use_slab = g.slab_cap != 0u && hi - lo + 8u <= g.slab_cap;
	    const uint32_t ts0 = MIFSK_WCLOCK();
	    if ( use_slab && ( ring || lo < slab_lo || hi > slab_hi ) ) {
		// Without a carrier the searches that follow advance through the
		// stream and reuse what is staged now: fill the slab.  With the
		// carrier held the next SCAN is many frames away: stage what this
		// search reads and no more.
		const uint32_t need = ( hi - lo + 7u ) & ~3u;
		const uint32_t take = ( carrier_held || ring ) ? need : g.slab_cap;
		slab_lo = plain_slab_origin(lo);
		slab_hi = lo + take;
		drop_prefetch();		// (the cursor left the lattice)
		stage_slab(base, lo, take);
		wave_lds_sync();
		    bump(MIFSK_CNT_STAGES);
	    }
	    const uint32_t ts1 = MIFSK_WCLOCK();
	    bool shared = false;
	    if constexpr ( NQ == kTiled ) {
		// long windows, the whole (rest of the) scan in this chunk, far enough from the
		// end of the stream that whole tile steps may be loaded behind every window
		// The carrier-held coarse scan runs the plan that also holds the fine scan's windows
		// (minimodem.c:1265 then :1373 at the same cursor): if the fine scan follows, the
		// partial sums are still on the tile and only its windows' assembly is left.
		uint32_t pi = zz.id & 3u, w_off = 0u;
		bool sum_it = true;
		if ( zz.id == 1u && cfg.seg[4].valid ) {
		    pi = 4u;
		} else if ( zz.id == 3u && cfg.seg[4].valid && un_valid && un_base == base ) {
		    pi = 4u;
		    w_off = cfg.seg_union_first_fine;
		    sum_it = false;
		}
		shared = g.tiled && !ring && zz.id < 4u && cfg.seg[pi].valid && c0 + Q == zz.J
		      && base + cfg.seg[pi].span_hi + cfg.bit_nsamples + 192u <= N
		      && base + cfg.seg[pi].span_hi + cfg.bit_nsamples + 192u >= base;
		if ( shared ) {
		    const bool kept = seg_correlate(base, zz, c0, pi, w_off, sum_it);
		    un_valid = kept && pi == 4u;
		    un_base = base;
		} else {
		    un_valid = false;		// (the windows go through the tile one by one)
		}
	    }
	    if ( !shared )
		scan_correlate(base, zz, c0, Q, use_slab);
	    wave_lds_sync();
	    const uint32_t ts2 = MIFSK_WCLOCK();
	    FrameOut f;
	    f.conf = 0.0f; f.ampl = 0.0f; f.bits = 0;
	    if ( lane < Q ) {
		uint32_t fb = 0u;
		f = frame_confidence_any(&mags[lane * nb], cfg.req_mask[kind], cfg.req_val[kind], nb, fb);
		if ( cnt_on && fb )
		    bump(MIFSK_CNT_CONF_FALLBACKS);
	    }
	    wave_lds_sync();			// mags[] is free again
	    cyc_s_stage += ts1 - ts0;
	    cyc_s_corr += ts2 - ts1;
	    cyc_s_conf += MIFSK_WCLOCK() - ts2;
	    bump(MIFSK_CNT_BATCHES);
	    bump(MIFSK_CNT_POSITIONS, Q);
	    if ( c0 == 0u ) {			// candidate 0's own score, for a rescan at this cursor
		k0.conf = lane_bcast(f.conf, 0);
		k0.ampl = lane_bcast(f.ampl, 0);
		const uint32_t blo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)f.bits, 0);
		const uint32_t bhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)( f.bits >> 32 ), 0);
		k0.bits = ( (uint64_t)bhi << 32 ) | blo;
		k0_valid = kind == 0u;
	    }
	    // fsk.c:492-501 over the chunk, in scan order on lane values
	    uint32_t win = ~0u;
	    if ( NQ <= 0 && r.conf < limit ) {
		// While the best so far is below the limit, the loop below ends at the FIRST candidate
		// that reaches the limit (it necessarily beats everything tried before it); if none
		// does, the winner is the first candidate attaining the chunk's maximum, provided
		// that exceeds the best so far (strict >; a NaN never wins).  Two ballots and a DPP
		// maximum instead of a readlane / compare / branch round trip per candidate.
		const bool cand = lane < Q && f.conf == f.conf;
		const unsigned long long hit = __ballot(cand && f.conf >= limit);
		if ( hit ) {
		    win = (uint32_t)__ffsll((long long)hit) - 1u;
		    r.conf = lane_bcast(f.conf, win);
		    done = true;
		} else {
		    const float best = wave_max_f32(cand ? f.conf : -INFINITY);
		    if ( r.conf < best ) {
			win = (uint32_t)__ffsll((long long)__ballot(cand && f.conf == best)) - 1u;
			r.conf = best;
		    }
		}
	    } else
	    for ( uint32_t i = 0; i < Q; i++ ) {
		const float c = lane_bcast(f.conf, i);
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
		r.ampl = lane_bcast(f.ampl, win);
		const uint32_t blo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)f.bits, (int)win);
		const uint32_t bhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)( f.bits >> 32 ), (int)win);
		r.bits = ( (uint64_t)bhi << 32 ) | blo;
		r.start = zz.at(c0 + win);
	    }
	}
	cyc_scan += MIFSK_WCLOCK() - t0;
	return r;
    }
};

// v[src lane] for a per-lane source index (ds_bpermute)
__device__ __forceinline__ float lane_gather( float v, uint32_t src )
{
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((int)( src << 2 ), __builtin_bit_cast(int, v)));
}
__device__ __forceinline__ uint64_t lane_gather64( uint64_t v, uint32_t src )
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)( src << 2 ), (int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)( src << 2 ), (int)(uint32_t)( v >> 32 ));
    return ( (uint64_t)hi << 32 ) | lo;
}

extern __shared__ __attribute__((aligned(16))) unsigned char mifsk_wave_smem[];
constexpr size_t kCntBytes = ( MIFSK_NCOUNTERS * sizeof(uint32_t) + 15u ) & ~(size_t)15;	// work counters, first in LDS

// (the wide-staging instantiation runs where a wave has >= 10 KiB of LDS to
// itself, i.e. at most 2-3 waves per SIMD: it may use 256 VGPRs)
#ifndef MIFSK_WAVE_OCC
#define MIFSK_WAVE_OCC 4	// waves per SIMD the narrow-staging instantiations are compiled for
#endif
// ST: the instantiation behind mifsk_demod_slab (state in, state out); the plain kernels do
// not carry its code or its registers
// RA: the instantiation that can do RING addressing and --auto-carrier; the plain ones do not
// carry that code or the registers it keeps alive either
template <int SV, int NQ, bool ST = false, bool RA = true>
__global__ __launch_bounds__(64, NQ == kTiled ? 2 : ( SV >= 10 ? 2 : MIFSK_WAVE_OCC ))
void demod_wave_kernel( const DevCfg *__restrict__ cfgp, const double *__restrict__ tw_default,
	mifsk_demod_io io, WaveGeom g, WaveAuto au )
{
    const DevCfg &cfg = *cfgp;
    const uint32_t s = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    const bool t0 = lane == 0;

    uint32_t *cnt = reinterpret_cast<uint32_t *>(mifsk_wave_smem);
    float2 *mags = reinterpret_cast<float2 *>(mifsk_wave_smem + kCntBytes);
    float *slab = reinterpret_cast<float *>(mifsk_wave_smem + kCntBytes + (size_t)g.mags_cap * sizeof(float2));
    if ( lane < MIFSK_NCOUNTERS )
	cnt[lane] = 0;

    const float *x = io.d_samples + (size_t)s * io.stream_stride;
    uint32_t N = io.d_nsamples ? io.d_nsamples[s] : io.nsamples;
    if ( io.nstreams > 1 && (size_t)N > io.stream_stride )
	N = (uint32_t)io.stream_stride;		// never trust a length beyond the row
    // How far this stream's row may be over-read (in samples from its start)
    // without leaving the batch: the rows after it, or for the last row its own
    // length.  The linear LATTICE fetches whole rounds with no per-lane bounds
    // logic and needs at least one round of room.
    const uint64_t rows_after = (uint64_t)( io.nstreams - 1 - (int)s ) * io.stream_stride;
    uint32_t safe_limit = rows_after == 0 ? N
			: rows_after > 0xFFFF0000ull ? 0xFFFF0000u : (uint32_t)rows_after;
    bool lattice_ok = g.lat_mode != LAT_NONE;
    if ( Wave<SV, NQ>::kLinear && safe_limit < Wave<SV, NQ>::kRoundFloats ) {
	lattice_ok = false;
	safe_limit = Wave<SV, NQ>::kRoundFloats;
    }
    float *ring = ( RA && g.ring_exact ) ? au.d_ring + (size_t)s * g.ring_stride : nullptr;
    const bool autodetect = RA && g.autodetect;
    if ( ring )
	lattice_ok = false;			// RING addressing: every frame through the general path
    // chained launches (launch_demod_wave): this call takes the stream up to au.limit only
    bool cut = false;
    if constexpr ( ST ) {
	if ( au.d_state && au.limit != 0u && au.limit < N ) {
	    N = au.limit;
	    cut = true;
	}
    }
    const double *tw = tw_default;
    double *tw_own = nullptr;
    if ( autodetect ) {
	tw_own = au.d_tw_scratch + (size_t)s * g.tw_entries * 4u;
	tw = tw_own;
    }



    WaveOut o;
    o.fcap = (uint32_t)( io.frames_cap > 0xFFFFFFFFull ? 0xFFFFFFFFull : io.frames_cap );
    o.ecap = (uint32_t)( io.episodes_cap > 0xFFFFFFFFull ? 0xFFFFFFFFull : io.episodes_cap );
    o.bytes = io.d_bytes ? io.d_bytes + (size_t)s * io.frames_cap : nullptr;
    o.bits = io.d_bits ? io.d_bits + (size_t)s * io.frames_cap : nullptr;
    o.frames = io.d_frames ? io.d_frames + (size_t)s * io.frames_cap : nullptr;
    o.eps = io.d_episodes ? io.d_episodes + (size_t)s * io.episodes_cap : nullptr;

    Wave<SV, NQ> ctx(cfg, g, tw, x, N, mags, slab, ring, safe_limit, cnt, io.d_counters != nullptr);

    if constexpr ( NQ == kTiled ) {
	if ( g.tiled && cfg.seg[1].valid && cfg.seg[3].valid )
	    ctx.plan_cache_fill();
    }

    // reference loop state (minimodem.c:1079-1088,1132-1133), uniform in the wave
    bool carrier = false;
    float confidence_total = 0.0f, amplitude_total = 0.0f;
    uint32_t nframes_decoded = 0;
    uint64_t carrier_nsamples = 0;
    uint32_t noconfidence = 0;
    uint32_t advance = 0;
    float track_amplitude = 0.0f, peak_confidence = 0.0f;
    // (base, rp): absolute index of samplebuf[0], and the file position;
    // samples_nvalid = rp - base
    uint32_t base = 0, rp = 0;
    const uint32_t bufsize = g.bufsize, half = g.bufsize / 2u;
    // --auto-carrier
    int carrier_band = -1;				// `static int carrier_band = -1`
    int first_band = -1;
    uint32_t b_mark = cfg.b_mark;

    uint32_t n_out_frames = 0, n_out_bytes = 0, n_out_eps = 0, ep_first = 0, ep_b_mark = 0;
    uint32_t status = 0;

    // mifsk_demod_slab: this row is the stream from index `origin` on, the loop resumes
    // from the state the call before left (minimodem.c:1079-1088,1132-1133,1144-1174).
    // Positions inside the kernel are relative to the row; what leaves it (frame starts,
    // episode frame indices, the saved state) counts from the start of the stream.
    const bool stateful = ST && au.d_state != nullptr;
    const bool last_slab = !stateful || au.final != 0u || ( au.limit != 0u && !cut );
    uint64_t origin = 0;
    uint32_t frame_base = 0;			// frames emitted by the calls before
    bool resumable = true;
    if ( stateful ) {
	if ( au.d_origin )
	    origin = au.d_origin[s];
	const mifsk_stream_state st = au.d_state[s];
	if ( st.flags & MIFSK_STATE_FINISHED ) {
	    resumable = false;			// nothing more to do for this stream
	    // (a stream whose loop was aborted stays aborted for its caller; a chained launch's
	    // later chunks leave the status word the failed chunk wrote alone -- below)
	    if ( au.append == 0u )
		status |= st.status & MIFSK_STREAM_ABORTED;
	} else if ( st.flags & MIFSK_STATE_STARTED ) {
	    if ( st.base < origin || st.base - origin > (uint64_t)N || st.rp < st.base ) {
		status |= MIFSK_STREAM_ABORTED;	// the caller dropped samples the loop still needs
		resumable = false;
	    } else {
		base = (uint32_t)( st.base - origin );
		rp = base + (uint32_t)( st.rp - st.base );
		advance = st.advance;
		carrier = ( st.flags & MIFSK_STATE_CARRIER ) != 0u;
		carrier_nsamples = st.carrier_nsamples;
		confidence_total = st.confidence_total;
		amplitude_total = st.amplitude_total;
		nframes_decoded = st.nframes_decoded;
		noconfidence = st.noconfidence;
		track_amplitude = st.track_amplitude;
		peak_confidence = st.peak_confidence;
		carrier_band = st.carrier_band;
		first_band = st.first_band;
		b_mark = st.b_mark;
		ep_b_mark = st.ep_b_mark;
		ep_first = st.ep_first;
		frame_base = (uint32_t)st.nframes_total;
		if ( au.append ) {
		    // the outputs continue where the call before stopped instead of at index 0
		    n_out_frames = frame_base;
		    n_out_bytes = st.nbytes_total;
		    n_out_eps = st.nepisodes_total;
		    status = st.status;
		    frame_base = 0u;
		}
		if ( autodetect && carrier_band >= 0 ) {	// the tones found before: this stream's table again
		    wave_build_table(tw_own, au.d_cs, (uint32_t)carrier_band,
				     (uint32_t)( carrier_band + g.b_shift ), cfg.bit_nsamples, g.fftsize, g.tw_entries);
		    ctx.load_resident_twiddles();
		}
	    }
	}
    }
    // With more of the stream to come, a pass of the loop is run only when a whole
    // samplebuf beyond its cursor is in the row: then every refill is a full half buffer
    // and every sample a search can read is the stream's, exactly as in a single call.
    // Lattice frames are accepted up to that horizon (N_lat), the general path stops at it.
    const uint32_t N_lat = last_slab ? N : ( N > g.bufsize ? N - g.bufsize : 0u );
    bool paused = false;


    uint32_t cyc_bulk = 0, cyc_general = 0, cyc_b_replay = 0, cyc_b_out = 0;
    const uint32_t t_start = MIFSK_WCLOCK();
#ifdef MIFSK_PROFILE
    const uint32_t t_wall0 = (uint32_t)wall_clock64();
#endif


    const ZigZag zc0(cfg, 0u), zc1(cfg, 1u), zf0(cfg, 2u), zf1(cfg, 3u);
    const uint32_t la = cfg.lock_advance;
    // the episodes' running totals of confidence and amplitude are reported in episode records
    // and carried in the saved state: nowhere else (the lattice replay skips them otherwise)
    // (a chained launch saves state too, but only for its own next chunk, which wants what this
    // one wants)
    const bool want_totals = o.eps != nullptr || ( ST && au.append == 0u );

    // every pass through the loop moves the cursor forward (or ends the loop):
    // a bound far above anything reachable turns a logic error into a flagged
    // stream instead of a hung GPU
    uint32_t guard = 0;
    const uint32_t guard_max = 2u * N + 1024u;
    for (;;) {
	if ( ++guard > guard_max ) {
	    status |= MIFSK_STREAM_ABORTED;
	    break;
	}
	if ( !resumable )
	    break;

	// ------------------------------------------------------------------
	// Bulk acceptance of lattice frames.  While carrier is held and the
	// cursor lands on the lattice, the reference's iteration for frame k
	// reduces to: first try wins the coarse scan (c >= limit), no refine
	// (c >= 0.75 peak), no squelch (a >= 0.25 track, c > threshold).  Those
	// predicates and the f32 state recurrences are replayed here in frame
	// order from the scored (confidence, amplitude) pairs; the first frame
	// that fails any of them falls through to the general path below.
	// (Every frame accepted here lies wholly inside the stream, where
	// samples_nvalid >= half the buffer >= everything a search reads: the
	// launcher enables the lattice only for such geometries.)
	// ------------------------------------------------------------------
	if ( lattice_ok && carrier && advance && base <= N_lat && advance <= N_lat - base ) {
	    const uint32_t t_bulk = MIFSK_WCLOCK();
	    const uint32_t first = cfg.try_first[1];
	    const uint32_t nb = base + advance;		// cursor of the next iteration
	    const uint32_t p = nb + first;
	    uint32_t e0 = ctx.lattice_lookup(p);
	    // frames from cursor nb on that still see expect_nsamples (minimodem.c:1229)
	    const uint32_t room = N_lat - nb >= cfg.expect_nsamples
				? udiv_magic(N_lat - nb - cfg.expect_nsamples, la, cfg.la_magic) + 1u : 0u;
	    if ( e0 == ~0u && room && !ctx.pause ) {
		uint32_t F = ctx.spec < room ? ctx.spec : room;
		ctx.lattice_block(p, F);
		e0 = 0;
	    }
	    bool progressed = false, broke = false;
	    if ( e0 != ~0u ) {
		uint32_t K = ctx.lat_n - e0;
		K = K < room ? K : room;
		// this lane's candidate (lane k <-> entry e0 + k)
		const bool have = lane < K;
		float cv, av;
		if ( e0 == 0u ) {
		    cv = ctx.l_conf;
		    av = ctx.l_ampl;
		} else {
		    cv = lane_gather(ctx.l_conf, e0 + lane);
		    av = lane_gather(ctx.l_ampl, e0 + lane);
		}
		cv = have ? cv : 0.0f;
		av = have ? av : 0.0f;
		// Replay the f32 state recurrences over all K candidates as a lane
		// scan (replay_scan_asm): lane k computes the state AFTER frame k by
		// applying the reference's update to its lower neighbour's state.
		// With a search step of one sample a "refine" is a flag and no search
		// (minimodem.c:1357): the frame is an ordinary one whose only side
		// effect is the reset of the running peak -- replayed as such.
		const bool soft = cfg.try_step[1] <= 1u;
		const uint32_t t_rp = MIFSK_WCLOCK();
		float xt = ( track_amplitude + av ) / 2.0f;		// minimodem.c:1391
		float xpk = peak_confidence < cv ? cv : peak_confidence;	// :1392-1393
		if ( soft && cv < peak_confidence * 0.75f )
		    xpk = cv;						// :1278-1281, then :1392
		float xsc = confidence_total + cv;			// :1397-1398
		float xsa = amplitude_total + av;
		float my_t = track_amplitude, my_pk = peak_confidence;
		float my_sc = confidence_total, my_sa = amplitude_total;
		// (soft, and no loop state kept for a later call: the running peak is read by nothing)
		if ( soft && !ST )
		    replay_scan_track(xt, xsc, xsa, my_t, my_sc, my_sa, cv, av, K, want_totals);
		else if ( soft )
		    replay_scan_soft(xt, xpk, xsc, xsa, my_t, my_pk, my_sc, my_sa, cv, av, K, lane, want_totals);
		else
		    replay_scan_asm(xt, xpk, xsc, xsa, my_t, my_pk, my_sc, my_sa, cv, av, K, want_totals);
		const bool ok = have
		    && cv > 0.0f && cv >= cfg.search_limit	// fsk.c:492,499: first try ends the scan
		    && ( soft || !( cv < my_pk * 0.75f ) )	// minimodem.c:1278
		    && !( av < my_t * 0.25f )			// minimodem.c:1286
		    && !( cv <= cfg.conf_threshold );		// minimodem.c:1292
		const unsigned long long bad = __ballot(have && !ok);
		const uint32_t n = bad ? (uint32_t)__ffsll((long long)bad) - 1u : K;
		ctx.run += n;
		// (frame n is the next iteration's and not a trivial one: straight to the general
		// path below -- a second replay would only find n = 0 again)
		broke = n < K;
		if ( n < K ) {			// the lattice broke here: remember how long it held
		    ctx.spec = ctx.run < g.lat_fmin ? g.lat_fmin
			     : ( ctx.run < g.lat_fmax ? ctx.run : g.lat_fmax );
		    ctx.cold = ctx.run ? 0u : ctx.cold + 1u;
		    ctx.run = 0;
		    if ( ctx.cold >= 4u ) {	// it keeps missing: plain searches for a while
			ctx.cold = 0;
			ctx.pause = 32;
		    }
		} else if ( e0 + K == ctx.lat_n ) {	// a whole block held: speculate further
		    ctx.spec = 2u * ctx.spec < g.lat_fmax ? 2u * ctx.spec : g.lat_fmax;
		}
		const uint32_t t_out = MIFSK_WCLOCK();
		cyc_b_replay += t_out - t_rp;
		if ( n ) {
		    float track, peak, ctot, atot;
		    if ( n == K ) {
			track = lane_bcast(xt, K - 1u);
			peak = lane_bcast(xpk, K - 1u);
			ctot = lane_bcast(xsc, K - 1u);
			atot = lane_bcast(xsa, K - 1u);
		    } else {
			track = lane_bcast(my_t, n);
			peak = lane_bcast(my_pk, n);
			ctot = lane_bcast(my_sc, n);
			atot = lane_bcast(my_sa, n);
		    }
		    // outputs of frames 0..n-1, one lane each
		    const bool mine = lane < n;
		    const uint64_t fb = e0 == 0u ? ctx.l_bits : lane_gather64(ctx.l_bits, e0 + lane);
		    uint64_t db = 0;
		    bool suppressed = false;
		    if ( mine ) {
			db = data_bits_of(cfg, fb);
			suppressed = cfg.do_rx_sync && db == cfg.sync_byte;
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
		    carrier_nsamples += (uint64_t)n * ( cfg.frame_nsamples + first - cfg.overscan );
		    base = nb + ( n - 1u ) * la;
		    advance = la;
		    // the half-buffer refills those n iterations made (minimodem.c:1158-1174)
		    // (one refill per iteration whenever fewer than half a buffer is
		    // valid: after n iterations the file position is the first
		    // rp + m * half that leaves at least half a buffer beyond the cursor)
		    while ( rp < base + half && rp < N )
			rp += N - rp < half ? N - rp : half;
		    ctx.bump(MIFSK_CNT_BULK_FRAMES, n);
		    progressed = true;
		    cyc_b_out += MIFSK_WCLOCK() - t_out;
		}
	    }
	    cyc_bulk += MIFSK_WCLOCK() - t_bulk;	// (block evaluation included)
	    if ( progressed && !broke )
		continue;
	}
	const uint32_t t_gen = MIFSK_WCLOCK();

	// ------------------------------------------------------------------
	// one iteration of the reference's loop
	// ------------------------------------------------------------------
	if ( !last_slab ) {
	    if ( RA && ring ) {
		// RING addressing with more of the stream to come: a search reads nothing beyond
		// samples_nvalid that is not in the (persistent) ring already, so a pass needs
		// exactly what the reference's fread would deliver -- a full half buffer whenever
		// fewer than half is valid after the shift (minimodem.c:1146-1174)
		const uint64_t nb = (uint64_t)base + advance;
		const uint64_t nrp = advance == bufsize ? nb : (uint64_t)rp;
		if ( nrp < nb || nrp - nb < (uint64_t)half ) {
		    if ( nrp > (uint64_t)N || (uint64_t)N - nrp < (uint64_t)half ) {
			paused = true;		// the refill is not in the row yet
			break;
		    }
		}
	    } else if ( (uint64_t)base + advance + bufsize > (uint64_t)N ) {
		paused = true;			// it could read beyond the row: the next slab resumes here
		break;
	    }
	}
	if ( advance == bufsize ) {				// minimodem.c:1146-1149: samples_nvalid = 0
	    base += advance;
	    rp = base;
	    advance = 0;
	}
	if ( base > rp )
	    break;						// (cannot happen: samples_nvalid >= 0)
	if ( advance ) {					// :1150-1156
	    if ( advance > rp - base )
		break;
	    if ( ring ) {
		// memmove(samplebuf, samplebuf + advance, (size - advance) floats):
		// ascending 64-cell chunks, each read before it is written
		// (a cell is never read after it has been written: reads run `advance`
		// cells ahead of the writes; one fence at the end publishes the lot)
		const uint32_t cnt = bufsize - advance;
		for ( uint32_t j = 0; j < cnt; j += 64u ) {
		    const uint32_t c = j + lane;
		    if ( c < cnt )
			ring[c] = ring[c + advance];
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
	    }
	    base += advance;
	    advance = 0;
	}
	if ( rp - base < half ) {				// :1158-1174
	    const uint32_t r = N - rp < half ? N - rp : half;
	    if ( ring ) {
		const uint32_t nv = rp - base;
		for ( uint32_t j = lane; j < r; j += 64u )
		    ring[nv + j] = x[rp + j];
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
	    }
	    rp += r;
	}
	const uint32_t nvalid = rp - base;
	if ( nvalid == 0 )					// :1176
	    break;

	if ( autodetect && carrier_band < 0 ) {		// :1179-1220
	    uint32_t i = 0;
	    const float nps = g.nps;
	    int band = -1;
	    while ( (float)i + nps <= (float)nvalid ) {		// float arithmetic, as there
		band = wave_detect_window(x + base + i, (uint32_t)nps, au.d_cs, g.fftsize, g.nbands,
					  g.auto_threshold);
		ctx.bump(22);
		if ( band >= 0 )
		    break;
		i = (uint32_t)( (float)i + nps );
	    }
	    advance = (uint32_t)( (float)i + nps );		// :1193-1195
	    if ( advance > nvalid )
		advance = nvalid;
	    if ( band < 0 )
		continue;
	    const int b_space = band + g.b_shift;		// :1203-1213
	    if ( b_space < 1 || b_space >= (int)g.nbands )
		continue;
	    carrier_band = band;
	    b_mark = (uint32_t)band;
	    if ( first_band < 0 )
		first_band = band;
	    wave_build_table(tw_own, au.d_cs, (uint32_t)band, (uint32_t)b_space, cfg.bit_nsamples,
			     g.fftsize, g.tw_entries);
	    ctx.load_resident_twiddles();
	    ctx.lat_n = 0;				// scored with the old tones
	    ctx.slab_lo = ctx.slab_hi = 0;
	    ctx.un_valid = false;			// (partial sums on the tile: made with the old tones)
	}

	if ( nvalid < cfg.expect_nsamples )			// :1229
	    break;
	ctx.bump(MIFSK_CNT_ITERATIONS);

	const uint32_t ci = carrier ? 1u : 0u;
	const uint32_t try_max = cfg.try_max[ci];
	const uint32_t try_step = cfg.try_step[ci];
	const uint32_t try_first = cfg.try_first[ci];

	ScanResult sr =ctx.scan(base, carrier ? zc1 : zc0, try_first, cfg.search_limit,
				 carrier ? 0u : 1u, carrier);	// minimodem.c:1265-1274
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

	if ( confidence <= cfg.conf_threshold ) {		// minimodem.c:1292-1321
	    if ( ++noconfidence > 20u ) {
		carrier_band = -1;				// :1297
		    if ( carrier ) {

			if ( t0 && o.eps && n_out_eps < o.ecap ) {
			mifsk_episode e;
			e.carrier_nsamples = carrier_nsamples;
			e.first_frame = ep_first;
			e.nframes = nframes_decoded;
			e.confidence_total = confidence_total;
			e.amplitude_total = amplitude_total;
			e.end_reason = 1;
			e.b_mark = ep_b_mark;
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

	carrier_nsamples += cfg.frame_nsamples;			// minimodem.c:1324
	uint32_t flags = 0;
	if ( carrier ) {
	    carrier_nsamples += frame_start;			// minimodem.c:1329-1330
	    carrier_nsamples -= cfg.overscan;
	} else {
	    carrier = true;					// minimodem.c:1350-1353
	    refine = true;
	    flags |= MIFSK_FRAME_ACQUIRE;
	    ep_first = frame_base + n_out_frames;
	    ep_b_mark = b_mark;					// :1340,1344
	}

	if ( refine && confidence < INFINITY && try_step > 1u ) {	// minimodem.c:1357-1389
	    // `carrier` is already set: an acquiring frame is re-searched with
	    // the data string over the no-carrier range (minimodem.c:1378)
	    // (with the carrier held before this frame, the coarse scan above used the
	    // same first try and the same data string: its candidate 0 is reused)
	    ScanResult s2 =ctx.scan(base, ci ? zf1 : zf0, try_first, INFINITY, 0u, true, ci != 0u);
	    flags |= MIFSK_FRAME_REFINED;
	    ctx.bump(MIFSK_CNT_REFINES);
	    if ( s2.conf > confidence ) {
		bits = s2.bits;
		amplitude = s2.ampl;
		frame_start = s2.start;
	    }
	}

	track_amplitude = ( track_amplitude + amplitude ) / 2.0f;	// minimodem.c:1391-1400
	if ( peak_confidence < confidence )
	    peak_confidence = confidence;
	confidence_total += confidence;
	amplitude_total += amplitude;
	nframes_decoded++;
	noconfidence = 0;

	advance = frame_start + cfg.frame_nsamples - cfg.overscan;	// minimodem.c:1407

	bits = data_bits_of(cfg, bits);					// minimodem.c:1415-1428
	const bool suppressed = cfg.do_rx_sync && bits == cfg.sync_byte;	// minimodem.c:1436-1439
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
	if ( ctx.pause )
	    ctx.pause--;
	cyc_general += MIFSK_WCLOCK() - t_gen;
    }


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
	st.carrier_band = carrier_band;
	st.first_band = first_band;
	st.b_mark = b_mark;
	st.ep_b_mark = ep_b_mark;
	st.ep_first = ep_first;
	// (totals: read back rather than carried through the loop)
	const uint32_t b0 = au.append ? 0u : au.d_state[s].nbytes_total, e0 = au.append ? 0u : au.d_state[s].nepisodes_total;
	st.nbytes_total = b0 + n_out_bytes;
	st.nepisodes_total = e0 + n_out_eps;
	st.status = au.d_state[s].status | status;
	au.d_state[s] = st;
    } else if ( stateful && t0 && ( status & MIFSK_STREAM_ABORTED ) ) {
	// aborted in this call: no later call (or chunk of a chained launch) resumes the
	// stream from the state of the call before and re-emits over the same outputs
	mifsk_stream_state st = au.d_state[s];
	st.flags |= MIFSK_STATE_STARTED | MIFSK_STATE_FINISHED;
	st.status |= MIFSK_STREAM_ABORTED;
	au.d_state[s] = st;
    }
    if ( carrier && !paused && resumable ) {			// minimodem.c:1469-1474
	if ( t0 && o.eps && n_out_eps < o.ecap ) {
	    mifsk_episode e;
	    e.carrier_nsamples = carrier_nsamples;
	    e.first_frame = ep_first;
	    e.nframes = nframes_decoded;
	    e.confidence_total = confidence_total;
	    e.amplitude_total = amplitude_total;
	    e.end_reason = 2;
	    e.b_mark = ep_b_mark;
	    o.eps[n_out_eps] = e;
	}
	n_out_eps++;
    }
    // (append: a stream the call before finished keeps the outputs that call wrote)
    if ( t0 && !( ST && au.append != 0u && !resumable && status == 0u ) ) {
	if ( n_out_frames > o.fcap && ( o.bits || o.frames || o.bytes ) )
	    status |= MIFSK_STREAM_FRAMES_TRUNCATED;
	if ( n_out_eps > o.ecap && o.eps )
	    status |= MIFSK_STREAM_EPISODES_TRUNCATED;
	const KernArgs<WaveArgs>::ptr a = KernArgs<WaveArgs>::here();
	if ( a->io.d_nframes ) a->io.d_nframes[s] = n_out_frames;
	if ( a->io.d_nbytes ) a->io.d_nbytes[s] = n_out_bytes;
	if ( a->io.d_nepisodes ) a->io.d_nepisodes[s] = n_out_eps;
	if ( a->io.d_status ) a->io.d_status[s] = status;
	if ( RA && a->io.d_carrier_band && a->g.autodetect ) a->io.d_carrier_band[s] = first_band;
	if ( a->io.d_counters ) {
	    uint64_t *c = a->io.d_counters + (size_t)s * MIFSK_NCOUNTERS;
	    for ( int i = 0; i < MIFSK_NCOUNTERS; i++ )
		c[i] = cnt[i];			// (event counts; the cycle totals below are profile-build only)
	    c[MIFSK_CNT_CYC_TOTAL] = MIFSK_WCLOCK() - t_start;
	    c[MIFSK_CNT_CYC_PARALLEL] = ctx.cyc_scan;
	    c[MIFSK_CNT_CYC_WAIT] = ctx.cyc_block;
	    c[MIFSK_CNT_CYC_CONFIDENCE] = ctx.cyc_conf;
	    c[MIFSK_CNT_CYC_BULK] = cyc_bulk;
	    c[13] = NQ == kTiled ? ctx.cyc_g_pass : ctx.cyc_stage;
	    c[14] = NQ == kTiled ? ctx.cyc_g_asm : ctx.cyc_corr;
	    c[15] = ctx.cyc_g_redo;
	    c[16] = cyc_general;
	    c[17] = ctx.cyc_s_stage;
	    c[18] = ctx.cyc_s_corr;
	    c[19] = ctx.cyc_s_conf;
	    if ( NQ != kTiled ) {		// (profile build: replay and outputs of the bulk path)
		c[20] = cyc_b_replay;
		c[21] = cyc_b_out;
	    }

#ifdef MIFSK_PROFILE
	    // when this stream started and ended on the chip-wide 100 MHz clock
	    c[23] = ( wall_clock64() & 0xFFFFFFFFull ) | ( (uint64_t)t_wall0 << 32 );
	    // HW_REG_XCC_ID (gfx940+)
	    c[22] |= (uint64_t)( __builtin_amdgcn_s_getreg(( 3 << 11 ) | 20) & 15 ) << 32;
#endif
	}
    }
}

// ---------------------------------------------------------------------------
// launcher: occupancy and LDS geometry per configuration
// ---------------------------------------------------------------------------

static constexpr size_t kLdsPerCu = 160 * 1024;

namespace {

struct Plan {
    WaveGeom	g;
    int		sv;		// staging width of the kernel instantiation
    size_t	lds_bytes;
};

// windows of the first F frames of a block
inline uint32_t wins_of( const DevCfg &cfg, uint32_t F )
{
    return cfg.lat_grid ? F * ( cfg.n_bits - 1u ) + 1u : F * cfg.n_bits;
}
inline uint32_t rel_of( const DevCfg &cfg, uint32_t w )
{
    return cfg.lat_grid ? w * cfg.bit_nsamples
			: ( w / cfg.n_bits ) * cfg.lock_advance + cfg.bit_offset[w % cfg.n_bits];
}

// Geometry for a staging width `sv` within `budget` bytes of LDS per wave;
// false when it does not fit.
bool plan_for( const DevCfg &cfg, const WaveHostArgs &ha, int sv, size_t budget, Plan &out,
	       bool want_tile = false )
{
    const uint32_t B = cfg.bit_nsamples, nb = cfg.n_bits;
    WaveGeom g;
    std::memset(&g, 0, sizeof(g));
    const uint32_t round_floats = 64u * (uint32_t)sv * 4u;

    // The bulk path accepts a frame without looking at samples_nvalid: sound
    // when half the reference's buffer (what is always valid away from the end
    // of the stream) covers everything a carrier-held search reads and the
    // largest advance.
    const uint32_t half = ha.samplebuf_size / 2u;
    const bool lattice_sound = !ha.ring_exact
	&& half >= cfg.try_max[1] + cfg.last_reach
	&& half >= cfg.expect_nsamples + cfg.try_max[1]
	&& half > cfg.try_max[1] + cfg.frame_nsamples;
    g.lat_mode = lattice_sound ? LAT_DIRECT : LAT_NONE;
    g.lat_fmax = 64u;
    {
	uint32_t fmin = cfg.lat_grid ? 63u / ( nb - 1u ) : 64u / nb;	// one pass of lanes
	if ( fmin < 2u ) fmin = 2u;
	g.lat_fmin = fmin;
    }
    // LINEAR: window starts non-decreasing in window order and a round's span
    // within one staging pass
    if ( g.lat_mode != LAT_NONE && cfg.lat_linear ) {
	bool ordered = true;
	const uint32_t wtot = wins_of(cfg, 64u);
	for ( uint32_t w = 1; w < wtot; w++ )
	    ordered = ordered && rel_of(cfg, w) >= rel_of(cfg, w - 1);
	uint32_t rw = 0;
	for ( uint32_t cand = 64u; ordered && cand <= 1024u; cand += 64u ) {
	    bool fits = true;
	    for ( uint32_t w0 = 0; w0 < wtot && fits; w0 += cand ) {
		const uint32_t w1 = w0 + cand < wtot ? w0 + cand : wtot;
		fits = rel_of(cfg, w1 - 1) + B - rel_of(cfg, w0) <= round_floats;
	    }
	    if ( !fits )
		break;
	    rw = cand;
	}
	if ( rw ) {
	    g.lat_mode = LAT_LINEAR;
	    g.round_wins = rw;
	}
    }

    // samples one search must see at once (+ slack for the chunked correlator)
    const uint32_t reach = ( cfg.try_max[0] > cfg.try_max[1] ? cfg.try_max[0] : cfg.try_max[1] )
			 + cfg.last_reach + 8u;
    auto skewed_floats = [&]( uint32_t nsamp ) -> size_t {
	return ( (size_t)nsamp + (size_t)( nsamp / B + 2 ) * cfg.skew + 8 + 3 ) & ~(size_t)3;
    };
    uint32_t slab_cap = ( reach + 4u + 3u ) & ~3u;
    size_t scan_floats = skewed_floats(slab_cap);
    const size_t region_floats = g.lat_mode == LAT_LINEAR ? round_floats + 16u : 0u;
    if ( want_tile )
	scan_floats = 0;		// the tile instead of a slab

    for (;;) {
	// a SCAN chunk scores mags_cap / n_bits candidates at once: room for a
	// whole fine scan (<= 2 * 8 candidates) where it fits
	uint32_t mcap = g.lat_mode != LAT_NONE ? wins_of(cfg, g.lat_fmax) : 0u;
	if ( mcap < 16u * nb ) mcap = 16u * nb;
	g.mags_cap = ( mcap + 1u ) & ~1u;
	size_t sf = scan_floats > region_floats ? scan_floats : region_floats;
	// no SCAN slab: long windows go through a tile instead (corr_global_tiled)
	g.tiled = ( scan_floats == 0 && want_tile && !ha.ring_exact && B >= TILE_K ) ? 1u : 0u;
	if ( g.tiled ) {
	    if ( g.lat_mode == LAT_LINEAR )
		g.lat_mode = LAT_DIRECT;	// (that instantiation has no staged rounds)
	    sf = TILE_FLOATS;
	    bool any = false;
	    for ( int i = 0; i < 4; i++ )
		any = any || cfg.seg[i].valid;
	    if ( any )			// shared segments: the carrier-held scans' plan words, the list of
		sf += 2u * ( 2u * SEG_MAX + SEG_MAX / 4u ) + 64u;	// windows to sum again (the partial sums lie on the tile)
	}
	const size_t total = kCntBytes + (size_t)g.mags_cap * sizeof(float2) + sf * 4u + 16u;
	if ( total <= budget ) {
	    g.slab_floats = (uint32_t)sf;
	    g.slab_cap = 0;
	    if ( scan_floats ) {
		// the skewed slab may use the whole region
		size_t ns = sf * B / ( B + cfg.skew );
		ns = ns > 16 ? ns - 16 : 0;
		g.slab_cap = (uint32_t)( ns & ~(size_t)3 );
		if ( g.slab_cap < slab_cap )
		    g.slab_cap = slab_cap;
	    }
	    // DIRECT blocks of short windows (SAME) stream every lane's window from global memory,
	    // and the frames behind a refinement are read AGAIN by the block after it -- on a signal
	    // that is refined every fifth frame (SAME's 8-bit frames without start / stop bits) a
	    // block of a full pass of lanes (8 frames) throws three of them away: 2.3 x the
	    // algorithmic bytes moved, at 4.6 TB/s of fabric traffic.  A pass of lanes costs the
	    // same half full, so after a break the next block is as long as the lattice held last
	    // time, down to four frames (same-box: 7.46 -> 7.07 ms; 3, 5, 6: 7.23, 7.27, 7.21).
	    if ( g.lat_mode == LAT_DIRECT && !g.tiled && g.lat_fmin > 4u )
		g.lat_fmin = 4u;
	    if ( const char *e = experiment_env("MIFSK_LAT_FMIN") )	// experiments only
		if ( std::atoi(e) >= 1 )
		    g.lat_fmin = (uint32_t)std::atoi(e);
	    out.g = g;
	    out.sv = sv;
	    out.lds_bytes = total;
	    return true;
	}
	if ( g.lat_mode != LAT_NONE && g.lat_fmax / 2u >= g.lat_fmin && g.lat_fmax > 8u ) {
	    g.lat_fmax /= 2u;			// shorter blocks: fewer magnitude slots
	    continue;
	}
	if ( scan_floats > region_floats ) {
	    scan_floats = 0;			// SCAN streams its windows from global memory
	    continue;
	}
	return false;
    }
}

} // namespace

int launch_demod_wave( const DevCfg &cfg, const DevCfg *d_cfg, const double *d_tw,
	const mifsk_demod_io &io, const WaveHostArgs &ha, void *stream, LaunchInfo *plan_only )
{
    if ( io.nstreams <= 0 && !plan_only )
	return 0;
    const int ncu = ha.ncu > 0 ? ha.ncu : 256;
    // Waves per CU the batch can use (a wave is a workgroup): at least one per
    // SIMD, at most 16 (4 per SIMD at <= 128 VGPRs).  Each gets that share of
    // the CU's LDS; prefer the widest staging that fits, then fewer waves.
    uint32_t want = ( (uint32_t)io.nstreams + (uint32_t)ncu - 1u ) / (uint32_t)ncu;
    if ( want < 4u ) want = 4u;
    if ( want > 16u ) want = 16u;
    int force_sv = 0;
    if ( const char *e = experiment_env("MIFSK_WAVES_PER_CU") )	// experiments only
	want = (uint32_t)std::atoi(e) < 1u ? 1u : (uint32_t)std::atoi(e);
    if ( const char *e = experiment_env("MIFSK_SV") )
	force_sv = std::atoi(e);
    Plan plan;
    bool ok = false;
    // Long windows are better read through the tile at two waves per SIMD than
    // from a slab that leaves one wave per SIMD (tools/ubench/longwin.hip: 17 ms
    // against 45): a slab only while it fits 8 waves per CU then
    const bool tile_ok = !ha.ring_exact && cfg.bit_nsamples >= kTileMinBit && force_sv != 4;
    const uint32_t wmin = tile_ok ? 8u : 4u;
    for ( uint32_t wpc = want; wpc >= wmin && !ok; wpc -= ( wpc > 8u ? 4u : ( wpc > 4u ? 2u : 1u ) ) ) {
	const size_t budget = ( kLdsPerCu / wpc ) & ~(size_t)255;
	// The wide-staging instantiation is compiled for two waves per SIMD (256
	// VGPRs): worth it where rounds are staged through LDS (linear LATTICE) or
	// where no more than 8 waves per CU are wanted anyway
	const bool wide = force_sv ? force_sv == 10 : ( wpc <= 8u || cfg.lat_linear );
	ok = wide && plan_for(cfg, ha, 10, budget, plan) && plan.g.slab_cap != 0u
		  && ( plan.g.lat_mode == LAT_LINEAR || wpc <= 8u || force_sv == 10 );
	if ( !ok )
	    ok = plan_for(cfg, ha, 4, budget, plan) && plan.g.slab_cap != 0u;
	if ( wpc == 4u )
	    break;
    }
    if ( !ok && force_sv != 4 ) {
	// nothing keeps the SCAN slab in LDS (RTTY: 1056-sample windows, a 40 kB
	// span; 0.5 baud: 96000-sample windows): the windows come from global
	// memory through the tile (with the shared segments' plan words 12.9 kB), two waves per
	// SIMD: compiled for three (168 VGPRs) the instantiation spilled 44 VGPRs to scratch and
	// ran 4096 RTTY streams in 10.8 ms at 8 waves per CU; with 191 VGPRs and nothing spilled
	// the same 8 waves per CU take 9.3 ms (profiles/r03_history.md)
	for ( uint32_t wpc = want < 8u ? want : 8u; wpc >= 4u && !ok; wpc-- ) {
	    const size_t budget = ( kLdsPerCu / wpc ) & ~(size_t)255;
	    ok = plan_for(cfg, ha, 10, budget, plan, true) && plan.g.tiled;
	}
    }
    if ( !ok ) {
	// ... or straight into registers, a window per lane
	const size_t budget = ( kLdsPerCu / want ) & ~(size_t)255;
	ok = plan_for(cfg, ha, 4, budget, plan);
	if ( !ok )
	    return -12;
    }
    // Chained launches (WaveChain, mifsk_device.h): where the plain instantiation is one of the
    // resumable ones, the batch is more than the chip holds at once and the streams are long
    // enough to cut (a chunk's last samplebuf waits for the next chunk: at least 8 per chunk).
    uint32_t chain_g = 0, chain_k = 0;
    {
	const uint32_t nq0 = ( plan.g.lat_mode == LAT_LINEAR && cfg.bit_nsamples % 4u == 0u ) ? cfg.bit_nsamples / 4u : 0u;
	const bool st_kernel = plan.g.tiled || !( ( plan.sv == 10 && ( nq0 == 10u || nq0 == 5u ) ) || ( plan.sv == 4 && nq0 == 1u ) );
	const uint32_t by_lds = (uint32_t)( kLdsPerCu / ( plan.lds_bytes ? plan.lds_bytes : 1 ) );
	const uint32_t by_regs = ( plan.g.tiled || plan.sv == 10 ? 2u : 4u ) * 4u;
	const uint64_t slots = (uint64_t)( by_lds < by_regs ? by_lds : by_regs ) * (uint64_t)ncu;
	const bool allowed = ( plan_only ? ha.chain_ok : ha.chain != nullptr ) && st_kernel && !ha.d_state
			  && !ha.ring_exact && !io.d_counters && io.nstreams > 0;
	if ( allowed && (uint64_t)io.nstreams > slots && ha.samplebuf_size > 0u ) {
	    chain_g = 2u;
	    chain_k = io.nsamples / ( 8u * ha.samplebuf_size );
	    if ( chain_k > 8u ) chain_k = 8u;
	}
	if ( const char *e = experiment_env("MIFSK_CHAIN") ) {	// experiments and tests only: "G,K", any batch
	    int a = 0, b = 0;
	    if ( allowed && std::sscanf(e, "%d,%d", &a, &b) == 2 ) {
		chain_g = (uint32_t)( a < 0 ? 0 : a );
		chain_k = (uint32_t)( b < 0 ? 0 : b );
	    }
	}
	if ( chain_g > (uint32_t)WaveChain::kMaxGroups ) chain_g = (uint32_t)WaveChain::kMaxGroups;
	if ( chain_g > (uint32_t)io.nstreams ) chain_g = (uint32_t)io.nstreams;
	if ( chain_g < 1u || chain_k < 2u )
	    chain_g = chain_k = 0u;
    }
    // Whole rounds.  A batch of more streams than waves fit runs in rounds, and a last round
    // that is a fraction of one leaves the chip mostly idle while its chains finish (4096 RTTY
    // streams at 12 waves per CU are 1.33 rounds: measured 13.8 ms against 13.3 at 8-10).  Among
    // the occupancies this plan allows (down to two thirds of the most) take the one that wastes
    // the fewest wave slots over the whole batch, the higher one on a tie; the kernel is limited
    // to it by its LDS allocation.  (Not for chained launches: their slots are refilled as they
    // come free.)
    {
	const uint32_t by_lds = (uint32_t)( kLdsPerCu / ( plan.lds_bytes ? plan.lds_bytes : 1 ) );
	const uint32_t by_regs = ( plan.g.tiled || plan.sv == 10 ? 2u : 4u ) * 4u;
	const uint32_t most = by_lds < by_regs ? by_lds : by_regs;
	const uint32_t per_cu = ( (uint32_t)( io.nstreams > 0 ? io.nstreams : 0 ) + (uint32_t)ncu - 1u ) / (uint32_t)ncu;
	if ( most >= 3u && per_cu > most && !chain_g ) {
	    uint32_t best = most, best_waste = 0xFFFFFFFFu;
	    for ( uint32_t w = most; 3u * w >= 2u * most; w-- ) {
		const uint32_t waste = ( per_cu + w - 1u ) / w * w - per_cu;
		if ( waste < best_waste ) {
		    best_waste = waste;
		    best = w;
		}
	    }
	    if ( best < most ) {
		const size_t pad = ( kLdsPerCu / best ) & ~(size_t)255;	// exactly `best` of these fit a CU
		if ( pad > plan.lds_bytes && kLdsPerCu / pad == best )
		    plan.lds_bytes = pad;
	    }
	}
    }
    if ( const char *e = experiment_env("MIFSK_LDS_PAD") )	// experiments only: limit occupancy
	if ( (size_t)std::atoi(e) > plan.lds_bytes )
	    plan.lds_bytes = (size_t)std::atoi(e);
    WaveGeom &g = plan.g;
    g.bufsize = ha.samplebuf_size;
    g.ring_exact = ha.ring_exact ? 1u : 0u;
    g.ring_stride = ha.ring_stride;
    g.autodetect = ha.autodetect ? 1u : 0u;
    g.auto_threshold = ha.auto_threshold;
    g.nps = ha.nps;
    g.b_shift = ha.b_shift;
    g.fftsize = ha.fftsize;
    g.nbands = ha.nbands;
    g.tw_entries = ha.tw_entries;
    WaveAuto au;
    au.d_cs = ha.d_cs;
    au.d_tw_scratch = ha.d_tw_scratch;
    au.d_ring = ha.d_ring;
    au.d_state = ha.d_state;
    au.d_origin = ha.d_origin;
    au.final = ha.final ? 1u : 0u;
    au.limit = 0u;
    au.append = 0u;
    for ( int k = 0; k < 5; k++ ) {
	au.d_rot[k] = ha.d_rot[k];
	au.rot_stride[k] = ha.rot_stride[k];
    }

    // the instantiation: staging width x resident-table correlator for the bit
    // lengths that have one (linear LATTICE only)
    const uint32_t nq = ( g.lat_mode == LAT_LINEAR && cfg.bit_nsamples % 4u == 0u ) ? cfg.bit_nsamples / 4u : 0u;
    if ( plan_only ) {
	const bool lin = g.lat_mode == LAT_LINEAR;	// (generic instantiations: <., 0> linear lattice, <., -2> direct or none)
	plan_only->kernel = chain_g ? ( g.tiled ? "mifsk::demod_wave_kernel<10, -1, true>"
					 : plan.sv == 10 ? ( lin ? "mifsk::demod_wave_kernel<10, 0, true>" : "mifsk::demod_wave_kernel<10, -2, true>" )
							 : ( lin ? "mifsk::demod_wave_kernel<4, 0, true>" : "mifsk::demod_wave_kernel<4, -2, true>" ) )
			  : g.tiled ? "mifsk::demod_wave_kernel<10, -1>"
			  : plan.sv == 10 ? ( nq == 10u ? "mifsk::demod_wave_kernel<10, 10>"
					   : nq == 5u ? "mifsk::demod_wave_kernel<10, 5>"
					   : lin ? "mifsk::demod_wave_kernel<10, 0>" : "mifsk::demod_wave_kernel<10, -2>" )
					  : ( nq == 1u ? "mifsk::demod_wave_kernel<4, 1>"
					      : lin ? "mifsk::demod_wave_kernel<4, 0>" : "mifsk::demod_wave_kernel<4, -2>" );
	plan_only->workgroup_size = 64;
	plan_only->lds_bytes = (uint32_t)plan.lds_bytes;
	plan_only->lattice_mode = g.lat_mode;
	plan_only->frames_per_block = g.lat_mode != LAT_NONE ? g.lat_fmax : 0u;
	plan_only->waves_per_simd = g.tiled || plan.sv == 10 ? 2u : 4u;
	plan_only->chain_groups = chain_g;
	plan_only->chain_chunks = chain_k;
	return 0;
    }
    hipStream_t st = (hipStream_t)stream;
    if ( chain_g ) {
	const WaveChain &ch = *ha.chain;
	if ( (size_t)io.nstreams > ch.state_cap )
	    return -12;
	const bool lin = g.lat_mode == LAT_LINEAR;
	// which resumable instantiation: staging width x lattice kind x (--auto-carrier or not)
#define MIFSK_CHAIN_PICK(RA_)											\
	( g.tiled ? reinterpret_cast<const void *>(&demod_wave_kernel<10, kTiled, true, RA_>)			\
	  : plan.sv == 10 ? ( lin ? reinterpret_cast<const void *>(&demod_wave_kernel<10, 0, true, RA_>)		\
				  : reinterpret_cast<const void *>(&demod_wave_kernel<10, kDirect, true, RA_>) )	\
			  : ( lin ? reinterpret_cast<const void *>(&demod_wave_kernel<4, 0, true, RA_>)		\
				  : reinterpret_cast<const void *>(&demod_wave_kernel<4, kDirect, true, RA_>) ) )
	const void *fn = ha.autodetect ? MIFSK_CHAIN_PICK(true) : MIFSK_CHAIN_PICK(false);
#undef MIFSK_CHAIN_PICK
	if ( hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds_bytes) != hipSuccess )
	    return -5;
	hipEvent_t fork = (hipEvent_t)ch.ev_fork;
	if ( hipEventRecord(fork, st) != hipSuccess )
	    return -5;
	mifsk_demod_io gio[WaveChain::kMaxGroups];
	uint32_t glo[WaveChain::kMaxGroups];
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
	    if ( io.d_carrier_band ) o.d_carrier_band = io.d_carrier_band + lo;
	    if ( o.nstreams > 0
		    && hipMemsetAsync(ch.d_state + lo, 0, (size_t)o.nstreams * sizeof(mifsk_stream_state), gs) != hipSuccess )
		return -5;
	}
	const uint32_t chunk = ( io.nsamples + chain_k - 1u ) / chain_k;
	au.append = 1u;
	au.d_origin = nullptr;
	for ( uint32_t k = 0; k < chain_k; k++ ) {
	    const bool last = k + 1u == chain_k;
	    au.final = last ? 1u : 0u;
	    au.limit = last ? 0u : ( k + 1u ) * chunk;
	    for ( uint32_t gi = 0; gi < chain_g; gi++ ) {
		if ( gio[gi].nstreams <= 0 )
		    continue;
		hipStream_t gs = (hipStream_t)ch.streams[gi];
		au.d_state = ch.d_state + glo[gi];
		if ( ha.d_tw_scratch )		// (--auto-carrier: the group's streams' own tables)
		    au.d_tw_scratch = ha.d_tw_scratch + (size_t)glo[gi] * g.tw_entries * 4u;
		void *kargs[] = { (void *)&d_cfg, (void *)&d_tw, (void *)&gio[gi], (void *)&g, (void *)&au };
		(void)hipLaunchKernel(fn, dim3((unsigned)gio[gi].nstreams), dim3(64), kargs, plan.lds_bytes, gs);
	    }
	}
	const bool launched = hipGetLastError() == hipSuccess;
	for ( uint32_t gi = 0; gi < chain_g; gi++ ) {
	    (void)hipEventRecord((hipEvent_t)ch.ev_done[gi], (hipStream_t)ch.streams[gi]);
	    (void)hipStreamWaitEvent(st, (hipEvent_t)ch.ev_done[gi], 0);
	}
	return launched ? 0 : -5;
    }
#define MIFSK_WAVE_LAUNCH_ST(SV_, NQ_)										\
    do {													\
	const void *fn = reinterpret_cast<const void *>(&demod_wave_kernel<SV_, NQ_, true>);			\
	if ( hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds_bytes)		\
		!= hipSuccess )											\
	    return -5;												\
	hipLaunchKernelGGL((demod_wave_kernel<SV_, NQ_, true>), dim3((unsigned)io.nstreams), dim3(64),		\
			   plan.lds_bytes, st, d_cfg, d_tw, io, g, au);						\
    } while (0)
    if ( ha.d_state ) {
	// mifsk_demod_slab: the instantiations with the state code, generic correlators
	const bool lin = g.lat_mode == LAT_LINEAR;
	if ( g.tiled )                 MIFSK_WAVE_LAUNCH_ST(10, kTiled);
	else if ( plan.sv == 10 && lin ) MIFSK_WAVE_LAUNCH_ST(10, 0);
	else if ( plan.sv == 10 )      MIFSK_WAVE_LAUNCH_ST(10, kDirect);
	else if ( lin )                MIFSK_WAVE_LAUNCH_ST(4, 0);
	else                           MIFSK_WAVE_LAUNCH_ST(4, kDirect);
	return hipGetLastError() == hipSuccess ? 0 : -5;
    }
#undef MIFSK_WAVE_LAUNCH_ST
    // (RING addressing and --auto-carrier have their own instantiations: the plain ones carry
    // neither that code nor the registers it keeps alive)
#define MIFSK_WAVE_LAUNCH_RA(SV_, NQ_, RA_)									\
    do {													\
	const void *fn = reinterpret_cast<const void *>(&demod_wave_kernel<SV_, NQ_, false, RA_>);		\
	if ( hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds_bytes)		\
		!= hipSuccess )											\
	    return -5;												\
	hipLaunchKernelGGL((demod_wave_kernel<SV_, NQ_, false, RA_>), dim3((unsigned)io.nstreams), dim3(64),	\
				   plan.lds_bytes, st, d_cfg, d_tw, io, g, au);						\
    } while (0)
#define MIFSK_WAVE_LAUNCH(SV_, NQ_)										\
    do {													\
	if ( ha.ring_exact || ha.autodetect )									\
	    MIFSK_WAVE_LAUNCH_RA(SV_, NQ_, true);								\
	else													\
	    MIFSK_WAVE_LAUNCH_RA(SV_, NQ_, false);								\
    } while (0)
    if ( g.tiled ) {
	MIFSK_WAVE_LAUNCH(10, kTiled);				// RTTY and slower
    } else if ( plan.sv == 10 ) {
	if ( nq == 10u )     MIFSK_WAVE_LAUNCH(10, 10);		// 1200 baud at 48 kHz
	else if ( nq == 5u ) MIFSK_WAVE_LAUNCH(10, 5);		// 2400 baud; 1200 baud at 24 kHz
	else if ( g.lat_mode == LAT_LINEAR ) MIFSK_WAVE_LAUNCH(10, 0);
	else                 MIFSK_WAVE_LAUNCH(10, kDirect);
    } else {
	if ( nq == 1u )      MIFSK_WAVE_LAUNCH(4, 1);		// 12000 baud
	else if ( g.lat_mode == LAT_LINEAR ) MIFSK_WAVE_LAUNCH(4, 0);
	else                 MIFSK_WAVE_LAUNCH(4, kDirect);	// SAME
    }
#undef MIFSK_WAVE_LAUNCH
#undef MIFSK_WAVE_LAUNCH_RA
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

} // namespace mifsk
