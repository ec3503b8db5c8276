// mifsk_devlib.h -- device-side building blocks shared by the receive-loop
// kernels (mifsk_kernels.hip: one workgroup per stream, master + worker waves;
// mifsk_wave.hip: one wavefront per stream).  Everything here is arithmetic or
// a memory idiom with a fixed operation order: both kernels produce the
// oracle's results bit for bit because they share these sequences.
// gfx950 only; included by .hip files only.
#pragma once

#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>
#include <cstdint>

#include "mifsk_device.h"
#include "mifsk_devmath.h"

namespace mifsk {

constexpr int P_CAP = 64;	// candidate positions per batch (= one wave of lanes)
constexpr int W_CAP = 448;	// bit windows per batch (LDS scratch)

// ---------------------------------------------------------------------------
// arithmetic shared by every kernel
// ---------------------------------------------------------------------------

// A kernel's by-value arguments as they lie in the kernarg segment, re-read where
// they are used.  The receive-loop kernels take ONE struct; what the cold ends of
// the loop need from it (output pointers, capacities, the final counts) is loaded
// there through this pointer -- scalar loads that hit the scalar cache -- instead
// of sitting in (or being spilled from) scalar registers for the whole loop: the
// compiler hoists loads of plain kernel arguments to the top of the kernel, the
// empty asm makes each use site's loads its own.
template <typename Args>
struct KernArgs {
    typedef __attribute__((address_space(4))) const Args *ptr;
    static __device__ __forceinline__ ptr here()
    {
	ptr p = (ptr)__builtin_amdgcn_kernarg_segment_ptr();
	asm volatile("" : "+s"(p));
	return p;
    }
};

struct FrameOut {
    float	conf;
    float	ampl;
    uint64_t	bits;
};

// fsk_frame_analyze after the per-bit magnitudes are known (fsk.c:199-212,
// 271-342, 439-441).  `mags[k]` = (mark, space) magnitude of bit k.  One lane
// runs this for one candidate position; every operation is f32, in the
// reference's order (contraction is disabled for this file).  The magnitudes
// are fetched from LDS eight at a time so that the loads (and, in the second
// pass, the independent divisions) overlap; the running sums stay sequential.
constexpr int CCH = 5;

__device__ __forceinline__ FrameOut
frame_confidence( const float2 *mags, uint64_t req_mask, uint64_t req_val, uint32_t n_bits )
{
    FrameOut out;
    out.conf = 0.0f;
    out.ampl = 0.0f;
    out.bits = 0;

    uint64_t bits = 0;
    float total_sig = 0.0f, total_noise = 0.0f;
    float mark_sig = 0.0f, space_sig = 0.0f;
    uint32_t n_mark = 0;
    const uint32_t last = n_bits - 1u;
    for ( uint32_t k0 = 0; k0 < n_bits; k0 += CCH ) {
	float2 m[CCH];
#pragma unroll
	for ( int j = 0; j < CCH; j++ )
	    m[j] = mags[k0 + j < last ? k0 + j : last];
#pragma unroll
	for ( int j = 0; j < CCH; j++ ) {
	    if ( k0 + j < n_bits ) {
		const bool one = m[j].x > m[j].y;		// fsk.c:161 (strict)
		const float sig = one ? m[j].x : m[j].y;
		const float noise = one ? m[j].y : m[j].x;
		bits |= (uint64_t)( one ? 1u : 0u ) << ( k0 + j );
		total_sig += sig;				// fsk.c:278
		if ( noise > FLT_EPSILON )			// fsk.c:279
		    total_noise += noise;
		if ( one ) {
		    mark_sig += sig;
		    n_mark++;
		} else {
		    space_sig += sig;
		}
	    }
	}
    }
    // a required bit that came out wrong rejects the frame with confidence 0
    // and bits/ampl untouched (fsk.c:211-212,486-487)
    if ( ( bits ^ req_val ) & req_mask )
	return out;
    const uint32_t n_space = n_bits - n_mark;

    const float snr = total_sig / total_noise;		// fsk.c:292
    const float avg_sig = total_sig / (float)(int)n_bits;	// fsk.c:295 (int n_bits)
    if ( n_mark )
	mark_sig /= (float)n_mark;			// fsk.c:298-301
    if ( n_space )
	space_sig /= (float)n_space;

    float divergence = 0.0f;				// fsk.c:305-313
    for ( uint32_t k0 = 0; k0 < n_bits; k0 += CCH ) {
	float term[CCH];
#pragma unroll
	for ( int j = 0; j < CCH; j++ ) {
	    const float2 m = mags[k0 + j < last ? k0 + j : last];
	    const bool one = m.x > m.y;
	    const float sig = one ? m.x : m.y;
	    const float cls = one ? mark_sig : space_sig;
	    term[j] = fabsf(sig - cls) / cls;
	}
#pragma unroll
	for ( int j = 0; j < CCH; j++ )
	    if ( k0 + j < n_bits )
		divergence += term[j];
    }
    divergence *= 2.0f;
    divergence /= (float)(int)n_bits;

    out.conf = snr * (1.0f - divergence);		// fsk.c:336
    out.ampl = avg_sig;					// fsk.c:342
    out.bits = bits;					// fsk.c:439-441
    return out;
}

// 1 / c for a finite, non-zero float c, in double, to within 2^-52 relative: v_rcp_f64 and
// two Newton steps.
__device__ __forceinline__ double rcp_of_float( float c )
{
    const double d = (double)c;
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(e, r, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(e, r, r);
    return r;
}

// IEEE float division x / c through a reciprocal in double: (float)((double)x * rc) with
// rc = rcp_of_float(c).  EXACT -- the correctly rounded quotient -- whenever the result is a
// normal float (or zero, infinite or NaN): the exact quotient of two 24-bit significands lies
// at least 2^-48 (relative) away from every midpoint of two adjacent floats (X / C - M / 2^24
// = (X 2^24 - M C) / (C 2^24), a non-zero integer over less than 2^48), and x * rc is within
// 2^-51 of it, so rounding the double to float cannot cross a midpoint.  (Ties exist only for
// denormal results: callers fall back to the division there.)  Three instructions per quotient
// where the division takes ten -- and the reference's divergence pass divides the eleven bits
// of a frame by one of only two class means (fsk.c:305-313).
__device__ __forceinline__ float div_by_rcp( float x, double rc )
{
    return (float)( (double)x * rc );
}

__device__ __forceinline__ bool float_is_plain( float c )	// finite and not zero
{
    return c != 0.0f && fabsf(c) < INFINITY;			// (false for NaN)
}

__device__ __forceinline__ bool float_is_subnormal( float t )
{
    return t != 0.0f && fabsf(t) < FLT_MIN;
}

// The same for a frame length known at compile time: every magnitude is loaded
// once (all loads in flight together), the per-bit signal levels stay in
// registers for the divergence pass, nothing is computed for padding slots.
// Operation for operation the sequence above -- with the divisions of the divergence pass
// and the two by the constant frame length made through reciprocals (div_by_rcp: the same
// quotients, bit for bit; any lane whose divisor or result is not a plain float sends its
// wave through the divisions proper).
template <int NB>
__device__ __forceinline__ FrameOut
frame_confidence_fixed( const float2 *mags, uint64_t req_mask, uint64_t req_val, uint32_t &fell_back )
{
    FrameOut out;
    out.conf = 0.0f;
    out.ampl = 0.0f;
    out.bits = 0;

    float2 m[NB];
#pragma unroll
    for ( int k = 0; k < NB; k++ )
	m[k] = mags[k];
    uint32_t bits = 0;					// NB <= 32 here
    float sig[NB];
    float total_sig = 0.0f, total_noise = 0.0f;
    float mark_sig = 0.0f, space_sig = 0.0f;
    uint32_t n_mark = 0;
#pragma unroll
    for ( int k = 0; k < NB; k++ ) {
	const bool one = m[k].x > m[k].y;		// fsk.c:161 (strict)
	sig[k] = one ? m[k].x : m[k].y;
	const float noise = one ? m[k].y : m[k].x;
	bits |= ( one ? 1u : 0u ) << k;
	total_sig += sig[k];				// fsk.c:278
	if ( noise > FLT_EPSILON )			// fsk.c:279
	    total_noise += noise;
	if ( one ) {
	    mark_sig += sig[k];
	    n_mark++;
	} else {
	    space_sig += sig[k];
	}
    }
    if ( ( (uint64_t)bits ^ req_val ) & req_mask )	// fsk.c:211-212,486-487
	return out;
    const uint32_t n_space = (uint32_t)NB - n_mark;

    const float snr = total_sig / total_noise;		// fsk.c:292
    constexpr double kRcpNB = 1.0 / (double)NB;
    float avg_sig = div_by_rcp(total_sig, kRcpNB);	// fsk.c:295
    if ( n_mark )
	mark_sig /= (float)n_mark;			// fsk.c:298-301
    if ( n_space )
	space_sig /= (float)n_space;

    float term[NB];					// fsk.c:305-313
    const double rc_mark = rcp_of_float(mark_sig), rc_space = rcp_of_float(space_sig);
    bool odd = ( n_mark && !float_is_plain(mark_sig) ) || ( n_space && !float_is_plain(space_sig) )
	    || float_is_subnormal(avg_sig);
#pragma unroll
    for ( int k = 0; k < NB; k++ ) {
	const bool one = ( bits >> k ) & 1u;
	const float cls = one ? mark_sig : space_sig;
	term[k] = div_by_rcp(fabsf(sig[k] - cls), one ? rc_mark : rc_space);
	odd = odd || float_is_subnormal(term[k]);
    }
    float divergence = 0.0f;
#pragma unroll
    for ( int k = 0; k < NB; k++ )
	divergence += term[k];
    divergence *= 2.0f;
    float div_n = div_by_rcp(divergence, kRcpNB);
    odd = odd || float_is_subnormal(div_n);
    if ( __any(odd) ) {
	// a divisor that is zero, infinite or NaN, or a subnormal quotient, somewhere in the
	// wave: the divisions proper, for everybody (uniform branch, practically never taken)
	fell_back = 1u;			// (MIFSK_CNT_CONF_FALLBACKS: tests/test_gpu_parity.py drives this arm)
	avg_sig = total_sig / (float)NB;
#pragma unroll
	for ( int k = 0; k < NB; k++ ) {
	    const float cls = ( bits >> k ) & 1u ? mark_sig : space_sig;
	    term[k] = fabsf(sig[k] - cls) / cls;
	}
	divergence = 0.0f;
#pragma unroll
	for ( int k = 0; k < NB; k++ )
	    divergence += term[k];
	divergence *= 2.0f;
	div_n = divergence / (float)NB;
    }
    divergence = div_n;

    out.conf = snr * (1.0f - divergence);		// fsk.c:336
    out.ampl = avg_sig;					// fsk.c:342
    out.bits = bits;					// fsk.c:439-441
    return out;
}

// The same pass written for a wave that runs ALONE and whose time is latency, not throughput:
// the master of the workgroup engine (mifsk_kernels.hip), whose scoring of a batch is a link of
// the stream's serial chain.  A wave issues in order; the sequence above, as hipcc schedules it,
// is one dependent chain after another -- a conditional running sum is an add and a select on
// the chain, 22 dependent instructions for 11 bits, four such sums one behind the other, then
// three divisions each inside its own branch, then eleven convert / multiply / convert chains
// -- and a dependent instruction issues every ~8 cycles where independent ones issue every 2-4.
// Here the same operations on the same values are laid out in stages whose instructions are
// independent of one another, pinned with scheduling barriers:
//   * the per-bit selections first (nothing depends on another bit);
//   * the four running sums as four chains in lock-step, one ADD per bit on each chain: a
//     conditional sum `if (c) s += v` is `s += c ? v : 0.0f` -- identical for every value that
//     can occur (the sums start at +0.0 and every addend is a magnitude, +0.0 or positive or
//     NaN, so no sum is ever -0.0, and x + 0.0f == x for every other x, NaN included);
//   * the three quotients without branches (`if (n) s /= n` as `s = n ? s / n : s`: x / 0 is
//     computed and dropped), so that their sequences interleave;
//   * the divergence terms stage by stage across the bits (differences, converts, products,
//     converts back), then the one chain that cannot be avoided: their sum in bit order.
// Bit for bit the results of frame_confidence_fixed (every parity test runs through it).
template <int NB>
__device__ __forceinline__ FrameOut
frame_confidence_staged( const float2 *mags, uint64_t req_mask, uint64_t req_val, uint32_t &fell_back )
{
    FrameOut out;
    out.conf = 0.0f;
    out.ampl = 0.0f;
    out.bits = 0;

    float2 m[NB];
#pragma unroll
    for ( int k = 0; k < NB; k++ )
	m[k] = mags[k];
    uint32_t bits = 0;					// NB <= 32 here
    float sig[NB], nz[NB], mk[NB], sp[NB];
#pragma unroll
    for ( int k = 0; k < NB; k++ ) {
	const bool one = m[k].x > m[k].y;		// fsk.c:161 (strict)
	sig[k] = one ? m[k].x : m[k].y;
	const float noise = one ? m[k].y : m[k].x;
	bits |= ( one ? 1u : 0u ) << k;
	nz[k] = noise > FLT_EPSILON ? noise : 0.0f;	// fsk.c:279
	mk[k] = one ? sig[k] : 0.0f;
	sp[k] = one ? 0.0f : sig[k];
    }
    if ( ( (uint64_t)bits ^ req_val ) & req_mask )	// fsk.c:211-212,486-487
	return out;
    __builtin_amdgcn_sched_barrier(0);
    float total_sig = 0.0f, total_noise = 0.0f;
    float mark_sig = 0.0f, space_sig = 0.0f;
#pragma unroll
    for ( int k = 0; k < NB; k++ ) {
	total_sig += sig[k];				// fsk.c:278
	total_noise += nz[k];
	mark_sig += mk[k];
	space_sig += sp[k];
	__builtin_amdgcn_sched_barrier(0);
    }
    const uint32_t n_mark = (uint32_t)__popc(bits);
    const uint32_t n_space = (uint32_t)NB - n_mark;

    const float snr = total_sig / total_noise;		// fsk.c:292
    constexpr double kRcpNB = 1.0 / (double)NB;
    float avg_sig = div_by_rcp(total_sig, kRcpNB);	// fsk.c:295
    const float mark_q = mark_sig / (float)n_mark;	// fsk.c:298-301
    const float space_q = space_sig / (float)n_space;
    mark_sig = n_mark ? mark_q : mark_sig;
    space_sig = n_space ? space_q : space_sig;

    const double rc_mark = rcp_of_float(mark_sig), rc_space = rcp_of_float(space_sig);
    bool odd = ( n_mark && !float_is_plain(mark_sig) ) || ( n_space && !float_is_plain(space_sig) )
	    || float_is_subnormal(avg_sig);
    __builtin_amdgcn_sched_barrier(0);
    float dif[NB];					// fsk.c:305-313
    double rcs[NB];
#pragma unroll
    for ( int k = 0; k < NB; k++ ) {
	const bool one = ( bits >> k ) & 1u;
	dif[k] = fabsf(sig[k] - ( one ? mark_sig : space_sig ));
	rcs[k] = one ? rc_mark : rc_space;
    }
    __builtin_amdgcn_sched_barrier(0);
    double prod[NB];
#pragma unroll
    for ( int k = 0; k < NB; k++ )
	prod[k] = (double)dif[k] * rcs[k];
    __builtin_amdgcn_sched_barrier(0);
    float term[NB];
#pragma unroll
    for ( int k = 0; k < NB; k++ ) {
	term[k] = (float)prod[k];			// div_by_rcp(dif[k], rcs[k])
	odd = odd || float_is_subnormal(term[k]);
    }
    __builtin_amdgcn_sched_barrier(0);
    float divergence = 0.0f;
#pragma unroll
    for ( int k = 0; k < NB; k++ )
	divergence += term[k];
    divergence *= 2.0f;
    float div_n = div_by_rcp(divergence, kRcpNB);
    odd = odd || float_is_subnormal(div_n);
    if ( __any(odd) ) {
	// (as in frame_confidence_fixed: the divisions proper, for everybody, practically never)
	fell_back = 1u;
	avg_sig = total_sig / (float)NB;
#pragma unroll
	for ( int k = 0; k < NB; k++ ) {
	    const float cls = ( bits >> k ) & 1u ? mark_sig : space_sig;
	    term[k] = fabsf(sig[k] - cls) / cls;
	}
	divergence = 0.0f;
#pragma unroll
	for ( int k = 0; k < NB; k++ )
	    divergence += term[k];
	divergence *= 2.0f;
	div_n = divergence / (float)NB;
    }
    divergence = div_n;

    out.conf = snr * (1.0f - divergence);		// fsk.c:336
    out.ampl = avg_sig;					// fsk.c:342
    out.bits = bits;					// fsk.c:439-441
    return out;
}

// frame lengths with a specialised confidence pass: start + 8 data + stop with
// the previous stop bit (11), the 7-bit variant (10); everything else is generic
__device__ __forceinline__ FrameOut
frame_confidence_any( const float2 *mags, uint64_t req_mask, uint64_t req_val, uint32_t n_bits,
	uint32_t &fell_back )
{
    if ( n_bits == 11u )
	return frame_confidence_fixed<11>(mags, req_mask, req_val, fell_back);
    if ( n_bits == 10u )
	return frame_confidence_fixed<10>(mags, req_mask, req_val, fell_back);
    if ( n_bits == 8u )					// RTTY "10ddddd1", SAME "dddddddd"
	return frame_confidence_fixed<8>(mags, req_mask, req_val, fell_back);
    return frame_confidence(mags, req_mask, req_val, n_bits);
}

// ... for the lone, latency-bound wave (the workgroup engine's master)
__device__ __forceinline__ FrameOut
frame_confidence_any_staged( const float2 *mags, uint64_t req_mask, uint64_t req_val, uint32_t n_bits,
	uint32_t &fell_back )
{
    if ( n_bits == 11u )
	return frame_confidence_staged<11>(mags, req_mask, req_val, fell_back);
    if ( n_bits == 10u )
	return frame_confidence_staged<10>(mags, req_mask, req_val, fell_back);
    if ( n_bits == 8u )
	return frame_confidence_staged<8>(mags, req_mask, req_val, fell_back);
    return frame_confidence(mags, req_mask, req_val, n_bits);
}

// what one fsk_find_frame() returns (fsk.c:504-511)
struct ScanResult {
    float	conf;
    float	ampl;
    uint64_t	bits;
    uint32_t	start;
};

// The reference's zig-zag scan order (fsk.c:477-484): first, first+s, first-s,
// first+2s, first-2s, ...; an up-step reaching try_max ends the scan, a
// down-step below 0 is skipped.  Closed form: U up-positions (u = 0..U-1),
// D valid down-positions (u = 1..D), J = U + D candidates in total.
struct ZigZag {
    uint32_t first, step, U, D, J;
    uint32_t id;		// which of the receive loop's four scans (2 * fine + carrier), or 4: none of them
    __device__ __forceinline__ ZigZag( uint32_t f, uint32_t mx, uint32_t s )
    {
	id = 4u;
	first = f;
	step = s;
	if ( (int)f >= (int)mx || s == 0 ) {
	    U = D = J = 0;
	} else {
	    U = ( mx - f - 1 ) / s + 1;
	    const uint32_t dmax = f / s;
	    D = U - 1 < dmax ? U - 1 : dmax;
	    J = U + D;
	}
    }
    // one of the receive loop's four scans, its counts made on the host
    // (DevCfg::zz_up / zz_down): kind = 2 * fine + carrier
    __device__ __forceinline__ ZigZag( const DevCfg &cfg, uint32_t kind )
    {
	first = cfg.try_first[kind & 1u];
	step = ( kind & 2u ) ? cfg.try_step_fine[kind & 1u] : cfg.try_step[kind & 1u];
	U = cfg.zz_up[kind & 3u];
	D = cfg.zz_down[kind & 3u];
	J = U + D;
	id = kind & 3u;
    }
    // i-th candidate (0-based, scan order)
    // (as selects, not branches: per-lane candidate indices diverge, and a divergent branch
    // costs the wave both sides plus the EXEC bookkeeping)
    __device__ __forceinline__ uint32_t at( uint32_t i ) const
    {
	const bool zig = i <= 2u * D;			// the alternating part: +1, -1, +2, -2, ...
	const uint32_t u = zig ? ( i + 1u ) >> 1 : i - D;	// (i == 0: u = 0)
	const uint32_t off = u * step;
	return ( zig && ( i & 1u ) == 0u ) ? first - off : first + off;
    }
};

constexpr int STAGE_VEC = 10;	// float4 per thread per staging round

// rel / bit_nsamples without a hardware divide: magic = floor(2^32 / B)
// under-estimates the quotient by at most one
__device__ __forceinline__ void divmod_bit( const DevCfg &cfg, uint32_t rel, uint32_t &q, uint32_t &r )
{
    q = __umulhi(rel, cfg.div_magic);
    r = rel - q * cfg.bit_nsamples;
    if ( r >= cfg.bit_nsamples ) {
	q++;
	r -= cfg.bit_nsamples;
    }
}

// x / d for a divisor whose magic = floor(2^32 / d) was computed on the host
__device__ __forceinline__ uint32_t udiv_magic( uint32_t x, uint32_t d, uint32_t magic )
{
    uint32_t q = __umulhi(x, magic);
    if ( x - q * d >= d )
	q++;
    return q;
}

// One aligned float4 of the stream at sample index a (a % 4 == 0), RAW: the
// address is clamped into the row and nothing is done with the data, so that a
// run of these loads is issued back to back and stays in flight together (any
// branch or select on the loaded value makes the compiler drain vmcnt per
// load).  Rows are padded to whole float4s (stream_stride % 4 == 0 and
// N <= stream_stride), so the access is always inside the row.
__device__ __forceinline__ float4 load4_raw( const float *__restrict__ x, uint32_t a, uint32_t N )
{
    const uint32_t aa = a < N ? a : 0u;
    return *reinterpret_cast<const float4 *>(x + aa);	// 16 B per lane, coalesced
}

// ... and the masking that goes with it, applied when the data is consumed:
// samples at or beyond N read as 0.0
__device__ __forceinline__ float4 mask4( float4 s, uint32_t a, uint32_t N )
{
    s.x = a < N ? s.x : 0.0f;
    s.y = ( a < N && a + 1 < N ) ? s.y : 0.0f;
    s.z = ( a < N && a + 2 < N ) ? s.z : 0.0f;
    s.w = ( a < N && a + 3 < N ) ? s.w : 0.0f;
    return s;
}

// Write the float4 loaded from stream index a = org4 + first into a skewed slab
// whose row 0 starts `head` samples after org4 (`first` = offset from org4).
// The LDS word of slab-relative sample rel is rel + (rel / B) * skew: rows of
// one bit length with `skew` pad words in between, so lanes whose windows start
// a whole number of bits apart read different banks.  Samples at or beyond N
// are written as 0.0; samples before row 0 or past `cap` are dropped.
__device__ __forceinline__ void store4_skewed( const DevCfg &cfg, float *slab, uint32_t cap,
	uint32_t first, uint32_t head, float4 s, uint32_t a, uint32_t N )
{
    const uint32_t B = cfg.bit_nsamples, skew = cfg.skew;
    const uint32_t rel0 = first >= head ? first - head : 0u;
    uint32_t q, r;
    divmod_bit(cfg, rel0, q, r);
    const uint32_t idx0 = rel0 + q * skew;
    if ( first >= head && rel0 + 3 < cap && a + 3 < N && a + 3 >= a && B >= 4 ) {
	// common case: four in-range samples, at most one row boundary inside
	float *d = slab + idx0;
	d[0] = s.x;
	d[1 + ( r + 1 >= B ? skew : 0u )] = s.y;
	d[2 + ( r + 2 >= B ? skew : 0u )] = s.z;
	d[3 + ( r + 3 >= B ? skew : 0u )] = s.w;
	return;
    }
    s = mask4(s, a, N);
    const float e[4] = { s.x, s.y, s.z, s.w };
    uint32_t idx = idx0;
#pragma unroll
    for ( int j = 0; j < 4; j++ ) {
	const uint32_t rel = first + j - head;		// meaningful when first + j >= head
	if ( first + j >= head && rel < cap ) {
	    slab[idx] = e[j];
	    idx++;
	    if ( ++r == B ) {
		r = 0;
		idx += skew;
	    }
	}
    }
}

// Make this wave's LDS writes visible to its other lanes.  LDS operations of one
// wave execute in order, so no hardware wait is needed; wavefront-scope fences
// only stop the compiler from reordering across the point.  (An inline-asm
// wait with a "memory" clobber, or a workgroup-scope fence, makes hipcc drain
// vmcnt as well -- which would stall on the global loads prefetched for the
// next batch.)
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t wave_min_u32( uint32_t v )
{
#pragma unroll
    for ( int o = 32; o > 0; o >>= 1 ) {
	const uint32_t t = (uint32_t)__shfl_xor((int)v, o);
	v = t < v ? t : v;
    }
    return v;
}

__device__ __forceinline__ uint32_t wave_max_u32( uint32_t v )
{
#pragma unroll
    for ( int o = 32; o > 0; o >>= 1 ) {
	const uint32_t t = (uint32_t)__shfl_xor((int)v, o);
	v = t > v ? t : v;
    }
    return v;
}

// databits.h:21-46
__device__ __forceinline__ uint64_t bit_window( uint64_t v, uint32_t offset, uint32_t bits )
{
    if ( bits >= 64 )
	return v >> offset;
    const uint64_t mask = ( 1ULL << bits ) - 1ULL;
    return ( v >> offset ) & mask;
}

__device__ __forceinline__ uint64_t bit_reverse( uint64_t v, uint32_t bits )
{
    uint32_t out = 0;		// the reference accumulates in 32 bits
    while ( bits-- ) {
	out = ( out << 1 ) | (uint32_t)( v & 1ULL );
	v >>= 1;
    }
    return out;
}

// frame word -> data bits handed to the databits decoder (minimodem.c:1415-1428)
__device__ __forceinline__ uint64_t data_bits_of( const DevCfg &cfg, uint64_t bits )
{
    if ( cfg.has_stopbits )
	bits >>= 1;
    bits = bit_window(bits, cfg.nstartbits, cfg.n_data_bits);
    if ( cfg.msb_first )
	bits = bit_reverse(bits, cfg.n_data_bits);
    return bits;
}

// v of the lane below (DPP wave_shr:1); lane 0 reads 0
__device__ __forceinline__ float wave_shr1( float v )
{
    const int i = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, i, 0x138, 0xf, 0xf, false));
}

// The lane scan of the bulk replay (master_loop): `steps` = K - 1 applications of
//   t <- (t[l-1] + a) * 0.5, pk <- max(pk[l-1], c), sc <- sc[l-1] + c, sa <- sa[l-1] + a
// with the lower neighbour read through DPP wave_shr:1.  Ping-pong between two
// register sets (the DPP source is never the destination); lane 0 is left alone
// by every DPP instruction (no valid source, bound_ctrl off), and `tmp` holds
// 2 * t in lane 0 so that the plain multiply reproduces its t.  v_max_f32 is the
// reference's `if (pk < c) pk = c` for every input that can occur (pk is never
// NaN; a NaN c leaves pk alone in both).  Asm because the compiler neither folds
// the shuffles into the arithmetic nor keeps lane 0 out of it (15 instructions
// per step instead of 5).
// Finally b* (seeded by the caller with the state before frame 0) receive the
// lower neighbour's result, i.e. the state BEFORE each lane's frame -- inside the
// asm as well: around a DPP intrinsic the compiler narrows EXEC to the lanes whose
// result is used, and a lane whose SOURCE lane is masked off is not written.
// totals == false: the running sums of confidence and amplitude (xsc, xsa: what a NOCARRIER
// line reports, minimodem.c:271-275) are not wanted -- no episode records, no saved state --
// and their two instructions per step are left out; xsc / xsa / bsc / bsa are then untouched.
__device__ __forceinline__ void replay_scan_asm( float &xt, float &xpk, float &xsc, float &xsa,
	float &bt, float &bpk, float &bsc, float &bsa, float cv, float av, uint32_t K, bool totals = true )
{
    float yt = xt, ypk = xpk, ysc = xsc, ysa = xsa;
    float tmp = xt + xt;
    const uint32_t pairs = K / 2u;		// 2 * pairs >= K - 1 steps
    if ( !totals ) {
#define MIFSK_SCAN_STEP_LEAN(ST, SPK, DT, DPK)								\
	"v_add_f32_dpp %[tmp], " ST ", %[av] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"		\
	"v_max_f32_dpp " DPK ", " SPK ", %[cv] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"		\
	"s_nop 0\n\t"											\
	"v_mul_f32_e32 " DT ", 0.5, %[tmp]\n\t"								\
	"s_nop 1\n\t"
	// (DT and DPK are read through DPP by the next step: two wait states after each write)
	uint32_t n = pairs;
	asm volatile(
	    "s_nop 1\n\t"
	    "s_cmp_eq_u32 %[n], 0\n\t"
	    "s_cbranch_scc1 2f\n\t"
	    "1:\n\t"
	    MIFSK_SCAN_STEP_LEAN("%[xt]", "%[xpk]", "%[yt]", "%[ypk]")
	    "s_sub_u32 %[n], %[n], 1\n\t"
	    MIFSK_SCAN_STEP_LEAN("%[yt]", "%[ypk]", "%[xt]", "%[xpk]")
	    "s_cmp_lg_u32 %[n], 0\n\t"
	    "s_cbranch_scc1 1b\n\t"
	    "s_nop 1\n\t"
	    "2:\n\t"
	    "v_mov_b32_dpp %[bt], %[xt] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
	    "v_mov_b32_dpp %[bpk], %[xpk] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
	    "s_nop 1\n\t"
	    : [bt] "+v"(bt), [bpk] "+v"(bpk), [xt] "+v"(xt), [xpk] "+v"(xpk),
	      [yt] "+v"(yt), [ypk] "+v"(ypk), [tmp] "+v"(tmp), [n] "+s"(n)
	    : [cv] "v"(cv), [av] "v"(av)
	    : "scc");
#undef MIFSK_SCAN_STEP_LEAN
	return;
    }
#define MIFSK_SCAN_STEP(ST, SPK, SSC, SSA, DT, DPK, DSC, DSA)						\
	"v_add_f32_dpp %[tmp], " ST ", %[av] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"		\
	"v_mul_f32_e32 " DT ", 0.5, %[tmp]\n\t"	/* >= 2 instructions before DT is read by DPP */	\
	"v_max_f32_dpp " DPK ", " SPK ", %[cv] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"		\
	"v_add_f32_dpp " DSC ", " SSC ", %[cv] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"		\
	"v_add_f32_dpp " DSA ", " SSA ", %[av] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
    uint32_t n = pairs;
    // s_nop: a VGPR written by a VALU instruction may be read through DPP only two
    // wait states later, and the compiler's hazard recogniser does not look into
    // (or out of) an asm block
    asm volatile(
	"s_nop 1\n\t"
	"s_cmp_eq_u32 %[n], 0\n\t"
	"s_cbranch_scc1 2f\n\t"
	"1:\n\t"
	MIFSK_SCAN_STEP("%[xt]", "%[xpk]", "%[xsc]", "%[xsa]", "%[yt]", "%[ypk]", "%[ysc]", "%[ysa]")
	"s_sub_u32 %[n], %[n], 1\n\t"
	MIFSK_SCAN_STEP("%[yt]", "%[ypk]", "%[ysc]", "%[ysa]", "%[xt]", "%[xpk]", "%[xsc]", "%[xsa]")
	"s_cmp_lg_u32 %[n], 0\n\t"
	"s_cbranch_scc1 1b\n\t"
	"s_nop 1\n\t"
	"2:\n\t"
	"v_mov_b32_dpp %[bt], %[xt] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
	"v_mov_b32_dpp %[bpk], %[xpk] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
	"v_mov_b32_dpp %[bsc], %[xsc] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
	"v_mov_b32_dpp %[bsa], %[xsa] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
	"s_nop 1\n\t"
	: [bt] "+v"(bt), [bpk] "+v"(bpk), [bsc] "+v"(bsc), [bsa] "+v"(bsa), [xt] "+v"(xt), [xpk] "+v"(xpk), [xsc] "+v"(xsc), [xsa] "+v"(xsa),
	  [yt] "+v"(yt), [ypk] "+v"(ypk), [ysc] "+v"(ysc), [ysa] "+v"(ysa),
	  [tmp] "+v"(tmp), [n] "+s"(n)
	: [cv] "v"(cv), [av] "v"(av)
	: "scc");
#undef MIFSK_SCAN_STEP
}

// The same lane scan for modes whose carrier-held search step is one sample
// (12000 baud at 48 kHz): there "refine" is a flag and no search
// (minimodem.c:1357 requires try_step_nsamples > 1), so a frame whose confidence
// falls below 0.75 x the running peak is an ordinary frame with one side effect,
//     peak_confidence = 0  (:1281)   and then   peak_confidence = confidence  (:1392-1393),
// i.e. the peak recurrence is  pk <- (c < 0.75 pk) ? c : max(pk, c)  instead of a
// plain running maximum: three more instructions per step than replay_scan_asm
// (the threshold 0.75 * pk[l-1] and the maximum through DPP, then a select).
// Lane 0 keeps its seed as there: the DPP instructions never write it, its
// threshold stays -inf (so the comparison is false) and its `mx` stays the seed.
__device__ __forceinline__ void replay_scan_soft( float &xt, float &xpk, float &xsc, float &xsa,
	float &bt, float &bpk, float &bsc, float &bsa, float cv, float av, uint32_t K, uint32_t lane,
	bool totals = true )
{
    (void)lane;
    float yt = xt, ypk = xpk, ysc = xsc, ysa = xsa;
    float tmp = xt + xt;
    float thr = -INFINITY, mx = xpk;
    const float k075 = 0.75f;
    const uint32_t pairs = K / 2u;		// 2 * pairs >= K - 1 steps
    if ( !totals ) {
	// (replay_scan_asm: without the two running sums; the step's other six instructions and
	// their spacing are those of the full step below)
#define MIFSK_SOFT_STEP_LEAN(ST, SPK, DT, DPK)								\
	"v_add_f32_dpp %[tmp], " ST ", %[av] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"		\
	"v_mul_f32_dpp %[thr], " SPK ", %[k075] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"		\
	"v_max_f32_dpp %[mx], " SPK ", %[cv] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"		\
	"v_mul_f32_e32 " DT ", 0.5, %[tmp]\n\t"								\
	"v_cmp_lt_f32_e32 vcc, %[cv], %[thr]\n\t"								\
	"v_cndmask_b32_e32 " DPK ", %[mx], %[cv], vcc\n\t"
	uint32_t n = pairs;
	asm volatile(
	    "s_nop 1\n\t"
	    "s_cmp_eq_u32 %[n], 0\n\t"
	    "s_cbranch_scc1 2f\n\t"
	    "1:\n\t"
	    MIFSK_SOFT_STEP_LEAN("%[xt]", "%[xpk]", "%[yt]", "%[ypk]")
	    "s_sub_u32 %[n], %[n], 1\n\t"
	    "s_nop 0\n\t"
	    MIFSK_SOFT_STEP_LEAN("%[yt]", "%[ypk]", "%[xt]", "%[xpk]")
	    "s_cmp_lg_u32 %[n], 0\n\t"
	    "s_nop 0\n\t"
	    "s_cbranch_scc1 1b\n\t"
	    "s_nop 1\n\t"
	    "2:\n\t"
	    "v_mov_b32_dpp %[bt], %[xt] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
	    "v_mov_b32_dpp %[bpk], %[xpk] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
	    "s_nop 1\n\t"
	    : [bt] "+v"(bt), [bpk] "+v"(bpk), [xt] "+v"(xt), [xpk] "+v"(xpk),
	      [yt] "+v"(yt), [ypk] "+v"(ypk), [tmp] "+v"(tmp), [thr] "+v"(thr), [mx] "+v"(mx), [n] "+s"(n)
	    : [cv] "v"(cv), [av] "v"(av), [k075] "v"(k075)
	    : "scc", "vcc");
#undef MIFSK_SOFT_STEP_LEAN
	return;
    }
#define MIFSK_SOFT_STEP(ST, SPK, SSC, SSA, DT, DPK, DSC, DSA)						\
	"v_add_f32_dpp %[tmp], " ST ", %[av] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"		\
	"v_mul_f32_dpp %[thr], " SPK ", %[k075] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"		\
	"v_max_f32_dpp %[mx], " SPK ", %[cv] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"		\
	"v_mul_f32_e32 " DT ", 0.5, %[tmp]\n\t"								\
	"v_cmp_lt_f32_e32 vcc, %[cv], %[thr]\n\t"								\
	"v_add_f32_dpp " DSC ", " SSC ", %[cv] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"		\
	"v_add_f32_dpp " DSA ", " SSA ", %[av] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"		\
	"v_cndmask_b32_e32 " DPK ", %[mx], %[cv], vcc\n\t"
    uint32_t n = pairs;
    // (s_nop: a VGPR written by a VALU instruction may be read through DPP only two
    // wait states later; inside a step every DPP source was written >= 2 instructions
    // earlier, the last-written DPK is first read three instructions into the next step)
    asm volatile(
	"s_nop 1\n\t"
	"s_cmp_eq_u32 %[n], 0\n\t"
	"s_cbranch_scc1 2f\n\t"
	"1:\n\t"
	MIFSK_SOFT_STEP("%[xt]", "%[xpk]", "%[xsc]", "%[xsa]", "%[yt]", "%[ypk]", "%[ysc]", "%[ysa]")
	"s_sub_u32 %[n], %[n], 1\n\t"
	"s_nop 0\n\t"
	MIFSK_SOFT_STEP("%[yt]", "%[ypk]", "%[ysc]", "%[ysa]", "%[xt]", "%[xpk]", "%[xsc]", "%[xsa]")
	"s_cmp_lg_u32 %[n], 0\n\t"
	"s_nop 0\n\t"
	"s_cbranch_scc1 1b\n\t"
	"s_nop 1\n\t"
	"2:\n\t"
	"v_mov_b32_dpp %[bt], %[xt] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
	"v_mov_b32_dpp %[bpk], %[xpk] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
	"v_mov_b32_dpp %[bsc], %[xsc] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
	"v_mov_b32_dpp %[bsa], %[xsa] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
	"s_nop 1\n\t"
	: [bt] "+v"(bt), [bpk] "+v"(bpk), [bsc] "+v"(bsc), [bsa] "+v"(bsa),
	  [xt] "+v"(xt), [xpk] "+v"(xpk), [xsc] "+v"(xsc), [xsa] "+v"(xsa),
	  [yt] "+v"(yt), [ypk] "+v"(ypk), [ysc] "+v"(ysc), [ysa] "+v"(ysa),
	  [tmp] "+v"(tmp), [thr] "+v"(thr), [mx] "+v"(mx), [n] "+s"(n)
	: [cv] "v"(cv), [av] "v"(av), [k075] "v"(k075)
	: "scc", "vcc");
#undef MIFSK_SOFT_STEP
}

// ... and where the running peak is DEAD.  With a carrier-held search step of one sample the
// refine rule never searches (minimodem.c:1357 wants try_step_nsamples > 1) and sets no flag:
// peak_confidence is then read by nothing but its own update (:1278-1282,1392-1393) -- no
// predicate, no output, no episode total depends on it.  A call that keeps no loop state for a
// later one (one launch = whole streams) leaves it out: the amplitude tracker alone, two
// instructions per frame instead of six (and the two running sums when episode totals are
// wanted).  xpk / bpk are left as they came in.
__device__ __forceinline__ void replay_scan_track( float &xt, float &xsc, float &xsa,
	float &bt, float &bsc, float &bsa, float cv, float av, uint32_t K, bool totals )
{
    float yt = xt, ysc = xsc, ysa = xsa;
    float tmp = xt + xt;
    uint32_t n = K / 2u;			// 2 * pairs >= K - 1 steps
    if ( !totals ) {
#define MIFSK_TRACK_STEP(ST, DT)										\
	"v_add_f32_dpp %[tmp], " ST ", %[av] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"		\
	"v_mul_f32_e32 " DT ", 0.5, %[tmp]\n\t"								\
	"s_nop 1\n\t"		/* DT is read through DPP by the next step: two wait states */
	asm volatile(
	    "s_nop 1\n\t"
	    "s_cmp_eq_u32 %[n], 0\n\t"
	    "s_cbranch_scc1 2f\n\t"
	    "1:\n\t"
	    MIFSK_TRACK_STEP("%[xt]", "%[yt]")
	    "s_sub_u32 %[n], %[n], 1\n\t"
	    MIFSK_TRACK_STEP("%[yt]", "%[xt]")
	    "s_cmp_lg_u32 %[n], 0\n\t"
	    "s_cbranch_scc1 1b\n\t"
	    "2:\n\t"
	    "v_mov_b32_dpp %[bt], %[xt] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
	    "s_nop 1\n\t"
	    : [bt] "+v"(bt), [xt] "+v"(xt), [yt] "+v"(yt), [tmp] "+v"(tmp), [n] "+s"(n)
	    : [av] "v"(av)
	    : "scc");
#undef MIFSK_TRACK_STEP
	return;
    }
#define MIFSK_TRACK_STEP_T(ST, SSC, SSA, DT, DSC, DSA)							\
	"v_add_f32_dpp %[tmp], " ST ", %[av] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"		\
	"v_add_f32_dpp " DSC ", " SSC ", %[cv] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"		\
	"v_mul_f32_e32 " DT ", 0.5, %[tmp]\n\t"								\
	"v_add_f32_dpp " DSA ", " SSA ", %[av] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"		\
	"s_nop 1\n\t"		/* DT, DSA: two wait states before the next step reads them through DPP */
    asm volatile(
	"s_nop 1\n\t"
	"s_cmp_eq_u32 %[n], 0\n\t"
	"s_cbranch_scc1 2f\n\t"
	"1:\n\t"
	MIFSK_TRACK_STEP_T("%[xt]", "%[xsc]", "%[xsa]", "%[yt]", "%[ysc]", "%[ysa]")
	"s_sub_u32 %[n], %[n], 1\n\t"
	MIFSK_TRACK_STEP_T("%[yt]", "%[ysc]", "%[ysa]", "%[xt]", "%[xsc]", "%[xsa]")
	"s_cmp_lg_u32 %[n], 0\n\t"
	"s_cbranch_scc1 1b\n\t"
	"2:\n\t"
	"v_mov_b32_dpp %[bt], %[xt] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
	"v_mov_b32_dpp %[bsc], %[xsc] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
	"v_mov_b32_dpp %[bsa], %[xsa] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
	"s_nop 1\n\t"
	: [bt] "+v"(bt), [bsc] "+v"(bsc), [bsa] "+v"(bsa), [xt] "+v"(xt), [xsc] "+v"(xsc), [xsa] "+v"(xsa),
	  [yt] "+v"(yt), [ysc] "+v"(ysc), [ysa] "+v"(ysa), [tmp] "+v"(tmp), [n] "+s"(n)
	: [cv] "v"(cv), [av] "v"(av)
	: "scc");
#undef MIFSK_TRACK_STEP_T
}

// maximum of v over the wave's 64 lanes (NaN never wins: v_max_f32 returns the other operand),
// as a DPP reduction: row_shr 1, 2, 3 / 4 / 8 within each row of 16, then row_bcast 15 and 31
// across the rows; the result is lane 63's.  (A loop of v_readlane + compare + branch over a
// handful of candidates costs a VALU -> SALU -> branch round trip per candidate.)
__device__ __forceinline__ float wave_max_f32( float v )
{
    const float ninf = -INFINITY;
#define MIFSK_MAX_DPP(CTRL, RM)											\
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, ninf),			\
		__builtin_bit_cast(int, v), (CTRL), (RM), 0xf, false)))
    MIFSK_MAX_DPP(0x111, 0xf);		// row_shr:1
    MIFSK_MAX_DPP(0x112, 0xf);		// row_shr:2
    MIFSK_MAX_DPP(0x114, 0xf);		// row_shr:4  (lane k of a row now holds the max of lanes k-7 .. k)
    MIFSK_MAX_DPP(0x118, 0xf);		// row_shr:8  (lane 15 of each row: the row's max)
    MIFSK_MAX_DPP(0x142, 0xa);		// row_bcast:15 into rows 1 and 3
    MIFSK_MAX_DPP(0x143, 0xc);		// row_bcast:31 into rows 2 and 3
#undef MIFSK_MAX_DPP
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ float lane_bcast( float v, uint32_t src )
{
    return __builtin_bit_cast(float,
	    __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), (int)src));
}

// LATTICE, linear variant (cfg.lat_linear): the bit length, every bit offset and
// the frame step are multiples of 4 samples, so every window of the wave starts
// a multiple of 16 bytes after the first one.  The region then holds the span
// [lo, hi) unskewed with sample lo at its (16-byte aligned) base:
//   stage     one (possibly unaligned) 16-byte global load and ONE ds_write_b128
//             per lane per KiB -- no per-sample address arithmetic at all;
//   correlate two ds_read_b128 per 8 samples (lanes one bit length apart are
//             2-way bank conflicted on b128, the same LDS time as the
//             conflict-free b32 reads of the skewed layout, a quarter of the
//             instructions).
// Same arithmetic, same order: results are identical to the skewed variant.
typedef float float4_u __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ float4 load4_unaligned( const float *__restrict__ x, uint32_t a, uint32_t N )
{
    const uint32_t aa = ( a + 3 < N && a + 3 >= a ) ? a : 0u;	// tail vectors are re-read by element
    const float4_u s = *reinterpret_cast<const float4_u *>(x + aa);
    return make_float4(s.x, s.y, s.z, s.w);
}

// ---------------------------------------------------------------------------
// Correlators with the twiddles in VECTOR registers (both engines).
//
// One lane = one bit window, so at any step every lane needs the SAME twiddle
// w[n].  Instead of streaming the table through the scalar cache (s_load into
// 64 SGPRs, one wait per half chunk -- exposed when a SIMD runs a single wave),
// the table is spread over the lanes: a "group" holds entries 16g .. 16g+15,
// entry (lane & 15) in each lane, identical in the four 16-lane rows.  gfx950's
// DPP64 form
//     v_fmac_f64_dpp acc, w, x  row_newbcast:J
// computes acc += w[lane J of this lane's row] * x in one instruction
// (tools/ubench/dpp_fmac.hip checks it bit for bit against fma(__shfl(w, ..))
// and measures 1.08-1.12 x the issue time of a plain v_fmac_f64).  No scalar
// loads, no SGPRs, nothing to wait for: sample n costs one convert and four of
// these, in index order -- the oracle's sums in the oracle's order.
// Every lane of the wave must be active (a broadcast from a disabled lane
// writes nothing): idle lanes shadow a real window.
// ---------------------------------------------------------------------------

struct TwGroup {
    double	w[4];		// cos_mark, -sin_mark, cos_space, -sin_space of entry 16 g + (lane & 15)
};

typedef double double2_a16 __attribute__((ext_vector_type(2)));

// group g of the table (zero-padded to whole groups plus one)
__device__ __forceinline__ TwGroup tw_group_load( const double *__restrict__ tw, uint32_t g, uint32_t lane )
{
    const double *t = tw + 4 * (size_t)( 16u * g + ( lane & 15u ) );
    const double2_a16 a = *reinterpret_cast<const double2_a16 *>(t);
    const double2_a16 b = *reinterpret_cast<const double2_a16 *>(t + 2);
    TwGroup G;
    G.w[0] = a.x; G.w[1] = a.y; G.w[2] = b.x; G.w[3] = b.y;
    return G;
}

template <int J>
__device__ __forceinline__ void fmac_bcast( double &acc, const double &w, double xd )
{
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
	: "+v"(acc) : "v"(w), "v"(xd), "i"(J));
}

// DPP reads EXEC and its source VGPR without the interlocks ordinary VALU
// operands have, and the compiler does not know these asm statements are DPP:
// pad once before a run of them (5 wait states after an EXEC write by SALU, 2
// after a VALU write of the broadcast register)
__device__ __forceinline__ void dpp_settle()
{
    asm volatile("s_nop 4");
}

template <int J>
__device__ __forceinline__ void fma4_bcast( double (&acc)[4], const TwGroup &G, float x )
{
    const double xd = (double)x;
    fmac_bcast<J>(acc[0], G.w[0], xd);
    fmac_bcast<J>(acc[1], G.w[1], xd);
    fmac_bcast<J>(acc[2], G.w[2], xd);
    fmac_bcast<J>(acc[3], G.w[3], xd);
}

// four consecutive samples at rows J0 .. J0+3 of group G
template <int J0>
__device__ __forceinline__ void quad_bcast( double (&acc)[4], const TwGroup &G, const float4 &s )
{
    fma4_bcast<J0>(acc, G, s.x);
    fma4_bcast<J0 + 1>(acc, G, s.y);
    fma4_bcast<J0 + 2>(acc, G, s.z);
    fma4_bcast<J0 + 3>(acc, G, s.w);
}

// a whole group: 16 samples
__device__ __forceinline__ void group_bcast( double (&acc)[4], const TwGroup &G,
	const float4 &s0, const float4 &s1, const float4 &s2, const float4 &s3 )
{
    quad_bcast<0>(acc, G, s0);
    quad_bcast<4>(acc, G, s1);
    quad_bcast<8>(acc, G, s2);
    quad_bcast<12>(acc, G, s3);
}

// the first `cnt` (1..15) samples of a group, one uniform test per sample
__device__ __forceinline__ void group_bcast_tail( double (&acc)[4], const TwGroup &G,
	const float4 &s0, const float4 &s1, const float4 &s2, const float4 &s3, uint32_t cnt )
{
#define MIFSK_TAIL(J, X) if ( cnt > (J) ) fma4_bcast<J>(acc, G, X)
    MIFSK_TAIL(0, s0.x);  MIFSK_TAIL(1, s0.y);  MIFSK_TAIL(2, s0.z);   MIFSK_TAIL(3, s0.w);
    MIFSK_TAIL(4, s1.x);  MIFSK_TAIL(5, s1.y);  MIFSK_TAIL(6, s1.z);   MIFSK_TAIL(7, s1.w);
    MIFSK_TAIL(8, s2.x);  MIFSK_TAIL(9, s2.y);  MIFSK_TAIL(10, s2.z);  MIFSK_TAIL(11, s2.w);
    MIFSK_TAIL(12, s3.x); MIFSK_TAIL(13, s3.y); MIFSK_TAIL(14, s3.z);  MIFSK_TAIL(15, s3.w);
#undef MIFSK_TAIL
}

// Window of NQ * 4 samples (NQ <= 12) in LDS at a 16-byte aligned address, its
// table resident in registers (tg[g] = group g): everything unrolled, all the
// ds_read_b128 free to issue ahead of the FMAs.
template <int NQ>
__device__ __forceinline__ void corr_lds_fixed( const TwGroup (&tg)[3], const float *p, double (&acc)[4] )
{
    static_assert(NQ >= 1 && NQ <= 12, "three resident groups");
    float4 xs[NQ];
#pragma unroll
    for ( int q = 0; q < NQ; q++ )
	xs[q] = *reinterpret_cast<const float4 *>(p + 4 * q);
    dpp_settle();
    if constexpr ( NQ == 1 ) {
	// A window of four samples (12000 baud at 48 kHz): sample 0's four FMAs and the zeroing
	// of the accumulators are a fifth of its arithmetic.  Entry 0 of both tables is exactly
	// (1.0, -0.0) (cos 0, -sin 0), and the caller's accumulators are +0.0: fma(x, 1.0, +0.0)
	// is x, fma(x, -0.0, +0.0) is x * -0.0 -- up to the SIGN of a zero, which no later
	// operation can see (it is added to, and squared); NaN and infinity propagate the same.
	const double x0 = (double)xs[0].x;
	acc[0] = x0;
	acc[2] = x0;
	acc[1] = x0 * -0.0;
	acc[3] = acc[1];
	fma4_bcast<1>(acc, tg[0], xs[0].y);
	fma4_bcast<2>(acc, tg[0], xs[0].z);
	fma4_bcast<3>(acc, tg[0], xs[0].w);
	return;
    }
#define MIFSK_QUAD(Q)							\
    if ( (Q) < NQ ) {							\
	if ( (Q) % 4 == 0 ) quad_bcast<0>(acc, tg[(Q) / 4], xs[(Q) < NQ ? (Q) : 0]);	\
	if ( (Q) % 4 == 1 ) quad_bcast<4>(acc, tg[(Q) / 4], xs[(Q) < NQ ? (Q) : 0]);	\
	if ( (Q) % 4 == 2 ) quad_bcast<8>(acc, tg[(Q) / 4], xs[(Q) < NQ ? (Q) : 0]);	\
	if ( (Q) % 4 == 3 ) quad_bcast<12>(acc, tg[(Q) / 4], xs[(Q) < NQ ? (Q) : 0]);	\
    }
    MIFSK_QUAD(0) MIFSK_QUAD(1) MIFSK_QUAD(2) MIFSK_QUAD(3) MIFSK_QUAD(4) MIFSK_QUAD(5)
    MIFSK_QUAD(6) MIFSK_QUAD(7) MIFSK_QUAD(8) MIFSK_QUAD(9) MIFSK_QUAD(10) MIFSK_QUAD(11)
#undef MIFSK_QUAD
}

// The same in two halves (quads [0, H) then [H, NQ)), for kernels whose register
// budget cannot hold a whole window's samples at once (128 VGPRs at four waves
// per SIMD, where the other waves cover the second batch's LDS latency).
template <int NQ>
__device__ __forceinline__ void corr_lds_fixed_halves( const TwGroup (&tg)[3], const float *p, double (&acc)[4] )
{
    static_assert(NQ >= 2 && NQ <= 12, "three resident groups");
    constexpr int H = ( NQ + 1 ) / 2;
    float4 xs[H];
#pragma unroll
    for ( int q = 0; q < H; q++ )
	xs[q] = *reinterpret_cast<const float4 *>(p + 4 * q);
    dpp_settle();
#define MIFSK_QUADH(Q, X)								\
    {											\
	if ( (Q) % 4 == 0 ) quad_bcast<0>(acc, tg[(Q) / 4 < 3 ? (Q) / 4 : 0], X);		\
	if ( (Q) % 4 == 1 ) quad_bcast<4>(acc, tg[(Q) / 4 < 3 ? (Q) / 4 : 0], X);		\
	if ( (Q) % 4 == 2 ) quad_bcast<8>(acc, tg[(Q) / 4 < 3 ? (Q) / 4 : 0], X);		\
	if ( (Q) % 4 == 3 ) quad_bcast<12>(acc, tg[(Q) / 4 < 3 ? (Q) / 4 : 0], X);		\
    }
    if ( 0 < H ) MIFSK_QUADH(0, xs[0])
    if ( 1 < H ) MIFSK_QUADH(1, xs[1 < H ? 1 : 0])
    if ( 2 < H ) MIFSK_QUADH(2, xs[2 < H ? 2 : 0])
    if ( 3 < H ) MIFSK_QUADH(3, xs[3 < H ? 3 : 0])
    if ( 4 < H ) MIFSK_QUADH(4, xs[4 < H ? 4 : 0])
    if ( 5 < H ) MIFSK_QUADH(5, xs[5 < H ? 5 : 0])
    __builtin_amdgcn_sched_barrier(0);
    float4 ys[H];
#pragma unroll
    for ( int q = 0; q < H; q++ )
	ys[q] = *reinterpret_cast<const float4 *>(p + 4 * ( H + q < NQ ? H + q : NQ - 1 ));
    if ( H + 0 < NQ ) MIFSK_QUADH(H + 0, ys[0])
    if ( H + 1 < NQ ) MIFSK_QUADH(H + 1, ys[1 < H ? 1 : 0])
    if ( H + 2 < NQ ) MIFSK_QUADH(H + 2, ys[2 < H ? 2 : 0])
    if ( H + 3 < NQ ) MIFSK_QUADH(H + 3, ys[3 < H ? 3 : 0])
    if ( H + 4 < NQ ) MIFSK_QUADH(H + 4, ys[4 < H ? 4 : 0])
    if ( H + 5 < NQ ) MIFSK_QUADH(H + 5, ys[5 < H ? 5 : 0])
#undef MIFSK_QUADH
}

// Window of nq * 4 samples in LDS at a 16-byte aligned address, any length: one
// table group (two 16-byte global loads per lane, L1-resident) and four
// ds_read_b128 per 16 samples, the next group's loads issued before this
// group's FMAs.  Reads up to 12 samples past the window (the region has slack).
__device__ __forceinline__ void corr_lds_stream( const double *__restrict__ tw, const float *p,
	uint32_t nq, uint32_t lane, double (&acc)[4] )
{
    const uint32_t ng = ( nq + 3u ) >> 2;
    TwGroup G = tw_group_load(tw, 0, lane);
    float4 s0 = *reinterpret_cast<const float4 *>(p);
    float4 s1 = *reinterpret_cast<const float4 *>(p + 4);
    float4 s2 = *reinterpret_cast<const float4 *>(p + 8);
    float4 s3 = *reinterpret_cast<const float4 *>(p + 12);
    // every group but the last, with the next one's loads in flight (nothing is
    // left outstanding at the end: a load nobody waits for makes hipcc drain
    // vmcnt at the next join, and with it the caller's prefetch)
    for ( uint32_t g = 0; g + 1u < ng; g++ ) {
	const TwGroup Gn = tw_group_load(tw, g + 1u, lane);
	const float *pn = p + 16u * ( g + 1u );
	const float4 n0 = *reinterpret_cast<const float4 *>(pn);
	const float4 n1 = *reinterpret_cast<const float4 *>(pn + 4);
	const float4 n2 = *reinterpret_cast<const float4 *>(pn + 8);
	const float4 n3 = *reinterpret_cast<const float4 *>(pn + 12);
	dpp_settle();						// (G may just have been copied)
	group_bcast(acc, G, s0, s1, s2, s3);
	G = Gn;
	s0 = n0; s1 = n1; s2 = n2; s3 = n3;
    }
    const uint32_t left = ( nq - 4u * ( ng - 1u ) ) * 4u;	// samples in the last group
    dpp_settle();
    if ( left >= 16u )
	group_bcast(acc, G, s0, s1, s2, s3);
    else
	group_bcast_tail(acc, G, s0, s1, s2, s3, left);
}

// The same with only the TABLE a group ahead: the window's own samples are read
// from LDS when their group is due (16 registers instead of 32).  For waves
// that also hold a staging round in registers (the workgroup engine's workers)
// at four waves per SIMD, where the other waves cover the LDS latency.
__device__ __forceinline__ void corr_lds_stream_lean( const double *__restrict__ tw, const float *p,
	uint32_t nq, uint32_t lane, double (&acc)[4] )
{
    const uint32_t ng = ( nq + 3u ) >> 2;
    TwGroup G = tw_group_load(tw, 0, lane);
    for ( uint32_t g = 0; g + 1u < ng; g++ ) {		// (nothing left in flight at the end)
	const TwGroup Gn = tw_group_load(tw, g + 1u, lane);
	const float *pc = p + 16u * g;
	const float4 s0 = *reinterpret_cast<const float4 *>(pc);
	const float4 s1 = *reinterpret_cast<const float4 *>(pc + 4);
	const float4 s2 = *reinterpret_cast<const float4 *>(pc + 8);
	const float4 s3 = *reinterpret_cast<const float4 *>(pc + 12);
	dpp_settle();
	group_bcast(acc, G, s0, s1, s2, s3);
	G = Gn;
    }
    const float *pc = p + 16u * ( ng - 1u );
    const float4 s0 = *reinterpret_cast<const float4 *>(pc);
    const float4 s1 = *reinterpret_cast<const float4 *>(pc + 4);
    const float4 s2 = *reinterpret_cast<const float4 *>(pc + 8);
    const float4 s3 = *reinterpret_cast<const float4 *>(pc + 12);
    const uint32_t left = ( nq - 4u * ( ng - 1u ) ) * 4u;	// samples in the last group
    dpp_settle();
    if ( left >= 16u )
	group_bcast(acc, G, s0, s1, s2, s3);
    else
	group_bcast_tail(acc, G, s0, s1, s2, s3, left);
}

// Window of B samples read straight from global memory at x + a (any
// alignment), any length: 64 bytes per lane per 16 samples, next group in
// flight.  The caller guarantees a + 16 * ceil(B / 16) <= N (the last group is
// loaded whole; samples beyond B are not accumulated).
__device__ __forceinline__ void corr_global_stream( const double *__restrict__ tw,
	const float *__restrict__ xa, uint32_t B, uint32_t lane, double (&acc)[4] )
{
    const uint32_t ng = ( B + 15u ) >> 4;
    TwGroup G = tw_group_load(tw, 0, lane);
    float4_u c0 = *reinterpret_cast<const float4_u *>(xa);
    float4_u c1 = *reinterpret_cast<const float4_u *>(xa + 4);
    float4_u c2 = *reinterpret_cast<const float4_u *>(xa + 8);
    float4_u c3 = *reinterpret_cast<const float4_u *>(xa + 12);
    for ( uint32_t g = 0; g + 1u < ng; g++ ) {		// (see corr_lds_stream: nothing left in flight)
	const TwGroup Gn = tw_group_load(tw, g + 1u, lane);
	const float *pn = xa + 16u * ( g + 1u );
	const float4_u d0 = *reinterpret_cast<const float4_u *>(pn);
	const float4_u d1 = *reinterpret_cast<const float4_u *>(pn + 4);
	const float4_u d2 = *reinterpret_cast<const float4_u *>(pn + 8);
	const float4_u d3 = *reinterpret_cast<const float4_u *>(pn + 12);
	dpp_settle();						// (G may just have been copied)
	group_bcast(acc, G, make_float4(c0.x, c0.y, c0.z, c0.w), make_float4(c1.x, c1.y, c1.z, c1.w),
		    make_float4(c2.x, c2.y, c2.z, c2.w), make_float4(c3.x, c3.y, c3.z, c3.w));
	G = Gn;
	c0 = d0; c1 = d1; c2 = d2; c3 = d3;
    }
    const uint32_t left = B - 16u * ( ng - 1u );
    const float4 s0 = make_float4(c0.x, c0.y, c0.z, c0.w), s1 = make_float4(c1.x, c1.y, c1.z, c1.w);
    const float4 s2 = make_float4(c2.x, c2.y, c2.z, c2.w), s3 = make_float4(c3.x, c3.y, c3.z, c3.w);
    dpp_settle();
    if ( left >= 16u )
	group_bcast(acc, G, s0, s1, s2, s3);
    else
	group_bcast_tail(acc, G, s0, s1, s2, s3, left);
}

// Long windows read from global memory, TILED: one window per lane advancing in
// lockstep means one load instruction touches 64 different cache lines, which the
// CU's address path serves at ~1.2 cycles per line (78-90 cycles per
// global_load_dwordx4, tools/ubench/ta_rate.hip) and which uses 16 of the 128
// bytes each line fill brings in.  Here the wave fetches TILE_K samples of every
// window per step with loads in which TILE_K / 4 adjacent lanes read the
// contiguous bytes of ONE window, parks them in an LDS tile -- row w = the TILE_K
// samples of window w, rows TILE_ROW floats apart -- and every lane reads its own
// row back with aligned ds_read_b128 (row stride 4 words off a multiple of 32:
// conflict-free in every 8-lane group).  One tile per wave, no barrier: a wave's LDS
// operations execute in order.  tools/ubench/longwin.hip: 33.7 -> 17.5 ms for the RTTY
// batch with 64 samples per step.
// [r3] 32 samples per step (one 128-byte line per window and load): the receive loop of
// these modes is a serial chain per stream whose time does not depend on how many other
// streams share the CU (measured: 6 / 5 / 4 / 3 waves per CU -> 14.4 / 17.7 / 17.7 / 28.5 ms
// for 4096 RTTY streams), so what counts is how many waves fit: a 9 kB tile, with the
// shared segments' partial sums aliased onto it, lets twelve waves share a CU's LDS
// where the 17 kB tile allowed six.
constexpr uint32_t TILE_K = 32u;
constexpr uint32_t TILE_ROW = TILE_K + 4u;
constexpr uint32_t TILE_FLOATS = 64u * TILE_ROW;
constexpr uint32_t TILE_LPW = TILE_K / 4u;		// lanes that fetch one window's step (16 bytes each)
constexpr uint32_t TILE_WPL = 64u / TILE_LPW;		// windows per load instruction
constexpr int TILE_GPS = (int)( TILE_K / 16u );		// groups of 16 samples per step
constexpr uint32_t kTileMinBit = 256u;			// bit lengths from here on may go through the tile

// NLD loads per step: windows 0 .. TILE_WPL NLD - 1.  Straight-line steps: every load
// and store of a step is unconditional (a conditional load leaves the compiler with
// an outstanding counter at the join and it waits for everything there).
template <int NLD>
__device__ __forceinline__ void corr_global_tiled_n( const double *__restrict__ tw, const float *__restrict__ x,
	uint32_t a, uint32_t B, uint32_t lane, float *tile, double (&acc)[4] )
{
    const uint32_t sub = lane % TILE_LPW, grp = lane / TILE_LPW;
    // load i of a step: lanes TILE_LPW j .. TILE_LPW j + TILE_LPW - 1 read samples of window TILE_WPL i + j
    uint32_t off[NLD];
#pragma unroll
    for ( int i = 0; i < NLD; i++ )
	off[i] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)( ( TILE_WPL * (uint32_t)i + grp ) << 2 ), (int)a) + 4u * sub;
    float *wr = tile + grp * TILE_ROW + 4u * sub;	// + TILE_WPL i rows
    const float *rd = tile + lane * TILE_ROW;
    const uint32_t ngf = B >> 4;			// whole groups of 16 samples
    const uint32_t tail = B & 15u;			// samples of the last, short group
    const uint32_t nfull = ngf / (uint32_t)TILE_GPS;	// steps of whole groups
    const uint32_t rest = ngf % (uint32_t)TILE_GPS;	// whole groups of the last, short step
    const uint32_t last_step = ( rest | tail ) ? nfull : nfull - 1u;
    float4 L[NLD];
#define MIFSK_TILE_FETCH(S)								\
    _Pragma("unroll")									\
    for ( int i = 0; i < NLD; i++ ) {							\
	const float4_u v = *reinterpret_cast<const float4_u *>(x + off[i] + TILE_K * (S));	\
	L[i] = make_float4(v.x, v.y, v.z, v.w);						\
    }
#define MIFSK_TILE_WRITE()								\
    _Pragma("unroll")									\
    for ( int i = 0; i < NLD; i++ )							\
	*reinterpret_cast<float4 *>(wr + TILE_WPL * (uint32_t)i * TILE_ROW) = L[i];
    MIFSK_TILE_FETCH(0u)
    TwGroup G = tw_group_load(tw, 0, lane);
    for ( uint32_t s = 0; s < nfull; s++ ) {
	MIFSK_TILE_WRITE()
	const uint32_t sn = s < last_step ? s + 1u : last_step;		// (the last step is fetched twice)
	MIFSK_TILE_FETCH(sn)
	float4 xs[TILE_GPS][4];
#pragma unroll
	for ( int h = 0; h < TILE_GPS; h++ )
#pragma unroll
	    for ( int j = 0; j < 4; j++ )
		xs[h][j] = *reinterpret_cast<const float4 *>(rd + 16 * h + 4 * j);
	// (the table is padded by a group: loading group g + 1 is always legal)
#pragma unroll
	for ( int h = 0; h < TILE_GPS; h++ ) {
	    const TwGroup Gn = tw_group_load(tw, (uint32_t)TILE_GPS * s + (uint32_t)h + 1u, lane);
	    dpp_settle();
	    group_bcast(acc, G, xs[h][0], xs[h][1], xs[h][2], xs[h][3]);
	    G = Gn;
	}
    }
    if ( rest | tail ) {
	// the short step: `rest` whole groups, then `tail` samples
	MIFSK_TILE_WRITE()
	float4 xs[4];
	for ( uint32_t h = 0; h < rest; h++ ) {
	    const TwGroup Gn = tw_group_load(tw, (uint32_t)TILE_GPS * nfull + h + 1u, lane);
	    const float *r = rd + 16u * h;
#pragma unroll
	    for ( int j = 0; j < 4; j++ )
		xs[j] = *reinterpret_cast<const float4 *>(r + 4 * j);
	    dpp_settle();
	    group_bcast(acc, G, xs[0], xs[1], xs[2], xs[3]);
	    G = Gn;
	}
	if ( tail ) {
	    const float *r = rd + 16u * rest;
#pragma unroll
	    for ( int j = 0; j < 4; j++ )
		xs[j] = *reinterpret_cast<const float4 *>(r + 4 * j);
	    dpp_settle();
	    group_bcast_tail(acc, G, xs[0], xs[1], xs[2], xs[3], tail);
	}
    }
#undef MIFSK_TILE_FETCH
#undef MIFSK_TILE_WRITE
}

// ---------------------------------------------------------------------------
// Shared segments (SegPlan, mifsk_device.h; Wave::seg_correlate): through the same tile
// every lane sums a SEGMENT of its own length `len` (at most the pass's lock-step length)
// -- samples at or beyond a lane's length enter as 0.0, which leaves its sums untouched
// -- and keeps the sum of x^2 over its segment for the error bound.  One group of 16
// samples; groups that lie below the pass's shortest segment need no mask (`whole`).
// ---------------------------------------------------------------------------
typedef float float2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void seg_group( double (&acc)[4], float2v &esum, const TwGroup &G,
	float4 s0, float4 s1, float4 s2, float4 s3, uint32_t n0, uint32_t len, bool whole )
{
    if ( !whole ) {
#define MIFSK_SEG_MASK(V, J) V = ( n0 + (J) < len ) ? V : 0.0f
	MIFSK_SEG_MASK(s0.x, 0u);  MIFSK_SEG_MASK(s0.y, 1u);  MIFSK_SEG_MASK(s0.z, 2u);  MIFSK_SEG_MASK(s0.w, 3u);
	MIFSK_SEG_MASK(s1.x, 4u);  MIFSK_SEG_MASK(s1.y, 5u);  MIFSK_SEG_MASK(s1.z, 6u);  MIFSK_SEG_MASK(s1.w, 7u);
	MIFSK_SEG_MASK(s2.x, 8u);  MIFSK_SEG_MASK(s2.y, 9u);  MIFSK_SEG_MASK(s2.z, 10u); MIFSK_SEG_MASK(s2.w, 11u);
	MIFSK_SEG_MASK(s3.x, 12u); MIFSK_SEG_MASK(s3.y, 13u); MIFSK_SEG_MASK(s3.z, 14u); MIFSK_SEG_MASK(s3.w, 15u);
#undef MIFSK_SEG_MASK
    }
    dpp_settle();
    group_bcast(acc, G, s0, s1, s2, s3);
    // sum of squares, two samples per instruction (v_pk_fma_f32): the error bound takes
    // sum |x| <= sqrt(n * sum x^2) from it
    esum = __builtin_elementwise_fma(float2v{s0.x, s0.y}, float2v{s0.x, s0.y}, esum);
    esum = __builtin_elementwise_fma(float2v{s0.z, s0.w}, float2v{s0.z, s0.w}, esum);
    esum = __builtin_elementwise_fma(float2v{s1.x, s1.y}, float2v{s1.x, s1.y}, esum);
    esum = __builtin_elementwise_fma(float2v{s1.z, s1.w}, float2v{s1.z, s1.w}, esum);
    esum = __builtin_elementwise_fma(float2v{s2.x, s2.y}, float2v{s2.x, s2.y}, esum);
    esum = __builtin_elementwise_fma(float2v{s2.z, s2.w}, float2v{s2.z, s2.w}, esum);
    esum = __builtin_elementwise_fma(float2v{s3.x, s3.y}, float2v{s3.x, s3.y}, esum);
    esum = __builtin_elementwise_fma(float2v{s3.z, s3.w}, float2v{s3.z, s3.w}, esum);
}

//   a     absolute start of this lane's window (idle lanes: any valid window)
//   nwin  lanes 0 .. nwin-1 hold windows (uniform)
// The caller guarantees a + TILE_K * ceil(B / TILE_K) <= N for every lane (whole steps
// are loaded; what lies beyond B is not accumulated).
__device__ __forceinline__ void corr_global_tiled( const double *__restrict__ tw, const float *__restrict__ x,
	uint32_t a, uint32_t nwin, uint32_t B, uint32_t lane, float *tile, double (&acc)[4] )
{
    if ( nwin > 32u )
	corr_global_tiled_n<(int)( 64u / TILE_WPL )>(tw, x, a, B, lane, tile, acc);
    else
	corr_global_tiled_n<(int)( 32u / TILE_WPL )>(tw, x, a, B, lane, tile, acc);
}

// Window in a slab WITHOUT pad words (cfg.skew == 0), at any alignment: the
// sample addresses are the window start plus constants, no per-sample row test.
__device__ __forceinline__ void corr_slab_plain( const double *__restrict__ tw, const float *p, uint32_t B,
	uint32_t lane, double (&acc)[4] )
{
    const uint32_t ng = ( B + 15u ) >> 4;
    const uint32_t last = B - 1u;
    TwGroup G = tw_group_load(tw, 0, lane);
    for ( uint32_t g = 0; g + 1u < ng; g++ ) {
	const TwGroup Gn = tw_group_load(tw, g + 1u, lane);
	const float *q = p + 16u * g;
	float xs[16];
#pragma unroll
	for ( int j = 0; j < 16; j++ )
	    xs[j] = q[j];
	dpp_settle();
	group_bcast(acc, G, make_float4(xs[0], xs[1], xs[2], xs[3]), make_float4(xs[4], xs[5], xs[6], xs[7]),
		    make_float4(xs[8], xs[9], xs[10], xs[11]), make_float4(xs[12], xs[13], xs[14], xs[15]));
	G = Gn;
    }
    {
	float xs[16];
#pragma unroll
	for ( int j = 0; j < 16; j++ ) {
	    uint32_t n = 16u * ( ng - 1u ) + (uint32_t)j;
	    n = n < last ? n : last;			// (uniform) never past the window
	    xs[j] = p[n];
	}
	const float4 s0 = make_float4(xs[0], xs[1], xs[2], xs[3]), s1 = make_float4(xs[4], xs[5], xs[6], xs[7]);
	const float4 s2 = make_float4(xs[8], xs[9], xs[10], xs[11]), s3 = make_float4(xs[12], xs[13], xs[14], xs[15]);
	const uint32_t left = B - 16u * ( ng - 1u );
	dpp_settle();
	if ( left >= 16u )
	    group_bcast(acc, G, s0, s1, s2, s3);
	else
	    group_bcast_tail(acc, G, s0, s1, s2, s3, left);
    }
}

// Window held in a SKEWED slab (rows of one bit length, `skew` pad words in
// between; see store4_skewed), starting `rel` samples after slab row 0: one
// ds_read_b32 per sample, 16 at a time.
__device__ __forceinline__ void corr_skewed_stream( const DevCfg &cfg, const double *__restrict__ tw,
	const float *slab, uint32_t rel, uint32_t lane, double (&acc)[4] )
{
    if ( cfg.skew == 0u ) {
	corr_slab_plain(tw, slab + rel, cfg.bit_nsamples, lane, acc);
	return;
    }
    const uint32_t B = cfg.bit_nsamples;
    uint32_t row, col;
    divmod_bit(cfg, rel, row, col);
    const float *p = slab + rel + row * cfg.skew;
    const uint32_t wrap = B - col;	// first n that falls into the next row
    const uint32_t skew = cfg.skew;
    const uint32_t last = B - 1u;
    const uint32_t ng = ( B + 15u ) >> 4;
    TwGroup G = tw_group_load(tw, 0, lane);
    // 16 samples of group g (clamped inside the window)
#define MIFSK_SKEWED_LOAD(XS, GI)							\
    _Pragma("unroll")									\
    for ( int j = 0; j < 16; j++ ) {							\
	uint32_t n = 16u * (GI) + (uint32_t)j;						\
	n = n < last ? n : last;			/* (uniform) never past the window */	\
	XS[j] = p[n + ( n >= wrap ? skew : 0u )];					\
    }
    // every group but the last with the next group's twiddles in flight; nothing
    // is left outstanding at the end (see corr_lds_stream)
    for ( uint32_t g = 0; g + 1u < ng; g++ ) {
	const TwGroup Gn = tw_group_load(tw, g + 1u, lane);
	float xs[16];
	MIFSK_SKEWED_LOAD(xs, g)
	dpp_settle();
	group_bcast(acc, G, make_float4(xs[0], xs[1], xs[2], xs[3]), make_float4(xs[4], xs[5], xs[6], xs[7]),
		    make_float4(xs[8], xs[9], xs[10], xs[11]), make_float4(xs[12], xs[13], xs[14], xs[15]));
	G = Gn;
    }
    {
	float xs[16];
	MIFSK_SKEWED_LOAD(xs, ng - 1u)
	const float4 s0 = make_float4(xs[0], xs[1], xs[2], xs[3]), s1 = make_float4(xs[4], xs[5], xs[6], xs[7]);
	const float4 s2 = make_float4(xs[8], xs[9], xs[10], xs[11]), s3 = make_float4(xs[12], xs[13], xs[14], xs[15]);
	const uint32_t left = B - 16u * ( ng - 1u );
	dpp_settle();
	if ( left >= 16u )
	    group_bcast(acc, G, s0, s1, s2, s3);
	else
	    group_bcast_tail(acc, G, s0, s1, s2, s3, left);
    }
#undef MIFSK_SKEWED_LOAD
}

} // namespace mifsk
