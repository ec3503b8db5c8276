// mifsk_devmath.h -- device arithmetic shared by the kernels (included by .hip only)
#ifndef MIFSK_DEVMATH_H
#define MIFSK_DEVMATH_H

#include <hip/hip_runtime.h>

namespace mifsk {

// |X[b]| * scalar, as the reference computes it (fsk.c:107-114): the FFT output
// is a pair of floats; hypotf in glibc 2.35 is (float)sqrt((double)re*re +
// (double)im*im) bit for bit -- pinned to the running C library by a sweep of
// 2 x 2^24 pairs in the CPU suite (tests/test_host_math.py; tools/hypotf_check.c 29 compares
// 10^9) -- with C's one special rule hypot(+-inf, NaN) = +inf, where this returns NaN.  That
// pair arises only in a window whose FIRST sample is infinite (-sin 0 = -0.0, inf * -0.0 =
// NaN) and then in both bands alike: the reference sees bit 0 with signal = noise = inf and
// its confidence sum is inf / inf, here it is NaN / x -- a NaN confidence either way, which
// never wins a search (fsk.c:492; tests/test_gpu_parity.py: non-finite samples, runs of
// infinities three bit windows long included).  f64 sqrt on gfx950 is correctly rounded.
// sqrt() of a double that is 0 or at least 2^-298 (a sum of squares of floats):
// the compiler's own correctly rounded sequence for gfx950 -- v_rsq_f64, one
// coupled Newton step on (g ~ sqrt s, h ~ 1/(2 sqrt s)), two residual
// corrections -- without the exponent pre-scaling it wraps around it for
// arguments below 2^-767, which cannot occur here (5 of its 18 instructions).
// Same instructions, same order: identical results (every parity test compares
// magnitudes bit for bit).
__device__ __forceinline__ double sqrt_sumsq( double s )
{
    const double r = __builtin_amdgcn_rsq(s);
    double g = s * r;
    double h = r * 0.5;
    const double e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g);
    double d = __builtin_fma(-g, g, s);
    h = __builtin_fma(h, e, h);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, s);
    g = __builtin_fma(d, h, g);
    return ( s == 0.0 || s == __builtin_inf() ) ? s : g;
}

__device__ __forceinline__ float band_mag( double re, double im, float scalar )
{
    const float fr = (float)re, fi = (float)im;
    // (the squares of two floats are exact in double, so the one rounding of their sum is
    // the fma's: the same s as mul, mul, add with an instruction less)
    const double s = __builtin_fma((double)fr, (double)fr, (double)fi * (double)fi);
    return (float)sqrt_sumsq(s) * scalar;
}

// (float)sqrt(s) WITHOUT the correctly rounded double in between, where that is provably the
// same float.  v_rsq_f64 is good to 2^29 ulp (a relative 2^-23: the ISA's figure); one coupled
// Newton step squares that: |g - sqrt s| <= 1.5 * 2^-46 g + a few roundings < 2^-45 g, and the
// correctly rounded z = RN53(sqrt s) that sqrt_sumsq() returns lies within 2^-53 more.  g and z
// round to the same float unless a rounding boundary of the float format -- a midpoint between
// neighbours: the 29 mantissa bits below a float's precision reading 2^28 -- lies between them,
// i.e. within 2^-45 * 2^53 = 2^8 units of g's last place.  kSqrtGuard = 2^12 units either side of
// the midpoint are treated as "too close" (sixteen times the bound: it would still hold for an
// rsq four times worse than specified), and s must be finite and at least 2^-250 so that the
// result is a NORMAL float of at least 2^-125 (below that the float's last place is not bit 29
// of the double's mantissa: subnormal-scale audio, tests/test_gpu_parity.py) -- anything else,
// in any lane, sends the whole wave through the exact sequence (2^-16 of the values).
// tests/test_gpu_math.py compares both paths on 2^32 sums of squares, boundary cases included.
constexpr uint32_t kSqrtGuard = 1u << 12;

__device__ __forceinline__ double sqrt_newton1( double s, bool &unsafe )
{
    const double r = __builtin_amdgcn_rsq(s);
    double g = s * r;
    const double h = r * 0.5;
    const double e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g);
    const uint32_t glo = (uint32_t)__double_as_longlong(g);
    const uint32_t shi = (uint32_t)( (unsigned long long)__double_as_longlong(s) >> 32 );
    // low 29 bits within kSqrtGuard of 2^28  <=>  ((glo << 3) - ((2^28 - G) << 3)) < (2 G << 3)
    const bool near = ( glo << 3 ) - ( ( 0x10000000u - kSqrtGuard ) << 3 ) < ( ( 2u * kSqrtGuard ) << 3 );
    // 2^-250 <= s < inf (high word between those of 2^-250 and of infinity; NaN and negatives fall outside)
    const bool in_range = shi - 0x30500000u < 0x7FF00000u - 0x30500000u;
    unsafe = near || !in_range;
    return g;
}

// the two magnitudes of a bit window (mark, space) by the exact sequence: where windows are long
// (SAME, RTTY: two square roots per 92 or 1056 samples) the short path's test and branch cost
// more than its five instructions save (same-box: SAME +0.7 %, RTTY +0.4 % with it)
__device__ __forceinline__ float2 band_mag2_exact( double re0, double im0, double re1, double im1, float scalar )
{
    return make_float2(band_mag(re0, im0, scalar), band_mag(re1, im1, scalar));
}

// ... and with the short square root: band_mag() of each, one test for both (bit windows of 4
// to 48 samples: 12000 baud -3.2 %, Bell-202 -0.7 %)
__device__ __forceinline__ float2 band_mag2( double re0, double im0, double re1, double im1, float scalar )
{
    const float fr0 = (float)re0, fi0 = (float)im0, fr1 = (float)re1, fi1 = (float)im1;
    const double s0 = __builtin_fma((double)fr0, (double)fr0, (double)fi0 * (double)fi0);
    const double s1 = __builtin_fma((double)fr1, (double)fr1, (double)fi1 * (double)fi1);
#ifdef MIFSK_EXACT_SQRT		/* (measurement builds: make variant TAG=exactsqrt DEFS=-DMIFSK_EXACT_SQRT) */
    return make_float2((float)sqrt_sumsq(s0) * scalar, (float)sqrt_sumsq(s1) * scalar);
#else
    bool u0, u1;
    float m0 = (float)sqrt_newton1(s0, u0), m1 = (float)sqrt_newton1(s1, u1);
    if ( __builtin_expect(__any(u0 || u1), 0) ) {
	m0 = (float)sqrt_sumsq(s0);
	m1 = (float)sqrt_sumsq(s1);
    }
    return make_float2(m0 * scalar, m1 * scalar);
#endif
}

} // namespace mifsk

#endif
