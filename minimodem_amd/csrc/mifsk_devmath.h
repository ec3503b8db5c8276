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

} // namespace mifsk

#endif
