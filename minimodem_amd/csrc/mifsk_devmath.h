// mifsk_devmath.h -- device arithmetic shared by the kernels (included by .hip only)
#ifndef MIFSK_DEVMATH_H
#define MIFSK_DEVMATH_H

#include <hip/hip_runtime.h>

namespace mifsk {

// |X[b]| * scalar, as the reference computes it (fsk.c:107-114): the FFT output
// is a pair of floats; hypotf in glibc 2.35 is exactly
// (float)sqrt((double)re*re + (double)im*im) (verified exhaustively on the
// host, tests/test_host_math.py); f64 sqrt on gfx950 is correctly rounded.
__device__ __forceinline__ float band_mag( double re, double im, float scalar )
{
    const float fr = (float)re, fi = (float)im;
    const double s = (double)fr * (double)fr + (double)fi * (double)fi;
    return (float)sqrt(s) * scalar;
}

} // namespace mifsk

#endif
