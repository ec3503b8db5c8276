// mifsk_carrier.hip -- --auto-carrier for a whole batch: what
// main() does before its first search when -a is given (minimodem.c:1179-1220)
// with fsk_detect_carrier (fsk.c:543-581) inside, one workgroup per stream.
//
// Until a carrier is found the reference's loop only shifts, refills and scans
// its buffer, so the set of scan windows is fixed by the buffer arithmetic
// alone; the kernel replays that arithmetic on (pos, nvalid) -- every thread
// the same scalar code -- and evaluates each window's spectrum across the
// threads.  Same sums in the same order as oracle/fsk_oracle.c:
// X[b] = sum_n x[n] e^{-2 pi i (b n mod N)/N} in f64 fma, n ascending.
// gfx950 only.
#include <hip/hip_runtime.h>

#include <cerrno>
#include <cstdint>

#include "mifsk_device.h"
#include "mifsk_devmath.h"

namespace mifsk {

constexpr int SCAN_BLOCK = 256;

// fsk_detect_carrier over one window: the band (>= 1) with the largest
// magnitude among those not below the threshold, the lowest such band on a
// tie (the reference keeps the first strict maximum), or -1.
__device__ int detect_carrier_window( const float *__restrict__ w, uint32_t n_win,
	const double *__restrict__ cs, uint32_t fftsize, uint32_t nbands, float threshold,
	float *s_mag, int *s_band )
{
    const float magscalar = 1.0f / ( (float)n_win / 2.0f );		// fsk.c:553
    float best = 0.0f;
    int best_band = -1;
    for ( uint32_t b = 1u + threadIdx.x; b < nbands; b += SCAN_BLOCK ) {
	double re = 0.0, im = 0.0;
	uint32_t k = 0;						// (b * n) mod fftsize
	for ( uint32_t n = 0; n < n_win; n++ ) {
	    const double x = (double)w[n];			// same address in every lane: one broadcast load
	    re = fma(x, cs[2 * (size_t)k], re);
	    im = fma(x, cs[2 * (size_t)k + 1], im);
	    k += b;
	    if ( k >= fftsize )
		k -= fftsize;
	}
	const float mag = band_mag(re, im, magscalar);
	if ( mag < threshold )					// fsk.c:570-571
	    continue;
	if ( best < mag ) {					// fsk.c:572-575
	    best = mag;
	    best_band = (int)b;
	}
    }
    s_mag[threadIdx.x] = best;
    s_band[threadIdx.x] = best_band;
    __syncthreads();
    for ( int o = SCAN_BLOCK / 2; o > 0; o >>= 1 ) {
	if ( (int)threadIdx.x < o ) {
	    const float m2 = s_mag[threadIdx.x + o];
	    const int b2 = s_band[threadIdx.x + o];
	    const float m1 = s_mag[threadIdx.x];
	    const int b1 = s_band[threadIdx.x];
	    // candidates have band >= 0 and a non-NaN magnitude > 0
	    const bool take = b2 >= 0 && ( b1 < 0 || m2 > m1 || ( m2 == m1 && b2 < b1 ) );
	    if ( take ) {
		s_mag[threadIdx.x] = m2;
		s_band[threadIdx.x] = b2;
	    }
	}
	__syncthreads();
    }
    const int band = s_band[0];
    __syncthreads();						// s_* are reused by the next window
    return band;
}

__global__ __launch_bounds__(SCAN_BLOCK)
void carrier_scan_kernel( CarrierScanArgs a )
{
    __shared__ float s_mag[SCAN_BLOCK];
    __shared__ int s_band[SCAN_BLOCK];
    const uint32_t s = blockIdx.x;
    const float *x = a.d_samples + (size_t)s * a.stream_stride;
    const uint32_t N = a.d_nsamples ? a.d_nsamples[s] : a.nsamples;
    const float nps = a.nsamples_per_scan;
    const uint32_t bufsize = a.samplebuf_size, half = bufsize / 2;

    // (pos, nvalid): absolute index of samplebuf[0] and samples_nvalid
    uint32_t pos = 0, nvalid = 0, advance = 0;
    int band = -1;
    for (;;) {
	if ( advance == bufsize ) {				// minimodem.c:1146-1149
	    nvalid = 0;
	    pos += advance;
	    advance = 0;
	}
	if ( advance ) {					// :1150-1156
	    if ( advance > nvalid )
		break;
	    pos += advance;
	    nvalid -= advance;
	}
	if ( nvalid < half ) {					// :1158-1174
	    const uint32_t got = pos + nvalid;
	    const uint32_t left = N > got ? N - got : 0u;
	    nvalid += left < half ? left : half;
	}
	if ( nvalid == 0 )					// :1176
	    break;
	uint32_t i = 0;
	band = -1;
	while ( (float)i + nps <= (float)nvalid ) {		// :1185-1192 (float arithmetic, as there)
	    band = detect_carrier_window(x + pos + i, (uint32_t)nps, a.d_cs, a.fftsize, a.nbands,
					 a.threshold, s_mag, s_band);
	    if ( band >= 0 )
		break;
	    i = (uint32_t)( (float)i + nps );
	}
	advance = (uint32_t)( (float)i + nps );			// :1193-1195
	if ( advance > nvalid )
	    advance = nvalid;
	if ( band < 0 )
	    continue;
	const int b_space = band + a.b_shift;			// :1209-1213
	if ( b_space < 1 || b_space >= (int)a.nbands ) {
	    band = -1;
	    continue;
	}
	break;
    }
    if ( threadIdx.x == 0 ) {
	a.d_band[s] = band;
	a.d_start[s] = band >= 0 ? pos : N;			// no carrier: nothing to search
    }
}

int launch_carrier_scan( const CarrierScanArgs &a, void *stream )
{
    if ( a.nstreams <= 0 )
	return 0;
    hipLaunchKernelGGL(carrier_scan_kernel, dim3((unsigned)a.nstreams), dim3(SCAN_BLOCK), 0,
		       (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -EIO;
}

} // namespace mifsk
