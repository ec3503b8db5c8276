// Micro-benchmark (gfx950): what a wave64 global_load_dwordx4 costs the CU's
// address/L1 path as a function of how the 64 lane addresses are laid out, with
// everything cache-resident (small footprint per wave).  16 waves per CU, each
// issuing `iters` loads with 8 in flight; reports CU cycles per wave-load.
//   hipcc --offload-arch=gfx950 -O3 -o ta_rate ta_rate.hip && ./ta_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if ( e != hipSuccess ) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float __attribute__((ext_vector_type(4), aligned(4))) f4u;

// lane offset (in floats) for pattern p
__device__ __forceinline__ int lane_off( int p, int lane )
{
    switch ( p ) {
    case 0: return lane * 4;				// contiguous 1 KiB
    case 1: return lane * 32;				// one 128-byte line per lane
    case 2: return lane * 32 + 1;			// the same, 4 bytes off alignment
    case 3: return ( lane >> 2 ) * 32 + ( lane & 3 ) * 4;	// quads: 64 contiguous bytes, a line per quad
    case 4: return ( lane >> 2 ) * 32 + ( lane & 3 ) * 4 + 9;	// the same, unaligned (straddles 64 B)
    case 5: return 0;					// every lane the same 16 bytes
    case 6: return ( lane & 3 ) * 4;			// 4 distinct addresses, 64 contiguous bytes
    case 7: return ( lane & 15 ) * 8;			// 16 x 32 bytes (table group: rows read the same)
    case 8: return ( lane >> 4 ) * 32 + ( lane & 15 ) * 4 ;	// rows of 16: 256 contiguous bytes... (64 B per quad)
    case 9: return ( lane >> 1 ) * 32 + ( lane & 1 ) * 4;	// pairs: 32 contiguous bytes, a line per pair
    default: return lane * 16;				// 64-byte stride: two lanes per line
    }
}

__global__ __launch_bounds__(64) void k( const float *x, int iters, int pattern, int step, float *out )
{
    const int lane = threadIdx.x;
    const float *p = x + (size_t)blockIdx.x * 8192 + lane_off(pattern, lane);
    float acc = 0.f;
    for ( int i = 0; i < iters; i += 8 ) {
	f4u v[8];
#pragma unroll
	for ( int j = 0; j < 8; j++ )
	    v[j] = *reinterpret_cast<const f4u *>(p + ( ( i + j ) * step & 2047 ));
#pragma unroll
	for ( int j = 0; j < 8; j++ )
	    acc += v[j].x + v[j].w;
    }
    out[blockIdx.x * 64 + lane] = acc;
}

int main()
{
    const int nwaves = 4096, iters = 16384;
    float *x, *out;
    CK(hipMalloc(&x, (size_t)( nwaves + 2 ) * 8192 * 4));
    CK(hipMemset(x, 0, (size_t)( nwaves + 2 ) * 8192 * 4));
    CK(hipMalloc(&out, nwaves * 64 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[] = { "contiguous 1 KiB", "a 128-byte line per lane", "a line per lane, +4 bytes",
	"64 B per quad, a line per quad", "64 B per quad, unaligned", "all lanes one address",
	"4 addresses (lane & 3)", "16 x 32 B (lane & 15)", "256 B per row of 16", "32 B per pair, a line per pair",
	"64-byte stride" };
    for ( int step : { 4, 16 } ) {
	printf("each lane advances %d bytes per load\n", step * 4);
	for ( int p = 0; p <= 10; p++ ) {
	    hipLaunchKernelGGL(k, dim3(nwaves), dim3(64), 0, 0, x, 1024, p, step, out);
	    CK(hipDeviceSynchronize());
	    CK(hipEventRecord(e0));
	    hipLaunchKernelGGL(k, dim3(nwaves), dim3(64), 0, 0, x, iters, p, step, out);
	    CK(hipEventRecord(e1));
	    CK(hipDeviceSynchronize());
	    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
	    // 16 waves per CU: CU cycles per wave-load at 2.4 GHz
	    const double cyc = ms * 1e-3 * 2.4e9 / ( 16.0 * iters );
	    printf("  %-34s %7.3f ms  %6.1f CU-cycles per wave-load\n", names[p], ms, cyc);
	}
    }
    return 0;
}
