// Microbenchmark + semantics check (gfx950): v_fmac_f64_dpp ... row_newbcast:J
//   d[lane] += src0[16 * (lane / 16) + J] * src1[lane]
// i.e. an f64 FMA whose first factor is broadcast from lane J of the lane's own
// 16-lane row.  The correlators keep the twiddle table spread over the lanes of
// a row and broadcast entry n to every lane when sample n is consumed: no scalar
// loads, no SGPRs.  This file checks (1) the result against a plain fma() with
// __shfl, bit for bit, and (2) the issue rate against a plain v_fmac_f64 with an
// SGPR operand.   hipcc --offload-arch=gfx950 -O3 -o dpp_fmac dpp_fmac.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

template <int J> __device__ __forceinline__ void fmac_bcast( double &acc, const double &w, double xd )
{
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
	: "+v"(acc) : "v"(w), "v"(xd), "i"(J));
}

__global__ void check_kernel( const double *w, const double *x, double *got, double *want )
{
    const int lane = threadIdx.x;
    const double wv = w[lane], xv = x[lane];
    double a = 0.25 * lane;
    double b = a;
#define STEP(J) fmac_bcast<J>(a, wv, xv); b = fma(__shfl(wv, ( lane & ~15 ) + J), xv, b);
    STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7)
    STEP(8) STEP(9) STEP(10) STEP(11) STEP(12) STEP(13) STEP(14) STEP(15)
    got[lane] = a;
    want[lane] = b;
}

template <bool DPP>
__global__ __launch_bounds__(64) void rate_kernel( const double *w, double *out, int iters, double sc )
{
    const int lane = threadIdx.x;
    const double wv = w[lane & 63];
    double a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3;
    double x = 1.0 + 1e-9 * lane;
    for ( int i = 0; i < iters; i++ ) {
	if ( DPP ) {
	    fmac_bcast<0>(a0, wv, x); fmac_bcast<1>(a1, wv, x); fmac_bcast<2>(a2, wv, x); fmac_bcast<3>(a3, wv, x);
	    fmac_bcast<4>(a0, wv, x); fmac_bcast<5>(a1, wv, x); fmac_bcast<6>(a2, wv, x); fmac_bcast<7>(a3, wv, x);
	    fmac_bcast<8>(a0, wv, x); fmac_bcast<9>(a1, wv, x); fmac_bcast<10>(a2, wv, x); fmac_bcast<11>(a3, wv, x);
	    fmac_bcast<12>(a0, wv, x); fmac_bcast<13>(a1, wv, x); fmac_bcast<14>(a2, wv, x); fmac_bcast<15>(a3, wv, x);
	} else {
#pragma unroll
	    for ( int j = 0; j < 4; j++ ) {
		asm volatile("v_fmac_f64_e32 %0, %4, %5\n\tv_fmac_f64_e32 %1, %4, %5\n\t"
			     "v_fmac_f64_e32 %2, %4, %5\n\tv_fmac_f64_e32 %3, %4, %5"
			     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(sc), "v"(x));
	    }
	}
    }
    out[blockIdx.x * 64 + lane] = a0 + a1 + a2 + a3;
}

int main()
{
    double hw[64], hx[64], *dw, *dx, *dg, *dwant;
    for ( int i = 0; i < 64; i++ ) { hw[i] = 1.0 / ( 3.0 + i ); hx[i] = 0.7 + 0.01 * i; }
    hipMalloc(&dw, 512); hipMalloc(&dx, 512); hipMalloc(&dg, 512); hipMalloc(&dwant, 512);
    hipMemcpy(dw, hw, 512, hipMemcpyHostToDevice); hipMemcpy(dx, hx, 512, hipMemcpyHostToDevice);
    check_kernel<<<1, 64>>>(dw, dx, dg, dwant);
    double g[64], wnt[64];
    hipMemcpy(g, dg, 512, hipMemcpyDeviceToHost); hipMemcpy(wnt, dwant, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for ( int i = 0; i < 64; i++ ) bad += memcmp(&g[i], &wnt[i], 8) != 0;
    printf("row_newbcast semantics: %d / 64 lanes differ from fma(__shfl(w, row*16+J), x, acc)\n", bad);

    double *dout; hipMalloc(&dout, 4096 * 64 * 8);
    const int iters = 20000;
    for ( int blocks : { 1024, 4096 } ) {
	for ( int dpp = 0; dpp < 2; dpp++ ) {
	    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	    for ( int rep = 0; rep < 2; rep++ ) {
		hipEventRecord(e0);
		if ( dpp ) rate_kernel<true><<<blocks, 64>>>(dw, dout, iters, 0.999);
		else rate_kernel<false><<<blocks, 64>>>(dw, dout, iters, 0.999);
		hipEventRecord(e1); hipEventSynchronize(e1);
	    }
	    float ms; hipEventElapsedTime(&ms, e0, e1);
	    const double inst = (double)blocks * iters * 16.0;
	    printf("%s  %5d waves: %.3f ms  %.2f wave-FMA/ns  (%.2f cycles per FMA per SIMD at 2.4 GHz, %d wave(s)/SIMD)\n",
		   dpp ? "v_fmac_f64_dpp row_newbcast" : "v_fmac_f64 sgpr            ", blocks, ms,
		   inst / ( ms * 1e6 ), ( ms * 1e-3 * 2.4e9 ) / ( inst / 1024.0 ), blocks / 1024);
	}
    }
    return bad != 0;
}
