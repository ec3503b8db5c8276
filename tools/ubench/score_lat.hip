// Latency of the per-frame confidence pass on ONE wave running alone (gfx950): the workgroup
// engine's master scores a batch of lattice frames -- one lane per frame -- between two
// barriers, and that pass is a link of every stream's serial chain.  Cycles (s_memtime) per
// call of frame_confidence_fixed<11> (as hipcc schedules the plain sequence) and of
// frame_confidence_staged<11> (the same operations laid out for instruction-level
// parallelism), with 7 / 24 / 64 lanes active; a dependent f32 chain and an independent one of
// the same length for scale.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iminimodem_amd/csrc -Iinclude -o score_lat tools/ubench/score_lat.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "mifsk_devlib.h"

using namespace mifsk;

template <int MODE>
__global__ __launch_bounds__(64) void k_score( const float2 *gm, uint32_t nact, int iters, uint32_t *cyc, float *sink )
{
    __shared__ float2 mags[64 * 11];
    for ( int i = threadIdx.x; i < 64 * 11; i += 64 )
	mags[i] = gm[i];
    __syncthreads();
    const uint32_t lane = threadIdx.x;
    float acc = 0.0f;
    uint64_t req_mask = 0x403ull, req_val = 0x401ull;	// "10dddddddd1": bit 0 = 0? (bit k = k-th char)
    asm volatile("" : "+s"(req_mask), "+s"(req_val));
    const uint32_t t0 = (uint32_t)clock64();
    for ( int it = 0; it < iters; it++ ) {
	FrameOut f;
	f.conf = 0.0f; f.ampl = 0.0f; f.bits = 0;
	if ( lane < nact ) {
	    if ( MODE == 0 ) { uint32_t fb = 0u; f = frame_confidence_fixed<11>(&mags[lane * 11], req_mask, req_val, fb); }
	    if ( MODE == 1 ) { uint32_t fb = 0u; f = frame_confidence_staged<11>(&mags[lane * 11], req_mask, req_val, fb); }
	}
	acc += f.conf + f.ampl + (float)f.bits;
	asm volatile("" : "+v"(acc));
	// (the next call reads what this one "wrote": keeps the calls apart like the barrier does)
	asm volatile("" ::: "memory");		// (nothing of a call may be hoisted out of the loop)
	__builtin_amdgcn_wave_barrier();
    }
    const uint32_t t1 = (uint32_t)clock64();
    if ( lane == 0 ) cyc[0] = t1 - t0;
    sink[lane] = acc;
}

template <bool DEP>
__global__ __launch_bounds__(64) void k_chain( int iters, uint32_t *cyc, float *sink )
{
    float a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
    const float k = 1.0001f;
    const uint32_t t0 = (uint32_t)clock64();
    for ( int it = 0; it < iters; it++ ) {
#pragma unroll
	for ( int j = 0; j < 64; j++ ) {
	    if ( DEP ) { a = a * k + 1.0f; a = a * k + 1.0f; a = a * k + 1.0f; a = a * k + 1.0f; }
	    else { a = a * k + 1.0f; b = b * k + 1.0f; c = c * k + 1.0f; d = d * k + 1.0f; }
	}
    }
    const uint32_t t1 = (uint32_t)clock64();
    if ( threadIdx.x == 0 ) cyc[0] = t1 - t0;
    sink[threadIdx.x] = a + b + c + d;
}

int main()
{
    std::vector<float2> hm(64 * 11);
    for ( int l = 0; l < 64; l++ )
	for ( int k = 0; k < 11; k++ ) {
	    const bool one = ( k == 1 ) ? false : ( k == 0 || k == 10 ) ? true : ( ( l * 7 + k * 3 ) % 5 < 2 );
	    // expect "10dddddddd1": char 0 = '1' (previous stop), char 1 = '0' (start), last '1'
	    hm[l * 11 + k] = one ? make_float2(0.9f + 0.01f * k, 0.1f + 0.002f * l) : make_float2(0.12f + 0.001f * l, 0.85f + 0.01f * k);
	}
    float2 *gm; uint32_t *cyc; float *sink;
    hipMalloc(&gm, hm.size() * sizeof(float2)); hipMalloc(&cyc, 4); hipMalloc(&sink, 256);
    hipMemcpy(gm, hm.data(), hm.size() * sizeof(float2), hipMemcpyHostToDevice);
    const int iters = 200;
    auto run = [&]( const char *name, auto launch, double per ) {
	uint32_t c = 0;
	for ( int r = 0; r < 3; r++ ) { launch(); hipDeviceSynchronize(); }
	hipMemcpy(&c, cyc, 4, hipMemcpyDeviceToHost);
	printf("%-44s %8.1f cycles per %s\n", name, (double)c / iters / per, per == 1.0 ? "call" : "instruction");
    };
    for ( uint32_t n : { 7u, 24u, 64u } ) {
	char nm[64];
	snprintf(nm, sizeof nm, "frame_confidence_fixed<11>, %u lanes", n);
	run(nm, [&] { hipLaunchKernelGGL(k_score<0>, dim3(1), dim3(64), 0, 0, gm, n, iters, cyc, sink); }, 1.0);
	snprintf(nm, sizeof nm, "frame_confidence_staged<11>, %u lanes", n);
	run(nm, [&] { hipLaunchKernelGGL(k_score<1>, dim3(1), dim3(64), 0, 0, gm, n, iters, cyc, sink); }, 1.0);
    }
    run("dependent v_fma_f32 chain", [&] { hipLaunchKernelGGL(k_chain<true>, dim3(1), dim3(64), 0, 0, iters, cyc, sink); }, 256.0);
    run("four independent v_fma_f32 chains", [&] { hipLaunchKernelGGL(k_chain<false>, dim3(1), dim3(64), 0, 0, iters, cyc, sink); }, 256.0);
    return 0;
}
