// Micro-benchmark / probe: can the workgroup engine's workers stage their
// rounds with gfx950's direct-to-LDS loads (global_load_lds_dwordx4) instead of
// global -> VGPR -> ds_write_b128, and what does that buy?
//
// Part 1 (semantics): one wave copies 1 KiB per instruction from a source that is only
// 4-byte aligned (the demod kernel's rounds start at an arbitrary sample), into an LDS
// destination given as a wave-uniform base + immediate offset; vmcnt(0), then ds_read by the
// issuing wave.  Checked against the source for every misalignment 0..3 floats.
//
// Part 2 (the worker loop's skeleton): 1024 workgroups of 192 threads (wave 0 idles like a
// master that is ahead; waves 1-2 are workers), four workgroups per CU (dynamic LDS pads the
// allocation to the demod kernel's), each worker walks its stream in chunks of SV KiB (worker
// w takes chunks w, w + 2, ...) and "correlates" a chunk with a dependent f64 FMA chain of
// `delay` steps per KiB on values read back from LDS -- the wave-level serial dependency of the real
// kernel.  Staging variants:
//   reg   : loads of chunk c+1 in flight in SV float4 registers while chunk c (in LDS) is
//           processed; wait, ds_write_b128 x SV, issue chunk c+2's loads   (the kernel today)
//   glds  : NSLOT LDS slots per worker, chunks c+1 .. c+NSLOT-1 in flight by LDS-DMA while
//           chunk c is processed; counted vmcnt before reading a slot
// Reports TB/s of samples read.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float float4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

template <int N> __device__ __forceinline__ void wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}
__device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// one KiB per call: lane i's 16 bytes from g + OFF land at lds_base + OFF + 16 i (g is the
// lane's own source address; the instruction's immediate offset applies to both sides)
template <int OFF>
__device__ __forceinline__ void glds16( const float *g, float *lds_base )
{
    __builtin_amdgcn_global_load_lds((glb_void *)g, (lds_void *)lds_base, 16, OFF, 0);
}

// ---- part 1 ---------------------------------------------------------------------------------
__global__ void probe_semantics( const float *src, float *out, int mis )
{
    float *lds = reinterpret_cast<float *>(smem);
    const int lane = threadIdx.x;
    for ( int i = lane; i < 2048; i += 64 ) lds[i] = -1.0f;
    __syncthreads();
    const float *g = src + mis + lane * 4;
    float *base = lds + 64;				// wave-uniform, 256 B into the allocation
    glds16<0>(g, base);
    glds16<1024>(g, base);			// (the immediate moves the global AND the LDS address)
    glds16<2048>(g, base);
    wait_vm<0>();
    for ( int i = lane; i < 1024; i += 64 ) out[i] = lds[i];
}

// ---- part 2 ---------------------------------------------------------------------------------
template <int SV>
__device__ __forceinline__ double consume( const float *slot, int lane, int delay, double acc )
{
    // every lane reads "its window" (SV float4, stride SV*4 floats like the kernel's 160 B) and
    // runs a dependent chain on it
    // (read in asm: with an LDS-DMA outstanding hipcc puts s_waitcnt vmcnt(0) in front of every
    // ds_read it knows about -- it cannot tell the slots apart -- which would serialise the ring)
    const uint32_t a = (uint32_t)(uintptr_t)( slot + lane * SV * 4 );
    float s = 0.f;
#pragma unroll
    for ( int i = 0; i < SV; i++ ) {
	float4 v;
	asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(a), "n"(i * 16) : "memory");
	s += v.x + v.y + v.z + v.w;
    }
    acc += (double)s;
    for ( int k = 0; k < delay * SV; k++ ) { acc = fma(acc, 1.0000001, 1e-9); asm volatile("" : "+v"(acc)); }
    return acc;
}

template <int SV>
__global__ __launch_bounds__(192)
void skel_reg( const float *__restrict__ x, size_t stride, int nchunks, int delay, int mis, double *out )
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if ( wave == 0 ) return;
    const int w = wave - 1;
    float *slot = reinterpret_cast<float *>(smem) + 1024 + w * ( SV * 256 );
    const float *base = x + (size_t)blockIdx.x * stride + mis + lane * 4;
    constexpr int CH = SV * 256;			// floats per chunk
    float4 buf[SV];
    double acc = 0.0;
#pragma unroll
    for ( int i = 0; i < SV; i++ ) {
	const float4_u v = *reinterpret_cast<const float4_u *>(base + (size_t)w * CH + i * 256);
	buf[i] = make_float4(v.x, v.y, v.z, v.w);
    }
    for ( int c = w; c < nchunks; c += 2 ) {
#pragma unroll
	for ( int i = 0; i < SV; i++ )
	    *reinterpret_cast<float4 *>(slot + lane * 4 + i * 256) = buf[i];
	const size_t nxt = (size_t)( c + 2 < nchunks ? c + 2 : c ) * CH;
#pragma unroll
	for ( int i = 0; i < SV; i++ ) {
	    const float4_u v = *reinterpret_cast<const float4_u *>(base + nxt + i * 256);
	    buf[i] = make_float4(v.x, v.y, v.z, v.w);
	}
	wait_lgkm();
	acc = consume<SV>(slot, lane, delay, acc);
	wait_lgkm();
    }
    if ( acc == 12345.678 ) out[0] = acc;
}

template <int SV, int I>
__device__ __forceinline__ void issue_chunk( const float *g, float *slot )
{
    if constexpr ( I < SV ) {
	// the immediate is the instruction's 13-bit signed field: hipcc (ROCm 7.2) silently wraps
	// a larger one (4096 -> -4096 with the same base registers), so move both bases instead
	glds16<( I % 4 ) * 1024>(g + ( I / 4 ) * 1024, slot + ( I / 4 ) * 1024);
	issue_chunk<SV, I + 1>(g, slot);
    }
}

template <int SV, int NSLOT>
__global__ __launch_bounds__(192)
void skel_glds( const float *__restrict__ x, size_t stride, int nchunks, int delay, int mis, double *out )
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if ( wave == 0 ) return;
    const int w = __builtin_amdgcn_readfirstlane(wave - 1);
    float *slots = reinterpret_cast<float *>(smem) + 1024 + w * ( NSLOT * SV * 256 );
    const float *base = x + (size_t)blockIdx.x * stride + mis + lane * 4;
    constexpr int CH = SV * 256;
    double acc = 0.0;
    // chunks of this worker: k = 0, 1, ... <-> absolute chunk w + 2k; slot of k = k % NSLOT
    const int mine = ( nchunks - w + 1 ) / 2;
#pragma unroll
    for ( int k = 0; k < NSLOT - 1; k++ )
	issue_chunk<SV, 0>(base + (size_t)( w + 2 * ( k < mine ? k : 0 ) ) * CH, slots + k * CH);
    int sl = 0, sl_in = NSLOT - 1;
    for ( int k = 0; k < mine; k++ ) {
	const int kn = k + NSLOT - 1 < mine ? k + NSLOT - 1 : k;
	issue_chunk<SV, 0>(base + (size_t)( w + 2 * kn ) * CH, slots + sl_in * CH);
	wait_vm<SV * ( NSLOT - 1 )>();			// chunk k has landed
	acc = consume<SV>(slots + sl * CH, lane, delay, acc);
	wait_lgkm();					// its reads are done before the slot is refilled
	sl = sl + 1 == NSLOT ? 0 : sl + 1;
	sl_in = sl_in + 1 == NSLOT ? 0 : sl_in + 1;
    }
    wait_vm<0>();
    if ( acc == 12345.678 ) out[0] = acc;
}

int main( int argc, char **argv )
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    // part 1
    {
	std::vector<float> h(4096);
	for ( int i = 0; i < 4096; i++ ) h[i] = (float)i;
	float *d, *o;
	CK(hipMalloc(&d, 4096 * 4)); CK(hipMalloc(&o, 1024 * 4));
	CK(hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice));
	for ( int mis = 0; mis < 4; mis++ ) {
	    hipLaunchKernelGGL(probe_semantics, dim3(1), dim3(64), 16384, 0, d, o, mis);
	    CK(hipDeviceSynchronize());
	    std::vector<float> r(1024);
	    CK(hipMemcpy(r.data(), o, 1024 * 4, hipMemcpyDeviceToHost));
	    int bad = 0, untouched_ok = 1;
	    for ( int i = 0; i < 64; i++ ) if ( r[i] != -1.0f ) untouched_ok = 0;
	    for ( int i = 0; i < 768; i++ ) if ( r[64 + i] != (float)( mis + i ) ) bad++;
	    for ( int i = 64 + 768; i < 1024; i++ ) if ( r[i] != -1.0f ) untouched_ok = 0;
	    printf("glds dwordx4, source misaligned by %d floats: %d of 768 words wrong, surroundings %s  (first words %g %g %g %g)\n",
		   mis, bad, untouched_ok ? "untouched" : "CLOBBERED", r[64], r[65], r[66], r[67]);
	}
    }
    // part 2
    const int nstreams = 1024; const size_t N = 480000; const size_t stride = N;
    float *d; CK(hipMalloc(&d, ( nstreams * stride + 65536 ) * sizeof(float)));
    CK(hipMemset(d, 0, ( nstreams * stride + 65536 ) * sizeof(float)));
    double *o; CK(hipMalloc(&o, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int lds_bytes = 38 * 1024;			// four workgroups per CU, like the demod kernel
    auto run = [&]( const char *name, auto kern, int sv, int delay, int mis ) {
	const int nchunks = (int)( N / ( sv * 256 ) ) - 1;
	CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
	float best = 1e9f;
	for ( int rep = 0; rep < 4; rep++ ) {
	    CK(hipEventRecord(e0));
	    hipLaunchKernelGGL(kern, dim3(nstreams), dim3(192), lds_bytes, 0, d, stride, nchunks, delay, mis, o);
	    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
	    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
	    if ( rep && ms < best ) best = ms;
	}
	const double bytes = (double)nstreams * nchunks * sv * 1024.0;
	printf("%-34s sv=%2d delay=%4d mis=%d : %.3f ms  %.2f TB/s\n", name, sv, delay, mis, best, bytes / best / 1e9);
    };
    const int mis = argc > 1 ? atoi(argv[1]) : 1;
    // dependent f64 FMAs per KiB (the kernel: ~2.2 k cycles per 10 KiB)
    for ( int delay : { 0, 12, 25, 40 } ) {
	run("reg  10 KiB, 1 ahead", skel_reg<10>, 10, delay, mis);
	run("reg   8 KiB, 1 ahead", skel_reg<8>, 8, delay, mis);
	run("glds  8 KiB x 2 slots (1 ahead)", skel_glds<8, 2>, 8, delay, mis);
	run("glds  5 KiB x 3 slots (2 ahead)", skel_glds<5, 3>, 5, delay, mis);
	run("glds  4 KiB x 4 slots (3 ahead)", skel_glds<4, 4>, 4, delay, mis);
	run("glds  2 KiB x 8 slots (7 ahead)", skel_glds<2, 8>, 2, delay, mis);
    }
    return 0;
}
