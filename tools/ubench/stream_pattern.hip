// Micro-benchmark: the demod kernel's HBM access pattern without the compute.
// 1024 workgroups x NW waves; wave w of block s walks stream s in steps of
// `step` floats, reading `chunk` float4 per lane per step (contiguous 64*16*chunk B).
// DEPTH = how many steps of loads are kept in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int CH, int DEPTH>
__global__ void pattern(const float* __restrict__ x, size_t stride, int nsteps, int step_floats,
                        int wave_off, int delay, float* out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* base = x + (size_t)blockIdx.x * stride + (size_t)wave * wave_off + lane * 4;
    float4 buf[DEPTH][CH];
    float acc = 0.f;
    #pragma unroll
    for (int d = 0; d < DEPTH; d++)
        #pragma unroll
        for (int i = 0; i < CH; i++)
            buf[d][i] = *reinterpret_cast<const float4*>(base + (size_t)d * step_floats + i * 256);
    for (int s = 0; s < nsteps; s += DEPTH) {
        #pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            float4 cur[CH];
            #pragma unroll
            for (int i = 0; i < CH; i++) cur[i] = buf[d][i];
            #pragma unroll
            for (int i = 0; i < CH; i++) acc += cur[i].x + cur[i].y + cur[i].z + cur[i].w;
            const size_t nxt = (size_t)(s + d + DEPTH) * step_floats;
            #pragma unroll
            for (int i = 0; i < CH; i++)
                buf[d][i] = *reinterpret_cast<const float4*>(base + nxt + i * 256);
            for (int k = 0; k < delay; k++) { acc = acc * 1.0000001f + 1e-9f; asm volatile("" : "+v"(acc)); }
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main(int argc, char** argv)
{
    const int nstreams = 1024; const size_t N = 480000; const size_t stride = N;
    float* d; hipMalloc(&d, (nstreams * stride + 65536) * sizeof(float));
    hipMemset(d, 0, (nstreams * stride + 65536) * sizeof(float));
    float* o; hipMalloc(&o, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern, int nw, int ch, int depth, int delay, int misalign) {
        const int step = ch * 256 * nw;               // floats per step per block
        const int nsteps = (int)(N / step) - depth - 1;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(nstreams), dim3(64 * nw), 0, 0, d + misalign, stride, nsteps - nsteps % depth, step, ch * 256, delay, o);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double bytes = (double)nstreams * (nsteps - nsteps % depth) * step * 4.0;
        printf("%-44s nw=%d ch=%d depth=%d delay=%d mis=%d : %.3f ms  %.2f TB/s\n", name, nw, ch, depth, delay, misalign, ms, bytes / ms / 1e9);
    };
    if (argc > 1) {     // single configuration (for counter collection): delay = argv[1]
        run("3 waves x 10 KB, depth 1", pattern<10,1>, 3, 10, 1, atoi(argv[1]), 0);
        return 0;
    }
    for (int delay : {0, 400, 1200, 2500}) {
        run("3 waves x 10 KB, depth 1", pattern<10,1>, 3, 10, 1, delay, 0);
        run("3 waves x 10 KB, depth 1, misaligned", pattern<10,1>, 3, 10, 1, delay, 1);
        run("3 waves x 5 KB, depth 2", pattern<5,2>, 3, 5, 2, delay, 0);
        run("3 waves x 2.5 KB(3) depth 4", pattern<3,4>, 3, 3, 4, delay, 0);
        run("4 waves x 10 KB, depth 1", pattern<10,1>, 4, 10, 1, delay, 0);
    }
    return 0;
}
