// Where do the four waves of each 256-thread workgroup land?  (diagnostic)
// Launches the demod kernel's geometry (1024 x 256 threads, 38 KB LDS each, so 4
// workgroups per CU) and prints, per CU, which SIMD every wave sits on.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <tuple>

__global__ __launch_bounds__(256, 4) void probe(unsigned *out, int spin)
{
    extern __shared__ unsigned char smem[];
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
    float a = threadIdx.x;
    for (int i = 0; i < spin; i++) { a = a * 1.0001f + 0.5f; asm volatile("" : "+v"(a)); }
    if (a == 1234.5f) smem[threadIdx.x] = 1;
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
}

int main()
{
    const int nb = 1024;
    unsigned *d; hipMalloc(&d, nb * 4 * 2 * sizeof(unsigned));
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 38 * 1024);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 38 * 1024, 0, d, 200000);
    std::vector<unsigned> h(nb * 8);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<std::tuple<unsigned,unsigned,unsigned,unsigned>, std::vector<std::tuple<int,int,unsigned,unsigned>>> cus;
    int hist[4][4] = {{0}};
    for (int b = 0; b < nb; b++) for (int w = 0; w < 4; w++) {
        unsigned hw = h[(b * 4 + w) * 2], xcc = h[(b * 4 + w) * 2 + 1] & 0xF;
        unsigned slot = hw & 0xF, simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        cus[{xcc, se, sh, cu}].push_back({b, w, simd, slot});
        hist[w][simd]++;
    }
    printf("CUs seen: %zu\n", cus.size());
    printf("wave index -> SIMD histogram:\n");
    for (int w = 0; w < 4; w++) printf("  wave %d: %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    int shown = 0;
    for (auto &kv : cus) {
        if (shown++ >= 3) break;
        printf("xcc %u se %u sh %u cu %u:", std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first), std::get<3>(kv.first));
        for (auto &t : kv.second) printf(" [b%d w%d simd%u slot%u]", std::get<0>(t), std::get<1>(t), std::get<2>(t), std::get<3>(t));
        printf("\n");
    }
    // how many wave-0s share a SIMD on the same CU
    int worst = 0, total = 0; double avgmax = 0;
    for (auto &kv : cus) {
        int c[4] = {0,0,0,0};
        for (auto &t : kv.second) if (std::get<1>(t) == 0) c[std::get<2>(t)]++;
        int m = 0; for (int i = 0; i < 4; i++) m = c[i] > m ? c[i] : m;
        worst = m > worst ? m : worst; avgmax += m; total++;
    }
    printf("wave-0s on the busiest SIMD of a CU: average %.2f, worst %d\n", avgmax / total, worst);
    return 0;
}
