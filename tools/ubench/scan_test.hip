// unit check of the replay lane scan (asm) against the sequential recurrences
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>
__device__ __forceinline__ void replay_scan_asm( float &xt, float &xpk, float &xsc, float &xsa,
	float cv, float av, uint32_t K )
{
    float yt = xt, ypk = xpk, ysc = xsc, ysa = xsa;
    float tmp = xt + xt;
    const uint32_t pairs = K / 2u;
    if ( pairs == 0 ) return;
#define STEP(ST, SPK, SSC, SSA, DT, DPK, DSC, DSA) \
	"v_add_f32_dpp %[tmp], " ST ", %[av] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
	"v_mul_f32_e32 " DT ", 0.5, %[tmp]\n\t" \
	"v_max_f32_dpp " DPK ", " SPK ", %[cv] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
	"v_add_f32_dpp " DSC ", " SSC ", %[cv] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
	"v_add_f32_dpp " DSA ", " SSA ", %[av] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
    uint32_t n = pairs;
    asm volatile("s_nop 1\n\t1:\n\t"
	STEP("%[xt]", "%[xpk]", "%[xsc]", "%[xsa]", "%[yt]", "%[ypk]", "%[ysc]", "%[ysa]")
	"s_sub_u32 %[n], %[n], 1\n\t"
	STEP("%[yt]", "%[ypk]", "%[ysc]", "%[ysa]", "%[xt]", "%[xpk]", "%[xsc]", "%[xsa]")
	"s_cmp_lg_u32 %[n], 0\n\ts_cbranch_scc1 1b\n\ts_nop 1\n\t"
	: [xt] "+v"(xt), [xpk] "+v"(xpk), [xsc] "+v"(xsc), [xsa] "+v"(xsa),
	  [yt] "+v"(yt), [ypk] "+v"(ypk), [ysc] "+v"(ysc), [ysa] "+v"(ysa), [tmp] "+v"(tmp), [n] "+s"(n)
	: [cv] "v"(cv), [av] "v"(av) : "scc");
}
__global__ void k(const float *c, const float *a, float *out, uint32_t K, float T, float PK, float SC, float SA)
{
    const uint32_t lane = threadIdx.x;
    const float cv = lane < K ? c[lane] : 0.f, av = lane < K ? a[lane] : 0.f;
    float xt = (T + av) / 2.0f, xpk = PK < cv ? cv : PK, xsc = SC + cv, xsa = SA + av;
    replay_scan_asm(xt, xpk, xsc, xsa, cv, av, K);
    out[lane * 4 + 0] = xt; out[lane * 4 + 1] = xpk; out[lane * 4 + 2] = xsc; out[lane * 4 + 3] = xsa;
}
int main()
{
    float *dc, *da, *dout; hipMalloc(&dc, 256); hipMalloc(&da, 256); hipMalloc(&dout, 1024);
    int bad = 0;
    for (int trial = 0; trial < 200; trial++) {
        uint32_t K = 1 + rand() % 64;
        std::vector<float> c(64), a(64), o(256);
        for (int i = 0; i < 64; i++) { c[i] = (rand() % 1000) / 97.0f; a[i] = (rand() % 1000) / 1013.0f; }
        float T = 0.77f, PK = 3.3f, SC = 100.25f, SA = 17.5f;
        hipMemcpy(dc, c.data(), 256, hipMemcpyHostToDevice); hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dc, da, dout, K, T, PK, SC, SA);
        hipMemcpy(o.data(), dout, 1024, hipMemcpyDeviceToHost);
        float t = T, pk = PK, sc = SC, sa = SA;
        for (uint32_t i = 0; i < K; i++) {
            t = (t + a[i]) / 2.0f; if (pk < c[i]) pk = c[i]; sc += c[i]; sa += a[i];
            if (o[i*4] != t || o[i*4+1] != pk || o[i*4+2] != sc || o[i*4+3] != sa) {
                if (bad < 8) printf("trial %d K %u lane %u: got %g %g %g %g want %g %g %g %g\n", trial, K, i, o[i*4], o[i*4+1], o[i*4+2], o[i*4+3], t, pk, sc, sa);
                bad++; break;
            }
        }
    }
    printf("%d bad trials of 200\n", bad);
    return 0;
}
