// What fits in the gap behind a v_mfma_f32_32x32x16_bf16 when ONE wave owns the SIMD?  (tools/probes/gap_probe.hip)
// A wave issues NIT x { MFMA (4 rotating accumulators) ; K fillers of one kind } and reports shader cycles per MFMA (s_memtime).
// 256 threads = 4 waves per workgroup = one wave per SIMD (160 KB of LDS per workgroup: one workgroup per CU), 256 workgroups.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/gap_probe.hip -o tools/probes/gap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int KIND, int K>
__device__ __forceinline__ void fillers(float (&x)[8], unsigned (&p)[4], u32x4 (&fr)[4], unsigned lds) {
#pragma unroll
    for (int i = 0; i < K; ++i) {
        if constexpr (KIND == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i % 8]));
        if constexpr (KIND == 1) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[i % 8]) : "v"(x[(i + 1) % 8]), "v"(x[(i + 2) % 8]));
        if constexpr (KIND == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p[i % 4]) : "v"(x[i % 8]), "v"(x[(i + 1) % 8]));
        if constexpr (KIND == 3) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[i % 4]) : "v"(lds), "n"(0));
        if constexpr (KIND == 4) asm volatile("s_nop 0");
        if constexpr (KIND == 5) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(x[i % 8]));
        if constexpr (KIND == 6) {  // the softmax mix: exp exp cvt max3 in rotation
            if (i % 4 < 2) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i % 8]));
            else if (i % 4 == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p[i % 4]) : "v"(x[i % 8]), "v"(x[(i + 1) % 8]));
            else asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[i % 8]) : "v"(x[(i + 1) % 8]), "v"(x[(i + 2) % 8]));
        }
    }
}

template <int KIND, int K, bool AGPR>
__global__ __launch_bounds__(256) void gap_kernel(float* out, long long* cyc, int nit) {
    extern __shared__ char smem[];
    f32x16 acc[4] = {};
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) a[e] = (__bf16)(0.01f * (threadIdx.x + e)), b[e] = (__bf16)(1.0f + 0.001f * e);
    float x[8];
    for (int e = 0; e < 8; ++e) x[e] = -0.001f * (threadIdx.x + e);
    unsigned p[4] = {};
    u32x4 fr[4] = {};
    const unsigned lds = (threadIdx.x & 63) * 16;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < nit; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[u]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(a), "v"(b));
            fillers<KIND, K>(x, p, fr, lds);
        }
    }
    // (LDS reads return in order and nothing consumes them here: no wait inside the loop, so the figure is the issue / return-path
    //  cost of a read, not its latency)
    asm volatile("s_waitcnt lgkmcnt(0)");
    const long long t1 = __builtin_readcyclecounter();
    asm volatile("s_nop 15\n s_nop 15");
    float s = 0;
    for (int u = 0; u < 4; ++u)
        for (int r = 0; r < 16; ++r) s += acc[u][r];
    for (int e = 0; e < 8; ++e) s += x[e];
    for (int e = 0; e < 4; ++e) s += p[e] + fr[e][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int K, bool AGPR>
void run(const char* name, float* out, long long* cyc) {
    const int nit = 2000;
    hipFuncSetAttribute((const void*)gap_kernel<KIND, K, AGPR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    gap_kernel<KIND, K, AGPR><<<256, 256, 160 * 1024>>>(out, cyc, nit);
    hipEventRecord(e0);
    gap_kernel<KIND, K, AGPR><<<256, 256, 160 * 1024>>>(out, cyc, nit);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double m = 0;
    for (auto v : h) m += v;
    m /= 256;
    printf("%-10s %s K=%d: %6.1f cycles/MFMA  (%.2f GHz)\n", name, AGPR ? "acc=AGPR" : "acc=VGPR", K, m / (4.0 * nit), m / (ms * 1e6));
}

template <int KIND, bool AGPR>
void sweep(const char* name, float* out, long long* cyc) {
    run<KIND, 0, AGPR>(name, out, cyc);
    run<KIND, 1, AGPR>(name, out, cyc);
    run<KIND, 2, AGPR>(name, out, cyc);
    run<KIND, 3, AGPR>(name, out, cyc);
    run<KIND, 4, AGPR>(name, out, cyc);
    run<KIND, 5, AGPR>(name, out, cyc);
    run<KIND, 6, AGPR>(name, out, cyc);
    run<KIND, 8, AGPR>(name, out, cyc);
}

int main() {
    float* out;
    long long* cyc;
    hipMalloc(&out, 256 * 256 * 4);
    hipMalloc(&cyc, 256 * 8);
    sweep<0, false>("v_exp", out, cyc);
    sweep<0, true>("v_exp", out, cyc);
    sweep<1, false>("v_max3", out, cyc);
    sweep<2, false>("v_cvt_pk", out, cyc);
    sweep<3, false>("ds_read128", out, cyc);
    sweep<4, false>("s_nop", out, cyc);
    sweep<5, false>("v_mul", out, cyc);
    sweep<6, false>("softmax", out, cyc);
    sweep<6, true>("softmax", out, cyc);
    return 0;
}
