// Standalone ablation timing of the two-workgroups-per-CU GEMM (csrc/gemm4w.hpp) next to the 8-phase kernel, N(0,1) bf16 operands:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -Iinclude tools/probes/gemm4w_probe.hip -o /tmp/gemm4w_probe
// Variants: the kernel's DBG bits (1 no epilogue, 2 no LDS-DMA in the loop, 4 no fragment reads, 8 no s_setprio, 16 pieces issued
// together after the barrier, 32 six reads ahead), and "1wg": 150 KB of LDS requested so that one workgroup is resident per CU.
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
bool g_prof_on = false;
void prof_begin(int, hipStream_t, const char*) {}
void prof_end(int, hipStream_t) {}
void l4p_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); }
#define GEMM_HAS_8P 1
#include <type_traits>
#include "../../l4p_amd/csrc/gemm8p.hpp"
#include "../../l4p_amd/csrc/gemm4w.hpp"
#include "../../l4p_amd/csrc/conv3_halo.hpp"
#define GEMM_T bf16_t
#define GEMM_FN launch_gemm_bf16
#include "../../l4p_amd/csrc/gemm_launch.inc"

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); }

template <int DBG, int PSTEP = 3>
static void run4w(const GemmParams& p, int lds) {
    auto kern = gemm4w_kernel<bf16_t, false, DBG, PSTEP>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int grid = ((p.M + 255) / 256) * ((p.N + 127) / 128);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, p);
}
template <class F>
static float time_us(F&& f, int it = 20) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(a, 0);
    for (int i = 0; i < it; ++i) f();
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / it * 1e3f;
}
int main() {
    struct Shape { int M, N, K, act; const char* name; } shapes[] = {
        {8192, 6144, 1408, 1, "fc1_b4 (GELU)"}, {131072, 2816, 1408, 0, "trk 2816"}, {131072, 704, 1408, 0, "trk_kv 704"},
        {1048576, 768, 352, 1, "K352 (GELU)"}, {8192, 4608, 1408, 0, "qkv-shaped"}, {8200, 1416, 1176, 0, "ragged K1176"}};
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto& s : shapes) {
        if (getenv("SHAPE") && !strstr(s.name, getenv("SHAPE"))) continue;
        const size_t na = (size_t)s.M * s.K, nw = (size_t)(s.N + 255) / 256 * 256 * s.K, nc = (size_t)s.M * s.N;
        std::vector<unsigned short> ha(na), hw(nw);
        std::vector<unsigned short> pool(1 << 20);
        for (auto& v : pool) v = f2bf(nd(rng));
        for (size_t i = 0; i < na; ++i) ha[i] = pool[(i * 2654435761u + (i >> 20)) & (pool.size() - 1)];
        const float ws = 1.f / sqrtf((float)s.K);
        for (auto& v : pool) v = f2bf(nd(rng) * ws);
        for (size_t i = 0; i < nw; ++i) hw[i] = pool[(i * 2246822519u + (i >> 20)) & (pool.size() - 1)];
        void *A, *W, *C, *C2; float* bias;
        hipMalloc(&A, na * 2); hipMalloc(&W, nw * 2); hipMalloc(&C, nc * 2); hipMalloc(&C2, nc * 2); hipMalloc(&bias, s.N * 4);
        hipMemcpy(A, ha.data(), na * 2, hipMemcpyHostToDevice); hipMemcpy(W, hw.data(), nw * 2, hipMemcpyHostToDevice);
        hipMemset(bias, 0, s.N * 4);
        GemmParams p; memset(&p, 0, sizeof(p));
        p.A = A; p.lda = s.K; p.W = W; p.ldw = s.K; p.M = s.M; p.N = s.N; p.K = s.K; p.bias = bias; p.out_T = C; p.ldc = s.N; p.act = s.act;
        const double fl = 2.0 * s.M * s.N * s.K;
        auto rep = [&](const char* what, float us) { printf("%-14s %-34s %9.1f us %8.1f TF/s\n", s.name, what, us, fl / us * 1e-6); fflush(stdout); };
        // correctness of the library form against the 8-phase kernel
        {
            GemmParams q = p; q.out_T = C2;
            hipMemset(C, 0, nc * 2); hipMemset(C2, 0, nc * 2);
            launch_8p<0, 2, 4>(q, 0);
            run4w<0>(p, Gemm4wCfg::LDS_BYTES);
            hipDeviceSynchronize();
            std::vector<unsigned short> h1(nc), h2(nc);
            hipMemcpy(h1.data(), C, nc * 2, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), C2, nc * 2, hipMemcpyDeviceToHost);
            size_t bad = 0; for (size_t i = 0; i < nc; ++i) bad += h1[i] != h2[i];
            printf("%-14s [check] %zu / %zu outputs differ between 4w and 8p\n", s.name, bad, nc);
        }
        const int L = Gemm4wCfg::LDS_BYTES, L1 = 150 * 1024;
        rep("8p 256x256", time_us([&] { launch_8p<0, 2, 4>(p, 0); }));
        rep("8p 256x256", time_us([&] { launch_8p<0, 2, 4>(p, 0); }));
        rep("4w", time_us([&] { run4w<0>(p, L); }));
        rep("4w 1wg/CU", time_us([&] { run4w<0>(p, L1); }));
        rep("4w noepi", time_us([&] { run4w<1>(p, L); }));
        rep("4w noepi 1wg/CU", time_us([&] { run4w<1>(p, L1); }));
        rep("4w noepi nodma", time_us([&] { run4w<1 | 2>(p, L); }));
        rep("4w noepi noreads", time_us([&] { run4w<1 | 4>(p, L); }));
        rep("4w noepi nodma noreads", time_us([&] { run4w<1 | 2 | 4>(p, L); }));
        rep("4w noprio", time_us([&] { run4w<8>(p, L); }));
        rep("4w pstep2", time_us([&] { run4w<0, 2>(p, L); }));
        rep("4w pstep4", time_us([&] { run4w<0, 4>(p, L); }));
        rep("4w pstep5", time_us([&] { run4w<0, 5>(p, L); }));
        rep("4w", time_us([&] { run4w<0>(p, L); }));
        rep("8p 256x256", time_us([&] { launch_8p<0, 2, 4>(p, 0); }));
        hipFree(A); hipFree(W); hipFree(C); hipFree(C2); hipFree(bias);
    }
    return 0;
}
