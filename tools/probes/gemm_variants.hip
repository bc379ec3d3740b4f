// Standalone timing of the bf16 GEMM kernel with parts disabled (GEMM_DBG_* in gemm.hpp).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <vector>
#include <cstdlib>
bool g_prof_on = false;
void prof_begin(int, hipStream_t, const char*) {}
void prof_end(int, hipStream_t) {}
void l4p_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); }
#define GEMM_PROBE_VARIANTS 1
#define GEMM_HAS_8P 1
#include <type_traits>
#include "../../l4p_amd/csrc/gemm8p.hpp"
#include "../../l4p_amd/csrc/gemm4w.hpp"
#define GEMM_T bf16_t
#define GEMM_FN launch_gemm_bf16
#include "../../l4p_amd/csrc/gemm_launch.inc"
int main(int argc, char** argv) {
    { const int gm = getenv("TILE_GM") ? atoi(getenv("TILE_GM")) : 0; hipMemcpyToSymbol(HIP_SYMBOL(g_tile_gm), &gm, sizeof(int)); }
    struct Shape { int M, N, K; const char* name; } shapes[] = {{2048, 4608, 1408, "qkv"}, {2048, 1408, 1408, "proj"},
        {2048, 6144, 1408, "fc1"}, {2048, 1408, 6144, "fc2"}, {8192, 6144, 1408, "fc1_b4"}, {8192, 1408, 6144, "fc2_b4"},
        {131072, 704, 1408, "trk_kv"}, {131072, 2816, 1408, "trk_up0"}, {32768, 6144, 1408, "fc1_b16"}, {8192, 4608, 1408, "qkv_b4"}, {8192, 1408, 1408, "proj_b4"}, {131072, 1408, 704, "trk_i2t"}, {1048576, 704, 352, "trk_up1"}, {262144, 256, 256, "dpt_1x1"}, {16384, 6144, 1408, "fc1_b8"}, {16384, 1408, 6144, "fc2_b8"}, {16384, 4608, 1408, "qkv_b8"}, {16384, 1408, 1408, "proj_b8"}};
    for (auto& s : shapes) {
        if (getenv("SHAPE") && strcmp(getenv("SHAPE"), s.name)) continue;
        const size_t na = (size_t)s.M * s.K, nw = (size_t)(s.N + 255) / 256 * 256 * s.K, nc = (size_t)s.M * s.N;
        std::vector<unsigned short> h(na > nw ? na : nw);
        for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3C00 + (unsigned short)(((i * 2654435761u) >> 20) & 0x3FF) + ((i & 1) << 15);
        void *A, *W, *C; float* bias;
        hipMalloc(&A, na * 2); hipMalloc(&W, nw * 2); hipMalloc(&C, nc * 2); hipMalloc(&bias, s.N * 4);
        hipMemcpy(A, h.data(), na * 2, hipMemcpyHostToDevice); hipMemcpy(W, h.data(), nw * 2, hipMemcpyHostToDevice);
        hipMemset(bias, 0, s.N * 4);
        GemmParams p; memset(&p, 0, sizeof(p));
        p.A = A; p.lda = s.K; p.W = W; p.ldw = s.K; p.M = s.M; p.N = s.N; p.K = s.K; p.bias = bias; p.out_T = C; p.ldc = s.N;
        {   // correctness of the selected variant against the plain 128x128 kernel
            void* Cr; hipMalloc(&Cr, nc * 2); hipMemset(Cr, 0, nc * 2); hipMemset(C, 0, nc * 2);
            GemmParams pr = p; pr.out_T = Cr;
            launch_cfg<128, 128, 0, true>(pr, 0);
            launch_gemm_bf16(0, p, 0);
            hipDeviceSynchronize();
            std::vector<unsigned short> h1(nc), h2(nc);
            hipMemcpy(h1.data(), Cr, nc * 2, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), C, nc * 2, hipMemcpyDeviceToHost);
            size_t bad = 0; for (size_t i = 0; i < nc; ++i) bad += h1[i] != h2[i];
            printf("  [check] %zu / %zu outputs differ from the 128x128 kernel\n", bad, nc);
            hipFree(Cr);
        }
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int i = 0; i < 5; ++i) launch_gemm_bf16(0, p, 0);
        hipEventRecord(a, 0);
        const int it = 50;
        for (int i = 0; i < it; ++i) launch_gemm_bf16(0, p, 0);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-8s M=%d N=%d K=%d: %7.2f us  %7.1f TF/s\n", s.name, s.M, s.N, s.K, ms / it * 1e3, 2.0 * s.M * s.N * s.K / (ms / it * 1e-3) / 1e12);
        hipFree(A); hipFree(W); hipFree(C); hipFree(bias);
    }
    return 0;
}
