// Standalone timing of the implicit-GEMM 3x3x3 conv kernels on the c3 shapes (band-size experiments: CONV_BT=n).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <cstdlib>
#include <vector>
bool g_prof_on = false;
void prof_begin(int, hipStream_t, const char*) {}
void prof_end(int, hipStream_t) {}
void l4p_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); }
#define GEMM_PROBE_VARIANTS 1
#define GEMM_HAS_8P 1
#include <type_traits>
#include "../../l4p_amd/csrc/gemm8p.hpp"
#define GEMM_T bf16_t
#define GEMM_FN launch_gemm_bf16
#include "../../l4p_amd/csrc/gemm_launch.inc"
int main(int argc, char** argv) {
    struct Shape { int B, T, H, W, Cin, Cout; const char* name; } shapes[] = {
        {4, 16, 64, 64, 256, 256, "rcu_64"}, {4, 16, 32, 32, 256, 256, "rcu_32"}, {4, 16, 128, 128, 256, 128, "head1"},
        {4, 16, 224, 224, 128, 128, "head2"}, {4, 16, 32, 32, 512, 256, "layer_rn1"}};
    const int bt = getenv("CONV_BT") ? atoi(getenv("CONV_BT")) : 0;
    hipMemcpyToSymbol(HIP_SYMBOL(g_conv_bt_override), &bt, sizeof(int));
    for (auto& s : shapes) {
        const long long M = (long long)s.B * s.T * s.H * s.W;
        const int K = 27 * s.Cin;
        const size_t na = (size_t)M * s.Cin, nw = (size_t)(s.Cout + 255) / 256 * 256 * K, nc = (size_t)M * s.Cout;
        std::vector<unsigned short> h(na > nw ? na : nw);
        unsigned st = 777u;
        for (size_t i = 0; i < h.size(); ++i) { st = st * 1664525u + 1013904223u; h[i] = (unsigned short)(0x3C00 + ((st >> 12) & 0x3FF) + ((st >> 31) << 15)); }
        void *A, *W, *C; float* bias;
        hipMalloc(&A, na * 2); hipMalloc(&W, nw * 2); hipMalloc(&C, nc * 2); hipMalloc(&bias, s.Cout * 4);
        hipMemcpy(A, h.data(), na * 2, hipMemcpyHostToDevice); hipMemcpy(W, h.data(), nw * 2, hipMemcpyHostToDevice);
        hipMemset(bias, 0, s.Cout * 4);
        GemmParams p; memset(&p, 0, sizeof(p));
        p.A = A; p.W = W; p.ldw = K; p.M = (int)M; p.N = s.Cout; p.K = K; p.bias = bias; p.out_T = C; p.ldc = s.Cout;
        p.Ti = p.To = s.T; p.Hi = p.Ho = s.H; p.Wi = p.Wo = s.W; p.Cin = s.Cin; p.st = p.sh = p.sw = 1;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int i = 0; i < 3; ++i) launch_gemm_bf16(1, p, 0);
        hipEventRecord(a, 0);
        const int it = 20;
        for (int i = 0; i < it; ++i) launch_gemm_bf16(1, p, 0);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-10s M=%lld N=%d K=%d bt=%d: %8.2f us  %7.1f TF/s\n", s.name, M, s.Cout, K, bt, ms / it * 1e3, 2.0 * M * s.Cout * K / (ms / it * 1e-3) / 1e12);
        hipFree(A); hipFree(W); hipFree(C); hipFree(bias);
    }
    return 0;
}
