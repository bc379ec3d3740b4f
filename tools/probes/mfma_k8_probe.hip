// Issue cost of the legacy K=8 bf16 MFMA (v_mfma_f32_32x32x8_bf16_1k) against the gfx950 K=16 form
// (v_mfma_f32_32x32x16_bf16) on one wave per SIMD: does head dim 88 = 5 x K16 + 1 x K8 save matrix-pipe cycles in QK^T?
// Prints shader cycles (s_memtime) per MFMA for independent accumulators (issue rate) and a dependent chain (latency).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE>  // 0: K16 independent, 1: K8 independent, 2: K16 dependent, 3: K8 dependent, 4: 5xK16+1xK8 pattern, 5: 6xK16 pattern
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
    bf16x8 a16, b16;
    s16x4 a8, b8;
    for (int i = 0; i < 8; ++i) { a16[i] = (__bf16)(threadIdx.x * 0.001f + i); b16[i] = (__bf16)(0.5f - i * 0.01f); }
    for (int i = 0; i < 4; ++i) { a8[i] = (short)(0x3f80 + threadIdx.x + i); b8[i] = (short)(0x3f00 + i); }
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a16, b16, acc[j], 0, 0, 0);
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a8, b8, acc[j], 0, 0, 0);
        } else if constexpr (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a16, b16, acc[0], 0, 0, 0);
        } else if constexpr (MODE == 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[0] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a8, b8, acc[0], 0, 0, 0);
        } else if constexpr (MODE == 4) {  // two score tiles as the attention kernel walks them: 5 x K16 + 1 x K8 each (12 MFMAs)
#pragma unroll
            for (int ks = 0; ks < 5; ++ks)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a16, b16, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a8, b8, acc[t], 0, 0, 0);
        } else {
#pragma unroll
            for (int ks = 0; ks < 6; ++ks)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a16, b16, acc[t], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int per_iter, int grid) {
    float* out; long long* cyc;
    hipMalloc(&out, grid * 256 * 4); hipMalloc(&cyc, grid * 8);
    const int iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, cyc, 100);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[4]; hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
    // s_memtime counts at a fixed 100 MHz on this family: report both the counter and the wall time per MFMA
    printf("%-34s grid %4d: counter %8.2f ticks / MFMA, wall %7.2f ns / MFMA per wave\n", name, grid,
           (double)h[0] / ((double)iters * per_iter), ms * 1e6 / ((double)iters * per_iter));
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int grid : {1, 256}) {
        run<0>("32x32x16 bf16, 4 independent", 4, grid);
        run<1>("32x32x8 bf16_1k, 4 independent", 4, grid);
        run<2>("32x32x16 bf16, dependent chain", 4, grid);
        run<3>("32x32x8 bf16_1k, dependent chain", 4, grid);
        run<5>("QK^T pattern 6 x K16 (96)", 12, grid);
        run<4>("QK^T pattern 5 x K16 + K8 (88)", 12, grid);
    }
    return 0;
}
