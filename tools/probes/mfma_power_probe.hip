// What does the matrix pipe SUSTAIN on this board when nothing but MFMAs run, on random operands?
// §6 of DESIGN.md argues that every kernel which fills the chip levels off at 1.05 - 1.3 PF because of the board's power limit, not
// because of issue slots.  This probe measures the ceiling that argument implies, with no memory system in the way: every wave keeps
// its operand fragments in registers and issues back-to-back MFMAs for ~0.3 s per case (long enough for the power controller to
// settle; a short burst runs at the boost clock and says nothing).
//   shapes:   16x16x32 bf16 (the GEMM / conv kernels' form) and 32x32x16 bf16 (the attention kernel's form)
//   operands: N(0, 1) bf16 / all zero
//   waves:    2 or 4 per SIMD (the GEMM / conv / attention kernels run 2)
//   +lds:     the fragments are re-read from LDS every iteration at the 8-phase GEMM's ratio (12 ds_read_b128 per 32 MFMAs of
//             16x16x32), to price what the LDS -> register traffic adds
// Prints sustained TFLOP/s (dense, 2*M*N*K per MFMA) per case, and the fraction of the 2.5 PF nominal peak.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_power_probe tools/probes/mfma_power_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// SHAPE 0: 16x16x32, 4 A x 4 B fragments, 16 accumulators (16 MFMAs per iteration, 16384 flops each)
// SHAPE 1: 32x32x16, 2 A x 2 B fragments, 4 accumulators (4 MFMAs per iteration, 32768 flops each) - run twice per iteration
template <int SHAPE, bool LDS>
__global__ __launch_bounds__(512) void k(const bf16x8* __restrict__ src, float* out, int iters) {
    __shared__ bf16x8 sm[8 * 512];
    const int tid = threadIdx.x;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = src[((blockIdx.x & 255) * 8 + i) * 512 + tid];
        b[i] = src[((blockIdx.x & 255) * 8 + 4 + i) * 512 + tid];
        sm[i * 512 + tid] = a[i];
        sm[(4 + i) * 512 + tid] = b[i];
    }
    __syncthreads();
    f32x4 c4[4][4];
    f32x16 c16[2][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) c4[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) c16[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
        if constexpr (LDS) {  // 6 reads per 16 MFMAs of the 16x16x32 form (= 12 per 32); the offset moves so nothing is hoisted
            const int o = (it & 1) * 0;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                asm volatile("ds_read_b128 %0, %1" : "=v"(a[i]) : "v"((unsigned)((i * 512 + tid + o) * 16)) : "memory");
                asm volatile("ds_read_b128 %0, %1" : "=v"(b[i]) : "v"((unsigned)(((4 + i) * 512 + tid + o) * 16)) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if constexpr (SHAPE == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) c4[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], c4[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        c16[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i + 2 * rep], b[j + 2 * rep], c16[i][j], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += c4[i][j][0] + c4[i][j][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) s += c16[i][j][0] + c16[i][j][15];
    out[blockIdx.x * 512 + tid] = s;
}

static unsigned short f2bf(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
}

template <int SHAPE, bool LDS>
static void run(const char* name, const bf16x8* src, float* out, int waves_per_simd, const char* data) {
    const int grid = 256 * waves_per_simd / 2;  // groups of 8 waves = 2 per SIMD: one (2 waves / SIMD) or two (4) groups per CU
    const double flops_iter = 16.0 * 2 * 16 * 16 * 32;  // per wave per iteration (both shapes)
    const int iters = 40000;
    hipLaunchKernelGGL((k<SHAPE, LDS>), dim3(grid), dim3(512), 0, 0, src, out, 2000);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    // three bursts of ten back-to-back launches (~0.1 - 0.2 s each), the last burst is what is reported
    float best_ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int l = 0; l < 10; ++l) hipLaunchKernelGGL((k<SHAPE, LDS>), dim3(grid), dim3(512), 0, 0, src, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&best_ms, e0, e1);
    }
    const double tf = flops_iter * iters * 10 * (double)grid * 8 / (best_ms * 1e-3) / 1e12;
    printf("%-28s %-6s %d wave/SIMD on %3d CUs: %8.1f TFLOP/s  (%.3f of 2.5 PF)  %.1f ms\n", name, data, waves_per_simd,
           grid >= 256 ? 256 : grid, tf, tf / 2500.0, best_ms);
    fflush(stdout);
}

int main() {
    const size_t n = 256ull * 8 * 512 * 8;  // bf16 elements
    std::vector<unsigned short> h(n);
    srand(1234);
    for (size_t i = 0; i < n; ++i) {
        const float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = rand() / (float)RAND_MAX;
        h[i] = f2bf(sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2));
    }
    bf16x8 *rnd, *zero;
    float* out;
    hipMalloc(&rnd, n * 2);
    hipMalloc(&zero, n * 2);
    hipMalloc(&out, 512 * 512 * 4);
    hipMemcpy(rnd, h.data(), n * 2, hipMemcpyHostToDevice);
    hipMemset(zero, 0, n * 2);
    for (int w : {1, 2}) {
        run<0, false>("16x16x32 bf16", rnd, out, w * 2, "randn");
        run<0, false>("16x16x32 bf16", zero, out, w * 2, "zeros");
        run<1, false>("32x32x16 bf16", rnd, out, w * 2, "randn");
        run<1, false>("32x32x16 bf16", zero, out, w * 2, "zeros");
        run<0, true>("16x16x32 bf16 + LDS reads", rnd, out, w * 2, "randn");
        run<1, true>("32x32x16 bf16 + LDS reads", rnd, out, w * 2, "randn");
    }
    return 0;
}
