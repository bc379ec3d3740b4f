// HBM write bandwidth of different store patterns (tuning aid for the GEMM epilogue).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
// pattern 0: wave writes 1 KB contiguous per instruction (lane*16)
// pattern 1: GEMM epilogue: lane (li = lane&15, kg = lane>>4) writes 16 B at row li, byte 32*kg + 16*g  (g = 0,1 back to back)
// pattern 2: full 128-B lines: 8 lanes per row, 8 rows per instruction
template <int PAT>
__global__ __launch_bounds__(512) void store_kernel(char* out, int N /*row bytes*/, int rows_per_block) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntn = N / 512;  // 256-column bf16 tiles
    const int tile = blockIdx.x, mt = tile / ntn, nt = tile % ntn;
    char* base = out + (long long)mt * 256 * N + nt * 512;  // tile = 256 rows x 512 bytes
    const int wr = wave >> 2, wc = wave & 3;
    u32x4 v = {(unsigned)tid, 1u, 2u, 3u};
    if (PAT == 0) {
        // tile rows are 512 B: wave writes rows [wave*32, +32): 2 rows (1 KB) per instruction -> 512 B contiguous per row
        for (int i = 0; i < 16; ++i) {
            const int row = wave * 32 + i * 2 + (lane >> 5);
            *(u32x4*)(base + (long long)row * N + (lane & 31) * 16) = v;
        }
    } else if (PAT == 1) {
        const int li = lane & 15, kg = lane >> 4;
        for (int i = 0; i < 8; ++i)
            for (int g = 0; g < 2; ++g) {
                const int row = wr * 128 + i * 16 + li;
                *(u32x4*)(base + (long long)row * N + wc * 128 + kg * 32 + g * 16) = v;
            }
    } else {
        for (int i = 0; i < 16; ++i) {
            const int row = wr * 128 + i * 8 + (lane >> 3);
            *(u32x4*)(base + (long long)row * N + wc * 128 + (lane & 7) * 16) = v;
        }
    }
}
int main() {
    const int M = 32768, N = 6144 * 2;
    char* out; hipMalloc(&out, (size_t)M * N);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = (M / 256) * (N / 512);
    for (int pat = 0; pat < 4; ++pat) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a, 0);
            for (int it = 0; it < 10; ++it) {
                if (pat == 0) hipLaunchKernelGGL(store_kernel<0>, dim3(blocks), dim3(512), 0, 0, out, N, 256);
                if (pat == 1) hipLaunchKernelGGL(store_kernel<1>, dim3(blocks), dim3(512), 0, 0, out, N, 256);
                if (pat == 2) hipLaunchKernelGGL(store_kernel<2>, dim3(blocks), dim3(512), 0, 0, out, N, 256);
                if (pat == 3) hipMemsetAsync(out, 1, (size_t)M * N, 0);
            }
            hipEventRecord(b, 0); hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b);
        }
        printf("pattern %d: %.1f us per 403 MB  -> %.2f TB/s\n", pat, ms / 10 * 1e3, (double)M * N / (ms / 10 * 1e-3) / 1e12);
    }
    return 0;
}
