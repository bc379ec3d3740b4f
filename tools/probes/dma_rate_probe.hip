// How fast can a CU pull L2-resident lines into LDS (LDS-DMA) or into registers?  Every workgroup re-reads the same 64 KB window of
// a 16 MB buffer (L2-hot after the first pass), 1 KB per wave instruction, full 128-byte lines; reports bytes / clock / CU.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/dma_rate_probe.hip -o build/dma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int MODE, int INFLIGHT>  // MODE 0: buffer_load .. lds, 1: global_load_lds, 2: global_load_dwordx4 to VGPRs (discarded)
__global__ __launch_bounds__(256) void k(const char* src, int iters, unsigned* sink, int ldsbytes_unused) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* base = src + (size_t)(blockIdx.x % 64) * 262144;  // 64 windows of 256 KB
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    const unsigned vo = (tid & 63) * 16 + wave * 1024;
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const unsigned so = ((it * 16 + u) & 15) * 4096;  // 16 x 4 KB = 64 KB window per workgroup
            if (MODE == 0)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(smem + u % 8 * 4096 + wave * 1024), 16, vo, so, 0, 0);
            else if (MODE == 1)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + so + vo),
                                                 (lptr_t)(smem + u % 8 * 4096 + wave * 1024), 16, 0, 0);
            else {
                const u32x4 v = *(const u32x4*)(base + so + vo);
                acc ^= v;
            }
            if (MODE != 2 && u % INFLIGHT == INFLIGHT - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc[0] == 0x12345678u) sink[0] = acc[1] + *(unsigned*)smem;
}
template <int MODE, int INFLIGHT>
static void run(const char* name, const char* src, unsigned* sink, int wg_per_cu, int lds) {
    auto kern = k<MODE, INFLIGHT>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int iters = 2000, grid = 256 * wg_per_cu;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, src, 50, sink, 0);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, src, iters, sink, 0);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)grid * iters * 16 * 4096;
    printf("%-44s wg/CU %d inflight %2d: %7.1f us  %6.2f TB/s  %5.1f B/clk/CU @2.1GHz\n", name, wg_per_cu, INFLIGHT, ms * 1e3, bytes / ms * 1e-9,
           bytes / (ms * 1e-3) / 256 / 2.1e9);
}
int main() {
    char* src; unsigned* sink;
    hipMalloc(&src, 64 * 262144 + 65536); hipMemset(src, 1, 64 * 262144 + 65536); hipMalloc(&sink, 64);
    run<0, 8>("buffer_load lds (L2-hot, full lines)", src, sink, 1, 150 * 1024);
    run<0, 8>("buffer_load lds", src, sink, 2, 64 * 1024);
    run<0, 16>("buffer_load lds", src, sink, 2, 64 * 1024);
    run<0, 4>("buffer_load lds", src, sink, 2, 64 * 1024);
    run<0, 8>("buffer_load lds", src, sink, 4, 32 * 1024);
    run<1, 8>("global_load_lds", src, sink, 1, 150 * 1024);
    run<1, 8>("global_load_lds", src, sink, 2, 64 * 1024);
    run<2, 8>("global_load_dwordx4 -> VGPR", src, sink, 1, 150 * 1024);
    run<2, 8>("global_load_dwordx4 -> VGPR", src, sink, 2, 64 * 1024);
    run<2, 8>("global_load_dwordx4 -> VGPR", src, sink, 8, 1024);
    return 0;
}
