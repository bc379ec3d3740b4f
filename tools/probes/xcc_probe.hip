// Which XCD does workgroup b run on?  (speed-only knowledge for the XCD-aware tile mappings)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int* out) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(x & 0xf);
}
int main() {
    for (int threads : {256, 512}) {
        const int n = 64;
        int* d;
        hipMalloc(&d, n * sizeof(int));
        hipLaunchKernelGGL(k, dim3(n), dim3(threads), 0, 0, d);
        std::vector<int> h(n);
        hipMemcpy(h.data(), d, n * sizeof(int), hipMemcpyDeviceToHost);
        printf("threads=%d:", threads);
        for (int i = 0; i < n; ++i) printf(" %d", h[i]);
        printf("\n");
        hipFree(d);
    }
    return 0;
}
