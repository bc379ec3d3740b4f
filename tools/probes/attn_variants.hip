// Standalone timing of the attention kernel with parts disabled (see ATTN_DBG_* in attention.hip).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I include -I l4p_amd/csrc [-DATTN_DBG_NOLOAD|-DATTN_DBG_NOCOMPUTE] ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include <vector>
bool g_prof_on = false;
void prof_begin(int, hipStream_t, const char*) {}
void prof_end(int, hipStream_t) {}
void l4p_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); }
#include "../../l4p_amd/csrc/attention.hip"
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 1, S = 2048, H = 16, Dh = 88;
    const size_t n = (size_t)B * S * H * 96;
    std::vector<unsigned short> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = 0x3C00 + (unsigned short)((i * 2654435761u) >> 22) % 0x300;  // bf16 in [0.0078, ~1)
    void *q, *kt, *vt, *out;
    hipMalloc(&q, n * 2); hipMalloc(&kt, n * 2); hipMalloc(&vt, n * 2); hipMalloc(&out, n * 2);
    hipMemcpy(q, h.data(), n * 2, hipMemcpyHostToDevice);
    hipMemcpy(kt, h.data(), n * 2, hipMemcpyHostToDevice);
    hipMemcpy(vt, h.data(), n * 2, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) launch_attention(L4P_BF16, q, kt, vt, out, B, S, H, Dh, 0.1066f, 0);
    hipEventRecord(a, 0);
    const int it = 50;
    for (int i = 0; i < it; ++i) launch_attention(L4P_BF16, q, kt, vt, out, B, S, H, Dh, 0.1066f, 0);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("B=%d: %.2f us per launch\n", B, ms / it * 1e3);
    return 0;
}
