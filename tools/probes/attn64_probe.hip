// Standalone timing / tracing of the 64-rows-per-wave attention kernel (attention64.hip) with parts disabled (ATTN64_DBG_*) and,
// with -DATTN64_TRACE, the s_memtime stamps of wave 0 of workgroup 0 at the phase boundaries of every KV step.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I include -I l4p_amd/csrc [-DATTN64_...] tools/probes/attn64_probe.hip -o ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include <vector>
#include <cstring>
bool g_prof_on = false;
void prof_begin(int, hipStream_t, const char*) {}
void prof_end(int, hipStream_t) {}
void l4p_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); }
int knob(int) { return 1; }
#include "../../l4p_amd/csrc/attention64.hip"
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4, S = 2048, H = 16, Dh = 88;
    const bool zeros = argc > 2 && atoi(argv[2]);  // all-zero operands: the same instruction stream at the clock an idle matrix pipe is granted
    const size_t n = (size_t)B * S * H * 96;
    std::vector<unsigned short> h(n);
    unsigned st = 12345u;
    for (size_t i = 0; i < n; ++i) {  // roughly N(0,1) bf16 values
        float acc = -6.f;
        for (int k = 0; k < 12; ++k) {
            st = st * 1664525u + 1013904223u;
            acc += (st >> 8) * (1.0f / 16777216.0f);
        }
        unsigned u;
        memcpy(&u, &acc, 4);
        h[i] = zeros ? 0 : (unsigned short)((u + 0x8000u) >> 16);
    }
    void *q, *kt, *vt, *out;
    hipMalloc(&q, n * 2); hipMalloc(&kt, n * 2); hipMalloc(&vt, n * 2); hipMalloc(&out, n * 2);
    hipMemcpy(q, h.data(), n * 2, hipMemcpyHostToDevice);
    hipMemcpy(kt, h.data(), n * 2, hipMemcpyHostToDevice);
    hipMemcpy(vt, h.data(), n * 2, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) launch_attention64(L4P_BF16, q, kt, vt, out, B, S, H, Dh, 0.1066f, 0);
    hipEventRecord(a, 0);
    const int it = 50;
    for (int i = 0; i < it; ++i) launch_attention64(L4P_BF16, q, kt, vt, out, B, S, H, Dh, 0.1066f, 0);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("B=%d%s: %.2f us per launch\n", B, zeros ? " zeros" : "", ms / it * 1e3);
#ifdef ATTN64_TRACE
    std::vector<long long> t(4096);
    hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(attn64::g_trace), 4096 * 8);
    printf("tile 0: prologue %lld  loop %lld  store %lld | tile 1: prologue %lld loop %lld store %lld (cycles)\n", t[1] - t[0], t[2] - t[1],
           t[3] - t[2], t[5] - t[4], t[6] - t[5], t[7] - t[6]);
    long long sum[6] = {};
    for (int i = 2; i < 30; ++i) {  // steady-state steps of the LAST tile walked (stamps are overwritten tile after tile)
        const long long* s = &t[8 + i * 8];
        const long long d[6] = {s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[5] - s[4], t[8 + (i + 1) * 8] - s[0]};
        for (int k = 0; k < 6; ++k) sum[k] += d[k];
        if (i < 8) printf("step %2d: dma %lld  rare-check %lld  QK phase %lld  PV phase %lld  barrier %lld | step %lld\n", i, d[0], d[1], d[2], d[3], d[4], d[5]);
    }
    printf("mean of steps 2..29: dma %.0f  check %.0f  QK %.0f  PV %.0f  barrier %.0f | step %.0f (48 MFMAs = 1536 cycles)\n", sum[0] / 28.0, sum[1] / 28.0,
           sum[2] / 28.0, sum[3] / 28.0, sum[4] / 28.0, sum[5] / 28.0);
#endif
    return 0;
}
