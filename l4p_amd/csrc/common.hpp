// Shared device helpers for the gfx950 (CDNA4) kernels of the L4P hot path.
// Everything here is wave64 / MFMA specific; there is no other target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#include "l4p_hip.h"  // dtype / error codes shared with the C ABI

// gfx950 hardware interaction (round 5, DESIGN.md "MFMA beside packed FP32"; reproducer tools/probes/mfma_valu_probe.py): while
// another wave of the same SIMD issues MFMAs, v_pk_{mul,add,fma}_f32 with op_sel[1] = 1 (the low result lane takes the HIGH dword
// of src1) computes its low result with src1 read as 0 in lanes 48..63.  The compiler picks that form by itself when it folds a
// swizzle into a packed multiply.  Files in which tools/check_isa.py finds the form are compiled without packed FP32 (NOPK_FILES
// in the Makefile); the link step runs the lint over the whole library.

// An MFMA operand fragment is always "8 consecutive k for one row":
// bf16 -> one 16x16x32 / 32x32x16 instruction, f32 -> 8 chained 16x16x4 / 32x32x2
// instructions that each consume one of the 8 elements.  The k-slot <-> element
// bijection is the same for both operands, so the contraction is exact either way.
template <typename T> struct Frag;
template <> struct Frag<bf16_t> { typedef bf16x8 type; };
template <> struct Frag<f16_t> { typedef f16x8 type; };
template <> struct Frag<float> { typedef f32x8 type; };
template <typename T> using vec8 = typename Frag<T>::type;  // 8 consecutive elements of a 16-bit engine type (one MFMA fragment)
// ... for code shared with the f32 engine whose 16-bit branch is dead there (a 16-byte placeholder keeps it compiling)
template <typename T> struct Half8 { typedef vec8<T> type; };
template <> struct Half8<float> { typedef bf16x8 type; };
template <typename T> using vec8h = typename Half8<T>::type;
template <typename T> struct Vec4T;
template <> struct Vec4T<bf16_t> { typedef bf16x4 type; };
template <> struct Vec4T<f16_t> { typedef f16x4 type; };
template <typename T> using vec4 = typename Vec4T<T>::type;
template <> struct Vec4T<float> { typedef bf16x4 type; };  // (placeholder: 16-bit branches that are dead in the f32 engine)
template <typename T> using vec4h = typename Vec4T<T>::type;
template <typename T> struct Elem16 { typedef T type; };
template <> struct Elem16<float> { typedef bf16_t type; };
template <typename T> using vec4e = typename Elem16<T>::type;  // element type of vec4h<T>

__device__ __forceinline__ f32x4 mma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma16(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mma32(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma16(f32x8 a, f32x8 b, f32x4 c) {
#pragma unroll
    for (int e = 0; e < 8; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], c, 0, 0, 0);
    return c;
}
__device__ __forceinline__ f32x16 mma32(bf16x8 a, bf16x8 b, f32x16 c) {
#ifdef ATTN_DBG_MFMA16  // (probe, WRONG results: every 32x32x16 MFMA issued as two 16x16x32 on the same operand registers - the
                        //  same matrix-pipe cycles, FLOPs and operand reads, a quarter of the accumulator registers per MFMA:
                        //  prices the MFMA shape's share of the attention kernel's power, tools/probes/ab_attn16.sh)
    const f32x4 q0 = {c[0], c[1], c[2], c[3]}, q1 = {c[4], c[5], c[6], c[7]}, q2 = {c[8], c[9], c[10], c[11]},
                q3 = {c[12], c[13], c[14], c[15]};
    const f32x4 n0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, q2, 0, 0, 0);
    const f32x4 n1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, q3, 0, 0, 0);  // (operands swapped: not the same value as n0)
    return (f32x16){n0[0], n0[1], n0[2], n0[3], n1[0], n1[1], n1[2], n1[3], q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
}
__device__ __forceinline__ f32x16 mma32(f32x8 a, f32x8 b, f32x16 c) {
#pragma unroll
    for (int e = 0; e < 8; ++e) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], c, 0, 0, 0);
    return c;
}

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float v) { return (f16_t)v; }
template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }

// exact (erf) GELU, matches torch.nn.GELU() default
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// The same GELU for the bf16 engine, branch free and ~3x fewer VALU slots than erff (which runs both of its divergent
// branches in every wave): erfc by Abramowitz & Stegun 7.1.26 (|abs error| <= 1.5e-7, far below the bf16 rounding of the
// operands and of the stored result), evaluated on the erfc side so the negative tail has no 1 - erf cancellation.
__device__ __forceinline__ float gelu_erf_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
    float poly = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    poly = __builtin_fmaf(poly, t, 1.421413741f);
    poly = __builtin_fmaf(poly, t, -0.284496736f);
    poly = __builtin_fmaf(poly, t, 0.254829592f);
    const float pe = poly * t * __builtin_amdgcn_exp2f(z * z * -1.4426950408889634f);  // erfc(|x| / sqrt 2)
    return 0.5f * x * (x >= 0.f ? 2.0f - pe : pe);
}
// Polynomial form for the bf16 engine, no transcendental: x * Phi(x) with Phi(x) - 1/2 = x P(x^2) on |x| <= 4.5 (P: degree-9
// weighted-minimax fit of (erf(x / sqrt 2) / 2) / x in x^2, fitted in float64; evaluated in float by Horner: |error| <= 8e-5
// absolute over all x, 2e-6 relative for x >= 4.5 where Phi is clamped to its value at 4.5) - 13 plain VALU against the
// erfc form's 16 + two quarter-rate ops (24 issue slots).  A bf16 ulp at |y| = 1 is 4e-3.
__device__ __forceinline__ float gelu_poly(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -4.5f, 4.5f);
    const float u = xc * xc;
    float p = -1.405532334e-12f;
    p = __builtin_fmaf(p, u, 1.702386847e-10f);
    p = __builtin_fmaf(p, u, -9.213366003e-09f);
    p = __builtin_fmaf(p, u, 2.962938120e-07f);
    p = __builtin_fmaf(p, u, -6.369978978e-06f);
    p = __builtin_fmaf(p, u, 9.790158587e-05f);
    p = __builtin_fmaf(p, u, -1.122762531e-03f);
    p = __builtin_fmaf(p, u, 9.833131509e-03f);
    p = __builtin_fmaf(p, u, -6.633633733e-02f);
    p = __builtin_fmaf(p, u, 3.988829162e-01f);
    return x * __builtin_fmaf(xc, p, 0.5f);
}
template <typename T> __device__ __forceinline__ float gelu_for(float x);  // GELU at the precision of engine dtype T
template <> __device__ __forceinline__ float gelu_for<float>(float x) { return gelu_erf(x); }
#ifdef L4P_GELU_ERFC  // (A/B aid: the erfc form; measured in one run: maskdot GEMM 1291 -> 1148 us, fc1 142.7 -> 136.6 us, c3 +0.8 %)
template <> __device__ __forceinline__ float gelu_for<bf16_t>(float x) { return gelu_erf_fast(x); }
#else
template <> __device__ __forceinline__ float gelu_for<bf16_t>(float x) { return gelu_poly(x); }
#endif
// (half has three more mantissa bits than bf16: the polynomial's 8e-5 absolute error would be a sixth of its ulp at 1; the
//  erfc form's 1.5e-7 is not seen)
template <> __device__ __forceinline__ float gelu_for<f16_t>(float x) { return gelu_erf_fast(x); }

// host side: engine dtype codes (include/l4p_hip.h)
static inline bool is16(int dtype) { return dtype == L4P_BF16 || dtype == L4P_F16; }
static inline int esize_of(int dtype) { return dtype == L4P_F32 ? 4 : 2; }
static inline bool dtype_ok(int dtype) { return dtype == L4P_BF16 || dtype == L4P_F32 || dtype == L4P_F16; }
// run `...` with T16 = the 16-bit engine type of `dtype` (bf16_t / f16_t); the caller has checked is16(dtype)
#define L4P_WITH_T16(dtype, T16, ...)      \
    do {                                   \
        if ((dtype) == L4P_F16) {          \
            typedef f16_t T16;             \
            __VA_ARGS__;                   \
        } else {                           \
            typedef bf16_t T16;            \
            __VA_ARGS__;                   \
        }                                  \
    } while (0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

#include "prof.hpp"

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) per (kernel instantiation, DEVICE), raised whenever a launch needs more than
// the largest size set so far: the attribute is a property of the function on one device, a host may drive several devices from
// several threads, and some kernels' LDS size depends on the geometry (a mini model's launch must not pin the limit below what
// the full-size model asks for later).  
#include <atomic>
#include <mutex>
struct lds_attr_state {
    std::atomic<int> set[64] = {};
    std::mutex mu;  // raising the limit is serialised, so a smaller request can never overwrite a larger one
};
template <class K>
static inline hipError_t lds_attr_once(lds_attr_state& st, K kern, int lds) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const int want = lds > 0 ? lds : 1;
    std::atomic<int>& cur = st.set[dev & 63];
    if (cur.load(std::memory_order_acquire) >= want) return hipSuccess;
    std::lock_guard<std::mutex> lock(st.mu);
    if (cur.load(std::memory_order_acquire) >= want) return hipSuccess;
    e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) cur.store(want, std::memory_order_release);
    return e;
}

// host-side error plumbing (defined in api.hip)
void l4p_set_error(const char* fmt, ...);
#define HIP_TRY(expr)                                                                  \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess) {                                                        \
            l4p_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return L4P_E_HIP;                                                          \
        }                                                                              \
    } while (0)
