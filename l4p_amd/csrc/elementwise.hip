// HBM-bound kernels of the encoder path: LayerNorm, tubelet im2col (patch embed gather), casts.
// All are one-pass, 16-byte vectorised, one wave per row where a row reduction is needed.
#include <cstdlib>
#include <type_traits>

#include "common.hpp"

// ---------------------------------------------------------------------------------------------
// LayerNorm over the last dim of a float [M][C] residual stream (reference: nn.LayerNorm eps=1e-6
// in the encoder, l4p_videomae.py:177; eps=1e-5 in the SAM two-way transformer, transformer.py:145-153;
// LayerNorm3d over channels, mask_decoder.py:145-157 == the same row LN in channels-last layout).
// Two-pass in registers (mean, then centred variance) like ATen.  out_T and/or out_f32 may be given.
// ---------------------------------------------------------------------------------------------
// XT: the input row is stored as T (the bf16 engine's activations) instead of float - half the bytes in; it may alias
// out_T (a wave holds its rows in registers before it stores anything).
// ROWS: rows per wave (all requested before the first is reduced).  Measured on the 1M x 352 LayerNorm + GELU of the tracker:
// 1 row 556 us, 2 rows 696 us, 4 rows 889 us - that launch is VALU-bound (GELU on 369 M elements), not latency-bound, and
// more rows only add register pressure; every launcher uses ROWS = 1.
// RES: the row normalised is x[row % x_mod] + delta[row] (delta in the engine dtype): the tracker's "keys += attention
// output; keys = LayerNorm(keys)" (sam/transformer.py:183-185) with the sum formed HERE instead of in the projection's
// epilogue - the float key stream is read once by this kernel instead of read + written by the GEMM and read again.
#ifdef LN_NT  // (probe: streaming hints on the LayerNorm's row loads / stores.  Measured, same call: the tracker's 369 MB-per-stream
              //  key LayerNorms 470 -> 449 us / 581 -> 582 us, the encoder's 8192-row ones 29.3 -> 31.9 us - their output is the
              //  next GEMM's operand and wants to stay in cache; c3 +-0.4 %: not adopted)
#define LN_ST(ptr, val) __builtin_nontemporal_store((val), (ptr))
#define LN_LD(ptr) __builtin_nontemporal_load(ptr)
#else
#define LN_ST(ptr, val) (*(ptr) = (val))
#define LN_LD(ptr) (*(ptr))
#endif
// y = (v - mean) * rstd * g + b, written ONCE: the chained key LayerNorm below re-derives another launch's float result from its
// inputs and stored statistics, bit for bit - both must round the same way whatever the compiler would contract in each context
__device__ __forceinline__ float ln_affine(float v, float mean, float rstd, float g, float b) {
    return __builtin_fmaf((v - mean) * rstd, g, b);
}
template <typename T, int MAXV, bool XT = false, int ROWS = 1, bool RES = false, bool PART = false>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, T* out_T, float* out_f32, int M,
                                                        int C, const float* __restrict__ add, int add_mod, T* out_T2, int act,
                                                        const T* __restrict__ delta = nullptr, int x_mod = 0,
                                                        const float* __restrict__ x_shared = nullptr, int x_period = 1,
                                                        int x_split = 0, float* out_sum = nullptr,
                                                        const float* __restrict__ part = nullptr, int nsplit = 0,
                                                        const float* __restrict__ pbias = nullptr, float* __restrict__ out_stats = nullptr) {
    // (x and the outputs are NOT restrict-qualified: the tracker normalises its key stream and the up-scaled activation in
    //  place; a wave has its whole row in registers - every store depends on the row statistics - before it writes)
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS;
    if (row0 >= M) return;
    const int nv = C >> 2;
    // every load of the rows is issued up front (x, gamma, beta, the optional addend): ONE exposed memory round trip per
    // wave instead of three dependent ones (x -> statistics -> gamma/beta/add) - the kernel is latency-, not bandwidth-bound
    f32x4 v[ROWS][MAXV], g[MAXV], bb[MAXV], av[ROWS][MAXV];
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int row = row0 + r < M ? row0 + r : M - 1;  // (a clamped duplicate row is loaded but never stored)
        // (RES, x_shared: rows p = row % x_period >= x_split come from the shared set x_shared[p - x_split] - the part of the
        //  tracker's key stream that is still the same for every track)
        const int prow = RES && x_shared ? row % x_period : 0;
        const f32x4* xr = RES && x_shared && prow >= x_split
                              ? (const f32x4*)(x_shared + (long long)(prow - x_split) * C)
                              : (const f32x4*)(x + (long long)(RES && x_mod > 0 ? row % x_mod : row) * C);
        const f32x4* ar = out_T2 ? (const f32x4*)(add + (long long)(row % add_mod) * C) : nullptr;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int idx = lane + i * 64;
            const bool in = idx < nv;
            if (XT && sizeof(T) == 2) {  // (branch-free: the load is clamped, not predicated, so all slots are requested at once)
                const vec4h<T> t = ((const vec4h<T>*)((const vec4e<T>*)x + (long long)row * C))[in ? idx : 0];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[r][i][k] = in ? (float)t[k] : 0.f;
            } else {
                v[r][i] = in ? LN_LD(xr + idx) : z;
            }
            if constexpr (RES) {
#ifdef LN_OLD_PART  // (A/B aid: the round-4 shape - one instantiation, run-time branch, one slice per round trip)
                if (part) {
                    f32x4 d = pbias ? ((const f32x4*)pbias)[in ? idx : 0] : z;
                    for (int sl = 0; sl < nsplit; ++sl) {
                        const f32x4 t = ((const f32x4*)(part + ((long long)sl * M + row) * C))[in ? idx : 0];
#pragma unroll
                        for (int k = 0; k < 4; ++k) d[k] += t[k];
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[r][i][k] += in ? d[k] : 0.f;
                } else
#endif
                if constexpr (PART) {  // the addend is bias + the float partials of a split-K projection, summed in slice order
                    f32x4 d = pbias ? ((const f32x4*)pbias)[in ? idx : 0] : z;
                    // (four slices requested at a time - a slice past the last is a clamped duplicate that is not added: with a plain
                    //  loop over a run-time nsplit every slice was its own memory round trip, 24 in a row per lane at batch 1)
                    for (int sl = 0; sl < nsplit; sl += 4) {
                        f32x4 t[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int su = sl + u < nsplit ? sl + u : sl;
                            t[u] = ((const f32x4*)(part + ((long long)su * M + row) * C))[in ? idx : 0];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (sl + u < nsplit) {
#pragma unroll
                                for (int k = 0; k < 4; ++k) d[k] += t[u][k];
                            }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[r][i][k] += in ? d[k] : 0.f;
                } else if constexpr (sizeof(T) == 2) {
                    const vec4h<T> t = LN_LD((const vec4h<T>*)((const vec4e<T>*)delta + (long long)row * C) + (in ? idx : 0));
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[r][i][k] += in ? (float)t[k] : 0.f;
                } else {
                    const f32x4 t = ((const f32x4*)((const float*)delta + (long long)row * C))[in ? idx : 0];
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[r][i][k] += in ? t[k] : 0.f;
                }
            }
            av[r][i] = (in && ar) ? ar[idx] : z;
        }
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + i * 64;
        const bool in = idx < nv;
        g[i] = in ? ((const f32x4*)gamma)[idx] : z;
        bb[i] = in ? ((const f32x4*)beta)[idx] : z;
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int row = row0 + r;
        if (row >= M) break;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) s += v[r][i][0] + v[r][i][1] + v[r][i][2] + v[r][i][3];
        const float mean = wave_sum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            if (lane + i * 64 < nv) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float d = v[r][i][k] - mean;
                    {  // (product rounded, then added - never contracted: the chained LayerNorm must round like the plain one in every
                       //  instantiation; left to the compiler the f32 kernels fused it and the bf16 ones did not)
#pragma clang fp contract(off)
                        q += d * d;
                    }
                }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
        if constexpr (RES) {  // (mean, rstd) of the row: what layernorm_chain_kernel re-derives this launch's float result from
            if (out_stats && lane == 0) {
                out_stats[2ll * row] = mean;
                out_stats[2ll * row + 1] = rstd;
            }
        }
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int idx = lane + i * 64;
            if (idx < nv) {
                f32x4 y;
#pragma unroll
                for (int k = 0; k < 4; ++k) y[k] = ln_affine(v[r][i][k], mean, rstd, g[i][k], bb[i][k]);
                if (act == L4P_ACT_GELU) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) y[k] = gelu_for<T>(y[k]);
                }
                if (out_T2) {  // T(y + add[row % add_mod]): the "+ positional / + prompt token" operand of the tracker
                    if (sizeof(T) == 2) {
                        vec4h<T> o;
#pragma unroll
                        for (int k = 0; k < 4; ++k) o[k] = (vec4e<T>)(y[k] + av[r][i][k]);
                        LN_ST((vec4h<T>*)((vec4e<T>*)out_T2 + (long long)row * C) + idx, o);
                    } else {
                        f32x4 o;
#pragma unroll
                        for (int k = 0; k < 4; ++k) o[k] = y[k] + av[r][i][k];
                        ((f32x4*)((float*)out_T2 + (long long)row * C))[idx] = o;
                    }
                }
                if (out_f32) LN_ST((f32x4*)(out_f32 + (long long)row * C) + idx, y);
                if constexpr (RES) {  // pre-norm residual stream (the encoder): the SUM x + delta is what lives on, y is only xn
                    if (out_sum) LN_ST((f32x4*)(out_sum + (long long)row * C) + idx, v[r][i]);
                }
                if (out_T) {
                    if (sizeof(T) == 2) {
                        vec4h<T> o;
#pragma unroll
                        for (int k = 0; k < 4; ++k) o[k] = (vec4e<T>)y[k];
                        LN_ST((vec4h<T>*)((vec4e<T>*)out_T + (long long)row * C) + idx, o);
                    } else {
                        ((f32x4*)((float*)out_T + (long long)row * C))[idx] = y;
                    }
                }
            }
        }
    }
}

template <typename T, int MAXV>
static void launch_ln_t(const float* x, const float* gamma, const float* beta, float eps, void* out_T, float* out_f32, int M,
                        int C, const float* add, int add_mod, void* out_T2, int act, hipStream_t stream) {
    hipLaunchKernelGGL((layernorm_kernel<T, MAXV>), dim3((M + 3) / 4), dim3(256), 0, stream, x, gamma, beta, eps, (T*)out_T, out_f32,
                       M, C, add, add_mod, (T*)out_T2, act);
}
template <typename T>
static void launch_ln(const float* x, const float* gamma, const float* beta, float eps, void* out_T, float* out_f32, int M, int C,
                      const float* add, int add_mod, void* out_T2, int act, hipStream_t stream) {
    // float4 slots per lane: 2 covers C <= 512 (DPT / tracker up-scaling), 6 the 1408-wide streams, 8 the 2048 limit
    if (C <= 512)
        launch_ln_t<T, 2>(x, gamma, beta, eps, out_T, out_f32, M, C, add, add_mod, out_T2, act, stream);
    else if (C <= 1536)
        launch_ln_t<T, 6>(x, gamma, beta, eps, out_T, out_f32, M, C, add, add_mod, out_T2, act, stream);
    else
        launch_ln_t<T, 8>(x, gamma, beta, eps, out_T, out_f32, M, C, add, add_mod, out_T2, act, stream);
}

int launch_layernorm_ex(int dtype, const float* x, const float* gamma, const float* beta, float eps, void* out_T,
                        float* out_f32, int M, int C, const float* add, int add_mod, void* out_T2, int act,
                        hipStream_t stream) {
    if (C % 4 || C > 2048 || (out_T2 && (!add || add_mod <= 0))) {
        l4p_set_error("layernorm: C=%d must be a multiple of 4 and <= 2048 (and out_T2 needs add/add_mod)", C);
        return L4P_E_INVALID;
    }
    ProfScope prof(PROF_LAYERNORM, stream, "M%d C%d add%d T2%d act%d f32%d", M, C, add != nullptr, out_T2 != nullptr, act, out_f32 != nullptr);
    if (is16(dtype)) L4P_WITH_T16(dtype, T16, launch_ln<T16>(x, gamma, beta, eps, out_T, out_f32, M, C, add, add_mod, out_T2, act, stream));
    else
        launch_ln<float>(x, gamma, beta, eps, out_T, out_f32, M, C, add, add_mod, out_T2, act, stream);
    HIP_TRY(hipGetLastError());
    return 0;
}

// (the token-ordered form of the first window's key LayerNorms: key_ln_tracks_kernel below)
static bool key_ln_tracks_fits(int M, int C, int x_mod, int add_mod, const void* add, const void* out_T, const void* out_T2);
template <typename T, bool CHAIN, bool XT = false>
static void launch_key_ln_tracks(const float* xs, int P, const void* dprev, const float* stats_in, const float* g0, const float* b0,
                                 const void* delta, const float* gamma, const float* beta, float eps, void* out_T, float* out_f32, int M, int C,
                                 const float* add, void* out_T2, float* out_stats, hipStream_t stream, const float* x_track = nullptr,
                                 const float* x_half = nullptr, int x_split = 0);

// y = LayerNorm(x[row % x_mod] + delta[row]) with the tracker's outputs (see layernorm_kernel RES); out_sum (may alias x when
// x_mod = 0): the float sum itself - the encoder's pre-norm residual stream, where y is only the next linear's input;
// part / nsplit / pbias: the addend is bias + nsplit float partials [nsplit][M][C] of a split-K projection instead of delta
int launch_layernorm_res(int dtype, const float* x, int x_mod, const void* delta_T, const float* gamma, const float* beta, float eps,
                         void* out_T, float* out_f32, int M, int C, const float* add, int add_mod, void* out_T2, const float* x_shared,
                         int x_period, int x_split, hipStream_t stream, float* out_sum, const float* part, int nsplit,
                         const float* pbias, float* out_stats) {
    if (part && (nsplit < 1 || nsplit > 16)) {
        l4p_set_error("layernorm_res: 1 <= nsplit <= 16 slices of float partials");
        return L4P_E_INVALID;
    }
    if (x_shared && (x_period <= 0 || x_split < 0 || x_split > x_period)) {
        l4p_set_error("layernorm_res: shared rows need 0 <= x_split <= x_period, x_period > 0");
        return L4P_E_INVALID;
    }
    if (!x_shared) x_period = 1, x_split = 0;
    if (C % 4 || C > 1536 || (!delta_T && !part) || (out_T2 && (!add || add_mod <= 0))) {
        l4p_set_error("layernorm_res: C=%d must be a multiple of 4 and <= 1536, delta or partials must be given (and out_T2 needs add/add_mod)", C);
        return L4P_E_INVALID;
    }
    ProfScope prof(PROF_LAYERNORM, stream, "M%d C%d res T2%d f32%d", M, C, out_T2 != nullptr, out_f32 != nullptr);
    if (!part && !out_sum && delta_T && key_ln_tracks_fits(M, C, x_mod, add_mod, add, out_T, out_T2) &&
        (x_mod > 0 ? !x_shared : (!x_shared || x_period == add_mod))) {
        if (x_mod > 0) {
            if (is16(dtype)) L4P_WITH_T16(dtype, T16, (launch_key_ln_tracks<T16, false>(x, add_mod, nullptr, nullptr, nullptr, nullptr, delta_T, gamma, beta, eps, out_T, out_f32, M, C, add, out_T2, out_stats, stream)));
            else
                launch_key_ln_tracks<float, false>(x, add_mod, nullptr, nullptr, nullptr, nullptr, delta_T, gamma, beta, eps, out_T, out_f32, M, C, add, out_T2, out_stats, stream);
        } else {  // the tracks' own float rows (possibly normalised in place), the common rows of a half-shared layer from x_shared
            if (is16(dtype)) L4P_WITH_T16(dtype, T16, (launch_key_ln_tracks<T16, false, true>(nullptr, add_mod, nullptr, nullptr, nullptr, nullptr, delta_T, gamma, beta, eps, out_T, out_f32, M, C, add, out_T2, out_stats, stream, x, x_shared, x_split)));
            else
                launch_key_ln_tracks<float, false, true>(nullptr, add_mod, nullptr, nullptr, nullptr, nullptr, delta_T, gamma, beta, eps, out_T, out_f32, M, C, add, out_T2, out_stats, stream, x, x_shared, x_split);
        }
        HIP_TRY(hipGetLastError());
        return 0;
    }
    const dim3 grid((M + 3) / 4);
#define LN_RES_LAUNCH(TT, MV, PT)                                                                                                          \
    hipLaunchKernelGGL((layernorm_kernel<TT, MV, false, 1, true, PT>), grid, dim3(256), 0, stream, x, gamma, beta, eps, (TT*)out_T, out_f32, M, \
                       C, add, add_mod, (TT*)out_T2, (int)L4P_ACT_NONE, (const TT*)delta_T, x_mod, x_shared, x_period, x_split, out_sum, part,  \
                       nsplit, pbias, out_stats)
    // (the split-K-partials form is its own instantiation: its four-slices-at-a-time requests double the register count, which the
    //  delta form - the encoder's and the tracker's rows at 8 waves per SIMD - must not pay)
#ifdef LN_OLD_PART
    const float* const part_sel = nullptr;
#else
    const float* const part_sel = part;
#endif
    if (is16(dtype)) L4P_WITH_T16(dtype, T16, {
        if (C <= 512) {
            if (part_sel) LN_RES_LAUNCH(T16, 2, true); else LN_RES_LAUNCH(T16, 2, false);
        } else {
            if (part_sel) LN_RES_LAUNCH(T16, 6, true); else LN_RES_LAUNCH(T16, 6, false);
        }
    }); else {
        if (C <= 512) {
            if (part_sel) LN_RES_LAUNCH(float, 2, true); else LN_RES_LAUNCH(float, 2, false);
        } else {
            if (part_sel) LN_RES_LAUNCH(float, 6, true); else LN_RES_LAUNCH(float, 6, false);
        }
    }
#undef LN_RES_LAUNCH
    HIP_TRY(hipGetLastError());
    return 0;
}

// Chained key LayerNorm (the tracker's second layer in a FIRST window, sam/transformer.py:183-185): the layer before normalised
// xs[row % x_mod] + dprev[row] - float rows common to all tracks plus that layer's update in the engine dtype - and stored only its
// engine-dtype outputs and (mean, rstd) per row.  Its float result, the residual this layer adds to, is re-derived here from those
// (ln_affine: bit for bit what it would have stored), so the float key master [N * P][C] is neither written nor read: 1.1 GB of
// 3.7 GB per 64 tracks.  One wave per row, every load issued before the first use, C <= 1536.
template <typename T>
__global__ __launch_bounds__(256) void layernorm_chain_kernel(const float* __restrict__ xs, int x_mod, const T* __restrict__ dprev,
                                                              const float* __restrict__ stats, const float* __restrict__ g0,
                                                              const float* __restrict__ b0, const T* __restrict__ delta,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                              T* __restrict__ out_T, float* __restrict__ out_f32, int M, int C,
                                                              const float* __restrict__ add, int add_mod, T* __restrict__ out_T2) {
    constexpr int MAXV = 6;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nv = C >> 2;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 v[MAXV], d1[MAXV], av[MAXV];
    const f32x4* xr = (const f32x4*)(xs + (long long)(row % x_mod) * C);
    const f32x4* ar = out_T2 ? (const f32x4*)(add + (long long)(row % add_mod) * C) : nullptr;
    const float mean0 = stats[2ll * row], rstd0 = stats[2ll * row + 1];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + i * 64;
        const bool in = idx < nv;
        v[i] = in ? xr[idx] : z;
        if (sizeof(T) == 2) {
            const vec4h<T> t = ((const vec4h<T>*)((const vec4e<T>*)dprev + (long long)row * C))[in ? idx : 0];
            const vec4h<T> u = ((const vec4h<T>*)((const vec4e<T>*)delta + (long long)row * C))[in ? idx : 0];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[i][k] += in ? (float)t[k] : 0.f;
                d1[i][k] = in ? (float)u[k] : 0.f;
            }
        } else {
            const f32x4 t = ((const f32x4*)((const float*)dprev + (long long)row * C))[in ? idx : 0];
            const f32x4 u = ((const f32x4*)((const float*)delta + (long long)row * C))[in ? idx : 0];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[i][k] += in ? t[k] : 0.f;
                d1[i][k] = in ? u[k] : 0.f;
            }
        }
        av[i] = (in && ar) ? ar[idx] : z;
    }
    // the previous layer's float result (its out_f32, had it stored one) + this layer's update
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + i * 64;
        if (idx < nv) {
            const f32x4 gg = ((const f32x4*)g0)[idx], bb = ((const f32x4*)b0)[idx];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[i][k] = ln_affine(v[i][k], mean0, rstd0, gg[k], bb[k]) + d1[i][k];
        } else {
            v[i] = z;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        if (lane + i * 64 < nv) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d = v[i][k] - mean;
                {  // (product rounded, then added - never contracted: the chained LayerNorm must round like the plain one in every
                       //  instantiation; left to the compiler the f32 kernels fused it and the bf16 ones did not)
#pragma clang fp contract(off)
                        q += d * d;
                    }
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + i * 64;
        if (idx < nv) {
            const f32x4 gg = ((const f32x4*)gamma)[idx], bb = ((const f32x4*)beta)[idx];
            f32x4 y;
#pragma unroll
            for (int k = 0; k < 4; ++k) y[k] = ln_affine(v[i][k], mean, rstd, gg[k], bb[k]);
            if (out_f32) ((f32x4*)(out_f32 + (long long)row * C))[idx] = y;
            if (sizeof(T) == 2) {
                vec4h<T> o, o2;
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = (vec4e<T>)y[k], o2[k] = (vec4e<T>)(y[k] + av[i][k]);
                if (out_T) ((vec4h<T>*)((vec4e<T>*)out_T + (long long)row * C))[idx] = o;
                if (out_T2) ((vec4h<T>*)((vec4e<T>*)out_T2 + (long long)row * C))[idx] = o2;
            } else {
                if (out_T) ((f32x4*)((float*)out_T + (long long)row * C))[idx] = y;
                if (out_T2) {
                    f32x4 o2;
#pragma unroll
                    for (int k = 0; k < 4; ++k) o2[k] = y[k] + av[i][k];
                    ((f32x4*)((float*)out_T2 + (long long)row * C))[idx] = o2;
                }
            }
        }
    }
}
// ---------------------------------------------------------------------------------------------
// The two key LayerNorms of a FIRST window again (layer 0: layernorm_kernel<RES> with every float row shared; layer 1:
// layernorm_chain_kernel), with the work laid out by TOKEN instead of by row (round 5).  In the row kernels a wave reads, beside
// the 5.6 - 11 KB that belong to its (track, token) row, the token's shared float key row and positional row (11 KB, from L2 /
// the Infinity Cache: the 2 x 11.5 MB sets do not fit the 4 MB L2) and two or four parameter vectors (11 - 22 KB through the
// vector L1): three to five times the unique bytes through the CU's 64 B/clk memory pipe.  tools/probes/stream_bw: the unique
// traffic alone streams in 183 / 250 us, with the two shared rows per row 301 / 371 us; the row kernels took 474 / 495 us.
// Here a wave owns ONE token p and walks tracks n0 .. n0 + nt - 1 of it (rows n * P + p): the shared rows are loaded ONCE into
// registers, the parameter vectors ONCE per workgroup into LDS, and the per-track operands of track n + 1 are requested before
// track n is reduced.  Per-row arithmetic is the row kernels' statement for statement (ln_affine, the uncontracted variance,
// the same summation order): bit-identical outputs (tests/test_track_gpu.py).
// ---------------------------------------------------------------------------------------------
// XT (later windows, layernorm_kernel<RES> with x_mod = 0): the float rows are the tracks' own key master x_track[row] (normalised
// in place when out_f32 aliases it) - or, for tokens p >= x_split of a half-shared layer 0, still the common rows x_half[p - x_split]:
// a wave-uniform choice, its token decides.  Per-track float rows travel with the track's update.
template <typename T, bool CHAIN, bool XT = false>
__global__ __launch_bounds__(256) void key_ln_tracks_kernel(const float* xs, int P, const T* __restrict__ dprev,
                                                            const float* __restrict__ stats_in, const float* __restrict__ g0,
                                                            const float* __restrict__ b0, const T* __restrict__ delta,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                            T* __restrict__ out_T, float* out_f32, int N, int C,
                                                            const float* __restrict__ add, T* __restrict__ out_T2,
                                                            float* __restrict__ out_stats, int tpw, const float* x_track = nullptr,
                                                            const float* __restrict__ x_half = nullptr, int x_split = 0) {
    static_assert(!(CHAIN && XT), "the chained form exists for first windows only");
    constexpr int MAXV = 6;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [gamma | beta | g0 | b0][C] floats
    f32x4* const sp = (f32x4*)smem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = C >> 2;
    const int ptiles = P >> 2;
    const int p = ((int)blockIdx.x % ptiles) * 4 + wave;
    const int n0 = ((int)blockIdx.x / ptiles) * tpw;
    const int n1 = n0 + tpw < N ? n0 + tpw : N;
    for (int i = threadIdx.x; i < nv; i += 256) {
        sp[i] = ((const f32x4*)gamma)[i];
        sp[nv + i] = ((const f32x4*)beta)[i];
        if constexpr (CHAIN) {
            sp[2 * nv + i] = ((const f32x4*)g0)[i];
            sp[3 * nv + i] = ((const f32x4*)b0)[i];
        }
    }
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 xv[MAXV], av[MAXV];
    typedef typename std::conditional<sizeof(T) == 2, vec4h<T>, f32x4>::type op_t;
    op_t dp[MAXV], dl[MAXV];
    float mean0 = 0.f, rstd0 = 0.f;
    const float* xsrc = XT ? (x_half && p >= x_split ? x_half + (long long)(p - x_split) * C : nullptr) : xs + (long long)p * C;
    const bool per_track = XT && xsrc == nullptr;  // (wave-uniform)
    const f32x4* xr = (const f32x4*)xsrc;
    const f32x4* ar = (const f32x4*)(add + (long long)p * C);
    auto request = [&](int n, op_t* dpv, op_t* dlv, float& m0, float& r0) {
        const long long row = (long long)n * P + p;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int idx = lane + i * 64;
            const int ci = idx < nv ? idx : 0;
            if constexpr (CHAIN) dpv[i] = ((const op_t*)(dprev + row * C))[ci];
            dlv[i] = ((const op_t*)(delta + row * C))[ci];
            if constexpr (XT) {
                if (per_track) xv[i] = idx < nv ? ((const f32x4*)(x_track + row * C))[idx] : z;
            }
        }
        if constexpr (CHAIN) m0 = stats_in[2 * row], r0 = stats_in[2 * row + 1];
    };
    request(n0, dp, dl, mean0, rstd0);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + i * 64;
        if (!per_track) xv[i] = idx < nv ? xr[idx] : z;
        av[i] = idx < nv ? ar[idx] : z;
    }
    __syncthreads();
    for (int n = n0; n < n1; ++n) {
        const long long row = (long long)n * P + p;
        f32x4 v[MAXV];
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int idx = lane + i * 64;
            const bool in = idx < nv;
            if constexpr (CHAIN) {
                // the previous layer's float result (x + its update, normalised with the stored statistics) + this layer's update
                f32x4 t = xv[i];
#pragma unroll
                for (int k = 0; k < 4; ++k) t[k] += in ? (float)dp[i][k] : 0.f;
                if (in) {
                    const f32x4 gg = sp[2 * nv + idx], bb = sp[3 * nv + idx];
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[i][k] = ln_affine(t[k], mean0, rstd0, gg[k], bb[k]) + (float)dl[i][k];
                } else {
                    v[i] = z;
                }
            } else {
                v[i] = xv[i];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[i][k] += in ? (float)dl[i][k] : 0.f;
            }
        }
        // (the operand registers are free once the row is formed: the next track's operands travel under this track's
        //  reductions and stores)
        if (n + 1 < n1) request(n + 1, dp, dl, mean0, rstd0);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        const float mean = wave_sum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            if (lane + i * 64 < nv) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float d = v[i][k] - mean;
                    {
#pragma clang fp contract(off)
                        q += d * d;
                    }
                }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
        if constexpr (!CHAIN) {
            if (out_stats && lane == 0) {
                out_stats[2 * row] = mean;
                out_stats[2 * row + 1] = rstd;
            }
        }
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int idx = lane + i * 64;
            if (idx < nv) {
                const f32x4 gg = sp[idx], bb = sp[nv + idx];
                f32x4 y;
#pragma unroll
                for (int k = 0; k < 4; ++k) y[k] = ln_affine(v[i][k], mean, rstd, gg[k], bb[k]);
                if (out_f32) ((f32x4*)(out_f32 + row * C))[idx] = y;
                if (sizeof(T) == 2) {
                    vec4h<T> o, o2;
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = (vec4e<T>)y[k], o2[k] = (vec4e<T>)(y[k] + av[i][k]);
                    ((vec4h<T>*)((vec4e<T>*)out_T + row * C))[idx] = o;
                    ((vec4h<T>*)((vec4e<T>*)out_T2 + row * C))[idx] = o2;
                } else {
                    ((f32x4*)((float*)out_T + row * C))[idx] = y;
                    f32x4 o2;
#pragma unroll
                    for (int k = 0; k < 4; ++k) o2[k] = y[k] + av[i][k];
                    ((f32x4*)((float*)out_T2 + row * C))[idx] = o2;
                }
            }
        }
    }
}
// the token-ordered form applies when every float row is shared with period P = add_mod, both T outputs are wanted and the
// rows are whole tracks; tracks per wave: 8 (4096 workgroups for 64 tracks of 2048 tokens)
static bool key_ln_tracks_fits(int M, int C, int x_mod, int add_mod, const void* add, const void* out_T, const void* out_T2) {
    // (x_mod = 0: the tracks' own float rows; else every float row is shared with the positional rows' period)
    return knob(KNOB_LN_TRACKS) && (x_mod == 0 || x_mod == add_mod) && add_mod > 0 && add_mod % 4 == 0 && M % add_mod == 0 && C % 4 == 0 &&
           C <= 1536 && add && out_T && out_T2;
}
template <typename T, bool CHAIN, bool XT>
static void launch_key_ln_tracks(const float* xs, int P, const void* dprev, const float* stats_in, const float* g0, const float* b0,
                                 const void* delta, const float* gamma, const float* beta, float eps, void* out_T, float* out_f32, int M, int C,
                                 const float* add, void* out_T2, float* out_stats, hipStream_t stream, const float* x_track, const float* x_half,
                                 int x_split) {
    const int N = M / P, tpw = N < 8 ? N : 8;  // (4 / 8 / 16 tracks per wave measure the same: c3 LayerNorm class 6.38 / 6.38 / 6.41 ms)
    const dim3 grid((P / 4) * ((N + tpw - 1) / tpw));
    hipLaunchKernelGGL((key_ln_tracks_kernel<T, CHAIN, XT>), grid, dim3(256), (size_t)(CHAIN ? 4 : 2) * C * sizeof(float), stream, xs, P, (const T*)dprev,
                       stats_in, g0, b0, (const T*)delta, gamma, beta, eps, (T*)out_T, out_f32, N, C, add, (T*)out_T2, out_stats, tpw, x_track,
                       x_half, x_split);
}

int launch_layernorm_chain(int dtype, const float* xs, int x_mod, const void* dprev_T, const float* stats, const float* g0, const float* b0,
                           const void* delta_T, const float* gamma, const float* beta, float eps, void* out_T, float* out_f32, int M, int C,
                           const float* add, int add_mod, void* out_T2, hipStream_t stream) {
    if (C % 4 || C > 1536 || x_mod < 1 || !xs || !dprev_T || !stats || !delta_T || (out_T2 && (!add || add_mod <= 0))) {
        l4p_set_error("layernorm_chain: C=%d must be a multiple of 4 and <= 1536; shared rows, both updates and the statistics are required", C);
        return L4P_E_INVALID;
    }
    ProfScope prof(PROF_LAYERNORM, stream, "M%d C%d chain T2%d f32%d", M, C, out_T2 != nullptr, out_f32 != nullptr);
    if (key_ln_tracks_fits(M, C, x_mod, add_mod, add, out_T, out_T2)) {
        if (is16(dtype)) L4P_WITH_T16(dtype, T16, (launch_key_ln_tracks<T16, true>(xs, x_mod, dprev_T, stats, g0, b0, delta_T, gamma, beta, eps, out_T, out_f32, M, C, add, out_T2, nullptr, stream)));
        else
            launch_key_ln_tracks<float, true>(xs, x_mod, dprev_T, stats, g0, b0, delta_T, gamma, beta, eps, out_T, out_f32, M, C, add, out_T2, nullptr, stream);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    const dim3 grid((M + 3) / 4);
    if (is16(dtype)) L4P_WITH_T16(dtype, T16, hipLaunchKernelGGL(layernorm_chain_kernel<T16>, grid, dim3(256), 0, stream, xs, x_mod, (const T16*)dprev_T, stats, g0, b0,
                           (const T16*)delta_T, gamma, beta, eps, (T16*)out_T, out_f32, M, C, add, add_mod, (T16*)out_T2));
    else
        hipLaunchKernelGGL(layernorm_chain_kernel<float>, grid, dim3(256), 0, stream, xs, x_mod, (const float*)dprev_T, stats, g0, b0,
                           (const float*)delta_T, gamma, beta, eps, (float*)out_T, out_f32, M, C, add, add_mod, (float*)out_T2);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm (+ GELU) of SHORT rows stored in the engine dtype, C <= 512 (round 5): the tracker's LayerNorm3d after its first
// up-scaling, [N * 8 P][352] in place.  One wave per row leaves 20 of 64 lanes idle in the second slot of a 352-wide row, pays
// two 6-step wave reductions per row and reads gamma / beta (2.8 KB) through the L1 for every 1.4 KB row.  Here a row belongs to
// ONE DPP ROW of 16 lanes (lane k holds the quads k, k + 16, ...: 88 quads = 5.5 slots, 92 % of the lanes busy), the statistics are
// two 4-step rotations inside the DPP row, a wave works on 4 rows at a time and walks RPW / 4 such groups with gamma / beta in
// registers, the next group's rows requested as soon as the current ones are converted.  Row sums are formed in a different
// order than layernorm_kernel's: equal to it to float rounding, not bit for bit.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float row16_sum(float v) {
    // rotations inside a row of 16 lanes: every lane ends with the row's sum (same order in every lane's view up to rotation)
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));  // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));  // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));  // row_ror:1
    return v;
}
template <typename T, int SLOTS, int RPW>
__global__ __launch_bounds__(256) void ln_rows16_kernel(const T* x, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                        T* out, long long M, int C, int act) {
    static_assert(sizeof(T) == 2 && RPW % 4 == 0, "16-bit rows, four at a time");
    const int lane = threadIdx.x & 63, k = lane & 15, sub = lane >> 4;
    const int nv = C >> 2;
    const long long row_base = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row_base >= M) return;
    f32x4 g[SLOTS], bb[SLOTS];
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) {
        const int q = k + 16 * j;
        g[j] = q < nv ? ((const f32x4*)gamma)[q] : z;
        bb[j] = q < nv ? ((const f32x4*)beta)[q] : z;
    }
    vec4h<T> raw[SLOTS];
    auto request = [&](long long row) {
        const long long rr = row < M ? row : M - 1;  // (a clamped duplicate row is loaded but never stored)
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) {
            const int q = k + 16 * j;
            raw[j] = ((const vec4h<T>*)((const vec4e<T>*)x + rr * C))[q < nv ? q : 0];
        }
    };
    request(row_base + sub);
    for (int it = 0; it < RPW / 4; ++it) {
        const long long row = row_base + 4 * it + sub;
        f32x4 v[SLOTS];
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) {
            const bool in = k + 16 * j < nv;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[j][e] = in ? (float)raw[j][e] : 0.f;
        }
        if (it + 1 < RPW / 4 && row_base + 4 * (it + 1) < M) request(row + 4);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) s += v[j][0] + v[j][1] + v[j][2] + v[j][3];
        const float mean = row16_sum(s) / (float)C;
        float qq = 0.f;
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) {
            if (k + 16 * j < nv) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = v[j][e] - mean;
                    qq += d * d;
                }
            }
        }
        const float rstd = rsqrtf(row16_sum(qq) / (float)C + eps);
        if (row < M) {
#pragma unroll
            for (int j = 0; j < SLOTS; ++j) {
                const int q = k + 16 * j;
                if (q < nv) {
                    vec4h<T> o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float y = ln_affine(v[j][e], mean, rstd, g[j][e], bb[j][e]);
                        if (act == L4P_ACT_GELU) y = gelu_for<T>(y);
                        o[e] = (vec4e<T>)y;
                    }
                    ((vec4h<T>*)((vec4e<T>*)out + row * C))[q] = o;
                }
            }
        }
    }
}

// LayerNorm of rows stored in the engine dtype (bf16 engine: bf16 in, bf16 out, possibly in place; f32 engine: the plain kernel)
int launch_layernorm_T(int dtype, const void* x_T, const float* gamma, const float* beta, float eps, void* out_T, int M, int C,
                       int act, hipStream_t stream) {
    if (!is16(dtype))
        return launch_layernorm_ex(dtype, (const float*)x_T, gamma, beta, eps, out_T, nullptr, M, C, nullptr, 0, nullptr, act, stream);
    if (C % 4 || C > 2048) {
        l4p_set_error("layernorm_T: C=%d must be a multiple of 4 and <= 2048", C);
        return L4P_E_INVALID;
    }
    ProfScope prof(PROF_LAYERNORM, stream, "M%d C%d inT act%d", M, C, act);
    if (knob(KNOB_LN_ROWS16) && C <= 512 && C % 4 == 0 && M >= 4096) {  // short rows, many of them: one DPP row of 16 lanes per row
        constexpr int RPW = 16;
        const dim3 grid16((unsigned)(((long long)M + 4 * RPW - 1) / (4 * RPW)));
        L4P_WITH_T16(dtype, T16, {
            if (C <= 384)
                hipLaunchKernelGGL((ln_rows16_kernel<T16, 6, RPW>), grid16, dim3(256), 0, stream, (const T16*)x_T, gamma, beta, eps, (T16*)out_T, (long long)M, C, act);
            else
                hipLaunchKernelGGL((ln_rows16_kernel<T16, 8, RPW>), grid16, dim3(256), 0, stream, (const T16*)x_T, gamma, beta, eps, (T16*)out_T, (long long)M, C, act);
        });
        HIP_TRY(hipGetLastError());
        return 0;
    }
    const dim3 grid((M + 3) / 4);
    const float* xf = (const float*)x_T;  // (re-typed inside the kernel)
    L4P_WITH_T16(dtype, T16, {
        if (C <= 512)
            hipLaunchKernelGGL((layernorm_kernel<T16, 2, true>), grid, dim3(256), 0, stream, xf, gamma, beta, eps, (T16*)out_T,
                               (float*)nullptr, M, C, (const float*)nullptr, 0, (T16*)nullptr, act);
        else if (C <= 1536)
            hipLaunchKernelGGL((layernorm_kernel<T16, 6, true>), grid, dim3(256), 0, stream, xf, gamma, beta, eps, (T16*)out_T,
                               (float*)nullptr, M, C, (const float*)nullptr, 0, (T16*)nullptr, act);
        else
            hipLaunchKernelGGL((layernorm_kernel<T16, 8, true>), grid, dim3(256), 0, stream, xf, gamma, beta, eps, (T16*)out_T,
                               (float*)nullptr, M, C, (const float*)nullptr, 0, (T16*)nullptr, act);
    });
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_layernorm(int dtype, const float* x, const float* gamma, const float* beta, float eps, void* out_T,
                     float* out_f32, int M, int C, hipStream_t stream) {
    return launch_layernorm_ex(dtype, x, gamma, beta, eps, out_T, out_f32, M, C, nullptr, 0, nullptr, L4P_ACT_NONE, stream);
}

// ---------------------------------------------------------------------------------------------
// float -> T cast (hook features handed to the DPT decoders), 16 bytes in per lane.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void cast_kernel(const float* __restrict__ x, T* __restrict__ y, long long n4) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const f32x4 v = ((const f32x4*)x)[i];
        if (sizeof(T) == 2) {
            vec4h<T> o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (vec4e<T>)v[k];
            ((vec4h<T>*)y)[i] = o;
        } else {
            ((f32x4*)y)[i] = v;
        }
    }
}

int launch_cast(int dtype, const float* x, void* y, long long n, hipStream_t stream) {
    if (n % 4) {
        l4p_set_error("cast: n must be a multiple of 4");
        return L4P_E_INVALID;
    }
    const long long n4 = n / 4;
    const int grid = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    ProfScope prof(PROF_ELEMENTWISE, stream, "cast");
    if (is16(dtype)) L4P_WITH_T16(dtype, T16, hipLaunchKernelGGL(cast_kernel<T16>, dim3(grid), dim3(256), 0, stream, x, (T16*)y, n4));
    else
        hipLaunchKernelGGL(cast_kernel<float>, dim3(grid), dim3(256), 0, stream, x, (float*)y, n4);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Tubelet gather for the patch-embed conv (reference PatchEmbed, modeling_finetune.py:269-283):
// Conv3d(3 -> C, kernel = stride = (pt, ph, pw)) == gather + GEMM.  Row = token (t', h', w')
// row-major, column k = ((c*pt + dt)*ph + dh)*pw + dw, zero-padded to Kp columns.
// rgb: [B][3][T][H][W] float.  One thread per output element; writes are fully coalesced, reads
// are pw-float runs that stay inside L2 (the whole clip is 9.6 MB).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void patch_gather_kernel(const float* __restrict__ rgb, T* __restrict__ out, int B, int Cin, int Tt, int Hh,
                                    int Ww, int pt, int ph, int pw, int Kp) {
    const int nT = Tt / pt, nH = Hh / ph, nW = Ww / pw;
    const int K = Cin * pt * ph * pw;
    const long long total = (long long)B * nT * nH * nW * Kp;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kp);
        long long tok = i / Kp;
        float v = 0.f;
        if (k < K) {
            const int dw = k % pw;
            int r = k / pw;
            const int dh = r % ph;
            r /= ph;
            const int dt = r % pt;
            const int c = r / pt;
            const int w = (int)(tok % nW);
            long long r2 = tok / nW;
            const int hh = (int)(r2 % nH);
            r2 /= nH;
            const int t = (int)(r2 % nT);
            const int b = (int)(r2 / nT);
            v = rgb[((((long long)b * Cin + c) * Tt + (t * pt + dt)) * Hh + (hh * ph + dh)) * Ww + (w * pw + dw)];
        }
        out[i] = from_f32<T>(v);
    }
}

int launch_patch_gather(int dtype, const float* rgb, void* out, int B, int Cin, int T, int H, int W, int pt, int ph,
                        int pw, int Kp, hipStream_t stream) {
    if (T % pt || H % ph || W % pw || Kp < Cin * pt * ph * pw) {
        l4p_set_error("patch_gather: bad geometry");
        return L4P_E_INVALID;
    }
    const long long total = (long long)B * (T / pt) * (H / ph) * (W / pw) * Kp;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    ProfScope prof(PROF_ELEMENTWISE, stream, "patch_gather");
    if (is16(dtype)) L4P_WITH_T16(dtype, T16, hipLaunchKernelGGL(patch_gather_kernel<T16>, dim3(grid), dim3(256), 0, stream, rgb, (T16*)out, B, Cin, T,
                           H, W, pt, ph, pw, Kp));
    else
        hipLaunchKernelGGL(patch_gather_kernel<float>, dim3(grid), dim3(256), 0, stream, rgb, (float*)out, B, Cin, T, H,
                           W, pt, ph, pw, Kp);
    HIP_TRY(hipGetLastError());
    return 0;
}
