// MFMA GEMM / implicit-GEMM conv3d for gfx950.
//
//   out[m][n] = epilogue( sum_k A[m][k] * W[n][k] )        (both operands k-contiguous)
//
// One kernel template serves: the encoder linears (reference modeling_finetune.py:169-190, :62-69),
// the patch-embed GEMM (:276-283), every Conv3d 1x1x1 / 3x3x3 and ConvTranspose3d k==stride of the
// DPT decoders (dpt_block.py:29-90,93-157,255-278,406-414) and the SAM-style tracker's projections
// (sam/transformer.py:223-245, mask_decoder.py:58-66).
//
// CDNA4 mapping:
//  * operands are swapped into the MFMA (weights = A operand, activations = B operand) so that a lane
//    ends up with 4*TN *consecutive output columns of one row*: bias / residual / store are 16-byte
//    vector accesses and every output row is written in 64..256-byte runs.
//  * LDS tiles have 128-byte rows, XOR-swizzled at 16-byte granularity (chunk ^= (row>>1)&7) so the
//    non-contiguous 16-lane groups of ds_read_b128 hit 16 distinct slots.
//  * tiles are staged by LDS-DMA (global_load_lds, 16 B per lane, swizzle applied to the per-lane SOURCE chunk);
//    the register-staged loader remains for the conv path with a fused input ReLU.  Next tile's loads are issued
//    before the current tile's MFMAs (double-buffered LDS, one barrier per k-tile).
//  * T = bf16 uses v_mfma_f32_16x16x32_bf16, T = float uses 8x v_mfma_f32_16x16x4_f32 on the same
//    fragment registers (exact-f32 parity mode).
#pragma once
#include "common.hpp"

#include <type_traits>

template <int I, int N, class F>
__device__ __forceinline__ void static_for_(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_<I + 1, N>(f);
    }
}

enum { EPI_DENSE = L4P_EPI_DENSE, EPI_QKV = L4P_EPI_QKV, EPI_CONVT = L4P_EPI_CONVT, EPI_MASKDOT = L4P_EPI_MASKDOT };
enum { ACT_NONE = L4P_ACT_NONE, ACT_GELU = L4P_ACT_GELU, ACT_RELU = L4P_ACT_RELU };

// The kernel parameter block IS the public descriptor (include/l4p_hip.h): plain pointers and ints.
typedef l4p_gemm_desc GemmParams;

// 16 zero bytes in global memory: the source of LDS-DMA chunks that must read as zero (conv padding, k tail)
__device__ __attribute__((aligned(16))) static const unsigned g_zero_chunk[4] = {0u, 0u, 0u, 0u};

// ---- epilogue shared by every GEMM kernel: the lane (li, kg) owns rows m_wave0 + 16*i + li and the 4*TN consecutive
//      columns n_wave0 + 4*TN*kg + [0, 4*TN) of its wave's tile (acc[i][j][r] = column 4*TN*kg + 4*j + r) -------------
// One row (16*i + li) of the lane's tile: bias, activation, residuals, scatter / store.
template <typename T, int NV>
__device__ __forceinline__ void gemm_epilogue_row(const GemmParams& p, float (&v)[NV], const float (&bv)[NV],
                                                  const long long (&coff)[NV / 8], int m, int nb);

// L4P_EPI_MASKDOT epilogue (see include/l4p_hip.h): activated outputs x the three hyper-network vectors of the row's
// query, summed over the 32-column chunk the lane shares with 1 (NV = 16) or 3 (NV = 8) neighbours.  Kept apart from
// gemm_epilogue so that its 3 x NV hyper-vector registers never weigh on the ordinary epilogues.
// 16-bit engines, NV = 16 (the 8-phase / two-workgroup kernels' wave tile of 64 columns), GELU: the form the tracker runs a
// million rows through per clip.  Round 5: (1) the activation is compile time (the generic form tested p.act per VALUE: ~115
// scalar branches per 16-row block); (2) GELU is the clamped polynomial evaluated on PAIRS (v_pk_fma_f32: 7 issues per value
// against 13; the clamp is a bare v_med3_f32 - fmed3 through the compiler canonicalises its operand first, two more v_max per
// value); (3) the 3 x 64 dot products of a row with its query's hyper-network vectors are a matrix product and run on the matrix
// pipe: the activated row, rounded to T as the reference's autocast holds it (mask_decoder.py:136-139), IS an MFMA operand
// fragment - lane (row li, column group kg) holds 16 consecutive channels = two 8-element fragments under the k-slot map
// (kg, e) <-> channel 16 kg + e (+ 8) - and hyper^T [16 (3 used) x 64 channels] the other operand under the same map, so
// D = hyper^T x G^T leaves d0..d2 of row li in ONE lane (kg = 0) with no cross-lane sum.  The two 32-column chunks of the
// contract (include/l4p_hip.h) come from two products whose hyper operand is zeroed for the other chunk's lanes: 4 MFMAs
// 16x16x32 per 16 rows replace 48 FMAs + 6 lane exchanges per lane.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float med3_bare(float x, float lo, float hi) {
    float r;
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "s"(hi));
    return r;
}
// gelu_poly (common.hpp) on NP pairs at once, coefficient by coefficient: a pair's Horner chain is serially dependent and a dependent
// packed op needs a wait state (one pair at a time the compiler emitted an s_nop behind each of its 12 packed ops); NP independent
// chains interleaved fill those slots.  xc = x clamped to [-4.5, 4.5].
template <int NP>
__device__ __forceinline__ void gelu_poly2(f32x2_t (&x)[NP], const f32x2_t (&xc)[NP]) {
    f32x2_t u[NP], q[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) u[k] = xc[k] * xc[k];
#pragma unroll
    for (int k = 0; k < NP; ++k) q[k] = __builtin_elementwise_fma((f32x2_t){-1.405532334e-12f, -1.405532334e-12f}, u[k], (f32x2_t){1.702386847e-10f, 1.702386847e-10f});
    constexpr float cf[8] = {-9.213366003e-09f, 2.962938120e-07f, -6.369978978e-06f, 9.790158587e-05f, -1.122762531e-03f, 9.833131509e-03f,
                             -6.633633733e-02f, 3.988829162e-01f};
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int k = 0; k < NP; ++k) q[k] = __builtin_elementwise_fma(q[k], u[k], (f32x2_t){cf[c], cf[c]});
#pragma unroll
    for (int k = 0; k < NP; ++k) q[k] = __builtin_elementwise_fma(xc[k], q[k], (f32x2_t){0.5f, 0.5f});
#pragma unroll
    for (int k = 0; k < NP; ++k) x[k] = x[k] * q[k];
}
template <typename T, int TM>
__device__ __forceinline__ void gemm_epilogue_maskdot_mfma(const GemmParams& p, f32x4 (&acc)[TM][4], int m_wave0, int n_wave0, int li,
                                                           int kg) {
    static_assert(sizeof(T) == 2, "16-bit engines");
    if (n_wave0 >= p.N) return;  // (N % 64 == 0 is checked by the launcher: a wave's 64 columns are in or out together)
    const int nb = n_wave0 + 16 * kg;
    const int co = n_wave0 % p.Cout;  // first channel of the wave's 64 columns inside their tap (Cout % 64 == 0)
    f32x2_t bv[8];
#pragma unroll
    for (int c = 0; c < 16; c += 4) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 b4 = p.bias ? *(const f32x4*)(p.bias + nb + c) : z;
        bv[c / 2] = (f32x2_t){b4[0], b4[1]};
        bv[c / 2 + 1] = (f32x2_t){b4[2], b4[3]};
    }
    const float neg = -4.5f;
    int hq = -1;
    vec8<T> he[2], ho[2];  // hyper^T fragments (row h = li of hyper^T, channels co + 16 kg + 8 ks + e): chunk 0 (kg < 2) / chunk 1
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mb = m_wave0 + i * 16;  // (hyper_rows % 16 == 0, checked by the launcher: one query per 16-row block)
        const int qn = (mb < p.M ? mb : p.M - 1) / p.hyper_rows;
        if (qn != hq) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
                if (li < 3) {
                    const float* hp = p.hyper + ((long long)qn * 3 + li) * p.Cout + co + 16 * kg + 8 * ks;
                    a = *(const f32x4*)hp;
                    b = *(const f32x4*)(hp + 4);
                }
                vec8<T> f;
#pragma unroll
                for (int e = 0; e < 4; ++e) f[e] = (T)a[e], f[4 + e] = (T)b[e];
                const vec8<T> zf = {};
                he[ks] = kg < 2 ? f : zf;
                ho[ks] = kg < 2 ? zf : f;
            }
            hq = qn;
        }
        vec8<T> g[2];
#pragma unroll
        for (int jh = 0; jh < 2; ++jh) {  // 8 values = 4 pairs = one operand fragment at a time
            f32x2_t x[4], xc[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = 2 * jh + (k >> 1), h2 = k & 1;
                x[k] = (f32x2_t){acc[i][j][2 * h2], acc[i][j][2 * h2 + 1]} + bv[2 * j + h2];
                xc[k] = (f32x2_t){med3_bare(x[k][0], neg, 4.5f), med3_bare(x[k][1], neg, 4.5f)};
            }
            // (the polynomial's 8e-5 absolute error is below a bf16 ulp but a sixth of a half ulp at 1: the half engine evaluates its
            //  own GELU - gelu_for<f16_t>, the erfc form - as every other f16 kernel and the all-VALU form of this epilogue do)
            if constexpr (std::is_same<T, bf16_t>::value) {
                gelu_poly2<4>(x, xc);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) x[k] = (f32x2_t){gelu_for<T>(x[k][0]), gelu_for<T>(x[k][1])};
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) g[jh][2 * k] = (T)x[k][0], g[jh][2 * k + 1] = (T)x[k][1];
        }
        f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0;
        d0 = mma16(he[0], g[0], d0);
        d0 = mma16(he[1], g[1], d0);
        d1 = mma16(ho[0], g[0], d1);
        d1 = mma16(ho[1], g[1], d1);
        const int m = mb + li;
        if (kg == 0 && m < p.M) {  // D[h][row]: lane (column = row li, kg = 0) holds h = 0 .. 3
            float* op = p.out_f32 + (long long)(n_wave0 >> 5) * 3 * p.M + m;  // [chunk][i][m]: 16 consecutive rows per store
            op[0] = d0[0];
            op[(long long)p.M] = d0[1];
            op[2 * (long long)p.M] = d0[2];
            op[3 * (long long)p.M] = d1[0];
            op[4 * (long long)p.M] = d1[1];
            op[5 * (long long)p.M] = d1[2];
        }
    }
}

template <typename T, int TM, int TN>
__device__ __forceinline__ void gemm_epilogue_maskdot(const GemmParams& p, f32x4 (&acc)[TM][TN], int m_wave0, int n_wave0, int li,
                                                      int kg) {
    constexpr int NV = 4 * TN;
    if constexpr (sizeof(T) == 2 && TN == 4) {
        // (wave uniform; Cout % 64: a wave's 64 columns lie inside one tap; hyper_rows % 16: one query per 16-row block)
        if (p.act == ACT_GELU && p.Cout % 64 == 0 && p.N % 64 == 0 && p.hyper_rows % 16 == 0 && !(p.tuning & 4)) {
            gemm_epilogue_maskdot_mfma<T, TM>(p, acc, m_wave0, n_wave0, li, kg);
            return;
        }
    }
    const int nb = n_wave0 + NV * kg;
    if (nb >= p.N) return;  // (N % 32 == 0: the lanes sharing a chunk leave together)
    const int co = nb % p.Cout;
    float bv[NV], hv[3][NV];
#pragma unroll
    for (int c = 0; c < NV; c += 4) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 b4 = p.bias ? *(const f32x4*)(p.bias + nb + c) : z;
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[c + e] = b4[e];
    }
    int hq = -1;
    const bool writer = NV == 16 ? (kg & 1) == 0 : kg == 0;
    // fully unrolled over the rows (static accumulator indexing; rolled with a row switch it measured 8 % slower: 1123 vs 1032 us
    // on the c3 shape)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        float v[NV];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * j + r] = acc[i][j][r];
        const int m = m_wave0 + i * 16 + li;
        const bool ok = m < p.M;
        const int qn = (ok ? m : p.M - 1) / p.hyper_rows;
        if (qn != hq) {  // (a tile normally lies inside one query: loaded once)
#pragma unroll
            for (int h = 0; h < 3; ++h)
#pragma unroll
                for (int c = 0; c < NV; c += 4) {
                    const f32x4 h4 = *(const f32x4*)(p.hyper + ((long long)qn * 3 + h) * p.Cout + co + c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) hv[h][c + e] = h4[e];
                }
            hq = qn;
        }
        float d0 = 0.f, d1 = 0.f, d2 = 0.f;
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            float x = v[c] + bv[c];
            if (p.act == ACT_GELU)
                x = gelu_for<T>(x);
            else if (p.act == ACT_RELU)
                x = fmaxf(x, 0.f);
            d0 += x * hv[0][c];
            d1 += x * hv[1][c];
            d2 += x * hv[2][c];
        }
        // sum over the lane groups that share the 32-column chunk: v_permlane16_swap / v_permlane32_swap (gfx950 VALU lane
        // exchanges; [0] + [1] = own + partner in every lane, the same two addends as a shuffle) instead of __shfl_xor, which
        // is a ds_bpermute round trip through the LDS crossbar per value (24 dependent ones per wave and tile)
        auto xsum16 = [](float d) {
            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(d), __float_as_uint(d), false, false);
            return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        };
        auto xsum32 = [](float d) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(d), __float_as_uint(d), false, false);
            return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        };
        d0 = xsum16(d0);
        d1 = xsum16(d1);
        d2 = xsum16(d2);
        if (NV == 8) {
            d0 = xsum32(d0);
            d1 = xsum32(d1);
            d2 = xsum32(d2);
        }
        if (writer && ok) {
            float* op = p.out_f32 + (long long)(nb >> 5) * 3 * p.M + m;  // [chunk][i][m]: 16 consecutive rows per store
            op[0] = d0;
            op[(long long)p.M] = d1;
            op[2 * (long long)p.M] = d2;
        }
    }
}

// ROLLED = false: the row loop is fully unrolled (registers die row by row: 120 VGPRs for the 128x128 kernel, which
// keeps 2-3 workgroups per CU).  ROLLED = true (gemm8p.hpp, alone on its CU with registers to spare): the row body -
// ~1k instructions with every epilogue flavour and the integer divisions of the scatter paths - exists ONCE; TM
// unrolled copies overflow the instruction cache (measured: 15 us per 256x256 tile) and nothing hides those misses.
template <typename T, int TM, int TN, bool ROLLED = false>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[TM][TN], int m_wave0, int n_wave0, int li,
                                              int kg) {
    constexpr int ES = sizeof(T);
    constexpr int NV = 4 * TN;
    const int nb = n_wave0 + NV * kg;
    // the lane's bias values are row independent: fetch them once, ahead of the row loop (a load inside the loop
    // costs one exposed L2 round trip per row when the workgroup is alone on its CU)
    float bv[NV];
#pragma unroll
    for (int g8 = 0; g8 < NV; g8 += 8) {
        const bool ok = p.bias && nb + g8 < p.N;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 b0 = ok ? *(const f32x4*)(p.bias + nb + g8) : z, b1 = ok ? *(const f32x4*)(p.bias + nb + g8 + 4) : z;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bv[g8 + q] = b0[q];
            bv[g8 + 4 + q] = b1[q];
        }
    }
    // column part of the output offset (row independent: the ConvTranspose tap decomposition costs five integer
    // divisions, which used to be repeated for every row)
    long long coff[NV / 8];
#pragma unroll
    for (int g8 = 0; g8 < NV; g8 += 8) {
        const int n = nb + g8;
        if (p.epi == EPI_CONVT) {
            const int tap = n / p.Cout, co = n - tap * p.Cout;
            const int dw = tap % p.kw, dh = (tap / p.kw) % p.kh, dt = tap / (p.kw * p.kh);
            coff[g8 / 8] = (((long long)dt * (p.Hi * p.kh) + dh) * (p.Wi * p.kw) + dw) * p.Cout + co;
        } else {
            coff[g8 / 8] = n;
        }
    }
    if (ROLLED) {
#pragma nounroll
        for (int i = 0; i < TM; ++i) {
            float v[NV];
#define L4P_ACC_ROW(I)                                                                                   \
    case I:                                                                                              \
        if (I < TM) {                                                                                    \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                               \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) v[4 * j + r] = acc[I < TM ? I : 0][j][r]; \
        }                                                                                                \
        break;
            switch (i) {  // wave-uniform: keeps the accumulators statically indexed
                L4P_ACC_ROW(0) L4P_ACC_ROW(1) L4P_ACC_ROW(2) L4P_ACC_ROW(3) L4P_ACC_ROW(4) L4P_ACC_ROW(5) L4P_ACC_ROW(6) L4P_ACC_ROW(7)
            }
#undef L4P_ACC_ROW
            static_assert(TM <= 8, "extend the accumulator row switch");
            const int m = m_wave0 + i * 16 + li;
            if (m < p.M) gemm_epilogue_row<T, NV>(p, v, bv, coff, m, nb);
        }
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float v[NV];
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[4 * j + r] = acc[i][j][r];
            const int m = m_wave0 + i * 16 + li;
            if (m < p.M) gemm_epilogue_row<T, NV>(p, v, bv, coff, m, nb);
        }
    }
}

template <typename T, int NV>
__device__ __forceinline__ void gemm_epilogue_row(const GemmParams& p, float (&v)[NV], const float (&bv)[NV],
                                                  const long long (&coff)[NV / 8], int m, int nb) {
    constexpr int ES = sizeof(T);
    {
        // float residuals of the whole row segment, loaded before any of its outputs is stored (in-place updates alias)
        float rv[NV], rv2[NV];
        const bool res32 = p.res1 && p.res_f32;
        if (res32) {
            const long long rrow = (long long)(p.res_mod > 0 ? (m % p.res_mod) : m) * p.ldr;
#pragma unroll
            for (int g8 = 0; g8 < NV; g8 += 8) {
                const int n = nb + g8;
                f32x4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = r0, s0 = r0, s1 = r0;
                if (n < p.N) {
                    r0 = *(const f32x4*)((const float*)p.res1 + rrow + n);
                    r1 = *(const f32x4*)((const float*)p.res1 + rrow + n + 4);
                    if (p.res2) {
                        s0 = *(const f32x4*)((const float*)p.res2 + rrow + n);
                        s1 = *(const f32x4*)((const float*)p.res2 + rrow + n + 4);
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    rv[g8 + q] = r0[q];
                    rv[g8 + 4 + q] = r1[q];
                    rv2[g8 + q] = s0[q];
                    rv2[g8 + 4 + q] = s1[q];
                }
            }
        }
        // row part of the output offset
        long long rowoff;
        if (p.epi == EPI_CONVT) {
            int wi, hi, ti, b;
            if (((p.Wi & (p.Wi - 1)) | (p.Hi & (p.Hi - 1)) | (p.Ti & (p.Ti - 1))) == 0) {  // power-of-two grid: shifts
                const int sw = __builtin_ctz(p.Wi), sh = __builtin_ctz(p.Hi), st = __builtin_ctz(p.Ti);
                wi = m & (p.Wi - 1);
                hi = (m >> sw) & (p.Hi - 1);
                ti = (m >> (sw + sh)) & (p.Ti - 1);
                b = m >> (sw + sh + st);
            } else {
                wi = m % p.Wi;
                int r = m / p.Wi;
                hi = r % p.Hi;
                r /= p.Hi;
                ti = r % p.Ti;
                b = r / p.Ti;
            }
            rowoff = ((((long long)b * p.Ti + ti) * p.kt * (p.Hi * p.kh) + hi * p.kh) * (p.Wi * p.kw) + wi * p.kw) * p.Cout;
        } else {
            const long long pm = p.c_gr > 0 ? (long long)(m / p.c_gr) * p.c_gs + p.c_go + (m % p.c_gr) : m;
            rowoff = pm * p.ldc;
        }
#pragma unroll
        for (int g8 = 0; g8 < NV; g8 += 8) {
            const int n = nb + g8;
            if (n >= p.N) continue;  // N % 8 == 0 is required
            float* vv = v + g8;
#pragma unroll
            for (int q = 0; q < 8; ++q) vv[q] += bv[g8 + q];
            if (p.act == ACT_GELU) {
#pragma unroll
                for (int q = 0; q < 8; ++q) vv[q] = gelu_for<T>(vv[q]);
            } else if (p.act == ACT_RELU) {
#pragma unroll
                for (int q = 0; q < 8; ++q) vv[q] = fmaxf(vv[q], 0.0f);
            }
            if (p.epi == EPI_QKV && n < p.H * p.Dp && p.q_scale != 0.f) {  // q = q * scale (modeling_finetune.py:180)
#pragma unroll
                for (int q = 0; q < 8; ++q) vv[q] *= p.q_scale;
            }
            const long long off = rowoff + coff[g8 / 8];  // element offset of the 8 outputs
            if (res32) {
#pragma unroll
                for (int q = 0; q < 8; ++q) vv[q] += rv[g8 + q];
                if (p.res2) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) vv[q] += rv2[g8 + q];
                }
            } else if (p.res1) {
                const long long roff = (long long)(p.res_mod > 0 ? (m % p.res_mod) : m) * p.ldr + n;
                const T* rp = (const T*)p.res1 + roff;
#pragma unroll
                for (int q = 0; q < 8; ++q) vv[q] += to_f32<T>(rp[q]);
                if (p.res2) {
                    const T* sp = (const T*)p.res2 + roff;
#pragma unroll
                    for (int q = 0; q < 8; ++q) vv[q] += to_f32<T>(sp[q]);
                }
            }
            if (p.epi == EPI_QKV && n >= p.H * p.Dp && n < 2 * p.H * p.Dp) {
                // K, stored in the attention kernel's LDS tile order (attention.hip): 8-element groups
                // [b][h][kv block][k-step][key][half ^ ((key >> 3) & 1)], KVB keys per block
                constexpr int KVB = 128 / ES;
                const int nk = n - p.H * p.Dp;
                const int h = nk / p.Dp, d0 = nk - h * p.Dp;
                const int ks = d0 >> 4, half = (d0 >> 3) & 1;
                const int b = m / p.S, s = m - b * p.S;
                const int kb = s / KVB, key = s - kb * KVB;
                const long long g8 =
                    ((((long long)(b * p.H + h) * (p.S / KVB) + kb) * (p.Dp / 16) + ks) * KVB + key) * 2 + (half ^ ((key >> 3) & 1));
                T* kp = (T*)p.k_tiled + g8 * 8;
                if (ES == 2) {
                    vec8h<T> o;
#pragma unroll
                    for (int q = 0; q < 8; ++q) o[q] = (T)vv[q];
                    *(vec8h<T>*)kp = o;
                } else {
                    *(f32x4*)kp = (f32x4){vv[0], vv[1], vv[2], vv[3]};
                    *(f32x4*)((float*)kp + 4) = (f32x4){vv[4], vv[5], vv[6], vv[7]};
                }
                continue;
            }
            if (p.epi == EPI_QKV && n >= 2 * p.H * p.Dp) {
                // V, stored transposed: vt[((b*H + h)*Dp + d)*S + s]
                const int nv = n - 2 * p.H * p.Dp;
                const int h = nv / p.Dp, d = nv - h * p.Dp;
                const int b = m / p.S, s = m - b * p.S;
                T* vp = (T*)p.vt + ((long long)(b * p.H + h) * p.Dp + d) * p.S + s;
#pragma unroll
                for (int q = 0; q < 8; ++q) vp[(long long)q * p.S] = from_f32<T>(vv[q]);
                continue;
            }
            if (p.out_f32) {
                float* op = p.out_f32 + off;
                *(f32x4*)op = (f32x4){vv[0], vv[1], vv[2], vv[3]};
                *(f32x4*)(op + 4) = (f32x4){vv[4], vv[5], vv[6], vv[7]};
            }
            if (p.out_relu_T) {  // second output: relu(v) as T (pre-activated input of the next ResidualConvUnit conv)
                T* op = (T*)p.out_relu_T + off;
                if (ES == 2) {
                    vec8h<T> o;
#pragma unroll
                    for (int q = 0; q < 8; ++q) o[q] = (T)fmaxf(vv[q], 0.f);
                    *(vec8h<T>*)op = o;
                } else {
                    *(f32x4*)op = (f32x4){fmaxf(vv[0], 0.f), fmaxf(vv[1], 0.f), fmaxf(vv[2], 0.f), fmaxf(vv[3], 0.f)};
                    *(f32x4*)((float*)op + 4) = (f32x4){fmaxf(vv[4], 0.f), fmaxf(vv[5], 0.f), fmaxf(vv[6], 0.f), fmaxf(vv[7], 0.f)};
                }
            }
            if (p.out_T) {
                T* op = (T*)p.out_T + off;
                if (ES == 2) {
                    vec8h<T> o;
#pragma unroll
                    for (int q = 0; q < 8; ++q) o[q] = (T)vv[q];
#ifdef GEMM_DBG_NOSTORE  // (tools/probes: epilogue arithmetic without the output traffic)
                    asm volatile("" ::"v"(o), "v"(op));
#else
                    *(vec8h<T>*)op = o;
#endif
                } else {
                    *(f32x4*)op = (f32x4){vv[0], vv[1], vv[2], vv[3]};
                    *(f32x4*)((float*)op + 4) = (f32x4){vv[4], vv[5], vv[6], vv[7]};
                }
            }
        }
    }
}

// Tile walk of the implicit-GEMM conv.  Every input voxel feeds 27 taps; with workgroups handed out in plain row order
// over eight XCDs the three t-planes a tile touches are 2 x (Ho*Wo*Cin) bytes apart and each XCD's 4 MB L2 re-fetched
// them from HBM (PMC: 7.9 GB read per launch of the 16x224x224x128 conv against 0.8 GB of input).  Here an XCD owns a
// contiguous range of tiles (the caller's XCD remap) and position n of the walk is the m-tile
//   (b, band, t, i):  i fastest inside a band of BT tiles (a few image rows), then t, then the next band
// so the rows of planes t-1 / t / t+1 that a band needs are still in L2 when the walk moves to t+1.
#ifdef GEMM_PROBE_VARIANTS
__device__ int g_conv_bt_override = 0;  // (tools/probes: band size experiments)
#endif
__device__ __forceinline__ int conv_tile_walk(const GemmParams& p, int n, int BM, int es) {
    const int P = p.Ho * p.Wo;  // output voxels per (b, t) plane
    if (P % BM) return n;
    const int TP = P / BM;
    const long long tile_bytes = (long long)BM * p.sh * p.sw * p.Cin * es;
    int bt = (int)((2ll << 20) / 3 / (tile_bytes > 0 ? tile_bytes : 1));
#ifdef GEMM_PROBE_VARIANTS
    if (g_conv_bt_override > 0) bt = g_conv_bt_override;
#endif
    if (bt > TP) bt = TP;
    if (bt < 1) bt = 1;
    while (TP % bt) --bt;  // largest divisor of TP that keeps three planes of a band within ~2 MB
    const int i = n % bt;
    int r = n / bt;
    const int t = r % p.To;
    r /= p.To;
    const int band = r % (TP / bt), b = r / (TP / bt);
    return (b * p.To + t) * TP + band * bt + i;
}

// Lean epilogue for the plain dense family (L4P_EPI_DENSE, no row maps, no broadcast residual): what the large GEMMs and
// convs of the 8-phase kernel mostly run.  The generic row body above decides every flavour (scatter forms, row maps,
// residual kinds, activations) at run time inside the row loop - ~600 issued instructions per row with SGPR spills,
// measured at 9 us per 256 x 256 tile for a bias + bf16 store (27 of fc1's 157 us at batch 4).  Here activation and
// residual kind are template parameters, the loop body is a few dozen instructions, and the residual of row i + 1 is
// requested before row i is finished (one exposed round trip per tile instead of one per row).
// RES: 0 none, 1 float (p.res1 [+ p.res2]), 2 T.
template <int I, int N, class F>
__device__ __forceinline__ void epi_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        epi_static_for<I + 1, N>(f);
    }
}

// logical row m -> physical row of the output / residual: the descriptor's optional one-level row map (l4p_gemm_desc.c_*)
struct EpiRowMapDesc {
    const GemmParams& p;
    __device__ __forceinline__ long long operator()(int m) const {
        return p.c_gr > 0 ? (long long)(m / p.c_gr) * p.c_gs + p.c_go + (m % p.c_gr) : m;
    }
};

// RowMap: any callable int -> long long (conv3_halo.hpp passes the voxel-tile map of its 3-D output tiles)
template <typename T, int TM, int TN, int ACT, int RES, class RowMap>
__device__ __forceinline__ void gemm_epilogue_dense(const GemmParams& p, f32x4 (&acc)[TM][TN], int m_wave0, int n_wave0, int li,
                                                    int kg, const RowMap& crow) {
    constexpr int ES = sizeof(T);
    constexpr int NV = 4 * TN, NG = NV / 8;
    const int nb = n_wave0 + NV * kg;
    bool gok[NG];
    float bv[NV];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        gok[g] = nb + 8 * g < p.N;  // N % 8 == 0
        const bool ok = p.bias && gok[g];
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 b0 = ok ? *(const f32x4*)(p.bias + nb + 8 * g) : z, b1 = ok ? *(const f32x4*)(p.bias + nb + 8 * g + 4) : z;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bv[8 * g + q] = b0[q];
            bv[8 * g + 4 + q] = b1[q];
        }
    }
    const bool two = RES == 2 && p.res2 != nullptr;  // (a second residual only exists in the engine dtype: the DPT fusion convs)
    // Residual registers: a ring of RD rows (float: NV values per residual and row; T: NV / 2 dwords).  The rows of a lane are
    // 16 apart, every row is its own HBM round trip, and nothing else runs on the CU while its workgroup is in the epilogue: with
    // one row of look-ahead the 8 rows of a wave were 8 dependent round trips (the tall tracker GEMM with a float residual
    // in + out spent more time here than in its main loop).  RD rows are requested before the first one is needed.
#ifndef GEMM_EPI_RES_DEPTH
#define GEMM_EPI_RES_DEPTH 4
#endif
    constexpr int RD = RES == 0 ? 1 : RES == 2 ? 2 : (GEMM_EPI_RES_DEPTH < TM ? GEMM_EPI_RES_DEPTH : TM);
    f32x4 rf[RD][RES == 1 ? 2 * NG : 1][1];
    u32x4 rt[RD][RES == 2 ? NG : 1][2];
    // (row map of the output and the residual: logical row m lives at physical row crow(m))
    auto load_res = [&](int m, auto slot_) {
        constexpr int sl = decltype(slot_)::value;
        const long long roff = crow(m) * p.ldr + nb;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (!gok[g]) continue;
            if (RES == 1) {
                rf[sl][2 * g][0] = *(const f32x4*)((const float*)p.res1 + roff + 8 * g);
                rf[sl][2 * g + 1][0] = *(const f32x4*)((const float*)p.res1 + roff + 8 * g + 4);
            } else if (RES == 2) {
                rt[sl][g][0] = *(const u32x4*)((const T*)p.res1 + roff + 8 * g);
                if (two) rt[sl][g][1] = *(const u32x4*)((const T*)p.res2 + roff + 8 * g);
            }
        }
    };
    if (RES != 0) {
        epi_static_for<0, RD>([&](auto d_) {
            constexpr int d = decltype(d_)::value;
            if (m_wave0 + d * 16 + li < p.M) load_res(m_wave0 + d * 16 + li, d_);
        });
    }
    // fully unrolled over the TM rows: the body is lean enough for that (~40 instructions a row), and a rolled loop makes
    // hipcc index the accumulator rows through scratch memory (one round trip per row: measured slower than the generic body)
    epi_static_for<0, TM>([&](auto i_) {
        constexpr int i = decltype(i_)::value, sl = i % RD;
        float v[NV];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * j + r] = acc[i][j][r];
        const int m = m_wave0 + i * 16 + li;
        if (m >= p.M) return;  // (rows only run out at the bottom of the matrix: nothing after this row either)
        if constexpr (ACT == ACT_GELU && std::is_same<T, bf16_t>::value && NV % 8 == 0) {
            // gelu_poly on four interleaved pairs (gelu_poly2: same operations per element, bit for bit; one pair at a time the
            // compiler put a wait state behind every packed op of the serial Horner chain)
#pragma unroll
            for (int c = 0; c < NV; c += 8) {
                f32x2_t x[4], xc[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    x[k] = (f32x2_t){v[c + 2 * k] + bv[c + 2 * k], v[c + 2 * k + 1] + bv[c + 2 * k + 1]};
                    xc[k] = (f32x2_t){med3_bare(x[k][0], -4.5f, 4.5f), med3_bare(x[k][1], -4.5f, 4.5f)};
                }
                gelu_poly2<4>(x, xc);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[c + 2 * k] = x[k][0], v[c + 2 * k + 1] = x[k][1];
            }
        } else {
#pragma unroll
            for (int c = 0; c < NV; ++c) {
                float x = v[c] + bv[c];
                if (ACT == ACT_GELU)
                    x = gelu_for<T>(x);
                else if (ACT == ACT_RELU)
                    x = fmaxf(x, 0.f);
                v[c] = x;
            }
        }
        if (RES == 1) {
#pragma unroll
            for (int g = 0; g < 2 * NG; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q) v[4 * g + q] += rf[sl][g][0][q];
        } else if (RES == 2) {
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (ES == 2) {
                        const vec8h<T> a = __builtin_bit_cast(vec8h<T>, rt[sl][g][0]), b = __builtin_bit_cast(vec8h<T>, rt[sl][g][1]);
                        v[8 * g + q] += two ? (float)a[q] + (float)b[q] : (float)a[q];
                    }
                }
        }
        // row i + RD's residual is requested now, into the slot this row has just released: its values are copies in
        // registers, so an in-place update of THIS row (out aliasing res1) cannot reach them, and rows never overlap
        if constexpr (RES != 0 && i + RD < TM) {
            if (m + 16 * RD < p.M) load_res(m + 16 * RD, std::integral_constant<int, sl>{});
        }
        const long long off = crow(m) * p.ldc + nb;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (!gok[g]) continue;
            const float* vv = v + 8 * g;
            if (p.out_f32) {
                float* op = p.out_f32 + off + 8 * g;
                *(f32x4*)op = (f32x4){vv[0], vv[1], vv[2], vv[3]};
                *(f32x4*)(op + 4) = (f32x4){vv[4], vv[5], vv[6], vv[7]};
            }
            if (ES == 2) {
                if (p.out_relu_T) {
                    vec8h<T> o;
#pragma unroll
                    for (int q = 0; q < 8; ++q) o[q] = (T)fmaxf(vv[q], 0.f);
                    *(vec8h<T>*)((T*)p.out_relu_T + off + 8 * g) = o;
                }
                if (p.out_T) {
                    vec8h<T> o;
#pragma unroll
                    for (int q = 0; q < 8; ++q) o[q] = (T)vv[q];
#ifdef GEMM_DBG_NOSTORE
                    asm volatile("" ::"v"(o));
#else
                    *(vec8h<T>*)((T*)p.out_T + off + 8 * g) = o;
#endif
                }
            }
        }
    });
}

template <typename T, int TM, int TN, int ACT, int RES>
__device__ __forceinline__ void gemm_epilogue_dense(const GemmParams& p, f32x4 (&acc)[TM][TN], int m_wave0, int n_wave0, int li,
                                                    int kg) {
    gemm_epilogue_dense<T, TM, TN, ACT, RES>(p, acc, m_wave0, n_wave0, li, kg, EpiRowMapDesc{p});
}

// Lean form of the L4P_EPI_QKV epilogue (bf16): what a lane's two 8-column groups are (q / k / v, head, dim) does not
// depend on the row, so it is decided once; a row costs one division (m -> batch, token), the bias, the q scale and its stores.
template <typename T, int TM, int TN>
__device__ __forceinline__ void gemm_epilogue_qkv(const GemmParams& p, f32x4 (&acc)[TM][TN], int m_wave0, int n_wave0, int li, int kg) {
    static_assert(sizeof(T) == 2, "bf16 kernels only");
    constexpr int NV = 4 * TN, NG = NV / 8, KVB = 64;
    const int nb = n_wave0 + NV * kg, HD = p.H * p.Dp;
    int kind[NG];          // 0 q, 1 k, 2 v, -1 out of range
    long long cbase[NG];   // column part of the element offset
    int kflip[NG];
    float bv[NV], qs[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int n = nb + 8 * g;
        kind[g] = n >= p.N ? -1 : (n < HD ? 0 : (n < 2 * HD ? 1 : 2));
        qs[g] = (kind[g] == 0 && p.q_scale != 0.f) ? p.q_scale : 1.f;
        kflip[g] = 0;
        cbase[g] = n;
        if (kind[g] == 1) {  // K tile order: 8-element groups [b][h][kv block][k-step][key][half ^ ((key >> 3) & 1)]
            const int nk = n - HD, h = nk / p.Dp, d0 = nk - h * p.Dp;
            kflip[g] = (d0 >> 3) & 1;
            cbase[g] = ((long long)h * (p.S / KVB) * (p.Dp / 16) + (d0 >> 4)) * KVB;  // + (b*H*(S/KVB) + kb)*(Dp/16)*KVB + key
        } else if (kind[g] == 2) {  // V^T: vt[((b*H + h)*Dp + d)*S + s]
            const int nv = n - 2 * HD, h = nv / p.Dp, d = nv - h * p.Dp;
            cbase[g] = ((long long)h * p.Dp + d) * p.S;
        }
        const bool ok = p.bias && kind[g] >= 0;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 b0 = ok ? *(const f32x4*)(p.bias + n) : z, b1 = ok ? *(const f32x4*)(p.bias + n + 4) : z;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bv[8 * g + q] = b0[q];
            bv[8 * g + 4 + q] = b1[q];
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m_wave0 + i * 16 + li;
        if (m >= p.M) continue;
        const int b = m / p.S, s_ = m - b * p.S;
        const int kb = s_ / KVB, key = s_ - kb * KVB;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (kind[g] < 0) continue;
            vec8h<T> o;
#pragma unroll
            for (int q = 0; q < 8; ++q) o[q] = (T)((acc[i][2 * g + q / 4][q % 4] + bv[8 * g + q]) * qs[g]);
            if (kind[g] == 0) {
                *(vec8h<T>*)((T*)p.out_T + (long long)m * p.ldc + cbase[g]) = o;
            } else if (kind[g] == 1) {
                const long long g8 = (cbase[g] + ((long long)b * p.H * (p.S / KVB) + kb) * (p.Dp / 16) * KVB + key) * 2 + (kflip[g] ^ ((key >> 3) & 1));
                *(vec8h<T>*)((T*)p.k_tiled + g8 * 8) = o;
            } else {
                T* vp = (T*)p.vt + (long long)b * p.H * p.Dp * p.S + cbase[g] + s_;
#pragma unroll
                for (int q = 0; q < 8; ++q) vp[(long long)q * p.S] = o[q];
            }
        }
    }
}

// Lean form of the L4P_EPI_CONVT epilogue (ConvTranspose3d with kernel == stride, bf16 output, bias, no activation):
// the tap / output-channel part of the scatter offset is per column group, the voxel part per row.
template <typename T, int TM, int TN>
__device__ __forceinline__ void gemm_epilogue_convt(const GemmParams& p, f32x4 (&acc)[TM][TN], int m_wave0, int n_wave0, int li, int kg) {
    static_assert(sizeof(T) == 2, "bf16 kernels only");
    constexpr int NV = 4 * TN, NG = NV / 8;
    const int nb = n_wave0 + NV * kg;
    bool gok[NG];
    long long coff[NG];
    float bv[NV];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int n = nb + 8 * g;
        gok[g] = n < p.N;
        const int tap = n / p.Cout, co = n - tap * p.Cout;
        const int dw = tap % p.kw, dh = (tap / p.kw) % p.kh, dt = tap / (p.kw * p.kh);
        coff[g] = (((long long)dt * (p.Hi * p.kh) + dh) * (p.Wi * p.kw) + dw) * p.Cout + co;
        const bool ok = p.bias && gok[g];
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 b0 = ok ? *(const f32x4*)(p.bias + n) : z, b1 = ok ? *(const f32x4*)(p.bias + n + 4) : z;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bv[8 * g + q] = b0[q];
            bv[8 * g + 4 + q] = b1[q];
        }
    }
    const bool pow2 = ((p.Wi & (p.Wi - 1)) | (p.Hi & (p.Hi - 1)) | (p.Ti & (p.Ti - 1))) == 0;
    const int sw = __builtin_ctz(p.Wi), sh = __builtin_ctz(p.Hi), st = __builtin_ctz(p.Ti);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m_wave0 + i * 16 + li;
        if (m >= p.M) continue;
        int wi, hi, ti, b;
        if (pow2) {
            wi = m & (p.Wi - 1);
            hi = (m >> sw) & (p.Hi - 1);
            ti = (m >> (sw + sh)) & (p.Ti - 1);
            b = m >> (sw + sh + st);
        } else {
            wi = m % p.Wi;
            int r = m / p.Wi;
            hi = r % p.Hi;
            r /= p.Hi;
            ti = r % p.Ti;
            b = r / p.Ti;
        }
        const long long rowoff = ((((long long)b * p.Ti + ti) * p.kt * (p.Hi * p.kh) + hi * p.kh) * (p.Wi * p.kw) + wi * p.kw) * p.Cout;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (!gok[g]) continue;
            vec8h<T> o;
#pragma unroll
            for (int q = 0; q < 8; ++q) o[q] = (T)(acc[i][2 * g + q / 4][q % 4] + bv[8 * g + q]);
            *(vec8h<T>*)((T*)p.out_T + rowoff + coff[g]) = o;
        }
    }
}

// the plain dense family (activation x residual kind); false = a combination that has no lean body.  Host-side twin of the
// condition: dense_epilogue_is_lean() in gemm_launch.inc.
template <typename T, int TM, int TN, class RowMap>
__device__ __forceinline__ bool gemm_epilogue_dense_cases(const GemmParams& p, f32x4 (&acc)[TM][TN], int m_wave0, int n_wave0,
                                                          int li, int kg, const RowMap& crow) {
    if (p.epi != EPI_DENSE || p.res_mod > 0) return false;
    const int res = !p.res1 ? 0 : (p.res_f32 ? 1 : 2);
    if (res == 1 && p.res2) return false;  // (two float residuals: no caller; the generic body handles it)
#define L4P_EPI_CASE(A, R)                                                                    \
    if (p.act == A && res == R) {                                                             \
        gemm_epilogue_dense<T, TM, TN, A, R>(p, acc, m_wave0, n_wave0, li, kg, crow);         \
        return true;                                                                          \
    }
    L4P_EPI_CASE(ACT_NONE, 0)
    L4P_EPI_CASE(ACT_GELU, 0)
    L4P_EPI_CASE(ACT_RELU, 0)
    L4P_EPI_CASE(ACT_NONE, 1)
    L4P_EPI_CASE(ACT_NONE, 2)
    L4P_EPI_CASE(ACT_RELU, 2)
#undef L4P_EPI_CASE
    return false;
}

// run-time selection of the specialisation (wave-uniform); false = not a plain dense epilogue, use the generic one
template <typename T, int TM, int TN>
__device__ __forceinline__ bool gemm_epilogue_dense_dispatch(const GemmParams& p, f32x4 (&acc)[TM][TN], int m_wave0, int n_wave0,
                                                             int li, int kg) {
    static_assert(sizeof(T) == 2, "bf16 kernels only");
    if (p.tuning & 1) return false;
    if (p.epi == EPI_QKV && !p.res1 && p.act == ACT_NONE && p.c_gr == 0) {
        gemm_epilogue_qkv<T, TM, TN>(p, acc, m_wave0, n_wave0, li, kg);
        return true;
    }
    if (p.epi == EPI_CONVT && !p.res1 && p.act == ACT_NONE && p.out_T && !p.out_f32 && !p.out_relu_T) {
        gemm_epilogue_convt<T, TM, TN>(p, acc, m_wave0, n_wave0, li, kg);
        return true;
    }
    return gemm_epilogue_dense_cases<T, TM, TN>(p, acc, m_wave0, n_wave0, li, kg, EpiRowMapDesc{p});
}

// GLDS = true: tiles are staged with global_load_lds (LDS-DMA, no VGPR round trip, no ds_write); the LDS image is
// lane-linear, so the XOR swizzle is applied to the per-lane SOURCE chunk and again on the fragment reads.
// GLDS = false: global -> register -> ds_write staging (needed for the fused input ReLU of the conv loader).
// STAGES = 3 (LDS-DMA only): two k-tiles stay in flight; the per-iteration wait is a COUNTED s_waitcnt vmcnt(loads of one
// tile) followed by a raw s_barrier, so the next tile's DMA is not drained at the barrier (a __syncthreads() would emit
// vmcnt(0)).  Order per iteration: wait(own loads of tile kt) -> barrier (everyone's tile kt landed, everyone finished
// reading tile kt-1) -> issue tile kt+2 into the buffer tile kt-1 used -> MFMAs on tile kt.
// (the kernel body is a device function of (descriptor, workgroup index): gemm_kernel runs it for one problem,
//  gemm_group_kernel for up to four problems in ONE launch, see below)
// GROUPW: row-grouped weights (l4p_gemm_desc.w_gr): the tile's row group selects the weight matrix and the bias row.  Its own
// instantiation - the descriptor is copied and patched per workgroup, which the plain kernels must not pay for.
template <typename T, int BM, int BN, int WM, int WN, int MODE, bool GLDS, int STAGES = 2, bool GROUPW = false>
__device__ __forceinline__ void gemm_body(const GemmParams& p_in, const int wg_index) {
    static_assert(STAGES == 2 || (STAGES >= 3 && STAGES <= 5 && GLDS), "deeper pipelines need LDS-DMA staging");
    static_assert(!GROUPW || MODE == 0, "row-grouped weights: dense GEMM");
    GemmParams patched;
    const GemmParams* pp = &p_in;
    if constexpr (GROUPW) {
        const int ntn_ = (p_in.N + BN - 1) / BN;
        const int grp = ((wg_index / ntn_) * BM) / p_in.w_gr;  // (no split-K, no XCD remap in MODE 0: tile = wg_index, m-major)
        patched = p_in;
        patched.W = (const T*)p_in.W + (long long)grp * p_in.w_gs;
        if (p_in.bias) patched.bias = p_in.bias + (long long)grp * p_in.b_gs;
        if (p_in.o_gs) {
            if (p_in.out_T) patched.out_T = (T*)p_in.out_T + (long long)grp * p_in.o_gs;
            if (p_in.out_f32) patched.out_f32 = p_in.out_f32 + (long long)grp * p_in.o_gs;
        }
        pp = &patched;
    }
    const GemmParams& p = *pp;
    constexpr int NT = WM * WN * 64;
    constexpr int ES = sizeof(T);
    constexpr int BK = 128 / ES;   // 64 bf16 / 32 f32 per LDS row
    constexpr int EPC = 16 / ES;   // elements per 16-byte chunk
    constexpr int CPF = 8 * ES / 16;  // chunks per fragment (1 bf16, 2 f32)
    constexpr int KK = BK / 32;
    constexpr int A_IT = BM * 8 / NT, W_IT = BN * 8 / NT;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    typedef typename Frag<T>::type frag_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                       // [STAGES][BM*128]
    char* Ws = smem + STAGES * BM * 128;   // [STAGES][BN*128]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int ntn = (p.N + BN - 1) / BN;
    const int ntiles = ntn * ((p.M + BM - 1) / BM);
    const int ksplit = wg_index / ntiles;  // split-K slice of this workgroup
    int tile = wg_index - ksplit * ntiles;
    if (MODE == 1 && p.splitk <= 1) {  // conv: XCD-contiguous tile ranges (workgroup b runs on XCD b % 8)
        const int xcd = tile & 7, idx = tile >> 3, q = ntiles >> 3, r = ntiles & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int mt = tile / ntn;
    const int nt = tile % ntn;
    if (MODE == 1 && p.splitk <= 1) mt = conv_tile_walk(p, mt, BM, ES);
    const int m0 = mt * BM, n0 = nt * BN;

    const int crow = tid >> 3, cc = tid & 7;  // this thread's (row, LDS slot) inside a staging pass
    // chunk of the 128-byte source row this thread fetches: with LDS-DMA the swizzle moves to the source side
    const int cs = GLDS ? (cc ^ ((crow >> 1) & 7)) : cc;
    // The W tile is read with the permuted row map (rows 16g + 4j + r of a wave's range, see wrow[] below), so it gets
    // its own XOR phase: distinct over (g, r >> 1) where the A tile's phase is distinct over 8 consecutive row pairs.
    auto sww = [](int row) { return ((((row % (16 * TN)) / (4 * TN)) & 3) << 1) | ((row >> 1) & 1); };
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- per-thread source addressing -------------------------------------------------------
    const T* a_base[A_IT];
    unsigned a_mask[A_IT];
    const T* w_base[W_IT];
    int cs_w[W_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        int m = m0 + crow + i * (NT / 8);
        if (m >= p.M) m = p.M - 1;
        if (MODE == 0) {
            const long long pm = p.a_gr > 0 ? (long long)(m / p.a_gr) * p.a_gs + p.a_go + (m % p.a_gr) : m;
            a_base[i] = (const T*)p.A + pm * p.lda + cs * EPC;
            a_mask[i] = 0;
        } else {
            int wo = m % p.Wo;
            int r = m / p.Wo;
            int ho = r % p.Ho;
            r /= p.Ho;
            int to = r % p.To;
            int b = r / p.To;
            int ti = to * p.st, hi = ho * p.sh, wi = wo * p.sw;
            unsigned mask = 0;
#pragma unroll
            for (int tap = 0; tap < 27; ++tap) {
                int dt = tap / 9 - 1, dh = (tap / 3) % 3 - 1, dw = tap % 3 - 1;
                bool ok = (unsigned)(ti + dt) < (unsigned)p.Ti && (unsigned)(hi + dh) < (unsigned)p.Hi &&
                          (unsigned)(wi + dw) < (unsigned)p.Wi;
                mask |= (ok ? 1u : 0u) << tap;
            }
            a_mask[i] = mask;
            long long vox = (((long long)b * p.Ti + ti) * p.Hi + hi) * p.Wi + wi;
            a_base[i] = (const T*)p.A + vox * p.Cin + cs * EPC;
        }
    }
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
        int n = n0 + crow + i * (NT / 8);
        // (plain weights are packed with rows padded to the tile, packing.py; a GROUP's matrix has exactly N rows and the next
        //  group's - or nothing - behind them: rows past N re-read row N - 1, their outputs are discarded by the epilogue)
        if constexpr (GROUPW) n = n < p.N ? n : p.N - 1;
        cs_w[i] = GLDS ? (cc ^ sww(crow + i * (NT / 8))) : cc;
        w_base[i] = (const T*)p.W + (long long)n * p.ldw + cs_w[i] * EPC;
    }

    const int nk = (p.K + BK - 1) / BK;
    const int kpc = (MODE == 1) ? (p.Cin / BK) : 1;  // k-tiles per conv tap

    u32x4 ra[A_IT], rw[W_IT];

    auto load_tile = [&](int kt) {
        if (MODE == 0) {
            const int k = kt * BK + cs * EPC;
            const bool kin = k < p.K;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                u32x4 z = {0, 0, 0, 0};
                ra[i] = kin ? *(const u32x4*)(a_base[i] + kt * BK) : z;
            }
#pragma unroll
            for (int i = 0; i < W_IT; ++i) {
                u32x4 z = {0, 0, 0, 0};
                rw[i] = (kt * BK + cs_w[i] * EPC) < p.K ? *(const u32x4*)(w_base[i] + kt * BK) : z;
            }
        } else {
            const int tap = kt / kpc;
            const int ci0 = (kt - tap * kpc) * BK;
            const int dt = tap / 9 - 1, dh = (tap / 3) % 3 - 1, dw = tap % 3 - 1;
            const long long toff = (((long long)dt * p.Hi + dh) * p.Wi + dw) * p.Cin + ci0;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                u32x4 z = {0, 0, 0, 0};
                const bool ok = (a_mask[i] >> tap) & 1u;
                u32x4 v = ok ? *(const u32x4*)(a_base[i] + toff) : z;
                if (p.relu_in) {
                    if (ES == 2) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            unsigned x = v[q];
                            unsigned neg = ((x >> 15) & 0x00010001u) * 0xFFFFu;
                            v[q] = x & ~neg;
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = __float_as_uint(fmaxf(__uint_as_float(v[q]), 0.0f));
                    }
                }
                ra[i] = v;
            }
#pragma unroll
            for (int i = 0; i < W_IT; ++i) rw[i] = *(const u32x4*)(w_base[i] + kt * BK);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int row = crow + i * (NT / 8);
            *(u32x4*)(As + buf * BM * 128 + row * 128 + ((cc ^ ((row >> 1) & 7)) << 4)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < W_IT; ++i) {
            const int row = crow + i * (NT / 8);
            *(u32x4*)(Ws + buf * BN * 128 + row * 128 + ((cc ^ sww(row)) << 4)) = rw[i];
        }
    };

    // LDS-DMA staging: one global_load_lds per 16-byte chunk; destination = wave-uniform base + lane*16
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    // conv position of the LDS-DMA stream (see issue_tile); starts at this workgroup's first k-tile (split-K: not 0)
    int cp_kt = 0, cp_tap = 0, cp_ci0 = 0;
    long long cp_off = 0;
    if (MODE == 1) {
        const int nsp = p.splitk > 1 ? p.splitk : 1;
        cp_kt = (int)((long long)nk * ksplit / nsp);
        cp_tap = cp_kt / kpc;
        cp_ci0 = (cp_kt - cp_tap * kpc) * BK;
        const int tp = cp_tap < 27 ? cp_tap : 26;
        cp_off = (((long long)(tp / 9 - 1) * p.Hi + ((tp / 3) % 3 - 1)) * p.Wi + (tp % 3 - 1)) * p.Cin;
    }
    auto issue_tile = [&](int kt, int buf) {
        const char* zero = (const char*)g_zero_chunk;
        if (MODE == 0) {
            const bool kin = (kt * BK + cs * EPC) < p.K;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const char* src = kin ? (const char*)(a_base[i] + kt * BK) : zero;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(As + buf * BM * 128 + (wave_u * 64 + i * NT) * 16), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < W_IT; ++i) {
                const char* src = (kt * BK + cs_w[i] * EPC) < p.K ? (const char*)(w_base[i] + kt * BK) : zero;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Ws + buf * BN * 128 + (wave_u * 64 + i * NT) * 16), 16, 0, 0);
            }
        } else {
            // (tap, channel slice) of k-tile kt, carried as counters: tiles are issued in k order, one further per call
            while (cp_kt < kt) {
                ++cp_kt;
                cp_ci0 += BK;
                if (cp_ci0 == p.Cin) {
                    cp_ci0 = 0;
                    ++cp_tap;
                    const int tp = cp_tap < 27 ? cp_tap : 26;
                    cp_off = (((long long)(tp / 9 - 1) * p.Hi + ((tp / 3) % 3 - 1)) * p.Wi + (tp % 3 - 1)) * p.Cin;
                }
            }
            const int tap = cp_tap;
            const long long toff = cp_off + cp_ci0;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const bool ok = (a_mask[i] >> tap) & 1u;
                const char* src = ok ? (const char*)(a_base[i] + toff) : zero;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(As + buf * BM * 128 + (wave_u * 64 + i * NT) * 16), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < W_IT; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t)(const char*)(w_base[i] + kt * BK),
                                                 (lptr_t)(Ws + buf * BN * 128 + (wave_u * 64 + i * NT) * 16), 16, 0, 0);
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int li = lane & 15, kg = lane >> 4;
    // fragment rows inside the block tile
    int xrow[TM], wrow[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) xrow[i] = wm * (TM * 16) + i * 16 + li;
#pragma unroll
    for (int j = 0; j < TN; ++j) wrow[j] = wn * (TN * 16) + 4 * TN * (li >> 2) + 4 * j + (li & 3);

    // split-K: this workgroup contracts k-tiles [kt0, kt1) only and leaves a float partial (see below)
    const int nsplit = p.splitk > 1 ? p.splitk : 1;
    int kt0 = (int)((long long)nk * ksplit / nsplit), kt1 = (int)((long long)nk * (ksplit + 1) / nsplit);
    if (MODE == 0 && p.kw_cols > 0) {  // block-structured weights (l4p_gemm_desc.kw_cols): only the k-tiles of this tile's column group
        const int grp = n0 / p.kw_cols;
        kt0 = (grp * p.kw_len) / BK;
        kt1 = ((grp + 1) * p.kw_len + BK - 1) / BK;
        kt1 = kt1 < nk ? kt1 : nk;
    }
    if (STAGES >= 3) {
#pragma unroll
        for (int s = 0; s < STAGES - 1; ++s)
            if (kt0 + s < kt1) issue_tile(kt0 + s, s);
    } else if (GLDS) {
        issue_tile(kt0, 0);
        __syncthreads();
    } else {
        load_tile(kt0);
        store_tile(0);
        __syncthreads();
    }

    for (int kt = kt0; kt < kt1; ++kt) {
        const int cur = STAGES >= 3 ? (kt - kt0) % STAGES : (kt - kt0) & 1;
        if (STAGES >= 3) {
            // tile kt has landed once only the tiles issued after it (at most STAGES - 2 of them) are still in flight
            const int ahead = kt1 - 1 - kt < STAGES - 2 ? kt1 - 1 - kt : STAGES - 2;
            if (ahead >= 3)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (A_IT + W_IT)) : "memory");
            else if (ahead == 2)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (A_IT + W_IT)) : "memory");
            else if (ahead == 1)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_IT + W_IT) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + STAGES - 1 < kt1) issue_tile(kt + STAGES - 1, (kt + STAGES - 1 - kt0) % STAGES);
        } else if (kt + 1 < kt1) {
#ifndef GEMM_DBG_NOLOAD  // (tools/probes/gemm_variants.hip: ablation timing)
            if (GLDS)
                issue_tile(kt + 1, cur ^ 1);
            else
                load_tile(kt + 1);
#endif
        }
        const char* Ab = As + cur * BM * 128;
        const char* Wb = Ws + cur * BN * 128;
#if !defined(GEMM_DBG_NOMMA) && !defined(GEMM_NO_HANDSCHED)
        if constexpr (ES == 2 && GLDS && STAGES == 2) {
            // Hand-scheduled k-tile (bf16, LDS-DMA staging).  hipcc answers every LDS wait with lgkmcnt(0) while an LDS-DMA is
            // in flight (four full drains of the read queue per k-tile in its own schedule).  The fragment reads are therefore
            // inline-asm ds_read_b128 with hand-counted lgkmcnt: LDS reads return in order, so MFMA m may issue as soon as the
            // only outstanding reads are those requested after its two operands.  Read order per k-step: A_0, W_0..W_{TN-1},
            // A_1..A_{TM-1}; MFMA order per k-step: i-major; PRE reads run ahead, one more is requested behind each MFMA.
            #ifndef GEMM_HS_PRE
#define GEMM_HS_PRE (TM + TN)
#endif
            constexpr int RPK = TM + TN, NR = KK * RPK, NM = KK * TM * TN, PRE = (GEMM_HS_PRE) < NR ? (GEMM_HS_PRE) : NR;
            const unsigned a_base = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)Ab + (wm * (TM * 16) + li) * 128;
            const unsigned w_base = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)Wb +
                                    (wn * (TN * 16) + 4 * TN * (li >> 2) + (li & 3)) * 128;
            const int swa = (li >> 1) & 7, swq = (((li >> 2) & 3) << 1) | ((li >> 1) & 1);
            unsigned a_ad[KK], w_ad[KK];
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                a_ad[kk] = a_base + (((kk * 4 + kg) ^ swa) << 4);
                w_ad[kk] = w_base + (((kk * 4 + kg) ^ swq) << 4);
            }
            u32x4 fr[NR];
            auto rd = [&fr, &a_ad, &w_ad](auto r_) {  // (explicit captures: asm operands do not trigger implicit capture)
                constexpr int r = decltype(r_)::value, kk = r / RPK, q = r % RPK;
                if constexpr (q == 0)
                    asm volatile("ds_read_b128 %0, %1" : "=v"(fr[r]) : "v"(a_ad[kk]) : "memory");
                else if constexpr (q <= TN)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[r]) : "v"(w_ad[kk]), "n"((q - 1) * 512) : "memory");
                else
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[r]) : "v"(a_ad[kk]), "n"((q - TN) * 2048) : "memory");
            };
            static_for_<0, (PRE < NR ? PRE : NR)>(rd);
            static_for_<0, NM>([&](auto m_) {
                constexpr int m = decltype(m_)::value, kk = m / (TM * TN), i = (m % (TM * TN)) / TN, j = m % TN;
                constexpr int ra = kk * RPK + (i == 0 ? 0 : TN + i), rw = kk * RPK + 1 + j;  // reads holding A_i / W_j
                constexpr int need = ra > rw ? ra : rw;
                constexpr int issued = (PRE + m < NR) ? PRE + m : NR;
                static_assert(need < issued, "operand requested before use");
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(issued - need - 1 > 15 ? 15 : issued - need - 1) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                acc[i][j] = mma16(__builtin_bit_cast(frag_t, fr[rw]), __builtin_bit_cast(frag_t, fr[ra]), acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (PRE + m < NR) rd(std::integral_constant<int, PRE + m>{});
            });
        } else
#endif
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            frag_t xf[TM], wf[TN];
            const int c0 = kk * 2 * ES + kg * CPF;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = xrow[i];
                const int sw = (row >> 1) & 7;
                u32x4* d = (u32x4*)&xf[i];
#pragma unroll
                for (int q = 0; q < CPF; ++q) d[q] = *(const u32x4*)(Ab + row * 128 + (((c0 + q) ^ sw) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wrow[j];
                const int sw = sww(row);
                u32x4* d = (u32x4*)&wf[j];
#pragma unroll
                for (int q = 0; q < CPF; ++q) d[q] = *(const u32x4*)(Wb + row * 128 + (((c0 + q) ^ sw) << 4));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
#ifdef GEMM_DBG_NOMMA  // keep the fragment reads alive, skip the matrix pipe
                    asm volatile("" ::"v"(wf[j]), "v"(xf[i]));
#else
                    acc[i][j] = mma16(wf[j], xf[i], acc[i][j]);
#endif
                }
        }
        if (!GLDS && kt + 1 < kt1) store_tile(cur ^ 1);
        if (STAGES == 2) __syncthreads();
    }

    if (p.splitk > 1) {
        // raw float partial [ksplit][M][N]; bias / activation / residuals / conversion happen in splitk_finish_kernel
        const int nb2 = n0 + wn * (TN * 16) + 4 * TN * kg;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * (TM * 16) + i * 16 + li;
            if (m >= p.M) continue;
            float* pp = p.partial + ((long long)ksplit * p.M + m) * p.N + nb2;
#pragma unroll
            for (int j = 0; j < TN; ++j)
                if (nb2 + 4 * j < p.N) *(f32x4*)(pp + 4 * j) = acc[i][j];
        }
        return;
    }

    if (p.epi == EPI_MASKDOT) {
        gemm_epilogue_maskdot<T, TM, TN>(p, acc, m0 + wm * (TM * 16), n0 + wn * (TN * 16), li, kg);
        return;
    }
    if constexpr (ES == 2) {  // plain dense / QKV epilogues: the lean specialisations
        if (gemm_epilogue_dense_dispatch<T, TM, TN>(p, acc, m0 + wm * (TM * 16), n0 + wn * (TN * 16), li, kg)) return;
    }
    gemm_epilogue<T, TM, TN>(p, acc, m0 + wm * (TM * 16), n0 + wn * (TN * 16), li, kg);
}

template <typename T, int BM, int BN, int WM, int WN, int MODE, bool GLDS, int STAGES = 2, bool GROUPW = false>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel(const GemmParams p) {
    gemm_body<T, BM, BN, WM, WN, MODE, GLDS, STAGES, GROUPW>(p, (int)blockIdx.x);
}

// Up to L4P_GEMM_GROUP_MAX independent dense GEMMs as ONE launch: workgroups [first[g], first[g + 1]) run problem g.  For the
// tracker's token-side projections (self-attention q / k / v, image -> token k / v, the three hyper-network MLP stages): each is
// a 20 us launch that fills a quarter of the chip, and they come in independent groups of two or three.
struct GemmGroupParams {
    GemmParams p[L4P_GEMM_GROUP_MAX];
    int first[L4P_GEMM_GROUP_MAX + 1];
};
template <typename T, int BM, int BN, int WM, int WN, bool GLDS, int STAGES>
__global__ __launch_bounds__(WM* WN * 64) void gemm_group_kernel(const GemmGroupParams g) {
    const int b = (int)blockIdx.x;
    int i = 0;
#pragma unroll
    for (int k = 1; k < L4P_GEMM_GROUP_MAX; ++k)
        if (b >= g.first[k]) i = k;
    // (a uniform switch keeps the descriptor accesses static: kernel arguments are read through scalar loads)
    switch (i) {
        case 0: gemm_body<T, BM, BN, WM, WN, 0, GLDS, STAGES>(g.p[0], b - g.first[0]); break;
        case 1: gemm_body<T, BM, BN, WM, WN, 0, GLDS, STAGES>(g.p[1], b - g.first[1]); break;
        case 2: gemm_body<T, BM, BN, WM, WN, 0, GLDS, STAGES>(g.p[2], b - g.first[2]); break;
        default: gemm_body<T, BM, BN, WM, WN, 0, GLDS, STAGES>(g.p[3], b - g.first[3]); break;
    }
}

// host launcher (gemm.hip)
int launch_gemm(int dtype, int mode, const GemmParams& p, hipStream_t stream);
int launch_gemm_group(int dtype, const GemmParams* p, int n, hipStream_t stream);
