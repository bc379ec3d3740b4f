// Encoder self-attention, chip-filling batches (>= 256 tiles of 256 query rows): ONE wave per SIMD, 64 query rows per wave.
// (reference modeling_finetune.py:169-190; operands, layouts and the arithmetic are those of attention.hip's hand-scheduled
// form - S^T = K Q^T, deferred running maximum riding in the head-dim padding, denominator out of the MFMA - so the two kernels
// agree to the rounding of P.)
//
// Why a second kernel: the 8-wave form (attention.hip, QS = 2) gives every wave 32 query rows, so each 32x32x16 MFMA needs its own
// ds_read_b128 fragment (K for S^T, V^T for O^T) and the CU's LDS pipe is as busy as its matrix pipe.  Here a workgroup is 4 waves
// = one 256-row Q tile of one (batch, head); a wave owns TWO 32-row query blocks and the whole 512-entry register file of its
// SIMD (S ping-pong 128, O 96, Q 48, P 32 + the early part of the next P, fragments in flight), so every K / V^T fragment read
// feeds two MFMAs: half the LDS reads, half the LDS-DMA requests and half the waves per barrier for the same MFMA count.
// 256 persistent workgroups (one per CU) walk the tiles; the next tile's K_0 / K_1 / V_0 ride the rings through the seam and
// its Q rows are staged in LDS by DMA a few blocks before the end (as in the 8-wave form).
//
// Per KV block (64 keys) a wave issues 48 MFMAs (24 for S_{j+1}^T = K_{j+1} Q^T, 24 for O^T += V_j^T P_j^T), 24 ds_read_b128, 6 LDS-DMA
// requests, 64 v_exp + 32 v_cvt_pk + 32 v_max3: ~3.3 single-issue instructions per MFMA gap (the guide's budget for one wave per
// SIMD is 5).  With no second wave on the SIMD to fill bubbles the vector work is spread evenly: the exponentials of block j run
// under the QK^T MFMAs of block j+1 EXCEPT their first E pairs, which run one phase earlier (under the PV MFMAs of block j-1,
// beside the row maximum of S_j): with the deferred maximum an exponential does not wait for the maximum - when the rare rescale
// does trigger, the early pairs are recomputed from the (still intact) scores.
#include <cstdlib>
#include <type_traits>

#include "common.hpp"

namespace attn64 {

#ifdef ATTN64_TRACE  // (tools/probes/attn64_probe.hip: s_memtime stamps of wave 0 of workgroup 0 at the phase boundaries of a KV step)
__device__ long long g_trace[4096];
#define A64_STAMP(slot)                                                                    \
    do {                                                                                   \
        if (blockIdx.x == 0 && threadIdx.x == 0 && (slot) < 4096) g_trace[slot] = __builtin_readcyclecounter(); \
    } while (0)
// inside a KV step the stamps stay in scalar registers (s_memtime returns through lgkmcnt like an LDS read: waiting for each one
// would drain the fragment reads in flight and time a different kernel) and are stored behind the step's closing barrier - only in
// steps that store them: an s_memtime whose destination the compiler considers dead lands in registers it has handed to something else
#define A64_STEP_STAMP(i)                                                    \
    do {                                                                     \
        if constexpr (HAS_NEXT) asm volatile("s_memtime %0" : "=s"(ts_[i])); \
    } while (0)
#else
#define A64_STAMP(slot) \
    do {                \
    } while (0)
#define A64_STEP_STAMP(i) \
    do {                  \
    } while (0)
#endif

// the two constant pieces of a tile (attention.hip): row DH of V^T reads ones (the denominator comes out of the MFMA), head dims DH, DH+1
// of K read 1.0 (they meet the two halves of -m in Q's padding)
__device__ __attribute__((aligned(16))) static const unsigned short g_ones_bf16[8] = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
__device__ __attribute__((aligned(16))) static const unsigned short g_ones_f16[8] = {0x3C00, 0x3C00, 0x3C00, 0x3C00, 0x3C00, 0x3C00, 0x3C00, 0x3C00};
__device__ __attribute__((aligned(16))) static const unsigned short g_kone_bf16[8] = {0x3F80, 0x3F80, 0, 0, 0, 0, 0, 0};
__device__ __attribute__((aligned(16))) static const unsigned short g_kone_f16[8] = {0x3C00, 0x3C00, 0, 0, 0, 0, 0, 0};

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ unsigned pack2(float a, float b, bf16_t) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    const bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ unsigned pack2(float a, float b, f16_t) {
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
    const f16x2 v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(unsigned, v);
}

#ifndef ATTN64_EARLY
#define ATTN64_EARLY 14
#endif
#ifndef ATTN64_PRE
#define ATTN64_PRE 4
#endif

template <typename T, int DH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn64_kernel(
    const T* __restrict__ q, const T* __restrict__ kt, const T* __restrict__ vt, T* __restrict__ out, int S, int H, float c_scale, int ntiles) {
    typedef typename Frag<T>::type frag_t;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    constexpr int DP = 96, KVB = 64, ES = 2, QB = 2;
    constexpr int NKS = DP / 16, NST = KVB / 32, NDT = DP / 32;  // 6 k-steps, 2 score tiles, 3 output d tiles
    constexpr int KBYTES = NKS * KVB * 2 * 8 * ES, VBYTES = DP * 128;  // 12 KB each
    constexpr int K_IT = KBYTES / 4096, V_IT = VBYTES / 4096;
    constexpr int QROWS = 128 * QB, QBYTES = QROWS * DP * ES;
    constexpr int DT_L = DH / 32, I_L = DH % 32, HI_L = (I_L >> 2) & 1, R_L = (I_L & 3) + 4 * (I_L >> 3);  // the denominator row of O^T
    constexpr int KS_P = DH / 16, HI_P = (DH / 8) & 1;  // k-step / lane half whose Q fragment holds dims DH .. DH+7
    static_assert(DH % 8 == 0 && DH < DP, "the first padding chunk starts at DH and carries -m / the denominator");
    constexpr int NP = QB * NST * 16 / 2;  // 32 pairs of scores per lane and KV block
    constexpr int E = ATTN64_EARLY;        // pairs exponentiated one phase early
    constexpr float RESCALE_THR = 8.f;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;                // [2][KBYTES]
    char* Vs = smem + 2 * KBYTES;   // [2][VBYTES]
    char* Qs = Vs + 2 * VBYTES;     // [QROWS][DP]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, hi = lane >> 5;
    const int nqb = S / QROWS, units = ntiles / nqb;  // units = B * H
    auto decode = [&](int v, int& bh_, int& qrow0_) __attribute__((always_inline)) {  // (XCD-aware: all query blocks of one (batch, head) on one XCD, attention.hip)
        int unit, qb;
        if ((units & 7) == 0) {
            const int xcd = v & 7, j = v >> 3;
            unit = xcd + 8 * (j / nqb);
            qb = j % nqb;
        } else {
            unit = v / nqb;
            qb = v % nqb;
        }
        bh_ = unit;
        qrow0_ = (unit / H) * S + qb * QROWS;
    };
    const int nkb = S / KVB;

    // ---- LDS-DMA: every request is (wave-uniform 64-bit base) + (one lane-constant 32-bit offset) --------------------------------
    // A K tile is a linear 12 KB copy: chunk c = tid + 256 i.  A V^T tile row d = tid / 8 + 32 i is 128 bytes at row stride S; the XOR
    // swizzle goes through the source chunk.  The lanes of the two constant pieces (pass i = 2 only: waves 2, 3 for K; lanes 0-7 of
    // wave 3 for V^T) fetch the constants instead.
    const unsigned k_lane = tid * 16;
    const int vrow0 = tid >> 3, vslot = tid & 7;
    const unsigned v_lane = (unsigned)(vrow0 * S * ES) + ((vslot ^ ((vrow0 >> 1) & 7)) * 16);
    const char* ones = std::is_same<T, f16_t>::value ? (const char*)g_ones_f16 : (const char*)g_ones_bf16;
    const char* kone = std::is_same<T, f16_t>::value ? (const char*)g_kone_f16 : (const char*)g_kone_bf16;
    bool kpad2, vone2;
    {
        const int c = tid + 512, key = (c % (2 * KVB)) >> 1;
        kpad2 = (c / (2 * KVB)) * 16 + (((c & 1) ^ ((key >> 3) & 1)) << 3) == DH;
        vone2 = vrow0 + 64 == DH;
        static_assert(DH >= 64 && DH - 64 < 32, "the constant pieces sit in the third pass of the tile copy");
    }
    // (bhx = batch * H + head of the tile the operands belong to: wave-uniform, kept scalar by the callers)
    auto issue_k = [&](int bhx, int kb, int buf) __attribute__((always_inline)) {
        const char* base = (const char*)kt + ((long long)bhx * nkb + kb) * KBYTES;
#pragma unroll
        for (int i = 0; i < K_IT; ++i) {
            const char* src = base + i * 4096 + k_lane;
            if (i == 2) src = kpad2 ? kone : src;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Ks + buf * KBYTES + (wave * 64 + i * 256) * 16), 16, 0, 0);
        }
    };
    auto issue_v = [&](int bhx, int kb, int buf) __attribute__((always_inline)) {
        const char* base = (const char*)vt + (long long)bhx * DP * S * ES + (long long)kb * 128;
#pragma unroll
        for (int i = 0; i < V_IT; ++i) {
            const char* src = base + (long long)i * 32 * S * ES + v_lane;
            if (i == 2) src = vone2 ? ones : src;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Vs + buf * VBYTES + (wave * 64 + i * 256) * 16), 16, 0, 0);
        }
    };
    auto issue_q = [&](int qrow0n, int bhn) __attribute__((always_inline)) {  // the next tile's Q rows -> Qs, a linear [QROWS][DP] image
        constexpr int CPR = DP * ES / 16, NT = 256;
        const char* qb_ = (const char*)(q + (long long)qrow0n * ((long long)H * DP) + (long long)(bhn % H) * DP);
        // (an opaque zero: the per-lane offsets below are invariant in the KV loop; hoisted out of it they would be held - spilled -
        //  through the whole tile)
        int z;
        asm volatile("v_mov_b32 %0, 0" : "=v"(z));
        const int t_ = tid + z;
#pragma unroll
        for (int i = 0; i < QROWS * CPR / NT; ++i) {
            const int c = i * NT + t_, row = c / CPR, pc = c - row * CPR;
            __builtin_amdgcn_global_load_lds((gptr_t)(qb_ + (unsigned)(row * (H * DP * ES) + pc * 16)), (lptr_t)(Qs + (i * NT + wave * 64) * 16), 16,
                                             0, 0);
        }
    };

    frag_t qf[QB][NKS];
    f32x16 s_a[QB][NST], s_b[QB][NST];
    unsigned pw[NP];      // P of the block the next step multiplies with V^T, packed pairs; filled early for p < E
    float mx[QB] = {0.f, 0.f};

    // fragment addresses (lane-constant)
    const int krow = (lq & ~12) | ((lq & 4) << 1) | ((lq & 8) >> 1);
    int koff[NST];
#pragma unroll
    for (int t = 0; t < NST; ++t) {
        const int key = t * 32 + krow;
        koff[t] = (key * 2 + (hi ^ ((key >> 3) & 1))) * 8 * ES;
    }
    const int vsw = (lq >> 1) & 7;
    const unsigned lds_k = (unsigned)(size_t)(__attribute__((address_space(3))) char*)Ks;
    const unsigned lds_v = (unsigned)(size_t)(__attribute__((address_space(3))) char*)Vs;
    unsigned ka[NST], va[NST * 2];  // + ring slot and k-step / d-tile as instruction immediates
#pragma unroll
    for (int t = 0; t < NST; ++t) ka[t] = lds_k + koff[t];
#pragma unroll
    for (int tj = 0; tj < NST * 2; ++tj) va[tj] = lds_v + lq * 128 + ((((tj * 16 + hi * 8) >> 3) ^ vsw) << 4);

    // the softmax scale times log2(e) is folded into Q (c_scale == 1: q arrives pre-scaled, nothing to do and no second rounding)
    auto q_prescale = [&]() __attribute__((always_inline)) {
        if (c_scale != 1.0f) {
#pragma unroll
            for (int b = 0; b < QB; ++b)
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                    for (int e = 0; e < 8; ++e) qf[b][ks][e] = from_f32<T>(to_f32<T>(qf[b][ks][e]) * c_scale);
        }
    };
    for (int v = blockIdx.x; v < ntiles; v += (int)gridDim.x) {
        int bh, qrow0;
        decode(v, bh, qrow0);
        const int vn = v + (int)gridDim.x;
        const bool has_next_tile = vn < ntiles;
        int bhn = 0, qrow0n = 0;
        if (has_next_tile) decode(vn, bhn, qrow0n);
        bh = __builtin_amdgcn_readfirstlane(bh), qrow0 = __builtin_amdgcn_readfirstlane(qrow0);
        bhn = __builtin_amdgcn_readfirstlane(bhn), qrow0n = __builtin_amdgcn_readfirstlane(qrow0n);
        const bool first_tile = v == (int)blockIdx.x;
        const int h = bh % H;
        const int nit = nkb;
        if (first_tile) {
            issue_k(bh, 0, 0);
            issue_v(bh, 0, 0);
            issue_k(bh, 1, 1);
#pragma unroll
            for (int b = 0; b < QB; ++b) {
                const T* qp = q + (long long)(qrow0 + wave * 64 + b * 32 + lq) * ((long long)H * DP) + (long long)h * DP;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) *(u32x4*)&qf[b][ks] = *(const u32x4*)(qp + ks * 16 + hi * 8);
            }
            q_prescale();
        }

        f32x16 o[QB][NDT];
#pragma unroll
        for (int b = 0; b < QB; ++b)
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[b][dt][r] = 0.f;
        float m_run[QB] = {0.f, 0.f};

        A64_STAMP(first_tile ? 0 : 4);
        // ---- tile prologue: S_0 and its row maxima ---------------------------------------------------------------------------
        __syncthreads();  // K_0 / V_0 / K_1 (and, past the first tile, Q in Qs) have landed; the previous tile's reads are done
        if (!first_tile) {
#pragma unroll
            for (int b = 0; b < QB; ++b)
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks)
                    *(u32x4*)&qf[b][ks] = *(const u32x4*)(Qs + ((wave * 64 + b * 32 + lq) * DP + ks * 16 + hi * 8) * ES);
            q_prescale();
        }
#pragma unroll
        for (int t = 0; t < NST; ++t) {
#pragma unroll
            for (int b = 0; b < QB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) s_a[b][t][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const frag_t kf = *(const frag_t*)(Ks + ks * (KVB * 2 * 8 * ES) + koff[t]);
#pragma unroll
                for (int b = 0; b < QB; ++b) s_a[b][t] = mma32(kf, qf[b][ks], s_a[b][t]);
            }
        }
#pragma unroll
        for (int b = 0; b < QB; ++b) {
            float m = s_a[b][0][0];
#pragma unroll
            for (int t = 0; t < NST; ++t)
#pragma unroll
                for (int r = (t == 0 ? 1 : 0); r < 16; ++r) m = fmaxf(m, s_a[b][t][r]);
            mx[b] = fmaxf(m, __shfl_xor(m, 32));
        }
        __syncthreads();  // every wave has read K_0 before the first iteration re-stages its slot
        A64_STAMP(first_tile ? 1 : 5);

        // One KV step = 48 MFMA gaps: 24 of S_{it+1}^T = K_{it+1} Q^T (fragment r = g / 2: k-step r / 2, score tile r % 2; query block g % 2),
        // then 24 of O^T += V_it^T P_it^T (fragment 12 + i: key step i / 3, d tile i % 3).  CUR = it & 1 is a template argument, so every LDS
        // address is (lane-constant register) + immediate.  What rides in the gaps (one wave per SIMD: nothing else hides a bubble,
        // so the vector work is placed, not left to the scheduler):
        //   QK^T gaps 1, 5, 9, 13, 17, 21: the six LDS-DMA requests of K_{it+2} and V_{it+1} (landed long before the closing barrier)
        //   the other 18 QK^T gaps:        one "late" pair of P_it each (pairs E .. 31: 2 v_exp + the v_cvt_pk of the PREVIOUS pair, so
        //                                  no instruction waits for the transcendental in front of it)
        //   PV gaps 0 .. 21:               the 32 v_max3 of the row maxima of S_{it+1} (3 in every third gap, 1 beside the first eight
        //                                  pairs) and the E "early" pairs of P_{it+1}; gaps 22, 23: the two half-wave exchanges of the maxima
        constexpr int E_ = E;
        static_assert(E_ == 14, "the gap tables below place 18 late and 14 early pairs");
        // (NO captures: everything the step touches outside itself comes in as a reference / value parameter under its own name.  A
        //  capturing lambda keeps its ~35 captured addresses in one closure object that every access reads; with four inlined copies of
        //  this body that object had more uses than the optimiser's scalar-replacement pass accepts (1024) - it was left in memory, and
        //  with it every array it points to: the whole kernel state went to scratch.)
#define A64_STATE                                                                                                                        \
    bh, bhn, nit, has_next_tile, qrow0n, kt, vt, nkb, S, k_lane, v_lane, kpad2, vone2, kone, ones, Ks, Vs, wave, hi, issue_q, mx, m_run, o, \
        qf, pw, ka, va
        auto step = [](int it, auto has_next, auto cur_, f32x16 (*s_cur)[NST], f32x16 (*s_nxt)[NST], const int bh, const int bhn, const int nit,
                       const bool has_next_tile, const int qrow0n, const T* kt, const T* vt, const int nkb, const int S, const unsigned k_lane,
                       const unsigned v_lane, const bool kpad2, const bool vone2, const char* kone, const char* ones, char* Ks, char* Vs,
                       const int wave, const int hi, auto& issue_q, float (&mx)[QB], float (&m_run)[QB], f32x16 (&o)[QB][NDT],
                       frag_t (&qf)[QB][NKS], unsigned (&pw)[NP], unsigned (&ka)[NST], unsigned (&va)[NST * 2]) __attribute__((always_inline)) {
            constexpr bool HAS_NEXT = decltype(has_next)::value;
            constexpr int CUR = decltype(cur_)::value;
#ifdef ATTN64_TRACE
            unsigned long long ts_[6] = {};
#endif
            A64_STEP_STAMP(0);
            // the tiles the requests of this step fetch: K block it + 2 and V^T block it + 1 - of the next tile past this one's end (and,
            // on the last tile, of this one again: a request nobody reads, cheaper than a branch in the MFMA stream)
            const bool k_in = it + 2 < nit, v_in = it + 1 < nit;
            const int k_bh = k_in || !has_next_tile ? bh : bhn, k_kb = k_in ? it + 2 : it + 2 - nit;
            const int v_bh = v_in || !has_next_tile ? bh : bhn, v_kb = v_in ? it + 1 : 0;
            const char* k_src = (const char*)kt + ((long long)k_bh * nkb + k_kb) * KBYTES;
            const char* v_src = (const char*)vt + (long long)v_bh * DP * S * ES + (long long)v_kb * 128;
            auto dma = [&](auto n_) __attribute__((always_inline)) {  // request n of the step: 0-2 K passes, 3-5 V^T passes
                constexpr int n = decltype(n_)::value, i = n % 3;
#ifndef ATTN64_DBG_NOLOAD
                // (the lanes of the constant pieces take their 16 bytes from the constants instead: a select on the source address, no
                //  branch - control flow inside the MFMA stream kept the optimiser from dissolving the closures, everything went to scratch)
                if constexpr (n < 3) {
                    const char* src = k_src + i * 4096 + k_lane;
                    if constexpr (i == 2) src = kpad2 ? kone : src;
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Ks + CUR * KBYTES + (wave * 64 + i * 256) * 16), 16, 0, 0);
                } else {
                    const char* src = v_src + (long long)i * 32 * S * ES + v_lane;
                    if constexpr (i == 2) src = vone2 ? ones : src;
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Vs + (CUR ^ 1) * VBYTES + (wave * 64 + i * 256) * 16), 16, 0, 0);
                }
#endif
            };
#ifndef ATTN64_DBG_NOLOAD
            if (has_next_tile && it == nit - 4) issue_q(qrow0n, bhn);
#endif
            A64_STEP_STAMP(1);

            // rare side path (always at the first block): move the reference, rescale O, rewrite -m in Q's padding dims, shift this
            // block's scores in place and redo the early pairs
            if (it == 0 || __any(fmaxf(mx[0], mx[1]) > RESCALE_THR)) {
#pragma unroll
                for (int b = 0; b < QB; ++b) {
                    const float want = m_run[b] + (it == 0 ? mx[b] : fmaxf(mx[b], 0.f));
                    const T m_hi = from_f32<T>(want), m_lo = from_f32<T>(want - to_f32<T>(m_hi));
                    const float m_new = to_f32<T>(m_hi) + to_f32<T>(m_lo);
                    const float delta = m_new - m_run[b];
                    if (it != 0) {
                        // (O was last written by inline-asm MFMAs the compiler's hazard recogniser does not see: a full QK^T phase
                        //  lies between them and this read in program order, the nops make it independent of that)
                        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
                        const float alpha = __builtin_amdgcn_exp2f(-delta);
                        // (through asm on accumulator operands, so that O never has a home in the architectural registers)
#pragma unroll
                        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                float e = o[b][dt][r], tmp;
                                asm volatile("v_accvgpr_read_b32 %1, %0\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 %0, %1"
                                             : "+a"(e), "=&v"(tmp)
                                             : "v"(alpha));
                                o[b][dt][r] = e;
                            }
                    }
                    m_run[b] = m_new;
                    if (hi == HI_P) {
                        qf[b][KS_P][0] = -m_hi;
                        qf[b][KS_P][1] = -m_lo;
                    }
#pragma unroll
                    for (int t = 0; t < NST; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) s_cur[b][t][r] -= delta;
                }
                static_for<0, E_>([&](auto p_) __attribute__((always_inline)) {
                    constexpr int p = decltype(p_)::value, e4 = p & 3, b = (p >> 2) % QB, tj = p / (4 * QB), t = tj >> 1, j = tj & 1, r = 8 * j + 2 * e4;
                    pw[p] = pack2(__builtin_amdgcn_exp2f(s_cur[b][t][r]), __builtin_amdgcn_exp2f(s_cur[b][t][r + 1]), T{});
                });
            }

            constexpr int PRE = ATTN64_PRE;  // fragment reads in flight ahead of the quad of MFMAs that consumes them (even)
            static_assert(PRE % 2 == 0 && PRE >= 2, "fragments are requested and awaited two at a time");
            constexpr int NR = HAS_NEXT ? 24 : 12, R0 = HAS_NEXT ? 0 : 12;
            u32x4 fr[24];
            auto rd = [&fr, &ka, &va](auto i_) __attribute__((always_inline)) {
                constexpr int r = R0 + decltype(i_)::value;
#ifdef ATTN64_DBG_NOLDSREAD  // (probe: wrong results)
                if constexpr (r >= 2) {
                    asm volatile("" : "=v"(fr[r]));
                    return;
                }
#endif
                if constexpr (r < 12)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[r]) : "v"(ka[r % NST]), "n"((CUR ^ 1) * KBYTES + (r / NST) * (KVB * 2 * 16)) : "memory");
                else
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[r]) : "v"(va[(r - 12) / NDT]), "n"(CUR * VBYTES + ((r - 12) % NDT) * 4096) : "memory");
            };
            // ---- the vector micro-operations ----------------------------------------------------------------------------------------
            // a P pair in two halves: X = the two exponentials, C = the pack (issued one pair later, behind the next pair's X)
            float xa[NP], xb[NP];
            auto X = [&](auto p_, f32x16 (*s)[NST]) __attribute__((always_inline)) {
#ifdef ATTN64_DBG_NOSOFTMAX  // (probe: wrong results - P = 1 everywhere)
                xa[decltype(p_)::value] = xb[decltype(p_)::value] = 1.0f;
                return;
#endif
                constexpr int p = decltype(p_)::value, e4 = p & 3, b = (p >> 2) % QB, tj = p / (4 * QB), t = tj >> 1, j = tj & 1, r = 8 * j + 2 * e4;
                xa[p] = __builtin_amdgcn_exp2f(s[b][t][r]);
                xb[p] = __builtin_amdgcn_exp2f(s[b][t][r + 1]);
            };
            unsigned pe[E_];  // the early pairs of P_{it+1}: they replace pw[0 .. E-1] once the PV MFMAs have consumed those
            auto Cw = [&](auto p_) __attribute__((always_inline)) {  // the pack of a pair of P_it
#ifdef ATTN64_DBG_NOSOFTMAX
                pw[decltype(p_)::value] = 0x3F803F80u;
                return;
#endif
                constexpr int p = decltype(p_)::value;
                pw[p] = pack2(xa[p], xb[p], T{});
            };
            auto Ce = [&](auto p_) __attribute__((always_inline)) {  // the pack of an early pair of P_{it+1}
#ifdef ATTN64_DBG_NOSOFTMAX
                pe[decltype(p_)::value] = 0x3F803F80u;
                return;
#endif
                constexpr int p = decltype(p_)::value;
                pe[p] = pack2(xa[p], xb[p], T{});
                // (pinned: the early pairs are only USED on the path that skips the next step's rare branch, and the optimiser sank
                //  their 28 exponentials + 14 packs out of the PV gaps into that path's own block - 250 cycles beside no MFMA)
                asm volatile("" : "+v"(pe[p]));
            };
            float mxn[QB] = {-INFINITY, -INFINITY};
            auto M = [&](auto u_) __attribute__((always_inline)) {  // two more scores into the running maximum of their query block: u = 0 .. 31, tile 0 first
#ifdef ATTN64_DBG_NOSOFTMAX
                mxn[0] = mxn[1] = 0.f;
                return;
#endif
                constexpr int u = decltype(u_)::value, b = u & 1, t = u >> 4, e = ((u >> 1) & 7) * 2;
                mxn[b] = fmaxf(fmaxf(mxn[b], s_nxt[b][t][e]), s_nxt[b][t][e + 1]);
            };
            auto F = [&](auto b_) __attribute__((always_inline)) {  // lanes l and l + 32 hold the two key halves of one query
                constexpr int b = decltype(b_)::value;
                float m = mxn[b], t;
                asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %1, %0\n\tv_max_f32 %0, %0, %1" : "+v"(m), "=&v"(t));
                mx[b] = m;
            };
            A64_STEP_STAMP(2);
            static_for<0, PRE>(rd);
            if constexpr (!HAS_NEXT) {  // the last block of a tile: no scores to build, its late pairs up front
                static_for<E_, NP>([&](auto p_) __attribute__((always_inline)) { X(p_, s_cur); });
                static_for<E_, NP - 1>(Cw);
            }
            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            // (ONE flat loop over the MFMAs, the gap work written out in its body: closures nested deeper than this are not
            //  dissolved by the optimiser any more and every array they reach ends up in scratch memory)
            static_for<0, 2 * NR>([&](auto n_) __attribute__((always_inline)) {
                constexpr int n = decltype(n_)::value, m = n / 2, r = R0 + m, b = n % 2;
                if constexpr (n % 4 == 0) {  // a quad: two fragments, four MFMAs, one wait
                    constexpr int issued = (PRE + m < NR) ? PRE + m : NR;
                    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(issued - m - 2) : "memory");
                }
                if constexpr (r == 12 && b == 0 && HAS_NEXT) A64_STEP_STAMP(3);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (r < 12) {
                    s_nxt[b][r % NST] = mma32(__builtin_bit_cast(frag_t, fr[r]), qf[b][r / NST], r < NST ? zero16 : s_nxt[b][r % NST]);
                } else {
                    // O^T lives in the accumulator half of the register file (the "a" constraint): nothing but MFMAs touches
                    // it in the steady state, and the 256 architectural registers are left to S, Q, P and the fragments
                    constexpr int i = r - 12, tj = i / NDT, w0 = (tj * QB + b) * 4;
                    const u32x4 pf_ = {pw[w0], pw[w0 + 1], pw[w0 + 2], pw[w0 + 3]};
                    if constexpr (std::is_same<T, f16_t>::value)
                        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(o[b][i % NDT]) : "v"(fr[r]), "v"(pf_));
                    else
                        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o[b][i % NDT]) : "v"(fr[r]), "v"(pf_));
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (n % 4 < 2 && PRE + m + n % 4 < NR) rd(std::integral_constant<int, PRE + m + n % 4>{});
                if constexpr (r < 12) {  // ---- the vector work behind QK^T MFMA g ----
                    constexpr int g = 2 * r + b;
                    if constexpr (g % 4 == 1) {
                        dma(std::integral_constant<int, g / 4>{});
                    } else {
                        constexpr int k = g - (g + 2) / 4;  // 0 .. 17: the k-th late pair
                        X(std::integral_constant<int, E_ + k>{}, s_cur);
                        if constexpr (k > 0) Cw(std::integral_constant<int, E_ + k - 1>{});
                    }
                } else {  // ---- behind PV MFMA g (HAS_NEXT only, except the pending pack) ----
                    constexpr int g = 2 * (r - 12) + b;
                    if constexpr (g == 0) Cw(std::integral_constant<int, NP - 1>{});  // (the last late pair; the key step 3 MFMAs consume it)
                    // (the last block of a tile has no QK^T phase: its six requests - the next tile's K_1 and V_0 - ride here)
                    if constexpr (!HAS_NEXT && g % 4 == 1) dma(std::integral_constant<int, g / 4>{});
                    if constexpr (HAS_NEXT) {
                        if constexpr (g < 22) {
                            constexpr int kk = g - (g + 2) / 3;  // early pairs placed in gaps < g (gaps 0, 3, 6 .. carry maxima only)
                            constexpr int nm_before = (g + 2) / 3 * 3 + (kk < 8 ? kk : 8);  // maxima placed in gaps < g
                            if constexpr (g % 3 == 0) {
                                static_for<nm_before, nm_before + 3>(M);
                            } else {
                                if constexpr (kk < 8) M(std::integral_constant<int, nm_before>{});
                                X(std::integral_constant<int, kk>{}, s_nxt);
                                if constexpr (kk > 0) Ce(std::integral_constant<int, kk - 1>{});
                            }
                        } else if constexpr (g == 22) {
                            Ce(std::integral_constant<int, E_ - 1>{});
                            F(std::integral_constant<int, 0>{});
                        } else {
                            F(std::integral_constant<int, 1>{});
                        }
                    }
                }
            });
            if constexpr (HAS_NEXT) {
#pragma unroll
                for (int p = 0; p < E_; ++p) pw[p] = pe[p];
                A64_STEP_STAMP(4);
#ifndef ATTN64_DBG_NOBARRIER
                __syncthreads();
#endif
                A64_STEP_STAMP(5);
#ifdef ATTN64_TRACE
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (blockIdx.x == 0 && threadIdx.x == 0 && 8 + it * 8 + 5 < 4096)
                    for (int i = 0; i < 6; ++i) g_trace[8 + it * 8 + i] = (long long)ts_[i];
#endif
            }
        };
        {
            int it = 0;
            for (; it + 2 < nit; it += 2) {
                step(it, std::true_type{}, std::integral_constant<int, 0>{}, s_a, s_b, A64_STATE);
                step(it + 1, std::true_type{}, std::integral_constant<int, 1>{}, s_b, s_a, A64_STATE);
            }
            step(it, std::true_type{}, std::integral_constant<int, 0>{}, s_a, s_b, A64_STATE);
            step(it + 1, std::false_type{}, std::integral_constant<int, 1>{}, s_b, s_a, A64_STATE);
        }

        A64_STAMP(first_tile ? 2 : 6);
        // ---- normalise and store (as attention.hip: one 16-byte store per pair of 8-column groups) ----------------------------------
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // (inline-asm MFMA results read by VALU below)
        auto o_get = [&](int b, int dt, int r) __attribute__((always_inline)) {
            float x;
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(o[b][dt][r]));
            return x;
        };
#pragma unroll
        for (int b = 0; b < QB; ++b) {
            float l_tot = o_get(b, DT_L, R_L);
            {
                const float other = __shfl_xor(l_tot, 32);
                if (hi != HI_L) l_tot = other;
            }
            const float inv = 1.0f / l_tot;
            T* op = out + (long long)(qrow0 + wave * 64 + b * 32 + lq) * ((long long)H * DH) + (long long)h * DH;
            constexpr int NG = DH / 8;
            u32x2 pk[NG];
#pragma unroll
            for (int k = 0; k < NG; ++k)
                pk[k] = (u32x2){pack2(o_get(b, k >> 2, 4 * (k & 3)) * inv, o_get(b, k >> 2, 4 * (k & 3) + 1) * inv, T{}),
                                pack2(o_get(b, k >> 2, 4 * (k & 3) + 2) * inv, o_get(b, k >> 2, 4 * (k & 3) + 3) * inv, T{})};
#pragma unroll
            for (int k = 0; k + 1 < NG; k += 2) {
                const auto x = __builtin_amdgcn_permlane32_swap(pk[k][0], pk[k + 1][0], false, false);
                const auto y = __builtin_amdgcn_permlane32_swap(pk[k][1], pk[k + 1][1], false, false);
                *(u32x4*)(op + 8 * k + 8 * hi) = (u32x4){x[0], y[0], x[1], y[1]};
            }
            if constexpr (NG & 1) *(u32x2*)(op + 8 * (NG - 1) + 4 * hi) = pk[NG - 1];
        }
        A64_STAMP(first_tile ? 3 : 7);
    }  // tile loop
}

}  // namespace attn64

// scale as launch_attention (0 = q pre-scaled by head_dim^-0.5 * log2(e)); S % 256 == 0, S / 64 even and >= 4
template <typename T>
static int launch_attn64_t(const void* q, const void* kt, const void* vt, void* out, int B, int S, int H, int Dh, float scale, hipStream_t stream) {
    const float c_scale = scale > 0.f ? scale * 1.4426950408889634f : 1.0f;
    const int ntiles = (S / 256) * H * B;
    const int grid = ntiles > 256 ? 256 : ntiles;
    const size_t lds = 2 * (size_t)(12288 + 12288) + (size_t)256 * 96 * 2;
    ProfScope prof(PROF_ATTENTION, stream);
    if (Dh == 88) {
        auto kern = attn64::attn64_kernel<T, 88>;
        static lds_attr_state attr_done;
        HIP_TRY(lds_attr_once(attr_done, kern, (int)lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, (const T*)q, (const T*)kt, (const T*)vt, (T*)out, S, H, c_scale, ntiles);
    } else {
        auto kern = attn64::attn64_kernel<T, 64>;
        static lds_attr_state attr_done;
        HIP_TRY(lds_attr_once(attr_done, kern, (int)lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, (const T*)q, (const T*)kt, (const T*)vt, (T*)out, S, H, c_scale, ntiles);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_attention64(int dtype, const void* q, const void* kt, const void* vt, void* out, int B, int S, int H, int Dh, float scale,
                       hipStream_t stream) {
    L4P_WITH_T16(dtype, T16, return launch_attn64_t<T16>(q, kt, vt, out, B, S, H, Dh, scale, stream));
    return 0;
}
