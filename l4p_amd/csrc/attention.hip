// Flash-style full spatio-temporal self-attention for the VideoMAE encoder blocks
// (reference modeling_finetune.py:169-190: q*scale @ k^T -> softmax -> @ v, 16 heads x 88 dims,
// 2048 tokens per window).  Never materialises the S x S score matrix.
//
// Inputs come straight from the QKV GEMM epilogue (gemm.hpp, EPI_QKV):
//   q  : [B*S][H*DP]            head dim zero-padded 88 -> DP = 96
//   kt : K in LDS tile order    8-element groups [b][h][S/KVB][DP/16][KVB][half ^ ((key>>3)&1)]
//   vt : [B][H][DP][S]          V transposed, so a PV operand fragment is 8 consecutive keys
// Output: out[B*S][H*DH] (token-major, head-major columns = transpose(1,2).reshape of the reference).
//
// CDNA4 mapping (32x32 MFMA, one wave = 32 query rows, 4 waves per workgroup):
//  * K / V^T tiles are staged by LDS-DMA (global_load_lds, 16 B per lane): no VGPR round trip, no
//    ds_write.  The K tile is a linear 12 KB copy (its swizzle was applied by the producer); the V^T
//    tile gets the XOR swizzle through the per-lane SOURCE address.  Both fragment reads are
//    conflict-free ds_read_b128.  LDS double-buffered, one barrier per KV block.
//  * S^T = K Q^T ("swapped" operands): each lane owns ONE query column, so row max and rescale are
//    lane-local plus a single lane<->lane+32 exchange.
//  * K rows are read with bits 2/3 of the row index swapped, which makes the 8 scores a lane holds
//    for a k-step 8 CONSECUTIVE keys: P goes from the S accumulator straight into the PV operand.
//  * O^T = V^T P^T keeps the output accumulator column = query, so the online-softmax rescale is a
//    plain per-lane multiply — and it is skipped (wave-uniformly) whenever no row maximum moved.
//  * the softmax denominator comes out of the MFMA: padding row d = DH of the V^T tile is sourced
//    from a constant "ones" chunk, so O^T[DH][q] = sum_k P[q][k] with exactly the weights used for O.
//  * exp via v_exp_f32 (exp2 of non-positive arguments), scale*log2(e) folded into one fma.
//  * the KV loop is software pipelined (S of block j+1 is built while P of block j is formed; the two score register
//    sets are used ping-pong) and, for bf16, hand scheduled (template flag HS: inline-asm fragment reads with counted
//    lgkmcnt, the vector work sliced behind the MFMAs) - see the comments at `step` below.
#include <cstdlib>
#include <type_traits>

#include "common.hpp"

int launch_attention64(int dtype, const void* q, const void* kt, const void* vt, void* out, int B, int S, int H, int Dh, float scale,
                       hipStream_t stream);  // attention64.hip

__device__ __attribute__((aligned(16))) static const unsigned short g_ones_bf16[8] = {0x3F80, 0x3F80, 0x3F80, 0x3F80,
                                                                                         0x3F80, 0x3F80, 0x3F80, 0x3F80};
__device__ __attribute__((aligned(16))) static const unsigned short g_ones_f16[8] = {0x3C00, 0x3C00, 0x3C00, 0x3C00,
                                                                                        0x3C00, 0x3C00, 0x3C00, 0x3C00};
__device__ __attribute__((aligned(16))) static const float g_ones_f32[4] = {1.f, 1.f, 1.f, 1.f};
// K-tile chunk holding head dims DH .. DH+7 (all padding) in the deferred-maximum form: dims DH and DH+1 read as 1.0, so
// the two bf16 halves of -m that sit in the same dims of Q are added to every score by the QK^T MFMA itself
__device__ __attribute__((aligned(16))) static const unsigned short g_kone_bf16[8] = {0x3F80, 0x3F80, 0, 0, 0, 0, 0, 0};
__device__ __attribute__((aligned(16))) static const unsigned short g_kone_f16[8] = {0x3C00, 0x3C00, 0, 0, 0, 0, 0, 0};

// SPLIT = 2: the workgroup has 8 waves; waves 0-3 walk the even KV blocks and waves 4-7 the odd ones for the
// SAME 128 query rows, and the two partial (m, O, denominator) states are merged through LDS at the end.
// Used when the launch has too few workgroups to put two of them on a CU (batch 1: 256 workgroups), so
// every SIMD still holds two waves whose MFMA / VALU / wait phases overlap.
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// QS = 2 ("query split", batch >= 4): the workgroup has 8 waves that cover 256 query rows (waves 0-3 the first 128, waves
// 4-7 the second) of ONE (batch, head) and SHARE one K / V^T tile ring: a tile is fetched and written into LDS once per 256
// query rows instead of once per 128 (half the L2 -> LDS bytes and LDS-DMA writes per MFMA), every wave issues 3 LDS-DMA
// requests per KV block instead of 6, and a batch-4 launch is ONE round of 512 workgroups (two per CU, four waves per SIMD).
// HS (bf16 only): the KV-block body is hand scheduled - see the comment at its definition below.
template <typename T, int DP, int KVB, int DH, int SPLIT, bool HS = false, int QS = 1>
__global__ __launch_bounds__(256 * SPLIT * QS) void attn_kernel(const T* __restrict__ q, const T* __restrict__ kt,
                                                           const T* __restrict__ vt, T* __restrict__ out, int S, int H,
                                                           float c_scale, int ntiles) {
    typedef typename Frag<T>::type frag_t;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    constexpr int ES = sizeof(T);
    constexpr int EPC = 16 / ES;
    constexpr int CPF = 8 * ES / 16;       // 16-byte chunks per 8-element fragment
    constexpr int NKS = DP / 16;           // k-steps of the QK^T contraction
    constexpr int NST = KVB / 32;          // 32-key score tiles per KV block
    constexpr int NDT = DP / 32;           // 32-wide output d tiles
    constexpr int KBYTES = NKS * KVB * 2 * 8 * ES;  // 12 KB
    constexpr int VBYTES = DP * 128;                // KVB * ES == 128: 12 KB
    static_assert(KVB * ES == 128, "V^T tile rows are 128 bytes");
    static_assert(KBYTES % 4096 == 0 && VBYTES % 4096 == 0, "tile = whole LDS-DMA passes of 256 lanes");
    static_assert(DH < DP && DH % 4 == 0, "one padding row carries the denominator");
    constexpr int K_IT = KBYTES / 4096, V_IT = VBYTES / 4096;
    // DEFER: the hand-scheduled body keeps a deferred running maximum (see `step`)
    constexpr bool DEFER = HS;
    constexpr int DT_L = DH / 32, I_L = DH % 32;                       // where the denominator row lands in O^T
    constexpr int HI_L = (I_L >> 2) & 1, R_L = (I_L & 3) + 4 * (I_L >> 3);

    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(SPLIT == 1 || QS == 1, "KV split and query split are alternatives");
    const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);  // KV-split / query-split group of this wave
    const int kgrp = QS > 1 ? 0 : grp;                                  // KV blocks walked: every one (QS) or every SPLIT-th
    char* Ks = smem + kgrp * 2 * (KBYTES + VBYTES);  // [2][KBYTES]   (per KV-split group; shared by the query-split groups)
    char* Vs = Ks + 2 * KBYTES;                      // [2][VBYTES]
    // PERSIST (no KV split): a workgroup walks tiles blockIdx.x, + gridDim.x, ...; the NEXT tile's K_0 / K_1 / V_0 ride the rings
    // through the seam (requested by the last two iterations of the current tile) and its Q rows are fetched into Qs by
    // LDS-DMA a few iterations before the end, so a seam costs one barrier instead of a round trip to HBM with every
    // workgroup of the chip fetching its prologue at the same time (measured: ~3.8 us per seam at batch 4).
#ifdef ATTN_NO_PERSIST
    constexpr bool PERSIST = false;
#else
    // (Only the 8-wave query-split form: the 4-wave form would need 276 registers, i.e. one wave per SIMD.  -DATTN_SEAM_STEP
    //  goes one step further - the last step of a tile builds the next tile's first scores from Q read out of Qs, so a seam is
    //  the output store only - but it needs ~10 registers more than the 256 two waves per SIMD leave: hipcc spills pointers
    //  to scratch inside the steady-state loop; kept for a future hand-allocated form.)
    constexpr bool PERSIST = SPLIT == 1 && QS > 1;
#endif
    constexpr int QROWS = 128 * QS, QBYTES = QROWS * DP * ES;
    char* Qs = smem + 2 * (KBYTES + VBYTES);         // [QROWS][DP] (PERSIST only)

    const int tid = threadIdx.x & 255, lane = tid & 63;  // thread / wave index inside the group
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, hi = lane >> 5;
    // XCD-aware work mapping: workgroup L runs on XCD L % 8 (observed dispatch order; speed only, not
    // correctness).  All S/128 query blocks of one (batch, head) are given to the SAME XCD so that head's
    // K / V^T (786 KB) is fetched into one L2 instead of eight.
    const int nqb = S / (128 * QS), units = ntiles / nqb;  // units = B * H
    auto decode = [&](int v, int& bh_, int& qrow0_) {  // tile v -> (batch * H + head, first query row of the tile in its batch item)
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
        qrow0_ = (unit / H) * S + qb * (128 * QS);  // row of q / out: batch * S + token
    };
    const int qloc = (QS > 1 ? grp * 128 : 0) + wave * 32 + lq;  // this lane's query row inside the tile
    const int nkb = S / KVB;

    // ---- LDS-DMA sources: lane-constant parts (the tile adds batch-head offsets koff / voff, wave-uniform) ----------------
    const char* kbase = (const char*)kt + tid * 16;  // + kb*KBYTES + i*4096
    // V^T: LDS position (row d, slot) <- chunk slot ^ ((d >> 1) & 7) of that row
    const int vrow0 = tid >> 3, vslot = tid & 7;
    const int vchunk = vslot ^ ((vrow0 >> 1) & 7);  // (d >> 1) & 7 is the same for d = vrow0 + 32*i
    const char* vsrc[V_IT];
    bool vones[V_IT];
#pragma unroll
    for (int i = 0; i < V_IT; ++i) {
        const int d = vrow0 + 32 * i;
        vones[i] = d == DH;
        vsrc[i] = (const char*)(vt + (long long)d * S) + vchunk * 16;  // + voff + kb*128
    }
    const char* ones = std::is_same<T, f16_t>::value ? (const char*)g_ones_f16 : ES == 2 ? (const char*)g_ones_bf16 : (const char*)g_ones_f32;
    // chunk c = tid + 256 i of a K tile is (k-step c / (2 KVB), key (c % (2 KVB)) / 2, half (c & 1) ^ ((key >> 3) & 1))
    bool kpad[K_IT];
#pragma unroll
    for (int i = 0; i < K_IT; ++i) {
        const int c = tid + 256 * i, key = (c % (2 * KVB)) >> 1;
        kpad[i] = DEFER && (c / (2 * KVB)) * 16 + (((c & 1) ^ ((key >> 3) & 1)) << 3) == DH;
    }
    const char* kone = std::is_same<T, f16_t>::value ? (const char*)g_kone_f16 : (const char*)g_kone_bf16;
    // QS: the 768 chunks of a tile are spread over 512 lanes: chunk gt (all lanes) + one more for half of the waves (K: chunk
    // 512 + gt from waves 0-3; V^T: chunk gt - 256 from waves 4-7, its first pass being chunks 256 + gt) - 3 requests per wave
    const int gt = threadIdx.x, gwave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* qk_src[2];
    const char* qv_src[2];
    bool qk_pad[2], qv_ones[2];
    if constexpr (QS > 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = i == 0 ? gt : 512 + (gt & 255), key = (c % (2 * KVB)) >> 1;
            qk_pad[i] = DEFER && (c / (2 * KVB)) * 16 + (((c & 1) ^ ((key >> 3) & 1)) << 3) == DH;
            qk_src[i] = (const char*)kt + c * 16;
            const int cv = i == 0 ? 256 + gt : (gt & 255), d = cv >> 3, slot = cv & 7;
            qv_ones[i] = d == DH;
            qv_src[i] = (const char*)(vt + (long long)d * S) + (slot ^ ((d >> 1) & 7)) * 16;
        }
    }
    // (tko / tvo: byte offset of the tile's (batch, head) in kt / vt relative to the per-lane pointers, which are those of
    //  the CURRENT tile: 0 for it, the distance to the next tile's head for the prefetch across the seam)
    auto issue_k = [&](long long tko, int kb, int buf) {
        if constexpr (QS > 1) {
            const char* s0 = qk_pad[0] ? kone : qk_src[0] + tko + (long long)kb * KBYTES;
            __builtin_amdgcn_global_load_lds((gptr_t)s0, (lptr_t)(Ks + buf * KBYTES + (gwave * 64) * 16), 16, 0, 0);
            if (gwave < 4) {
                const char* s1 = qk_pad[1] ? kone : qk_src[1] + tko + (long long)kb * KBYTES;
                __builtin_amdgcn_global_load_lds((gptr_t)s1, (lptr_t)(Ks + buf * KBYTES + (512 + gwave * 64) * 16), 16, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < K_IT; ++i) {
            const char* src = kpad[i] ? kone : kbase + tko + (long long)kb * KBYTES + i * 4096;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Ks + buf * KBYTES + (wave * 64 + i * 256) * 16), 16, 0, 0);
        }
    };
    auto issue_v = [&](long long tvo, int kb, int buf) {
        if constexpr (QS > 1) {
            const char* s0 = qv_ones[0] ? ones : qv_src[0] + tvo + (long long)kb * 128;
            __builtin_amdgcn_global_load_lds((gptr_t)s0, (lptr_t)(Vs + buf * VBYTES + (256 + gwave * 64) * 16), 16, 0, 0);
            if (gwave >= 4) {
                const char* s1 = qv_ones[1] ? ones : qv_src[1] + tvo + (long long)kb * 128;
                __builtin_amdgcn_global_load_lds((gptr_t)s1, (lptr_t)(Vs + buf * VBYTES + ((gwave - 4) * 64) * 16), 16, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < V_IT; ++i) {
            const char* src = vones[i] ? ones : vsrc[i] + tvo + (long long)kb * 128;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Vs + buf * VBYTES + (wave * 64 + i * 256) * 16), 16, 0, 0);
        }
    };
    // the next tile's Q rows -> Qs, a linear [QROWS][DP] image: chunk c = pass * NT + thread is (row c / CPR, 16-byte piece c % CPR)
    auto issue_q = [&](int qrow0n, int bhn) {
        constexpr int CPR = DP * ES / 16, NT = 256 * QS;
        static_assert((QROWS * CPR) % NT == 0, "whole LDS-DMA passes");
        const char* qb_ = (const char*)(q + (long long)qrow0n * ((long long)H * DP) + (long long)(bhn % H) * DP);
        // (an opaque zero: the per-lane address arithmetic below is invariant in the KV loop, and hoisted out of it, it would
        //  hold ~25 registers through the whole tile - 231 + 25 no longer fits two waves per SIMD)
        int z;
        asm volatile("v_mov_b32 %0, 0" : "=v"(z));
        const int t_ = (int)threadIdx.x + z;
#pragma unroll
        for (int i = 0; i < QROWS * CPR / NT; ++i) {
            const int c = i * NT + t_, row = c / CPR, pc = c - row * CPR;
            __builtin_amdgcn_global_load_lds((gptr_t)(qb_ + (long long)row * ((long long)H * DP * ES) + pc * 16),
                                             (lptr_t)(Qs + (i * NT + (int)(threadIdx.x >> 6) * 64) * 16), 16, 0, 0);
        }
    };

    // ================================ tile loop ================================
    int bh_at = 0;  // the (batch, head) the per-lane source pointers currently point at
    // State that crosses a tile seam (PERSIST): the LAST pipeline step of a tile already builds the first scores of the next
    // tile (its Q fragments are read from Qs into qf once the current tile's last QK^T is done, the K ring already holds the
    // next K_0), so a seam is the output store of the finished tile and nothing else.
    frag_t qf[NKS];             // Q fragments (B operand: column = query, 8 consecutive d per k-step half)
    f32x16 s_a[NST], s_b[NST];  // score registers, used ping-pong (no copy between pipeline steps)
    float mx = 0.f;             // row maximum of the scores the next step consumes
    auto q_from_lds = [&]() {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            u32x4* d = (u32x4*)&qf[ks];
#pragma unroll
            for (int c = 0; c < CPF; ++c) d[c] = *(const u32x4*)(Qs + (qloc * DP + ks * 16 + hi * 8 + c * EPC) * ES);
        }
    };
    // HS: the softmax scale (times log2 e) is folded into Q once (the reference scales q before q k^T as well,
    // modeling_finetune.py:180), so the scores come out of the MFMA in the exp2 domain
    auto q_prescale = [&]() {
        if constexpr (DEFER) {
            if (c_scale != 1.0f) {  // (pre-scaled q: nothing to do, and no second rounding)
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                    for (int e = 0; e < 8; ++e) qf[ks][e] = from_f32<T>(to_f32<T>(qf[ks][e]) * c_scale);
            }
        }
    };
    for (int v = blockIdx.x; v < ntiles; v += PERSIST ? (int)gridDim.x : ntiles) {
    int bh, qrow0;
    decode(v, bh, qrow0);
    const int vn = v + (int)gridDim.x;
    const bool has_next_tile = PERSIST && vn < ntiles;
    int bhn = 0, qrow0n = 0;
    if (has_next_tile) decode(vn, bhn, qrow0n);
    bh = __builtin_amdgcn_readfirstlane(bh), qrow0 = __builtin_amdgcn_readfirstlane(qrow0);  // (wave-uniform: keep them scalar)
    bhn = __builtin_amdgcn_readfirstlane(bhn), qrow0n = __builtin_amdgcn_readfirstlane(qrow0n);
    // the per-lane source pointers are moved to this tile's (batch, head); the next tile is addressed relative to them
    {
        const long long dk = (long long)(bh - bh_at) * nkb * KBYTES, dv = (long long)(bh - bh_at) * DP * S * ES;
        kbase += dk;
#pragma unroll
        for (int i = 0; i < V_IT; ++i) vsrc[i] += dv;
        if constexpr (QS > 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i) qk_src[i] += dk, qv_src[i] += dv;
        }
        bh_at = bh;
    }
    const long long tk = 0, tv = 0;                                                                  // this tile's K / V^T
    const long long tkn = (long long)(bhn - bh) * nkb * KBYTES, tvn = (long long)(bhn - bh) * DP * S * ES;  // the next tile's
    // later tiles: K_0, K_1, V_0 were requested and Q, the first scores and their row maximum left by the previous tile's last step
    const bool first_tile = !PERSIST || v == (int)blockIdx.x;
    const int q_row = qrow0 + qloc;                // row of q / out (batch * S + token)
    const int h = bh % H;
    const int nit = nkb / SPLIT;
    if (first_tile) {
        issue_k(tk, kgrp, 0);
        issue_v(tv, kgrp, 0);
        if (nit > 1) issue_k(tk, SPLIT + kgrp, 1);
        const T* qp = q + (long long)q_row * ((long long)H * DP) + (long long)h * DP;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            u32x4* d = (u32x4*)&qf[ks];
#pragma unroll
            for (int c = 0; c < CPF; ++c) d[c] = *(const u32x4*)(qp + ks * 16 + hi * 8 + c * EPC);
        }
        q_prescale();
    }

    f32x16 o[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    // DEFER: scores are kept RELATIVE to the running reference m_run.  -m_run rides in the head-dim padding: Q carries it
    // (as two bf16 halves, 16 mantissa bits) in dims DH, DH+1, where the K tile reads 1.0 (g_kone_bf16, substituted by the
    // tile DMA below), so S = K Q^T comes out of the MFMA as score - m_run and P = exp2(S) needs no subtraction.  m_run
    // starts at 0 and is set by the first block.
    float m_run = DEFER ? 0.f : -INFINITY;
    constexpr int KS_P = DH / 16, HI_P = (DH / 8) & 1;  // k-step / lane half whose Q fragment holds dims DH .. DH+7
    static_assert(DH % 8 == 0, "the first padding chunk starts at DH");

    // K tile row read by this lane for score-tile row i = lq: swap bits 2 and 3
    const int krow = (lq & ~12) | ((lq & 4) << 1) | ((lq & 8) >> 1);
    // byte offsets of this lane's fragments inside the tiles
    int koff[NST];
#pragma unroll
    for (int t = 0; t < NST; ++t) {
        const int key = t * 32 + krow;
        koff[t] = (key * 2 + (hi ^ ((key >> 3) & 1))) * 8 * ES;  // + ks * KVB * 2 * 8 * ES
    }
    int voff[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) voff[dt] = (dt * 32 + lq) * 128;
    const int vsw = (lq >> 1) & 7;  // ((dt*32 + lq) >> 1) & 7

    // S^T = K Q^T for one KV block: rows = keys (permuted), cols = queries
    auto qk_tile = [&](const char* Kb, f32x16* s) {
#pragma unroll
        for (int t = 0; t < NST; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                frag_t kf;
                u32x4* d = (u32x4*)&kf;
#pragma unroll
                for (int c = 0; c < CPF; ++c) d[c] = *(const u32x4*)(Kb + ks * (KVB * 2 * 8 * ES) + koff[t] + c * 16);
                s[t] = mma32(kf, qf[ks], s[t]);
            }
        }
    };
    auto row_max = [&](const f32x16* s) {
        float mx = s[0][0];
#pragma unroll
        for (int t = 0; t < NST; ++t)
#pragma unroll
            for (int r = (t == 0 ? 1 : 0); r < 16; ++r) mx = fmaxf(mx, s[t][r]);
        return fmaxf(mx, __shfl_xor(mx, 32));
    };

    // Software pipeline over this group's KV blocks (block j of the group = global block j*SPLIT + grp):
    //   iteration j:  [ S_{j+1} = K_{j+1} Q^T  (MFMA)  ||  P_j = exp2(S_j*c - m)  (VALU) ]
    //                 [ O^T += V_j^T P_j^T     (MFMA)  ||  row max of S_{j+1}     (VALU) ]
    // so each MFMA batch has independent VALU work to issue under it.  K runs one block ahead of V in the LDS rings.
#ifdef ATTN_SEAM_STEP
    if (first_tile)  // (later tiles: s_a and mx come out of the previous tile's last step)
#endif
    {
        __syncthreads();  // K_0 / V_0 / K_1 (and, past the first tile, Q in Qs) have landed; the previous tile's reads are done
#ifndef ATTN_SEAM_STEP
        if (!first_tile) {
            q_from_lds();
            q_prescale();
        }
#endif
        qk_tile(Ks, s_a);
        mx = row_max(s_a);
        __syncthreads();  // every wave has read K_0 before the first iteration re-stages its slot
    }

    // one pipeline step; HAS_NEXT is a compile-time flag so that the steady-state body is ONE basic block in which the
    // scheduler is free to interleave the two MFMA batches with the VALU work (the last block is peeled)
    // HS, deferred maximum: the reference m_run only has to keep the exponents in range (P is rounded to bf16, whose
    // relative precision does not depend on magnitude, and the denominator is accumulated from the same rounded P), so it
    // is moved — O rescaled, Q's padding dims rewritten, this block's scores shifted in place — only when some row's block
    // maximum exceeds it by more than RESCALE_THR (P <= 2^THR otherwise), and always at the first block.  The block body
    // has no multiply-add in front of the exponential.
    constexpr float RESCALE_THR = 8.f;
    // SEAM (compile time): the last step of a tile that has a successor - its "next" scores are the next tile's first ones
    auto step = [&](int it, auto has_next, auto seam_, f32x16* s_cur, f32x16* s_nxt) __attribute__((always_inline)) {
        constexpr bool HAS_NEXT = decltype(has_next)::value, SEAM = decltype(seam_)::value;
        const int cur = it & 1;
#ifndef ATTN_DBG_NOLOAD  // (tools/probes/attn_variants.hip: compute-only timing)
#ifdef ATTN_DMA_IN_SLOTS_ALL  // (probe)
        constexpr bool DMA_IN_SLOTS = HS;
#else
        constexpr bool DMA_IN_SLOTS = HS && SPLIT > 1;  // measured: +4 % for the KV-split (batch 1) form, -2 % otherwise
#endif
        if (!DMA_IN_SLOTS) {
            // (K_{it} in ring slot cur and V_{it-1} in slot cur^1 were consumed last iteration)  Past the end of this tile the
            // rings carry on with the next tile's first blocks (nit is even: its block j lands in slot j & 1 as well)
            if (it + 2 < nit) issue_k(tk, (it + 2) * SPLIT + kgrp, cur);
            else if (has_next_tile) issue_k(tkn, it + 2 - nit, cur);
            if (it + 1 < nit) issue_v(tv, (it + 1) * SPLIT + kgrp, cur ^ 1);
            else if (has_next_tile) issue_v(tvn, 0, cur ^ 1);
            if (PERSIST && has_next_tile && it == nit - 4) issue_q(qrow0n, bhn);
        }
#endif
#ifdef ATTN_DBG_NOCOMPUTE  // (probe: staging-only timing)
        __syncthreads();
        return;
#endif
        if constexpr (DEFER) {
            // (mx = row maximum of s_cur, relative to m_run)  The rare side path: move the reference, rescale O, rewrite
            // -m_run in Q's padding dims and shift this block's scores in place; the block body below is the same either way.
            if (it == 0 || __any(mx > RESCALE_THR)) {
                const float want = m_run + (it == 0 ? mx : fmaxf(mx, 0.f));
                const T m_hi = from_f32<T>(want), m_lo = from_f32<T>(want - to_f32<T>(m_hi));
                const float m_new = to_f32<T>(m_hi) + to_f32<T>(m_lo);  // the reference the MFMA will really subtract
                const float delta = m_new - m_run;
                if (it != 0) {  // (at the first block O is still zero, and exp2(-delta) may overflow for very negative scores)
                    const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
                }
                m_run = m_new;
                if (hi == HI_P) {
                    qf[KS_P][0] = -m_hi;
                    qf[KS_P][1] = -m_lo;
                }
#pragma unroll
                for (int t = 0; t < NST; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s_cur[t][r] -= delta;
            }
        } else {
            // ---- running max / rescale (lane-local; skipped wave-uniformly when no row maximum moved) -----
            const float m_new = fmaxf(m_run, mx * c_scale);
            if (__any(m_new > m_run)) {
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
                m_run = m_new;
            }
        }
        if constexpr (HS) {
            // Hand-scheduled block body.  hipcc emits lgkmcnt(0) for every LDS wait of its own while an LDS-DMA is in
            // flight, which makes each MFMA wait for its own ds_read; and it clusters the VALU work away from the MFMAs.
            // Here the 24 fragment reads (12 K for S_{j+1} = K Q^T, then 12 V^T for O^T += V^T P^T) are inline-asm
            // ds_read_b128 with hand-counted lgkmcnt (LDS reads return in order: MFMA m may issue once only the reads
            // requested after its operand are outstanding); PRE reads run ahead.  Behind every MFMA sits a slice of the
            // block's VALU work: the exp2 / bf16 packing of P_j under the QK MFMAs, the row max of S_{j+1} under the PV ones.
            typedef __attribute__((ext_vector_type(2))) float f32x2;
#ifndef ATTN_HS_PRE
#define ATTN_HS_PRE 6
#endif
            constexpr int PRE = ATTN_HS_PRE;  // fragment reads in flight ahead of the MFMA that consumes them
            constexpr int NR = HAS_NEXT ? 24 : 12, R0 = HAS_NEXT ? 0 : 12;  // the last block has no next scores to build
            const unsigned lds_k = (unsigned)(size_t)(__attribute__((address_space(3))) char*)Ks;
            const unsigned lds_v = (unsigned)(size_t)(__attribute__((address_space(3))) char*)Vs;
            const unsigned kb = lds_k + (cur ^ 1) * KBYTES, vb = lds_v + cur * VBYTES;
            unsigned ka[NST], va[NST * 2];
#pragma unroll
            for (int t = 0; t < NST; ++t) ka[t] = kb + koff[t];
#pragma unroll
            for (int tj = 0; tj < NST * 2; ++tj) va[tj] = vb + lq * 128 + ((((tj * 16 + hi * 8) >> 3) ^ vsw) << 4);
            u32x4 fr[24];
            auto rd = [&fr, &ka, &va](auto i_) {  // (explicit captures: asm operands do not trigger implicit capture)
                constexpr int r = R0 + decltype(i_)::value;
#ifdef ATTN_DBG_NOLDSREAD
                if constexpr (r >= 2) {
                    asm volatile("" : "=v"(fr[r]));
                    return;
                }
#endif
                if constexpr (r < 12)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[r]) : "v"(ka[r % NST]), "n"((r / NST) * (KVB * 2 * 16)) : "memory");
                else
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[r]) : "v"(va[(r - 12) / NDT]), "n"(((r - 12) % NDT) * 4096) : "memory");
            };
            const f32x2 c2 = {c_scale, c_scale}, m2 = {-m_run, -m_run};
            frag_t pf[NST][2];
#ifdef ATTN_DBG_NOSOFTMAX  // (tools/probes: ablation timing, wrong results)
#pragma unroll
            for (int t = 0; t < NST; ++t)
#pragma unroll
                for (int j = 0; j < 2; ++j) asm volatile("" : "=v"(pf[t][j]));
#endif
            auto exp_pair = [&](auto p_) {  // P elements 2p, 2p+1 of the 32 scores of this lane
                constexpr int e = decltype(p_)::value * 2, t = e / 16, r = e % 16, j = r / 8, ee = r % 8;
#ifdef ATTN_DBG_NOSOFTMAX
                return;
#endif
                f32x2 x = {s_cur[t][r], s_cur[t][r + 1]};
                if constexpr (!DEFER) x = __builtin_elementwise_fma(x, c2, m2);
#ifdef ATTN_DBG_NOEXP  // (tools/probes: ablation timing, wrong results)
                pf[t][j][ee] = from_f32<T>(x[0]);
                pf[t][j][ee + 1] = from_f32<T>(x[1]);
#else
                pf[t][j][ee] = from_f32<T>(__builtin_amdgcn_exp2f(x[0]));
                pf[t][j][ee + 1] = from_f32<T>(__builtin_amdgcn_exp2f(x[1]));
#endif
            };
            float mxn = -INFINITY;
            auto max_pair = [&](auto p_) {
                constexpr int e = decltype(p_)::value * 2, t = e / 16, r = e % 16;
#ifndef ATTN_DBG_NOMAX
                mxn = fmaxf(fmaxf(mxn, s_nxt[t][r]), s_nxt[t][r + 1]);
#endif
            };
            if constexpr (SEAM) {  // the current tile's last QK^T is done: qf takes the next tile's Q (prefetched into Qs)
                q_from_lds();
                q_prescale();
            }
            static_for<0, PRE>(rd);
            if constexpr (!HAS_NEXT) static_for<0, 16>(exp_pair);
            __builtin_amdgcn_s_setprio(1);  // the MFMA run outranks the co-resident waves' vector work (+4 % at batch 4)
            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            static_for<0, NR>([&](auto m_) {
                constexpr int m = decltype(m_)::value, r = R0 + m;
                constexpr int issued = (PRE + m < NR) ? PRE + m : NR;
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(issued - m - 1) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (r < 12) {
                    s_nxt[r % NST] = mma32(__builtin_bit_cast(frag_t, fr[r]), qf[r / NST], r < NST ? zero16 : s_nxt[r % NST]);
                } else {
                    constexpr int i = r - 12;
                    o[i % NDT] = mma32(__builtin_bit_cast(frag_t, fr[r]), pf[(i / NDT) >> 1][(i / NDT) & 1], o[i % NDT]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (PRE + m < NR) rd(std::integral_constant<int, PRE + m>{});
                // KV-split form: one LDS-DMA request of the tiles the NEXT iterations read behind each of the first MFMAs
                // instead of a block of six at the top of the iteration
                if constexpr (!DMA_IN_SLOTS) {
                } else if constexpr (m < K_IT) {
                    if (it + 2 < nit) {
                        const char* src = kpad[m < K_IT ? m : 0] ? kone : kbase + tk + (long long)((it + 2) * SPLIT + grp) * KBYTES + m * 4096;
                        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Ks + cur * KBYTES + (wave * 64 + m * 256) * 16), 16, 0, 0);
                    }
                } else if constexpr (m < K_IT + V_IT && HAS_NEXT) {
                    constexpr int i = m - K_IT;
                    const char* src = vones[i] ? ones : vsrc[i] + tv + (long long)((it + 1) * SPLIT + grp) * 128;
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Vs + (cur ^ 1) * VBYTES + (wave * 64 + i * 256) * 16), 16, 0, 0);
                }
                // VALU slice riding behind this MFMA: 16 pairs over 12 slots (2 pairs in the first four, then 1)
                constexpr int slot = r < 12 ? r : r - 12;
                constexpr int p0 = slot < 4 ? 2 * slot : 4 + slot, np = slot < 4 ? 2 : 1;
                if constexpr (r < 12) {
                    static_for<p0, p0 + np>(exp_pair);
                } else if constexpr (HAS_NEXT) {
                    static_for<p0, p0 + np>(max_pair);
                }
            });
            __builtin_amdgcn_s_setprio(0);
            if (HAS_NEXT) {
                {
                    // lanes l and l + 32 hold the two key halves of one query: v_permlane32_swap exchanges them without the
                    // LDS round trip (and the lgkmcnt(0) wait) of ds_bpermute: [0] = (own | lower), [1] = (upper | own)
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mxn), __float_as_uint(mxn), false, false);
                    mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
                }
#ifndef ATTN_DBG_NOBARRIER
                __syncthreads();
#endif
            }
            return;
        }
        // ---- next block's scores (MFMA) alongside this block's exponentials (VALU) ----------------
        if constexpr (SEAM) {  // (see the hand-scheduled body)
            q_from_lds();
            q_prescale();
        }
        if (HAS_NEXT) qk_tile(Ks + (cur ^ 1) * KBYTES, s_nxt);
        frag_t pf[NST][2];
        {
            typedef __attribute__((ext_vector_type(2))) float f32x2;
            const f32x2 c2 = {c_scale, c_scale}, m2 = {-m_run, -m_run};
#pragma unroll
            for (int t = 0; t < NST; ++t)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {  // two scores per v_pk_fma_f32
                        const f32x2 a = {s_cur[t][8 * j + e], s_cur[t][8 * j + e + 1]};
                        const f32x2 x = __builtin_elementwise_fma(a, c2, m2);
                        pf[t][j][e] = from_f32<T>(__builtin_amdgcn_exp2f(x[0]));
                        pf[t][j][e + 1] = from_f32<T>(__builtin_amdgcn_exp2f(x[1]));
                    }
        }
        // ---- O^T += V^T P^T (MFMA) alongside the row max of the next scores (VALU) ------------------
        const char* Vb = Vs + cur * VBYTES;
#pragma unroll
        for (int t = 0; t < NST; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c0 = ((t * 32 + j * 16 + hi * 8) * ES) >> 4;  // first 16-byte chunk of the 8 keys
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    frag_t vf;
                    u32x4* d = (u32x4*)&vf;
#pragma unroll
                    for (int c = 0; c < CPF; ++c) d[c] = *(const u32x4*)(Vb + voff[dt] + (((c0 + c) ^ vsw) << 4));
                    o[dt] = mma32(vf, pf[t][j], o[dt]);
                }
            }
        if (HAS_NEXT) {
            mx = row_max(s_nxt);
            __syncthreads();  // (drains the LDS-DMA issued at the top: vmcnt(0) + barrier; a 3-deep ring with a counted
                              //  vmcnt was measured and is no faster: the DMA latency is not what bounds the loop)
        }
    };
    {
        int it = 0;
        for (; it + 2 < nit; it += 2) {
            step(it, std::true_type{}, std::false_type{}, s_a, s_b);
            step(it + 1, std::true_type{}, std::false_type{}, s_b, s_a);
        }
        if (it + 2 == nit) {
            step(it, std::true_type{}, std::false_type{}, s_a, s_b);
#ifdef ATTN_SEAM_STEP  // (experiment: the last step of a tile also builds the next tile's first scores - see the note at PERSIST)
            if (PERSIST && has_next_tile)
                step(it + 1, std::true_type{}, std::integral_constant<bool, PERSIST>{}, s_b, s_a);  // leaves the next tile's first scores in s_a
            else
#endif
                step(it + 1, std::false_type{}, std::false_type{}, s_b, s_a);
        } else {
            step(it, std::false_type{}, std::false_type{}, s_a, s_b);  // (a single KV block: never persistent)
        }
    }

    // ---- SPLIT: merge the partial states of the KV groups through LDS (tile buffers are free now) -----
    if (SPLIT > 1) {
        __syncthreads();  // the other group may still be reading its tiles where the exchange buffer goes
        float* xch = (float*)smem;  // [4 waves][NDT*16 + 1][64 lanes]
        constexpr int NX = NDT * 16 + 1;
        if (grp == 1) {
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) xch[(wave * NX + dt * 16 + r) * 64 + lane] = o[dt][r];
            xch[(wave * NX + NDT * 16) * 64 + lane] = m_run;
        }
        __syncthreads();
        if (grp == 1) return;
        const float m1 = xch[(wave * NX + NDT * 16) * 64 + lane];
        const float m = fmaxf(m_run, m1);
        const float a0 = __builtin_amdgcn_exp2f(m_run - m), a1 = __builtin_amdgcn_exp2f(m1 - m);
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] = o[dt][r] * a0 + xch[(wave * NX + dt * 16 + r) * 64 + lane] * a1;
    }

    // ---- normalise and store: lane owns query q_row, d = 32*dt + (r&3) + 8*(r>>2) + 4*hi -------------
    float l_tot = o[DT_L][R_L];
    {
        const float other = __shfl_xor(l_tot, 32);
        if (hi != HI_L) l_tot = other;
    }
    const float inv = 1.0f / l_tot;
    T* op = out + (long long)q_row * ((long long)H * DH) + (long long)h * DH;
#ifndef ATTN_NO_WIDE_STORE
    if constexpr (ES == 2) {
        // The row is split across the half-waves in 4-column pieces (lane: columns 8k + 4hi .. + 3 of every 8-column group
        // k).  One v_permlane32_swap per dword of a group PAIR (k, k+1) hands the upper half's group-k piece down and the
        // lower half's group-(k+1) piece up: lanes 0-31 then hold columns 8k .. 8k+7, lanes 32-63 columns 8k+8 .. 8k+15 ->
        // one 16-byte store per pair instead of two 8-byte ones (the store tail is bound by the number of store
        // instructions, not by bytes).
        constexpr int NG = DH / 8;
        u32x2 pk[NG];
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            vec4h<T> v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (vec4e<T>)(o[k >> 2][4 * (k & 3) + e] * inv);
            pk[k] = __builtin_bit_cast(u32x2, v);
        }
#pragma unroll
        for (int k = 0; k + 1 < NG; k += 2) {
            const auto x = __builtin_amdgcn_permlane32_swap(pk[k][0], pk[k + 1][0], false, false);
            const auto y = __builtin_amdgcn_permlane32_swap(pk[k][1], pk[k + 1][1], false, false);
            *(u32x4*)(op + 8 * k + 8 * hi) = (u32x4){x[0], y[0], x[1], y[1]};
        }
        if constexpr (NG & 1) *(u32x2*)(op + 8 * (NG - 1) + 4 * hi) = pk[NG - 1];
        continue;  // next tile
    }
#endif
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = dt * 32 + 8 * g + 4 * hi;
            if (d0 < DH) {
                if (ES == 2) {
                    vec4h<T> v;
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = (vec4e<T>)(o[dt][4 * g + k] * inv);
                    *(vec4h<T>*)(op + d0) = v;
                } else {
                    *(f32x4*)(op + d0) = (f32x4){o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv,
                                                 o[dt][4 * g + 3] * inv};
                }
            }
        }
    }  // tile loop
}

template <typename T, int KVB, int DH, int SPLIT, bool HS = false, int QS = 1>
static int launch_attn_t(const void* q, const void* kt, const void* vt, void* out, int B, int S, int H, float scale,
                         hipStream_t stream) {
    constexpr int DP = 96;
    const int ntiles = (S / (128 * QS)) * H * B;
    // query-split form: persistent workgroups (one 8-wave workgroup per CU) that walk the tiles with the next tile's operands
    // prefetched across the seam; the Q staging buffer sits behind the tile rings
    static const int persist_env = getenv("L4P_ATTN_PERSIST") ? atoi(getenv("L4P_ATTN_PERSIST")) : 1;
    const bool persist = SPLIT == 1 && QS > 1 && persist_env && S / KVB >= 4 && (S / KVB) % 2 == 0;  // (the kernel's PERSIST)
    const int slots = 256;  // one 8-wave workgroup per CU (250 registers: two waves per SIMD)
    const int grid = persist && ntiles > slots ? slots : ntiles;
    const size_t lds = SPLIT * 2 * (size_t)(DP / 16 * KVB * 2 * 8 * sizeof(T) + DP * 128) + (QS > 1 ? (size_t)128 * QS * DP * sizeof(T) : 0);
    auto kern = attn_kernel<T, DP, KVB, DH, SPLIT, HS, QS>;
    static lds_attr_state attr_done;
    HIP_TRY(lds_attr_once(attr_done, kern, (int)lds));
    // scale == 0 (L4P_ATTN_PRESCALED): q already carries head_dim^-0.5 * log2(e); the kernels then multiply by exactly 1
    const float c_scale = scale > 0.f ? scale * 1.4426950408889634f : 1.0f;
    ProfScope prof(PROF_ATTENTION, stream);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256 * SPLIT * QS), lds, stream, (const T*)q, (const T*)kt, (const T*)vt, (T*)out,
                       S, H, c_scale, ntiles);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <typename T16>
static int launch_attention16(int variant, bool split, const void* q, const void* kt, const void* vt, void* out, int B, int S, int H,
                              int Dh, float scale, hipStream_t stream) {
    if (variant != 1) {
        // query split (8 waves share the K / V^T tiles): when the launch still fills the chip with 256-row workgroups, two per CU
        const bool qsplit = !split && S % 256 == 0 && (long long)(S / 256) * H * B >= 512 && variant != 2;
        if (Dh == 88 && qsplit) return launch_attn_t<T16, 64, 88, 1, true, 2>(q, kt, vt, out, B, S, H, scale, stream);
        if (Dh == 88)
            return split ? launch_attn_t<T16, 64, 88, 2, true>(q, kt, vt, out, B, S, H, scale, stream)
                         : launch_attn_t<T16, 64, 88, 1, true>(q, kt, vt, out, B, S, H, scale, stream);
        return split ? launch_attn_t<T16, 64, 64, 2, true>(q, kt, vt, out, B, S, H, scale, stream)
                     : launch_attn_t<T16, 64, 64, 1, true>(q, kt, vt, out, B, S, H, scale, stream);
    }
    if (Dh == 88)
        return split ? launch_attn_t<T16, 64, 88, 2>(q, kt, vt, out, B, S, H, scale, stream)
                     : launch_attn_t<T16, 64, 88, 1>(q, kt, vt, out, B, S, H, scale, stream);
    return split ? launch_attn_t<T16, 64, 64, 2>(q, kt, vt, out, B, S, H, scale, stream)
                 : launch_attn_t<T16, 64, 64, 1>(q, kt, vt, out, B, S, H, scale, stream);
}

// scale = Dh^-0.5 (reference :150)
int launch_attention(int dtype, const void* q, const void* kt, const void* vt, void* out, int B, int S, int H, int Dh,
                     float scale, hipStream_t stream) {
    if (S % 128 || (Dh != 88 && Dh != 64) || !(scale >= 0.f)) {
        l4p_set_error("attention: need S %% 128 == 0, head_dim in {88, 64} and scale >= 0 (S=%d Dh=%d scale=%g)", S, Dh, (double)scale);
        return L4P_E_INVALID;
    }
    // too few workgroups for two per CU (256 CUs): split the KV range over two wave groups inside each workgroup
    const bool split = (long long)(S / 128) * H * B < 512 && (S / 64) % 2 == 0;
    static const int variant = getenv("L4P_ATTN_VARIANT") ? atoi(getenv("L4P_ATTN_VARIANT")) : 0;  // tuning aid: 1 = compiler-scheduled body
    // chip-filling 16-bit launches: one wave per SIMD, 64 query rows per wave (attention64.hip)
    if (is16(dtype) && variant == 0 && knob(KNOB_ATTN64) && S % 256 == 0 && (S / 64) % 2 == 0 && S / 64 >= 4 && (long long)(S / 256) * H * B >= 256)
        return launch_attention64(dtype, q, kt, vt, out, B, S, H, Dh, scale, stream);
    if (is16(dtype)) L4P_WITH_T16(dtype, T16, return launch_attention16<T16>(variant, split, q, kt, vt, out, B, S, H, Dh, scale, stream));
    if (Dh == 88) return launch_attn_t<float, 32, 88, 1>(q, kt, vt, out, B, S, H, scale, stream);
    return launch_attn_t<float, 32, 64, 1>(q, kt, vt, out, B, S, H, scale, stream);
}
