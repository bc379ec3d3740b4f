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
#include <cstdlib>
#include <type_traits>

#include "common.hpp"

__device__ __attribute__((aligned(16))) static const unsigned short g_ones_bf16[8] = {0x3F80, 0x3F80, 0x3F80, 0x3F80,
                                                                                         0x3F80, 0x3F80, 0x3F80, 0x3F80};
__device__ __attribute__((aligned(16))) static const float g_ones_f32[4] = {1.f, 1.f, 1.f, 1.f};

// SPLIT = 2: the workgroup has 8 waves; waves 0-3 walk the even KV blocks and waves 4-7 the odd ones for the
// SAME 128 query rows, and the two partial (m, O, denominator) states are merged through LDS at the end.
// Used when the launch has too few workgroups to put two of them on a CU (batch 1: 256 workgroups), so
// every SIMD still holds two waves whose MFMA / VALU / wait phases overlap.
template <typename T, int DP, int KVB, int DH, int SPLIT>
__global__ __launch_bounds__(256 * SPLIT) void attn_kernel(const T* __restrict__ q, const T* __restrict__ kt,
                                                           const T* __restrict__ vt, T* __restrict__ out, int S, int H,
                                                           float c_scale) {
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
    constexpr int DT_L = DH / 32, I_L = DH % 32;                       // where the denominator row lands in O^T
    constexpr int HI_L = (I_L >> 2) & 1, R_L = (I_L & 3) + 4 * (I_L >> 3);

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);  // KV-split group of this wave
    char* Ks = smem + grp * 2 * (KBYTES + VBYTES);  // [2][KBYTES]   (per group)
    char* Vs = Ks + 2 * KBYTES;                      // [2][VBYTES]

    const int tid = threadIdx.x & 255, lane = tid & 63;  // thread / wave index inside the group
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, hi = lane >> 5;
    // XCD-aware work mapping: workgroup L runs on XCD L % 8 (observed dispatch order; speed only, not
    // correctness).  All S/128 query blocks of one (batch, head) are given to the SAME XCD so that head's
    // K / V^T (786 KB) is fetched into one L2 instead of eight.
    const int nqb = S / 128, units = gridDim.x / nqb;  // units = B * H
    int unit, qb;
    if ((units & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        unit = xcd + 8 * (j / nqb);
        qb = j % nqb;
    } else {
        unit = blockIdx.x / nqb;
        qb = blockIdx.x % nqb;
    }
    const int b = unit / H, h = unit % H;
    const int q_row = qb * 128 + wave * 32 + lq;

    // Q fragments (B operand: column = query, 8 consecutive d per k-step half)
    frag_t qf[NKS];
    {
        const T* qp = q + ((long long)b * S + q_row) * ((long long)H * DP) + (long long)h * DP;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            u32x4* d = (u32x4*)&qf[ks];
#pragma unroll
            for (int c = 0; c < CPF; ++c) d[c] = *(const u32x4*)(qp + ks * 16 + hi * 8 + c * EPC);
        }
    }

    // ---- LDS-DMA sources -----------------------------------------------------------------------------
    const int nkb = S / KVB;
    const char* kbase = (const char*)kt + ((long long)(b * H + h) * nkb) * KBYTES + tid * 16;  // + kb*KBYTES + i*4096
    // V^T: LDS position (row d, slot) <- chunk slot ^ ((d >> 1) & 7) of that row
    const int vrow0 = tid >> 3, vslot = tid & 7;
    const int vchunk = vslot ^ ((vrow0 >> 1) & 7);  // (d >> 1) & 7 is the same for d = vrow0 + 32*i
    const char* vsrc[V_IT];
    bool vones[V_IT];
#pragma unroll
    for (int i = 0; i < V_IT; ++i) {
        const int d = vrow0 + 32 * i;
        vones[i] = d == DH;
        vsrc[i] = (const char*)(vt + (((long long)b * H + h) * DP + d) * S) + vchunk * 16;  // + kb*128
    }
    const char* ones = ES == 2 ? (const char*)g_ones_bf16 : (const char*)g_ones_f32;
    auto issue_k = [&](int kb, int buf) {
#pragma unroll
        for (int i = 0; i < K_IT; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(kbase + (long long)kb * KBYTES + i * 4096),
                                             (lptr_t)(Ks + buf * KBYTES + (wave * 64 + i * 256) * 16), 16, 0, 0);
    };
    auto issue_v = [&](int kb, int buf) {
#pragma unroll
        for (int i = 0; i < V_IT; ++i) {
            const char* src = vones[i] ? ones : vsrc[i] + (long long)kb * 128;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Vs + buf * VBYTES + (wave * 64 + i * 256) * 16), 16, 0, 0);
        }
    };

    f32x16 o[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY;

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
    const int nit = nkb / SPLIT;
    issue_k(grp, 0);
    issue_v(grp, 0);
    if (nit > 1) issue_k(SPLIT + grp, 1);
    __syncthreads();
    f32x16 s_cur[NST], s_nxt[NST];
    qk_tile(Ks, s_cur);
    float mx = row_max(s_cur);

    // one pipeline step; HAS_NEXT is a compile-time flag so that the steady-state body is ONE basic block in which the
    // scheduler is free to interleave the two MFMA batches with the VALU work (the last block is peeled)
    auto step = [&](int it, auto has_next) {
        constexpr bool HAS_NEXT = decltype(has_next)::value;
        const int cur = it & 1;
#ifndef ATTN_DBG_NOLOAD  // (tools/probes/attn_variants.hip: compute-only timing)
        if (it + 2 < nit) issue_k((it + 2) * SPLIT + grp, cur);      // K_{it} (ring slot cur) was consumed last iteration
        if (HAS_NEXT) issue_v((it + 1) * SPLIT + grp, cur ^ 1);      // V_{it-1} (slot cur^1) was consumed last iteration
#endif
#ifdef ATTN_DBG_NOCOMPUTE  // (probe: staging-only timing)
        __syncthreads();
        return;
#endif
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
        // ---- next block's scores (MFMA) alongside this block's exponentials (VALU) ----------------
        if (HAS_NEXT) qk_tile(Ks + (cur ^ 1) * KBYTES, s_nxt);
        frag_t pf[NST][2];
#pragma unroll
        for (int t = 0; t < NST; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    pf[t][j][e] = from_f32<T>(__builtin_amdgcn_exp2f(__builtin_fmaf(s_cur[t][8 * j + e], c_scale, -m_run)));
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
#pragma unroll
            for (int t = 0; t < NST; ++t) s_cur[t] = s_nxt[t];
            __syncthreads();  // (drains the LDS-DMA issued at the top: vmcnt(0) + barrier)
        }
    };
    for (int it = 0; it + 1 < nit; ++it) step(it, std::true_type{});
    step(nit - 1, std::false_type{});

    // ---- SPLIT: merge the partial states of the KV groups through LDS (tile buffers are free now) -----
    if (SPLIT > 1) {
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
    T* op = out + ((long long)b * S + q_row) * ((long long)H * DH) + (long long)h * DH;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = dt * 32 + 8 * g + 4 * hi;
            if (d0 < DH) {
                if (ES == 2) {
                    bf16x4 v;
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = (bf16_t)(o[dt][4 * g + k] * inv);
                    *(bf16x4*)(op + d0) = v;
                } else {
                    *(f32x4*)(op + d0) = (f32x4){o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv,
                                                 o[dt][4 * g + 3] * inv};
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// Ping-pong form for the bf16 engine (the CDNA4 two-waves-per-SIMD recipe).  512 threads = two groups of 4 waves, 32
// query rows per wave, 256 query rows per workgroup, and BOTH groups walk the same K / V^T tiles, so every staged tile
// feeds twice the MFMA work of the 128-row kernel above (the L2 -> LDS stream was a third of its time).  Every SIMD hosts
// one wave of each group and the groups run ONE BARRIER APART: while a wave is in its matrix segment
//       M(j):   O^T += V_j^T P_j^T ;  S_{j+1} = K_{j+1} Q^T            (24 MFMAs at raised priority, fragments from LDS)
// its SIMD partner is in its vector segment
//       V(j+1): row max, running max / rescale, P_{j+1} = exp2(S_{j+1} c - m), one tile of LDS-DMA
// so the matrix pipe and the VALU of a SIMD work at the same time instead of taking turns inside one wave.
// Tiles live in 3-deep rings.  Group 0's vector segment (global slot 2j+1) stages K_{j+3}, group 1's (slot 2j+2) stages
// V_{j+3}; each replaces the tile both groups finished reading one slot earlier, is waited for by a COUNTED vmcnt at the end
// of the stager's next vector segment (the LDS-DMA queue is never drained in the loop) and is first read two slots later.
template <int DH>
__global__ __launch_bounds__(512) void attn_pp_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ kt,
                                                      const bf16_t* __restrict__ vt, bf16_t* __restrict__ out, int S, int H,
                                                      float c_scale) {
    typedef bf16_t T;
    typedef bf16x8 frag_t;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    constexpr int DP = 96, KVB = 64, NKS = DP / 16, NST = KVB / 32, NDT = DP / 32, QB = 256;
    constexpr int KBYTES = NKS * KVB * 2 * 8 * 2, VBYTES = DP * 128;  // 12 KB each
    constexpr int NLD = KBYTES / 4096;                                // LDS-DMA instructions per tile per wave (4 waves)
    static_assert(KBYTES == VBYTES && KBYTES % 4096 == 0, "one tile = NLD passes of a 4-wave group");
    constexpr int DT_L = DH / 32, I_L = DH % 32;
    constexpr int HI_L = (I_L >> 2) & 1, R_L = (I_L & 3) + 4 * (I_L >> 3);

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;               // K ring [3][KBYTES]
    char* Vs = smem + 3 * KBYTES;  // V ring [3][VBYTES]
    const unsigned lds_k = (unsigned)(size_t)(__attribute__((address_space(3))) char*)Ks;  // LDS byte addresses (inline-asm reads)
    const unsigned lds_v = lds_k + 3 * KBYTES;
    const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
    const int tid = threadIdx.x & 255, lane = tid & 63;  // position inside the group
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, hi = lane >> 5;
    // all query blocks of one (batch, head) on one XCD (workgroup L runs on XCD L % 8): its K / V^T stay in one L2
    const int nqb = S / QB, units = gridDim.x / nqb;
    int unit, qb;
    if ((units & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        unit = xcd + 8 * (j / nqb);
        qb = j % nqb;
    } else {
        unit = blockIdx.x / nqb;
        qb = blockIdx.x % nqb;
    }
    const int b = unit / H, h = unit % H;
    const int q_row = qb * QB + grp * 128 + wave * 32 + lq;

    frag_t qf[NKS];
    {
        const T* qp = q + ((long long)b * S + q_row) * ((long long)H * DP) + (long long)h * DP;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[ks] = *(const frag_t*)(qp + ks * 16 + hi * 8);
    }
    const int nit = S / KVB;
    // ---- LDS-DMA sources: a group stages one whole tile (NLD x 256 lanes x 16 B) ----
    const char* kbase = (const char*)kt + ((long long)(b * H + h) * nit) * KBYTES + tid * 16;
    const int vrow0 = tid >> 3, vslot = tid & 7;
    const int vchunk = vslot ^ ((vrow0 >> 1) & 7);
    const char* vsrc[NLD];
    bool vones[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int d = vrow0 + 32 * i;
        vones[i] = d == DH;
        vsrc[i] = (const char*)(vt + (((long long)b * H + h) * DP + d) * S) + vchunk * 16;
    }
    const char* ones = (const char*)g_ones_bf16;
    // block index past the end: the last block is staged again into a free slot so the counted waits stay exact
    auto issue_k = [&](int j, int slot) {
        const int kb = j < nit ? j : nit - 1;
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(kbase + (long long)kb * KBYTES + i * 4096),
                                             (lptr_t)(Ks + slot * KBYTES + (wave * 64 + i * 256) * 16), 16, 0, 0);
    };
    auto issue_v = [&](int j, int slot) {
        const int kb = j < nit ? j : nit - 1;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const char* src = vones[i] ? ones : vsrc[i] + (long long)kb * 128;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Vs + slot * VBYTES + (wave * 64 + i * 256) * 16), 16, 0, 0);
        }
    };

    f32x16 o[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY;
    const int krow = (lq & ~12) | ((lq & 4) << 1) | ((lq & 8) >> 1);
    int koff[NST];
#pragma unroll
    for (int t = 0; t < NST; ++t) {
        const int key = t * 32 + krow;
        koff[t] = (key * 2 + (hi ^ ((key >> 3) & 1))) * 16;
    }
    int voff[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) voff[dt] = (dt * 32 + lq) * 128;
    const int vsw = (lq >> 1) & 7;

    f32x16 s[NST];
    frag_t pf[NST][2];
    // vector segment: online softmax of the scores in s -> P fragments
    auto softmax = [&]() {
        float mx = s[0][0];
#pragma unroll
        for (int t = 0; t < NST; ++t)
#pragma unroll
            for (int r = (t == 0 ? 1 : 0); r < 16; ++r) mx = fmaxf(mx, s[t][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx * c_scale);
        if (__any(m_new > m_run)) {
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            m_run = m_new;
        }
#pragma unroll
        for (int t = 0; t < NST; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    pf[t][j][e] = (bf16_t)__builtin_amdgcn_exp2f(__builtin_fmaf(s[t][8 * j + e], c_scale, -m_run));
    };
    auto seg_barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: the group stages K_0..K_2 / V_0..V_2 (group 0 the K tiles, group 1 the V tiles); S_0, P_0 ----
    if (grp == 0) {
        issue_k(0, 0);
        issue_k(1, 1);
        issue_k(2, 2);
    } else {
        issue_v(0, 0);
        issue_v(1, 1);
        issue_v(2, 2);
    }
    __syncthreads();  // (vmcnt(0) + barrier)
#pragma unroll
    for (int t = 0; t < NST; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) s[t] = mma32(*(const frag_t*)(Ks + ks * (KVB * 2 * 16) + koff[t]), qf[ks], s[t]);
    }
    softmax();
    seg_barrier();
#ifndef ATTN_PP_NOSTAGGER
    if (grp == 1) seg_barrier();  // second group runs one barrier behind
#endif

    int sv = 0, sk = 1;  // ring slots of V_j and K_{j+1}
    for (int j = 0; j < nit; ++j) {
        // ---- M(j): S_{j+1} = K_{j+1} Q^T first, then O^T += V_j^T P_j^T.
        //      The fragment reads are inline-asm ds_read_b128 with hand-counted lgkmcnt waits: while an LDS-DMA
        //      (global_load_lds) is in flight hipcc turns every lgkmcnt wait of its own into lgkmcnt(0), which drains the
        //      whole read queue in front of each MFMA.  LDS reads return in order, so MFMA k may start once the reads
        //      issued after its operand are the only ones outstanding.  15 reads (12 K + 3 V) go out up front, the other
        //      9 V reads ride behind the first 9 MFMAs. ----
        {
            const unsigned vb = lds_v + sv * VBYTES, kb = lds_k + sk * KBYTES;
            u32x4 vf[NST * 2 * NDT], kf[NKS * NST];
#define L4P_DS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define L4P_LGKM(n)                                          \
    do {                                                     \
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory"); \
        __builtin_amdgcn_sched_barrier(0);                   \
    } while (0)
            unsigned ka[NST], va[NST * 2];
#pragma unroll
            for (int t = 0; t < NST; ++t) ka[t] = kb + koff[t];
#pragma unroll
            for (int tj = 0; tj < NST * 2; ++tj) va[tj] = vb + lq * 128 + ((((tj * 16 + hi * 8) >> 3) ^ vsw) << 4);
#define L4P_K_READ(i) L4P_DS_READ(kf[i], ka[(i) % NST], ((i) / NST) * (KVB * 2 * 16))
#define L4P_V_READ(i) L4P_DS_READ(vf[i], va[(i) / NDT], ((i) % NDT) * 4096)
            static_assert(NKS * NST == 12 && NST * 2 * NDT == 12, "hand-counted schedule below");
            L4P_K_READ(0); L4P_K_READ(1); L4P_K_READ(2); L4P_K_READ(3); L4P_K_READ(4); L4P_K_READ(5);
            L4P_K_READ(6); L4P_K_READ(7); L4P_K_READ(8); L4P_K_READ(9); L4P_K_READ(10); L4P_K_READ(11);
            L4P_V_READ(0); L4P_V_READ(1); L4P_V_READ(2);
#ifndef ATTN_PP_NOPRIO
            __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
            for (int t = 0; t < NST; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
            // QK MFMA k consumes read k; reads issued so far = 15 + min(k, 9)  ->  allowed outstanding = 14 + min(k, 9) - k
#define L4P_QK(k, cnt)                                                                                        \
    L4P_LGKM(cnt);                                                                                            \
    s[(k) % NST] = mma32(__builtin_bit_cast(frag_t, kf[k]), qf[(k) / NST], s[(k) % NST]);                     \
    __builtin_amdgcn_sched_barrier(0);
            L4P_QK(0, 14) L4P_V_READ(3);  L4P_QK(1, 14) L4P_V_READ(4);  L4P_QK(2, 14) L4P_V_READ(5);
            L4P_QK(3, 14) L4P_V_READ(6);  L4P_QK(4, 14) L4P_V_READ(7);  L4P_QK(5, 14) L4P_V_READ(8);
            L4P_QK(6, 14) L4P_V_READ(9);  L4P_QK(7, 14) L4P_V_READ(10); L4P_QK(8, 14) L4P_V_READ(11);
            L4P_QK(9, 14) L4P_QK(10, 13) L4P_QK(11, 12)
            // PV MFMA i consumes V read i = read 12 + i of 24  ->  allowed outstanding = 11 - i
#define L4P_PV(i)                                                                                             \
    L4P_LGKM(11 - (i));                                                                                       \
    o[(i) % NDT] = mma32(__builtin_bit_cast(frag_t, vf[i]), pf[((i) / NDT) >> 1][((i) / NDT) & 1], o[(i) % NDT]); \
    __builtin_amdgcn_sched_barrier(0);
            L4P_PV(0) L4P_PV(1) L4P_PV(2) L4P_PV(3) L4P_PV(4) L4P_PV(5) L4P_PV(6) L4P_PV(7) L4P_PV(8) L4P_PV(9) L4P_PV(10) L4P_PV(11)
#undef L4P_PV
#undef L4P_QK
#undef L4P_K_READ
#undef L4P_V_READ
#undef L4P_LGKM
#undef L4P_DS_READ
#ifndef ATTN_PP_NOPRIO
            __builtin_amdgcn_s_setprio(0);
#endif
        }
        seg_barrier();
        // ---- V(j+1): stage block j+3 into the slot block j used (K: slot of K_j = sk - 1; V: slot of V_j = sv) ----
#ifndef ATTN_PP_NOLOAD  // (tools/probes/attn_variants.hip ablations)
        if (grp == 0)
            issue_k(j + 3, sk == 0 ? 2 : sk - 1);
        else
            issue_v(j + 3, sv);
#endif
#ifndef ATTN_PP_NOSOFTMAX
        if (j + 1 < nit) softmax();
#else
        asm volatile("" ::"v"(s[0]), "v"(s[1]));
#endif
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");  // the tile staged one iteration ago has landed
        seg_barrier();
        sv = sv == 2 ? 0 : sv + 1;
        sk = sk == 2 ? 0 : sk + 1;
    }
#ifndef ATTN_PP_NOSTAGGER
    if (grp == 0) seg_barrier();  // pairs with the delayed group's last barrier
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup's LDS allocation

    float l_tot = o[DT_L][R_L];
    {
        const float other = __shfl_xor(l_tot, 32);
        if (hi != HI_L) l_tot = other;
    }
    const float inv = 1.0f / l_tot;
    T* op = out + ((long long)b * S + q_row) * ((long long)H * DH) + (long long)h * DH;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = dt * 32 + 8 * g + 4 * hi;
            if (d0 < DH) {
                bf16x4 v;
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = (bf16_t)(o[dt][4 * g + k] * inv);
                *(bf16x4*)(op + d0) = v;
            }
        }
}

template <int DH>
static int launch_attn_pp(const void* q, const void* kt, const void* vt, void* out, int B, int S, int H, float scale,
                          hipStream_t stream) {
    const size_t lds = 3 * (size_t)(12288 + 12288);
    auto kern = attn_pp_kernel<DH>;
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    ProfScope prof(PROF_ATTENTION, stream);
    hipLaunchKernelGGL(kern, dim3((S / 256) * H * B), dim3(512), lds, stream, (const bf16_t*)q, (const bf16_t*)kt,
                       (const bf16_t*)vt, (bf16_t*)out, S, H, scale * 1.4426950408889634f);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <typename T, int KVB, int DH, int SPLIT>
static int launch_attn_t(const void* q, const void* kt, const void* vt, void* out, int B, int S, int H, float scale,
                         hipStream_t stream) {
    constexpr int DP = 96;
    const size_t lds = SPLIT * 2 * (size_t)(DP / 16 * KVB * 2 * 8 * sizeof(T) + DP * 128);
    auto kern = attn_kernel<T, DP, KVB, DH, SPLIT>;
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const float c_scale = scale * 1.4426950408889634f;
    ProfScope prof(PROF_ATTENTION, stream);
    hipLaunchKernelGGL(kern, dim3((S / 128) * H * B), dim3(256 * SPLIT), lds, stream, (const T*)q, (const T*)kt, (const T*)vt, (T*)out,
                       S, H, c_scale);
    HIP_TRY(hipGetLastError());
    return 0;
}

// scale = Dh^-0.5 (reference :150)
int launch_attention(int dtype, const void* q, const void* kt, const void* vt, void* out, int B, int S, int H, int Dh,
                     float scale, hipStream_t stream) {
    if (S % 128 || (Dh != 88 && Dh != 64)) {
        l4p_set_error("attention: need S %% 128 == 0 and head_dim in {88, 64} (S=%d Dh=%d)", S, Dh);
        return L4P_E_INVALID;
    }
    // too few workgroups for two per CU (256 CUs): split the KV range over two wave groups inside each workgroup
    const bool split = (long long)(S / 128) * H * B < 512 && (S / 64) % 2 == 0;
    // L4P_ATTN_VARIANT=2 selects the experimental ping-pong kernel (measured 63 / 144 us at batch 1 / 4 against 38 / 132 us
    // for the kernels below: its matrix and vector segments do not yet overlap on a SIMD; kept for the next tuning round)
    static const int variant = getenv("L4P_ATTN_VARIANT") ? atoi(getenv("L4P_ATTN_VARIANT")) : 0;
    if (dtype == L4P_BF16 && variant == 2 && S % 256 == 0 && S / 64 >= 4)
        return Dh == 88 ? launch_attn_pp<88>(q, kt, vt, out, B, S, H, scale, stream)
                        : launch_attn_pp<64>(q, kt, vt, out, B, S, H, scale, stream);
    if (dtype == L4P_BF16) {
        if (Dh == 88)
            return split ? launch_attn_t<bf16_t, 64, 88, 2>(q, kt, vt, out, B, S, H, scale, stream)
                         : launch_attn_t<bf16_t, 64, 88, 1>(q, kt, vt, out, B, S, H, scale, stream);
        return split ? launch_attn_t<bf16_t, 64, 64, 2>(q, kt, vt, out, B, S, H, scale, stream)
                     : launch_attn_t<bf16_t, 64, 64, 1>(q, kt, vt, out, B, S, H, scale, stream);
    }
    if (Dh == 88) return launch_attn_t<float, 32, 88, 1>(q, kt, vt, out, B, S, H, scale, stream);
    return launch_attn_t<float, 32, 64, 1>(q, kt, vt, out, B, S, H, scale, stream);
}
