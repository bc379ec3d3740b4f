// Two-workgroups-per-CU bf16 MFMA GEMM for gfx950: the form of the large dense GEMMs (same math, descriptor and epilogues as
// gemm.hpp / gemm8p.hpp; reference call sites listed in gemm.hpp) whose END-OF-TILE work overlaps another tile's main loop.
//
// Why: the 8-phase kernel (gemm8p.hpp) owns its CU - one 8-wave workgroup, 128 KB of LDS - so while a tile's epilogue runs (bias /
// GELU / mask product in VALU, the chip-wide store burst of a round) the CU's matrix pipe idles: 9 of the GEMM class's 36 ms in
// the batch-4 all-heads step by a no-epilogue build.  A persistent 8-wave form cannot hide it (stores and the next tile's loads
// share vmcnt and retire in order; measured slower).  Here a workgroup is HALF a CU's worth of waves:
//
//  * 256 threads = 4 waves (one per SIMD), workgroup tile 256 x 128, wave tile 128 x 64 (the 8-phase kernel's: 8 x 4 MFMA tiles,
//    the same fragment reads per MFMA), 72 KB of LDS, <= 256 VGPRs: TWO workgroups are resident per CU and share each SIMD's
//    matrix pipe.  They are independent - own barrier, own tile - and the SIMD arbitrates by age: the older workgroup's waves
//    win the pipe, finish first, and their epilogue (and the next workgroup's prologue in that slot) runs under the younger
//    workgroup's main loop.  After the first tile the two slots of a CU stay about half a tile apart by themselves.
//  * k-step = 32 (one 16x16x32 MFMA deep), three LDS stages of 24 KB (A 256 rows + W 128 rows of 64 bytes), two in flight.
//    One raw s_barrier per k-step (32 MFMAs per wave): wait (counted vmcnt) for the own pieces of stage t -> barrier (everyone's
//    stage t landed, everyone has finished reading stage t - 1) -> stage t + 2 into the slot stage t - 1 used, its six LDS-DMA
//    pieces spread behind the MFMAs -> 12 ds_read_b128 (inline asm, hand-counted lgkmcnt) + 32 MFMAs.
//  * staging by buffer_load ... lds (LDS-DMA through a buffer descriptor): the per-lane byte offset of a piece is loop invariant
//    (one VGPR), the k offset is the instruction's SCALAR offset - no vector address arithmetic in the loop at all (the
//    global_load_lds form of the other kernels re-derives a 64-bit address per piece and k-tile).  K % 32 == 0 is required
//    (the launcher sends other shapes to the 8-phase kernel).
//  * 64-byte LDS rows: chunk c of row r sits at c ^ (((r >> 2) & 1) << 1) in the A tile and at c ^ (((r >> 5) & 1) << 1) in
//    the W tile (whose fragment rows are the permuted 16 q + 4 j + r of the epilogue's column order): both conflict-free for the
//    four 16-lane groups of ds_read_b128; the swizzle is applied to the per-lane SOURCE chunk of the LDS-DMA.
//
// L2 -> LDS bytes per output are 1.5x the 256 x 256 tile's (A 256 + W 128 rows per 32 768 outputs against 512 per 65 536).
#pragma once
#include "gemm.hpp"

struct Gemm4wCfg {
    static constexpr int BM = 256, BN = 128, BK = 32, STAGES = 3;
    static constexpr int A_STAGE = BM * 64, W_STAGE = BN * 64, STAGE = A_STAGE + W_STAGE;
    static constexpr int LDS_BYTES = STAGES * STAGE;
};

template <bool SPLITK = false>
__global__ __launch_bounds__(256, 2) void gemm4w_kernel(const GemmParams p) {
    typedef bf16_t T;
    typedef Gemm4wCfg Cfg;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK, TM = 8, TN = 4;
    constexpr int A_ST = Cfg::A_STAGE, ST = Cfg::STAGE;
    constexpr int NLD = 6;  // LDS-DMA pieces per wave and stage (4 A passes + 2 W passes of 64 rows)
    typedef __attribute__((address_space(3))) void* lptr_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];  // [3][A 256 x 64 B | W 128 x 64 B]

    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, li = lane & 15, kg = lane >> 4;

    // ---- workgroup -> tile: XCD x owns a contiguous range of the banded linear order (gemm8p.hpp); the 64 tiles an XCD runs at a
    //      time are 8 tile rows x a band of <= 8 tile columns ----
    const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM, ntiles = ntn * ntm;
    const int nsplit = SPLITK ? p.splitk : 1;
    const int ksplit = SPLITK ? (int)blockIdx.x / ntiles : 0;
    const int bid = SPLITK ? (int)blockIdx.x - ksplit * ntiles : (int)blockIdx.x;
    int m0, n0;
    {
        const int xcd = bid & 7, idx = bid >> 3, q = ntiles >> 3, r = ntiles & 7;
        const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        const int nb = (ntn + 7) >> 3, bw = (ntn + nb - 1) / nb, full = (nb - 1) * ntm * bw;
        int mt, nt_;
        if (nb == 1) {
            mt = tile / ntn;
            nt_ = tile % ntn;
        } else if (tile < full) {
            const int band = tile / (ntm * bw), rem = tile - band * (ntm * bw);
            mt = rem / bw;
            nt_ = band * bw + rem % bw;
        } else {
            const int w = ntn - (nb - 1) * bw, rem = tile - full;
            mt = rem / w;
            nt_ = (nb - 1) * bw + rem % w;
        }
        m0 = mt * BM, n0 = nt_ * BN;
    }

    // ---- staging: lane (srow, slot) of a 64-row pass fetches source chunk cs of its row ----
    const int srow = tid >> 2, slot = tid & 3;
    const int cs_a = slot ^ (((srow >> 2) & 1) << 1), cs_w = slot ^ (((srow >> 5) & 1) << 1);
    constexpr unsigned OOB = 0x80000000u;  // num_records of both descriptors (every offset is below it: checked by the launcher)
    unsigned a_vo[4], w_vo[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + i * 64 + srow;
        if (m >= p.M) m = p.M - 1;  // rows past M are computed on a clamped row and never stored
        const long long pm = p.a_gr > 0 ? (long long)(m / p.a_gr) * p.a_gs + p.a_go + (m % p.a_gr) : m;
        a_vo[i] = (unsigned)(pm * p.lda * 2 + cs_a * 16);  // (< 2^31: checked by the launcher)
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int n = n0 + i * 64 + srow;
        if (n >= p.N) n = p.N - 1;
        w_vo[i] = (unsigned)((long long)n * p.ldw * 2 + cs_w * 16);
    }
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)OOB, 0x00020000);

    const int nk_all = (p.K + BK - 1) / BK;
    const int kt0 = SPLITK ? (int)((long long)nk_all * ksplit / nsplit) : 0;
    const int nk = SPLITK ? (int)((long long)nk_all * (ksplit + 1) / nsplit) - kt0 : nk_all;

    // one piece (q = 0..3: A pass q, 4..5: W pass q - 4) of k-step kt into stage slot sl
    auto stage_piece = [&](auto q_, int kt, auto sl_) {
        constexpr int q = decltype(q_)::value, sl = decltype(sl_)::value;
        const int kabs = kt0 + kt;
        const unsigned vo = q < 4 ? a_vo[q < 4 ? q : 0] : w_vo[q >= 4 ? q - 4 : 0];
        char* dst = smem + sl * ST + (q < 4 ? q * 64 * 64 : A_ST + (q - 4) * 64 * 64) + wave * (16 * 64);
        if (q < 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lptr_t)dst, 16, vo, kabs * (BK * 2), 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lptr_t)dst, 16, vo, kabs * (BK * 2), 0, 0);
    };
    auto stage_all = [&](int kt, auto sl_) { static_for_<0, NLD>([&](auto q_) { stage_piece(q_, kt, sl_); }); };

    // ---- fragment read addresses (LDS byte addresses; stage, A tile i and W tile j are instruction offsets) ----
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)smem;
    const unsigned a_ad = lds0 + (wr * 128 + li) * 64 + ((kg ^ (((li >> 2) & 1) << 1)) << 4);
    const unsigned w_ad = lds0 + A_ST + (wc * 64 + 16 * (li >> 2) + (li & 3)) * 64 + ((kg ^ (((li >> 3) & 1) << 1)) << 4);

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- prologue: stages 0 and 1 ----
    stage_all(0, std::integral_constant<int, 0>{});
    if (nk > 1) stage_all(1, std::integral_constant<int, 1>{});

    // One k-step on stage slot SL.  Reads (in order): A_0, W_0..W_3, A_1..A_7; MFMAs i-major; MFMA m may issue once the only
    // outstanding reads are those requested after its operands (LDS reads return in order).  All 12 reads are requested up front
    // - the other workgroup's wave on this SIMD owns the matrix pipe meanwhile -, the six pieces of stage t + 2 follow behind
    // MFMAs 2, 7, 12, ... (an LDS-DMA issue is ~60 cycles of this wave's issue time: inside the MFMA shadow it is free).
#ifndef GEMM4W_PRE
#define GEMM4W_PRE 12
#endif
#ifndef GEMM4W_PRIO
#define GEMM4W_PRIO 1
#endif
    auto kstep = [&](int t, auto sl_) {
        constexpr int SL = decltype(sl_)::value, NXT = (SL + 2) % 3;
        constexpr int NR = TM + TN, NM = TM * TN, PRE = GEMM4W_PRE < NR ? GEMM4W_PRE : NR;
        if (t + 1 < nk)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");  // own pieces of stage t landed (stage t + 1 may fly)
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const bool more = t + 2 < nk;
        u32x4 fr[NR];
        auto rd = [&fr, a_ad, w_ad](auto r_) {
            constexpr int r = decltype(r_)::value;
            if constexpr (r == 0)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[r]) : "v"(a_ad), "n"(SL * ST) : "memory");
            else if constexpr (r <= TN)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[r]) : "v"(w_ad), "n"(SL * ST + (r - 1) * 256) : "memory");
            else
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[r]) : "v"(a_ad), "n"(SL * ST + (r - TN) * 1024) : "memory");
        };
        static_for_<0, PRE>(rd);
        if (GEMM4W_PRIO) __builtin_amdgcn_s_setprio(1);
        static_for_<0, NM>([&](auto m_) {
            constexpr int m = decltype(m_)::value, i = m / TN, j = m % TN;
            constexpr int ra = i == 0 ? 0 : TN + i, rw = 1 + j;
            constexpr int need = ra > rw ? ra : rw;
            constexpr int issued = (PRE + m < NR) ? PRE + m : NR;
            static_assert(need < issued, "operand requested before use");
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(issued - need - 1 > 15 ? 15 : issued - need - 1) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            acc[i][j] = mma16(__builtin_bit_cast(bf16x8, fr[rw]), __builtin_bit_cast(bf16x8, fr[ra]), acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PRE + m < NR) rd(std::integral_constant<int, PRE + m>{});
            if constexpr (m % 5 == 2 && m / 5 < NLD) {
                if (more) stage_piece(std::integral_constant<int, m / 5>{}, t + 2, std::integral_constant<int, NXT>{});
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        if (GEMM4W_PRIO) __builtin_amdgcn_s_setprio(0);
    };
    for (int t = 0; t < nk; t += 3) {
        kstep(t, std::integral_constant<int, 0>{});
        if (t + 1 < nk) kstep(t + 1, std::integral_constant<int, 1>{});
        if (t + 2 < nk) kstep(t + 2, std::integral_constant<int, 2>{});
    }

    const int mw = m0 + wr * (TM * 16), nw = n0 + wc * (TN * 16);
    if constexpr (SPLITK) {  // raw float partial [ksplit][M][N]; finished by splitk_finish_kernel (or the caller, tuning bit 1)
        const int nb2 = nw + 4 * TN * kg;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = mw + i * 16 + li;
            if (m >= p.M) continue;
            float* pp = p.partial + ((long long)ksplit * p.M + m) * p.N + nb2;
#pragma unroll
            for (int j = 0; j < TN; ++j)
                if (nb2 + 4 * j < p.N) *(f32x4*)(pp + 4 * j) = acc[i][j];
        }
        return;
    } else {
#ifdef GEMM_DBG_NOEPI  // (tools/probes: main-loop-only timing)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(acc[i][j]));
#else
        if (p.epi == EPI_MASKDOT)
            gemm_epilogue_maskdot<T, TM, TN>(p, acc, mw, nw, li, kg);
        else if (!gemm_epilogue_dense_dispatch<T, TM, TN>(p, acc, mw, nw, li, kg))
            gemm_epilogue<T, TM, TN, true>(p, acc, mw, nw, li, kg);
#endif
    }
}
