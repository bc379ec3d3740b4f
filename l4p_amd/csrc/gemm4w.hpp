// Two-workgroups-per-CU bf16 MFMA GEMM for gfx950: the form of the large dense GEMMs (same math, descriptor and epilogues as
// gemm.hpp / gemm8p.hpp; reference call sites listed in gemm.hpp) whose END-OF-TILE work overlaps another tile's main loop.
//
// Why: the 8-phase kernel (gemm8p.hpp) owns its CU - one 8-wave workgroup, 128 KB of LDS - so while a tile's epilogue runs (bias /
// GELU / mask product in VALU, the chip-wide store burst of a round) the CU's matrix pipe idles: 9 of the GEMM class's 36 ms in
// the batch-4 all-heads step by a no-epilogue build.  A persistent 8-wave form cannot hide it (stores and the next tile's loads
// share vmcnt and retire in order; measured slower).  Here a workgroup is HALF a CU's worth of waves:
//
//  * 256 threads = 4 waves (one per SIMD), workgroup tile 256 x 128, wave tile 128 x 64 (the 8-phase kernel's: 8 x 4 MFMA tiles,
//    the same fragment reads per MFMA), 80 KB of LDS, <= 256 VGPRs: TWO workgroups are resident per CU and share each SIMD's
//    matrix pipe.  They are independent - own barriers, own tile - and the SIMD arbitrates by age: the older workgroup's waves
//    win the pipe, finish first, and their epilogue (and the next workgroup's prologue in that slot) runs under the younger
//    workgroup's main loop.
//  * k-tile = 64 with 128-byte LDS rows, i.e. every LDS-DMA piece (1 KB per wave instruction) is EIGHT FULL 128-byte lines.  A
//    first form of this kernel used 32-deep stages of 64-byte rows (three stages fit the 80 KB): a piece then touches sixteen
//    half lines, and the CU's address / tag path - not L2, not the piece count - bounded the main loop: 1085 -> 1423 TF/s by
//    fetching whole lines in an otherwise identical loop (tools/probes/gemm4w_probe.hip, M131072 N2816 K1408).
//  * 80 KB = A double-buffered (2 x 256 rows x 128 B) + W SINGLE-buffered (128 rows x 128 B).  A k-tile:
//        wait vmcnt(0) (own pieces of k-tile t) | barrier 1 (everyone's landed; everyone is done with A(t-1))
//        8 W fragment reads (all of W(t) a wave needs: 32 VGPRs, held for the k-tile) + the first A fragments
//        MFMAs 0-3 | barrier 2 (every wave holds W(t) in registers: the W buffer is free)
//        MFMAs 4-63 with, behind them: the remaining A fragment reads (a 6-slot register ring, one read per 4 MFMAs, inline asm
//        with hand-counted lgkmcnt) and the 12 LDS-DMA pieces of k-tile t+1 (8 A into the other A buffer, 4 W), one per
//        PSTEP MFMAs so that the last is issued well before the k-tile ends.
//  * staging by buffer_load ... lds (LDS-DMA through a buffer descriptor): the per-lane byte offset of a piece is loop invariant
//    (one VGPR), the k offset is the instruction's SCALAR offset - no vector address arithmetic in the loop (the global_load_lds
//    form of the other kernels re-derives a 64-bit address per piece and k-tile).  Chunks past K (K % 8 == 0; last k-tile only)
//    read as zero through the descriptor's range check: the lane's offset is moved past num_records (soffset is not checked).
//  * swizzles as gemm.hpp: A chunk ^= (row >> 1) & 7; W (natural n order in LDS, fragment rows 16 q + 4 j + r) chunk ^=
//    (q << 1) | ((row >> 1) & 1): conflict-free ds_read_b128; applied to the per-lane SOURCE chunk of the LDS-DMA.
//
// L2 -> LDS bytes per output are 1.5x the 256 x 256 tile's (A 256 + W 128 rows per 32 768 outputs against 512 per 65 536).
#pragma once
#include "gemm.hpp"

struct Gemm4wCfg {
    static constexpr int BM = 256, BN = 128, BK = 64;
    static constexpr int A_BUF = BM * 128, W_BUF = BN * 128;
    static constexpr int LDS_BYTES = 2 * A_BUF + W_BUF;
};

// DBG (tools/probes/gemm4w_probe.hip only; 0 in the library): 1 no epilogue, 2 no LDS-DMA in the loop, 4 no fragment reads,
// 8 no s_setprio
template <typename T, bool SPLITK = false, int DBG = 0, int PSTEP = 3>
__global__ __launch_bounds__(256, 2) void gemm4w_kernel(const GemmParams p) {
    typedef Gemm4wCfg Cfg;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK, TM = 8, TN = 4;
    constexpr int A_BUF = Cfg::A_BUF;
    constexpr int NPA = 8, NPW = 4, NLD = NPA + NPW;  // LDS-DMA pieces per wave and k-tile (passes of 32 rows)
    typedef __attribute__((address_space(3))) void* lptr_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];  // [A buffer 0 | A buffer 1 | W]

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

    // ---- staging: lane (crow, cc) of a 32-row pass fetches source chunk cs of its row ----
    const int crow = tid >> 3, cc = tid & 7;
    const int cs_a = cc ^ ((crow >> 1) & 7);
    constexpr unsigned OOB = 0x80000000u;  // num_records of both descriptors (every valid offset is below it: the launcher checks)
    unsigned a_vo[NPA], w_vo[NPW];
    auto cs_w = [&](int i) {  // W rows are read through the permuted fragment row map: their own XOR phase (gemm.hpp, sww)
        const int row = i * 32 + crow;
        return cc ^ (((((row & 63) >> 4) & 3) << 1) | ((row >> 1) & 1));
    };
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
        int m = m0 + i * 32 + crow;
        if (m >= p.M) m = p.M - 1;  // rows past M are computed on a clamped row and never stored
        const long long pm = p.a_gr > 0 ? (long long)(m / p.a_gr) * p.a_gs + p.a_go + (m % p.a_gr) : m;
        a_vo[i] = (unsigned)(pm * p.lda * 2 + cs_a * 16);
    }
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        int n = n0 + i * 32 + crow;
        if (n >= p.N) n = p.N - 1;
        w_vo[i] = (unsigned)((long long)n * p.ldw * 2 + cs_w(i) * 16);
    }
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)OOB, 0x00020000);

    const int nk_all = (p.K + BK - 1) / BK;
    const int kt0 = SPLITK ? (int)((long long)nk_all * ksplit / nsplit) : 0;
    const int nk = SPLITK ? (int)((long long)nk_all * (ksplit + 1) / nsplit) - kt0 : nk_all;

    // piece q (0..7: A pass q, 8..11: W pass q - 8) of k-tile kt; A pieces go to A buffer ab.  or_a / or_w[q & 1]: zero, or - in
    // the matrix's last, partial k-tile - bit 31 for the lanes whose chunk lies at or past K: the offset then falls outside the
    // descriptor's range and the piece reads as zero there (tail_or below; the XOR phase of a W row depends on the pass parity only)
    unsigned or_a = 0u, or_w[2] = {0u, 0u};
    auto tail_or = [&](int kt) {
        const int krem = p.K - (kt0 + kt) * BK;  // valid k in this k-tile (>= 64 except in the tail tile)
        or_a = cs_a * 8 >= krem ? OOB : 0u;
        or_w[0] = cs_w(0) * 8 >= krem ? OOB : 0u;
        or_w[1] = cs_w(1) * 8 >= krem ? OOB : 0u;
    };
    auto stage_piece = [&](auto q_, int kt, auto ab_) {
        constexpr int q = decltype(q_)::value, ab = decltype(ab_)::value;
        const int kabs = kt0 + kt;
        const unsigned vo = q < NPA ? (a_vo[q < NPA ? q : 0] | or_a) : (w_vo[q >= NPA ? q - NPA : 0] | or_w[q & 1]);
        char* dst = smem + (q < NPA ? ab * A_BUF + q * 4096 : 2 * A_BUF + (q - NPA) * 4096) + wave * 1024;
        if (q < NPA)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lptr_t)dst, 16, vo, kabs * (BK * 2), 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lptr_t)dst, 16, vo, kabs * (BK * 2), 0, 0);
    };

    // ---- fragment read addresses (LDS byte addresses; buffer, A tile i and W tile j are instruction offsets) ----
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)smem;
    const int swa = (li >> 1) & 7, swq = (((li >> 2) & 3) << 1) | ((li >> 1) & 1);
    unsigned a_ad[2], w_ad[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        a_ad[kk] = lds0 + (wr * 128 + li) * 128 + (((kk * 4 + kg) ^ swa) << 4);                                  // + i * 2048
        w_ad[kk] = lds0 + 2 * A_BUF + (wc * 64 + 16 * (li >> 2) + (li & 3)) * 128 + (((kk * 4 + kg) ^ swq) << 4);  // + j * 512
    }

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- prologue: k-tile 0 ----
    const bool ktail = (p.K & (BK - 1)) != 0;  // the matrix's last k-tile is partial
    if (ktail && kt0 + 1 == nk_all) tail_or(0);
    static_for_<0, NLD>([&](auto q_) { stage_piece(q_, 0, std::integral_constant<int, 0>{}); });

    // One k-tile on A buffer AB.  LDS reads in issue order: W(kk, j) (8), then A fragment g = kk * 8 + i (16); before MFMA 0 the W
    // reads and PREA A fragments are requested, fragment g + PREA behind the first MFMA of fragment g.  MFMA m: kk = m / 32,
    // i = (m % 32) / 4, j = m % 4.  LDS reads return in order: MFMA 4 g may issue once at most (issued - 9 - g) reads are outstanding.
    constexpr int PREA = 4, RA = PREA + 2;  // A fragments requested ahead; register ring (a slot is rewritten two fragments later)
    auto ktile = [&](int t, auto ab_) {
        constexpr int AB = decltype(ab_)::value;
        constexpr bool PRIO = !(DBG & 8);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // own pieces of k-tile t have landed
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const bool more = t + 1 < nk && !(DBG & 2);
        if (ktail && kt0 + t + 2 == nk_all) tail_or(t + 1);  // (wave-uniform; once per tile)
        u32x4 fw[2][TN], fa[RA];
        auto rd_w = [&fw, &w_ad](auto r_) {
            constexpr int r = decltype(r_)::value, kk = r / TN, j = r % TN;
            if constexpr ((DBG & 4) != 0)
                asm volatile("v_mov_b32 %0, 0x3f803f80\n\tv_mov_b32 %1, 0x3f803f80\n\tv_mov_b32 %2, 0x3f803f80\n\tv_mov_b32 %3, 0x3f803f80"
                             : "=v"(fw[kk][j][0]), "=v"(fw[kk][j][1]), "=v"(fw[kk][j][2]), "=v"(fw[kk][j][3]));
            else
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fw[kk][j]) : "v"(w_ad[kk]), "n"(j * 512) : "memory");
        };
        auto rd_a = [&fa, &a_ad](auto g_) {
            constexpr int g = decltype(g_)::value, kk = g / TM, i = g % TM;
            if constexpr ((DBG & 4) != 0)
                asm volatile("v_mov_b32 %0, 0x3f803f80\n\tv_mov_b32 %1, 0x3f803f80\n\tv_mov_b32 %2, 0x3f803f80\n\tv_mov_b32 %3, 0x3f803f80"
                             : "=v"(fa[g % RA][0]), "=v"(fa[g % RA][1]), "=v"(fa[g % RA][2]), "=v"(fa[g % RA][3]));
            else
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[g % RA]) : "v"(a_ad[kk]), "n"(AB * A_BUF + i * 2048) : "memory");
        };
        static_for_<0, 2 * TN>(rd_w);
        static_for_<0, PREA>(rd_a);
        if (PRIO) __builtin_amdgcn_s_setprio(1);
        static_for_<0, 2 * TM * TN>([&](auto m_) {
            constexpr int m = decltype(m_)::value, kk = m / (TM * TN), i = (m % (TM * TN)) / TN, j = m % TN, g = m / TN;
            if constexpr (j == 0) {  // first MFMA of fragment g: reads issued so far = 8 + min(16, PREA + g), fragment g is read 8 + g
                constexpr int issued = 2 * TN + (PREA + g < 2 * TM ? PREA + g : 2 * TM);
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(issued - (2 * TN + g) - 1) : "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[i][j] = mma16(__builtin_bit_cast(vec8<T>, fw[kk][j]), __builtin_bit_cast(vec8<T>, fa[g % RA]), acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (j == 0 && g + PREA < 2 * TM) rd_a(std::integral_constant<int, (g + PREA < 2 * TM ? g + PREA : 0)>{});
            if constexpr (m == TN - 1) {  // every wave holds W(t) in registers (the wait of MFMA 0 covered the eight W reads)
                if (PRIO) __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_s_barrier();
                if (PRIO) __builtin_amdgcn_s_setprio(1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (m >= TN && (m - TN) % PSTEP == PSTEP - 1 && (m - TN) / PSTEP < NLD) {
                if (more) stage_piece(std::integral_constant<int, (m - TN) / PSTEP>{}, t + 1, std::integral_constant<int, AB ^ 1>{});
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        if (PRIO) __builtin_amdgcn_s_setprio(0);
    };
    for (int t = 0; t < nk; t += 2) {
        ktile(t, std::integral_constant<int, 0>{});
        if (t + 1 < nk) ktile(t + 1, std::integral_constant<int, 1>{});
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
        if constexpr ((DBG & 1) != 0) {  // main-loop-only timing
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(acc[i][j]));
        } else if (p.epi == EPI_MASKDOT)
            gemm_epilogue_maskdot<T, TM, TN>(p, acc, mw, nw, li, kg);
        else if (!gemm_epilogue_dense_dispatch<T, TM, TN>(p, acc, mw, nw, li, kg))
            gemm_epilogue<T, TM, TN, true>(p, acc, mw, nw, li, kg);
    }
}
