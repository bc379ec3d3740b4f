// Deep-pipelined bf16 MFMA GEMM / implicit-GEMM conv3d for LARGE problems on gfx950 (same math, descriptor and
// epilogue as gemm.hpp; reference call sites listed there).  Where gemm.hpp's 128x128 tile is bound by the per-CU
// global->LDS path (64 FLOP/B) and drains its LDS-DMA at every barrier, this kernel follows the CDNA4 playbook's
// "8-phase" structure:
//
//  * 512 threads = 8 waves, wave tile 128 x 64 (acc 8 x 4 MFMA tiles), workgroup tile (WR*128) x (WC*64):
//    256 x 256 (WR=2, WC=4; 128 FLOP/B) or 512 x 128 (WR=4, WC=2; narrow-N convs).  One workgroup per CU.
//  * a k-tile (64 deep) lives in LDS as four HALF-tiles: A0/A1 = the first/second 64 rows of every wave's 128 rows,
//    W0/W1 = the first/second 32 columns (as 4 x 8-column groups, see the epilogue's column order) of every wave's 64.
//    Two k-tile buffers.  128-byte rows, XOR-swizzled at 16 B exactly like gemm.hpp (conflict-free ds_read_b128),
//    filled by LDS-DMA with the swizzle applied to the per-lane SOURCE chunk.
//  * each k-tile is four PHASES, one C quadrant (64 x 32, 16 MFMAs over k = 64) each:
//        P1: read W0 + A0 | stage W1(t+1) | Q00      P2: read W1 | stage A1(t+1) | Q01
//        P3: read A1      | stage A0(t+2) | Q11      P4:    -    | stage W0(t+2) | Q10
//    A phase = [ds_reads, one half-tile of LDS-DMA, counted vmcnt] s_barrier [16 MFMAs at raised priority] s_barrier.
//  * the two wave groups (waves 0-3 / 4-7: one wave of each per SIMD) run ONE BARRIER APART, so on every SIMD one
//    wave's MFMA segment overlaps its partner's read/stage segment.
//  * the LDS-DMA queue is never drained in the loop: after its own issue every phase waits vmcnt(INFLIGHT), i.e. for
//    the half-tile issued four phases earlier.  Hazards (both groups, one barrier apart):
//        RAW: data waited for in phase q is first read in phase q+1        (W1: q=P1',r=P2'; A1: P2'/P3'; A0: P3'/P1''; W0: P4'/P1'')
//        WAR: a half-tile is re-staged >= 2 phases after its last ds_read  (W1: read P2, staged P1'; A1: P3/P2'; A0: P1/P3; W0: P1/P4)
//    Past the last k-tile the phases stage zero chunks so the counted waits stay exact.
#pragma once
#include "gemm.hpp"

#ifdef GEMM_PROBE_VARIANTS
static __device__ int g_tile_gm = 0;
#endif
// SPLITK is a compile-time form: the k-range arithmetic it adds to the staging segments of every phase cost the plain kernel
// 1.7 % of the c3 step when it was a run-time condition
template <int WR, int WC, int TM, int TN>
struct Gemm8pCfg {
    static constexpr int BM = WR * TM * 16, BN = WC * TN * 16;
    static constexpr int A_HALF = WR * TM * 8 * 128, W_HALF = (WC * TN * 8 + 63) / 64 * 64 * 128;
    static constexpr int LDS_BYTES = 2 * (2 * A_HALF + 2 * W_HALF);
};
// TM x TN (round 3): MFMA tiles per wave.  8 x 4 (wave tile 128 x 64) is the default form described above; 4 x 6 (wave tile
// 64 x 96, WR = 4, WC = 2: workgroup tile 256 x 192) exists for the encoder's N = 1408 / 4608 linears at batch 4: 256 x 256 tiles
// give 192 / 576 tiles on 256 CUs (0.75 / 2.25 rounds: a CU that has a tile computes 65 536 outputs where its fair share is 45 056),
// 256 x 192 tiles give 256 / 768 (whole rounds of 49 152 outputs).  Same phases with quadrants of (TM/2) x (TN/2) tiles.
// PERSIST (round 3 EXPERIMENT, measured slower and not instantiated by default - see launch_8p; MODE 0 without split-K): the grid is one workgroup per CU and a workgroup walks tiles bid, bid + gridDim.x, ...
// Between the main loop of a tile and its epilogue the NEXT tile's prologue (k-tile 0 and half of k-tile 1: six half-tile stages) is
// requested - the LDS buffers are free by then and no epilogue of this mode touches LDS - so the next main loop starts on data that
// landed under the epilogue instead of paying a workgroup launch + an exposed HBM / L2 round trip per tile.  The per-lane source
// pointers of the next tile live only across those six stage calls (they are recomputed at the top of the next iteration from an
// opaque copy of the tile origin): the epilogue's register budget is unchanged.  Matters most where tiles are short: the mask
// product (K = 352: 5.5 k-tiles per tile, 48 - 96 tiles per CU).
template <typename T, int MODE, int WR, int WC, bool SPLITK = false, int TM = 8, int TN = 4, bool PERSIST = false>
__global__ __launch_bounds__(512) void gemm8p_kernel(const GemmParams p) {
    static_assert(!PERSIST || (MODE == 0 && !SPLITK), "persistent form: dense GEMM without split-K");
    static_assert(WR * WC == 8 && (WC == 2 || WC == 4), "8 waves");
    static_assert(TM % 2 == 0 && TN % 2 == 0 && (WR * TM) % 8 == 0, "half tiles; A half-tile = whole 64-row staging passes");
    constexpr int BM = WR * TM * 16, BN = WC * TN * 16, BK = 64;
    constexpr int RH = TM * 8, CH = TN * 8;               // rows of A / of W that one wave owns in a half-tile
    constexpr int A_PASS = WR * RH / 64, W_PASS = (WC * CH + 63) / 64;  // 512-lane LDS-DMA passes (64 rows each) per half-tile
    // (a W half-tile is allocated as whole passes: rows past WC * CH are filler - zero chunks, never read - so that every
    //  lane issues the same number of requests and the counted vmcnt waits hold for all waves)
    constexpr int A_HALF = WR * RH * 128, W_HALF = W_PASS * 64 * 128;  // bytes
    constexpr int BUF = 2 * A_HALF + 2 * W_HALF;
    constexpr int INFLIGHT = 2 * A_PASS + 2 * W_PASS;    // loads of the four most recent half-tiles
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][A0 | A1 | W0 | W1]

    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int wr = wave / WC, wc = wave % WC, grp = wave >> 2;
    // per-lane constants (lane_setup): PERSIST re-derives them from an opaque copy of the thread index after every epilogue, so that
    // none of them - and nothing the compiler would hoist out of the tile loop on their account - is live across an epilogue
    int li, kg, srow, slot, cs_a, cs_w[W_PASS], a_off[2], w_off[2];
    auto lane_setup = [&](int tid) {
        const int lane = tid & 63;
        li = lane & 15, kg = lane >> 4;
        srow = tid >> 3, slot = tid & 7;
        cs_a = slot ^ ((srow >> 1) & 7);
        // (W rows are stored per wave column block as (q, 4 jj + r): q = the lane group that owns the column in the epilogue; the
        //  XOR phase is distinct over (q, r >> 1), conflict-free for the fragment reads of either wave tile shape - brute-forced)
#pragma unroll
        for (int i = 0; i < W_PASS; ++i) {
            const int lr = srow + 64 * i;
            cs_w[i] = slot ^ (((((lr % CH) / (2 * TN)) & 3) << 1) | ((lr >> 1) & 1));
        }
        // fragment read offsets (bytes inside a half-tile); the k-step's chunk is XORed with the row phase
        const int sw_a = (li >> 1) & 7, sw_w = (((li >> 2) & 3) << 1) | ((li >> 1) & 1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            a_off[kk] = (wr * RH + li) * 128 + (((kk * 4 + kg) ^ sw_a) << 4);                                // + ii * 2048
            w_off[kk] = (wc * CH + 2 * TN * (li >> 2) + (li & 3)) * 128 + (((kk * 4 + kg) ^ sw_w) << 4);  // + jj * 512
        }
    };
    lane_setup((int)threadIdx.x);

    // ---- workgroup -> tile: XCD x gets a contiguous range of tiles (workgroup b runs on XCD b % 8), n fastest, so the
    //      tiles that share an A row panel / the W panels stay inside one L2 ----
    const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM, ntiles = ntn * ntm;
    // split-K (few tiles, long K: the batch-1 MLP-out projection): slice ksplit of a tile contracts k-tiles [kt0, kt0 + nk) and
    // leaves a float partial that splitk_finish_kernel sums (fixed order) and finishes, as in gemm_kernel
    const int nsplit = SPLITK ? p.splitk : 1;
    const int ksplit = SPLITK ? (int)blockIdx.x / ntiles : 0;
    const int bid0 = SPLITK ? (int)blockIdx.x - ksplit * ntiles : (int)blockIdx.x;
    // Linear order of the tiles: column BANDS of <= 8 tile columns, inside a band m-major with the band's columns fastest.
    // The ~32 tiles an XCD runs at a time are consecutive in this order, i.e. a block of ~(32 / band width) tile rows x the
    // band: they share that many A panels and <= 8 W panels, all walking k together, so each panel k-tile is fetched into the
    // XCD's L2 once for the whole block.  (Plain row-major order made a round touch every W panel of a wide N: 2 + 24 panels
    // for the MLP's first linear instead of 4 + 8; PMC: 2.4x the operand bytes fetched.)
    // (PERSIST: workgroup b's j-th tile is "workgroup" b + j * gridDim.x of the plain launch; gridDim.x is a multiple of 8, so
    //  the XCD of a tile and the block of tiles an XCD works on at a time are those of the plain launch)
    auto decode_tile = [&](int bid, int& m0_, int& n0_) {
        int tile;
        {
            const int xcd = bid & 7, idx = bid >> 3, q = ntiles >> 3, r = ntiles & 7;
            tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        }
        int mt, nt_;
        {
            const int nb = (ntn + 7) >> 3, bw = (ntn + nb - 1) / nb, full = (nb - 1) * ntm * bw;
#ifdef GEMM_PROBE_VARIANTS
            const bool banded = g_tile_gm >= 0;
#else
            const bool banded = true;
#endif
            if (!banded || nb == 1) {
                mt = tile / ntn;
                nt_ = tile % ntn;
            } else if (tile < full) {
                const int band = tile / (ntm * bw), rem = tile - band * (ntm * bw);
                mt = rem / bw;
                nt_ = band * bw + rem % bw;
            } else {
                const int w = ntn - (nb - 1) * bw, rem = tile - full;  // last band: the remaining w columns
                mt = rem / w;
                nt_ = (nb - 1) * bw + rem % w;
            }
        }
        if (MODE == 1) mt = conv_tile_walk(p, mt, BM, 2);
        m0_ = mt * BM, n0_ = nt_ * BN;
    };
    int bid = bid0, m0, n0;
    decode_tile(bid, m0, n0);

    // ---- staging sources (one 16-byte chunk per lane per pass) ----
    const char* a_src[A_PASS][2];
    unsigned a_mask[A_PASS][2];
    const char* w_src[W_PASS][2];
    auto setup_src = [&](int m0, int n0) {
#pragma unroll
    for (int i = 0; i < A_PASS; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int lra = i * 64 + srow;  // LDS row of the A half-tile: (wave row block, row inside the block's half)
            int m = m0 + (lra / RH) * (TM * 16) + h * RH + lra % RH;
            if (m >= p.M) m = p.M - 1;
            if (MODE == 0) {
                const long long pm = p.a_gr > 0 ? (long long)(m / p.a_gr) * p.a_gs + p.a_go + (m % p.a_gr) : m;
                a_src[i][h] = (const char*)((const T*)p.A + pm * p.lda + cs_a * 8);
                a_mask[i][h] = 0;
            } else {
                int wo = m % p.Wo;
                int r = m / p.Wo;
                int ho = r % p.Ho;
                r /= p.Ho;
                int to = r % p.To;
                int b = r / p.To;
                const int ti = to * p.st, hi = ho * p.sh, wi = wo * p.sw;
                unsigned mask = 0;
#pragma unroll
                for (int tap = 0; tap < 27; ++tap) {
                    const int dt = tap / 9 - 1, dh = (tap / 3) % 3 - 1, dw = tap % 3 - 1;
                    const bool ok = (unsigned)(ti + dt) < (unsigned)p.Ti && (unsigned)(hi + dh) < (unsigned)p.Hi &&
                                    (unsigned)(wi + dw) < (unsigned)p.Wi;
                    mask |= (ok ? 1u : 0u) << tap;
                }
                a_mask[i][h] = mask;
                const long long vox = (((long long)b * p.Ti + ti) * p.Hi + hi) * p.Wi + wi;
                a_src[i][h] = (const char*)((const T*)p.A + vox * p.Cin + cs_a * 8);
            }
        }
#pragma unroll
    for (int i = 0; i < W_PASS; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int lr = srow + 64 * i, within = lr % CH;
            int n = n0 + (lr / CH) * (TN * 16) + (within / (2 * TN)) * (4 * TN) + 4 * (h * (TN / 2)) + within % (2 * TN);
            if (n >= p.N) n = p.N - 1;  // columns past N are computed on a clamped row and never stored
            w_src[i][h] = (const char*)((const T*)p.W + (long long)n * p.ldw + cs_w[i] * 8);
            if (lr >= WC * CH) w_src[i][h] = nullptr;  // filler rows of the padded half-tile
        }
    };
    setup_src(m0, n0);

    const int nk_all = (p.K + BK - 1) / BK;
    const int kt0 = SPLITK ? (int)((long long)nk_all * ksplit / nsplit) : 0;
    const int nk = SPLITK ? (int)((long long)nk_all * (ksplit + 1) / nsplit) - kt0 : nk_all;
    const int kpc = (MODE == 1) ? (p.Cin / BK) : 1;
    const char* zero = (const char*)g_zero_chunk;

    // Conv (MODE 1): k-tile kt is (tap kt / kpc, channel slice kt % kpc).  Both A streams (h = 0, 1) are staged in
    // k-tile order, one k-tile further at every call, so (tap, channel offset, tap address offset) are carried as wave-uniform
    // counters instead of being re-derived from kt with an integer division and 64-bit multiplies at every call (that
    // arithmetic sat in the staging segment of every phase: ~40 instructions per call).
    struct ConvPos {
        int kt, tap, ci0;
        long long off;  // byte offset of (tap, ci0) relative to the lane's centre voxel
    } cpos[2];
    auto conv_tap_off = [&](int tap) {
        const int dt = tap / 9 - 1, dh = (tap / 3) % 3 - 1, dw = tap % 3 - 1;
        return ((((long long)dt * p.Hi + dh) * p.Wi + dw) * p.Cin) * 2;
    };
    if (MODE == 1) {
#pragma unroll
        for (int h = 0; h < 2; ++h) cpos[h] = ConvPos{0, 0, 0, conv_tap_off(0)};
    }
    auto stage_a = [&](int h, int kt, int buf) {
        long long off;
        int tap = 0;
        if (MODE == 0) {
            off = (long long)(kt0 + kt) * (BK * 2);
        } else {
            ConvPos& c = cpos[h];
            while (c.kt < kt0 + kt) {  // (one step per call in the main loop; the prologue's first calls start at 0)
                ++c.kt;
                c.ci0 += BK;
                if (c.ci0 == p.Cin) {
                    c.ci0 = 0;
                    ++c.tap;
                    c.off = conv_tap_off(c.tap < 27 ? c.tap : 26);
                }
            }
            tap = c.tap;
            off = c.off + (long long)c.ci0 * 2;
        }
#pragma unroll
        for (int i = 0; i < A_PASS; ++i) {
            bool ok;
            if (MODE == 0)
                ok = (!SPLITK || kt < nk) && (kt0 + kt) * BK + cs_a * 8 < p.K;
            else
                ok = kt < nk && ((a_mask[i][h] >> tap) & 1u);
            const char* src = ok ? a_src[i][h] + off : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + buf * BUF + h * A_HALF + (i * 512 + wave * 64) * 16), 16,
                                             0, 0);
        }
    };
    auto stage_w = [&](int h, int kt, int buf) {
#pragma unroll
        for (int i = 0; i < W_PASS; ++i) {
            const bool ok = (!SPLITK || kt < nk) && (kt0 + kt) * BK + cs_w[i] * 8 < p.K && w_src[i][h] != nullptr;
            const char* src = ok ? w_src[i][h] + (long long)(kt0 + kt) * (BK * 2) : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)src,
                                             (lptr_t)(smem + buf * BUF + 2 * A_HALF + h * W_HALF + (i * 512 + wave * 64) * 16), 16, 0, 0);
        }
    };

    f32x4 acc[TM][TN];
    vec8<T> xa[TM / 2][2], wb[2][TN / 2][2];

    auto read_a = [&](int h, int buf) {
        const char* base = smem + buf * BUF + h * A_HALF;
#pragma unroll
        for (int ii = 0; ii < TM / 2; ++ii)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) xa[ii][kk] = *(const vec8<T>*)(base + a_off[kk] + ii * 2048);
    };
    auto read_w = [&](int h, int buf) {
        const char* base = smem + buf * BUF + 2 * A_HALF + h * W_HALF;
#pragma unroll
        for (int jj = 0; jj < TN / 2; ++jj)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) wb[h][jj][kk] = *(const vec8<T>*)(base + w_off[kk] + jj * 512);
    };
    auto quadrant = [&](int qa, int qw) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int ii = 0; ii < TM / 2; ++ii)
#pragma unroll
                for (int jj = 0; jj < TN / 2; ++jj)
                    acc[qa * (TM / 2) + ii][qw * (TN / 2) + jj] =
                        mma16(wb[qw][jj][kk], xa[ii][kk], acc[qa * (TM / 2) + ii][qw * (TN / 2) + jj]);
        __builtin_amdgcn_s_setprio(0);
    };
    // end of a phase's read/stage segment: counted wait for the half-tile staged four phases ago, then the barrier
    auto seg_end = [&]() {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto phase_end = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: k-tile 0 complete, A0/W0 of k-tile 1 ----
    auto prologue_stage = [&]() {
        stage_a(0, 0, 0);
        stage_w(0, 0, 0);
        stage_w(1, 0, 0);
        stage_a(1, 0, 0);
        stage_a(0, 1, 1);
        stage_w(0, 1, 1);
    };
    prologue_stage();
    for (;;) {  // tile loop (one pass unless PERSIST)
    // (past a seam the epilogue's stores are younger than the prologue's requests and memory operations retire in order: the
    //  counted wait still means "at most the four youngest half-tile stages are outstanding")
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");  // A0(0), W0(0) landed
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 1) __builtin_amdgcn_s_barrier();  // second wave group runs one barrier behind
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto ktile = [&](int t, auto cur_c) {
        constexpr int cur = decltype(cur_c)::value, nxt = cur ^ 1;
        // P1
        read_w(0, cur);
        read_a(0, cur);
        stage_w(1, t + 1, nxt);
        seg_end();
        quadrant(0, 0);
        phase_end();
        // P2
        read_w(1, cur);
        stage_a(1, t + 1, nxt);
        seg_end();
        quadrant(0, 1);
        phase_end();
        // P3
        read_a(1, cur);
        stage_a(0, t + 2, cur);
        seg_end();
        quadrant(1, 1);
        phase_end();
        // P4
        stage_w(0, t + 2, cur);
        seg_end();
        quadrant(1, 0);
        phase_end();
    };
    for (int t = 0; t < nk; t += 2) {
        ktile(t, std::integral_constant<int, 0>{});
        if (t + 1 < nk) ktile(t + 1, std::integral_constant<int, 1>{});
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();  // pairs with the trailing barrier of the delayed group
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (only zero-chunk dummies are still in flight)

    // ---- seam (PERSIST): every wave is past its last fragment read (the barrier above), the LDS-DMA queue is empty: request the
    //      next tile's prologue now, then run this tile's epilogue under it ----
    int nm0 = 0, nn0 = 0;
    bool more = false;
    if constexpr (PERSIST) {
        const int nbid = bid + (int)gridDim.x;
        more = nbid < ntiles;
        if (more) {
            bid = nbid;
            decode_tile(nbid, nm0, nn0);
            setup_src(nm0, nn0);
            prologue_stage();
        }
    }

    if constexpr (SPLITK) {  // raw float partial [ksplit][M][N]; bias / activation / residuals / conversion: splitk_finish_kernel
        const int nb2 = n0 + wc * (TN * 16) + 4 * TN * kg;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wr * (TM * 16) + i * 16 + li;
            if (m >= p.M) continue;
            float* pp = p.partial + ((long long)ksplit * p.M + m) * p.N + nb2;
#pragma unroll
            for (int j = 0; j < TN; ++j)
                if (nb2 + 4 * j < p.N) *(f32x4*)(pp + 4 * j) = acc[i][j];
        }
        return;
    }
#ifdef GEMM_DBG_NOEPI  // (tools/probes/gemm_variants.hip: main-loop-only timing)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(acc[i][j]));
#else
#ifdef GEMM_DBG_ONLY_FAST  // (probe: the kernel with nothing but the lean bias + store epilogue)
    gemm_epilogue_dense<T, TM, TN, ACT_NONE, 0>(p, acc, m0 + wr * (TM * 16), n0 + wc * (TN * 16), li, kg);
    return;
#endif
    // (the mask-product epilogue exists for the 8 x 4 wave tile only - launch_gemm never sends it to another form)
    if (TN == 4 && p.epi == EPI_MASKDOT) {
        if constexpr (TN == 4) gemm_epilogue_maskdot<T, TM, TN>(p, acc, m0 + wr * (TM * 16), n0 + wc * (TN * 16), li, kg);
    } else if (!gemm_epilogue_dense_dispatch<T, TM, TN>(p, acc, m0 + wr * (TM * 16), n0 + wc * (TN * 16), li, kg)) {
        // (the generic row body - every scatter form, row map and residual kind decided at run time - is compiled into the 8 x 4
        //  form only: beside the 4 x 6 form's lean epilogues it pushed hipcc into scratch; launch_gemm sends the 4 x 6 form
        //  lean epilogues only, epilogue_is_lean_8p)
        if constexpr (TN == 4) gemm_epilogue<T, TM, TN, true>(p, acc, m0 + wr * (TM * 16), n0 + wc * (TN * 16), li, kg);
    }
#endif
    if constexpr (!PERSIST) {
        break;
    } else {
        if (!more) break;
        // the next tile becomes the current one; its source pointers are recomputed from an OPAQUE copy of its origin, so that
        // the ones computed at the seam are dead across the epilogue
        m0 = nm0, n0 = nn0;
        asm volatile("" : "+s"(m0), "+s"(n0));
        int t_ = (int)threadIdx.x;
        asm volatile("" : "+v"(t_));
        lane_setup(t_);
        setup_src(m0, n0);
    }
    }  // tile loop
}
