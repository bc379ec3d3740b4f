// 3x3x3 conv (stride 1, pad 1, channels-last, bf16) with the input block staged ONCE per channel slice in LDS ("halo"
// tile) and the 27 taps walked out of LDS — the full-resolution convs of the DPT decoders (reference dpt_block.py:110-157
// ResidualConvUnit, :406-414 head; dpt_head.py:41-86).  Same math, descriptor and epilogues as the implicit-GEMM form
// (gemm.hpp / gemm8p.hpp MODE 1), different data movement:
//
//  * implicit GEMM streams, per (tap, channel slice), the A rows of its tile from global memory: every input element
//    travels L2 -> LDS 27 times (PMC: 7x the operand bytes even reach the fabric) and the A stream is half of the
//    kernel's L2 -> LDS traffic.
//  * here a workgroup owns an output BLOCK of TT x TH x TW = 2 x (8|16) x 16 voxels and all BN = 256 | 128 output
//    channels.  For a slice of CK = 32 input channels it holds the (TT+2) x (TH+2) x (TW+2) halo block in LDS (64-byte
//    rows, zero rows outside the volume: no masks in the loop) and runs the 27 taps as 27 k-tiles whose A fragments are
//    ds_read_b128 at a tap-dependent row offset.  Only the weights stream: one BN x 32 tile per k-tile through a 4-slot
//    LDS-DMA ring.  L2 -> LDS bytes per MFMA drop by ~45 % (256 x 256: 64 KB -> 35 KB per 64-deep k step).
//  * the halo is a SINGLE buffer refilled plane by plane while it is in use: t-plane 0 is only read by the nine dt = -1
//    taps and plane 1 by the eighteen dt <= 0 taps, so the next slice's planes 0 / 1 are fetched during taps 9.. / 18..
//    of the current slice, and planes 2 / 3 during the first taps of their own slice (first needed at tap 9 / 18).
//  * LDS rows are 64 bytes = 4 chunks; the chunk is XORed with ((x >> 2) & 1) << 1, x = the row's position along w inside
//    its halo row (W tiles: the MFMA row).  A fragment is 16 consecutive rows of one halo row, so among the four lanes
//    that share a 256-byte bank row (rows 4 apart) the XOR term alternates, which makes the four 16-lane groups of
//    ds_read_b128 conflict-free at ANY row alignment, i.e. for every tap offset (checked exhaustively).  The term depends
//    on the lane and on dw only: a lane keeps three fragment base addresses (dw = -1, 0, 1) and a k-tile's eight A
//    fragment reads are ONE add + immediates.  W tiles are stored in fragment order (LDS row = wave column block * 64 +
//    j * 16 + MFMA row).  All staging is LDS-DMA with the swizzle applied to the per-lane SOURCE chunk.
//  * pipeline: the 8-phase kernel's structure with a 32-deep k-tile = two phases of 16 MFMAs per wave
//        P1: read W(t) + A rows 0-63 | stage W(t+3)        | barrier | 16 MFMAs | barrier
//        P2: read A rows 64-127      | stage a halo piece  | vmcnt   | barrier | 16 MFMAs | barrier
//    the two 4-wave groups run one barrier apart (one wave's MFMA segment over its SIMD partner's read / stage
//    segment).  Hazards: W(t+3) overwrites the slot of W(t-1), last read three barriers earlier by the delayed group;
//    W(t+1) is waited for (vmcnt(2 * W_PASS): everything but the two youngest W tiles; halo pieces in flight only make
//    the wait stricter) at the end of P2(t) and first read after the following barrier; a halo plane is refilled >= 4
//    barriers after its last read and first read >= 6 k-tiles after its refill was issued.
#pragma once
#include "gemm.hpp"

template <int WR, int WC>
struct ConvHaloCfg {
    static constexpr int BM = WR * 128, BN = WC * 64, CK = 32;
    static constexpr int TT = 2, TW = 16, TH = BM / (TT * TW);
    static constexpr int HH = TH + 2, HW = TW + 2, PR = HH * HW, NPL = TT + 2;
    static constexpr int PRP = (PR + 15) / 16 * 16;  // plane stride in rows: whole waves of LDS-DMA slots (4 slots per row), the
                                                      // rows PR .. PRP-1 are filler that is written (zeros) and never read
    static constexpr int HALO_BYTES = NPL * PRP * 64;
    static constexpr int WSLOT = BN * 64, NSLOT = 4;
    static constexpr int LDS_BYTES = HALO_BYTES + NSLOT * WSLOT;
    static constexpr int W_PASS = BN * 4 / 512;            // LDS-DMA instructions per lane and W tile
    static constexpr int H_PASS = (PRP * 4 + 511) / 512;   // ... per halo plane
};

// UPS (round 5): the halo block is not copied but COMPUTED - the conv's input is the bilinear (align_corners) up-sampling of the
// low-resolution volume p.A [B][Ti][ups_hi][ups_wi][Cin] to (Hi, Wi) (dpt_head.py:79-84: interpolate -> head conv), which then never
// exists in memory (822 MB written and read per dense head and step at the full geometry).  A halo slot = 8 channels of one voxel:
// its lane requests the four source taps (4 x 16 bytes) where the copy form issued one LDS-DMA, and one k-tile later - the loads have
// landed behind that k-tile's MFMAs - forms sum w_ab * v_ab in the order and with the rounding of upsample_line (dpt_ops.hip: the
// fused conv equals up-sample + conv bit for bit) and writes the 16 bytes into the halo with ds_write_b128.  The refill schedule and
// its hazards are the copy form's with every write one k-tile later (later than a plane's last read, >= 5 k-tiles before its first).
template <typename T, int WR, int WC, bool UPS = false>
__global__ __launch_bounds__(512) void conv3_halo_kernel(const GemmParams p) {
    static_assert(WR * WC == 8 && (WC == 2 || WC == 4), "8 waves");
    typedef ConvHaloCfg<WR, WC> Cfg;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, TM = 8, TN = 4;
    constexpr int TT = Cfg::TT, TH = Cfg::TH, TW = Cfg::TW, HH = Cfg::HH, HW = Cfg::HW, PR = Cfg::PR, PRP = Cfg::PRP;
    constexpr int WSLOT = Cfg::WSLOT, W_PASS = Cfg::W_PASS, H_PASS = Cfg::H_PASS;
    static_assert(2 * H_PASS <= 9, "planes 2 / 3 are refilled before tap 9");
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];  // [halo NPL x PRP rows x 64 B][W ring 4 x BN x 64 B]
    char* const wring = smem + Cfg::HALO_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC, grp = wave >> 2;
    const int li = lane & 15, kg = lane >> 4;

    // ---- workgroup -> output block: XCD x (workgroup b runs on XCD b % 8) owns a contiguous range of blocks in the order
    //      (batch, h block, w block, t block fastest - below): the ~32 blocks an XCD runs at a time are a compact slab whose halos
    //      overlap in its L2 ----
    const int nbw = p.Wo / TW, nbh = p.Ho / TH, nbt = p.To / TT;
    const int ntiles = (p.M / BM);
    int tile;
    {
        const int bid = (int)blockIdx.x, xcd = bid & 7, idx = bid >> 3, q = ntiles >> 3, r = ntiles & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // order of the blocks: T fastest, then w, then h (round 6; round 3 had w, t, h).  The ~32 blocks an XCD runs at a time are then ALL
    // t blocks of a few neighbouring w positions: the two halo planes a block shares with each t neighbour (half of its halo) and
    // the halo columns it shares with its w neighbours are fetched into the XCD's L2 once.  Expected fetch per output plane:
    // 18/16 (t) x ~1.03 (w) x 18/16 (h) = 1.30 against 1.6 for the w-fastest order, whose 32 resident blocks are 14 w positions
    // x 2.3 t blocks (PMC, round 5: 1216 MB fetched per head-conv launch against 680 MB of operands).  -DCONV_HALO_ORDER_WTH: round 3's.
    int rem = tile;
#ifdef CONV_HALO_ORDER_WTH
    const int bw = rem % nbw;
    rem /= nbw;
    const int bt = rem % nbt;
    rem /= nbt;
#else
    const int bt = rem % nbt;
    rem /= nbt;
    const int bw = rem % nbw;
    rem /= nbw;
#endif
    const int bh = rem % nbh, bb = rem / nbh;
    const int t0 = bt * TT, h0 = bh * TH, w0 = bw * TW;

    // ---- W staging: LDS slot s = pass * 512 + tid  ->  (LDS row s >> 2 = colblock * 64 + j * 16 + a, physical chunk s & 3) ----
    const char* w_src[W_PASS];
#pragma unroll
    for (int i = 0; i < W_PASS; ++i) {
        const int s = i * 512 + tid, row = s >> 2, cp = s & 3;
        const int cb = row >> 6, j = (row >> 4) & 3, a = row & 15;
        const int n = cb * 64 + 16 * (a >> 2) + 4 * j + (a & 3);  // the output column this MFMA row accumulates (gemm.hpp)
        const int c = cp ^ (((a >> 2) & 1) << 1);
        w_src[i] = (const char*)((const T*)p.W + (long long)n * p.ldw) + c * 16;
    }
    // ---- halo staging: slot s = pass * 512 + tid of a plane -> (row r = s >> 2 = hh * HW + hw, physical chunk s & 3) ----
    unsigned h_off[H_PASS];  // byte offset of (h, w, chunk) inside a (b, t) plane of the input
    bool h_ok[H_PASS];
#pragma unroll
    for (int i = 0; i < H_PASS; ++i) {
        const int s = i * 512 + tid, r = s >> 2;
        const int hh = r / HW, hw = r - hh * HW;
        const int gh = h0 - 1 + hh, gw = w0 - 1 + hw;
        h_ok[i] = s < PR * 4 && (unsigned)gh < (unsigned)p.Hi && (unsigned)gw < (unsigned)p.Wi;
        const int c = (s & 3) ^ (((hw >> 2) & 1) << 1);  // logical chunk held by this physical slot
        h_off[i] = h_ok[i] ? (unsigned)(((long long)gh * p.Wi + gw) * p.Cin * 2 + c * 16) : 0u;
    }
    const long long plane_bytes = UPS ? (long long)p.ups_hi * p.ups_wi * p.Cin * 2 : (long long)p.Hi * p.Wi * p.Cin * 2;
    const char* zero = (const char*)g_zero_chunk;
    // UPS: source taps of this lane's slot of pass i: byte offset of tap (y0, x0) inside a low-resolution (b, t) plane, the byte
    // steps to y1 / x1 (0 at the border) and the two interpolation weights (ATen's index / lambda arithmetic, dpt_ops.hip src_index)
    unsigned u_off[UPS ? H_PASS : 1];  // (a multiple of 16: bit 0 = the y1 tap is one row on, bit 1 = the x1 tap is one voxel on)
    float u_lh[UPS ? H_PASS : 1], u_lw[UPS ? H_PASS : 1];
    if constexpr (UPS) {
#pragma unroll
        for (int i = 0; i < H_PASS; ++i) {
            const int s = i * 512 + tid, r = s >> 2;
            const int hh = r / HW, hw = r - hh * HW;
            const int gh = h0 - 1 + hh, gw = w0 - 1 + hw;
            const int c = (s & 3) ^ (((hw >> 2) & 1) << 1);
            auto src_index = [](int dst, int in, int out, int& i0, int& i1, float& lam) {
                const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
                const float src = scale * (float)dst;
                i0 = (int)src;
                if (i0 > in - 1) i0 = in - 1;
                lam = fminf(fmaxf(src - (float)i0, 0.f), 1.f);
                i1 = i0 + (i0 < in - 1 ? 1 : 0);
            };
            int y0 = 0, y1 = 0, x0 = 0, x1 = 0;
            float lh = 0.f, lw = 0.f;
            if (h_ok[i]) {
                src_index(gh, p.ups_hi, p.Hi, y0, y1, lh);
                src_index(gw, p.ups_wi, p.Wi, x0, x1, lw);
            }
            u_off[i] = (unsigned)(((long long)y0 * p.ups_wi + x0) * p.Cin * 2 + c * 16) | (y1 > y0 ? 1u : 0u) | (x1 > x0 ? 2u : 0u);
            u_lh[i] = lh, u_lw[i] = lw;
        }
    }
    // a pass = this lane's four taps, where the slot goes and whether it lies inside the volume
    struct UpsPass {
        u32x4 v[4];
        int lds;  // LDS byte offset of the slot (-1: nothing pending)
        bool ok;
        float lh, lw;
    };
    auto ups_load = [&](int plane, int cs, int pass, UpsPass& u) {
#pragma unroll
        for (int i = 0; i < H_PASS; ++i) {
            if (i != pass) continue;
            if (i * 512 + wave * 64 >= PRP * 4) continue;
            const int gt = t0 - 1 + plane;
            u.ok = h_ok[i] && (unsigned)gt < (unsigned)p.Ti;
            const char* src = (const char*)p.A + ((long long)bb * p.Ti + (u.ok ? gt : 0)) * plane_bytes + (u_off[i] & ~15u) + cs * 64;
            const unsigned dy = (u_off[i] & 1u) ? (unsigned)(p.ups_wi * p.Cin * 2) : 0u, dx = (u_off[i] & 2u) ? (unsigned)(p.Cin * 2) : 0u;
            u.v[0] = *(const u32x4*)src;
            u.v[1] = *(const u32x4*)(src + dx);
            u.v[2] = *(const u32x4*)(src + dy);
            u.v[3] = *(const u32x4*)(src + dy + dx);
            u.lh = u_lh[i], u.lw = u_lw[i];
            u.lds = (plane * PRP * 4 + i * 512 + tid) * 16;
        }
    };
    auto ups_store = [&](UpsPass& u) {
        if (u.lds < 0) return;
        // upsample_line<T, .., NT = 1, NH>: acc += (wh[a] * ww[b]) * v[a][b] over (a, b) = (0,0) (0,1) (1,0) (1,1), fused multiply-adds;
        // a zero weight adds exactly nothing, as the tap that form skips
        const float wh[2] = {1.f - u.lh, u.lh}, ww[2] = {1.f - u.lw, u.lw};
        // (the four products as single v_mul_f32: the vectoriser made packed multiplies with a swizzled operand of them - the form
        //  tools/check_isa.py bans)
        float wgt[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) asm("v_mul_f32 %0, %1, %2" : "=v"(wgt[a][b]) : "v"(wh[a]), "v"(ww[b]));
        vec8<T> o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float acc = 0.f;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const vec8<T> v = __builtin_bit_cast(vec8<T>, u.v[2 * a + b]);
                    acc = __builtin_fmaf(wgt[a][b], (float)v[k], acc);
                }
            o[k] = (T)acc;
        }
        const u32x4 zz = {0u, 0u, 0u, 0u};
        *(u32x4*)(smem + u.lds) = u.ok ? __builtin_bit_cast(u32x4, o) : zz;
        u.lds = -1;
    };
    UpsPass ub;  // the pass in flight in the main loop
    ub.lds = -1;
    auto ups_issue = [&](int plane, int cs, int pass) { ups_load(plane, cs, pass, ub); };
    auto ups_finish = [&]() { ups_store(ub); };
    auto stage_halo = [&](int plane, int cs, int pass) {  // one LDS-DMA instruction (per lane) of a halo plane
#pragma unroll
        for (int i = 0; i < H_PASS; ++i) {
            if (i != pass) continue;
            if (i * 512 + wave * 64 >= PRP * 4) continue;  // (whole waves past the plane issue nothing: wave-uniform)
            const int gt = t0 - 1 + plane;
            const bool ok = h_ok[i] && (unsigned)gt < (unsigned)p.Ti;
            const char* src = ok ? (const char*)p.A + ((long long)bb * p.Ti + gt) * plane_bytes + h_off[i] + cs * 64 : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + (plane * PRP * 4 + i * 512 + wave * 64) * 16), 16, 0, 0);
        }
    };
    const int ncs = p.Cin / 32, nk = ncs * 27;
    auto stage_w = [&](int t, int koff) {  // W tile of k-tile t (byte offset koff of its k range) into ring slot t & 3
#pragma unroll
        for (int i = 0; i < W_PASS; ++i) {
            const char* src = t < nk ? w_src[i] + koff : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(wring + (t & 3) * WSLOT + (i * 512 + wave * 64) * 16), 16, 0, 0);
        }
    };

    // ---- fragment read addresses ----
    // A: fragment f (0..7) of this wave = output rows (ft, fh) = block row wr * 8 + f, 16 voxels along w; its centre halo row
    const int F0 = wr * 8;
    const int rho_c = (F0 / TH + 1) * PRP + (F0 % TH + 1) * HW + 1 + li;  // + f * HW
    int a_base[3];  // byte address of fragment 0 at tap (0, 0, dw): the swizzle term follows the w position 1 + dw + li
#pragma unroll
    for (int d = 0; d < 3; ++d) a_base[d] = (rho_c + d - 1) * 64 + ((kg ^ ((((li + d) >> 2) & 1) << 1)) << 4);
    // W: fragment j = LDS rows wc * 64 + j * 16 + li
    const int w_off = (wc * 64 + li) * 64 + ((kg ^ (((li >> 2) & 1) << 1)) << 4);  // + j * 1024

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    vec8<T> xa[4], wb[4];

    auto read_w = [&](int t) {
        const char* base = wring + (t & 3) * WSLOT + w_off;
#pragma unroll
        for (int j = 0; j < 4; ++j) wb[j] = *(const vec8<T>*)(base + j * 1024);
    };
    auto read_a = [&](int half, int abase) {  // fragments half * 4 .. + 3; abase = this tap's address of fragment 0
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) xa[ii] = *(const vec8<T>*)(smem + abase + (half * 4 + ii) * (HW * 64));
    };
    auto mfma16 = [&](int half) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[half * 4 + ii][j] = mma16(wb[j], xa[ii], acc[half * 4 + ii][j]);
        __builtin_amdgcn_s_setprio(0);
    };
    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: the whole halo of slice 0 and W(0..2) ----
    const int tap_bytes = p.Cin * 2;
    if constexpr (UPS) {
        stage_w(0, 0);
        stage_w(1, tap_bytes);
        stage_w(2, 2 * tap_bytes);
        // (the accumulators are not live yet: two planes' passes are requested together before the first is consumed)
#pragma unroll
        for (int pl = 0; pl < Cfg::NPL; ++pl) {
            UpsPass pb[H_PASS];
#pragma unroll
            for (int i = 0; i < H_PASS; ++i) {
                pb[i].lds = -1;
                ups_load(pl, 0, i, pb[i]);
            }
#pragma unroll
            for (int i = 0; i < H_PASS; ++i) ups_store(pb[i]);
        }
    } else {
#pragma unroll
        for (int pl = 0; pl < Cfg::NPL; ++pl)
#pragma unroll
            for (int i = 0; i < H_PASS; ++i) stage_halo(pl, 0, i);
        stage_w(0, 0);
        stage_w(1, tap_bytes);
        stage_w(2, 2 * tap_bytes);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 1) __builtin_amdgcn_s_barrier();  // second wave group runs one barrier behind
    __builtin_amdgcn_sched_barrier(0);

    int tap = 0, cs = 0;          // k-tile t = cs * 27 + tap = tap (dt, dh, dw) of channel slice cs
    int dw = 0, dh = 0;           // dw + 1, dh + 1 of the tap
    int toff = (-PRP - HW) * 64;  // byte offset of the tap's (dt, dh) rows relative to the centre (dw is in a_base)
    int tap3 = 3, koff3 = 3 * tap_bytes;  // k-tile t + 3 (the one being staged): its tap and the byte offset of its k range
    for (int t = 0; t < nk; ++t) {
        const int abase = (dw == 0 ? a_base[0] : dw == 1 ? a_base[1] : a_base[2]) + toff;
        // ---- P1 ----
        read_w(t);
        read_a(0, abase);
        stage_w(t + 3, koff3);
        bar();
        mfma16(0);
        bar();
        // ---- P2 ----
        read_a(1, abase);
        if constexpr (UPS) {
            // one pass in flight (its four taps = 16 registers; the kernel sits at 256), requested at an EVEN tap and consumed two
            // k-tiles later: plane 2 of this slice at taps 0 2 4 (written by 6, first read at 9), plane 3 at 6 8 10 (by 12, read at 18),
            // the next slice's plane 0 at 12 14 16 (free since tap 9, written by 18) and plane 1 at 18 20 22 (free since 18, written by
            // 24); H_PASS == 3.  (Consumed ONE k-tile later the loads were still in flight: 2226 -> 3566 us.)
            static_assert(H_PASS == 3, "the fused loader's schedule is laid out for three passes per plane");
            bool issued = false;
            if ((tap & 1) == 0) {
                ups_finish();
                const int q = tap >> 1;  // 0 .. 13
                if (q < 3) {
                    if (cs > 0) ups_issue(2, cs, q), issued = true;
                } else if (q < 6) {
                    if (cs > 0) ups_issue(3, cs, q - 3), issued = true;
                } else if (q < 9) {
                    if (cs + 1 < ncs) ups_issue(0, cs + 1, q - 6), issued = true;
                } else if (q < 12) {
                    if (cs + 1 < ncs) ups_issue(1, cs + 1, q - 9), issued = true;
                }
            }
            // W(t+1) has landed: the two youngest W tiles - and the four tap loads of a pass in flight - may stay in flight
            if (issued) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * W_PASS + 4) : "memory");
            else if (ub.lds >= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * W_PASS + 4) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * W_PASS) : "memory");
        } else {
        if (cs + 1 < ncs) {  // next slice's planes 0 / 1 once the current slice no longer reads them
            if (tap >= 9 && tap < 9 + H_PASS) stage_halo(0, cs + 1, tap - 9);
            if (tap >= 18 && tap < 18 + H_PASS) stage_halo(1, cs + 1, tap - 18);
        }
        if (cs > 0) {        // this slice's planes 2 / 3 (first read at tap 9 / 18)
            if (tap < H_PASS) stage_halo(2, cs, tap);
            else if (tap < 2 * H_PASS) stage_halo(3, cs, tap - H_PASS);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * W_PASS) : "memory");  // W(t+1) (and everything older) has landed
        }
        bar();
        mfma16(1);
        bar();
        // next tap: dw fastest, then dh, then dt, then the next channel slice
        ++tap;
        if (++dw == 3) {
            dw = 0;
            toff += HW * 64;
            if (++dh == 3) {
                dh = 0;
                toff += (PRP - 3 * HW) * 64;
                if (tap == 27) {
                    tap = 0;
                    ++cs;
                    toff = (-PRP - HW) * 64;
                }
            }
        }
        koff3 += tap_bytes;
        if (++tap3 == 27) {  // k range of (tap 0, next slice): back by 27 taps, on by 32 channels
            tap3 = 0;
            koff3 += 64 - 27 * tap_bytes;
        }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();  // pairs with the trailing barrier of the delayed group
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (only zero-chunk dummies are still in flight)

    // ---- epilogue: the lean dense family on the block's rows.  Logical row (inside this block) r = f * 16 + li lives at
    //      output voxel (t0 + f / TH, h0 + f % TH, w0 + li) ----
    const long long origin = (((long long)bb * p.To + t0) * p.Ho + h0) * p.Wo + w0;
    const int m_tile = tile * BM;
    const long long hw_out = (long long)p.Ho * p.Wo;
    const int Wo = p.Wo;
    auto rowmap = [=](int m) -> long long {
        const int r = m - m_tile, f = r >> 4;
        return origin + (long long)(f / TH) * hw_out + (f % TH) * Wo + (r & 15);
    };
    gemm_epilogue_dense_cases<T, TM, TN>(p, acc, m_tile + wr * 128, wc * 64, li, kg, rowmap);
}
