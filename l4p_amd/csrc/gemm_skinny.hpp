// Dense GEMM for a HANDFUL of rows (M <= 128: the tracker's token-side projections of one rank's query shard - 6 prompt tokens per
// track, 8 tracks per rank of configs[4] -, sam/transformer.py:223-245, mask_decoder.py:160-180) against a weight matrix that comes
// from HBM / the Infinity Cache every time.
//
// Such a launch is pure memory LATENCY: 48 x 1408 activations, 1408 x 1408 weights (4 MB), 0.2 GFLOP.  The LDS-staged 128 x 64 kernel
// (gemm.hpp, four stages) runs it on 22 workgroups, each walking its 22 k-tiles with three in flight: 14 - 16 us per launch, ~40 such
// launches per window of the recursion, none of which shrinks with the query shard.  Here ONE WAVE owns a 16-row x 32-column block of
// the output and streams its operands straight from global memory into MFMA fragment registers (no LDS, no barrier): a ring of RING
// k-steps (3 x 16-byte loads per lane each) is requested before the first MFMA, so a wave keeps 54 KB in flight and the launch
// 130+ waves on as many CUs.
//
// The loads are inline assembly with HAND-COUNTED s_waitcnt vmcnt: loads return in order, so step s may run as soon as the only
// outstanding ones are those requested after its three (hipcc's own counting gives up at the ring's branches and drains the queue
// every step).  NKS > 0: K = 32 NKS is a compile-time constant, everything is unrolled and every count static (the tracker's three
// contraction lengths); NKS = 0: any K % 64 == 0 - whole turns of the ring with static counts, the last turns behind a full drain.
//
// BIT-IDENTICAL to the LDS-staged kernels: the same 16x16x32 MFMA on the same fragments (lane (li, kg) holds elements
// k0 + 8 kg .. + 8 of row li; weight rows in the permuted order 8 (li >> 2) + 4 j + (li & 3) that leaves a lane 8 consecutive output
// columns), accumulated over k in the same ascending order, and the same epilogue functions.  K % 64 == 0 (the staged kernels
// contract whole 64-wide k-tiles; their zero-filled tail is not reproduced here).
#pragma once
#include "gemm.hpp"

#ifndef SKINNY_RING
#define SKINNY_RING 18  // 18 x 3 fragments = 216 VGPRs (22 would spill the ring into the accumulator file); vmcnt(51) at a wait
#endif

// the contraction of one wave's block: xa / wa0 / wa1 = this lane's first fragment of the activation row and of its two weight rows
template <typename T, int NKS>
__device__ __forceinline__ void gemm_skinny_loop(const char* xa, const char* wa0, const char* wa1, const int K, f32x4 (&acc)[1][2]) {
    typedef typename Frag<T>::type frag_t;
    constexpr int RING = SKINNY_RING;
    static_assert((RING - 1) * 3 <= 63, "vmcnt range");
    u32x4 xq[RING], wq0[RING], wq1[RING];

    // the three fragments of the k-step `turn * RING + d` (64 bytes per k-step along a row) into ring slot d
    auto issue = [&xq, &wq0, &wq1](auto d_, const char* x, const char* w0, const char* w1) {
        constexpr int d = decltype(d_)::value;
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(xq[d]) : "v"(x), "n"(d * 64) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(wq0[d]) : "v"(w0), "n"(d * 64) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(wq1[d]) : "v"(w1), "n"(d * 64) : "memory");
    };
    // wait until at most CNT loads are outstanding; the slot's registers pass through the statement, so its consumers stay behind it
    auto landed = [&xq, &wq0, &wq1](auto d_, auto cnt_) {
        constexpr int d = decltype(d_)::value, cnt = decltype(cnt_)::value;
        asm volatile("s_waitcnt vmcnt(%3)" : "+v"(xq[d]), "+v"(wq0[d]), "+v"(wq1[d]) : "n"(cnt));
    };
    auto mfma = [&xq, &wq0, &wq1, &acc](auto d_) {
        constexpr int d = decltype(d_)::value;
        const frag_t x = __builtin_bit_cast(frag_t, xq[d]);
        acc[0][0] = mma16(__builtin_bit_cast(frag_t, wq0[d]), x, acc[0][0]);
        acc[0][1] = mma16(__builtin_bit_cast(frag_t, wq1[d]), x, acc[0][1]);
    };

    if constexpr (NKS > 0) {
        constexpr int PRE = NKS < RING ? NKS : RING;
        static_for_<0, PRE>([&](auto d_) { issue(d_, xa, wa0, wa1); });
        static_for_<0, NKS>([&](auto s_) {
            constexpr int s = decltype(s_)::value, d = s % RING;
            constexpr int after = (NKS - 1 - s) < (RING - 1) ? (NKS - 1 - s) : (RING - 1);  // k-steps requested after this one
            landed(std::integral_constant<int, d>{}, std::integral_constant<int, 3 * after>{});
            mfma(std::integral_constant<int, d>{});
            if constexpr (s + RING < NKS) {
                constexpr long long turn = (s + RING) / RING;
                issue(std::integral_constant<int, d>{}, xa + turn * RING * 64, wa0 + turn * RING * 64, wa1 + turn * RING * 64);
            }
        });
    } else {
        const int nks = K >> 5;
        static_for_<0, RING>([&](auto d_) {
            if (decltype(d_)::value < nks) issue(d_, xa, wa0, wa1);
        });
        int s0 = 0;
        // whole turns whose successors are whole too: every step requests its successor a turn ahead, counts are static
        for (; s0 + 2 * RING <= nks; s0 += RING) {
            const char *xn = xa + (long long)(s0 + RING) * 64, *w0n = wa0 + (long long)(s0 + RING) * 64, *w1n = wa1 + (long long)(s0 + RING) * 64;
            static_for_<0, RING>([&](auto d_) {
                landed(d_, std::integral_constant<int, 3 * (RING - 1)>{});
                mfma(d_);
                issue(d_, xn, w0n, w1n);
            });
        }
        // the last one or two turns: drained in front of each
        for (; s0 < nks; s0 += RING) {
            const char *xn = xa + (long long)(s0 + RING) * 64, *w0n = wa0 + (long long)(s0 + RING) * 64, *w1n = wa1 + (long long)(s0 + RING) * 64;
            static_for_<0, RING>([&](auto d_) { landed(d_, std::integral_constant<int, 0>{}); });
            static_for_<0, RING>([&](auto d_) {
                constexpr int d = decltype(d_)::value;
                if (s0 + d < nks) {
                    mfma(d_);
                    if (s0 + d + RING < nks) issue(d_, xn, w0n, w1n);
                }
            });
        }
    }
}

template <typename T, bool GROUPW = false>
__device__ __forceinline__ void gemm_skinny_body(const GemmParams& p_in, const int wg_index) {
    static_assert(sizeof(T) == 2, "16-bit engines");
    const int lane = threadIdx.x & 63;
    const int li = lane & 15, kg = lane >> 4;
    const int ntn = (p_in.N + 31) >> 5;
    const int mt = wg_index / ntn, nt = wg_index - mt * ntn;
    const int m0 = mt * 16, n0 = nt * 32;
    // row-grouped weights (l4p_gemm_desc.w_gr, a multiple of 16): the block's row group selects the weight matrix, the bias row and
    // (o_gs) the output column block - as the GROUPW instantiation of gemm_body does per tile
    GemmParams patched;
    const GemmParams* pp = &p_in;
    if constexpr (GROUPW) {
        const int grp = m0 / p_in.w_gr;
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

    int m = m0 + li;
    if (m >= p.M) m = p.M - 1;  // (rows past M re-read the last row; the epilogue drops them)
    const long long pm = p.a_gr > 0 ? (long long)(m / p.a_gr) * p.a_gs + p.a_go + (m % p.a_gr) : m;
    const char* xa = (const char*)((const T*)p.A + pm * p.lda + kg * 8);
    // (plain weights are packed with their rows padded to a multiple of 128: rows past N exist and are dropped by the epilogue; a
    //  GROUP's matrix has exactly N rows and the next group's - or nothing - behind them: rows past N re-read row N - 1)
    int wr0 = n0 + 8 * (li >> 2) + (li & 3), wr1 = wr0 + 4;
    if constexpr (GROUPW) {
        wr0 = wr0 < p.N ? wr0 : p.N - 1;
        wr1 = wr1 < p.N ? wr1 : p.N - 1;
    }
    const char* wa0 = (const char*)((const T*)p.W + (long long)wr0 * p.ldw + kg * 8);
    const char* wa1 = (const char*)((const T*)p.W + (long long)wr1 * p.ldw + kg * 8);

    f32x4 acc[1][2];
    acc[0][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc[0][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    switch (p.K) {  // (wave-uniform; the epilogue exists once behind it)
        case 704: gemm_skinny_loop<T, 22>(xa, wa0, wa1, p.K, acc); break;
        case 1408: gemm_skinny_loop<T, 44>(xa, wa0, wa1, p.K, acc); break;
        case 2048: gemm_skinny_loop<T, 64>(xa, wa0, wa1, p.K, acc); break;
        default: gemm_skinny_loop<T, 0>(xa, wa0, wa1, p.K, acc); break;
    }
    if (gemm_epilogue_dense_dispatch<T, 1, 2>(p, acc, m0, n0, li, kg)) return;
    gemm_epilogue<T, 1, 2>(p, acc, m0, n0, li, kg);
}

template <typename T, bool GROUPW = false>
__global__ __launch_bounds__(64) void gemm_skinny_kernel(const GemmParams p) {
    gemm_skinny_body<T, GROUPW>(p, (int)blockIdx.x);
}

// up to L4P_GEMM_GROUP_MAX independent problems as one launch (see gemm_group_kernel): workgroups [first[g], first[g + 1]) run problem g
template <typename T>
__global__ __launch_bounds__(64) void gemm_skinny_group_kernel(const GemmGroupParams g) {
    const int b = (int)blockIdx.x;
    int i = 0;
#pragma unroll
    for (int k = 1; k < L4P_GEMM_GROUP_MAX; ++k)
        if (b >= g.first[k]) i = k;
    // (i is wave-uniform: the descriptor is read through scalar loads at a scalar offset, and the body exists once)
    gemm_skinny_body<T>(g.p[i], b - g.first[i]);
}
