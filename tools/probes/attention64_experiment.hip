// Encoder self-attention, 64 query rows per wave (bf16; reference modeling_finetune.py:169-190).
//
// Same inputs, layouts and arithmetic as attention.hip's hand-scheduled deferred-maximum kernel (q pre-scaled or scaled
// in registers, K tile order, V^T, -m carried in the head-dim padding of Q against 1.0 in K, denominator from the ones row
// of V^T).  What differs is the shape of the work: a workgroup of 4 waves owns 256 query rows, each wave TWO 32-row query
// tiles, and lives ALONE on its CU (one wave per SIMD, the whole 512-entry register file: O^T 96 + two score sets 128 +
// Q 48 + P 32 registers), so that every K / V^T fragment read from LDS feeds two MFMAs instead of one: half the
// ds_read_b128 and half the fragment waits per FLOP of attention.hip's 32-row form (whose ablation put 19 % of its time on
// the fragment reads).  With no second wave on the SIMD to overlap with, the overlap is inside the wave: the vector work
// of a block (exp2 / bf16 packing of P_j, row maximum of S_{j+1}) is sliced behind the 48 MFMAs of the block.
// EXPERIMENT, not part of libl4p_hip.so (round 2; to try it: copy into l4p_amd/csrc/, declare launch_attention64 in attention.hip
// and call it from launch_attention for bf16, S % 256 == 0 and >= 256 workgroups).  Correct (tests/test_kernels_gpu.py's
// attention cases pass through it) and NOT faster than the 32-row kernel: 108.6-109.3 vs 106.5-108.5 us at batch 4, 198.5 vs
// 193.5 us at batch 8 (tools/attn_time.py, random data, same call), for every read-ahead depth from 3 to 14.  Its first form
// used the MFMA builtins: hipcc parked Q in AGPRs and copied it back before every use (56 v_accvgpr_* per 48-MFMA block, 114 us).
// This form has no copies in the steady-state block (48 MFMA, 24 ds_read_b128, 64 v_exp, 32 v_cvt_pk, 38 v_max, nothing else) and
// half the fragment reads per FLOP of the 32-row kernel - so the number of fragment reads is not what bounds either kernel.
// Register plan (one wave per SIMD: 256 architected VGPRs + 256 AGPRs): the output accumulators O^T (96) and the Q fragments
// (48) live in AGPRs for the whole kernel - every MFMA is written as inline asm so that its operand classes are chosen here
// (PV: AGPR accumulator; QK^T: Q read as an AGPR B operand), which hipcc's builtins do not allow (round 2's builtin form of this
// kernel parked Q in AGPRs and copied it back before every use: 56 v_accvgpr_* per 48-MFMA block, 5 % slower than the 32-row
// kernel).  VGPRs hold the two score sets (128), P (32), the fragment read-ahead and addresses.
// Hazards hipcc cannot see through inline asm are kept by construction: a score tile is read by VALU no sooner than two MFMAs
// after its last MFMA (8-pass MFMA: 11 wait states); O^T / Q are touched by VALU only on the rare rescale path and in the
// epilogue, behind explicit s_nop padding that is tied to the registers by asm operands.
#include <cstdlib>
#include <type_traits>

#include "common.hpp"

__device__ __attribute__((aligned(16))) static const unsigned short g64_ones_bf16[8] = {0x3F80, 0x3F80, 0x3F80, 0x3F80,
                                                                                           0x3F80, 0x3F80, 0x3F80, 0x3F80};
__device__ __attribute__((aligned(16))) static const unsigned short g64_kone_bf16[8] = {0x3F80, 0x3F80, 0, 0, 0, 0, 0, 0};

template <int I, int N, class F>
__device__ __forceinline__ void static_for64(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for64<I + 1, N>(f);
    }
}

// MFMAs with explicit operand classes.  S^T tile: VGPR accumulator, K fragment in VGPRs, Q fragment in AGPRs.
__device__ __forceinline__ void mfma_s_first(f32x16& d, const u32x4& k, const u32x4& q) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(d) : "v"(k), "a"(q));
}
__device__ __forceinline__ void mfma_s(f32x16& d, const u32x4& k, const u32x4& q) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(k), "a"(q));
}
// O^T tile: AGPR accumulator, V^T and P fragments in VGPRs
__device__ __forceinline__ void mfma_o(f32x16& d, const u32x4& v, const u32x4& pfrag) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(v), "v"(pfrag));
}

template <int DH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn64_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ kt, const bf16_t* __restrict__ vt, bf16_t* __restrict__ out, int S,
    int H, float c_scale) {
    typedef bf16_t T;
    typedef bf16x8 frag_t;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    constexpr int DP = 96, KVB = 64, QT = 2;
    constexpr int NKS = DP / 16, NST = KVB / 32, NDT = DP / 32;
    constexpr int KBYTES = NKS * KVB * 2 * 16, VBYTES = DP * 128;  // 12 KB each
    constexpr int K_IT = KBYTES / 4096, V_IT = VBYTES / 4096;
    constexpr int DT_L = DH / 32, I_L = DH % 32;  // where the denominator row lands in O^T
    constexpr int HI_L = (I_L >> 2) & 1, R_L = (I_L & 3) + 4 * (I_L >> 3);
    constexpr int KS_P = DH / 16, HI_P = (DH / 8) & 1;  // k-step / lane half whose Q fragment holds dims DH .. DH+7
    static_assert(DH % 8 == 0 && DH < DP, "padding starts at DH");
    constexpr float RESCALE_THR = 8.f;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;              // [2][KBYTES]
    char* Vs = smem + 2 * KBYTES;  // [2][VBYTES]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, hi = lane >> 5;
    // all query blocks of one (batch, head) on the same XCD (workgroup L runs on XCD L % 8): its K / V^T stay in one L2
    const int nqb = S / 256, units = gridDim.x / nqb;
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
    const int q_row0 = qb * 256 + wave * 64 + lq;  // + 32 * qt

    u32x4 qf[QT][NKS];  // (only ever an "a" operand: lives in AGPRs)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const T* qp = q + ((long long)b * S + q_row0 + 32 * qt) * ((long long)H * DP) + (long long)h * DP;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            frag_t f = *(const frag_t*)(qp + ks * 16 + hi * 8);
            if (c_scale != 1.0f) {  // (not pre-scaled: fold scale * log2 e here, a second bf16 rounding of q)
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = (bf16_t)((float)f[e] * c_scale);
            }
            qf[qt][ks] = __builtin_bit_cast(u32x4, f);
        }
    }

    // ---- LDS-DMA sources (as attention.hip) ----
    const int nkb = S / KVB;
    const char* kbase = (const char*)kt + ((long long)(b * H + h) * nkb) * KBYTES + tid * 16;
    const int vrow0 = tid >> 3, vslot = tid & 7;
    const int vchunk = vslot ^ ((vrow0 >> 1) & 7);
    const char* vsrc[V_IT];
    bool vones[V_IT], kpad[K_IT];
#pragma unroll
    for (int i = 0; i < V_IT; ++i) {
        const int d = vrow0 + 32 * i;
        vones[i] = d == DH;
        vsrc[i] = (const char*)(vt + (((long long)b * H + h) * DP + d) * S) + vchunk * 16;
    }
#pragma unroll
    for (int i = 0; i < K_IT; ++i) {
        const int c = tid + 256 * i, key = (c % (2 * KVB)) >> 1;
        kpad[i] = (c / (2 * KVB)) * 16 + (((c & 1) ^ ((key >> 3) & 1)) << 3) == DH;
    }
    const char* ones = (const char*)g64_ones_bf16;
    const char* kone = (const char*)g64_kone_bf16;
    auto issue_k = [&](int kb, int buf) {
#pragma unroll
        for (int i = 0; i < K_IT; ++i) {
            const char* src = kpad[i] ? kone : kbase + (long long)kb * KBYTES + i * 4096;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Ks + buf * KBYTES + (wave * 64 + i * 256) * 16), 16, 0, 0);
        }
    };
    auto issue_v = [&](int kb, int buf) {
#pragma unroll
        for (int i = 0; i < V_IT; ++i) {
            const char* src = vones[i] ? ones : vsrc[i] + (long long)kb * 128;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Vs + buf * VBYTES + (wave * 64 + i * 256) * 16), 16, 0, 0);
        }
    };

    f32x16 o[QT][NDT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qt][dt][r] = 0.f;
    float m_run[QT] = {0.f, 0.f}, mx[QT];

    const int krow = (lq & ~12) | ((lq & 4) << 1) | ((lq & 8) >> 1);
    int koff[NST];
#pragma unroll
    for (int t = 0; t < NST; ++t) {
        const int key = t * 32 + krow;
        koff[t] = (key * 2 + (hi ^ ((key >> 3) & 1))) * 16;
    }
    const int vsw = (lq >> 1) & 7;

    const int nit = nkb;
    issue_k(0, 0);
    issue_v(0, 0);
    if (nit > 1) issue_k(1, 1);
    __syncthreads();
    f32x16 s_a[QT][NST], s_b[QT][NST];
    // block 0 scores (compiler scheduled: once per workgroup)
#pragma unroll
    for (int t = 0; t < NST; ++t) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const u32x4 kf = *(const u32x4*)(Ks + ks * (KVB * 2 * 16) + koff[t]);
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                if (ks == 0)
                    mfma_s_first(s_a[qt][t], kf, qf[qt][ks]);
                else
                    mfma_s(s_a[qt][t], kf, qf[qt][ks]);
            }
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(s_a[0][0]), "+v"(s_a[0][1]), "+v"(s_a[1][0]), "+v"(s_a[1][1]));  // MFMA -> VALU read
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float m = s_a[qt][0][0];
#pragma unroll
        for (int t = 0; t < NST; ++t)
#pragma unroll
            for (int r = (t == 0 ? 1 : 0); r < 16; ++r) m = fmaxf(m, s_a[qt][t][r]);
        mx[qt] = fmaxf(m, __shfl_xor(m, 32));
    }
    __syncthreads();  // every wave has read K_0 before the first iteration re-stages its slot

    auto step = [&](int it, auto has_next, f32x16 (*s_cur)[NST], f32x16 (*s_nxt)[NST]) __attribute__((always_inline)) {
        constexpr bool HAS_NEXT = decltype(has_next)::value;
        const int cur = it & 1;
        if (it + 2 < nit) issue_k(it + 2, cur);      // K_{it} (ring slot cur) was consumed last iteration
        if (HAS_NEXT) issue_v(it + 1, cur ^ 1);      // V_{it-1} (slot cur^1) was consumed last iteration
        // ---- deferred maximum: the rare side path (see attention.hip) ----
        if (it == 0 || __any(fmaxf(mx[0], mx[1]) > RESCALE_THR)) {
            // (the previous block's last PV MFMAs may still be writing O^T: 8 passes + 3 wait states before VALU reads it)
            asm volatile("s_nop 7\n\ts_nop 7"
                         : "+a"(o[0][0]), "+a"(o[0][1]), "+a"(o[0][2]), "+a"(o[1][0]), "+a"(o[1][1]), "+a"(o[1][2]));
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const float want = m_run[qt] + (it == 0 ? mx[qt] : fmaxf(mx[qt], 0.f));
                const T m_hi = (T)want, m_lo = (T)(want - (float)m_hi);
                const float m_new = (float)m_hi + (float)m_lo;
                const float delta = m_new - m_run[qt];
                if (it != 0) {
                    const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[qt][dt][r] *= alpha;
                }
                m_run[qt] = m_new;
                if (hi == HI_P) {  // dims DH, DH+1 = the two bf16 halves of -m: dword 0 of the chunk
                    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
                    const bf16x2 mm = {-m_hi, -m_lo};
                    qf[qt][KS_P][0] = __builtin_bit_cast(unsigned, mm);
                }
#pragma unroll
                for (int t = 0; t < NST; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s_cur[qt][t][r] -= delta;
            }
            // (VALU wrote O^T / Q / the scores: keep the first MFMAs that read them a few wait states away)
            asm volatile("s_nop 4"
                         : "+a"(o[0][0]), "+a"(o[0][1]), "+a"(o[0][2]), "+a"(o[1][0]), "+a"(o[1][1]), "+a"(o[1][2]), "+a"(qf[0][KS_P]),
                           "+a"(qf[1][KS_P]));
        }
        // ---- hand-scheduled block body: 24 fragment reads (12 K, 12 V^T), each feeding QT MFMAs ----
#ifndef ATTN64_PRE
#define ATTN64_PRE 6
#endif
        constexpr int PRE = ATTN64_PRE;  // fragment reads in flight ahead of the MFMA pair that consumes them
        constexpr int NR = HAS_NEXT ? 24 : 12, R0 = HAS_NEXT ? 0 : 12;
        const unsigned lds_k = (unsigned)(size_t)(__attribute__((address_space(3))) char*)Ks;
        const unsigned lds_v = (unsigned)(size_t)(__attribute__((address_space(3))) char*)Vs;
        const unsigned kb = lds_k + (cur ^ 1) * KBYTES, vb = lds_v + cur * VBYTES;
        unsigned ka[NST], va[NST * 2];
#pragma unroll
        for (int t = 0; t < NST; ++t) ka[t] = kb + koff[t];
#pragma unroll
        for (int tj = 0; tj < NST * 2; ++tj) va[tj] = vb + lq * 128 + ((((tj * 16 + hi * 8) >> 3) ^ vsw) << 4);
        u32x4 fr[24];
        auto rd = [&fr, &ka, &va](auto i_) {
            constexpr int r = R0 + decltype(i_)::value;
            if constexpr (r < 12)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[r]) : "v"(ka[r % NST]), "n"((r / NST) * (KVB * 2 * 16)) : "memory");
            else
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[r]) : "v"(va[(r - 12) / NDT]), "n"(((r - 12) % NDT) * 4096) : "memory");
        };
        frag_t pf[QT][NST][2];
        auto exp_pair = [&](auto qt_, auto p_) {  // P elements 2p, 2p+1 of q-tile qt
            constexpr int qt = decltype(qt_)::value, e = decltype(p_)::value * 2, t = e / 16, r = e % 16, j = r / 8, ee = r % 8;
            pf[qt][t][j][ee] = (bf16_t)__builtin_amdgcn_exp2f(s_cur[qt][t][r]);
            pf[qt][t][j][ee + 1] = (bf16_t)__builtin_amdgcn_exp2f(s_cur[qt][t][r + 1]);
        };
        float mxn[QT] = {-INFINITY, -INFINITY};
        auto max_pair = [&](auto qt_, auto p_) {
            constexpr int qt = decltype(qt_)::value, e = decltype(p_)::value * 2, t = e / 16, r = e % 16;
            mxn[qt] = fmaxf(fmaxf(mxn[qt], s_nxt[qt][t][r]), s_nxt[qt][t][r + 1]);
        };
        static_for64<0, PRE>(rd);
        if constexpr (!HAS_NEXT) {
            static_for64<0, 16>([&](auto p_) {
                exp_pair(std::integral_constant<int, 0>{}, p_);
                exp_pair(std::integral_constant<int, 1>{}, p_);
            });
        }
        static_for64<0, NR>([&](auto m_) {
            constexpr int m = decltype(m_)::value, r = R0 + m;
            constexpr int issued = (PRE + m < NR) ? PRE + m : NR;
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(issued - m - 1) : "memory");
            constexpr int slot = r < 12 ? r : r - 12;
            constexpr int p0 = slot < 4 ? 2 * slot : 4 + slot, np = slot < 4 ? 2 : 1;
            static_for64<0, QT>([&](auto qt_) {
                constexpr int qt = decltype(qt_)::value;
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (r < NST) {
                    mfma_s_first(s_nxt[qt][r % NST], fr[r], qf[qt][r / NST]);
                } else if constexpr (r < 12) {
                    mfma_s(s_nxt[qt][r % NST], fr[r], qf[qt][r / NST]);
                } else {
                    constexpr int i = r - 12;
                    mfma_o(o[qt][i % NDT], fr[r], __builtin_bit_cast(u32x4, pf[qt][(i / NDT) >> 1][(i / NDT) & 1]));
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (qt == 0 && PRE + m < NR) rd(std::integral_constant<int, PRE + m>{});
                // the vector slice riding behind this MFMA: 16 pairs per q-tile over 12 slots (2 in the first four, then 1)
                if constexpr (r < 12) {
                    static_for64<p0, p0 + np>([&](auto p_) { exp_pair(qt_, p_); });
                } else if constexpr (HAS_NEXT) {
                    static_for64<p0, p0 + np>([&](auto p_) { max_pair(qt_, p_); });
                }
            });
        });
        if (HAS_NEXT) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mxn[qt]), __float_as_uint(mxn[qt]), false, false);
                mx[qt] = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            }
            __syncthreads();
        }
    };
    {
        int it = 0;
        for (; it + 2 < nit; it += 2) {
            step(it, std::true_type{}, s_a, s_b);
            step(it + 1, std::true_type{}, s_b, s_a);
        }
        if (it + 2 == nit) {
            step(it, std::true_type{}, s_a, s_b);
            step(it + 1, std::false_type{}, s_b, s_a);
        } else {
            step(it, std::false_type{}, s_a, s_b);
        }
    }

    // ---- normalise and store: lane owns query q_row0 + 32 qt, d = 32*dt + (r&3) + 8*(r>>2) + 4*hi ----
    asm volatile("s_nop 7\n\ts_nop 7" : "+a"(o[0][0]), "+a"(o[0][1]), "+a"(o[0][2]), "+a"(o[1][0]), "+a"(o[1][1]), "+a"(o[1][2]));
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float l_tot = o[qt][DT_L][R_L];
        {
            const float other = __shfl_xor(l_tot, 32);
            if (hi != HI_L) l_tot = other;
        }
        const float inv = 1.0f / l_tot;
        T* op = out + ((long long)b * S + q_row0 + 32 * qt) * ((long long)H * DH) + (long long)h * DH;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = dt * 32 + 8 * g + 4 * hi;
                if (d0 < DH) {
                    bf16x4 v;
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = (bf16_t)(o[qt][dt][4 * g + k] * inv);
                    *(bf16x4*)(op + d0) = v;
                }
            }
    }
}

template <int DH>
static int launch_attn64_t(const void* q, const void* kt, const void* vt, void* out, int B, int S, int H, float scale, hipStream_t stream) {
    const size_t lds = 2 * (size_t)(6 * 64 * 2 * 16 + 96 * 128);
    auto kern = attn64_kernel<DH>;
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const float c_scale = scale > 0.f ? scale * 1.4426950408889634f : 1.0f;
    ProfScope prof(PROF_ATTENTION, stream);
    hipLaunchKernelGGL(kern, dim3((S / 256) * H * B), dim3(256), lds, stream, (const bf16_t*)q, (const bf16_t*)kt, (const bf16_t*)vt,
                       (bf16_t*)out, S, H, c_scale);
    HIP_TRY(hipGetLastError());
    return 0;
}

// bf16, S % 256 == 0, head_dim 88 / 64; the caller (attention.hip:launch_attention) decides when this form is used
int launch_attention64(const void* q, const void* kt, const void* vt, void* out, int B, int S, int H, int Dh, float scale,
                       hipStream_t stream) {
    if (Dh == 88) return launch_attn64_t<88>(q, kt, vt, out, B, S, H, scale, stream);
    return launch_attn64_t<64>(q, kt, vt, out, B, S, H, scale, stream);
}
