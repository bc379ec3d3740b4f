// Flash-style full spatio-temporal self-attention for the VideoMAE encoder blocks
// (reference modeling_finetune.py:169-190: q*scale @ k^T -> softmax -> @ v, 16 heads x 88 dims,
// 2048 tokens per window).  Never materialises the S x S score matrix.
//
// Inputs come straight from the QKV GEMM epilogue (gemm.hpp, EPI_QKV):
//   qk : [B][S][2][H][DP]  (q | k, head dim zero-padded 88 -> DP = 96)
//   vt : [B][H][DP][S]     (V transposed, so a PV operand fragment is 8 consecutive keys)
// Output: out[B*S][H*Dh] (token-major, head-major columns = transpose(1,2).reshape of the reference).
//
// CDNA4 mapping (32x32 MFMA, one wave = 32 query rows, 4 waves per workgroup):
//  * S^T = K Q^T ("swapped" operands): each lane owns ONE query column, so row max / row sum /
//    rescale are lane-local plus a single lane<->lane+32 exchange.
//  * K rows are read with bits 2/3 of the row index swapped, which makes the 8 scores a lane holds
//    for a k-step 8 CONSECUTIVE keys: P goes from the S accumulator straight into the PV operand
//    (no LDS round trip, no cross-lane shuffle), and the V^T fragment is one ds_read_b128.
//  * O^T = V^T P^T keeps the output accumulator column = query, so the online-softmax rescale is a
//    plain per-lane multiply.
//  * K / V^T tiles are staged global -> registers -> LDS (loads issued before the MFMAs of the
//    current tile, written after them), LDS double-buffered, rows padded by 16 B (odd slot stride)
//    so both ds_read_b128 patterns are conflict-free.
#include "common.hpp"

template <typename T, int DP, int KVB>
__global__ __launch_bounds__(256) void attn_kernel(const T* __restrict__ qk, const T* __restrict__ vt,
                                                   T* __restrict__ out, int S, int H, int Dh, float c_scale) {
    typedef typename Frag<T>::type frag_t;
    constexpr int ES = sizeof(T);
    constexpr int EPC = 16 / ES;
    constexpr int CPF = 8 * ES / 16;
    constexpr int KSTR = DP * ES + 16;   // bytes per K row in LDS
    constexpr int VSTR = KVB * ES + 16;  // bytes per V^T row in LDS
    constexpr int NKS = DP / 16;         // k-steps of the QK^T contraction
    constexpr int NST = KVB / 32;        // 32-key score tiles per KV block
    constexpr int NDT = DP / 32;         // 32-wide output d tiles
    constexpr int KCH = KVB * (DP * ES / 16);  // 16-byte chunks in a K tile
    constexpr int VCH = DP * (KVB * ES / 16);
    constexpr int K_IT = KCH / 256, V_IT = VCH / 256;
    static_assert(KCH % 256 == 0 && VCH % 256 == 0, "tile chunking");
    constexpr int KBYTES = KVB * KSTR, VBYTES = DP * VSTR;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;               // [2][KBYTES]
    char* Vs = smem + 2 * KBYTES;  // [2][VBYTES]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, hi = lane >> 5;
    const int qb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int q_row = qb * 128 + wave * 32 + lq;
    const long long tok_stride = 2LL * H * DP;  // elements between consecutive tokens in qk

    // Q fragments (B operand: column = query, 8 consecutive d per k-step half)
    frag_t qf[NKS];
    {
        const T* qp = qk + ((long long)b * S + q_row) * tok_stride + (long long)h * DP;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            u32x4* d = (u32x4*)&qf[ks];
#pragma unroll
            for (int q = 0; q < CPF; ++q) d[q] = *(const u32x4*)(qp + ks * 16 + hi * 8 + q * EPC);
        }
    }

    const T* kbase = qk + (long long)b * S * tok_stride + (long long)(H + h) * DP;  // + key*tok_stride
    const T* vbase = vt + ((long long)b * H + h) * DP * S;                           // + d*S + key

    u32x4 rk[K_IT], rv[V_IT];
    auto load_kv = [&](int kb) {
#pragma unroll
        for (int i = 0; i < K_IT; ++i) {
            const int id = tid + i * 256;
            const int row = id / (DP * ES / 16), c = id % (DP * ES / 16);
            rk[i] = *(const u32x4*)(kbase + (long long)(kb * KVB + row) * tok_stride + c * EPC);
        }
#pragma unroll
        for (int i = 0; i < V_IT; ++i) {
            const int id = tid + i * 256;
            const int row = id / (KVB * ES / 16), c = id % (KVB * ES / 16);
            rv[i] = *(const u32x4*)(vbase + (long long)row * S + kb * KVB + c * EPC);
        }
    };
    auto store_kv = [&](int buf) {
#pragma unroll
        for (int i = 0; i < K_IT; ++i) {
            const int id = tid + i * 256;
            const int row = id / (DP * ES / 16), c = id % (DP * ES / 16);
            *(u32x4*)(Ks + buf * KBYTES + row * KSTR + c * 16) = rk[i];
        }
#pragma unroll
        for (int i = 0; i < V_IT; ++i) {
            const int id = tid + i * 256;
            const int row = id / (KVB * ES / 16), c = id % (KVB * ES / 16);
            *(u32x4*)(Vs + buf * VBYTES + row * VSTR + c * 16) = rv[i];
        }
    };

    f32x16 o[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // K row read by this lane for score-tile row i = lq: swap bits 2 and 3
    const int krow = (lq & ~12) | ((lq & 4) << 1) | ((lq & 8) >> 1);

    const int nkb = S / KVB;
    load_kv(0);
    store_kv(0);
    __syncthreads();

    for (int kb = 0; kb < nkb; ++kb) {
        const int cur = kb & 1;
        if (kb + 1 < nkb) load_kv(kb + 1);
        const char* Kb = Ks + cur * KBYTES;
        const char* Vb = Vs + cur * VBYTES;

        // ---- S^T tiles: rows = keys (permuted), cols = queries --------------------------------
        f32x16 s[NST];
#pragma unroll
        for (int t = 0; t < NST; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                frag_t kf;
                u32x4* d = (u32x4*)&kf;
#pragma unroll
                for (int q = 0; q < CPF; ++q)
                    d[q] = *(const u32x4*)(Kb + (t * 32 + krow) * KSTR + (ks * 16 + hi * 8) * ES + q * 16);
                s[t] = mma32(kf, qf[ks], s[t]);
            }
        }

        // ---- online softmax (all per-query state is lane-local) -------------------------------
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < NST; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[t][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx * c_scale);
        const float alpha = exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int t = 0; t < NST; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = exp2f(s[t][r] * c_scale - m_new);
                s[t][r] = pv;
                psum += pv;
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;

        // ---- O^T += V^T P^T ---------------------------------------------------------------------
#pragma unroll
        for (int t = 0; t < NST; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                frag_t pf;
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[e] = from_f32<T>(s[t][8 * j + e]);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    frag_t vf;
                    u32x4* d = (u32x4*)&vf;
#pragma unroll
                    for (int q = 0; q < CPF; ++q)
                        d[q] = *(const u32x4*)(Vb + (dt * 32 + lq) * VSTR + (t * 32 + j * 16 + hi * 8) * ES + q * 16);
                    o[dt] = mma32(vf, pf, o[dt]);
                }
            }

        if (kb + 1 < nkb) store_kv(cur ^ 1);
        __syncthreads();
    }

    // ---- normalise and store: lane owns query q_row, d = 32*dt + (r&3) + 8*(r>>2) + 4*hi -------------
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    T* op = out + ((long long)b * S + q_row) * ((long long)H * Dh) + (long long)h * Dh;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = dt * 32 + 8 * g + 4 * hi;
            if (d0 < Dh) {  // Dh % 4 == 0
                if (ES == 2) {
                    bf16x4 v;
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = (bf16_t)(o[dt][4 * g + q] * inv);
                    *(bf16x4*)(op + d0) = v;
                } else {
                    *(f32x4*)(op + d0) = (f32x4){o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv,
                                                 o[dt][4 * g + 3] * inv};
                }
            }
        }
}

template <typename T, int KVB>
static int launch_attn_t(const void* qk, const void* vt, void* out, int B, int S, int H, int Dh, float scale,
                         hipStream_t stream) {
    constexpr int DP = 96;
    constexpr int ES = sizeof(T);
    const size_t lds = 2 * (KVB * (DP * ES + 16) + DP * (KVB * ES + 16));
    auto kern = attn_kernel<T, DP, KVB>;
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const float c_scale = scale * 1.4426950408889634f;
    ProfScope prof(PROF_ATTENTION, stream);
    hipLaunchKernelGGL(kern, dim3(S / 128, H, B), dim3(256), lds, stream, (const T*)qk, (const T*)vt, (T*)out, S, H,
                       Dh, c_scale);
    HIP_TRY(hipGetLastError());
    return 0;
}

// qk: [B][S][2][H][96], vt: [B][H][96][S], out: [B*S][H*Dh]; scale = Dh^-0.5 (reference :150)
int launch_attention(int dtype, const void* qk, const void* vt, void* out, int B, int S, int H, int Dh, float scale,
                     hipStream_t stream) {
    if (S % 128 || Dh > 96 || Dh % 4) {
        l4p_set_error("attention: need S %% 128 == 0 and head_dim <= 96, multiple of 4 (S=%d Dh=%d)", S, Dh);
        return L4P_E_INVALID;
    }
    if (dtype == L4P_BF16) return launch_attn_t<bf16_t, 64>(qk, vt, out, B, S, H, Dh, scale, stream);
    return launch_attn_t<float, 32>(qk, vt, out, B, S, H, Dh, scale, stream);
}
