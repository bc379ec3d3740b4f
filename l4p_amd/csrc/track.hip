// Kernels specific to the SAM-style point tracker (reference sparse_heads.py, sam/*.py).
// The heavy projections run in gemm.hpp; what lives here is the small-token attention in its three
// shapes, prompt-token construction, the per-query key initialisation, the hyper-network mask
// product, the fused (trilinear up-sampling + soft-argmax + spatial means) read-out that never
// materialises the [N,3,16,224,224] logits, and the integer/boolean sliding-window bookkeeping.
#include "common.hpp"

template <typename T> struct Vec8;  // 8 consecutive elements of T as floats
template <> struct Vec8<bf16_t> {
    static __device__ __forceinline__ void load(const bf16_t* p, float* v) {
        const bf16x8 t = *(const bf16x8*)p;
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (float)t[k];
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float* v) {
        bf16x8 t;
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = (bf16_t)v[k];
        *(bf16x8*)p = t;
    }
};
template <> struct Vec8<f16_t> {
    static __device__ __forceinline__ void load(const f16_t* p, float* v) {
        const f16x8 t = *(const f16x8*)p;
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (float)t[k];
    }
    static __device__ __forceinline__ void store(f16_t* p, const float* v) {
        f16x8 t;
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = (f16_t)v[k];
        *(f16x8*)p = t;
    }
};
template <> struct Vec8<float> {
    static __device__ __forceinline__ void load(const float* p, float* v) {
        const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = a[k];
            v[4 + k] = b[k];
        }
    }
    static __device__ __forceinline__ void store(float* p, const float* v) {
        *(f32x4*)p = (f32x4){v[0], v[1], v[2], v[3]};
        *(f32x4*)(p + 4) = (f32x4){v[4], v[5], v[6], v[7]};
    }
};

template <typename T> struct Vec4;  // 4 consecutive elements (head dims only need % 4 == 0)
template <> struct Vec4<bf16_t> {
    static __device__ __forceinline__ void load(const bf16_t* p, float* v) {
        const bf16x4 t = *(const bf16x4*)p;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (float)t[k];
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float* v) {
        bf16x4 t;
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = (bf16_t)v[k];
        *(bf16x4*)p = t;
    }
};
template <> struct Vec4<f16_t> {
    static __device__ __forceinline__ void load(const f16_t* p, float* v) {
        const f16x4 t = *(const f16x4*)p;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (float)t[k];
    }
    static __device__ __forceinline__ void store(f16_t* p, const float* v) {
        f16x4 t;
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = (f16_t)v[k];
        *(f16x4*)p = t;
    }
};
template <> struct Vec4<float> {
    static __device__ __forceinline__ void load(const float* p, float* v) {
        const f32x4 a = *(const f32x4*)p;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = a[k];
    }
    static __device__ __forceinline__ void store(float* p, const float* v) { *(f32x4*)p = (f32x4){v[0], v[1], v[2], v[3]}; }
};

// -------------------------------------------------------------------------------------------------
// Prompt tokens (prompt_encoder.py:78-121,196-203,221-232; mask_decoder.py:107-113):
// tokens[n] = [mask_tok0, mask_tok1, mask_tok2, point PE + label emb, not-a-point, feature + feature emb]
// queries [N][3] = (t, x, y) in window time / pixels; labels [N] float {0,1,2}; pfeat [N][C]; plabel [N].
// -------------------------------------------------------------------------------------------------
// TT: the engine type of the optional second output (tokens_T: the same values rounded once - what l4p_cast makes of `tokens`; the
// window call takes both from one launch)
template <typename TT>
__global__ void track_tokens_kernel(const float* __restrict__ queries, const float* __restrict__ labels,
                                    const float* __restrict__ pfeat, const float* __restrict__ plabel,
                                    const float* __restrict__ gauss, const float* __restrict__ mask_tokens,
                                    const float* __restrict__ point_emb0, const float* __restrict__ point_emb1,
                                    const float* __restrict__ not_a_point, const float* __restrict__ feat_emb0,
                                    const float* __restrict__ feat_emb1, float* __restrict__ tokens, int N, int C, float T,
                                    float H, float W, TT* __restrict__ tokens_T) {
    const int n = blockIdx.x;
    const int half = C / 2;
    const float ct = 2.f * (queries[n * 3 + 0] / T) - 1.f;
    const float cx = 2.f * (queries[n * 3 + 1] / W) - 1.f;
    const float cy = 2.f * (queries[n * 3 + 2] / H) - 1.f;
    const float lab = labels[n], pl = plabel[n];
    float* out = tokens + (long long)n * 6 * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        out[0 * C + c] = mask_tokens[0 * C + c];
        out[1 * C + c] = mask_tokens[1 * C + c];
        out[2 * C + c] = mask_tokens[2 * C + c];
        const int j = c < half ? c : c - half;
        float a = ct * gauss[j];
        a += cx * gauss[half + j];
        a += cy * gauss[2 * half + j];
        a = 6.283185307179586f * a;
        float pe = c < half ? sinf(a) : cosf(a);
        if (lab == 0.f) pe += point_emb0[c];
        if (lab == 1.f) pe += point_emb1[c];
        out[3 * C + c] = pe;
        out[4 * C + c] = not_a_point[c];
        float fe = 0.f;
        if (pl == 0.f) fe = pfeat[(long long)n * C + c] + feat_emb0[c];
        if (pl == 1.f) fe = pfeat[(long long)n * C + c] + feat_emb1[c];
        out[5 * C + c] = fe;
        if (tokens_T) {
            TT* ot = tokens_T + (long long)n * 6 * C;
            ot[0 * C + c] = from_f32<TT>(mask_tokens[0 * C + c]);
            ot[1 * C + c] = from_f32<TT>(mask_tokens[1 * C + c]);
            ot[2 * C + c] = from_f32<TT>(mask_tokens[2 * C + c]);
            ot[3 * C + c] = from_f32<TT>(pe);
            ot[4 * C + c] = from_f32<TT>(not_a_point[c]);
            ot[5 * C + c] = from_f32<TT>(fe);
        }
    }
}

// -------------------------------------------------------------------------------------------------
// keys = enc_last (broadcast over queries) + history   (sparse_heads.py:341-346), emitted as
// float (residual master), T (values) and T(keys + dense PE) (keys of the attention).
// -------------------------------------------------------------------------------------------------
// shared8 < per_q8 (later windows, hist_uniform == 2): rows from shared_from on are the same for every track (encoder feature +
// the learned mask token), so only track 0 forms them - its float rows also go to k32_shared [P - shared_from][C], the residual
// master l4p_layernorm_res reads for those rows of EVERY track (it normalises the key stream in place, so the shared rows cannot
// live inside it).  The other tracks' rows past shared_from are neither read (hist) nor written.
template <typename T>
__global__ void track_keys_init_kernel(const float* __restrict__ enc, const float* __restrict__ hist,
                                       const float* __restrict__ pos, float* __restrict__ k32, T* __restrict__ kT,
                                       T* __restrict__ kP, long long per_q8, long long total8, long long shared8,
                                       float* __restrict__ k32_shared) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total8; i += (long long)gridDim.x * blockDim.x) {
        const long long e8 = i % per_q8, e = e8 * 8;
        if (e8 >= shared8 && i >= per_q8) continue;
        float a[8], h[8], p[8], s[8];
        Vec8<float>::load(enc + e, a);
        Vec8<float>::load(hist + i * 8, h);
        Vec8<float>::load(pos + e, p);
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] += h[k];
        Vec8<float>::store(k32 + i * 8, a);
        if (k32_shared && e8 >= shared8) Vec8<float>::store(k32_shared + (e8 - shared8) * 8, a);
        Vec8<T>::store(kT + i * 8, a);
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] = a[k] + p[k];
        Vec8<T>::store(kP + i * 8, s);
    }
}

// out[r][:] = v[:] for r in [0, rows): broadcast fill (history mask token, sparse_heads.py:418-427)
__global__ void fill_rows_kernel(float* __restrict__ out, const float* __restrict__ v, long long rows, int C,
                                 long long group_rows, long long group_stride, long long group_off) {
    const long long total = rows * (C / 4);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / (C / 4);
        const int c4 = (int)(i % (C / 4));
        const long long row = (r / group_rows) * group_stride + group_off + (r % group_rows);
        ((f32x4*)(out + row * C))[c4] = ((const f32x4*)v)[c4];
    }
}

// -------------------------------------------------------------------------------------------------
// Attention among the 6 prompt tokens (sam/transformer.py:223-245, self_attn): q,k,v [N][6][D], heads x hd.
// One wave per (query n, head); lanes split the head dim.
// -------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void self_attn6_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                        const T* __restrict__ v, T* __restrict__ out, int D, int hd,
                                                        float scale) {
    const int n = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    const long long base = (long long)n * 6 * D + (long long)h * hd;
    float s[6][6];
    if (hd <= 192) {
        // Head dims of up to three 64-lane chunks (the tracker: 176): all 54 operand elements of a lane are requested before the first
        // product - the rolled form below re-read q / k per (i, j) pair, 36 x 2 dependent round trips per launch (27 us for 0.1 MFLOP).
        // Same products, same order, same sums.
        float qv[6][3], kv[6][3], vv[6][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int d = lane + 64 * c;
            const bool ok = d < hd;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                qv[i][c] = ok ? (float)q[base + (long long)i * D + d] : 0.f;
                kv[i][c] = ok ? (float)k[base + (long long)i * D + d] : 0.f;
                vv[i][c] = ok ? (float)v[base + (long long)i * D + d] : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                float acc = 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (lane + 64 * c < hd) acc += qv[i][c] * kv[j][c];
                s[i][j] = wave_sum(acc) * scale;
            }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            float m = s[i][0];
#pragma unroll
            for (int j = 1; j < 6; ++j) m = fmaxf(m, s[i][j]);
            float z = 0.f;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                s[i][j] = expf(s[i][j] - m);
                z += s[i][j];
            }
            const float iz = 1.f / z;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int d = lane + 64 * c;
                if (d < hd) {
                    float o = 0.f;
#pragma unroll
                    for (int j = 0; j < 6; ++j) o += s[i][j] * iz * vv[j][c];
                    out[base + (long long)i * D + d] = from_f32<T>(o);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float acc = 0.f;
            for (int d = lane; d < hd; d += 64) acc += (float)q[base + (long long)i * D + d] * (float)k[base + (long long)j * D + d];
            s[i][j] = wave_sum(acc) * scale;
        }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float m = s[i][0];
#pragma unroll
        for (int j = 1; j < 6; ++j) m = fmaxf(m, s[i][j]);
        float z = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            s[i][j] = expf(s[i][j] - m);
            z += s[i][j];
        }
        const float iz = 1.f / z;
        for (int d = lane; d < hd; d += 64) {
            float o = 0.f;
#pragma unroll
            for (int j = 0; j < 6; ++j) o += s[i][j] * iz * (float)v[base + (long long)j * D + d];
            out[base + (long long)i * D + d] = from_f32<T>(o);
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Prompt tokens attending to the P image tokens (cross_attn_token_to_image / final attention):
// q [N][6][D], k, v [N][P][D] -> out [N][6][D]; D = heads*hd, hd % 8 == 0, hd <= 96.
// One workgroup per (n, head): scores for all 6 x P pairs live in LDS, softmax per token, then P.V with
// the head dim split in 8-wide column groups and the keys split across thread groups.
// -------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void t2i_attn_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                       const T* __restrict__ v, T* __restrict__ out, int P, int D, int hd,
                                                       float scale, long long kv_stride, const float* __restrict__ pre = nullptr,
                                                       long long ld_pre = 0, int heads = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sc = (float*)smem;      // [6][P]
    float* qs = sc + 6 * P;        // [6][hd]
    float* red = qs + 6 * 96;      // [256] scratch, later [ngroups][6][hd]
    const int n = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
    const T* qp = q + (long long)n * 6 * D + (long long)h * hd;
    const T* kp = k + (long long)n * kv_stride + (long long)h * hd;  // kv_stride = P*D, or 0 when every query shares K / V
    const T* vp = v + (long long)n * kv_stride + (long long)h * hd;
    // pre: the scaled scores were formed elsewhere (folded keys: l4p_t2i_attn_scores) - pre[(n * P + p) * ld_pre + i * heads + h]
    if (pre) {
        const float* pr = pre + (long long)n * P * ld_pre + h;
        for (int p = tid; p < P; p += 256) {
            float sv[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) sv[i] = pr[(long long)p * ld_pre + i * heads];
#pragma unroll
            for (int i = 0; i < 6; ++i) sc[i * P + p] = sv[i];
        }
    } else
        for (int i = tid; i < 6 * hd; i += 256) qs[(i / hd) * 96 + (i % hd)] = (float)qp[(long long)(i / hd) * D + (i % hd)];
    __syncthreads();
    // scores
    // bf16, head dims in whole 16-byte chunks: a thread requests its WHOLE key row (hd / 8 chunks of 16 bytes) at once and the next
    // row before it uses the current one - the phase is bound by memory latency (a thread reads its rows alone), so what counts
    // is bytes in flight per thread: 2 x 176 here against 32 in the batched form below.  Same products in the same order.
    constexpr int MAXCH = 12;  // (174 VGPRs: two workgroups per CU instead of the three LDS would allow - measured equal to a
                               //  168-register build with three; the phase is VALU-bound past this point: 112.7 vs 122 us at 64 tracks)
    const bool whole_rows = sizeof(T) == 2 && (hd & 7) == 0 && (hd >> 3) <= MAXCH;
    if (whole_rows && !pre) {
        const int nch = hd >> 3;
        vec8h<T> cur[MAXCH], nxt[MAXCH];
        auto load_row = [&](int p, vec8h<T> (&r)[MAXCH]) {
            const vec8h<T>* kr = (const vec8h<T>*)((const vec4e<T>*)kp + (long long)p * D);
#pragma unroll
            for (int c = 0; c < MAXCH; ++c)
                if (c < nch) r[c] = kr[c];
        };
        if (tid < P) load_row(tid, cur);
        for (int p = tid; p < P; p += 256) {
            if (p + 256 < P) load_row(p + 256, nxt);
            float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < MAXCH; ++c) {
                if (c < nch) {
                    float kv[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) kv[e] = (float)cur[c][e];
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        const f32x4 q0 = *(const f32x4*)(qs + i * 96 + 8 * c), q1 = *(const f32x4*)(qs + i * 96 + 8 * c + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i] += q0[e] * kv[e];
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i] += q1[e] * kv[4 + e];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) sc[i * P + p] = acc[i] * scale;
#pragma unroll
            for (int c = 0; c < MAXCH; ++c) cur[c] = nxt[c];
        }
    }
    for (int p = tid; p < P && !whole_rows && !pre; p += 256) {
        float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const T* kr = kp + (long long)p * D;
        // the key row is requested in batches of 4 loads before any is used (a thread reads its row alone: without the
        // batching every 8-byte load was a separate exposed round trip)
        int d0 = 0;
        for (; d0 + 16 <= hd; d0 += 16) {
            float kv[16];
#pragma unroll
            for (int c = 0; c < 4; ++c) Vec4<T>::load(kr + d0 + 4 * c, kv + 4 * c);
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f32x4 q4 = *(const f32x4*)(qs + i * 96 + d0 + 4 * c);  // b128 LDS read (rows are 384 B apart)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i] += q4[e] * kv[4 * c + e];
                }
        }
        for (; d0 < hd; d0 += 4) {
            float kv[4];
            Vec4<T>::load(kr + d0, kv);
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i] += qs[i * 96 + d0 + e] * kv[e];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) sc[i * P + p] = acc[i] * scale;
    }
    __syncthreads();
    // softmax over P for each of the 6 tokens (in place: sc <- exp(s - max); z kept in registers of all threads)
    float zinv[6];
    for (int i = 0; i < 6; ++i) {
        float m = -INFINITY;
        for (int p = tid; p < P; p += 256) m = fmaxf(m, sc[i * P + p]);
        m = wave_max(m);
        if ((tid & 63) == 0) red[tid >> 6] = m;
        __syncthreads();
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        __syncthreads();
        float z = 0.f;
        for (int p = tid; p < P; p += 256) {
            const float e = expf(sc[i * P + p] - m);
            sc[i * P + p] = e;
            z += e;
        }
        z = wave_sum(z);
        if ((tid & 63) == 0) red[tid >> 6] = z;
        __syncthreads();
        zinv[i] = 1.f / (red[0] + red[1] + red[2] + red[3]);
        __syncthreads();
    }
    // P.V: column group cg (4 dims) x key group kg
    const int ncg = hd / 4;
    const int nkg = 256 / ncg;
    const int cg = tid % ncg, kg = tid / ncg;
    float o[6][4];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) o[i][e] = 0.f;
    if (kg < nkg) {
        int p = kg;
        for (; p + 7 * nkg < P; p += 8 * nkg) {  // eight value rows in flight (same summation order as the plain loop)
            float vv[8][4];
#pragma unroll
            for (int u = 0; u < 8; ++u) Vec4<T>::load(vp + (long long)(p + u * nkg) * D + cg * 4, vv[u]);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const float w = sc[i * P + p + u * nkg];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[i][e] += w * vv[u][e];
                }
        }
        for (; p < P; p += nkg) {
            float vv[4];
            Vec4<T>::load(vp + (long long)p * D + cg * 4, vv);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const float w = sc[i * P + p];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[i][e] += w * vv[e];
            }
        }
    }
    __syncthreads();
    // reduce across key groups through LDS (reuse sc: scores are no longer needed)
    float* acc = sc;  // [nkg][6][hd]
    if (kg < nkg) {
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[(kg * 6 + i) * hd + cg * 4 + e] = o[i][e];
    }
    __syncthreads();
    for (int idx = tid; idx < 6 * hd; idx += 256) {
        const int i = idx / hd, d = idx % hd;
        float s = 0.f;
        for (int g = 0; g < nkg; ++g) s += acc[(g * 6 + i) * hd + d];
        out[(long long)n * 6 * D + (long long)i * D + (long long)h * hd + d] = from_f32<T>(s * zinv[i]);
    }
}

// -------------------------------------------------------------------------------------------------
// Image tokens attending to the 6 prompt tokens (cross_attn_image_to_token):
// q [N][P][D], k, v [N][6][D] -> out [N][P][D].  One thread per (image token, head).
// -------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void i2t_attn_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                       const T* __restrict__ v, T* __restrict__ out, int P, int D, int hd,
                                                       int heads, float scale, long long q_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ks = (float*)smem;  // [6][D]
    float* vs = ks + 6 * D;    // [6][D]
    const int n = blockIdx.y, tid = threadIdx.x;
    for (int i = tid; i < 6 * D; i += 256) {
        ks[i] = (float)k[(long long)n * 6 * D + i];
        vs[i] = (float)v[(long long)n * 6 * D + i];
    }
    __syncthreads();
    const int work = blockIdx.x * 256 + tid;  // (p, h)
    if (work >= P * heads) return;
    const int p = work / heads, h = work % heads;
    const T* qp = q + (long long)n * q_stride + (long long)p * D + (long long)h * hd;  // q_stride = P*D, or 0 (shared queries)
    float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int d0 = 0; d0 < hd; d0 += 4) {
        float qv[4];
        Vec4<T>::load(qp + d0, qv);
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) s[i] += qv[e] * ks[i * D + h * hd + d0 + e];
    }
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        s[i] *= scale;
        m = fmaxf(m, s[i]);
    }
    float z = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        s[i] = expf(s[i] - m);
        z += s[i];
    }
    const float iz = 1.f / z;
    T* op = out + ((long long)n * P + p) * D + (long long)h * hd;
    for (int d0 = 0; d0 < hd; d0 += 4) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < 6; ++i) a += s[i] * vs[i * D + h * hd + d0 + e];
            o[e] = a * iz;
        }
        Vec4<T>::store(op + d0, o);
    }
}

// LDS-staged form of i2t_attn_kernel (same arithmetic, same order): a workgroup takes ROWS = 256 / heads consecutive image
// tokens; their q rows (ROWS x D, one contiguous span) arrive by coalesced 16-byte LDS-DMA, each thread (token, head) works
// on its head segment in LDS and overwrites it with its output, and the tile leaves with coalesced 16-byte stores.  The
// direct form reads and writes 8 bytes per lane at a head-dim stride (1.8 TB/s measured).  Needs D * sizeof(T) % 16 == 0.
template <typename T>
__global__ __launch_bounds__(256) void i2t_attn_lds_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                           const T* __restrict__ v, T* __restrict__ out, int P, int D, int hd,
                                                           int heads, float scale, long long q_stride) {
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ks = (float*)smem;  // [6][D]
    float* vs = ks + 6 * D;    // [6][D]
    T* tile = (T*)(smem + ((12 * D * 4 + 15) & ~15));  // [ROWS][D]
    const int n = blockIdx.y, tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rows = 256 / heads, p0 = blockIdx.x * rows;
    const int nrow = P - p0 < rows ? P - p0 : rows;
    const int chunks = nrow * D * (int)sizeof(T) / 16;
    const char* src = (const char*)(q + (long long)n * q_stride + (long long)p0 * D);
    for (int c = 0; c < chunks; c += 256)
        if (c + tid < chunks)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + (long long)(c + tid) * 16), (lptr_t)((char*)tile + (c + wave * 64) * 16), 16, 0, 0);
    for (int i = tid * 4; i < 6 * D; i += 1024) {  // D % 4 == 0: 4 elements per load (was one 2-byte load per element)
        Vec4<T>::load(k + (long long)n * 6 * D + i, ks + i);
        Vec4<T>::load(v + (long long)n * 6 * D + i, vs + i);
    }
    __syncthreads();
    const int pl = tid / heads, h = tid % heads;
    if (pl < nrow) {
        T* qp = tile + (long long)pl * D + (long long)h * hd;
        float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float* kh = ks + h * hd;  // (16-byte aligned: D % 4 == 0 and hd % 4 == 0, so the token rows are read as b128)
        const float* vh = vs + h * hd;
        for (int d0 = 0; d0 < hd; d0 += 4) {
            float qv[4];
            Vec4<T>::load(qp + d0, qv);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const f32x4 k4 = *(const f32x4*)(kh + i * D + d0);
#pragma unroll
                for (int e = 0; e < 4; ++e) s[i] += qv[e] * k4[e];
            }
        }
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            s[i] *= scale;
            m = fmaxf(m, s[i]);
        }
        float z = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            s[i] = expf(s[i] - m);
            z += s[i];
        }
        const float iz = 1.f / z;
        for (int d0 = 0; d0 < hd; d0 += 4) {
            float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const f32x4 v4 = *(const f32x4*)(vh + i * D + d0);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] += s[i] * v4[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] *= iz;
            Vec4<T>::store(qp + d0, o);  // in place: this thread is the only reader of its segment
        }
    }
    __syncthreads();
    char* dst = (char*)(out + ((long long)n * P + p0) * D);
    for (int c = tid; c < chunks; c += 256) *(u32x4*)(dst + (long long)c * 16) = *(const u32x4*)((const char*)tile + c * 16);
}

// -------------------------------------------------------------------------------------------------
// masks[n][m][vox] = sum_c hyper[n][m][c] * up[n][vox][c]   (mask_decoder.py:139), up channels-last T.
// -------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void mask_product_kernel(const T* __restrict__ up, const float* __restrict__ hyper,
                                                           float* __restrict__ masks, long long vox, int Cc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* hs = (float*)smem;  // [3][Cc]
    const int n = blockIdx.y;
    for (int i = threadIdx.x; i < 3 * Cc; i += 256) hs[i] = hyper[(long long)n * 3 * Cc + i];
    __syncthreads();
    for (long long p = blockIdx.x * 256LL + threadIdx.x; p < vox; p += (long long)gridDim.x * 256) {
        const T* xp = up + ((long long)n * vox + p) * Cc;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int c0 = 0; c0 < Cc; c0 += 8) {
            float x[8];
            Vec8<T>::load(xp + c0, x);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                a0 += x[e] * hs[c0 + e];
                a1 += x[e] * hs[Cc + c0 + e];
                a2 += x[e] * hs[2 * Cc + c0 + e];
            }
        }
        masks[((long long)n * 3 + 0) * vox + p] = a0;
        masks[((long long)n * 3 + 1) * vox + p] = a1;
        masks[((long long)n * 3 + 2) * vox + p] = a2;
    }
}

// -------------------------------------------------------------------------------------------------
// Read-out (sparse_heads.py:572-589,645-647) fused: trilinear resize (align_corners=False; identity in
// time when Tl == T) of the low-res logits [N][3][Tl][h][w] to H x W, then per (n, t):
//   channel 0 -> soft-argmax xy over H*W with pixel-centre grid (+0.5);  channel 1 -> spatial mean (vis);
//   channel 2 -> exp(spatial mean) (depth).
// One workgroup per (n, t); the three low-res maps sit in LDS.
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ void src_idx_nc(int dst, int in, int out, int& i0, int& i1, float& lam) {
    float src = ((float)in / (float)out) * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int)src;
    if (i0 > in - 1) i0 = in - 1;
    lam = fminf(fmaxf(src - (float)i0, 0.f), 1.f);
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
}

// A thread owns output columns x = tid, tid + 256, ... and walks down the rows: its x source indices / weight stay in
// registers and the row's y indices / weight are wave-uniform, so a sample is 4 LDS reads + 3 lerps (the flat-index form
// recomputed two source indices and an integer division per sample and was VALU-bound: 242 us per clip).
__global__ __launch_bounds__(256) void track_readout_kernel(const float* __restrict__ masks, float* __restrict__ traj,
                                                            float* __restrict__ vis, float* __restrict__ depth, int T,
                                                            int h, int w, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lo = (float*)smem;  // [3][h*w]
    __shared__ float red[4][8];
    const int n = blockIdx.x / T, t = blockIdx.x % T, tid = threadIdx.x;
    const int hw = h * w;
    for (int i = tid; i < 3 * hw; i += 256) {
        const int m = i / hw, r = i % hw;
        lo[i] = masks[(((long long)n * 3 + m) * T + t) * hw + r];
    }
    __syncthreads();
    auto lerp2 = [&](const float* b, int r0, int r1, int x0, int x1, float lx, float ly) -> float {
        const float top = (1.f - lx) * b[r0 + x0] + lx * b[r0 + x1];
        const float bot = (1.f - lx) * b[r1 + x0] + lx * b[r1 + x1];
        return (1.f - ly) * top + ly * bot;
    };
    // pass 1: an upper bound of channel 0 and the spatial sums of the up-sampled channels 1 and 2, all from the 64 x 64 source.
    // Bilinear interpolation is a convex combination, so max(source) bounds the up-sampled maximum - the soft-argmax below only
    // needs its exponents non-positive (softmax is shift-invariant) -, and the sum of an up-sampled channel is
    // sum_r b[r] * wy[row(r)] * wx[col(r)] with wx[i] / wy[i] = the total weight source column / row i receives from the W / H
    // output positions (accumulated per source index in a fixed order).  Replaces a second walk over all H x W samples of
    // three channels (12 LDS reads + 9 lerps per sample): 171 -> ~90 us per 64-track window.
    float* wxs = lo + 3 * hw;  // [w]
    float* wys = wxs + w;      // [h]
    for (int i = tid; i < w + h; i += 256) {
        const bool isx = i < w;
        const int idx = isx ? i : i - w, in = isx ? w : h, out = isx ? W : H;
        float acc = 0.f;
        for (int d = 0; d < out; ++d) {
            int i0, i1;
            float lam;
            src_idx_nc(d, in, out, i0, i1, lam);
            if (i0 == idx) acc += 1.f - lam;
            if (i1 == idx) acc += lam;
        }
        (isx ? wxs : wys)[idx] = acc;
    }
    __syncthreads();
    float mx = -INFINITY, s1 = 0.f, s2 = 0.f;
    for (int r = tid; r < hw; r += 256) {
        const float wgt = wys[r / w] * wxs[r % w];
        mx = fmaxf(mx, lo[r]);
        s1 += lo[hw + r] * wgt;
        s2 += lo[2 * hw + r] * wgt;
    }
    mx = wave_max(mx);
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if ((tid & 63) == 0) {
        red[tid >> 6][0] = mx;
        red[tid >> 6][1] = s1;
        red[tid >> 6][2] = s2;
    }
    __syncthreads();
    mx = fmaxf(fmaxf(red[0][0], red[1][0]), fmaxf(red[2][0], red[3][0]));
    s1 = red[0][1] + red[1][1] + red[2][1] + red[3][1];
    s2 = red[0][2] + red[1][2] + red[2][2] + red[3][2];
    __syncthreads();
    // pass 2: soft-argmax
    float z = 0.f, sx = 0.f, sy = 0.f;
    for (int x = tid; x < W; x += 256) {
        int x0, x1;
        float lx;
        src_idx_nc(x, w, W, x0, x1, lx);
        const float xc = (float)x + 0.5f;
        for (int y = 0; y < H; ++y) {
            int y0, y1;
            float ly;
            src_idx_nc(y, h, H, y0, y1, ly);
            // (expf, not v_exp_f32: the bare instruction is 13 us per launch faster, but its last-ulp differences moved one
            //  re-seeded query of the 4-window full-size bf16 run onto another frame - the track then differs by 7e-2 in depth)
            const float e = expf(lerp2(lo, y0 * w, y1 * w, x0, x1, lx, ly) - mx);
            z += e;
            sx += e * xc;
            sy += e * ((float)y + 0.5f);
        }
    }
    z = wave_sum(z);
    sx = wave_sum(sx);
    sy = wave_sum(sy);
    if ((tid & 63) == 0) {
        red[tid >> 6][0] = z;
        red[tid >> 6][1] = sx;
        red[tid >> 6][2] = sy;
    }
    __syncthreads();
    if (tid == 0) {
        z = red[0][0] + red[1][0] + red[2][0] + red[3][0];
        sx = red[0][1] + red[1][1] + red[2][1] + red[3][1];
        sy = red[0][2] + red[1][2] + red[2][2] + red[3][2];
        traj[((long long)n * 2 + 0) * T + t] = sx / z;
        traj[((long long)n * 2 + 1) * T + t] = sy / z;
        vis[(long long)n * T + t] = s1 / (float)(H * W);
        depth[(long long)n * T + t] = expf(s2 / (float)(H * W));
    }
}

// The same read-out on a 1024-thread workgroup (W <= 256; knob "readout_wide").  In the kernel above a thread owns an output column and
// walks its H rows: H x (bilinear sample + expf) serially per thread on 224 of 256 threads - 58 us per launch whatever the number of
// tracks, one launch per window.  Here the samples exp(logit - max) of RB rows at a time are formed by ALL threads into LDS, and the
// column threads then add them up in the SAME order (row after row: z, sum e x, sum e y per column, then the wave / workgroup sums of
// the kernel above): the same values in the same sums, bit for bit.
constexpr int READOUT_RB = 32;
__global__ __launch_bounds__(1024) void track_readout_wide_kernel(const float* __restrict__ masks, float* __restrict__ traj,
                                                                  float* __restrict__ vis, float* __restrict__ depth, int T,
                                                                  int h, int w, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lo = (float*)smem;  // [3][h*w]
    __shared__ float red[4][8];
    const int n = blockIdx.x / T, t = blockIdx.x % T, tid = threadIdx.x;
    const int hw = h * w;
    float* wxs = lo + 3 * hw;        // [w]
    float* wys = wxs + w;            // [h]
    float* tlx = wys + h;            // [W] x interpolation weight
    float* tly = tlx + W;            // [H]
    int* tx0 = (int*)(tly + H);      // [W] x0 | x1 << 16
    int* ty0 = tx0 + W;              // [H] y0 * w | (y1 * w) << 16
    float* eb = (float*)(ty0 + H);   // [RB][W]
    for (int i = tid; i < 3 * hw; i += 1024) {
        const int m = i / hw, r = i % hw;
        lo[i] = masks[(((long long)n * 3 + m) * T + t) * hw + r];
    }
    for (int i = tid; i < W + H; i += 1024) {
        const bool isx = i < W;
        const int d = isx ? i : i - W;
        int i0, i1;
        float lam;
        src_idx_nc(d, isx ? w : h, isx ? W : H, i0, i1, lam);
        if (isx) {
            tlx[d] = lam;
            tx0[d] = i0 | (i1 << 16);
        } else {
            tly[d] = lam;
            ty0[d] = (i0 * w) | ((i1 * w) << 16);
        }
    }
    __syncthreads();
    auto lerp2 = [&](const float* b, int r0, int r1, int x0, int x1, float lx, float ly) -> float {
        const float top = (1.f - lx) * b[r0 + x0] + lx * b[r0 + x1];
        const float bot = (1.f - lx) * b[r1 + x0] + lx * b[r1 + x1];
        return (1.f - ly) * top + ly * bot;
    };
    for (int i = tid; i < w + h; i += 1024) {
        const bool isx = i < w;
        const int idx = isx ? i : i - w, in = isx ? w : h, out = isx ? W : H;
        float acc = 0.f;
        for (int d = 0; d < out; ++d) {
            int i0, i1;
            float lam;
            src_idx_nc(d, in, out, i0, i1, lam);
            if (i0 == idx) acc += 1.f - lam;
            if (i1 == idx) acc += lam;
        }
        (isx ? wxs : wys)[idx] = acc;
    }
    __syncthreads();
    float mx = -INFINITY, s1 = 0.f, s2 = 0.f;
    if (tid < 256) {  // (the partial sums of the 256-thread kernel: thread tid takes r = tid, tid + 256, ...)
        for (int r = tid; r < hw; r += 256) {
            const float wgt = wys[r / w] * wxs[r % w];
            mx = fmaxf(mx, lo[r]);
            s1 += lo[hw + r] * wgt;
            s2 += lo[2 * hw + r] * wgt;
        }
        mx = wave_max(mx);
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if ((tid & 63) == 0) {
            red[tid >> 6][0] = mx;
            red[tid >> 6][1] = s1;
            red[tid >> 6][2] = s2;
        }
    }
    __syncthreads();
    mx = fmaxf(fmaxf(red[0][0], red[1][0]), fmaxf(red[2][0], red[3][0]));
    s1 = red[0][1] + red[1][1] + red[2][1] + red[3][1];
    s2 = red[0][2] + red[1][2] + red[2][2] + red[3][2];
    __syncthreads();
    // pass 2: soft-argmax
    float z = 0.f, sx = 0.f, sy = 0.f;
    const float xc = (float)tid + 0.5f;
    for (int yb = 0; yb < H; yb += READOUT_RB) {
        const int rows = H - yb < READOUT_RB ? H - yb : READOUT_RB;
        for (int i = tid; i < rows * W; i += 1024) {
            const int yy = i / W, x = i - yy * W;
            const int px = tx0[x], py = ty0[yb + yy];
            eb[i] = expf(lerp2(lo, py & 0xFFFF, py >> 16, px & 0xFFFF, px >> 16, tlx[x], tly[yb + yy]) - mx);
        }
        __syncthreads();
        if (tid < W) {
            for (int yy = 0; yy < rows; ++yy) {
                const float e = eb[yy * W + tid];
                z += e;
                sx += e * xc;
                sy += e * ((float)(yb + yy) + 0.5f);
            }
        }
        __syncthreads();
    }
    if (tid < 256) {
        z = wave_sum(z);
        sx = wave_sum(sx);
        sy = wave_sum(sy);
        if ((tid & 63) == 0) {
            red[tid >> 6][0] = z;
            red[tid >> 6][1] = sx;
            red[tid >> 6][2] = sy;
        }
    }
    __syncthreads();
    if (tid == 0) {
        z = red[0][0] + red[1][0] + red[2][0] + red[3][0];
        sx = red[0][1] + red[1][1] + red[2][1] + red[3][1];
        sy = red[0][2] + red[1][2] + red[2][2] + red[3][2];
        traj[((long long)n * 2 + 0) * T + t] = sx / z;
        traj[((long long)n * 2 + 1) * T + t] = sy / z;
        vis[(long long)n * T + t] = s1 / (float)(H * W);
        depth[(long long)n * T + t] = expf(s2 / (float)(H * W));
    }
}

// -------------------------------------------------------------------------------------------------
// Sliding-window bookkeeping (forward_windowed_core, sparse_heads.py:303-335, :366-393, :455-486).
// All integer / boolean decisions are made here, bit-for-bit as the reference's float comparisons.
//   prepare: q_off = cur_q with time shifted by -start; valid_t[n][j] = (j + start + 0.5 - cur_q.t >= 0);
//            valid_n = any(valid_t); label = 0/1 by valid_n, then 1 if ANY coordinate of cur_q equals the
//            original query (:330-332), else 2 if valid (:334-335).
//   commit : masked scatter of the window estimates into the clip buffers; prompt feature carry;
//            re-seed the query at argmax visibility over the overlap with the next window.
// -------------------------------------------------------------------------------------------------
__global__ void track_prepare_kernel(const float* __restrict__ cur_q, const float* __restrict__ orig_q, int start, int ws,
                                     float* __restrict__ q_off, float* __restrict__ labels,
                                     unsigned char* __restrict__ valid_t, unsigned char* __restrict__ valid_n, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float qt = cur_q[n * 3], qx = cur_q[n * 3 + 1], qy = cur_q[n * 3 + 2];
    bool any = false;
    for (int j = 0; j < ws; ++j) {
        const float tj = (float)(j + start) + 0.5f;  // arange + start + 0.5 (exact in float)
        const bool ok = (tj - qt) >= 0.f;
        valid_t[n * ws + j] = ok ? 1 : 0;
        any |= ok;
    }
    valid_n[n] = any ? 1 : 0;
    q_off[n * 3] = qt - (float)start;
    q_off[n * 3 + 1] = qx;
    q_off[n * 3 + 2] = qy;
    float lab = any ? 1.f : 0.f;
    const bool same = (qt == orig_q[n * 3]) || (qx == orig_q[n * 3 + 1]) || (qy == orig_q[n * 3 + 2]);
    if (same) lab = 1.f;
    if (any && !same) lab = 2.f;
    labels[n] = lab;
}

__global__ void track_commit_kernel(const float* __restrict__ w_traj, const float* __restrict__ w_vis,
                                    const float* __restrict__ w_depth, const unsigned char* __restrict__ valid_t,
                                    const unsigned char* __restrict__ valid_n, float* __restrict__ traj,
                                    float* __restrict__ vis, float* __restrict__ depth, int T, int start, int ws, int next_start,
                                    int last_window, float* __restrict__ cur_q, float* __restrict__ plabel,
                                    int* __restrict__ best_out, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    for (int j = 0; j < ws; ++j) {
        if (!valid_t[n * ws + j]) continue;
        vis[(long long)n * T + start + j] = w_vis[n * ws + j];
        depth[(long long)n * T + start + j] = w_depth[n * ws + j];
        traj[((long long)n * 2 + 0) * T + start + j] = w_traj[(n * 2 + 0) * ws + j];
        traj[((long long)n * 2 + 1) * T + start + j] = w_traj[(n * 2 + 1) * ws + j];
    }
    if (last_window) return;
    if (valid_n[n]) plabel[n] = 1.f;
    // argmax (first maximum) of the stitched visibility over [next_start, start + ws)
    int best = 0;
    float bv = vis[(long long)n * T + next_start];
    for (int j = 1; j < start + ws - next_start; ++j) {
        const float v = vis[(long long)n * T + next_start + j];
        if (v > bv) {
            bv = v;
            best = j;
        }
    }
    if (best_out) best_out[n] = best;
    const float nt = (float)best + (float)next_start + 0.5f;
    if (nt > cur_q[n * 3]) {
        cur_q[n * 3] = nt;
        cur_q[n * 3 + 1] = traj[((long long)n * 2 + 0) * T + next_start + best];
        cur_q[n * 3 + 2] = traj[((long long)n * 2 + 1) * T + next_start + best];
    }
}

// prompt feature carry: pfeat[n] = new[n] where valid_n (sparse_heads.py:389-393)
__global__ void masked_rows_copy_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                        const unsigned char* __restrict__ mask, int N, int C) {
    const long long total = (long long)N * (C / 4);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i / (C / 4));
        if (mask[n]) ((f32x4*)dst)[i] = ((const f32x4*)src)[i];
    }
}

// ------------------------------------------------------------------------------------------------
// Image -> token attention with the projections folded into the token side (sam/transformer.py:180-185, 223-245; see
// sparse_heads.py / api_trackwin.hip "folded i2t"): the scores of an image token against the 6 prompt tokens of its track arrive
// as a float row [tokens][heads] (column t * heads + h; scale and the query-bias term already inside), the softmax over the
// tokens of every head leaves as a row of the engine dtype padded to `ldp` columns - the A operand of P x V'.
// One thread per (row, head); same exponential and normalisation order as i2t_attn_kernel (expf, one reciprocal).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void i2t_probs_kernel(const float* __restrict__ s, long long lds_, int pairs, const float* __restrict__ cb, int rows_per_group,
                                 T* __restrict__ p, int ldp, long long M, int heads, int tokens) {
    const long long total = M * heads;
    const int HT = heads * tokens;
    for (long long w = blockIdx.x * (long long)blockDim.x + threadIdx.x; w < total; w += (long long)gridDim.x * blockDim.x) {
        const long long row = w / heads;
        const int h = (int)(w - row * heads);
        const float* sr = s + row * lds_ + h;
        const float* cr = cb ? cb + (row / rows_per_group) * HT + h : nullptr;  // the track's query-bias term per (token, head)
        float e[8];
        float m = -INFINITY;
        for (int t = 0; t < tokens; ++t) {
            float v = sr[t * heads];
            if (pairs) v += sr[HT + t * heads];  // scores against the low halves of the folded key matrix (bf16 engine)
            if (cr) v += cr[t * heads];
            e[t] = v;
            m = fmaxf(m, v);
        }
        float z = 0.f;
        for (int t = 0; t < tokens; ++t) {
            e[t] = expf(e[t] - m);
            z += e[t];
        }
        const float iz = 1.f / z;
        T* pr = p + row * ldp + h;
        for (int t = 0; t < tokens; ++t) pr[t * heads] = from_f32<T>(e[t] * iz);
        if (h == 0)
            for (int c = HT; c < ldp; ++c) pr[c] = from_f32<T>(0.f);  // padding columns of the k dimension
    }
}
// in float [G][R][C] -> out T [G][2 R][C]: rows [0, R) = T(x), rows [R, 2 R) = T(x - float(T(x))).  The folded key matrix of a track
// as a pair of bf16 matrices, for its 1408-term dot products with the keys (an option of the folded image -> token attention:
// measured to make no difference to the tracker's distance from the f32 engine, so it is off by default).
template <typename T>
__global__ void split_hilo_kernel(const float* __restrict__ in, T* __restrict__ out, int G, int R, long long C) {
    const long long total = (long long)G * R * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long c = i % C, gr = i / C;
        const long long g = gr / R, r = gr % R;
        const float x = in[i];
        const T hi = from_f32<T>(x);
        out[((g * 2 * R) + r) * C + c] = hi;
        out[((g * 2 * R) + R + r) * C + c] = from_f32<T>(x - to_f32<T>(hi));
    }
}
// in [G][R][C] -> out [G][C][Rp] (rows R .. Rp-1 of the k dimension zero): the folded value matrix of a track, transposed into
// the k-contiguous form the GEMM reads weights in.  Small (G * R * C elements = 4.3 M for 64 tracks): one thread per output element.
template <typename T>
__global__ void transpose_pad_kernel(const T* __restrict__ in, T* __restrict__ out, int G, int R, int C, int Rp) {
    const long long total = (long long)G * C * Rp;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i % Rp);
        const long long gc = i / Rp;
        const int c = (int)(gc % C);
        const long long g = gc / C;
        out[i] = r < R ? in[(g * R + r) * C + c] : from_f32<T>(0.f);
    }
}

// ------------------------------------------------------------------------------------------------
// delta = P x V' + b_out of the folded image -> token attention (bf16, k = 64): probs [N][P][64], V'^T [N][C][64] (l4p_transpose_pad),
// delta [N][P][C].  The product is 2 MFMA k-steps per output tile; what it costs is the [N * P, C] result (369 MB per 64 tracks).
// As a GEMM with one k-tile it ran 11 264 one-shot workgroups (load, 8 MFMAs per wave, store: 2.7 TB/s).  Here a workgroup owns
// 128 columns of a track and half of its rows: the 128 x 64 block of V'^T lives in registers (64 VGPRs), every wave streams 16-row
// tiles on its own - probs straight into the A fragments (2 x 16 bytes per lane, next tile requested before this one is used), 16
// MFMAs, + bias, rounded, through a private 4 KB LDS tile into whole 256-byte row segments - no workgroup barrier anywhere.
// Same products in the same order as the GEMM (two k-steps per accumulator): bit-identical to it.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void i2t_delta_kernel(const T* __restrict__ probs, const T* __restrict__ vt,
                                                        const float* __restrict__ bias, T* __restrict__ delta, int P, int C,
                                                        int rows_per_wg) {
    constexpr int K = 64, CW = 128;
    __shared__ __attribute__((aligned(16))) T tile[4][16 * CW];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, i16 = lane & 15;
    const int c0 = blockIdx.x * CW, n = blockIdx.y, r0 = blockIdx.z * rows_per_wg;
    const int rows = P - r0 < rows_per_wg ? P - r0 : rows_per_wg;
    const int ntile = rows / 16;
    const T* vb = vt + ((long long)n * C + c0) * K;
    vec8<T> fb[8][2];
    float bz[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) fb[j][kk] = *(const vec8<T>*)(vb + (long long)(16 * j + i16) * K + 32 * kk + 8 * g);
        bz[j] = bias ? bias[c0 + 16 * j + i16] : 0.f;
    }
    const T* pa = probs + ((long long)n * P + r0) * K + (long long)i16 * K + 8 * g;
    T* out = delta + ((long long)n * P + r0) * C + c0;
    T* tl = tile[wave];
    vec8<T> a0 = {}, a1 = {}, n0 = {}, n1 = {};
    int t = wave;
    if (t < ntile) {
        a0 = *(const vec8<T>*)(pa + (long long)t * 16 * K);
        a1 = *(const vec8<T>*)(pa + (long long)t * 16 * K + 32);
    }
    for (; t < ntile; t += 4) {
        if (t + 4 < ntile) {
            n0 = *(const vec8<T>*)(pa + (long long)(t + 4) * 16 * K);
            n1 = *(const vec8<T>*)(pa + (long long)(t + 4) * 16 * K + 32);
        }
        f32x4 acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[j] = mma16(a0, fb[j][0], (f32x4){0.f, 0.f, 0.f, 0.f});
            acc[j] = mma16(a1, fb[j][1], acc[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) tl[(4 * g + e) * CW + 16 * j + i16] = (T)(acc[j][e] + bz[j]);
        __builtin_amdgcn_wave_barrier();  // (the tile is this wave's own: LDS operations of a wave complete in order)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = q * 64 + lane, row = idx >> 4, part = idx & 15;
            const u32x4 v = *(const u32x4*)(tl + row * CW + part * 8);
            *(u32x4*)(out + (long long)(t * 16 + row) * C + part * 8) = v;
        }
        __builtin_amdgcn_wave_barrier();
        a0 = n0;
        a1 = n1;
    }
}

// ------------------------------------------------------------------------------------------------
// Token -> image attention with the VALUE projection folded away as well (sam/transformer.py:168-173,223-245; round 4).
//   out[t, head h] = sum_p prob[t,h,p] (keys[p] Wv_h^T + bv_h) = (sum_p prob[t,h,p] keys[p]) Wv_h^T + bv_h     (sum_p prob = 1)
// so the [P, C] x [C, C/2] value projection of every track (4.06 GF, a 2.9 MB tensor per track) becomes a context
// ctx[(t,h)] = prob[(t,h)] x keys - 48 x 2048 x 1408 per track, HBM-bound on ONE read of the keys - and a 48-row projection.
//   t2i_probs: scores [N][P][48] (float; column t * heads + h) -> e [N][P][48] (T) = exp(score - m) with m the column maximum over
//              the SPLIT of 256 keys the row belongs to, and stats [N][P / 256][2][48] (float): that maximum and the split's sum of e.
//              (the softmax over all P keys is assembled by t2i_ctx: a split's terms carry exp(m_split - M) / Z.  One pass, fully
//              coalesced - a track's [P][48] block is one contiguous run - over 8 x N workgroups; a whole-column softmax needs
//              either three passes of N workgroups (measured 91 us at 64 tracks) or 192-byte-strided column groups (44 us))
//   t2i_ctx:   ctx[(h * Rg + n * tokens + t)][c] = sum_splits scale * sum_{p in split} e[n][p][t * heads + h] * keys[n][p][c]
//              (rows grouped by head, Rg rows per group: the A operand of the row-grouped-weights GEMM against Wv's head blocks)
// ------------------------------------------------------------------------------------------------
constexpr int T2I_SPLIT = 256;  // keys per softmax split
template <typename T>
__global__ __launch_bounds__(192) void t2i_probs_kernel(const float* __restrict__ s, long long ld, T* __restrict__ probs,
                                                        float* __restrict__ stats, int P) {
    constexpr int HT = 48, NCG = 12, NRL = 16, NR = T2I_SPLIT / NRL;
    __shared__ __attribute__((aligned(16))) float red[NRL * HT];
    __shared__ __attribute__((aligned(16))) float stat[HT];
    const int sp_ = blockIdx.x, n = blockIdx.y, tid = threadIdx.x, cg = tid % NCG, rl = tid / NCG;
    const int p0 = sp_ * T2I_SPLIT;
    const float* sp = s + ((long long)n * P + p0) * ld + 4 * cg;
    f32x4 v[NR];
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int r = i * NRL + rl;
        v[i] = p0 + r < P ? *(const f32x4*)(sp + (long long)r * ld) : (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    }
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[i][e]);
    *(f32x4*)(red + rl * HT + 4 * cg) = m;
    __syncthreads();
    if (tid < HT) {
        float x = -INFINITY;
        for (int r = 0; r < NRL; ++r) x = fmaxf(x, red[r * HT + tid]);
        stat[tid] = x;
    }
    __syncthreads();
    m = *(const f32x4*)(stat + 4 * cg);
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // (rows past P: exp(-inf) = 0.  The split's sum is the sum of the ROUNDED terms, the ones the context product adds up:
            //  the weights then sum to one exactly, whatever the rounding did to a peaked column's few large terms)
            v[i][e] = to_f32<T>(from_f32<T>(expf(v[i][e] - m[e])));
            z[e] += v[i][e];
        }
    __syncthreads();
    *(f32x4*)(red + rl * HT + 4 * cg) = z;
    __syncthreads();
    float* st = stats + ((long long)n * gridDim.x + sp_) * 2 * HT;
    if (tid < HT) {
        float x = 0.f;
        for (int r = 0; r < NRL; ++r) x += red[r * HT + tid];
        st[tid] = stat[tid];
        st[HT + tid] = x;
    }
    T* pp = probs + ((long long)n * P + p0) * HT + 4 * cg;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int r = i * NRL + rl;
        if (p0 + r < P) {
            const float o[4] = {v[i][0], v[i][1], v[i][2], v[i][3]};
            Vec4<T>::store(pp + (long long)r * HT, o);
        }
    }
}
// scale[split][column] = exp(m_split - M) / Z from the splits' statistics of track n (M: the column maximum over all splits,
// Z = sum_splits z_split exp(m_split - M)); nsp <= 16
__device__ __forceinline__ void t2i_scales(const float* __restrict__ stats, int n, int nsp, float* scale /* LDS [16][48] */) {
    constexpr int HT = 48;
    const int tid = threadIdx.x;
    if (tid < HT) {
        const float* st = stats + (long long)n * nsp * 2 * HT;
        float M = -INFINITY;
        for (int q = 0; q < nsp; ++q) M = fmaxf(M, st[q * 2 * HT + tid]);
        float Z = 0.f;
        for (int q = 0; q < nsp; ++q) Z += st[q * 2 * HT + HT + tid] * expf(st[q * 2 * HT + tid] - M);
        const float iz = 1.f / Z;
        for (int q = 0; q < nsp; ++q) scale[q * HT + tid] = expf(st[q * 2 * HT + tid] - M) * iz;
    }
    __syncthreads();
}

// bf16: MFMA form.  Workgroup = (128 columns of C, track); 4 waves x 32 columns, all HT rows.  Per 32-key step a stage of
// keys [32][128] (8 KB) + probs [32][HT] (3 KB for HT = 48) lands by LDS-DMA (both are row-major images of global memory: the
// probs of 32 keys are one contiguous run); both MFMA operands are k(= key)-strided in those images and are read with the
// transposing LDS read (ds_read_b64_tr_b16: a 16-lane group reads a [4 keys][16 columns] block, lane i receives column i).
// 4-stage ring, two stages in flight behind the one being read, counted vmcnt + one raw barrier per step.  The kernel is bound by
// the single read of the keys (5.8 MB per track); per step a wave issues 10 LDS reads and 6 MFMAs.
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
template <typename T>
__device__ __forceinline__ vec8<T> tr_frag(const char* lo, int hi_off) {
    typedef __attribute__((address_space(3))) s16x4_t* lp_t;
    const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(lo));
    const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(lo + hi_off));
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    const s16x8_t r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(vec8<T>, r);
}
// (ST = 8, six stages in flight, was measured on a rank's 8-track shard - 88 workgroups, each alone on its CU: no faster, 49.7 vs 51.9 us;
//  tools/probes/ctx_ablate.sh: a step's 0.7 us is the stage wait + barrier (0.22 us) and the fragment reads -> MFMAs (0.27 us) one
//  after the other, not the memory round trip)
// shared_from < P: the rows p >= shared_from of EVERY track are read from track 0's block (later windows of the recursion, layer 0: the
// second temporal half of every track's keys is still track 0's; a multiple of 32).
template <typename T, int HT, int CW, int ST = 4, int UNR = 1>
__global__ __launch_bounds__(256) void t2i_ctx_mfma_kernel(const T* __restrict__ probs, const T* __restrict__ keys,
                                                           T* __restrict__ ctx, const float* __restrict__ stats, int P, int C, int heads,
                                                           int tokens, long long Rg, int shared_from) {
    static_assert(HT == 48, "t2i_scales");
    static_assert(ST == 4 || ST == 8 || ST == 12, "ring depth");
    constexpr int KT = 32, MT = HT / 16, NTW = CW / 64;  // NTW: 16-column tiles per wave
    constexpr int SPK = T2I_SPLIT / KT;                          // key steps per softmax split
    constexpr int KB = KT * CW * 2, PB = KT * HT * 2, SB = KB + PB;
    constexpr int KP = KB / 4096;  // 1 KB pieces of the key stage per wave
    constexpr int PW = PB / 1024;  // waves that carry a 1 KB piece of the probs stage (the others repeat the last piece)
    constexpr int CPR = CW / 8;    // 16-byte chunks per key row
    static_assert(HT % 16 == 0 && PB % 1024 == 0 && PW >= 1 && PW <= 4 && (CW == 64 || CW == 128), "stages in whole 1 KB pieces");
    typedef __attribute__((address_space(3))) void* lptr_t;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float scale[16 * HT];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c0 = blockIdx.x * CW, n = blockIdx.y;
    const char* kb_own = (const char*)(keys + (long long)n * P * C + c0);
    const char* kb_shared = (const char*)(keys + c0);
    const char* pb = (const char*)(probs + (long long)n * P * HT);
    const int ns = P / KT;
    auto issue = [&](int s) {
        char* dst = smem + (s % ST) * SB;
        const long long p0 = (long long)s * KT;
        const char* kb = p0 >= shared_from ? kb_shared : kb_own;
#pragma unroll
        for (int i = 0; i < KP; ++i) {
            const int q = wave * 64 + i * 256 + lane;  // 16-byte chunk of the [32][CW] tile: row q / CPR, part q % CPR
            const char* src = kb + (p0 + q / CPR) * C * 2 + (q % CPR) * 16;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + (wave * 64 + i * 256) * 16), 16, 0, 0);
        }
        const int pw = wave < PW ? wave : PW - 1;
        __builtin_amdgcn_global_load_lds((gptr_t)(pb + p0 * HT * 2 + (pw * 64 + lane) * 16), (lptr_t)(dst + KB + pw * 1024), 16, 0, 0);
    };
    const int g = lane >> 4, i16 = lane & 15;
    const int a_off = KB + ((8 * g + (i16 >> 2)) * HT + 4 * (i16 & 3)) * 2;
    const int b_off = ((8 * g + (i16 >> 2)) * CW + wave * (16 * NTW) + 4 * (i16 & 3)) * 2;
    f32x4 acc[MT][NTW], tot[MT][NTW];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[m][j] = tot[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // UNR stages are consumed per barrier (a "group"); AHG groups stay in flight behind the one being read, one more is requested into
    // the slots of the group read before (UNR = 1, ST = 4: the chip-filling form; UNR = 2, ST = 8: a launch of at most one workgroup
    // per CU, where a step is the stage wait + barrier and then the fragment reads -> MFMAs, one after the other: half the barriers, and
    // the LDS latency of two stages' fragments overlaps).  MFMAs run in stage order: the same sums in the same order.
    constexpr int NGS = ST / UNR, AHG = NGS - 2;
    static_assert(ST % UNR == 0 && AHG >= 1 && AHG <= 6, "ring of whole groups");
    const int ngr = ns / UNR;  // (the launcher guarantees ns % UNR == 0)
#pragma unroll
    for (int gq = 0; gq <= AHG; ++gq)
        if (gq < ngr) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) issue(gq * UNR + u);
        }
    t2i_scales(stats, n, (P + T2I_SPLIT - 1) / T2I_SPLIT, scale);
    for (int gi = 0; gi < ngr; ++gi) {
        const int ahead = ngr - 1 - gi < AHG ? ngr - 1 - gi : AHG;
        constexpr int LPG = UNR * (KP + 1);  // loads of a wave per group
        switch (ahead) {
            case 6: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * LPG > 63 ? 63 : 6 * LPG) : "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * LPG > 63 ? 63 : 5 * LPG) : "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * LPG > 63 ? 63 : 4 * LPG) : "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPG > 63 ? 63 : 3 * LPG) : "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPG) : "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPG) : "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if !(defined(CTX_ABL) && CTX_ABL == 3)  // (CTX_ABL: timing ablations of tools/probes/ctx_ablate.sh, wrong results)
        __builtin_amdgcn_s_barrier();
#endif
#if !(defined(CTX_ABL) && CTX_ABL == 2)
        if (gi + AHG + 1 < ngr) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) issue((gi + AHG + 1) * UNR + u);
        }
#endif
        vec8<T> fa[UNR][MT], fb[UNR][NTW];
#if !(defined(CTX_ABL) && CTX_ABL == 1)
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const char* base = smem + ((gi * UNR + u) % ST) * SB;
#pragma unroll
            for (int m = 0; m < MT; ++m) fa[u][m] = tr_frag<T>(base + a_off + m * 32, 4 * HT * 2);
#pragma unroll
            for (int j = 0; j < NTW; ++j) fb[u][j] = tr_frag<T>(base + b_off + j * 32, 4 * CW * 2);
        }
#endif
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int s = gi * UNR + u;
#if !(defined(CTX_ABL) && CTX_ABL == 1)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int j = 0; j < NTW; ++j) acc[m][j] = mma16(fa[u][m], fb[u][j], acc[m][j]);
#endif
            if ((s % SPK) == SPK - 1 || s == ns - 1) {  // end of a softmax split: its sum joins the total with the split's weight
                const float* sc = scale + (s / SPK) * HT + 4 * g;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const f32x4 w = *(const f32x4*)(sc + m * 16);
#pragma unroll
                    for (int j = 0; j < NTW; ++j) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) tot[m][j][e] += w[e] * acc[m][j][e];
                        acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int tp = m * 16 + 4 * g + e;  // score column t * heads + h
                const int t = tp / heads, h = tp - t * heads;
                if (t < tokens)
                    ctx[((long long)h * Rg + (long long)n * tokens + t) * C + c0 + wave * (16 * NTW) + j * 16 + i16] = (T)tot[m][j][e];
            }
}
// any dtype (the f32 engine): one thread per column, the HT running sums in registers, probs rows read as wave-uniform values
template <typename T, int HT>
__global__ __launch_bounds__(256) void t2i_ctx_kernel(const T* __restrict__ probs, const T* __restrict__ keys, T* __restrict__ ctx,
                                                      const float* __restrict__ stats, int P, int C, int heads, int tokens, long long Rg,
                                                      int shared_from) {
    __shared__ float scale[16 * HT];
    const int c = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
    t2i_scales(stats, n, (P + T2I_SPLIT - 1) / T2I_SPLIT, scale);
    if (c >= C) return;
    float acc[HT], tot[HT];
#pragma unroll
    for (int t = 0; t < HT; ++t) acc[t] = tot[t] = 0.f;
    const T* kp_own = keys + (long long)n * P * C + c;
    const T* kp_shared = keys + c;
    const T* pp = probs + (long long)n * P * HT;
    for (int p = 0; p < P; ++p) {
        const float kv = to_f32<T>((p >= shared_from ? kp_shared : kp_own)[(long long)p * C]);
#pragma unroll
        for (int t = 0; t < HT; ++t) acc[t] += to_f32<T>(pp[(long long)p * HT + t]) * kv;
        if ((p % T2I_SPLIT) == T2I_SPLIT - 1 || p == P - 1) {
#pragma unroll
            for (int t = 0; t < HT; ++t) {
                tot[t] += scale[(p / T2I_SPLIT) * HT + t] * acc[t];
                acc[t] = 0.f;
            }
        }
    }
#pragma unroll
    for (int tp = 0; tp < HT; ++tp) {
        const int t = tp / heads, h = tp - t * heads;
        if (t < tokens) ctx[((long long)h * Rg + (long long)n * tokens + t) * C + c] = from_f32<T>(tot[tp]);
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
#define GRID1D(total, cap) ((int)(((total) + 255) / 256 < (cap) ? ((total) + 255) / 256 : (cap)))

// tokens -> image attention from scores formed elsewhere (folded keys): softmax over the P keys per (track, token, head), then P.V
int launch_t2i_attn_scores(int dtype, const float* scores, long long ld_scores, const void* v, void* out, int N, int P, int D, int heads,
                           hipStream_t stream) {
    const int hd = heads > 0 ? D / heads : 0;
    if (heads < 1 || hd * heads != D || hd % 4 || hd > 96 || P % 4 || ld_scores < 6 * heads) {
        l4p_set_error("t2i_attn_scores: D = heads * hd with hd %% 4 == 0, hd <= 96, P %% 4 == 0, ld_scores >= 6 * heads");
        return L4P_E_INVALID;
    }
    ProfScope prof(PROF_TRACK, stream, "small_attn kind5 N%d P%d D%d", N, P, D);
    const int ncg = hd / 4, nkg = 256 / ncg;
    size_t lds = (size_t)(6 * P + 6 * 96 + 256) * 4;
    const size_t need2 = (size_t)nkg * 6 * hd * 4;
    if (need2 > (size_t)6 * P * 4) lds += need2 - (size_t)6 * P * 4;
    const long long kv_stride = (long long)P * D;
    if (is16(dtype)) L4P_WITH_T16(dtype, T16, {
        auto kb = t2i_attn_kernel<T16>;
        HIP_TRY(hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kb, dim3(N, heads), dim3(256), lds, stream, (const T16*)nullptr, (const T16*)nullptr, (const T16*)v,
                           (T16*)out, P, D, hd, 1.f, kv_stride, scores, ld_scores, heads);
    }); else {
        auto kf = t2i_attn_kernel<float>;
        HIP_TRY(hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kf, dim3(N, heads), dim3(256), lds, stream, (const float*)nullptr, (const float*)nullptr, (const float*)v,
                           (float*)out, P, D, hd, 1.f, kv_stride, scores, ld_scores, heads);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}
int launch_i2t_probs(int dtype, const float* s, long long lds_, int pairs, const float* cbias, int rows_per_group, void* p, int ldp,
                     long long M, int heads, int tokens, hipStream_t stream) {
    if (tokens < 1 || tokens > 8 || heads < 1 || ldp < tokens * heads || lds_ < (pairs ? 2 : 1) * tokens * heads || (cbias && rows_per_group < 1)) {
        l4p_set_error("i2t_probs: 1 <= tokens <= 8, ldp >= tokens * heads, score row stride >= (pairs ? 2 : 1) * tokens * heads");
        return L4P_E_INVALID;
    }
    ProfScope prof(PROF_TRACK, stream, "i2t_probs");
    const int grid = GRID1D(M * heads, 16384);
    if (is16(dtype)) L4P_WITH_T16(dtype, T16, hipLaunchKernelGGL(i2t_probs_kernel<T16>, dim3(grid), dim3(256), 0, stream, s, lds_, pairs, cbias, rows_per_group, (T16*)p, ldp, M,
                           heads, tokens));
    else
        hipLaunchKernelGGL(i2t_probs_kernel<float>, dim3(grid), dim3(256), 0, stream, s, lds_, pairs, cbias, rows_per_group, (float*)p, ldp, M, heads,
                           tokens);
    HIP_TRY(hipGetLastError());
    return 0;
}
int launch_split_hilo(int dtype, const float* in, void* out, int G, int R, long long C, hipStream_t stream) {
    if (G < 1 || R < 1 || C < 1) {
        l4p_set_error("split_hilo: bad shape");
        return L4P_E_INVALID;
    }
    ProfScope prof(PROF_TRACK, stream, "split_hilo");
    const int grid = GRID1D((long long)G * R * C, 16384);
    if (is16(dtype)) L4P_WITH_T16(dtype, T16, hipLaunchKernelGGL(split_hilo_kernel<T16>, dim3(grid), dim3(256), 0, stream, in, (T16*)out, G, R, C));
    else
        hipLaunchKernelGGL(split_hilo_kernel<float>, dim3(grid), dim3(256), 0, stream, in, (float*)out, G, R, C);
    HIP_TRY(hipGetLastError());
    return 0;
}
int launch_transpose_pad(int dtype, const void* in, void* out, int G, int R, int C, int Rp, hipStream_t stream) {
    if (G < 1 || R < 1 || C < 1 || Rp < R) {
        l4p_set_error("transpose_pad: bad shape");
        return L4P_E_INVALID;
    }
    ProfScope prof(PROF_TRACK, stream, "transpose_pad");
    const int grid = GRID1D((long long)G * C * Rp, 16384);
    if (is16(dtype)) L4P_WITH_T16(dtype, T16, hipLaunchKernelGGL(transpose_pad_kernel<T16>, dim3(grid), dim3(256), 0, stream, (const T16*)in, (T16*)out, G, R, C, Rp));
    else
        hipLaunchKernelGGL(transpose_pad_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)in, (float*)out, G, R, C, Rp);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_i2t_delta(int dtype, const void* probs, const void* vt, const float* bias, void* delta, int N, int P, int C, int K,
                     hipStream_t stream) {
    if (!is16(dtype) || K != 64 || C % 128 || P % 16 || N < 1) {
        l4p_set_error("i2t_delta: 16-bit engine, k = 64, C %% 128 == 0, P %% 16 == 0 (other shapes: the row-grouped-weights GEMM)");
        return L4P_E_INVALID;
    }
    // (a batched GEMM - profiled with the small / streaming products under the tag of the launch it replaces: the executed-shapes check sees it)
    ProfScope prof(PROF_GEMM_SMALL, stream, "M%lld N%d K%d epi0 act0 delta t16x128 wgrp", (long long)N * P, C, K);
    const int rs = P % 128 == 0 ? 2 : 1;
    L4P_WITH_T16(dtype, T16, hipLaunchKernelGGL(i2t_delta_kernel<T16>, dim3(C / 128, N, rs), dim3(256), 0, stream, (const T16*)probs, (const T16*)vt, bias,
                       (T16*)delta, P, C, P / rs));
    HIP_TRY(hipGetLastError());
    return 0;
}
int launch_t2i_probs(int dtype, const float* scores, long long ld_scores, void* probs, float* stats, int N, int P, int HT, hipStream_t stream) {
    if (N < 1 || P < 1 || P > 16 * T2I_SPLIT || HT != 48 || ld_scores < HT || ld_scores % 4) {
        l4p_set_error("t2i_probs: HT == 48, P <= 4096, score row stride >= HT and a multiple of 4");
        return L4P_E_INVALID;
    }
    ProfScope prof(PROF_TRACK, stream, "t2i_probs");
    const dim3 grid((P + T2I_SPLIT - 1) / T2I_SPLIT, N);
    if (is16(dtype)) L4P_WITH_T16(dtype, T16, hipLaunchKernelGGL(t2i_probs_kernel<T16>, grid, dim3(192), 0, stream, scores, ld_scores, (T16*)probs, stats, P));
    else
        hipLaunchKernelGGL(t2i_probs_kernel<float>, grid, dim3(192), 0, stream, scores, ld_scores, (float*)probs, stats, P);
    HIP_TRY(hipGetLastError());
    return 0;
}
int launch_t2i_context(int dtype, const void* probs, const float* stats, const void* keys, void* ctx, int N, int P, int C, int heads,
                       int tokens, long long Rg, int shared_from, hipStream_t stream) {
    if (N < 1 || heads * tokens != 48 || C % 64 || P % 32 || P < 96 || P > 16 * T2I_SPLIT || Rg < (long long)N * tokens || shared_from < 0 ||
        shared_from % 32) {
        l4p_set_error("t2i_context: heads * tokens == 48, C %% 64 == 0, P %% 32 == 0, 96 <= P <= 4096, Rg >= N * tokens, shared_from %% 32 == 0");
        return L4P_E_INVALID;
    }
    if (shared_from > P) shared_from = P;
    // (a batched GEMM: N x [HT x P] x [P x C]; profiled with the small / streaming products: the FLOP model's executed-shapes check sees it)
    ProfScope prof(PROF_GEMM_SMALL, stream, "M%d N%d K%d epi0 act0 ctx t48x%d", N * heads * tokens, C, P, C % 128 ? 64 : 128);
    if (is16(dtype)) L4P_WITH_T16(dtype, T16, {
        // at most one workgroup per CU (a rank's query shard): four 32-key stages per barrier on a twelve-stage ring (8 tracks: 53.8 -> 34.2 us;
        // two per barrier: 39.0), else - P not a multiple of 128 - two
        if (C % 128 == 0 && (C / 128) * N <= 256 && P % 128 == 0 && knob(KNOB_TRACK_DEEP)) {
            constexpr int lds = 12 * (32 * 128 * 2 + 32 * 48 * 2);
            auto kern = t2i_ctx_mfma_kernel<T16, 48, 128, 12, 4>;
            static lds_attr_state attr_quads;
            HIP_TRY(lds_attr_once(attr_quads, kern, lds));
            hipLaunchKernelGGL(kern, dim3(C / 128, N), dim3(256), lds, stream, (const T16*)probs, (const T16*)keys, (T16*)ctx, stats, P, C,
                               heads, tokens, Rg, shared_from);
        } else if (C % 128 == 0 && (C / 128) * N <= 256 && P % 64 == 0 && knob(KNOB_TRACK_DEEP)) {
            constexpr int lds = 8 * (32 * 128 * 2 + 32 * 48 * 2);
            auto kern = t2i_ctx_mfma_kernel<T16, 48, 128, 8, 2>;
            static lds_attr_state attr_pairs;
            HIP_TRY(lds_attr_once(attr_pairs, kern, lds));
            hipLaunchKernelGGL(kern, dim3(C / 128, N), dim3(256), lds, stream, (const T16*)probs, (const T16*)keys, (T16*)ctx, stats, P, C,
                               heads, tokens, Rg, shared_from);
        } else if (C % 128 == 0) {
            constexpr int lds = 4 * (32 * 128 * 2 + 32 * 48 * 2);
            hipLaunchKernelGGL((t2i_ctx_mfma_kernel<T16, 48, 128>), dim3(C / 128, N), dim3(256), lds, stream, (const T16*)probs,
                               (const T16*)keys, (T16*)ctx, stats, P, C, heads, tokens, Rg, shared_from);
        } else {
            constexpr int lds = 4 * (32 * 64 * 2 + 32 * 48 * 2);
            hipLaunchKernelGGL((t2i_ctx_mfma_kernel<T16, 48, 64>), dim3(C / 64, N), dim3(256), lds, stream, (const T16*)probs,
                               (const T16*)keys, (T16*)ctx, stats, P, C, heads, tokens, Rg, shared_from);
        }
    }); else {
        hipLaunchKernelGGL((t2i_ctx_kernel<float, 48>), dim3((C + 255) / 256, N), dim3(256), 0, stream, (const float*)probs,
                           (const float*)keys, (float*)ctx, stats, P, C, heads, tokens, Rg, shared_from);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}
int launch_track_tokens(const float* queries, const float* labels, const float* pfeat, const float* plabel,
                        const float* gauss, const float* mask_tokens, const float* pe0, const float* pe1,
                        const float* nap, const float* fe0, const float* fe1, float* tokens, int N, int C, int T, int H,
                        int W, hipStream_t stream, int dtype, void* tokens_T) {
    ProfScope prof(PROF_TRACK, stream, "track_tokens");
    if (tokens_T && is16(dtype)) L4P_WITH_T16(dtype, T16, hipLaunchKernelGGL(track_tokens_kernel<T16>, dim3(N), dim3(256), 0, stream, queries, labels, pfeat, plabel, gauss, mask_tokens, pe0, pe1, nap, fe0, fe1, tokens, N, C, (float)T, (float)H, (float)W, (T16*)tokens_T));
    else
        hipLaunchKernelGGL(track_tokens_kernel<float>, dim3(N), dim3(256), 0, stream, queries, labels, pfeat, plabel, gauss, mask_tokens,
                           pe0, pe1, nap, fe0, fe1, tokens, N, C, (float)T, (float)H, (float)W, (float*)tokens_T);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_track_keys_init(int dtype, const float* enc, const float* hist, const float* pos, float* k32, void* kT,
                           void* kP, int N, int P, int C, int shared_from, float* k32_shared, hipStream_t stream) {
    if (C % 8 || shared_from < 0 || shared_from > P || (shared_from > 0 && !k32_shared)) {
        l4p_set_error("track_keys_init: C %% 8 != 0, or shared_from outside [0, P], or shared rows without k32_shared");
        return L4P_E_INVALID;
    }
    const long long per_q8 = (long long)P * C / 8, total8 = per_q8 * N;
    const long long shared8 = shared_from > 0 ? (long long)shared_from * C / 8 : per_q8;  // (0: no shared rows)
    if (shared_from == 0) k32_shared = nullptr;
    ProfScope prof(PROF_TRACK, stream, "track_keys_init");
    if (is16(dtype)) L4P_WITH_T16(dtype, T16, hipLaunchKernelGGL(track_keys_init_kernel<T16>, dim3(GRID1D(total8, 16384)), dim3(256), 0, stream, enc, hist,
                           pos, k32, (T16*)kT, (T16*)kP, per_q8, total8, shared8, k32_shared));
    else
        hipLaunchKernelGGL(track_keys_init_kernel<float>, dim3(GRID1D(total8, 16384)), dim3(256), 0, stream, enc, hist, pos,
                           k32, (float*)kT, (float*)kP, per_q8, total8, shared8, k32_shared);
    HIP_TRY(hipGetLastError());
    return 0;
}

// Copies the block of `bytes` bytes at base + off to the same offset of the next n - 1 groups (group g at base + g * stride):
// the later-window form of the tracker shares what only depends on the second temporal half of the keys, which is the same
// for every track of a clip (encoder feature + the learned mask token), by computing it for track 0 and copying it
__global__ void broadcast_block_kernel(char* __restrict__ base, long long off, long long chunks, long long stride, int n) {
    const long long total = chunks * (n - 1);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long g = i / chunks + 1, c = i - (g - 1) * chunks;
        const u32x4 v = *(const u32x4*)(base + off + c * 16);
        __builtin_nontemporal_store(v, (u32x4*)(base + g * stride + off + c * 16));
    }
}
int launch_broadcast_block(void* base, long long off, long long bytes, long long stride, int n, hipStream_t stream) {
    if ((bytes | off | stride) % 16 || ((size_t)base & 15)) {
        l4p_set_error("broadcast_block: offsets / sizes must be multiples of 16 bytes");
        return L4P_E_INVALID;
    }
    if (n <= 1 || bytes == 0) return 0;
    ProfScope prof(PROF_TRACK, stream, "broadcast_block");
    hipLaunchKernelGGL(broadcast_block_kernel, dim3(GRID1D(bytes / 16 * (n - 1), 16384)), dim3(256), 0, stream, (char*)base, off,
                       bytes / 16, stride, n);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_fill_rows(float* out, const float* v, long long rows, int C, long long group_rows, long long group_stride,
                     long long group_off, hipStream_t stream) {
    ProfScope prof(PROF_TRACK, stream, "fill_rows");
    hipLaunchKernelGGL(fill_rows_kernel, dim3(GRID1D(rows * (C / 4), 16384)), dim3(256), 0, stream, out, v, rows, C,
                       group_rows, group_stride, group_off);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_small_attn(int dtype, int kind, const void* q, const void* k, const void* v, void* out, int N, int P, int D,
                      int heads, hipStream_t stream) {
    const int hd = D / heads;
    const float scale = 1.0f / sqrtf((float)hd);
    if (hd * heads != D || (kind != 0 && (hd % 4 || hd > 96)) || ((kind == 1 || kind == 3) && P % 4)) {  // (P % 4: 16-byte LDS rows)
        l4p_set_error("small_attn: unsupported head geometry D=%d heads=%d", D, heads);
        return L4P_E_INVALID;
    }
    ProfScope prof(PROF_TRACK, stream, "small_attn kind%d N%d P%d D%d", kind, N, P, D);
    if (kind == 0) {  // 6 x 6 self attention
        if (is16(dtype)) L4P_WITH_T16(dtype, T16, hipLaunchKernelGGL(self_attn6_kernel<T16>, dim3(N, heads), dim3(64), 0, stream, (const T16*)q,
                               (const T16*)k, (const T16*)v, (T16*)out, D, hd, scale));
        else
            hipLaunchKernelGGL(self_attn6_kernel<float>, dim3(N, heads), dim3(64), 0, stream, (const float*)q,
                               (const float*)k, (const float*)v, (float*)out, D, hd, scale);
    } else if (kind == 1 || kind == 3) {  // tokens -> image (3: one K / V set shared by every query)
        const long long kv_stride = kind == 1 ? (long long)P * D : 0;
        const int ncg = hd / 4, nkg = 256 / ncg;
        size_t lds = (size_t)(6 * P + 6 * 96 + 256) * 4;
        const size_t need2 = (size_t)nkg * 6 * hd * 4;
        if (need2 > (size_t)6 * P * 4) lds += need2 - (size_t)6 * P * 4;
        auto kf = t2i_attn_kernel<float>;
        if (is16(dtype)) L4P_WITH_T16(dtype, T16, {
            auto kb = t2i_attn_kernel<T16>;
            HIP_TRY(hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kb, dim3(N, heads), dim3(256), lds, stream, (const T16*)q, (const T16*)k,
                               (const T16*)v, (T16*)out, P, D, hd, scale, kv_stride, (const float*)nullptr, 0ll, 0);
        }); else {
            HIP_TRY(hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kf, dim3(N, heads), dim3(256), lds, stream, (const float*)q, (const float*)k,
                               (const float*)v, (float*)out, P, D, hd, scale, kv_stride, (const float*)nullptr, 0ll, 0);
        }
    } else if (kind == 2 || kind == 4) {  // image -> tokens (4: one query set shared by every track)
        const long long q_stride = kind == 2 ? (long long)P * D : 0;
        if (256 % heads == 0 && (D * (esize_of(dtype))) % 16 == 0) {
            const int rows = 256 / heads, es = esize_of(dtype);
            const size_t lds = (((size_t)12 * D * 4 + 15) & ~(size_t)15) + (size_t)rows * D * es;
            const dim3 grid((P + rows - 1) / rows, N);
            if (is16(dtype)) L4P_WITH_T16(dtype, T16, {
                auto kern = i2t_attn_lds_kernel<T16>;
                HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, (const T16*)q, (const T16*)k, (const T16*)v,
                                   (T16*)out, P, D, hd, heads, scale, q_stride);
            }); else {
                auto kern = i2t_attn_lds_kernel<float>;
                HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, (const float*)q, (const float*)k, (const float*)v,
                                   (float*)out, P, D, hd, heads, scale, q_stride);
            }
            HIP_TRY(hipGetLastError());
            return 0;
        }
        const size_t lds = (size_t)12 * D * 4;
        const dim3 grid((P * heads + 255) / 256, N);
        if (is16(dtype)) L4P_WITH_T16(dtype, T16, hipLaunchKernelGGL(i2t_attn_kernel<T16>, grid, dim3(256), lds, stream, (const T16*)q, (const T16*)k,
                               (const T16*)v, (T16*)out, P, D, hd, heads, scale, q_stride));
        else
            hipLaunchKernelGGL(i2t_attn_kernel<float>, grid, dim3(256), lds, stream, (const float*)q, (const float*)k,
                               (const float*)v, (float*)out, P, D, hd, heads, scale, q_stride);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// LDS-staged form (VT threads, one voxel each; 45 KB of LDS at bf16, three workgroups per CU so one's copy overlaps
// another's arithmetic): the workgroup copies VT voxels x Cc channels (one contiguous span of `up`) into LDS with coalesced
// 16-byte LDS-DMA and each thread then walks its own voxel row from LDS.  The direct form above reads 16 bytes per lane at
// a Cc*sizeof(T) stride (64 distinct lines per load instruction): 1.1 TB/s measured, this one streams.  Same summation order.
template <typename T, int VT>
__global__ __launch_bounds__(VT) void mask_product_lds_kernel(const T* __restrict__ up, const float* __restrict__ hyper,
                                                               float* __restrict__ masks, long long vox, int Cc) {
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* hs = (float*)smem;                 // [3][Cc]
    char* tile = smem + ((3 * Cc * 4 + 15) & ~15);  // [VT][Cc] T
    const int n = blockIdx.y, tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long p0 = (long long)blockIdx.x * VT;
    const int nvox = (int)(vox - p0 < VT ? vox - p0 : VT);
    const int chunks = nvox * Cc * (int)sizeof(T) / 16;
    const char* src = (const char*)(up + ((long long)n * vox + p0) * Cc);
    for (int c = 0; c < chunks; c += VT)
        if (c + tid < chunks)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + (long long)(c + tid) * 16), (lptr_t)(tile + (c + wave * 64) * 16), 16, 0, 0);
    for (int i = tid; i < 3 * Cc; i += VT) hs[i] = hyper[(long long)n * 3 * Cc + i];
    __syncthreads();
    if (tid < nvox) {
        const T* xp = (const T*)tile + (long long)tid * Cc;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int c0 = 0; c0 < Cc; c0 += 8) {
            float x[8];
            Vec8<T>::load(xp + c0, x);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                a0 += x[e] * hs[c0 + e];
                a1 += x[e] * hs[Cc + c0 + e];
                a2 += x[e] * hs[2 * Cc + c0 + e];
            }
        }
        const long long p = p0 + tid;
        masks[((long long)n * 3 + 0) * vox + p] = a0;
        masks[((long long)n * 3 + 1) * vox + p] = a1;
        masks[((long long)n * 3 + 2) * vox + p] = a2;
    }
}

int launch_mask_product(int dtype, const void* up, const float* hyper, float* masks, int N, long long vox, int Cc,
                        hipStream_t stream) {
    if (Cc % 8) {
        l4p_set_error("mask_product: C %% 8 != 0");
        return L4P_E_INVALID;
    }
    ProfScope prof(PROF_TRACK, stream, "mask_product");
    const int es = esize_of(dtype);
    const int VT = 128;
    const size_t lds_tile = ((size_t)3 * Cc * 4 + 15) / 16 * 16 + (size_t)VT * Cc * es;
    if (lds_tile <= 160 * 1024) {
        const dim3 grid((unsigned)((vox + VT - 1) / VT), N);
        if (is16(dtype)) L4P_WITH_T16(dtype, T16, {
            auto kern = mask_product_lds_kernel<T16, 128>;
            HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_tile));
            hipLaunchKernelGGL(kern, grid, dim3(128), lds_tile, stream, (const T16*)up, hyper, masks, vox, Cc);
        }); else {
            auto kern = mask_product_lds_kernel<float, 128>;
            HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_tile));
            hipLaunchKernelGGL(kern, grid, dim3(128), lds_tile, stream, (const float*)up, hyper, masks, vox, Cc);
        }
        HIP_TRY(hipGetLastError());
        return 0;
    }
    const dim3 grid(GRID1D(vox, 1024), N);
    const size_t lds = (size_t)3 * Cc * 4;
    if (is16(dtype)) L4P_WITH_T16(dtype, T16, hipLaunchKernelGGL(mask_product_kernel<T16>, grid, dim3(256), lds, stream, (const T16*)up, hyper, masks, vox, Cc));
    else
        hipLaunchKernelGGL(mask_product_kernel<float>, grid, dim3(256), lds, stream, (const float*)up, hyper, masks, vox, Cc);
    HIP_TRY(hipGetLastError());
    return 0;
}

// masks[n][i][t][y][x] = sum over the chunks of tap (y%2, x%2) of the L4P_EPI_MASKDOT partial sums [chunk][i][m] of row
// m = ((n*T + t)*h + y/2)*w + x/2 (see include/l4p_hip.h).  One thread per output voxel, fixed summation order.
__global__ void mask_gather_kernel(const float* __restrict__ partial, float* __restrict__ masks, int N, int T, int h, int w,
                                   int cpt) {
    const int H2 = 2 * h, W2 = 2 * w;
    const long long vox = (long long)T * H2 * W2, total = vox * N, M = (long long)N * T * h * w;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W2);
        long long r = i / W2;
        const int y = (int)(r % H2);
        r /= H2;
        const int t = (int)(r % T), n = (int)(r / T);
        const long long m = (((long long)n * T + t) * h + (y >> 1)) * w + (x >> 1);
        const int tap = (y & 1) * 2 + (x & 1);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int c = 0; c < cpt; ++c) {
            const float* pp = partial + (long long)(tap * cpt + c) * 3 * M + m;
            a0 += pp[0];
            a1 += pp[M];
            a2 += pp[2 * M];
        }
        const long long p = ((long long)t * H2 + y) * W2 + x;
        masks[((long long)n * 3 + 0) * vox + p] = a0;
        masks[((long long)n * 3 + 1) * vox + p] = a1;
        masks[((long long)n * 3 + 2) * vox + p] = a2;
    }
}

int launch_mask_gather(const float* partial, float* masks, int N, int T, int h, int w, int cpt, hipStream_t stream) {
    const long long total = (long long)N * T * 4 * h * w;
    ProfScope prof(PROF_TRACK, stream, "mask_gather");
    hipLaunchKernelGGL(mask_gather_kernel, dim3(GRID1D(total, 16384)), dim3(256), 0, stream, partial, masks, N, T, h, w, cpt);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_track_readout(const float* masks, float* traj, float* vis, float* depth, int N, int T, int h, int w, int H,
                         int W, hipStream_t stream) {
    // the 1024-thread form: few (track, frame) workgroups - a rank's query shard of the sharded long video - leave most of the chip idle
    // and the launch is one workgroup's serial walk; with many tracks the 256-thread form fills the chip just as well
    const size_t lds_wide = ((size_t)3 * h * w + w + h + 2 * (W + H) + (size_t)READOUT_RB * W) * 4;
    if (knob(KNOB_READOUT_WIDE) && W <= 256 && h * w < 65536 && N * T <= 512 && lds_wide <= 160 * 1024) {
        static lds_attr_state attr_wide;
        HIP_TRY(lds_attr_once(attr_wide, track_readout_wide_kernel, (int)lds_wide));
        ProfScope prof(PROF_TRACK, stream, "track_readout wide");
        hipLaunchKernelGGL(track_readout_wide_kernel, dim3(N * T), dim3(1024), lds_wide, stream, masks, traj, vis, depth, T, h, w, H, W);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    const size_t lds = ((size_t)3 * h * w + w + h) * 4;
    static lds_attr_state attr_done;
    HIP_TRY(lds_attr_once(attr_done, track_readout_kernel, (int)lds));
    ProfScope prof(PROF_TRACK, stream, "track_readout");
    hipLaunchKernelGGL(track_readout_kernel, dim3(N * T), dim3(256), lds, stream, masks, traj, vis, depth, T, h, w, H, W);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_track_prepare(const float* cur_q, const float* orig_q, int start, int ws, float* q_off, float* labels,
                         unsigned char* valid_t, unsigned char* valid_n, int N, hipStream_t stream) {
    ProfScope prof(PROF_TRACK, stream, "track_prepare");
    hipLaunchKernelGGL(track_prepare_kernel, dim3((N + 63) / 64), dim3(64), 0, stream, cur_q, orig_q, start, ws, q_off,
                       labels, valid_t, valid_n, N);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_track_commit(const float* w_traj, const float* w_vis, const float* w_depth, const unsigned char* valid_t,
                        const unsigned char* valid_n, float* traj, float* vis, float* depth, int T, int start, int ws,
                        int next_start, int last_window, float* cur_q, float* plabel, const float* new_pfeat, float* pfeat,
                        int* best_out, int N, int C, hipStream_t stream) {
    ProfScope prof(PROF_TRACK, stream, "track_commit");
    hipLaunchKernelGGL(track_commit_kernel, dim3((N + 63) / 64), dim3(64), 0, stream, w_traj, w_vis, w_depth, valid_t,
                       valid_n, traj, vis, depth, T, start, ws, next_start, last_window, cur_q, plabel, best_out, N);
    if (!last_window && new_pfeat && pfeat)
        hipLaunchKernelGGL(masked_rows_copy_kernel, dim3(GRID1D((long long)N * (C / 4), 4096)), dim3(256), 0, stream, pfeat,
                           new_pfeat, valid_n, N, C);
    HIP_TRY(hipGetLastError());
    return 0;
}
