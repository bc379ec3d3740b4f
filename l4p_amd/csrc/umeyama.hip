// Joint depth + camera seam alignment on the GPU (reference aligner.py:121-265, geometry_utils.py:13-53):
// overlap depth/pose pairs -> sampled world-space point maps -> RANSAC similarity (Umeyama) -> apply.
// The reference does this on the CPU with numpy + skimage.measure.ransac (randomised, unpinned version);
// this is a deterministic counter-based-RNG re-statement of the same estimator, validated against synthetic
// ground truth (tests/test_umeyama_gpu.py), not against the reference ("parity unpinned").
#include "common.hpp"

__device__ __forceinline__ unsigned hash_u32(unsigned x) {  // PCG-style integer hash (counter-based RNG)
    x = x * 747796405u + 2891336453u;
    const unsigned w = ((x >> ((x >> 28u) + 4u)) ^ x) * 277803737u;
    return (w >> 22u) ^ w;
}

// -------------------------------------------------------------------------------------------------
// EXACT q-quantile of n non-negative floats with torch.quantile's linear interpolation (aligner.py:187:
// torch.quantile(depth, 0.98)): pos = q (n - 1) in float, result = lerp(x_(lo), x_(lo+1), pos - lo) over the order
// statistics.  Non-negative floats order like their bit patterns, so x_(lo) is found by a three-pass radix select on the
// bits (11 + 11 + 10), one histogram + one pick per pass; x_(lo+1) is x_(lo) again when it has duplicates covering rank
// lo + 1, else the smallest element above it (one more pass).  Negative inputs (never produced: depth = exp(.)) are
// treated as 0.
// ws (uint): [0] prefix bits found so far, [1] rank still to skip inside the prefix, [2] count of x == x_(lo) after the
// last pass, [3] bits of min{x > x_(lo)}, [4 ..] 2048 histogram bins.
// -------------------------------------------------------------------------------------------------
#define QSEL_BINS 2048
// order-preserving map float -> unsigned for every finite value of either sign (negative: all bits flipped, non-negative:
// sign bit set; -0.0 lands just below +0.0 and both map back to a zero), and its inverse
__device__ __forceinline__ unsigned qsel_key(float v) {
    const unsigned b = __float_as_uint(v);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float qsel_unkey(unsigned k) { return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu)); }

// Depth values crowd into a handful of exponent bins, so the histogram is built per workgroup in LDS first (same-key lanes
// of a wave are merged with a ballot before they touch the bin: one LDS atomic per distinct key of the wave for the first
// few keys) and only the non-empty bins reach the global counters: 400 k global atomics on ~4 addresses cost 0.2 ms a pass.
template <int SHIFT, int BITS, unsigned PMASK>
__global__ __launch_bounds__(256) void qsel_hist_kernel(const float* __restrict__ x, long long n, unsigned* __restrict__ ws) {
    __shared__ unsigned h[1 << BITS];
    for (int i = threadIdx.x; i < (1 << BITS); i += 256) h[i] = 0;
    __syncthreads();
    const unsigned prefix = ws[0];
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i0 = blockIdx.x * (long long)blockDim.x; i0 < n; i0 += stride) {  // (wave-uniform trip count)
        const long long i = i0 + threadIdx.x;
        const unsigned k = i < n ? qsel_key(x[i]) : 0u;
        bool todo = i < n && (k & PMASK) == prefix;
        const unsigned bin = (k >> SHIFT) & ((1u << BITS) - 1);
#pragma unroll 1
        for (int round = 0; round < 4; ++round) {
            const unsigned long long act = __ballot(todo);
            if (!act) break;
            const int leader = __ffsll((long long)act) - 1;
            const unsigned lb = (unsigned)__shfl((int)bin, leader);
            const bool mine = todo && bin == lb;
            const unsigned long long grp = __ballot(mine);
            if ((int)(threadIdx.x & 63) == leader) atomicAdd(&h[lb], (unsigned)__popcll(grp));
            todo = todo && !mine;
        }
        if (todo) atomicAdd(&h[bin], 1u);  // many distinct keys in the wave: no contention to speak of
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (1 << BITS); i += 256)
        if (h[i]) atomicAdd(&ws[4 + i], h[i]);
}
// one workgroup: the bin that holds the wanted rank, by a block-wide prefix sum over the bins (a single thread walking 2048
// global counters one dependent load at a time cost 0.4 ms a pass), then the bins are cleared for the next pass
template <int SHIFT, int BITS, bool LAST>
__global__ __launch_bounds__(256) void qsel_pick_kernel(unsigned* __restrict__ ws) {
    constexpr int PER = (1 << BITS) / 256;
    __shared__ unsigned scan[256];
    const int tid = threadIdx.x;
    unsigned c[PER], sum = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        c[j] = ws[4 + tid * PER + j];
        sum += c[j];
    }
    scan[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {  // inclusive Hillis-Steele scan
        const unsigned v = tid >= o ? scan[tid - o] : 0u;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    const unsigned rank = ws[1], excl = scan[tid] - sum;
    __syncthreads();  // (everyone has read ws[1] before the owner rewrites it)
    if (sum > 0 && rank >= excl && rank < excl + sum) {
        unsigned r = rank - excl;
        int b = 0;
#pragma unroll
        for (int j = 0; j < PER - 1; ++j)
            if (b == j && r >= c[j]) {
                r -= c[j];
                b = j + 1;
            }
        ws[0] |= (unsigned)(tid * PER + b) << SHIFT;
        ws[1] = r;
        if (LAST) {
            unsigned cb = c[0];
#pragma unroll
            for (int j = 1; j < PER; ++j) cb = b == j ? c[j] : cb;
            ws[2] = cb;
        }
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) ws[4 + tid * PER + j] = 0;
}
__global__ void qsel_next_kernel(const float* __restrict__ x, long long n, unsigned* __restrict__ ws) {
    const unsigned v = ws[0];
    unsigned mn = 0xFFFFFFFFu;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const unsigned k = qsel_key(x[i]);
        if (k > v && k < mn) mn = k;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned other = __shfl_xor(mn, o);
        mn = other < mn ? other : mn;
    }
    if ((threadIdx.x & 63) == 0 && mn != 0xFFFFFFFFu) atomicMin(&ws[3], mn);
}
__global__ void qsel_finish_kernel(const unsigned* __restrict__ ws, float weight, float* __restrict__ out) {
    if (threadIdx.x != 0) return;
    const float a = qsel_unkey(ws[0]);
    // rank lo sits at offset ws[1] inside the ws[2] copies of a: rank lo + 1 is another copy unless it was the last one
    const float b = (ws[1] + 1 < ws[2] || ws[3] == 0xFFFFFFFFu) ? a : qsel_unkey(ws[3]);
    // ATen lerp: the weight < 0.5 form and its mirror (aten/src/ATen/native/Lerp.h)
    const float d = b - a;
    out[0] = weight < 0.5f ? __builtin_fmaf(weight, d, a) : b - d * (1.f - weight);
}

// -------------------------------------------------------------------------------------------------
// Sampled point maps (generate_point_map geometry_utils.py:13-53 on the frames ::step of the overlap):
// X = world_T_cam * [depth * K^-1 [x, y, 1]; 1].  One sample per thread; pixel chosen by a hash inside
// each stride-`ratio` cell (the reference takes a random 10 % permutation subset, aligner.py:214-220).
// depth: [F][H*W]; K, P: [F][16] row-major 4x4 (pixel intrinsics / world_T_cam); out: [F*spf][3].
// -------------------------------------------------------------------------------------------------
__global__ void pointmap_kernel(const float* __restrict__ depth, const float* __restrict__ K, const float* __restrict__ P,
                                float* __restrict__ out, int F, int H, int W, int ratio, unsigned seed, int spf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F * spf) return;
    const int f = i / spf, j = i % spf;
    int pix = j * ratio + (int)(hash_u32(seed ^ (unsigned)j * 2654435761u) % (unsigned)ratio);
    if (pix >= H * W) pix = H * W - 1;
    const float x = (float)(pix % W), y = (float)(pix / W);
    const float* k = K + f * 16;
    // inverse of the upper-left 3x3 (general, as torch.inverse does)
    const float a = k[0], b = k[1], c = k[2], d = k[4], e = k[5], g = k[6], h = k[8], l = k[9], m = k[10];
    const float det = a * (e * m - g * l) - b * (d * m - g * h) + c * (d * l - e * h);
    const float id = 1.f / det;
    const float i00 = (e * m - g * l) * id, i01 = (c * l - b * m) * id, i02 = (b * g - c * e) * id;
    const float i10 = (g * h - d * m) * id, i11 = (a * m - c * h) * id, i12 = (c * d - a * g) * id;
    const float i20 = (d * l - e * h) * id, i21 = (b * h - a * l) * id, i22 = (a * e - b * d) * id;
    const float z = depth[(long long)f * H * W + pix];
    const float cx = (i00 * x + i01 * y + i02) * z, cy = (i10 * x + i11 * y + i12) * z, cz = (i20 * x + i21 * y + i22) * z;
    const float* p = P + f * 16;
    out[i * 3 + 0] = p[0] * cx + p[1] * cy + p[2] * cz + p[3];
    out[i * 3 + 1] = p[4] * cx + p[5] * cy + p[6] * cz + p[7];
    out[i * 3 + 2] = p[8] * cx + p[9] * cy + p[10] * cz + p[11];
}

// -------------------------------------------------------------------------------------------------
// Umeyama similarity from accumulated moments (skimage.transform._geometric._umeyama):
// dst ~ s R src + t.  sums: n, mean_s[3], mean_d[3], cov[3][3] = E[(d-md)(s-ms)^T], var_s.
// -------------------------------------------------------------------------------------------------
__device__ void jacobi3(double A[3][3], double V[3][3]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) V[i][j] = i == j;
    for (int sweep = 0; sweep < 30; ++sweep) {
        if (fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]) < 1e-300) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (fabs(A[p][q]) < 1e-300) continue;
                const double th = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {
                    const double x = A[k][p], y = A[k][q];
                    A[k][p] = c * x - s * y;
                    A[k][q] = s * x + c * y;
                }
                for (int k = 0; k < 3; ++k) {
                    const double x = A[p][k], y = A[q][k];
                    A[p][k] = c * x - s * y;
                    A[q][k] = s * x + c * y;
                }
                for (int k = 0; k < 3; ++k) {
                    const double x = V[k][p], y = V[k][q];
                    V[k][p] = c * x - s * y;
                    V[k][q] = s * x + c * y;
                }
            }
    }
}

// model: [0..8] = s*R row-major, [9..11] = t, [12] = s
__device__ void umeyama_from_moments(const double ms[3], const double md[3], const double cov[3][3], double var_s,
                                     float* model) {
    double AtA[3][3], V[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) AtA[i][j] = cov[0][i] * cov[0][j] + cov[1][i] * cov[1][j] + cov[2][i] * cov[2][j];
    jacobi3(AtA, V);
    int o[3] = {0, 1, 2};
    for (int a = 0; a < 2; ++a)
        for (int b = a + 1; b < 3; ++b)
            if (AtA[o[b]][o[b]] > AtA[o[a]][o[a]]) {
                const int t = o[a];
                o[a] = o[b];
                o[b] = t;
            }
    double v1[3], v2[3], v3[3], u1[3], u2[3], u3[3];
    for (int i = 0; i < 3; ++i) {
        v1[i] = V[i][o[0]];
        v2[i] = V[i][o[1]];
    }
    v3[0] = v1[1] * v2[2] - v1[2] * v2[1];
    v3[1] = v1[2] * v2[0] - v1[0] * v2[2];
    v3[2] = v1[0] * v2[1] - v1[1] * v2[0];
    for (int i = 0; i < 3; ++i) {
        u1[i] = cov[i][0] * v1[0] + cov[i][1] * v1[1] + cov[i][2] * v1[2];
        u2[i] = cov[i][0] * v2[0] + cov[i][1] * v2[1] + cov[i][2] * v2[2];
    }
    const double s1 = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
    for (int i = 0; i < 3; ++i) u1[i] /= fmax(s1, 1e-300);
    const double dp = u1[0] * u2[0] + u1[1] * u2[1] + u1[2] * u2[2];
    for (int i = 0; i < 3; ++i) u2[i] -= dp * u1[i];
    const double s2 = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
    for (int i = 0; i < 3; ++i) u2[i] /= fmax(s2, 1e-300);
    u3[0] = u1[1] * u2[2] - u1[2] * u2[1];
    u3[1] = u1[2] * u2[0] - u1[0] * u2[2];
    u3[2] = u1[0] * u2[1] - u1[1] * u2[0];
    // sigma3 with the reflection sign folded in: u3^T cov v3 (negative when det(cov) < 0)
    double s3 = 0;
    for (int i = 0; i < 3; ++i) s3 += u3[i] * (cov[i][0] * v3[0] + cov[i][1] * v3[1] + cov[i][2] * v3[2]);
    const double scale = var_s > 0 ? (s1 + s2 + s3) / var_s : 1.0;
    double R[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[i][j] = u1[i] * v1[j] + u2[i] * v2[j] + u3[i] * v3[j];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) model[i * 3 + j] = (float)(scale * R[i][j]);
        model[9 + i] = (float)(md[i] - scale * (R[i][0] * ms[0] + R[i][1] * ms[1] + R[i][2] * ms[2]));
    }
    model[12] = (float)scale;
}

__device__ __forceinline__ float residual(const float* m, const float* s, const float* d) {
    const float rx = m[0] * s[0] + m[1] * s[1] + m[2] * s[2] + m[9] - d[0];
    const float ry = m[3] * s[0] + m[4] * s[1] + m[5] * s[2] + m[10] - d[1];
    const float rz = m[6] * s[0] + m[7] * s[1] + m[8] * s[2] + m[11] - d[2];
    return sqrtf(rx * rx + ry * ry + rz * rz);
}

// One workgroup per RANSAC trial: minimal-sample model, then inlier count + residual sum over all points.
// scores: [trials][2] (inliers as float, residual sum), models: [trials][13]
__global__ __launch_bounds__(256) void ransac_trials_kernel(const float* __restrict__ src, const float* __restrict__ dst,
                                                            int n, const float* __restrict__ q98, float thr_rel,
                                                            int min_samples, unsigned seed, float* __restrict__ scores,
                                                            float* __restrict__ models) {
    __shared__ float model[13];
    __shared__ float red[4][2];
    const int trial = blockIdx.x;
    if (threadIdx.x == 0) {
        double ms[3] = {0, 0, 0}, md[3] = {0, 0, 0}, cov[3][3] = {{0}}, var = 0;
        int idx[32];
        for (int j = 0; j < min_samples; ++j) idx[j] = (int)(hash_u32(seed + 7919u * trial + 104729u * j) % (unsigned)n);
        for (int j = 0; j < min_samples; ++j)
            for (int k = 0; k < 3; ++k) {
                ms[k] += src[idx[j] * 3 + k];
                md[k] += dst[idx[j] * 3 + k];
            }
        for (int k = 0; k < 3; ++k) {
            ms[k] /= min_samples;
            md[k] /= min_samples;
        }
        for (int j = 0; j < min_samples; ++j) {
            double ds[3], dd[3];
            for (int k = 0; k < 3; ++k) {
                ds[k] = src[idx[j] * 3 + k] - ms[k];
                dd[k] = dst[idx[j] * 3 + k] - md[k];
                var += ds[k] * ds[k];
            }
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) cov[a][b] += dd[a] * ds[b];
        }
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) cov[a][b] /= min_samples;
        var /= min_samples;
        umeyama_from_moments(ms, md, cov, var, model);
    }
    __syncthreads();
    const float thr = q98[0] * thr_rel;
    float cnt = 0.f, rs = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float r = residual(model, src + i * 3, dst + i * 3);
        if (r < thr) {
            cnt += 1.f;
            rs += r;
        }
    }
    cnt = wave_sum(cnt);
    rs = wave_sum(rs);
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6][0] = cnt;
        red[threadIdx.x >> 6][1] = rs;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        scores[trial * 2 + 0] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
        scores[trial * 2 + 1] = red[0][1] + red[1][1] + red[2][1] + red[3][1];
        for (int k = 0; k < 13; ++k) models[trial * 13 + k] = model[k];
    }
}

// Best trial (most inliers, ties -> smaller residual sum, as skimage), then re-estimate on its inliers.
// out: [16] row-major 4x4 similarity T = [sR | t; 0 0 0 1], out[16] = s, out[17] = inlier count
__global__ __launch_bounds__(256) void ransac_final_kernel(const float* __restrict__ src, const float* __restrict__ dst, int n,
                                                           const float* __restrict__ q98, float thr_rel, int trials,
                                                           const float* __restrict__ scores, const float* __restrict__ models,
                                                           float* __restrict__ out) {
    __shared__ float model[13];
    __shared__ double red[4][16];
    __shared__ double mom[16];
    if (threadIdx.x == 0) {
        int best = 0;
        for (int t = 1; t < trials; ++t)
            if (scores[t * 2] > scores[best * 2] || (scores[t * 2] == scores[best * 2] && scores[t * 2 + 1] < scores[best * 2 + 1]))
                best = t;
        for (int k = 0; k < 13; ++k) model[k] = models[best * 13 + k];
    }
    __syncthreads();
    const float thr = q98[0] * thr_rel;
    // pass 1: count + means over the inliers
    double v[16];
    for (int k = 0; k < 16; ++k) v[k] = 0;
    for (int i = threadIdx.x; i < n; i += 256)
        if (residual(model, src + i * 3, dst + i * 3) < thr) {
            v[0] += 1;
            for (int k = 0; k < 3; ++k) {
                v[1 + k] += src[i * 3 + k];
                v[4 + k] += dst[i * 3 + k];
            }
        }
    for (int k = 0; k < 7; ++k)
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 7; ++k) red[threadIdx.x >> 6][k] = v[k];
    __syncthreads();
    if (threadIdx.x == 0)
        for (int k = 0; k < 7; ++k) mom[k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
    __syncthreads();
    const double cntd = mom[0] > 0 ? mom[0] : 1.0;
    const double ms[3] = {mom[1] / cntd, mom[2] / cntd, mom[3] / cntd}, md[3] = {mom[4] / cntd, mom[5] / cntd, mom[6] / cntd};
    // pass 2: covariance + source variance
    for (int k = 0; k < 16; ++k) v[k] = 0;
    for (int i = threadIdx.x; i < n; i += 256)
        if (residual(model, src + i * 3, dst + i * 3) < thr) {
            double ds[3], dd[3];
            for (int k = 0; k < 3; ++k) {
                ds[k] = src[i * 3 + k] - ms[k];
                dd[k] = dst[i * 3 + k] - md[k];
                v[9] += ds[k] * ds[k];
            }
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) v[a * 3 + b] += dd[a] * ds[b];
        }
    for (int k = 0; k < 10; ++k)
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 10; ++k) red[threadIdx.x >> 6][k] = v[k];
    __syncthreads();
    if (threadIdx.x == 0) {
        double cov[3][3];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) cov[a][b] = (red[0][a * 3 + b] + red[1][a * 3 + b] + red[2][a * 3 + b] + red[3][a * 3 + b]) / cntd;
        const double var = (red[0][9] + red[1][9] + red[2][9] + red[3][9]) / cntd;
        float fin[13];
        if (mom[0] >= 3)
            umeyama_from_moments(ms, md, cov, var, fin);
        else
            for (int k = 0; k < 13; ++k) fin[k] = model[k];
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) out[i * 4 + j] = fin[i * 3 + j];
            out[i * 4 + 3] = fin[9 + i];
        }
        out[12] = out[13] = out[14] = 0.f;
        out[15] = 1.f;
        out[16] = fin[12];
        out[17] = (float)mom[0];
    }
}

// apply (aligner.py:239-265): pose <- T pose, rotation block / s ; pose: [16][T] row-major 4x4 per frame
__global__ void similarity_apply_pose_kernel(const float* __restrict__ sim, float* __restrict__ pose, int T) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    float P[16], R[16];
    for (int k = 0; k < 16; ++k) P[k] = pose[(long long)k * T + t];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float a = 0.f;
            for (int k = 0; k < 4; ++k) a += sim[i * 4 + k] * P[k * 4 + j];
            R[i * 4 + j] = a;
        }
    const float is = 1.f / sim[16];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[i * 4 + j] *= is;
    for (int k = 0; k < 16; ++k) pose[(long long)k * T + t] = R[k];
}

// prefix composition of per-seam similarities (the seam-local exchange of the sharded long video, l4p_amd/parallel.py): applying
// (sim_a, s_a) after (sim_r, s_r) is applying (sim_a sim_r, s_a s_r) - the rotation block of a pose is divided by the scale after
// each product, so it stays a rotation and the scales multiply.  rel: [n][B][18] seam records (seam i aligns window i + 1 to RAW
// window i), out: [n + 1][B][18]: out[0] = identity, out[w] = out[w - 1] o rel[w - 1] = the transform of window w into window 0's
// frame.  One thread per clip walks the seams in order (double accumulation, one rounding per record).
__global__ void similarity_prefix_kernel(const float* __restrict__ rel, float* __restrict__ out, int n, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double A[16], s = 1.0;
    for (int k = 0; k < 16; ++k) A[k] = (k % 5 == 0) ? 1.0 : 0.0;
    float* o = out + (long long)b * 18;
    for (int k = 0; k < 16; ++k) o[k] = (float)A[k];
    o[16] = 1.f, o[17] = 0.f;
    for (int w = 0; w < n; ++w) {
        const float* r = rel + ((long long)w * B + b) * 18;
        double C[16];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                double a = 0.0;
                for (int k = 0; k < 4; ++k) a += A[i * 4 + k] * (double)r[k * 4 + j];
                C[i * 4 + j] = a;
            }
        for (int k = 0; k < 16; ++k) A[k] = C[k];
        s *= (double)r[16];
        o = out + ((long long)(w + 1) * B + b) * 18;
        for (int k = 0; k < 16; ++k) o[k] = (float)A[k];
        o[16] = (float)s, o[17] = r[17];
    }
}

__global__ void scale_by_device_scalar_kernel(float* __restrict__ x, long long n, const float* __restrict__ s) {
    const float v = s[0];
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) x[i] *= v;
}

extern "C" {

// order statistic `lo` (0-based) of x[0..n) and the next one, blended with `weight` (ATen's lerp); weight 0: the statistic itself
static int qsel_launch(hipStream_t s, const float* x, long long n, unsigned lo, float weight, unsigned* ws, float* out);

/* exact q-quantile (torch.quantile, linear interpolation) of n finite floats. ws: >= L4P_QUANTILE_WS_UINTS uints. */
int l4p_quantile(l4p_stream s_, const float* x, long long n, float q, unsigned* ws, float* out) {
    hipStream_t s = (hipStream_t)s_;
    if (n < 1 || n > 0x7FFFFFFFll || !(q >= 0.f && q <= 1.f)) {
        l4p_set_error("l4p_quantile: need 1 <= n < 2^31 and 0 <= q <= 1 (n=%lld q=%g)", n, (double)q);
        return L4P_E_INVALID;
    }
    ProfScope prof(PROF_ELEMENTWISE, s, "l4p_quantile");
    // torch.quantile computes the rank in the input dtype: pos = q * (n - 1) rounded to float
    const float pos = q * (float)(n - 1);
    const float lo_f = floorf(pos);
    return qsel_launch(s, x, n, (unsigned)lo_f, pos - lo_f, ws, out);
}

/* the order statistic of rank `rank` (0-based, ascending) of n finite floats, exact.  torch.median(x) is rank (n - 1) / 2. */
int l4p_select_rank(l4p_stream s_, const float* x, long long n, long long rank, unsigned* ws, float* out) {
    hipStream_t s = (hipStream_t)s_;
    if (n < 1 || n > 0x7FFFFFFFll || rank < 0 || rank >= n) {
        l4p_set_error("l4p_select_rank: need 1 <= n < 2^31 and 0 <= rank < n (n=%lld rank=%lld)", n, rank);
        return L4P_E_INVALID;
    }
    ProfScope prof(PROF_ELEMENTWISE, s, "l4p_select_rank");
    return qsel_launch(s, x, n, (unsigned)rank, 0.f, ws, out);
}

// LinearAligner(method="median") (aligner.py:96-107): ratios f(target) / (f(pred) + 1e-8) in float, f = safe_inverse or identity
__global__ void ratio_kernel(const float* __restrict__ pred, const float* __restrict__ tgt, long long n, int inverse,
                             float* __restrict__ r) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float p = pred[i], t = tgt[i];
        const float af = inverse ? (p > 0.f ? 1.0f / p : 0.f) : p, bf = inverse ? (t > 0.f ? 1.0f / t : 0.f) : t;
        r[i] = bf / (af + 1e-8f);
    }
}
/* sol[0] = torch.median(ratios) = their order statistic (n - 1) / 2 (the LOWER median, as torch returns it), sol[1] = 0.
 * ratios: n floats of scratch; ws >= L4P_QUANTILE_WS_UINTS uints. */
int l4p_ratio_median_solve(l4p_stream s_, const float* pred, const float* target, long long n, int inverse, float* ratios,
                           unsigned* ws, float* sol) {
    hipStream_t s = (hipStream_t)s_;
    if (n < 1 || n > 0x7FFFFFFFll) {
        l4p_set_error("l4p_ratio_median_solve: need 1 <= n < 2^31 (n=%lld)", n);
        return L4P_E_INVALID;
    }
    ProfScope prof(PROF_ELEMENTWISE, s, "ratio_median");
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(ratio_kernel, dim3(grid), dim3(256), 0, s, pred, target, n, inverse & 1, ratios);
    HIP_TRY(hipMemsetAsync(sol + 1, 0, sizeof(float), s));
    return qsel_launch(s, ratios, n, (unsigned)((n - 1) / 2), 0.f, ws, sol);
}

static int qsel_launch(hipStream_t s, const float* x, long long n, unsigned lo, float weight, unsigned* ws, float* out) {
    unsigned init[4] = {0u, lo, 0u, 0xFFFFFFFFu};
    HIP_TRY(hipMemsetAsync(ws, 0, (4 + QSEL_BINS) * sizeof(unsigned), s));
    // (init is tiny and lives on the host stack: three 4-byte memsets keep the call free of host -> device copies)
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)(ws + 1), (int)init[1], 1, s));
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)(ws + 3), (int)init[3], 1, s));
    const int grid = (int)((n + 255) / 256 < 256 ? (n + 255) / 256 : 256);
    hipLaunchKernelGGL((qsel_hist_kernel<21, 11, 0u>), dim3(grid), dim3(256), 0, s, x, n, ws);
    hipLaunchKernelGGL((qsel_pick_kernel<21, 11, false>), dim3(1), dim3(256), 0, s, ws);
    hipLaunchKernelGGL((qsel_hist_kernel<10, 11, 0xFFE00000u>), dim3(grid), dim3(256), 0, s, x, n, ws);
    hipLaunchKernelGGL((qsel_pick_kernel<10, 11, false>), dim3(1), dim3(256), 0, s, ws);
    hipLaunchKernelGGL((qsel_hist_kernel<0, 10, 0xFFFFFC00u>), dim3(grid), dim3(256), 0, s, x, n, ws);
    hipLaunchKernelGGL((qsel_pick_kernel<0, 10, true>), dim3(1), dim3(256), 0, s, ws);
    hipLaunchKernelGGL(qsel_next_kernel, dim3(grid), dim3(256), 0, s, x, n, ws);
    hipLaunchKernelGGL(qsel_finish_kernel, dim3(1), dim3(64), 0, s, ws, weight, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

int l4p_point_map_samples(l4p_stream s_, const float* depth, const float* K, const float* P, float* out, int F, int H, int W,
                          int ratio, unsigned seed) {
    hipStream_t s = (hipStream_t)s_;
    const int spf = (H * W) / ratio;
    ProfScope prof(PROF_ELEMENTWISE, s, "l4p_point_map_samples");
    hipLaunchKernelGGL(pointmap_kernel, dim3((F * spf + 255) / 256), dim3(256), 0, s, depth, K, P, out, F, H, W, ratio, seed, spf);
    HIP_TRY(hipGetLastError());
    return 0;
}

/* RANSAC similarity dst ~ s R src + t over n correspondences.  thr = q98[0] * thr_rel.
 * ws: float[trials * 15].  out: float[18] = 4x4 T (row-major), s, inlier count. */
int l4p_similarity_ransac(l4p_stream s_, const float* src, const float* dst, int n, const float* q98, float thr_rel,
                          int trials, int min_samples, unsigned seed, float* ws, float* out) {
    hipStream_t s = (hipStream_t)s_;
    if (n < min_samples || min_samples < 3 || min_samples > 32 || trials < 1) {
        l4p_set_error("similarity_ransac: bad arguments n=%d min_samples=%d trials=%d", n, min_samples, trials);
        return L4P_E_INVALID;
    }
    ProfScope prof(PROF_ELEMENTWISE, s, "l4p_similarity_ransac");
    float* scores = ws;
    float* models = ws + 2 * trials;
    hipLaunchKernelGGL(ransac_trials_kernel, dim3(trials), dim3(256), 0, s, src, dst, n, q98, thr_rel, min_samples, seed, scores,
                       models);
    hipLaunchKernelGGL(ransac_final_kernel, dim3(1), dim3(256), 0, s, src, dst, n, q98, thr_rel, trials, scores, models, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

/* out[0] = identity, out[w] = out[w - 1] o rel[w - 1]: rel [n][B][18], out [n + 1][B][18] (see similarity_prefix_kernel) */
int l4p_similarity_prefix(l4p_stream s_, const float* rel, float* out, int n, int B) {
    hipStream_t s = (hipStream_t)s_;
    if (n < 0 || B < 1) {
        l4p_set_error("l4p_similarity_prefix: need n >= 0 seams and B >= 1 clips (n=%d B=%d)", n, B);
        return L4P_E_INVALID;
    }
    ProfScope prof(PROF_ELEMENTWISE, s, "l4p_similarity_prefix");
    hipLaunchKernelGGL(similarity_prefix_kernel, dim3((B + 63) / 64), dim3(64), 0, s, rel, out, n, B);
    HIP_TRY(hipGetLastError());
    return 0;
}

/* apply: pose [16][T] <- T pose with the rotation block divided by s; depth (n floats) *= s */
int l4p_similarity_apply(l4p_stream s_, const float* sim, float* pose, int T, float* depth, long long n) {
    hipStream_t s = (hipStream_t)s_;
    ProfScope prof(PROF_ELEMENTWISE, s, "l4p_similarity_apply");
    hipLaunchKernelGGL(similarity_apply_pose_kernel, dim3((T + 63) / 64), dim3(64), 0, s, sim, pose, T);
    if (depth && n > 0) {
        const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
        hipLaunchKernelGGL(scale_by_device_scalar_kernel, dim3(grid), dim3(256), 0, s, depth, n, sim + 16);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}
}
