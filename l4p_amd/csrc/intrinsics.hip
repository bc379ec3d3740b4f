// Camera intrinsics from a Pluecker ray map (reference compute_optimal_rotation_intrinsics,
// geometry_utils.py:409-456, used by rays_to_cameras_and_fixed_per_frame_intrinsics :493-579 when
// use_intrinsics=False): homography between the identity-K pixel rays and the predicted directions,
// H^-1 = K R, RQ decomposition.  The reference calls cv2.findHomography(RANSAC, 0.2) + cv2.RQDecomp3x3 on
// the CPU (randomised, unpinned version).  This is a deterministic restatement: Hartley-normalised DLT,
// iteratively re-estimated on its own consensus set (reprojection error < thr), then RQ with a positive
// diagonal.  Validated by recovering known intrinsics from synthetic ray maps ("parity unpinned").
#include "common.hpp"

// Eigenvector of the smallest eigenvalue of a symmetric positive semi-definite 9 x 9 matrix (the DLT normal matrix
// A^T A; its null / near-null vector is the homography), by shifted inverse iteration: Cholesky of M + eps I in place,
// then a few rounds of h <- normalise((M + eps I)^-1 h).  The smallest eigenvalue of a DLT system is at the noise level
// (exactly 0 for a minimal 4-point sample) while the next one is O(1) after Hartley normalisation, so every round
// gains orders of magnitude; 5 rounds are far beyond double precision for any usable sample.  All loops are fully
// unrolled: the 45 packed entries live in registers (the cyclic Jacobi sweep this replaces kept two 9 x 9 arrays in
// scratch memory and ran a fixed 60 sweeps: 4 ms per clip).
// m: lower triangle packed row-major, m[i*(i+1)/2 + j] (j <= i); destroyed.  h: out, unit length.
__device__ __forceinline__ void smallest_eigvec9(double (&m)[45], double (&h)[9]) {
    double tr = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) tr += m[i * (i + 1) / 2 + i];
    const double eps = tr * (1e-13 / 9.0) + 1e-300;
#pragma unroll
    for (int i = 0; i < 9; ++i) m[i * (i + 1) / 2 + i] += eps;
    double dinv[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        double sdiag = m[j * (j + 1) / 2 + j];
#pragma unroll
        for (int k = 0; k < j; ++k) sdiag -= m[j * (j + 1) / 2 + k] * m[j * (j + 1) / 2 + k];
        const double d = sqrt(fmax(sdiag, eps * 1e-3));
        dinv[j] = 1.0 / d;
        m[j * (j + 1) / 2 + j] = d;
#pragma unroll
        for (int i = j + 1; i < 9; ++i) {
            double v = m[i * (i + 1) / 2 + j];
#pragma unroll
            for (int k = 0; k < j; ++k) v -= m[i * (i + 1) / 2 + k] * m[j * (j + 1) / 2 + k];
            m[i * (i + 1) / 2 + j] = v * dinv[j];
        }
    }
    const double h0[9] = {0.3, -0.5, 0.7, 0.2, -0.9, 0.4, 0.6, -0.1, 0.8};
#pragma unroll
    for (int i = 0; i < 9; ++i) h[i] = h0[i];
#pragma unroll 1
    for (int it = 0; it < 5; ++it) {
#pragma unroll
        for (int i = 0; i < 9; ++i) {  // L y = h
            double v = h[i];
#pragma unroll
            for (int k = 0; k < i; ++k) v -= m[i * (i + 1) / 2 + k] * h[k];
            h[i] = v * dinv[i];
        }
#pragma unroll
        for (int i = 8; i >= 0; --i) {  // L^T x = y
            double v = h[i];
#pragma unroll
            for (int k = i + 1; k < 9; ++k) v -= m[k * (k + 1) / 2 + i] * h[k];
            h[i] = v * dinv[i];
        }
        double n2 = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) n2 += h[i] * h[i];
        const double inv = 1.0 / sqrt(fmax(n2, 1e-300));
#pragma unroll
        for (int i = 0; i < 9; ++i) h[i] *= inv;
    }
}

__device__ void inv3(const double* m, double* o) {
    const double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
    const double id = 1.0 / det;
    o[0] = (m[4] * m[8] - m[5] * m[7]) * id;
    o[1] = (m[2] * m[7] - m[1] * m[8]) * id;
    o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = (m[5] * m[6] - m[3] * m[8]) * id;
    o[4] = (m[0] * m[8] - m[2] * m[6]) * id;
    o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = (m[3] * m[7] - m[4] * m[6]) * id;
    o[7] = (m[1] * m[6] - m[0] * m[7]) * id;
    o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}
__device__ void mul3(const double* a, const double* b, double* o) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) o[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
}

// one workgroup per batch item; rays [B][6][T][h*w]; frame t0; out_K: [B][4][4][T] pixel-unit intrinsics of an H x W image
// (the same K for every frame: "fixed" intrinsics), diag: [B][2] = (consensus size, iterations used)
// per_frame (the reference's fixed_intrinsics = False branch, geometry_utils.py:582-654): grid (B, T), workgroup (b, t) estimates
// frame t's own K and writes it to frame t only, together with the rotation R of H^-1 = K R (out_R [B][9][T], row-major), which
// that branch uses as the camera rotation directly.  The minimal samples are hashed from the item index b * T + t.
__global__ __launch_bounds__(256) void rays_to_intrinsics_kernel(const float* __restrict__ rays, float* __restrict__ out_K,
                                                                 float* __restrict__ diag, int T, int h, int w, int H, int W,
                                                                 int t0_arg, float thr, float z_thr, float* __restrict__ out_R,
                                                                 int per_frame) {
    const int b = blockIdx.x, r = threadIdx.x, nr = h * w;
    const int t0 = per_frame ? (int)blockIdx.y : t0_arg;
    const int item = per_frame ? b * T + t0 : b;
    __shared__ double red[4][48];
    __shared__ double Hs[9];
    __shared__ double nrm[8];  // src: mx,my,s ; dst: mx,my,s ; count
    __shared__ int cnt_s;
    const long long plane = (long long)T * nr;
    bool valid = false;
    double x = 0, y = 0, u = 0, v = 0;
    if (r < nr) {
        const float* rp = rays + (long long)b * 6 * plane + (long long)t0 * nr + r;
        const double dx = rp[0], dy = rp[plane], dz = rp[2 * plane];
        x = (double)(r % w);
        y = (double)(r / w);
        valid = fabs(dz) > z_thr;  // the identity-K ray has z = 1/|(i,j,1)| > z_thr always
        if (valid) {
            u = dx / dz;
            v = dy / dz;
        }
    }
    // ---- RANSAC initialisation: 128 minimal (4-point) homographies, scored by consensus ------------
    __shared__ float lx[256], ly[256], lu[256], lv[256];
    __shared__ unsigned char lval[256];
    __shared__ float Hts[128][9];
    __shared__ int counts[128];
    lx[r] = (float)x;
    ly[r] = (float)y;
    lu[r] = (float)u;
    lv[r] = (float)v;
    lval[r] = valid ? 1 : 0;
    if (r < 128) counts[r] = 0;
    __syncthreads();
    if (r < 128) {
        int idx[4];
        bool ok = true;
        for (int k = 0; k < 4; ++k) {
            unsigned hsh = (unsigned)(item * 131 + r) * 2654435761u + 40503u * (unsigned)k;
            hsh ^= hsh >> 15;
            hsh *= 2246822519u;
            hsh ^= hsh >> 13;
            idx[k] = (int)(hsh % (unsigned)nr);
            ok = ok && lval[idx[k]];
            for (int q = 0; q < k; ++q) ok = ok && idx[q] != idx[k];
        }
        double Am[45];
#pragma unroll
        for (int i = 0; i < 45; ++i) Am[i] = 0;
        if (ok) {
            // Hartley normalisation of the 4 points
            double mx = 0, my = 0, mu = 0, mv = 0;
            for (int k = 0; k < 4; ++k) {
                mx += lx[idx[k]];
                my += ly[idx[k]];
                mu += lu[idx[k]];
                mv += lv[idx[k]];
            }
            mx /= 4; my /= 4; mu /= 4; mv /= 4;
            double ds = 0, dd = 0;
            for (int k = 0; k < 4; ++k) {
                ds += sqrt((lx[idx[k]] - mx) * (lx[idx[k]] - mx) + (ly[idx[k]] - my) * (ly[idx[k]] - my));
                dd += sqrt((lu[idx[k]] - mu) * (lu[idx[k]] - mu) + (lv[idx[k]] - mv) * (lv[idx[k]] - mv));
            }
            const double ss = ds > 0 ? 1.4142135623730951 * 4 / ds : 1.0, sd = dd > 0 ? 1.4142135623730951 * 4 / dd : 1.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double xn = (lx[idx[k]] - mx) * ss, yn = (ly[idx[k]] - my) * ss;
                const double un = (lu[idx[k]] - mu) * sd, vn = (lv[idx[k]] - mv) * sd;
                const double r1[9] = {-xn, -yn, -1, 0, 0, 0, un * xn, un * yn, un};
                const double r2[9] = {0, 0, 0, -xn, -yn, -1, vn * xn, vn * yn, vn};
#pragma unroll
                for (int i = 0; i < 9; ++i)
#pragma unroll
                    for (int j = 0; j <= i; ++j) Am[i * (i + 1) / 2 + j] += r1[i] * r1[j] + r2[i] * r2[j];
            }
            double Hn[9];
            smallest_eigvec9(Am, Hn);
            double tmp[9], Hd[9];
            const double Ts[9] = {ss, 0, -ss * mx, 0, ss, -ss * my, 0, 0, 1};
            const double Tdi[9] = {1 / sd, 0, mu, 0, 1 / sd, mv, 0, 0, 1};
            mul3(Hn, Ts, tmp);
            mul3(Tdi, tmp, Hd);
            for (int i = 0; i < 9; ++i) Hts[r][i] = (float)Hd[i];
        } else {
            for (int i = 0; i < 9; ++i) Hts[r][i] = 0.f;
        }
    }
    __syncthreads();
    if (valid) {
        for (int t = 0; t < 128; ++t) {
            const float* Ht = Hts[t];
            const float pw = Ht[6] * (float)x + Ht[7] * (float)y + Ht[8];
            const float pu = (Ht[0] * (float)x + Ht[1] * (float)y + Ht[2]) / pw, pv = (Ht[3] * (float)x + Ht[4] * (float)y + Ht[5]) / pw;
            const float e = sqrtf((pu - (float)u) * (pu - (float)u) + (pv - (float)v) * (pv - (float)v));
            if (e < thr) atomicAdd(&counts[t], 1);  // NaN (degenerate trial) compares false
        }
    }
    __syncthreads();
    __shared__ int best_t;
    if (threadIdx.x == 0) {
        int bt = 0;
        for (int t = 1; t < 128; ++t)
            if (counts[t] > counts[bt]) bt = t;
        best_t = bt;
    }
    __syncthreads();
    bool wgt = false;
    if (valid) {
        const float* Ht = Hts[best_t];
        const float pw = Ht[6] * (float)x + Ht[7] * (float)y + Ht[8];
        const float pu = (Ht[0] * (float)x + Ht[1] * (float)y + Ht[2]) / pw, pv = (Ht[3] * (float)x + Ht[4] * (float)y + Ht[5]) / pw;
        wgt = sqrtf((pu - (float)u) * (pu - (float)u) + (pv - (float)v) * (pv - (float)v)) < thr;
    }
    if (counts[best_t] < 8) wgt = valid;  // no usable minimal model: fall back to the plain least-squares start
    // ---- least-squares re-estimation on the consensus set, iterated until the set is stable --------
    int iters = 0;
    for (int it = 0; it < 8; ++it) {
        // ---- Hartley normalisation over the current consensus set ----
        double s7[7] = {wgt ? 1.0 : 0.0, wgt ? x : 0, wgt ? y : 0, wgt ? u : 0, wgt ? v : 0, 0, 0};
        for (int k = 0; k < 5; ++k)
            for (int o = 32; o > 0; o >>= 1) s7[k] += __shfl_xor(s7[k], o);
        __syncthreads();
        if ((threadIdx.x & 63) == 0)
            for (int k = 0; k < 5; ++k) red[threadIdx.x >> 6][k] = s7[k];
        __syncthreads();
        if (threadIdx.x == 0) {
            double t[5];
            for (int k = 0; k < 5; ++k) t[k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
            nrm[6] = t[0];
            const double n = t[0] > 0 ? t[0] : 1;
            nrm[0] = t[1] / n;
            nrm[1] = t[2] / n;
            nrm[3] = t[3] / n;
            nrm[4] = t[4] / n;
            cnt_s = (int)t[0];
        }
        __syncthreads();
        if (cnt_s < 4) break;
        double d2[2] = {0, 0};
        if (wgt) {
            d2[0] = sqrt((x - nrm[0]) * (x - nrm[0]) + (y - nrm[1]) * (y - nrm[1]));
            d2[1] = sqrt((u - nrm[3]) * (u - nrm[3]) + (v - nrm[4]) * (v - nrm[4]));
        }
        for (int k = 0; k < 2; ++k)
            for (int o = 32; o > 0; o >>= 1) d2[k] += __shfl_xor(d2[k], o);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) {
            red[threadIdx.x >> 6][0] = d2[0];
            red[threadIdx.x >> 6][1] = d2[1];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const double a = (red[0][0] + red[1][0] + red[2][0] + red[3][0]) / nrm[6];
            const double c = (red[0][1] + red[1][1] + red[2][1] + red[3][1]) / nrm[6];
            nrm[2] = a > 0 ? 1.4142135623730951 / a : 1.0;
            nrm[5] = c > 0 ? 1.4142135623730951 / c : 1.0;
        }
        __syncthreads();
        // ---- A^T A of the DLT system (2 rows per correspondence), 45 unique entries ----
        double acc[45];
        for (int k = 0; k < 45; ++k) acc[k] = 0;
        if (wgt) {
            const double xn = (x - nrm[0]) * nrm[2], yn = (y - nrm[1]) * nrm[2];
            const double un = (u - nrm[3]) * nrm[5], vn = (v - nrm[4]) * nrm[5];
            const double r1[9] = {-xn, -yn, -1, 0, 0, 0, un * xn, un * yn, un};
            const double r2[9] = {0, 0, 0, -xn, -yn, -1, vn * xn, vn * yn, vn};
            int k = 0;
            for (int i = 0; i < 9; ++i)
                for (int j = i; j < 9; ++j) acc[k++] = r1[i] * r1[j] + r2[i] * r2[j];
        }
        for (int k = 0; k < 45; ++k)
            for (int o = 32; o > 0; o >>= 1) acc[k] += __shfl_xor(acc[k], o);
        __syncthreads();
        if ((threadIdx.x & 63) == 0)
            for (int k = 0; k < 45; ++k) red[threadIdx.x >> 6][k] = acc[k];
        __syncthreads();
        if (threadIdx.x == 0) {
            double Am[45], Hn[9];
            int k = 0;
#pragma unroll
            for (int i = 0; i < 9; ++i)
#pragma unroll
                for (int j = i; j < 9; ++j) {  // (the sums were accumulated as the upper triangle, row-major)
                    Am[j * (j + 1) / 2 + i] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
                    ++k;
                }
            smallest_eigvec9(Am, Hn);
            // denormalise: H = Td^-1 Hn Ts
            const double Ts[9] = {nrm[2], 0, -nrm[2] * nrm[0], 0, nrm[2], -nrm[2] * nrm[1], 0, 0, 1};
            const double Tdi[9] = {1 / nrm[5], 0, nrm[3], 0, 1 / nrm[5], nrm[4], 0, 0, 1};
            double tmp[9];
            mul3(Hn, Ts, tmp);
            mul3(Tdi, tmp, Hs);
        }
        __syncthreads();
        // ---- new consensus set ----
        bool nw = false;
        if (valid) {
            const double pw = Hs[6] * x + Hs[7] * y + Hs[8];
            const double pu = (Hs[0] * x + Hs[1] * y + Hs[2]) / pw, pv = (Hs[3] * x + Hs[4] * y + Hs[5]) / pw;
            nw = sqrt((pu - u) * (pu - u) + (pv - v) * (pv - v)) < thr;
        }
        ++iters;
        const bool same = __all(nw == wgt);
        __shared__ int changed;
        if (threadIdx.x == 0) changed = 0;
        __syncthreads();
        if (!same && (threadIdx.x & 63) == 0) changed = 1;
        __syncthreads();
        // keep the previous set if the new one would be degenerate
        int c2 = nw ? 1 : 0;
        for (int o = 32; o > 0; o >>= 1) c2 += __shfl_xor(c2, o);
        __shared__ int newcnt[4];
        if ((threadIdx.x & 63) == 0) newcnt[threadIdx.x >> 6] = c2;
        __syncthreads();
        const int total = newcnt[0] + newcnt[1] + newcnt[2] + newcnt[3];
        if (!changed || total < 8) break;
        wgt = nw;
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    // ---- H^-1 = K R, RQ decomposition with positive diagonal, K / K[2][2] ----
    double A[9], Hm[9];
    for (int i = 0; i < 9; ++i) A[i] = Hs[i];
    const double det = A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
    if (det < 0)
        for (int i = 0; i < 9; ++i) A[i] = -A[i];
    inv3(A, Hm);
    // rows of Hm from the bottom up: Gram-Schmidt gives R's rows; K is upper triangular
    double q2[3], q1[3], q0[3], K[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const double* m0 = Hm;
    const double* m1 = Hm + 3;
    const double* m2 = Hm + 6;
    K[8] = sqrt(m2[0] * m2[0] + m2[1] * m2[1] + m2[2] * m2[2]);
    for (int i = 0; i < 3; ++i) q2[i] = m2[i] / K[8];
    K[5] = m1[0] * q2[0] + m1[1] * q2[1] + m1[2] * q2[2];
    double t1[3];
    for (int i = 0; i < 3; ++i) t1[i] = m1[i] - K[5] * q2[i];
    K[4] = sqrt(t1[0] * t1[0] + t1[1] * t1[1] + t1[2] * t1[2]);
    for (int i = 0; i < 3; ++i) q1[i] = t1[i] / K[4];
    K[2] = m0[0] * q2[0] + m0[1] * q2[1] + m0[2] * q2[2];
    K[1] = m0[0] * q1[0] + m0[1] * q1[1] + m0[2] * q1[2];
    double t0v[3];
    for (int i = 0; i < 3; ++i) t0v[i] = m0[i] - K[2] * q2[i] - K[1] * q1[i];
    K[0] = sqrt(t0v[0] * t0v[0] + t0v[1] * t0v[1] + t0v[2] * t0v[2]);
    for (int i = 0; i < 3; ++i) q0[i] = t0v[i] / K[0];
    if (out_R) {  // H^-1 = K R with a positive diagonal of K: the rows of R
        for (int j = 0; j < 3; ++j) {
            out_R[((long long)b * 9 + 0 + j) * T + t0] = (float)q0[j];
            out_R[((long long)b * 9 + 3 + j) * T + t0] = (float)q1[j];
            out_R[((long long)b * 9 + 6 + j) * T + t0] = (float)q2[j];
        }
    }
    for (int i = 0; i < 9; ++i) K[i] /= K[8];
    // ray-grid units -> pixel units of the H x W image: denormalize(normalize(K, h, w), H, W)  (geometry_utils.py:110-125,575-577)
    double Kp[16] = {0};
    for (int j = 0; j < 3; ++j) {
        Kp[0 * 4 + j] = K[j];
        Kp[1 * 4 + j] = K[3 + j];
        Kp[2 * 4 + j] = K[6 + j];
    }
    Kp[0 * 4 + 2] += 0.5;
    Kp[1 * 4 + 2] += 0.5;
    for (int j = 0; j < 4; ++j) {
        Kp[0 * 4 + j] = Kp[0 * 4 + j] / w * W;
        Kp[1 * 4 + j] = Kp[1 * 4 + j] / h * H;
    }
    Kp[0 * 4 + 2] -= 0.5;
    Kp[1 * 4 + 2] -= 0.5;
    Kp[15] = 1.0;
    for (int k = 0; k < 16; ++k) {
        if (per_frame)
            out_K[((long long)b * 16 + k) * T + t0] = (float)Kp[k];
        else
            for (int t = 0; t < T; ++t) out_K[((long long)b * 16 + k) * T + t] = (float)Kp[k];
    }
    if (diag) {
        diag[item * 2] = (float)nrm[6];
        diag[item * 2 + 1] = (float)iters;
    }
}

extern "C" {
/* K from the ray map of frame t0 (fixed intrinsics for the whole window).  rays: float [B][6][T][h][w] (h*w <= 256);
 * out_K: float [B][4][4][T] pixel-unit intrinsics of the H x W image; diag (optional): float [B][2]. */
int l4p_rays_to_intrinsics(l4p_stream s_, const float* rays, float* out_K, float* diag, int B, int T, int h, int w, int H,
                           int W, int t0, float reproj_thr) {
    hipStream_t s = (hipStream_t)s_;
    if (h * w > 256 || t0 < 0 || t0 >= T) {
        l4p_set_error("rays_to_intrinsics: unsupported ray map %dx%d / frame %d", h, w, t0);
        return L4P_E_INVALID;
    }
    ProfScope prof(PROF_ELEMENTWISE, s, "l4p_rays_to_intrinsics");
    hipLaunchKernelGGL(rays_to_intrinsics_kernel, dim3(B), dim3(256), 0, s, rays, out_K, diag, T, h, w, H, W, t0, reproj_thr,
                       1e-4f, (float*)nullptr, 0);
    HIP_TRY(hipGetLastError());
    return 0;
}
/* Per-frame variable intrinsics (rays_to_cameras_and_variable_per_frame_intrinsics, geometry_utils.py:582-654): every frame's own
 * K (out_K [B][4][4][T], pixel units of the H x W image) and the rotation of H^-1 = K R (out_R [B][9][T], row-major), which
 * that branch takes as the camera rotation; diag (optional): float [B*T][2]. */
int l4p_rays_to_intrinsics_frames(l4p_stream s_, const float* rays, float* out_K, float* out_R, float* diag, int B, int T, int h,
                                  int w, int H, int W, float reproj_thr) {
    hipStream_t s = (hipStream_t)s_;
    if (h * w > 256 || !out_K || !out_R) {
        l4p_set_error("rays_to_intrinsics_frames: unsupported ray map %dx%d or null output", h, w);
        return L4P_E_INVALID;
    }
    ProfScope prof(PROF_ELEMENTWISE, s, "l4p_rays_to_intrinsics_frames");
    hipLaunchKernelGGL(rays_to_intrinsics_kernel, dim3(B, T), dim3(256), 0, s, rays, out_K, diag, T, h, w, H, W, 0, reproj_thr, 1e-4f,
                       out_R, 1);
    HIP_TRY(hipGetLastError());
    return 0;
}
}
