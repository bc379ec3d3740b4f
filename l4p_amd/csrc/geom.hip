// Small geometry / alignment kernels that the reference runs as torch.linalg calls inside Python
// loops (with host syncs): window-seam depth alignment and ray-map -> camera pose.
#include "common.hpp"

// -------------------------------------------------------------------------------------------------
// LstSqAffineAligner (aligner.py:29-66): min_{s,t} || s*f(pred) + t - f(target) ||^2, f = safe_inverse
// (misc.py:48-62) or identity.  The reference builds a (401k x 2) matrix and calls torch.linalg.lstsq;
// the same minimiser is the 2x2 normal-equation solve over four sums, accumulated here in double.
// LinearAligner(method="mean") (aligner.py:69-118): s = mean(f(target) / (f(pred) + 1e-8)), t = 0 (mode bit 1).
// Two-stage reduction in a FIXED order (bit-reproducible, unlike atomics): every workgroup leaves its four partial
// sums in scratch[4 * block .. ], one workgroup then adds the L4P_AFFINE_BLOCKS partials in index order and solves.
// scratch: double[4 * L4P_AFFINE_BLOCKS].   sol: float[2] = (s, t).   mode: bit 0 = inverse, bit 1 = ratio mean.
// -------------------------------------------------------------------------------------------------
#define L4P_AFFINE_BLOCKS 1024
__device__ __forceinline__ float pre_fn(float x, int inverse) { return inverse ? (x > 0.f ? 1.0f / x : 0.f) : x; }

__global__ __launch_bounds__(256) void affine_sums_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                          long long n, int mode, double* __restrict__ scratch) {
    const int inverse = mode & 1, ratio = mode & 2;
    double sa = 0, saa = 0, sb = 0, sab = 0;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float af = pre_fn(pred[i], inverse), bf = pre_fn(tgt[i], inverse);
        if (ratio) {
            sb += (double)(bf / (af + 1e-8f));  // (the reference divides in float: aligner.py:103-104)
        } else {
            const double a = af, b = bf;
            sa += a;
            saa += a * a;
            sb += b;
            sab += a * b;
        }
    }
    __shared__ double red[4][4];
    double v[4] = {sa, saa, sb, sab};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 4; ++k) red[wave][k] = v[k];
    __syncthreads();
    if (threadIdx.x < 4)
        scratch[4 * blockIdx.x + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
// one workgroup of 256 threads: thread j sums partials j, j + 256, ... in order, then a fixed tree over the 256 threads
__global__ __launch_bounds__(256) void affine_solve_kernel(const double* __restrict__ scratch, int nblocks, long long n, int mode,
                                                           float* __restrict__ sol) {
    __shared__ double red[256][4];
    double v[4] = {0, 0, 0, 0};
    for (int b = threadIdx.x; b < nblocks; b += 256)
        for (int k = 0; k < 4; ++k) v[k] += scratch[4 * b + k];
    for (int k = 0; k < 4; ++k) red[threadIdx.x][k] = v[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            for (int k = 0; k < 4; ++k) red[threadIdx.x][k] += red[threadIdx.x + o][k];
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    const double N = (double)n, sa = red[0][0], saa = red[0][1], sb = red[0][2], sab = red[0][3];
    double s = 0.0, t = 0.0;
    if (mode & 2) {
        s = n > 0 ? sb / N : 1.0;
    } else {
        const double det = N * saa - sa * sa;
        if (fabs(det) > 0.0) {
            s = (N * sab - sa * sb) / det;
            t = (saa * sb - sa * sab) / det;
        }
    }
    sol[0] = (float)s;
    sol[1] = (float)t;
}
__global__ void affine_apply_kernel(const float* __restrict__ x, float* __restrict__ y, long long n, int inverse,
                                    const float* __restrict__ sol) {
    const float s = sol[0], t = sol[1];
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = s * pre_fn(x[i], inverse) + t;
        y[i] = pre_fn(v, inverse);
    }
}

int launch_affine_solve(const float* pred, const float* tgt, long long n, int mode, double* scratch, float* sol,
                        hipStream_t stream) {
    ProfScope prof(PROF_ELEMENTWISE, stream, "affine_solve");
    const int grid = (int)((n + 255) / 256 < L4P_AFFINE_BLOCKS ? (n + 255) / 256 : L4P_AFFINE_BLOCKS);
    if (grid > 0) hipLaunchKernelGGL(affine_sums_kernel, dim3(grid), dim3(256), 0, stream, pred, tgt, n, mode, scratch);
    hipLaunchKernelGGL(affine_solve_kernel, dim3(1), dim3(256), 0, stream, scratch, grid, n, mode, sol);
    HIP_TRY(hipGetLastError());
    return 0;
}
int launch_affine_apply(const float* x, float* y, long long n, int inverse, const float* sol, hipStream_t stream) {
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    ProfScope prof(PROF_ELEMENTWISE, stream, "affine_apply");
    if (grid > 0) hipLaunchKernelGGL(affine_apply_kernel, dim3(grid), dim3(256), 0, stream, x, y, n, inverse & 1, sol);
    HIP_TRY(hipGetLastError());
    return 0;
}

// -------------------------------------------------------------------------------------------------
// rays_to_cameras (geometry_utils.py:331-406) + pose = inv(extrinsics) (dense_heads.py:346-348).
// One workgroup per (b, t) frame, one thread per ray of the h x w (= 16 x 16) ray map:
//   * Pluecker (d, m) -> point p = d x (m/|d|), unit direction  (geometry_utils.py:308-328)
//   * camera centre = argmin sum_r |(I - dd^T)(c - p_r)|^2: 3x3 normal equations (:249-282)
//   * rotation = Kabsch between ideal pixel rays K'^-1 [i j 1]^T and the predicted directions (:285-305),
//     3x3 SVD by Jacobi in double; the det-sign fix of the reference is folded in by taking the third
//     singular vectors as cross products (u1 v1^T + u2 v2^T + (u1xu2)(v1xv2)^T).
//   * world_T_cam = [R^T | c] written row-major into out[b][16][T].
// K: pixel-unit intrinsics [B][4][4][T] for an H x W image (normalised / re-scaled to the ray grid as
// normalize_intrinsics + denormalize_intrinsics do, :110-125).
// -------------------------------------------------------------------------------------------------
__device__ void jacobi_eig3(double A[3][3], double V[3][3]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (off < 1e-300) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (fabs(A[p][q]) < 1e-300) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
}

// R_in (optional, [B][9][T] row-major): the rotation is GIVEN (the per-frame variable-intrinsics branch takes it from the RQ
// decomposition, geometry_utils.py:636-644) - only the camera centre is solved here; K is then unused.
__global__ __launch_bounds__(256) void rays_to_pose_kernel(const float* __restrict__ rays, const float* __restrict__ K,
                                                           float* __restrict__ out, int B, int T, int h, int w, int H,
                                                           int W, const float* __restrict__ R_in) {
    const int bt = blockIdx.x, b = bt / T, t = bt % T;
    const int r = threadIdx.x, nr = h * w;
    __shared__ double red[4][24];
    __shared__ double Kinv[9];
    if (threadIdx.x == 0 && R_in) {
        for (int i = 0; i < 9; ++i) Kinv[i] = i % 4 == 0 ? 1.0 : 0.0;  // (unused: the Kabsch sums are ignored below)
    } else if (threadIdx.x == 0) {
        // K' = denormalize(normalize(K, H, W), h, w), upper-left 3x3; then invert
        double k[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) k[i][j] = K[(((long long)b * 4 + i) * 4 + j) * T + t];
        k[0][2] += 0.5;
        k[1][2] += 0.5;
        for (int j = 0; j < 3; ++j) {
            k[0][j] = k[0][j] / W * w;
            k[1][j] = k[1][j] / H * h;
        }
        k[0][2] -= 0.5;
        k[1][2] -= 0.5;
        const double det = k[0][0] * (k[1][1] * k[2][2] - k[1][2] * k[2][1]) - k[0][1] * (k[1][0] * k[2][2] - k[1][2] * k[2][0]) +
                           k[0][2] * (k[1][0] * k[2][1] - k[1][1] * k[2][0]);
        const double id = 1.0 / det;
        Kinv[0] = (k[1][1] * k[2][2] - k[1][2] * k[2][1]) * id;
        Kinv[1] = (k[0][2] * k[2][1] - k[0][1] * k[2][2]) * id;
        Kinv[2] = (k[0][1] * k[1][2] - k[0][2] * k[1][1]) * id;
        Kinv[3] = (k[1][2] * k[2][0] - k[1][0] * k[2][2]) * id;
        Kinv[4] = (k[0][0] * k[2][2] - k[0][2] * k[2][0]) * id;
        Kinv[5] = (k[0][2] * k[1][0] - k[0][0] * k[1][2]) * id;
        Kinv[6] = (k[1][0] * k[2][1] - k[1][1] * k[2][0]) * id;
        Kinv[7] = (k[0][1] * k[2][0] - k[0][0] * k[2][1]) * id;
        Kinv[8] = (k[0][0] * k[1][1] - k[0][1] * k[1][0]) * id;
    }
    __syncthreads();
    double v[18];
    for (int k = 0; k < 18; ++k) v[k] = 0.0;
    if (r < nr) {
        const long long plane = (long long)T * nr;
        const float* rp = rays + (long long)b * 6 * plane + (long long)t * nr + r;
        const double d0 = rp[0], d1 = rp[plane], d2 = rp[2 * plane];
        double m0 = rp[3 * plane], m1 = rp[4 * plane], m2 = rp[5 * plane];
        const double dn = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
        m0 /= dn;
        m1 /= dn;
        m2 /= dn;
        const double p0 = d1 * m2 - d2 * m1, p1 = d2 * m0 - d0 * m2, p2 = d0 * m1 - d1 * m0;  // d x m
        const double dd = fmax(dn, 1e-12);                                                   // F.normalize eps
        const double u0 = d0 / dd, u1 = d1 / dd, u2 = d2 / dd;
        // P = I - u u^T (symmetric): P00 P01 P02 P11 P12 P22 ; rhs = P p
        const double P00 = 1 - u0 * u0, P01 = -u0 * u1, P02 = -u0 * u2, P11 = 1 - u1 * u1, P12 = -u1 * u2, P22 = 1 - u2 * u2;
        v[0] = P00; v[1] = P01; v[2] = P02; v[3] = P11; v[4] = P12; v[5] = P22;
        v[6] = P00 * p0 + P01 * p1 + P02 * p2;
        v[7] = P01 * p0 + P11 * p1 + P12 * p2;
        v[8] = P02 * p0 + P12 * p1 + P22 * p2;
        // ideal ray through pixel (i = column, j = row)
        const double pi = (double)(r % w), pj = (double)(r / w);
        double a0 = Kinv[0] * pi + Kinv[1] * pj + Kinv[2], a1 = Kinv[3] * pi + Kinv[4] * pj + Kinv[5],
               a2 = Kinv[6] * pi + Kinv[7] * pj + Kinv[8];
        const double an = sqrt(a0 * a0 + a1 * a1 + a2 * a2);
        a0 /= an; a1 /= an; a2 /= an;
        // Hm = B^T A, B = raw directions, A = ideal rays:  Hm[i][j] = sum d_i a_j
        v[9] = d0 * a0; v[10] = d0 * a1; v[11] = d0 * a2;
        v[12] = d1 * a0; v[13] = d1 * a1; v[14] = d1 * a2;
        v[15] = d2 * a0; v[16] = d2 * a1; v[17] = d2 * a2;
    }
#pragma unroll
    for (int k = 0; k < 18; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 18; ++k) red[wave][k] = v[k];
    __syncthreads();
    if (threadIdx.x != 0) return;
    double s[18];
    for (int k = 0; k < 18; ++k) s[k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
    // ---- centre: solve M c = rhs, M symmetric 3x3 ----
    const double M00 = s[0], M01 = s[1], M02 = s[2], M11 = s[3], M12 = s[4], M22 = s[5];
    const double c00 = M11 * M22 - M12 * M12, c01 = M02 * M12 - M01 * M22, c02 = M01 * M12 - M02 * M11;
    const double c11 = M00 * M22 - M02 * M02, c12 = M01 * M02 - M00 * M12, c22 = M00 * M11 - M01 * M01;
    const double det = M00 * c00 + M01 * c01 + M02 * c02;
    const double cx = (c00 * s[6] + c01 * s[7] + c02 * s[8]) / det;
    const double cy = (c01 * s[6] + c11 * s[7] + c12 * s[8]) / det;
    const double cz = (c02 * s[6] + c12 * s[7] + c22 * s[8]) / det;
    // ---- rotation: Hm = U S V^T ; Rpre = u1 v1^T + u2 v2^T + (u1 x u2)(v1 x v2)^T ; R = Rpre^T ----
    double Hm[3][3] = {{s[9], s[10], s[11]}, {s[12], s[13], s[14]}, {s[15], s[16], s[17]}};
    double A[3][3], V[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) A[i][j] = Hm[0][i] * Hm[0][j] + Hm[1][i] * Hm[1][j] + Hm[2][i] * Hm[2][j];
    jacobi_eig3(A, V);
    int i1 = 0;
    for (int k = 1; k < 3; ++k)
        if (A[k][k] > A[i1][i1]) i1 = k;
    int i2 = -1;
    for (int k = 0; k < 3; ++k)
        if (k != i1 && (i2 < 0 || A[k][k] > A[i2][i2])) i2 = k;
    double v1[3] = {V[0][i1], V[1][i1], V[2][i1]}, v2[3] = {V[0][i2], V[1][i2], V[2][i2]};
    double v3[3] = {v1[1] * v2[2] - v1[2] * v2[1], v1[2] * v2[0] - v1[0] * v2[2], v1[0] * v2[1] - v1[1] * v2[0]};
    double u1[3], u2[3];
    for (int i = 0; i < 3; ++i) {
        u1[i] = Hm[i][0] * v1[0] + Hm[i][1] * v1[1] + Hm[i][2] * v1[2];
        u2[i] = Hm[i][0] * v2[0] + Hm[i][1] * v2[1] + Hm[i][2] * v2[2];
    }
    double n1 = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
    for (int i = 0; i < 3; ++i) u1[i] /= n1;
    const double dp = u1[0] * u2[0] + u1[1] * u2[1] + u1[2] * u2[2];
    for (int i = 0; i < 3; ++i) u2[i] -= dp * u1[i];
    double n2 = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
    for (int i = 0; i < 3; ++i) u2[i] /= n2;
    double u3[3] = {u1[1] * u2[2] - u1[2] * u2[1], u1[2] * u2[0] - u1[0] * u2[2], u1[0] * u2[1] - u1[1] * u2[0]};
    double R[3][3];  // R = Rpre^T, Rpre[i][j] = sum_k u_k[i] v_k[j]
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[j][i] = u1[i] * v1[j] + u2[i] * v2[j] + u3[i] * v3[j];
    if (R_in)
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) R[i][j] = R_in[((long long)b * 9 + i * 3 + j) * T + t];
    // world_T_cam = inv([R | -R c]) = [R^T | c]
    float* op = out + (long long)b * 16 * T + t;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) op[(long long)(i * 4 + j) * T] = (float)R[j][i];
    }
    op[(long long)3 * T] = (float)cx;
    op[(long long)7 * T] = (float)cy;
    op[(long long)11 * T] = (float)cz;
    op[(long long)12 * T] = 0.f;
    op[(long long)13 * T] = 0.f;
    op[(long long)14 * T] = 0.f;
    op[(long long)15 * T] = 1.f;
}

int launch_rays_to_pose(const float* rays, const float* K, float* out, int B, int T, int h, int w, int H, int W,
                        hipStream_t stream) {
    if (h * w > 256) {
        l4p_set_error("rays_to_pose: ray map %dx%d larger than 256 rays per frame", h, w);
        return L4P_E_INVALID;
    }
    ProfScope prof(PROF_ELEMENTWISE, stream, "rays_to_pose");
    hipLaunchKernelGGL(rays_to_pose_kernel, dim3(B * T), dim3(256), 0, stream, rays, K, out, B, T, h, w, H, W, (const float*)nullptr);
    HIP_TRY(hipGetLastError());
    return 0;
}
int launch_rays_to_pose_rot(const float* rays, const float* R, float* out, int B, int T, int h, int w, hipStream_t stream) {
    if (h * w > 256 || !R) {
        l4p_set_error("rays_to_pose_rot: ray map %dx%d larger than 256 rays per frame, or no rotations", h, w);
        return L4P_E_INVALID;
    }
    ProfScope prof(PROF_ELEMENTWISE, stream, "rays_to_pose_rot");
    hipLaunchKernelGGL(rays_to_pose_kernel, dim3(B * T), dim3(256), 0, stream, rays, (const float*)nullptr, out, B, T, h, w, 1, 1, R);
    HIP_TRY(hipGetLastError());
    return 0;
}
