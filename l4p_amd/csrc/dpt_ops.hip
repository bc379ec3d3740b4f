// HBM-bound pieces of the DPT decoders: channels-last trilinear resize and the final 1x1x1
// projection (+exp) that also converts channels-last T back to the reference's NCDHW float layout.
#include <cstdlib>

#include "common.hpp"

// -------------------------------------------------------------------------------------------------
// Trilinear resize, channels-last [B][Ti][Hi][Wi][C] -> [B][To][Ho][Wo][C].
// Replaces F.interpolate(mode="trilinear") with align_corners=True (dpt_block.py:229-234,
// dpt_head.py:79-83) or False (sparse_heads.py:645-647).  Index/weight arithmetic follows ATen's
// area_pixel_compute_source_index + guard_index_and_lambda in float.
// One thread = 8 channels (16 B of bf16) of one output voxel; the 8 taps are 16-byte loads.
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ void src_index(int dst, int in, int out, bool align, int& i0, int& i1, float& lam) {
    float src;
    if (align) {
        const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
        src = scale * (float)dst;
    } else {
        const float scale = (float)in / (float)out;
        src = scale * ((float)dst + 0.5f) - 0.5f;
        if (src < 0.f) src = 0.f;
    }
    i0 = (int)src;
    if (i0 > in - 1) i0 = in - 1;
    lam = fminf(fmaxf(src - (float)i0, 0.f), 1.f);
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
}
// The same with the scale formed by the caller (ONCE, on the host: an IEEE single division there and here round alike) - the
// per-thread division sequence was a sixth of the up-sampling kernel's vector instructions.
__host__ __device__ __forceinline__ float src_scale(int in, int out, bool align) {
    return align ? (out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f) : (float)in / (float)out;
}
__device__ __forceinline__ void src_index_s(int dst, int in, float scale, bool align, int& i0, int& i1, float& lam) {
    float src;
    if (align) {
        src = scale * (float)dst;
    } else {
        src = scale * ((float)dst + 0.5f) - 0.5f;
        if (src < 0.f) src = 0.f;
    }
    i0 = (int)src;
    if (i0 > in - 1) i0 = in - 1;
    lam = fminf(fmaxf(src - (float)i0, 0.f), 1.f);
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
}

// One workgroup row = one output (b, to, ho) line (blockIdx.y): its t / h source indices and weights are wave-uniform
// scalars, and a thread only splits its x index into (wo, channel group) - the flat-index form spent five integer
// divisions per 16 output bytes.
// An axis that is not resized (lambda == 0: the time axis of the (1,2,2) resizes, every second line of an align-corners x2)
// has one tap, not two.  Which of the t / h taps exist is uniform over the line, so the tap set is chosen ONCE per workgroup
// (NT x NH in {1,2}^2) and inside a specialisation all 2*NT*NH loads of an output are requested before the first is used
// (the per-tap `if (weight == 0) continue` of the previous form kept every load behind a branch: one round trip each).
template <typename T, bool NT_STORE, int NT, int NH>
__device__ __forceinline__ void upsample_line(const T* __restrict__ xb, T* __restrict__ yl, int Hi, int Wi, int Wo, int C, int align,
                                              int t0, int t1, float lt, int h0, int h1, float lh, float sw, int cv_shift) {
    constexpr int V = 8;  // channels per thread
    const int cv = C / V;
    const float wt[2] = {NT == 2 ? 1.f - lt : 1.f, lt}, wh[2] = {NH == 2 ? 1.f - lh : 1.f, lh};
    const int ti[2] = {t0, t1}, hi[2] = {h0, h1};
    // the (t, h) source rows of the line: wave-uniform bases; a lane adds one 32-bit offset per w tap (a clip's input volume is far
    // below 2 GB) - the 64-bit per-tap address arithmetic was another fifth of the kernel's vector instructions
    const T* rowp[NT][NH];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int bb = 0; bb < NH; ++bb) rowp[a][bb] = xb + ((long long)ti[a] * Hi + hi[bb]) * Wi * C;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < Wo * cv; i += gridDim.x * 256) {
        const int wo = cv_shift >= 0 ? i >> cv_shift : i / cv, c = (i - wo * cv) * V;
        int w0, w1;
        float lw;
        src_index_s(wo, Wi, sw, align, w0, w1, lw);
        const float ww[2] = {1.f - lw, lw};
        const unsigned wofs[2] = {(unsigned)(w0 * C + c), (unsigned)(w1 * C + c)};
        float v[NT][NH][2][V];
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int bb = 0; bb < NH; ++bb)
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    const T* p = rowp[a][bb] + wofs[cc];
                    if (sizeof(T) == 2) {
                        const vec8h<T> t = *(const vec8h<T>*)p;
#pragma unroll
                        for (int k = 0; k < V; ++k) v[a][bb][cc][k] = (float)t[k];
                    } else {
                        const f32x4 q0 = *(const f32x4*)p, q1 = *(const f32x4*)((const float*)p + 4);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            v[a][bb][cc][k] = q0[k];
                            v[a][bb][cc][4 + k] = q1[k];
                        }
                    }
                }
        float acc[V];
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] = 0.f;
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int bb = 0; bb < NH; ++bb)
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    const float wgt = wt[a] * wh[bb] * ww[cc];  // same product, same tap order as the 8-tap form
#pragma unroll
                    for (int k = 0; k < V; ++k) acc[k] += wgt * v[a][bb][cc][k];
                }
        T* yp = yl + (unsigned)(i * V);  // (wo * C + c == i * V)
        if (sizeof(T) == 2) {
            vec8h<T> o;
#pragma unroll
            for (int k = 0; k < V; ++k) o[k] = (vec4e<T>)acc[k];
            if (NT_STORE)
                __builtin_nontemporal_store(o, (vec8h<T>*)yp);
            else
                *(vec8h<T>*)yp = o;
        } else {
            *(f32x4*)yp = (f32x4){acc[0], acc[1], acc[2], acc[3]};
            *(f32x4*)((float*)yp + 4) = (f32x4){acc[4], acc[5], acc[6], acc[7]};
        }
    }
}

template <typename T, bool NT = false>
__global__ __launch_bounds__(256) void upsample_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int Ti, int Hi, int Wi,
                                                       int To, int Ho, int Wo, int C, int align, float st, float sh, float sw, int cv_shift) {
    const int line = blockIdx.y + blockIdx.z * 65535;  // (b * To + to) * Ho + ho
    if (line >= B * To * Ho) return;
    const int ho = line % Ho, to = (line / Ho) % To, b = line / (Ho * To);
    int t0, t1, h0, h1;
    float lt, lh;
    src_index_s(to, Ti, st, align, t0, t1, lt);
    src_index_s(ho, Hi, sh, align, h0, h1, lh);
    // (the same for every lane of the line, but formed by vector float instructions: made scalar so that the row bases below are)
    t0 = __builtin_amdgcn_readfirstlane(t0), t1 = __builtin_amdgcn_readfirstlane(t1);
    h0 = __builtin_amdgcn_readfirstlane(h0), h1 = __builtin_amdgcn_readfirstlane(h1);
    const T* xb = x + (long long)b * Ti * Hi * Wi * C;
    T* yl = y + (long long)line * Wo * C;
    // (a zero weight removes the tap exactly as the skipped term did: 0 * finite contributes nothing to the sum)
    if (lt == 0.f) {
        if (lh == 0.f)
            upsample_line<T, NT, 1, 1>(xb, yl, Hi, Wi, Wo, C, align, t0, t1, lt, h0, h1, lh, sw, cv_shift);
        else
            upsample_line<T, NT, 1, 2>(xb, yl, Hi, Wi, Wo, C, align, t0, t1, lt, h0, h1, lh, sw, cv_shift);
    } else {
        if (lh == 0.f)
            upsample_line<T, NT, 2, 1>(xb, yl, Hi, Wi, Wo, C, align, t0, t1, lt, h0, h1, lh, sw, cv_shift);
        else
            upsample_line<T, NT, 2, 2>(xb, yl, Hi, Wi, Wo, C, align, t0, t1, lt, h0, h1, lh, sw, cv_shift);
    }
}

int launch_upsample(int dtype, const void* x, void* y, int B, int Ti, int Hi, int Wi, int To, int Ho, int Wo, int C,
                    int align, hipStream_t stream) {
    if (C % 8) {
        l4p_set_error("upsample: C=%d must be a multiple of 8", C);
        return L4P_E_INVALID;
    }
    const int lines = B * To * Ho, per_line = Wo * (C / 8);
    // two items per thread (fewer, longer workgroups: 2.55 -> 2.18 ms per c3 step) and non-temporal stores (the output,
    // up to 822 MB, is streamed to the next conv and never fits a cache: -> 2.02 ms); L4P_UPS_IPT / L4P_UPS_NT: tuning aids
    static const int ipt = getenv("L4P_UPS_IPT") ? atoi(getenv("L4P_UPS_IPT")) : 2;
    int gx = (per_line + 256 * ipt - 1) / (256 * ipt);
    gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
    const dim3 grid(gx, lines < 65535 ? lines : 65535, (lines + 65534) / 65535);
    ProfScope prof(PROF_ELEMENTWISE, stream, "upsample");
    if ((long long)Ti * Hi * Wi * C * esize_of(dtype) >= (1ll << 31) || (long long)Wo * C >= (1ll << 28)) {
        l4p_set_error("upsample: a clip's input volume must stay below 2 GB (32-bit tap offsets)");
        return L4P_E_INVALID;
    }
    const float st = src_scale(Ti, To, align != 0), sh = src_scale(Hi, Ho, align != 0), sw = src_scale(Wi, Wo, align != 0);
    const int cvn = C / 8;
    int cv_shift = -1;
    if ((cvn & (cvn - 1)) == 0)
        for (cv_shift = 0; (1 << cv_shift) < cvn; ++cv_shift) {
        }
    static const int nt = getenv("L4P_UPS_NT") ? atoi(getenv("L4P_UPS_NT")) : 1;
    if (is16(dtype) && nt) L4P_WITH_T16(dtype, T16, hipLaunchKernelGGL((upsample_kernel<T16, true>), grid, dim3(256), 0, stream, (const T16*)x, (T16*)y, B, Ti,
                           Hi, Wi, To, Ho, Wo, C, align, st, sh, sw, cv_shift));
    else if (is16(dtype)) L4P_WITH_T16(dtype, T16, hipLaunchKernelGGL(upsample_kernel<T16>, grid, dim3(256), 0, stream, (const T16*)x, (T16*)y, B, Ti,
                           Hi, Wi, To, Ho, Wo, C, align, st, sh, sw, cv_shift));
    else
        hipLaunchKernelGGL(upsample_kernel<float>, grid, dim3(256), 0, stream, (const float*)x, (float*)y, B, Ti, Hi,
                           Wi, To, Ho, Wo, C, align, st, sh, sw, cv_shift);
    HIP_TRY(hipGetLastError());
    return 0;
}

// -------------------------------------------------------------------------------------------------
// head2[2]: Conv3d 1x1x1 C -> Cout (Cout <= 8) + optional exp, channels-last T in, NCDHW float out
// (dpt_block.py:413; dense_heads.py:73,179,215; misc.py:23-24).  One thread per voxel: the row is
// 16-byte loads, the store is coalesced along the voxel index for each output channel.
// -------------------------------------------------------------------------------------------------
template <typename T, int C>
__global__ __launch_bounds__(256) void head_out_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ y,
                                                       long long vox_per_b, int B, int Cout, int post_exp) {
    __shared__ float ws[8 * C + 8];
    for (int i = threadIdx.x; i < Cout * C; i += blockDim.x) ws[i] = w[i];
    if (threadIdx.x < Cout) ws[8 * C + threadIdx.x] = bias[threadIdx.x];
    __syncthreads();
    const long long total = vox_per_b * B;
    for (long long m = blockIdx.x * (long long)blockDim.x + threadIdx.x; m < total;
         m += (long long)gridDim.x * blockDim.x) {
        float acc[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) acc[o] = 0.f;
        const T* xp = x + m * C;
#pragma unroll 4
        for (int c0 = 0; c0 < C; c0 += 8) {
            float v[8];
            if (sizeof(T) == 2) {
                const vec8h<T> t = *(const vec8h<T>*)(xp + c0);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = (float)t[k];
            } else {
                const f32x4 a = *(const f32x4*)(xp + c0), b = *(const f32x4*)((const float*)xp + c0 + 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v[k] = a[k];
                    v[4 + k] = b[k];
                }
            }
#pragma unroll
            for (int o = 0; o < 8; ++o)
                if (o < Cout) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[o] += v[k] * ws[o * C + c0 + k];
                }
        }
        const long long b = m / vox_per_b, v = m - b * vox_per_b;
#pragma unroll
        for (int o = 0; o < 8; ++o)
            if (o < Cout) {
                float r = acc[o] + ws[8 * C + o];
                if (post_exp) r = expf(r);
                y[(b * Cout + o) * vox_per_b + v] = r;
            }
    }
}

// LDS-staged form used when C * sizeof(T) == 256 (the bf16 engine): the workgroup copies 256 voxel rows (64 KB, one
// contiguous span) into LDS with coalesced 16-byte LDS-DMA, XOR-swizzled through the SOURCE chunk (LDS chunk q of row r
// holds source chunk q ^ (r & 15)) so the row-per-thread reads that follow are conflict-free.  The direct form above
// reads 16 bytes per lane at a 256-byte stride (64 lines per load instruction, 1.4 TB/s measured).  Same summation order.
template <typename T, int C>
__global__ __launch_bounds__(256) void head_out_lds_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y,
                                                           long long vox_per_b, int B, int Cout, int post_exp) {
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    constexpr int ROWB = C * (int)sizeof(T), CPR = ROWB / 16;  // bytes / 16-byte chunks per row
    static_assert(CPR == 16, "swizzle below assumes 16 chunks per row");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ws = (float*)smem;             // [8][C] + 8 bias
    char* tile = smem + (8 * C + 8) * 4;  // [256][ROWB]
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < Cout * C; i += 256) ws[i] = w[i];
    if (tid < Cout) ws[8 * C + tid] = bias[tid];
    const long long total = vox_per_b * B;
    const long long m0 = (long long)blockIdx.x * 256;
    const int nrow = (int)(total - m0 < 256 ? total - m0 : 256);
    // pass i stages rows i*16 .. i*16+15: thread -> (row = i*16 + tid/16, LDS chunk q = tid%16) <- source chunk q ^ (row & 15)
    const int q = tid & 15, r16 = tid >> 4;
    for (int i = 0; i < 16; ++i) {
        const int row = i * 16 + r16;
        if (row < nrow)
            __builtin_amdgcn_global_load_lds((gptr_t)((const char*)x + (m0 + row) * ROWB + ((q ^ (row & 15)) << 4)),
                                             (lptr_t)(tile + (i * 256 + wave * 64) * 16), 16, 0, 0);
    }
    __syncthreads();
    if (tid >= nrow) return;
    float acc[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = 0.f;
    const char* xr = tile + tid * ROWB;
#pragma unroll 4
    for (int c0 = 0; c0 < C; c0 += 8) {
        const char* cp = xr + ((((c0 * (int)sizeof(T)) >> 4) ^ (tid & 15)) << 4);  // (bf16: one chunk = 8 channels)
        float v[8];
        const vec8h<T> t = *(const vec8h<T>*)cp;
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (float)t[k];
#pragma unroll
        for (int o = 0; o < 8; ++o)
            if (o < Cout) {
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[o] += v[k] * ws[o * C + c0 + k];
            }
    }
    const long long m = m0 + tid;
    const long long b = m / vox_per_b, v = m - b * vox_per_b;
#pragma unroll
    for (int o = 0; o < 8; ++o)
        if (o < Cout) {
            float r = acc[o] + ws[8 * C + o];
            if (post_exp) r = expf(r);
            y[(b * Cout + o) * vox_per_b + v] = r;
        }
}

int launch_head_out(int dtype, const void* x, const float* w, const float* bias, float* y, long long vox_per_b, int B,
                    int C, int Cout, int post_exp, hipStream_t stream) {
    if (C != 128 || Cout < 1 || Cout > 8) {
        l4p_set_error("head_out: supports C == 128 and 1 <= Cout <= 8 (C=%d Cout=%d)", C, Cout);
        return L4P_E_INVALID;
    }
    const long long total = vox_per_b * B;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    ProfScope prof(PROF_ELEMENTWISE, stream, "head_out");
    if (is16(dtype)) L4P_WITH_T16(dtype, T16, {
        auto kern = head_out_lds_kernel<T16, 128>;
        const size_t lds = (8 * 128 + 8) * 4 + 256 * 256;
        HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)((total + 255) / 256)), dim3(256), lds, stream, (const T16*)x, w, bias, y, vox_per_b,
                           B, Cout, post_exp);
    }); else
        hipLaunchKernelGGL((head_out_kernel<float, 128>), dim3(grid), dim3(256), 0, stream, (const float*)x, w, bias, y,
                           vox_per_b, B, Cout, post_exp);
    HIP_TRY(hipGetLastError());
    return 0;
}
