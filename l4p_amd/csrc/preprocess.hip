// Clip preparation on the GPU: the caller side of the hot path.
//
// Replaces, for decoded uint8 frames that are already in HBM,
//   * VideoDataset.getitem_helper (reference l4p/data/video_dataset.py:86-93): per frame
//       PIL resize(BILINEAR) down to resize_size and back up ("antialias" blur), to_tensor (/255);
//   * L4PDataset.__getitem__ (l4p/data/l4p_dataset_mini.py:543-587): temporal mirror-padding (:126-190), spatial
//       F.interpolate(trilinear, frame count unchanged = per-frame bilinear) (:236-288), centre crop (:290-391),
//       ImageNet normalisation (:575-580), layout [3][T][h][w] float.
//
// This is byte / integer work bound by HBM, not MFMA work:
//  * `pil_resample_{h_lds,h,v}_kernel` are one pass of Pillow's 8-bit resampler (libImaging/Resample.c: 22-bit fixed-point
//    triangle filter whose support grows with the down-scale factor, uint8 between passes) — integer arithmetic, bit-exact.
//    Horizontal pass: a workgroup stages <= 16 contiguous image rows in LDS with dword loads; a thread owns an output
//    column (x, channel), holds that column's coefficients in registers and walks down the rows (the table-driven form
//    also produces any SUBSET of output columns: the up-scaling pass only makes the two columns each final output column
//    reads).  Vertical pass: 4 consecutive bytes of a row per thread (one dword load per tap, one dword store).
//  * `clip_resize_normalize_kernel` produces the network input directly.  An output pixel only touches <= 4 source
//    pixels, so the LAST blur pass (vertical up-scale back to the full frame height — the largest intermediate, as
//    big as the video itself) is never written: the 4 neighbours are evaluated on the fly from the horizontally
//    up-scaled rows with the same integer arithmetic (FUSE_V), then /255, the two lerps, (x - mean) / std.
//    The float arithmetic follows ATen's CPU kernel: source coordinate = ONE fma(scale, dst + 0.5, -0.5).
#include <cmath>
#include <cstdint>

#include "common.hpp"
#include "l4p_hip.h"

// This file is compiled with -ffp-contract=off (Makefile): no fused multiply-adds except the one written as fmaf
// (HIP's __fmul_rn / __fadd_rn are plain operators the compiler may still contract).  With that the float tensor equals
// the un-contracted float32 restatement in oracle/preprocess_oracle.py bit for bit, for both kernel instances.

namespace {

constexpr int PIL_BITS = 32 - 8 - 2;  // Resample.c PRECISION_BITS

__device__ __forceinline__ int clip8(int acc) {
    const int v = acc >> PIL_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// Horizontal pass.  src [rows][in_w][C] -> dst [rows][out_w][C]; bounds [out_w][2], kk [out_w][ksize].
// VEC consecutive bytes of the FLAT output per thread (a dword store; rows of 3-channel pixels are rarely a multiple of
// 4 bytes, the whole tensor usually is), each with its own (row, x, channel); the tail bytes are written one by one.
template <int VEC>
__global__ __launch_bounds__(256) void pil_resample_h_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                             long long rows, int in_w, int out_w, int C,
                                                             const int* __restrict__ bounds, const int* __restrict__ kk,
                                                             int ksize) {
    const int row_bytes = out_w * C;
    const long long total = rows * row_bytes;
    const long long nvec = (total + VEC - 1) / VEC;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < nvec; i += gridDim.x * 256ll) {
        const long long o = i * VEC;
        long long row = o / row_bytes;
        int r = (int)(o - row * row_bytes);
        int xx = r / C, c = r - xx * C;
        uint32_t packed = 0;
#pragma unroll
        for (int b = 0; b < VEC; ++b) {
            if (o + b < total) {
                const int x0 = bounds[2 * xx], n = bounds[2 * xx + 1];
                const int* k = kk + (long long)xx * ksize;
                const uint8_t* p = src + (row * in_w + x0) * C + c;
                int acc = 1 << (PIL_BITS - 1);
                for (int x = 0; x < n; ++x) acc += (int)p[x * C] * k[x];
                packed |= (uint32_t)clip8(acc) << (8 * b);
            }
            if (++c == C) {
                c = 0;
                if (++xx == out_w) {
                    xx = 0;
                    ++row;
                }
            }
        }
        if (VEC == 4 && o + 4 <= total) {
            *(uint32_t*)(dst + o) = packed;
        } else {
            for (int b = 0; b < VEC && o + b < total; ++b) dst[o + b] = (uint8_t)(packed >> (8 * b));
        }
    }
}

// Horizontal pass through LDS (launched when src is dword aligned and the filter has <= 17 taps): a workgroup takes R
// consecutive image rows — one contiguous byte range of the input — and copies it to LDS with coalesced dword loads.
// A thread owns an output column (x, channel): it reads that column's bounds and coefficients ONCE into registers and
// walks down the R rows, so the inner loop is KS independent LDS byte reads + multiply-adds and one byte store (a wave
// stores 64 consecutive bytes).  The global-memory form above re-read the coefficient row for every output byte and
// ran the 3.8x down-scale of a 50-frame 480x854 video at 0.5 TB/s.
template <int KS>
__global__ __launch_bounds__(256) void pil_resample_h_lds_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                                 long long rows, int in_w, int out_w, int C, int R,
                                                                 const int* __restrict__ bounds, const int* __restrict__ kk,
                                                                 int ksize) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int in_rb = in_w * C, out_rb = out_w * C;
    const long long total_in = rows * in_rb;
    const long long nblk = (rows + R - 1) / R;
    for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const long long r0 = blk * R;
        const int nr = (int)(rows - r0 < R ? rows - r0 : R);
        const long long start = r0 * in_rb, end = start + (long long)nr * in_rb;
        const long long a = start & ~3ll;
        const int ndw = (int)((end - a + 3) / 4);
        for (int i = threadIdx.x; i < ndw; i += 256) {
            const long long off = a + 4ll * i;
            uint32_t v = 0;
            if (off + 4 <= total_in) {
                v = *(const uint32_t*)(src + off);
            } else {  // the last dword of the tensor: never read past its end
                for (int b = 0; off + b < total_in; ++b) v |= (uint32_t)src[off + b] << (8 * b);
            }
            ((uint32_t*)lds)[i] = v;
        }
        __syncthreads();
        const uint8_t* rows_lds = lds + (int)(start - a);
        for (int col = threadIdx.x; col < out_rb; col += 256) {
            const int xx = col / C, c = col - xx * C;
            const int x0 = bounds[2 * xx], n = bounds[2 * xx + 1];
            int kreg[KS];
#pragma unroll
            for (int x = 0; x < KS; ++x) kreg[x] = x < n ? kk[(long long)xx * ksize + x] : 0;
            const uint8_t* p = rows_lds + x0 * C + c;
            uint8_t* d = dst + r0 * out_rb + col;
            for (int lr = 0; lr < nr; ++lr) {
                int acc = 1 << (PIL_BITS - 1);
#pragma unroll
                for (int x = 0; x < KS; ++x) acc += (int)p[x * C] * kreg[x];  // taps past n: zero coefficient, LDS slack
                *d = (uint8_t)clip8(acc);
                p += in_rb;
                d += out_rb;
            }
        }
        __syncthreads();
    }
}

// Vertical pass.  src [imgs][in_h][row_bytes] -> dst [imgs][out_h][row_bytes]; 4 bytes per thread (row_bytes % 4 == 0)
// or 1 byte per thread (VEC = 1).
template <int VEC>
__global__ __launch_bounds__(256) void pil_resample_v_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                             long long imgs, int in_h, int out_h, int row_bytes,
                                                             const int* __restrict__ bounds, const int* __restrict__ kk,
                                                             int ksize) {
    const int rv = row_bytes / VEC;
    const long long total = imgs * out_h * rv;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
        const long long line = i / rv;  // img * out_h + yy
        const int xb = (int)(i - line * rv) * VEC;
        const long long img = line / out_h;
        const int yy = (int)(line - img * out_h);
        const int y0 = bounds[2 * yy], n = bounds[2 * yy + 1];
        const int* k = kk + (long long)yy * ksize;
        const uint8_t* p = src + (img * in_h + y0) * row_bytes + xb;
        if constexpr (VEC == 4) {
            int a0 = 1 << (PIL_BITS - 1), a1 = a0, a2 = a0, a3 = a0;
            for (int y = 0; y < n; ++y) {
                const uint32_t v = *(const uint32_t*)(p + (long long)y * row_bytes);
                const int w = k[y];
                a0 += (int)(v & 255u) * w;
                a1 += (int)((v >> 8) & 255u) * w;
                a2 += (int)((v >> 16) & 255u) * w;
                a3 += (int)(v >> 24) * w;
            }
            *(uint32_t*)(dst + line * row_bytes + xb) =
                (uint32_t)clip8(a0) | ((uint32_t)clip8(a1) << 8) | ((uint32_t)clip8(a2) << 16) | ((uint32_t)clip8(a3) << 24);
        } else {
            int acc = 1 << (PIL_BITS - 1);
            for (int y = 0; y < n; ++y) acc += (int)p[(long long)y * row_bytes] * k[y];
            dst[line * row_bytes + xb] = (uint8_t)clip8(acc);
        }
    }
}

struct ClipArgs {
    const uint8_t* frames;   // FUSE_V: [n_src][src_h][in_w][3] (rows before the last vertical pass), else [n_src][in_h][in_w][3]
    const int* frame_index;  // [T_out] source frame of every output frame (stride, mirror-padding, temporal crop)
    float* out;              // [3][T_out][out_h][out_w]
    int T_out, in_h, in_w, src_h;
    int res_h, res_w;        // size F.interpolate resizes the frame to
    int i0, j0;              // centre-crop offsets inside the resized frame
    int out_h, out_w;
    int identity;            // resize factor exactly 1 on both axes: the reference skips F.interpolate
    const int* vbounds;      // FUSE_V: Pillow tables of the vertical pass src_h -> in_h
    const int* vkk;
    int vksize;
    const float* xl;         // compact-x mode: [out_w] horizontal lerp weights (l4p_resize_index_table), else NULL
    float mean[3], stdv[3];
};

// ATen area_pixel_compute_source_index (align_corners = False) + guard_index_and_lambda, float32, index as one fma
__host__ __device__ inline void src_index(float scale, int dst, int n_in, int& i0, int& i1, float& l1) {
    float s = fmaf(scale, (float)dst + 0.5f, -0.5f);
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i0 = i0 < n_in - 1 ? i0 : n_in - 1;
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    l1 = s - (float)i0;
    l1 = l1 < 0.f ? 0.f : (l1 > 1.f ? 1.f : l1);
}

// to_tensor: uint8 / 255 as a correctly rounded IEEE division (torch's .div(255)), never a reciprocal multiply
__device__ __forceinline__ float u8f(int v) { return __fdiv_rn((float)v, 255.f); }

template <bool FUSE_V>
__global__ __launch_bounds__(256) void clip_resize_normalize_kernel(ClipArgs a) {
    const int per_frame = a.out_h * a.out_w;
    const long long total = (long long)a.T_out * per_frame;
    const float sy = (float)a.in_h / (float)a.res_h, sx = (float)a.in_w / (float)a.res_w;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
        const int t = (int)(i / per_frame);
        const int r = (int)(i - (long long)t * per_frame);
        const int oy = r / a.out_w, ox = r - oy * a.out_w;
        int y0, y1, x0, x1;
        float ly, lx;
        if (a.identity) {
            y0 = y1 = oy + a.i0;
            x0 = x1 = ox + a.j0;
            ly = lx = 0.f;
        } else {
            src_index(sy, oy + a.i0, a.in_h, y0, y1, ly);
            if (a.xl) {  // compact rows: only the two columns each output column reads were produced, interleaved
                x0 = 2 * ox;
                x1 = 2 * ox + 1;
                lx = a.xl[ox];
            } else {
                src_index(sx, ox + a.j0, a.in_w, x0, x1, lx);
            }
        }
        const long long f = a.frame_index[t];
        const long long row_bytes = (long long)(a.xl ? 2 * a.out_w : a.in_w) * 3;
        const uint8_t* base = a.frames + f * a.src_h * row_bytes;
        // the four neighbours, 3 channels each
        int p[2][2][3];
#pragma unroll
        for (int yi = 0; yi < 2; ++yi) {
            const int y = yi ? y1 : y0;
            if constexpr (FUSE_V) {
                const int s0 = a.vbounds[2 * y], n = a.vbounds[2 * y + 1];
                const int* k = a.vkk + (long long)y * a.vksize;
                int acc[2][3];
#pragma unroll
                for (int xi = 0; xi < 2; ++xi)
#pragma unroll
                    for (int c = 0; c < 3; ++c) acc[xi][c] = 1 << (PIL_BITS - 1);
                for (int s = 0; s < n; ++s) {
                    const uint8_t* q = base + (long long)(s0 + s) * row_bytes;
                    const int w = k[s];
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        acc[0][c] += (int)q[x0 * 3 + c] * w;
                        acc[1][c] += (int)q[x1 * 3 + c] * w;
                    }
                }
#pragma unroll
                for (int xi = 0; xi < 2; ++xi)
#pragma unroll
                    for (int c = 0; c < 3; ++c) p[yi][xi][c] = clip8(acc[xi][c]);
            } else {
                const uint8_t* q = base + (long long)y * row_bytes;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    p[yi][0][c] = q[x0 * 3 + c];
                    p[yi][1][c] = q[x1 * 3 + c];
                }
            }
        }
        const float wy0 = 1.f - ly, wx0 = 1.f - lx;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v;
            if (a.identity) {
                v = u8f(p[0][0][c]);
            } else {
                // w0 * v0 + w1 * v1 as ATen writes it, un-contracted (FP_CONTRACT OFF above)
                const float top = __fadd_rn(__fmul_rn(u8f(p[0][0][c]), wx0), __fmul_rn(u8f(p[0][1][c]), lx));
                const float bot = __fadd_rn(__fmul_rn(u8f(p[1][0][c]), wx0), __fmul_rn(u8f(p[1][1][c]), lx));
                v = __fadd_rn(__fmul_rn(top, wy0), __fmul_rn(bot, ly));
            }
            a.out[((long long)c * a.T_out + t) * per_frame + r] = __fdiv_rn(__fsub_rn(v, a.mean[c]), a.stdv[c]);
        }
    }
}

int grid_for(long long threads) {
    const long long b = (threads + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

extern "C" {

int l4p_pil_coeffs(int in_size, int out_size, int* bounds, int* coeffs, int coeffs_cap, int* ksize_out) {
    if (in_size <= 0 || out_size <= 0 || !ksize_out) {
        l4p_set_error("pil_coeffs: bad sizes");
        return L4P_E_INVALID;
    }
    // Resample.c precompute_coeffs (bilinear: support 1, scaled by the down-scale factor) — double arithmetic as there
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    *ksize_out = ksize;
    if (!bounds || !coeffs) return 0;  // size query
    if ((long long)coeffs_cap < (long long)out_size * ksize) {
        l4p_set_error("pil_coeffs: coefficient buffer too small (%d < %lld)", coeffs_cap, (long long)out_size * ksize);
        return L4P_E_INVALID;
    }
    const double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        int* k = coeffs + (long long)xx * ksize;
        double w[64];
        if (xmax > 64) {  // (support > 31: a > 31x down-scale; not a video preprocessing case)
            l4p_set_error("pil_coeffs: down-scale factor too large");
            return L4P_E_INVALID;
        }
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) {
            double v = (x + xmin - center + 0.5) * ss;
            v = v < 0.0 ? -v : v;
            w[x] = v < 1.0 ? 1.0 - v : 0.0;
            ww += w[x];
        }
        int x = 0;
        for (; x < xmax; ++x) {
            const double v = ww != 0.0 ? w[x] / ww : w[x];
            k[x] = v < 0 ? (int)(-0.5 + v * (1 << PIL_BITS)) : (int)(0.5 + v * (1 << PIL_BITS));  // normalize_coeffs_8bpc
        }
        for (; x < ksize; ++x) k[x] = 0;
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    return 0;
}

int l4p_resize_index_table(int in_size, int res_size, int crop0, int out_size, int* i0, int* i1, float* lambda1) {
    if (in_size <= 0 || res_size <= 0 || crop0 < 0 || out_size <= 0 || crop0 + out_size > res_size || !i0 || !i1 || !lambda1) {
        l4p_set_error("resize_index_table: bad arguments");
        return L4P_E_INVALID;
    }
    const float scale = (float)in_size / (float)res_size;
    for (int j = 0; j < out_size; ++j) src_index(scale, j + crop0, in_size, i0[j], i1[j], lambda1[j]);  // the kernel's own rule
    return 0;
}

int l4p_pil_resample_u8(l4p_stream s, const unsigned char* src, unsigned char* dst, long long n_img, int in_h, int in_w,
                        int channels, int axis, int out_size, const int* bounds, const int* coeffs, int ksize) {
    if (n_img <= 0 || in_h <= 0 || in_w <= 0 || channels <= 0 || out_size <= 0 || ksize <= 0 || (axis != 0 && axis != 1)) {
        l4p_set_error("pil_resample_u8: bad arguments");
        return L4P_E_INVALID;
    }
    hipStream_t stream = (hipStream_t)s;
    if (axis == 1) {
        const long long rows = n_img * in_h;
        ProfScope prof(PROF_PREP, stream, "pil_h %dx%d->%d", in_h, in_w, out_size);
        const long long in_rb = (long long)in_w * channels;
        if ((uintptr_t)src % 4 == 0 && in_rb <= 32768 && ksize <= 17) {
            long long R = 32768 / in_rb;
            R = R > 16 ? 16 : R;
            const long long nblk = (rows + R - 1) / R;
            const dim3 grid((unsigned)(nblk > 16384 ? 16384 : nblk));
            const size_t lds = (size_t)(R * in_rb + 17 * channels + 16);  // + slack for the zero-weighted taps past a row's end
            auto go = [&](auto kern) {
                hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, src, dst, rows, in_w, out_size, channels, (int)R, bounds,
                                   coeffs, ksize);
            };
            if (ksize <= 3)
                go(pil_resample_h_lds_kernel<3>);
            else if (ksize <= 9)
                go(pil_resample_h_lds_kernel<9>);
            else
                go(pil_resample_h_lds_kernel<17>);
        } else if ((uintptr_t)dst % 4 == 0)
            hipLaunchKernelGGL(pil_resample_h_kernel<4>, dim3(grid_for((rows * out_size * channels + 3) / 4)), dim3(256), 0,
                               stream, src, dst, rows, in_w, out_size, channels, bounds, coeffs, ksize);
        else
            hipLaunchKernelGGL(pil_resample_h_kernel<1>, dim3(grid_for(rows * out_size * channels)), dim3(256), 0, stream, src,
                               dst, rows, in_w, out_size, channels, bounds, coeffs, ksize);
    } else {
        const int row_bytes = in_w * channels;
        ProfScope prof(PROF_PREP, stream, "pil_v %dx%d->%d", in_h, in_w, out_size);
        if (row_bytes % 4 == 0 && ((uintptr_t)src % 4 == 0) && ((uintptr_t)dst % 4 == 0))
            hipLaunchKernelGGL(pil_resample_v_kernel<4>, dim3(grid_for(n_img * out_size * (row_bytes / 4))), dim3(256), 0,
                               stream, src, dst, n_img, in_h, out_size, row_bytes, bounds, coeffs, ksize);
        else
            hipLaunchKernelGGL(pil_resample_v_kernel<1>, dim3(grid_for(n_img * out_size * (long long)row_bytes)), dim3(256),
                               0, stream, src, dst, n_img, in_h, out_size, row_bytes, bounds, coeffs, ksize);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int l4p_clip_resize_normalize(l4p_stream s, const unsigned char* frames, const int* frame_index, float* rgb_out, int T_out,
                              int in_h, int in_w, int res_h, int res_w, int crop_i0, int crop_j0, int out_h, int out_w,
                              const float* mean3, const float* std3, int src_h, const int* vbounds, const int* vcoeffs,
                              int vksize, const float* x_lambda) {
    if (T_out <= 0 || in_h <= 0 || in_w <= 0 || res_h <= 0 || res_w <= 0 || out_h <= 0 || out_w <= 0 || crop_i0 < 0 ||
        crop_j0 < 0 || crop_i0 + out_h > res_h || crop_j0 + out_w > res_w || !mean3 || !std3) {
        l4p_set_error("clip_resize_normalize: bad arguments (crop %d+%d of %d, %d+%d of %d)", crop_i0, out_h, res_h, crop_j0,
                      out_w, res_w);
        return L4P_E_INVALID;
    }
    const bool fuse = vbounds != nullptr;
    if (fuse && (!vcoeffs || vksize <= 0 || src_h <= 0)) {
        l4p_set_error("clip_resize_normalize: incomplete vertical-pass tables");
        return L4P_E_INVALID;
    }
    ClipArgs a{};
    a.frames = frames;
    a.frame_index = frame_index;
    a.out = rgb_out;
    a.T_out = T_out;
    a.in_h = in_h;
    a.in_w = in_w;
    a.src_h = fuse ? src_h : in_h;
    a.res_h = res_h;
    a.res_w = res_w;
    a.i0 = crop_i0;
    a.j0 = crop_j0;
    a.out_h = out_h;
    a.out_w = out_w;
    a.identity = (res_h == in_h && res_w == in_w) ? 1 : 0;  // l4p_dataset_mini.py:246-247
    if (a.identity && x_lambda) {
        l4p_set_error("clip_resize_normalize: compact rows make no sense without a resize");
        return L4P_E_INVALID;
    }
    a.vbounds = vbounds;
    a.vkk = vcoeffs;
    a.vksize = vksize;
    a.xl = x_lambda;
    for (int c = 0; c < 3; ++c) {
        a.mean[c] = mean3[c];
        a.stdv[c] = std3[c];
    }
    hipStream_t stream = (hipStream_t)s;
    ProfScope prof(PROF_PREP, stream, "clip_resize_normalize T%d %dx%d fuse%d", T_out, in_h, in_w, (int)fuse);
    const int grid = grid_for((long long)T_out * out_h * out_w);
    if (fuse)
        hipLaunchKernelGGL(clip_resize_normalize_kernel<true>, dim3(grid), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(clip_resize_normalize_kernel<false>, dim3(grid), dim3(256), 0, stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"
