// C ABI of libl4p_hip.so (see include/l4p_hip.h): error plumbing, kernel-level entry points and the
// engine that runs the encoder with a single call.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <unordered_map>
#include <vector>

#include <stdlib.h>

#include "engine.hpp"

static thread_local char g_err[512] = "";

void l4p_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int launch_layernorm_res(int dtype, const float* x, int x_mod, const void* delta_T, const float* gamma, const float* beta, float eps,
                         void* out_T, float* out_f32, int M, int C, const float* add, int add_mod, void* out_T2, const float* x_shared,
                         int x_period, int x_split, hipStream_t stream, float* out_sum = nullptr, const float* part = nullptr,
                         int nsplit = 0, const float* pbias = nullptr, float* out_stats = nullptr);

extern "C" {

const char* l4p_last_error(void) { return g_err; }
int l4p_abi_version(void) { return 10; }

// A HIP stream whose kernels run on CUs [first_cu, first_cu + n_cus) only (hipExtStreamCreateWithCUMask): the sharded long-video path
// gives the tracker's latency-bound kernel chain a slice of the chip of its own, beside the chip-filling decoders on the rest.
int l4p_stream_create_cu_mask(int first_cu, int n_cus, l4p_stream* out) {
    if (!out || first_cu < 0 || n_cus < 1 || first_cu + n_cus > 1024) {
        l4p_set_error("l4p_stream_create_cu_mask: bad CU range [%d, %d)", first_cu, first_cu + n_cus);
        return L4P_E_INVALID;
    }
    uint32_t mask[32] = {0};
    for (int c = first_cu; c < first_cu + n_cus; ++c) mask[c >> 5] |= 1u << (c & 31);
    hipStream_t s = nullptr;
    HIP_TRY(hipExtStreamCreateWithCUMask(&s, (uint32_t)((first_cu + n_cus + 31) / 32), mask));
    *out = (l4p_stream)s;
    return 0;
}
int l4p_stream_destroy(l4p_stream s) {
    HIP_TRY(hipStreamDestroy((hipStream_t)s));
    return 0;
}

int l4p_gemm(l4p_stream stream, int dtype, const l4p_gemm_desc* d) {
    if (!d) {
        l4p_set_error("l4p_gemm: null descriptor");
        return L4P_E_INVALID;
    }
    return launch_gemm(dtype, 0, *d, (hipStream_t)stream);
}
int l4p_gemm_group(l4p_stream stream, int dtype, const l4p_gemm_desc* d, int n) {
    if (!d) {
        l4p_set_error("l4p_gemm_group: null descriptors");
        return L4P_E_INVALID;
    }
    return launch_gemm_group(dtype, d, n, (hipStream_t)stream);
}
int l4p_conv3d_k3(l4p_stream stream, int dtype, const l4p_gemm_desc* d) {
    if (!d) {
        l4p_set_error("l4p_conv3d_k3: null descriptor");
        return L4P_E_INVALID;
    }
    return launch_gemm(dtype, 1, *d, (hipStream_t)stream);
}
int l4p_layernorm(l4p_stream stream, int dtype, const float* x, const float* gamma, const float* beta, float eps,
                  void* out_T, float* out_f32, int M, int C) {
    return launch_layernorm(dtype, x, gamma, beta, eps, out_T, out_f32, M, C, (hipStream_t)stream);
}
int l4p_attention(l4p_stream stream, int dtype, const void* q, const void* kt, const void* vt, void* out, int B, int S,
                  int H, int Dh, float scale) {
    return launch_attention(dtype, q, kt, vt, out, B, S, H, Dh, scale, (hipStream_t)stream);
}
int l4p_patch_gather(l4p_stream stream, int dtype, const float* rgb, void* out, int B, int Cin, int T, int H, int W,
                     int pt, int ph, int pw, int Kp) {
    return launch_patch_gather(dtype, rgb, out, B, Cin, T, H, W, pt, ph, pw, Kp, (hipStream_t)stream);
}
int l4p_cast(l4p_stream stream, int dtype, const float* x, void* y, long long n) {
    return launch_cast(dtype, x, y, n, (hipStream_t)stream);
}

int l4p_upsample_trilinear(l4p_stream stream, int dtype, const void* x, void* y, int B, int Ti, int Hi, int Wi, int To,
                           int Ho, int Wo, int C, int align_corners) {
    return launch_upsample(dtype, x, y, B, Ti, Hi, Wi, To, Ho, Wo, C, align_corners, (hipStream_t)stream);
}
int l4p_head_out(l4p_stream stream, int dtype, const void* x, const float* w, const float* bias, float* y,
                 long long vox_per_b, int B, int C, int Cout, int post_exp) {
    return launch_head_out(dtype, x, w, bias, y, vox_per_b, B, C, Cout, post_exp, (hipStream_t)stream);
}

int l4p_affine_align_solve(l4p_stream stream, const float* pred, const float* target, long long n, int inverse,
                           double* scratch, float* sol) {
    return launch_affine_solve(pred, target, n, inverse, scratch, sol, (hipStream_t)stream);
}
int l4p_affine_align_apply(l4p_stream stream, const float* x, float* y, long long n, int inverse, const float* sol) {
    return launch_affine_apply(x, y, n, inverse, sol, (hipStream_t)stream);
}
int l4p_rays_to_pose(l4p_stream stream, const float* rays, const float* K, float* out, int B, int T, int h, int w,
                     int H, int W) {
    return launch_rays_to_pose(rays, K, out, B, T, h, w, H, W, (hipStream_t)stream);
}
int l4p_rays_to_pose_rot(l4p_stream stream, const float* rays, const float* R, float* out, int B, int T, int h, int w) {
    return launch_rays_to_pose_rot(rays, R, out, B, T, h, w, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// engine
// ---------------------------------------------------------------------------------------------
int l4p_create(int device, int dtype, l4p_engine** out) {
    if (!out || !dtype_ok(dtype)) {
        l4p_set_error("l4p_create: bad arguments");
        return L4P_E_INVALID;
    }
    // the engine only records its device: the caller's current device is left as it was (every launch goes to the stream the
    // caller passes, and the host side makes that stream's device current around a call)
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) {
        l4p_set_error("l4p_create: no such device");
        return L4P_E_INVALID;
    }
    l4p_engine* e = new l4p_engine();
    e->device = device;
    e->dtype = dtype;
    *out = e;
    return 0;
}
int l4p_destroy(l4p_engine* e) {
    delete e;
    return 0;
}
int l4p_bind_weight(l4p_engine* e, const char* name, const void* dev_ptr, long long numel) {
    if (!e || !name || !dev_ptr) {
        l4p_set_error("l4p_bind_weight: null argument");
        return L4P_E_INVALID;
    }
    e->w[name] = Weight{dev_ptr, numel};
    return 0;
}

int l4p_encoder_configure(l4p_engine* e, const l4p_encoder_cfg* c) {
    if (!e || !c) {
        l4p_set_error("l4p_encoder_configure: null argument");
        return L4P_E_INVALID;
    }
    const int tokens = (c->frames / c->pt) * (c->img_h / c->ph) * (c->img_w / c->pw);
    if (c->heads * c->head_dim != c->dim || c->head_dim > 96 || c->head_dim % 4 || tokens % 128 || c->dim % 8 ||
        c->mlp_hidden % 8 || c->patch_kp % 8 || c->patch_kp < c->in_chans * c->pt * c->ph * c->pw) {
        l4p_set_error("l4p_encoder_configure: unsupported geometry (dim=%d heads=%d head_dim=%d tokens=%d)", c->dim,
                      c->heads, c->head_dim, tokens);
        return L4P_E_INVALID;
    }
    e->enc = *c;
    e->enc_tokens = tokens;
    e->enc_set = true;
    return 0;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct EncWs {
    float* x;
    char *xn, *qk, *vt, *ao, *hb;
    float* sk;  // split-K partials of fc2 (small batches only)
    size_t total;
};
// fc2 (K = mlp_hidden = 6144: 96 k-tiles in a row) at batch 1 gives each CU one latency-bound chain of k-tiles; two K slices
// double the workgroups in flight (69 -> ~50 us including the reduction pass).  Not worth it once M fills the chip.
// With enough k-tiles per slice the slices go to the 8-phase kernel instead: as many slices of the 256x256 tiles as fill the
// chip about once (batch 1: 48 tiles x 4 = 192 workgroups), the form gemm_launch.inc routes to that kernel.
static int enc_fc2_splitk(const l4p_engine* e, size_t M) {
    const l4p_encoder_cfg& c = e->enc;
    const size_t tiles = ((M + 127) / 128) * (((size_t)c.dim + 63) / 64);
    if (!(is16(e->dtype) && tiles < 512 && c.mlp_hidden >= 4096)) return 1;
    static const int env8 = getenv("L4P_FC2_SPLITK8") ? atoi(getenv("L4P_FC2_SPLITK8")) : -1;  // (A/B aid: 0 = the 2-slice form, n = n slices)
    const bool no8 = env8 == 0;
    const size_t t8 = ((M + 255) / 256) * (((size_t)c.dim + 255) / 256);
    const int sk8 = env8 > 1 ? env8 : (t8 > 0 ? (int)(200 / t8) : 0);  // (48 tiles: 4 slices 72.0 us, 5 slices 73.4, 3 slices 77.7; 2 slices of 128x128 tiles 76.1)
    if (!no8 && c.dim > 128 && sk8 >= 2 && sk8 <= 8 && t8 * sk8 >= 144 && c.mlp_hidden / sk8 >= 512) return sk8;
    return 2;
}
static EncWs enc_layout(const l4p_engine* e, int B, char* base) {
    const l4p_encoder_cfg& c = e->enc;
    const size_t es = esize_of(e->dtype);
    const size_t M = (size_t)B * e->enc_tokens;
    const size_t wide = (size_t)(c.dim > c.patch_kp ? c.dim : c.patch_kp);
    EncWs w;
    size_t off = 0;
    w.x = (float*)(base + off);
    off += align256(M * c.dim * 4);
    w.xn = base + off;
    off += align256(M * wide * es);
    w.qk = base + off;
    off += align256(M * 2 * c.heads * 96 * es);
    w.vt = base + off;
    off += align256(M * c.heads * 96 * es);
    w.ao = base + off;
    off += align256(M * c.dim * es);
    w.hb = base + off;
    off += align256(M * c.mlp_hidden * es);
    w.sk = (float*)(base + off);
    if (enc_fc2_splitk(e, M) > 1) off += align256((size_t)enc_fc2_splitk(e, M) * M * c.dim * 4);
    w.total = off;
    return w;
}

size_t l4p_encoder_workspace_bytes(const l4p_engine* e, int B) {
    if (!e || !e->enc_set || B <= 0) return 0;
    return enc_layout(e, B, nullptr).total;
}

#define GETW(var, key)                                                  \
    const void* var = e->find(key);                                     \
    if (!var) {                                                         \
        l4p_set_error("weight '%s' was never bound", std::string(key).c_str()); \
        return L4P_E_MISSING;                                           \
    }

int l4p_encoder_forward(l4p_engine* e, l4p_stream stream_, const float* rgb, int B, void* workspace, size_t ws_bytes,
                        int n_taps, const int* tap_layer, float* const* tap_f32, void* const* tap_T) {
    if (!e || !e->enc_set || !rgb || !workspace || B <= 0 || n_taps <= 0) {
        l4p_set_error("l4p_encoder_forward: bad arguments");
        return L4P_E_INVALID;
    }
    hipStream_t stream = (hipStream_t)stream_;
    const l4p_encoder_cfg& c = e->enc;
    const int dt = e->dtype;
    const size_t es = esize_of(dt);
    const int S = e->enc_tokens, M = B * S, C = c.dim, H = c.heads, Dh = c.head_dim, Dp = 96;
    EncWs w = enc_layout(e, B, (char*)workspace);
    if (ws_bytes < w.total) {
        l4p_set_error("l4p_encoder_forward: workspace too small (%zu < %zu)", ws_bytes, w.total);
        return L4P_E_INVALID;
    }
    int last = 0;
    for (int i = 0; i < n_taps; ++i) {
        if (tap_layer[i] < 0 || tap_layer[i] > c.depth) {
            l4p_set_error("l4p_encoder_forward: tap layer %d out of range", tap_layer[i]);
            return L4P_E_INVALID;
        }
        if (tap_layer[i] > last) last = tap_layer[i];
    }
    int rc;
    auto emit_taps = [&](int layer) -> int {
        for (int i = 0; i < n_taps; ++i) {
            if (tap_layer[i] != layer) continue;
            if (layer == c.depth) {
                // features_list[-1] = norm(x)  (l4p_videomae.py:115)
                const void* g = e->find("enc.norm.g");
                const void* b = e->find("enc.norm.b");
                if (!g || !b) {
                    l4p_set_error("weight 'enc.norm.*' was never bound");
                    return L4P_E_MISSING;
                }
                int r = launch_layernorm(dt, w.x, (const float*)g, (const float*)b, c.ln_eps, tap_T ? tap_T[i] : nullptr,
                                         tap_f32 ? tap_f32[i] : nullptr, M, C, stream);
                if (r) return r;
            } else {
                if (tap_f32 && tap_f32[i]) {
                    hipError_t he = hipMemcpyAsync(tap_f32[i], w.x, (size_t)M * C * 4, hipMemcpyDeviceToDevice, stream);
                    if (he != hipSuccess) {
                        l4p_set_error("tap copy failed: %s", hipGetErrorString(he));
                        return L4P_E_HIP;
                    }
                }
                if (tap_T && tap_T[i]) {
                    int r = launch_cast(dt, w.x, tap_T[i], (long long)M * C, stream);
                    if (r) return r;
                }
            }
        }
        return 0;
    };

    // ---- patch embed: gather -> GEMM (+bias, + fixed sinusoid table)   modeling_finetune.py:276-283,
    //      l4p_videomae.py:99-101
    {
        GETW(pw_, "enc.patch.w");
        GETW(pb_, "enc.patch.b");
        GETW(pos_, "enc.pos");
        rc = launch_patch_gather(dt, rgb, w.xn, B, c.in_chans, c.frames, c.img_h, c.img_w, c.pt, c.ph, c.pw, c.patch_kp,
                                 stream);
        if (rc) return rc;
        GemmParams p;
        memset(&p, 0, sizeof(p));
        p.A = w.xn;
        p.lda = c.patch_kp;
        p.W = pw_;
        p.ldw = c.patch_kp;
        p.M = M;
        p.N = C;
        p.K = c.patch_kp;
        p.bias = (const float*)pb_;
        p.res1 = pos_;
        p.res_f32 = 1;
        p.ldr = C;
        p.res_mod = S;
        p.out_f32 = w.x;
        p.ldc = C;
        rc = launch_gemm(dt, 0, p, stream);
        if (rc) return rc;
    }
    rc = emit_taps(0);
    if (rc) return rc;

    const float scale = 1.0f / sqrtf((float)Dh);
    // Deferred residual sums (bf16 engine): "x = x + proj(...)" and "x = x + fc2(...)" (modeling_finetune.py:247-248) are not formed
    // in the projections' epilogues - one round of tiles there means every CU reads and writes its slice of the float stream at
    // the same time with the matrix pipe idle (measured: 27 of proj's 56 us, 38 of fc2's 143 us at batch 4) - but in the LayerNorm
    // that follows, which reads x anyway (l4p_layernorm_res: y = LN(x + delta), x updated in place).  The projection leaves
    // bias + product in the engine dtype, which is what the reference's autocast linear returns before the float sum.
    // L4P_ENC_DEFER_RES=0: the fused-epilogue form (A/B aid).  The float engine keeps the fused form.
    static const int defer_env = getenv("L4P_ENC_DEFER_RES") ? atoi(getenv("L4P_ENC_DEFER_RES")) : 1;
    const bool defer = is16(dt) && defer_env != 0 && C % 4 == 0 && C <= 1536;
    void* const delta = w.qk;  // [M][C] engine dtype, in the q / k slot (free between the attention and the next QKV projection)
    static const int sk_in_ln = getenv("L4P_ENC_SK_IN_LN") ? atoi(getenv("L4P_ENC_SK_IN_LN")) : 1;  // (0: finish pass, A/B aid)
    bool pending = false;
    int pending_sk = 0;
    const float* pending_bias = nullptr;
    char key[96];
    for (int l = 0; l < last && l < c.depth; ++l) {
#define BW(var, suffix)                                  \
    snprintf(key, sizeof(key), "enc.blk%d." suffix, l);  \
    GETW(var, key)
        BW(ln1g, "ln1.g");
        BW(ln1b, "ln1.b");
        BW(qkvw, "qkv.w");
        BW(qkvb, "qkv.b");
        BW(projw, "proj.w");
        BW(projb, "proj.b");
        BW(ln2g, "ln2.g");
        BW(ln2b, "ln2.b");
        BW(fc1w, "fc1.w");
        BW(fc1b, "fc1.b");
        BW(fc2w, "fc2.w");
        BW(fc2b, "fc2.b");
#undef BW
        // x = x + proj(attn(norm1(x)))            modeling_finetune.py:247, :169-190
        if (pending) {  // x += the previous block's MLP output (left in the engine dtype), then norm1: one pass over x
            rc = launch_layernorm_res(dt, w.x, 0, delta, (const float*)ln1g, (const float*)ln1b, c.ln_eps, w.xn, nullptr, M, C, nullptr, 0,
                                      nullptr, nullptr, 1, 0, stream, w.x);
            pending = false;
        } else if (pending_sk > 0) {  // ... left as split-K partials: bias + slices summed here instead of in a finish pass
            rc = launch_layernorm_res(dt, w.x, 0, nullptr, (const float*)ln1g, (const float*)ln1b, c.ln_eps, w.xn, nullptr, M, C,
                                      nullptr, 0, nullptr, nullptr, 1, 0, stream, w.x, w.sk, pending_sk, pending_bias);
            pending_sk = 0;
        } else {
            rc = launch_layernorm(dt, w.x, (const float*)ln1g, (const float*)ln1b, c.ln_eps, w.xn, nullptr, M, C, stream);
        }
        if (rc) return rc;
        GemmParams p;
        memset(&p, 0, sizeof(p));
        p.A = w.xn;
        p.lda = C;
        p.W = qkvw;
        p.ldw = C;
        p.M = M;
        p.N = 3 * H * Dp;
        p.K = C;
        p.bias = (const float*)qkvb;
        p.out_T = w.qk;  // q: [M][H*Dp]; k (tile order) follows in the same workspace slot
        p.k_tiled = w.qk + (size_t)M * H * Dp * es;
        p.ldc = H * Dp;
        p.epi = EPI_QKV;
        p.vt = w.vt;
        p.S = S;
        p.H = H;
        p.Dp = Dp;
        p.q_scale = scale * 1.4426950408889634f;  // q leaves the projection in the exp2 domain: one rounding of q * scale
        rc = launch_gemm(dt, 0, p, stream);
        if (rc) return rc;
        rc = launch_attention(dt, w.qk, w.qk + (size_t)M * H * Dp * es, w.vt, w.ao, B, S, H, Dh, L4P_ATTN_PRESCALED, stream);
        if (rc) return rc;
        memset(&p, 0, sizeof(p));
        p.A = w.ao;
        p.lda = C;
        p.W = projw;
        p.ldw = C;
        p.M = M;
        p.N = C;
        p.K = C;
        p.bias = (const float*)projb;
        p.ldc = C;
        if (defer) {  // (q / k are dead once the attention has run: their slot takes the projection's output)
            p.out_T = delta;
        } else {
            p.res1 = w.x;
            p.res_f32 = 1;
            p.ldr = C;
            p.out_f32 = w.x;
        }
        rc = launch_gemm(dt, 0, p, stream);
        if (rc) return rc;
        // x = x + fc2(gelu(fc1(norm2(x))))         modeling_finetune.py:248, :62-69
        if (defer)
            rc = launch_layernorm_res(dt, w.x, 0, delta, (const float*)ln2g, (const float*)ln2b, c.ln_eps, w.xn, nullptr, M, C, nullptr, 0,
                                      nullptr, nullptr, 1, 0, stream, w.x);
        else
            rc = launch_layernorm(dt, w.x, (const float*)ln2g, (const float*)ln2b, c.ln_eps, w.xn, nullptr, M, C, stream);
        if (rc) return rc;
        memset(&p, 0, sizeof(p));
        p.A = w.xn;
        p.lda = C;
        p.W = fc1w;
        p.ldw = C;
        p.M = M;
        p.N = c.mlp_hidden;
        p.K = C;
        p.bias = (const float*)fc1b;
        p.act = ACT_GELU;
        p.out_T = w.hb;
        p.ldc = c.mlp_hidden;
        rc = launch_gemm(dt, 0, p, stream);
        if (rc) return rc;
        memset(&p, 0, sizeof(p));
        p.A = w.hb;
        p.lda = c.mlp_hidden;
        p.W = fc2w;
        p.ldw = c.mlp_hidden;
        p.M = M;
        p.N = C;
        p.K = c.mlp_hidden;
        p.bias = (const float*)fc2b;
        p.ldc = C;
        p.splitk = enc_fc2_splitk(e, (size_t)M);
        p.partial = p.splitk > 1 ? w.sk : nullptr;
        // the MLP's residual rides into the NEXT block's norm1 unless x itself is needed first (a tap after this block, the last
        // block) or the split-K finish pass forms the sum anyway
        bool tap_next = l + 1 >= last;
        for (int i = 0; i < n_taps; ++i) tap_next = tap_next || tap_layer[i] == l + 1;
        if (defer && !tap_next && p.splitk <= 1) {
            p.out_T = delta;
            pending = true;
        } else if (defer && sk_in_ln && !tap_next && p.splitk > 1 && p.splitk <= 16) {
            // batch 1 / 2: the K slices' float partials stay in w.sk and the next norm1 sums them (no finish pass: it read the
            // partials and read + wrote x only for the LayerNorm to read x again)
            p.tuning |= 2;
            pending_sk = p.splitk;
            pending_bias = (const float*)fc2b;
        } else {
            p.res1 = w.x;
            p.res_f32 = 1;
            p.ldr = C;
            p.out_f32 = w.x;
        }
        rc = launch_gemm(dt, 0, p, stream);
        if (rc) return rc;
        if (l + 1 < c.depth) {
            rc = emit_taps(l + 1);
            if (rc) return rc;
        }
    }
    if (last == c.depth) {
        rc = emit_taps(c.depth);
        if (rc) return rc;
    }
    (void)es;
    return 0;
}

}  // extern "C"
