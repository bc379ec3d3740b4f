// l4p_dpt_forward: the whole DPT decoder of one dense head (reference DPTOutputAdapter_fix.forward,
// dpt_head.py:41-86 + dpt_block.py:93-157,210-238,255-278,406-414) as ONE native call: ~45 kernel launches
// issued back to back from C++ on the caller's stream, intermediates bump-allocated from a caller-provided
// workspace.  Same kernels, same order and same arguments as the Python composition in
// l4p_amd/models/task_heads/dense_heads.py (dpt_decode), which remains as the readable statement of the graph
// and is asserted bit-identical in tests/test_encoder_dpt_gpu.py.
#include <string.h>

#include <string>

#include "engine.hpp"

namespace {

struct Bump {
    char* base;
    size_t off, cap;
    bool dry;  // size query: no memory behind it
    void* take(size_t bytes) {
        const size_t a = (off + 255) & ~(size_t)255;
        off = a + bytes;
        return dry ? (void*)1 : (off <= cap ? base + a : nullptr);
    }
};

struct Vol {  // channels-last activation [B][t][h][w][c] of engine dtype
    void* p;
    int t, h, w, c;
    long long vox(int B) const { return (long long)B * t * h * w; }
};

int splitk_for(long long M, int N, int K, int es) {  // keep in sync with l4p_amd/ops.py:splitk_for
    const int nk = (K + 128 / es - 1) / (128 / es);
    if (nk < 32) return 1;
    long long s;
    if (N >= 256) {  // 128x128 tiles, two workgroups per CU: aim at 512 workgroups (gemm_launch.inc picks the tile)
        const long long tiles = ((M + 127) / 128) * ((N + 127) / 128);
        if (tiles >= 400) return 1;
        s = 512 / tiles;
    } else {  // 128x64 tiles: ~1024 workgroups (4 per CU): one 4-wave workgroup per CU is latency-bound
        const long long tiles = ((M + 127) / 128) * ((N + 63) / 64);
        if (tiles >= 512) return 1;
        s = 1024 / tiles;
    }
    if (s > 16) s = 16;
    if (s > nk / 8) s = nk / 8;
    return s < 1 ? 1 : (int)s;
}

struct Ctx {
    const l4p_engine* e;
    hipStream_t st;
    int dt, es, B;
    Bump ws;
    std::string pre;
    int rc = 0;
    bool dry = false;

    const void* W(const std::string& k) {
        if (dry) return (const void*)1;
        const void* p = e->find(pre + k);
        if (!p && !rc) {
            l4p_set_error("weight '%s%s' was never bound", pre.c_str(), k.c_str());
            rc = L4P_E_MISSING;
        }
        return p;
    }
    void* alloc(size_t bytes) {
        void* p = ws.take(bytes);
        if (!p && !rc) {
            l4p_set_error("l4p_dpt_forward: workspace too small");
            rc = L4P_E_INVALID;
        }
        return p;
    }
    Vol vol(int t, int h, int w, int c) { return Vol{alloc((size_t)B * t * h * w * c * es), t, h, w, c}; }

    // out = act(x @ w^T + bias) on [vox][c] -> [vox][n]
    Vol dense(const Vol& x, const char* wk, const char* bk, int n) {
        Vol o = vol(x.t, x.h, x.w, n);
        if (rc || dry) return o;
        GemmParams p;
        memset(&p, 0, sizeof(p));
        p.A = x.p;
        p.lda = x.c;
        p.W = W(wk);
        p.ldw = x.c;
        p.M = (int)x.vox(B);
        p.N = n;
        p.K = x.c;
        p.bias = (const float*)W(bk);
        p.out_T = o.p;
        p.ldc = n;
        if (!rc) rc = launch_gemm(dt, 0, p, st);
        return o;
    }
    // ConvTranspose3d kernel == stride == k
    Vol convT(const Vol& x, const char* wk, const char* bk, int cout, const int k[3]) {
        Vol o = vol(x.t * k[0], x.h * k[1], x.w * k[2], cout);
        if (rc || dry) return o;
        GemmParams p;
        memset(&p, 0, sizeof(p));
        p.A = x.p;
        p.lda = x.c;
        p.W = W(wk);
        p.ldw = x.c;
        p.M = (int)x.vox(B);
        p.N = k[0] * k[1] * k[2] * cout;
        p.K = x.c;
        p.Ti = x.t;
        p.Hi = x.h;
        p.Wi = x.w;
        p.bias = (const float*)W(bk);
        p.out_T = o.p;
        p.epi = EPI_CONVT;
        p.kt = k[0];
        p.kh = k[1];
        p.kw = k[2];
        p.Cout = cout;
        if (!rc) rc = launch_gemm(dt, 0, p, st);
        return o;
    }
    // 3x3x3 conv, pad 1; optional bias / ReLU output / two T residuals / relu copy
    // ups_h / ups_w > 0: the conv reads x up-sampled (bilinear, align_corners) to ups_h x ups_w, formed in its loader
    Vol conv3(const Vol& x_in, const char* wk, const char* bk, int cout, const int s[3], int act, const void* r1, const void* r2,
              Vol* relu_copy, int ups_h = 0, int ups_w = 0) {
        Vol x = x_in;
        if (ups_h > 0) x.h = ups_h, x.w = ups_w;
        Vol o = vol((x.t - 1) / s[0] + 1, (x.h - 1) / s[1] + 1, (x.w - 1) / s[2] + 1, cout);
        if (relu_copy) *relu_copy = vol(o.t, o.h, o.w, cout);
        const long long M = o.vox(B);
        const int sk = ups_h > 0 ? 1 : splitk_for(M, cout, 27 * x.c, es);
        float* partial = sk > 1 ? (float*)alloc((size_t)sk * M * cout * 4) : nullptr;
        if (rc || dry) return o;
        GemmParams p;
        memset(&p, 0, sizeof(p));
        p.A = x.p;
        p.W = W(wk);
        p.ldw = 27 * x.c;
        p.M = (int)M;
        p.N = cout;
        p.K = 27 * x.c;
        p.Ti = x.t;
        p.Hi = x.h;
        p.Wi = x.w;
        p.Cin = x.c;
        p.To = o.t;
        p.Ho = o.h;
        p.Wo = o.w;
        p.st = s[0];
        p.sh = s[1];
        p.sw = s[2];
        p.bias = bk ? (const float*)W(bk) : nullptr;
        p.act = act;
        if (r1) {
            p.res1 = r1;
            p.res2 = r2;
            p.res_f32 = 0;
            p.ldr = cout;
        }
        p.out_T = o.p;
        p.ldc = cout;
        p.out_relu_T = relu_copy ? relu_copy->p : nullptr;
        p.splitk = sk;
        p.partial = partial;
        if (ups_h > 0) p.ups_hi = x_in.h, p.ups_wi = x_in.w;
        if (!rc) rc = launch_gemm(dt, 1, p, st);
        return o;
    }
    Vol resize(const Vol& x, int t, int h, int w) {
        if (t == x.t && h == x.h && w == x.w) return x;
        Vol o = vol(t, h, w, x.c);
        if (rc || dry) return o;
        rc = launch_upsample(dt, x.p, o.p, B, x.t, x.h, x.w, t, h, w, x.c, 1, st);
        return o;
    }
};

int run(Ctx& c, const l4p_dpt_cfg* cfg, const void* const* hooks, float* out) {
    static const int one[3] = {1, 1, 1};
    const int F = cfg->feature_dim;
    Vol lay[4], layr[4];
    for (int i = 0; i < 4; ++i) {
        Vol tok{(void*)hooks[i], cfg->nt, cfg->nh, cfg->nw, cfg->dim};
        char wk[32], bk[32];
        snprintf(wk, sizeof(wk), "act%d.0.w", i);
        snprintf(bk, sizeof(bk), "act%d.0.b", i);
        Vol a = c.dense(tok, wk, bk, cfg->layer_dims[i]);
        const int* sf = cfg->actpost[i];
        snprintf(wk, sizeof(wk), "act%d.1.w", i);
        snprintf(bk, sizeof(bk), "act%d.1.b", i);
        if (sf[0] > 0 || sf[1] > 0 || sf[2] > 0) {
            const int k[3] = {1 << sf[0], 1 << sf[1], 1 << sf[2]};
            a = c.convT(a, wk, bk, cfg->layer_dims[i], k);
        } else if (sf[0] < 0 || sf[1] < 0 || sf[2] < 0) {
            const int s[3] = {1 << -sf[0], 1 << -sf[1], 1 << -sf[2]};
            a = c.conv3(a, wk, bk, cfg->layer_dims[i], s, ACT_NONE, nullptr, nullptr, nullptr);
        }
        snprintf(wk, sizeof(wk), "rn%d.w", i);
        lay[i] = c.conv3(a, wk, nullptr, F, one, ACT_NONE, nullptr, nullptr, &layr[i]);
    }
    auto rcu = [&](int r, int u, const Vol& x, const Vol& xr, const void* extra, Vol* relu_copy) {
        char w1[40], b1[40], w2[40], b2[40];
        snprintf(w1, sizeof(w1), "ref%d.rcu%d.c1.w", r, u);
        snprintf(b1, sizeof(b1), "ref%d.rcu%d.c1.b", r, u);
        snprintf(w2, sizeof(w2), "ref%d.rcu%d.c2.w", r, u);
        snprintf(b2, sizeof(b2), "ref%d.rcu%d.c2.b", r, u);
        Vol y = c.conv3(xr, w1, b1, F, one, ACT_RELU, nullptr, nullptr, nullptr);
        return c.conv3(y, w2, b2, F, one, ACT_NONE, x.p, extra, relu_copy);
    };
    Vol path{};
    for (int r = 4; r >= 1; --r) {
        const int i = r - 1;
        Vol o, orl;
        if (r == 4) {
            o = lay[3];
            orl = layr[3];
        } else {
            Vol p4 = path;
            if (r == 3 && (p4.t != lay[2].t || p4.h != lay[2].h)) {
                l4p_set_error("l4p_dpt_forward: refinenet4 output needs cropping (unsupported token grid)");
                return L4P_E_INVALID;
            }
            o = rcu(r, 1, lay[i], layr[i], p4.p, &orl);  // path + RCU1(layer)
        }
        Vol o2 = rcu(r, 2, o, orl, nullptr, nullptr);
        char wk[32], bk[32];
        snprintf(wk, sizeof(wk), "ref%d.out.w", r);
        snprintf(bk, sizeof(bk), "ref%d.out.b", r);
        Vol oc = c.dense(o2, wk, bk, F);  // out_conv commuted in front of the (linear) up-sampling
        const int* fs = cfg->fusion[i];
        path = c.resize(oc, oc.t * fs[0], oc.h * fs[1], oc.w * fs[2]);
        if (c.rc) return c.rc;
    }
    Vol h1 = c.conv3(path, "head1.w", "head1.b", F / 2, one, ACT_NONE, nullptr, nullptr, nullptr);
    // dpt_head.py:79-84: interpolate -> head conv.  Knob conv_ups = 1 (off by default: measured slower, include/l4p_hip.h): where the
    // LDS-halo kernel's fused loader takes the shape (16-bit engines, time axis not resized, 128 output channels, whole 2 x 16 x 16
    // blocks: the full geometry) the up-sampled volume never exists
    Vol h2;
    const bool fuse = is16(c.dt) && knob(KNOB_CONV_UPS) && cfg->out_t == h1.t && (cfg->out_h != h1.h || cfg->out_w != h1.w) &&
                      cfg->last_dim == 128 && h1.c % 32 == 0 && cfg->out_t % 2 == 0 && cfg->out_h % 16 == 0 && cfg->out_w % 16 == 0 &&
                      (long long)c.B * cfg->out_t * cfg->out_h * cfg->out_w / 512 >= 192 &&
                      (long long)c.B * cfg->out_t * cfg->out_h * cfg->out_w * h1.c * 2 < (1ll << 32);
    if (fuse && !c.dry) {  // (the workspace is sized for the un-fused form: the knob may change between sizing and a forward)
        h2 = c.conv3(h1, "head2.w", "head2.b", cfg->last_dim, one, ACT_RELU, nullptr, nullptr, nullptr, cfg->out_h, cfg->out_w);
    } else {
        Vol h1u = c.resize(h1, cfg->out_t, cfg->out_h, cfg->out_w);
        h2 = c.conv3(h1u, "head2.w", "head2.b", cfg->last_dim, one, ACT_RELU, nullptr, nullptr, nullptr);
    }
    if (c.rc || c.dry) return c.rc;
    return launch_head_out(c.dt, h2.p, (const float*)c.W("out.w"), (const float*)c.W("out.b"), out,
                           (long long)h2.t * h2.h * h2.w, c.B, cfg->last_dim, cfg->out_ch, cfg->post_exp, c.st);
}

}  // namespace

extern "C" {

size_t l4p_dpt_workspace_bytes(const l4p_engine* e, const l4p_dpt_cfg* cfg, int B) {
    if (!e || !cfg || B <= 0) return 0;
    Ctx c;
    c.e = e;
    c.st = nullptr;
    c.dt = e->dtype;
    c.es = esize_of(e->dtype);
    c.B = B;
    c.ws = Bump{nullptr, 0, 0, true};
    c.dry = true;
    const void* hooks[4] = {(void*)1, (void*)1, (void*)1, (void*)1};
    run(c, cfg, hooks, nullptr);
    return c.ws.off + 256;
}

int l4p_dpt_forward(l4p_engine* e, l4p_stream stream, const char* task, const l4p_dpt_cfg* cfg, const void* const* hooks, int B,
                    void* workspace, size_t ws_bytes, float* out) {
    if (!e || !task || !cfg || !hooks || !workspace || !out || B <= 0) {
        l4p_set_error("l4p_dpt_forward: bad arguments");
        return L4P_E_INVALID;
    }
    Ctx c;
    c.e = e;
    c.st = (hipStream_t)stream;
    c.dt = e->dtype;
    c.es = esize_of(e->dtype);
    c.B = B;
    c.ws = Bump{(char*)workspace, 0, ws_bytes, false};
    c.pre = std::string("dpt.") + task + ".";
    return run(c, cfg, hooks, out);
}
}
