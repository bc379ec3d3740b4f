// l4p_track_window_forward: one window of the SAM-style tracker for the N queries of a clip
// (VideoMAETrack2DSamHead.forward / forward_single_batch, sparse_heads.py:497-667; PromptEncoder prompt_encoder.py:67-121;
// TwoWayTransformer sam/transformer.py:67-111,156-187; MaskDecoder.predict_masks mask_decoder.py:101-141; memory tokens
// sparse_heads.py:406-448,660-665) as ONE native call: ~125 kernel launches issued back to back from C++ on the caller's
// stream, intermediates bump-allocated from a caller-provided workspace.  Same kernels, same order, same arguments as the
// Python composition l4p_amd/models/task_heads/sparse_heads.py:_window (which remains as the readable statement of the
// graph, selectable with L4P_TRACK_PYTHON=1, and is asserted bit-identical in tests/test_track_gpu.py).  What it buys: the
// host issues one call per clip and window instead of ~125 ctypes calls with their tensor allocations — with 8 ranks
// sharing one host's cores that is what keeps the step from becoming launch-bound.
#include <stdlib.h>
#include <string.h>

#include <string>

#include "engine.hpp"

int launch_layernorm_ex(int dtype, const float* x, const float* gamma, const float* beta, float eps, void* out_T,
                        float* out_f32, int M, int C, const float* add, int add_mod, void* out_T2, int act,
                        hipStream_t stream);
int launch_layernorm_T(int dtype, const void* x_T, const float* gamma, const float* beta, float eps, void* out_T, int M, int C,
                       int act, hipStream_t stream);
int launch_layernorm_res(int dtype, const float* x, int x_mod, const void* delta_T, const float* gamma, const float* beta, float eps,
                         void* out_T, float* out_f32, int M, int C, const float* add, int add_mod, void* out_T2, const float* x_shared,
                         int x_period, int x_split, hipStream_t stream, float* out_sum = nullptr, const float* part = nullptr,
                         int nsplit = 0, const float* pbias = nullptr, float* out_stats = nullptr);
int launch_layernorm_chain(int dtype, const float* xs, int x_mod, const void* dprev_T, const float* stats, const float* g0, const float* b0,
                           const void* delta_T, const float* gamma, const float* beta, float eps, void* out_T, float* out_f32, int M, int C,
                           const float* add, int add_mod, void* out_T2, hipStream_t stream);
int launch_track_tokens(const float* queries, const float* labels, const float* pfeat, const float* plabel,
                        const float* gauss, const float* mask_tokens, const float* pe0, const float* pe1,
                        const float* nap, const float* fe0, const float* fe1, float* tokens, int N, int C, int T, int H,
                        int W, hipStream_t stream, int dtype = 0, void* tokens_T = nullptr);
int launch_track_keys_init(int dtype, const float* enc, const float* hist, const float* pos, float* k32, void* kT,
                           void* kP, int N, int P, int C, int shared_from, float* k32_shared, hipStream_t stream);
int launch_fill_rows(float* out, const float* v, long long rows, int C, long long group_rows, long long group_stride,
                     long long group_off, hipStream_t stream);
int launch_broadcast_block(void* base, long long off, long long bytes, long long stride, int n, hipStream_t stream);
int launch_small_attn(int dtype, int kind, const void* q, const void* k, const void* v, void* out, int N, int P, int D,
                      int heads, hipStream_t stream);
int launch_mask_gather(const float* partial, float* masks, int N, int T, int h, int w, int cpt, hipStream_t stream);
int launch_i2t_probs(int dtype, const float* s, long long lds_, int pairs, const float* cbias, int rows_per_group, void* p, int ldp,
                     long long M, int heads, int tokens, hipStream_t stream);
int launch_split_hilo(int dtype, const float* in, void* out, int G, int R, long long C, hipStream_t stream);
int launch_t2i_attn_scores(int dtype, const float* scores, long long ld_scores, const void* v, void* out, int N, int P, int D, int heads,
                           hipStream_t stream);
int launch_transpose_pad(int dtype, const void* in, void* out, int G, int R, int C, int Rp, hipStream_t stream);
int launch_i2t_delta(int dtype, const void* probs, const void* vt, const float* bias, void* delta, int N, int P, int C, int K,
                     hipStream_t stream);
int launch_t2i_probs(int dtype, const float* scores, long long ld_scores, void* probs, float* stats, int N, int P, int HT, hipStream_t stream);
int launch_t2i_context(int dtype, const void* probs, const float* stats, const void* keys, void* ctx, int N, int P, int C, int heads,
                       int tokens, long long Rg, int shared_from, hipStream_t stream);
int launch_track_readout(const float* masks, float* traj, float* vis, float* depth, int N, int T, int h, int w, int H,
                         int W, hipStream_t stream);

namespace {

struct Stack {  // bump allocator with LIFO release (mark / release) and a high-water mark for the size query
    char* base;
    size_t off, cap, peak;
    bool dry;
    void* take(size_t bytes) {
        const size_t a = (off + 255) & ~(size_t)255;
        off = a + bytes;
        if (off > peak) peak = off;
        return dry ? (void*)256 : (off <= cap ? base + a : nullptr);
    }
};

struct TW {
    const l4p_engine* e;
    hipStream_t st;
    int dt, es;
    Stack ws;
    int rc = 0;
    bool dry = false;

    const void* W(const std::string& k) {
        if (dry) return (const void*)256;
        const void* p = e->find("trk." + k);
        if (!p && !rc) {
            l4p_set_error("weight 'trk.%s' was never bound", k.c_str());
            rc = L4P_E_MISSING;
        }
        return p;
    }
    const float* Wf(const std::string& k) { return (const float*)W(k); }
    void* alloc(size_t bytes) {
        void* p = ws.take(bytes);
        if (!p && !rc) {
            l4p_set_error("l4p_track_window_forward: workspace too small");
            rc = L4P_E_INVALID;
        }
        return p;
    }
    float* f32(long long rows, int cols) { return (float*)alloc((size_t)rows * cols * 4); }
    void* T(long long rows, int cols) { return alloc((size_t)rows * cols * es); }

    // out = act(a[M][K] (row stride lda) @ w[n][K]^T + bias) (+ res1, float) -> out_T and / or out_f32, row stride ldc
    void gemm(const void* a, long long M, int K, long long lda, const std::string& wk, int n, bool bias, int act, const float* res1,
              int res_mod, float* out_f32, void* out_T, long long ldc, const int* a_map = nullptr, const int* c_map = nullptr) {
        if (rc || dry) return;
        GemmParams p = desc(a, M, K, lda, wk, n, bias, act, res1, res_mod, out_f32, out_T, ldc, a_map, c_map);
        if (!rc) rc = launch_gemm(dt, 0, p, st);
    }
    // several INDEPENDENT projections as one launch (l4p_gemm_group: bit-identical to separate launches)
    void group(const GemmParams* d, int n) {
        if (rc || dry) return;
        rc = launch_gemm_group(dt, d, n, st);
    }
    GemmParams desc(const void* a, long long M, int K, long long lda, const std::string& wk, int n, bool bias, int act, const float* res1,
                    int res_mod, float* out_f32, void* out_T, long long ldc, const int* a_map = nullptr, const int* c_map = nullptr) {
        GemmParams p;
        memset(&p, 0, sizeof(p));
        if (dry) return p;
        p.A = a;
        p.lda = lda;
        p.W = W(wk + ".w");
        p.ldw = K;
        p.M = (int)M;
        p.N = n;
        p.K = K;
        p.bias = bias ? Wf(wk + ".b") : nullptr;
        p.act = act;
        if (res1) {
            p.res1 = res1;
            p.res_f32 = 1;
            p.ldr = n;
            p.res_mod = res_mod;
        }
        p.out_f32 = out_f32;
        p.out_T = out_T;
        p.ldc = ldc;
        p.epi = EPI_DENSE;
        if (a_map) {
            p.a_gr = a_map[0];
            p.a_gs = a_map[1];
            p.a_go = a_map[2];
        }
        if (c_map) {
            p.c_gr = c_map[0];
            p.c_gs = c_map[1];
            p.c_go = c_map[2];
        }
        return p;
    }
    // x[M][K] T -> new T [M][n] = act(x W^T + b)
    void* proj(const void* x, long long M, int K, const std::string& wk, int n, int act = ACT_NONE) {
        void* o = T(M, n);
        gemm(x, M, K, K, wk, n, true, act, nullptr, 0, nullptr, o, n);
        return o;
    }
    // LayerNorm with the tracker's extras (see l4p_layernorm_ex)
    void ln(const float* x32, const std::string& key, float eps, void* oT, float* o32, long long M, int Cc, const float* add,
            int add_mod, void* oT2, int act) {
        if (rc || dry) return;
        rc = launch_layernorm_ex(dt, x32, Wf(key + ".g"), Wf(key + ".b"), eps, oT, o32, (int)M, Cc, add, add_mod, oT2, act, st);
    }
    void attn(int kind, const void* q, const void* k, const void* v, void* out, int N, int P, int D, int heads) {
        if (rc || dry) return;
        rc = launch_small_attn(dt, kind, q, k, v, out, N, P, D, heads, st);
    }
};

int run(TW& c, const l4p_track_cfg& g, const float* enc_last, float* hist, const float* q_off, const float* labels,
        const float* pfeat, const float* plabel, int N, int need_history, int hist_uniform, float* traj, float* vis, float* depth,
        float* new_pfeat) {
    const int P = g.tokens, Cc = g.dim, Dh = Cc / 2;
    const int T = g.T, H = g.H, W = g.W;
    const long long NP = (long long)N * P;
    // hist_uniform == 2: rows [P/2, P) of every track's keys are the same until the first image -> token update
    const bool half_shared = hist_uniform == 2 && N > 1 && P % 2 == 0;
    // hist_uniform == 2 / 4: a later window of a recursion - layer 0's token -> image attention in the folded form of the later layers
    // (the per-(track, head) attention kernel on projected keys / values is 120 us per window whatever the number of tracks, and the two
    // projections of N x P key rows go with it: 8 tracks 240 -> 110 us).  A window evaluated out of context (0) keeps the projected
    // form, which the first window's shared-key shortcut equals bit for bit.  L4P_TRACK_FOLD_L0=0: A/B aid.
    static const bool fold_l0_env = !(getenv("L4P_TRACK_FOLD_L0") && atoi(getenv("L4P_TRACK_FOLD_L0")) == 0);
    const bool fold_l0 = fold_l0_env && (hist_uniform == 2 || hist_uniform == 4);
    if (hist_uniform == 2 || hist_uniform == 4) hist_uniform = 0;
    // projection of per-track keys x [N*P][Cc] whose second temporal half is common to all tracks: the first halves of all
    // tracks (row-mapped GEMM over N * P/2 rows), track 0's second half, and a copy of that block to the other tracks
    auto proj_half_shared = [&](const void* x, const std::string& wk, int n) -> void* {
        void* o = c.T(NP, n);
        const int half = P / 2;
        const int m1[3] = {half, P, 0}, m2[3] = {half, P, half};
        c.gemm(x, (long long)N * half, Cc, Cc, wk, n, true, ACT_NONE, nullptr, 0, nullptr, o, n, m1, m1);
        c.gemm(x, half, Cc, Cc, wk, n, true, ACT_NONE, nullptr, 0, nullptr, o, n, m2, m2);
        if (!c.rc && !c.dry)
            c.rc = launch_broadcast_block(o, (long long)half * n * c.es, (long long)half * n * c.es, (long long)P * n * c.es, N, c.st);
        return o;
    };
    // ---- prompt tokens (prompt_encoder.py:78-121,196-203; mask_decoder.py:107-113) ----
    float* tok32 = c.f32(6ll * N, Cc);
    void* tokT = c.T(6ll * N, Cc);
    if (!c.rc && !c.dry) {
        c.rc = launch_track_tokens(q_off, labels, pfeat, plabel, c.Wf("gauss"), c.Wf("mask_tokens"), c.Wf("point_emb0"),
                                   c.Wf("point_emb1"), c.Wf("not_a_point"), c.Wf("feat_emb0"), c.Wf("feat_emb1"), tok32, N, Cc, T, H, W,
                                   c.st, c.dt, tokT);  // (both forms of the tokens from one launch: the values l4p_cast gives)
    }
    // ---- keys = enc_features[-1] + history (sparse_heads.py:341-346); one shared [P][C] set while every track still has
    //      the same history rows (first window), the per-track set from the first image -> token update on ----
    const float* pos = c.Wf("dense_pe");
    int Nk = hist_uniform ? 1 : N;
    float* ks32 = nullptr;
    void *ksT = nullptr, *ksP = nullptr;
    if (Nk == 1 && N > 1) {
        ks32 = c.f32(P, Cc);
        ksT = c.T(P, Cc);
        ksP = c.T(P, Cc);
    }
    float* k32 = c.f32(NP, Cc);
    void* kT = c.T(NP, Cc);
    void* kP = c.T(NP, Cc);
    const bool start_shared = Nk == 1 && N > 1;
    // half_shared: rows [P/2, P) are formed for track 0 only (proj_half_shared reads no others); their float master, which the
    // first layer's "keys = norm4(keys + ...)" needs for every track, lives in kh32 (that LayerNorm runs in place over k32)
    float* kh32 = half_shared ? c.f32(P / 2, Cc) : nullptr;
    if (!c.rc && !c.dry)
        c.rc = launch_track_keys_init(c.dt, enc_last, hist, pos, start_shared ? ks32 : k32, start_shared ? ksT : kT,
                                      start_shared ? ksP : kP, Nk, P, Cc, half_shared ? P / 2 : 0, kh32, c.st);
    float* cur32 = start_shared ? ks32 : k32;  // the key set the next projection reads
    void* curT = start_shared ? ksT : kT;
    void* curP = start_shared ? ksP : kP;

    const float* q32 = nullptr;
    const void* qT = tokT;
    const void* qP = tokT;
    float* x32 = c.f32(6ll * N, Cc);
    auto ln_tokens = [&](const std::string& key) {  // (q32, qT, qP) = LN(x32), T(LN), T(LN + token PE)
        float* o32 = c.f32(6ll * N, Cc);
        void* oT = c.T(6ll * N, Cc);
        void* oP = c.T(6ll * N, Cc);
        c.ln(x32, key, 1e-5f, oT, o32, 6ll * N, Cc, tok32, 6 * N, oP, ACT_NONE);
        q32 = o32;
        qT = oT;
        qP = oP;
    };
    // Token -> image attention with the keys' projection folded into the tokens (packing.py fold_t2i; every track owns all rows of
    // its keys): Q' = q_tok x kfold^T [N][HT][C], scores = kP x Q'^T (row-grouped weights), softmax over the P keys and P.V in
    // l4p_t2i_attn_scores.  The [N * P, C/2] key projection (2 * P * C * C/2 FLOP per track, 4.06 GF) and its tensor disappear; the
    // value projection: see fold_v below.
    static const bool fold_t2i_env = !(getenv("L4P_TRACK_FOLD_T2I") && atoi(getenv("L4P_TRACK_FOLD_T2I")) == 0);
    const int HTk = 6 * g.sam_heads;
    const bool fold_t2i_ok = fold_t2i_env && P % 128 == 0 && HTk <= 64;
    // ... and the VALUE projection too (l4p_t2i_context: out = (probs x keys) Wv_h^T + bv_h): softmax to probs [N][P][HT]
    // (l4p_t2i_probs), the context of every (token, head) against the keys WITHOUT the positional term - one read of the keys, MFMA -
    // and the 48 context rows of each track through their head's block of W_v (row-grouped weights over head-major rows, every group
    // writing its own column block of `ta`: o_gs).  Another 4.06 GF and a [P, C/2] tensor per track gone.  L4P_TRACK_FOLD_T2I_V=0:
    // the projected values + l4p_t2i_attn_scores.
    static const bool fold_v_env = !(getenv("L4P_TRACK_FOLD_T2I_V") && atoi(getenv("L4P_TRACK_FOLD_T2I_V")) == 0);
    const bool fold_v = fold_t2i_ok && fold_v_env && HTk == 48 && Cc % 128 == 0 && (Dh / g.sam_heads) % 8 == 0 && P % 32 == 0 && P >= 96 && P <= 4096;
    const long long RgT = (6ll * N + 127) / 128 * 128;  // rows of a head group of the context (and of `ta`, whose rows past 6 N are scratch)
    // the folded weights are block-structured (packing.py fold_i2t / fold_t2i: head h's C columns meet head h's C/2/heads inputs only):
    // their products walk only the k-tiles of a tile's head (l4p_gemm_desc.kw_cols: bit-identical, a quarter of the weight bytes)
    static const bool kwin_env = !(getenv("L4P_TRACK_KWIN") && atoi(getenv("L4P_TRACK_KWIN")) == 0);
    const int kw_cols = kwin_env && Cc % 128 == 0 ? Cc : 0, kw_len = Dh / g.sam_heads;
    // hs: rows [P/2, P) of keysP / keysT exist for track 0 only (a later window's layer 0): the scores of the two halves are two
    // row-mapped launches, the context product reads those rows from track 0, the projected values (fold_v off) are formed once and copied
    auto t2i_folded = [&](const void* tq, const std::string& prefix, const void* keysP, const void* keysT, void* ta, bool hs) {
        const long long KW = (long long)g.sam_heads * Cc;
        void* qf = c.T((long long)N * HTk + 128, Cc);  // Q' [N][HT][C] (+ slack rows under the last tile)
        if (!c.rc && !c.dry) {
            GemmParams pq = c.desc(tq, 6ll * N, Dh, Dh, prefix + ".kfold", (int)KW, false, ACT_NONE, nullptr, 0, nullptr, qf, KW);
            pq.kw_cols = kw_cols, pq.kw_len = kw_len;
            if (!c.rc) c.rc = launch_gemm(c.dt, 0, pq, c.st);
        }
        float* sc = c.f32(NP, HTk);
        if (!c.rc && !c.dry) {
            GemmParams p;
            memset(&p, 0, sizeof(p));
            p.A = keysP, p.lda = Cc, p.W = qf, p.ldw = Cc, p.M = (int)NP, p.N = HTk, p.K = Cc;
            p.out_f32 = sc, p.ldc = HTk, p.epi = EPI_DENSE;
            p.w_gr = P, p.w_gs = (long long)HTk * Cc, p.b_gs = 0;
            if (hs) {
                const int half = P / 2;
                p.M = N * half, p.w_gr = half;
                p.a_gr = half, p.a_gs = P, p.a_go = 0, p.c_gr = half, p.c_gs = P, p.c_go = 0;  // first halves: every track's own
                c.rc = launch_gemm(c.dt, 0, p, c.st);
                p.a_gs = 0, p.a_go = half, p.c_go = half;                                      // second halves: track 0's rows
                if (!c.rc) c.rc = launch_gemm(c.dt, 0, p, c.st);
            } else {
                c.rc = launch_gemm(c.dt, 0, p, c.st);
            }
        }
        if (fold_v) {
            const int hd = Dh / g.sam_heads;
            void* pr = c.T(NP, HTk);
            // (measured at full size, tools/probes/foldv_precision.py: keeping the context rows as [hi | lo] bf16 pairs changes nothing;
            //  what does matter is that the softmax sums are those of the ROUNDED terms, see t2i_probs_kernel)
            void* cx = c.T((long long)g.sam_heads * RgT, Cc);
            float* stt = c.f32((long long)N * ((P + 255) / 256), 2 * HTk);  // per 256-key split: column maxima, sums
            if (!c.rc && !c.dry) c.rc = launch_t2i_probs(c.dt, sc, HTk, pr, stt, N, P, HTk, c.st);
            if (!c.rc && !c.dry) c.rc = launch_t2i_context(c.dt, pr, stt, keysT, cx, N, P, Cc, g.sam_heads, 6, RgT, hs ? P / 2 : P, c.st);
            if (!c.rc && !c.dry) {
                GemmParams p = c.desc(cx, (long long)g.sam_heads * RgT, Cc, Cc, prefix + ".v", hd, true, ACT_NONE, nullptr, 0, nullptr, ta, Dh);
                p.w_gr = (int)RgT, p.w_gs = (long long)hd * Cc, p.b_gs = hd, p.o_gs = hd;
                p.c_gr = (int)RgT, p.c_gs = 0, p.c_go = 0;
                if (!c.rc) c.rc = launch_gemm(c.dt, 0, p, c.st);
            }
        } else {
            void* tv = hs ? proj_half_shared(keysT, prefix + ".v", Dh) : c.proj(keysT, NP, Cc, prefix + ".v", Dh);
            if (!c.rc && !c.dry) c.rc = launch_t2i_attn_scores(c.dt, sc, HTk, tv, ta, N, P, Dh, g.sam_heads, c.st);
        }
    };
    // chained key LayerNorm (see the image -> token block below): L4P_TRACK_LN_CHAIN=0 keeps the float key master
    static const bool chain_env = !(getenv("L4P_TRACK_LN_CHAIN") && atoi(getenv("L4P_TRACK_LN_CHAIN")) == 0);
    const bool may_chain = chain_env && start_shared && g.sam_depth >= 2 && Cc <= 1536;
    float* chain_stats = may_chain ? c.f32(NP, 2) : nullptr;
    // (a third layer would need the chained layer's float result: with the master's storage holding layer 0's update it gets its own)
    float* kc32 = may_chain && g.sam_depth > 2 ? c.f32(NP, Cc) : nullptr;
    float* cur32_after_chain = nullptr;
    bool chained = false;
    std::string chain_norm;
    for (int l = 0; l < g.sam_depth; ++l) {
        const std::string lo = "l" + std::to_string(l) + ".";
        const bool shared = Nk == 1 && N > 1;  // keys still common to all tracks (only in layer 0 of a first window)
        // --- self attention of the prompt tokens (transformer.py:159-166) ---
        const size_t mark_self = c.ws.off;
        {
            void *sq = c.T(6ll * N, Cc), *sk = c.T(6ll * N, Cc), *sv = c.T(6ll * N, Cc);
            const GemmParams qkv[3] = {
                c.desc(qP, 6ll * N, Cc, Cc, lo + "self.q", Cc, true, ACT_NONE, nullptr, 0, nullptr, sq, Cc),
                c.desc(qP, 6ll * N, Cc, Cc, lo + "self.k", Cc, true, ACT_NONE, nullptr, 0, nullptr, sk, Cc),
                c.desc(qT, 6ll * N, Cc, Cc, lo + "self.v", Cc, true, ACT_NONE, nullptr, 0, nullptr, sv, Cc)};
            c.group(qkv, 3);
            void* sa = c.T(6ll * N, Cc);
            c.attn(0, sq, sk, sv, sa, N, 6, Cc, g.sam_heads);
            c.gemm(sa, 6ll * N, Cc, Cc, lo + "self.out", Cc, true, ACT_NONE, q32, 0, x32, nullptr, Cc);
        }
        c.ws.off = mark_self;
        ln_tokens(lo + "norm1");
        // --- tokens -> image (transformer.py:168-173) ---
        size_t mark = c.ws.off;
        {
            void* tq = c.proj(qP, 6ll * N, Cc, lo + "t2i.q", Dh);
            const bool hs = half_shared && l == 0;
            const bool folded = fold_t2i_ok && (l >= 1 || fold_l0) && !shared && (!hs || P % 256 == 0);
            void* tv = folded ? nullptr : hs ? proj_half_shared(curT, lo + "t2i.v", Dh) : c.proj(curT, (long long)Nk * P, Cc, lo + "t2i.v", Dh);
            void* ta = c.T(RgT, Dh);
            // (layer 0 of a FIRST window - keys still common to all tracks - and of a window evaluated out of context keeps the
            //  projected form: the projection of the common rows is one small GEMM, and the per-track evaluation of the same window
            //  (the equality test of that shortcut) stays bit-identical to it; layer 0 of a later window: see fold_l0 above)
            if (folded) {
                t2i_folded(tq, lo + "t2i", curP, curT, ta, hs);
            } else {
                void* tk = hs ? proj_half_shared(curP, lo + "t2i.k", Dh) : c.proj(curP, (long long)Nk * P, Cc, lo + "t2i.k", Dh);
                c.attn(shared ? 3 : 1, tq, tk, tv, ta, N, P, Dh, g.sam_heads);
            }
            c.gemm(ta, 6ll * N, Dh, Dh, lo + "t2i.out", Cc, true, ACT_NONE, q32, 0, x32, nullptr, Cc);
        }
        c.ws.off = mark;
        ln_tokens(lo + "norm2");
        // --- MLP (transformer.py:175-178), ReLU ---
        mark = c.ws.off;
        {
            void* hdn = c.proj(qT, 6ll * N, Cc, lo + "mlp1", g.sam_mlp, ACT_RELU);
            c.gemm(hdn, 6ll * N, g.sam_mlp, g.sam_mlp, lo + "mlp2", Cc, true, ACT_NONE, q32, 0, x32, nullptr, Cc);
        }
        c.ws.off = mark;
        ln_tokens(lo + "norm3");
        // --- image -> tokens (transformer.py:180-185): keys are updated in place ---
        mark = c.ws.off;
        {
            void *ik = c.T(6ll * N, Dh), *iv = c.T(6ll * N, Dh);
            const GemmParams kv[2] = {c.desc(qP, 6ll * N, Cc, Cc, lo + "i2t.k", Dh, true, ACT_NONE, nullptr, 0, nullptr, ik, Dh),
                                      c.desc(qT, 6ll * N, Cc, Cc, lo + "i2t.v", Dh, true, ACT_NONE, nullptr, 0, nullptr, iv, Dh)};
            c.group(kv, 2);
            // keys = norm4(keys + out_proj(attention)): the update leaves the projection in the engine dtype (`delta`) and the
            // LayerNorm forms the sum (l4p_layernorm_res): the float key stream is read once per layer instead of read + written
            // by the projection's epilogue and read again.  While the keys are still common to all tracks the float residual is
            // row m % P of the common set.
            // Chained LayerNorm (first window: this layer's float residual is the SHARED key set): the layer stores its update, its
            // engine-dtype results and (mean, rstd) per row; the next layer re-derives this layer's float result from those
            // (l4p_layernorm_chain) - the float key master is neither written nor read, and its storage holds this layer's update.
            const bool chain_next = chain_env && shared && l + 1 < g.sam_depth && Cc <= 1536;
            void* delta = chain_next ? (void*)k32 : c.T(NP, Cc);
            const int HT = 6 * g.sam_heads;               // (token, head) pairs of a track: the folded k dimension
            const int HTp = (HT + 63) / 64 * 64;
            // Folded form (packing.py fold_i2t; every track owns its keys, all rows of kP exist): the 2048 x N image tokens never
            // pass through i2t.q / i2t.out.  Token side: K' = k_tok x qfold^T [N][HT][C], c = k_tok x cfold^T [N][HT],
            // V' = v_tok x ofold^T [N][HT][C] (three small GEMMs), V'^T [N][C][HTp].  Image side: scores = kP x K'^T + c (row-grouped
            // weights: a track's rows meet that track's K'), softmax over the tokens of each head, delta = P x V' + b_out.
            // Per 64 tracks: 0.37 GB read + 0.37 GB written and 35 GFLOP, where the projections moved 1.3 GB and 520 GFLOP.
            static const bool fold_env = !(getenv("L4P_TRACK_FOLD_I2T") && atoi(getenv("L4P_TRACK_FOLD_I2T")) == 0);
            // (keys still common to all tracks: the same GEMM on row m % P of the common set; later windows, layer 0: the second
            //  temporal half of every track is track 0's - two launches over the half blocks, row-mapped like proj_half_shared)
            const bool hs0 = half_shared && l == 0;
            const bool fold = fold_env && P % 256 == 0 && HT <= 64;
            if (fold) {
                const long long KW = (long long)g.sam_heads * Cc;  // columns of K' / V' per token row
                // optional (bf16 engine): K' as a PAIR of bf16 matrices (rows [0, HT) = hi, [HT, 2 HT) = lo per track; l4p_split_hilo)
                // for its 1408-term products with the keys; the two score halves and c are summed in l4p_i2t_probs.
                // MEASURED (tools/probes/fold_precision.py, per-track distance of the bf16 engine from the f32 engine over 4 - 6 windows):
                // with K' rounded ONCE to bf16 the folded form is already as close to f32 as the projected form (traj 3e-4, depth
                // 6e-3 per track, both forms) - the pair changes nothing and is off (L4P_TRACK_FOLD_PAIR=1 turns it on).
                static const bool pair_env = getenv("L4P_TRACK_FOLD_PAIR") && atoi(getenv("L4P_TRACK_FOLD_PAIR")) == 1;
                const bool pair = pair_env && is16(c.dt);
                const int NS = pair ? 2 * HT : HT;               // score columns
                float* kf32 = pair ? c.f32(6ll * N, (int)KW) : nullptr;
                void* kf = c.T((long long)N * NS + 128, Cc);     // K' [N][NS][C] (+ slack rows under the last tile)
                void* vf = c.T(6ll * N * g.sam_heads, Cc);
                float* cf = c.f32(6ll * N, g.sam_heads);
                void* vt = c.T((long long)N * Cc + 128, HTp);    // V'^T [N][C][HTp] (+ slack rows)
                GemmParams tk[3] = {
                    c.desc(ik, 6ll * N, Dh, Dh, lo + "i2t.qfold", (int)KW, false, ACT_NONE, nullptr, 0, pair ? kf32 : nullptr, pair ? nullptr : kf, KW),
                    c.desc(iv, 6ll * N, Dh, Dh, lo + "i2t.ofold", (int)KW, false, ACT_NONE, nullptr, 0, nullptr, vf, KW),
                    c.desc(ik, 6ll * N, Dh, Dh, lo + "i2t.cfold", g.sam_heads, false, ACT_NONE, nullptr, 0, cf, nullptr, g.sam_heads)};
                tk[0].kw_cols = tk[1].kw_cols = kw_cols;
                tk[0].kw_len = tk[1].kw_len = kw_len;
                c.group(tk, 3);
                if (pair && !c.rc && !c.dry) c.rc = launch_split_hilo(c.dt, kf32, kf, N, HT, Cc, c.st);
                if (!c.rc && !c.dry) c.rc = launch_transpose_pad(c.dt, vf, vt, N, HT, Cc, HTp, c.st);
                float* sc = c.f32(NP, NS);
                void* pr = c.T(NP, HTp);
                if (!c.rc && !c.dry) {
                    GemmParams p;
                    memset(&p, 0, sizeof(p));
                    p.A = curP, p.lda = Cc, p.W = kf, p.ldw = Cc, p.M = (int)NP, p.N = NS, p.K = Cc;
                    p.out_f32 = sc, p.ldc = NS, p.epi = EPI_DENSE;
                    p.w_gr = P, p.w_gs = (long long)NS * Cc, p.b_gs = 0;
                    if (shared) {
                        p.a_gr = P, p.a_gs = 0, p.a_go = 0;  // every track reads the common key rows
                        c.rc = launch_gemm(c.dt, 0, p, c.st);
                    } else if (hs0) {
                        const int half = P / 2;
                        p.M = N * half, p.w_gr = half;
                        p.a_gr = half, p.a_gs = P, p.a_go = 0, p.c_gr = half, p.c_gs = P, p.c_go = 0;  // first halves: every track's own
                        c.rc = launch_gemm(c.dt, 0, p, c.st);
                        p.a_gs = 0, p.a_go = half, p.c_go = half;                                      // second halves: track 0's rows
                        if (!c.rc) c.rc = launch_gemm(c.dt, 0, p, c.st);
                    } else {
                        c.rc = launch_gemm(c.dt, 0, p, c.st);
                    }
                    if (!c.rc) c.rc = launch_i2t_probs(c.dt, sc, NS, pair ? 1 : 0, cf, P, pr, HTp, NP, g.sam_heads, 6, c.st);
                    static const bool delta_env = !(getenv("L4P_TRACK_DELTA_KERNEL") && atoi(getenv("L4P_TRACK_DELTA_KERNEL")) == 0);
                    if (!c.rc && delta_env && is16(c.dt) && HTp == 64 && Cc % 128 == 0 && P % 16 == 0) {
                        // (its own streaming kernel, bit-identical to the GEMM below: see i2t_delta_kernel)
                        c.rc = launch_i2t_delta(c.dt, pr, vt, c.Wf(lo + "i2t.out.b"), delta, N, P, Cc, HTp, c.st);
                    } else if (!c.rc) {
                        memset(&p, 0, sizeof(p));
                        p.A = pr, p.lda = HTp, p.W = vt, p.ldw = HTp, p.M = (int)NP, p.N = Cc, p.K = HTp;
                        p.bias = c.Wf(lo + "i2t.out.b"), p.out_T = delta, p.ldc = Cc, p.epi = EPI_DENSE;
                        p.w_gr = P, p.w_gs = (long long)Cc * HTp, p.b_gs = 0;
                        c.rc = launch_gemm(c.dt, 0, p, c.st);
                    }
                }
            } else {
                void* iq = half_shared && l == 0 ? proj_half_shared(curP, lo + "i2t.q", Dh)
                                                 : c.proj(curP, (long long)Nk * P, Cc, lo + "i2t.q", Dh);
                void* ia = c.T(NP, Dh);
                c.attn(shared ? 4 : 2, iq, ik, iv, ia, N, P, Dh, g.sam_heads);
                c.gemm(ia, NP, Dh, Dh, lo + "i2t.out", Cc, true, ACT_NONE, nullptr, 0, nullptr, delta, Cc);
            }
            // (after the last layer nothing adds to the float keys any more: only the T copies are written)
            float* o32 = l + 1 < g.sam_depth && !chain_next ? k32 : nullptr;
            if (chained) {  // (the previous layer left its update in k32's storage and its statistics in chain_stats)
                if (!c.rc && !c.dry)
                    c.rc = launch_layernorm_chain(c.dt, ks32, P, k32, chain_stats, c.Wf(chain_norm + ".g"), c.Wf(chain_norm + ".b"), delta,
                                                  c.Wf(lo + "norm4.g"), c.Wf(lo + "norm4.b"), 1e-5f, kT, l + 1 < g.sam_depth ? kc32 : nullptr,
                                                  (int)NP, Cc, pos, P, kP, c.st);
            } else if (!c.rc && !c.dry) {
                c.rc = launch_layernorm_res(c.dt, cur32, shared ? P : 0, delta, c.Wf(lo + "norm4.g"), c.Wf(lo + "norm4.b"), 1e-5f, kT, o32,
                                            (int)NP, Cc, pos, P, kP, half_shared && l == 0 ? kh32 : nullptr, P, P / 2, c.st, nullptr, nullptr, 0,
                                            nullptr, chain_next ? chain_stats : nullptr);
            }
            if (chained) cur32_after_chain = kc32;
            chained = chain_next;
            chain_norm = lo + "norm4";
            if (shared) {  // from here on every track owns its keys
                Nk = N;
                curT = kT;
                curP = kP;
            }
            cur32 = cur32_after_chain ? cur32_after_chain : k32;
            cur32_after_chain = nullptr;
        }
        c.ws.off = mark;
    }
    // --- final tokens -> image attention (transformer.py:103-109) ---
    size_t mark = c.ws.off;
    {
        void* fq = c.proj(qP, 6ll * N, Cc, "final.q", Dh);
        void* fa = c.T(RgT, Dh);
        if (fold_t2i_ok && Nk == N) {
            t2i_folded(fq, "final", curP, curT, fa, false);
        } else {
            void* fv = c.proj(curT, (long long)Nk * P, Cc, "final.v", Dh);
            void* fk = c.proj(curP, (long long)Nk * P, Cc, "final.k", Dh);
            c.attn(1, fq, fk, fv, fa, N, P, Dh, g.sam_heads);
        }
        c.gemm(fa, 6ll * N, Dh, Dh, "final.out", Cc, true, ACT_NONE, q32, 0, x32, nullptr, Cc);
    }
    c.ws.off = mark;
    void* hsT = c.T(6ll * N, Cc);
    c.ln(x32, "norm_final", 1e-5f, hsT, x32, 6ll * N, Cc, nullptr, 0, nullptr, ACT_NONE);

    // --- hyper-network MLPs on the 3 mask tokens (mask_decoder.py:130-133,160-180) ---
    const int d1 = Cc / g.out_dim_factor, d1p = (d1 + 31) / 32 * 32, cpt = d1p / 32;
    float* hyper = c.f32(3ll * N, d1p);
    if (!c.rc && !c.dry) {
        if (hipMemsetAsync(hyper, 0, (size_t)3 * N * d1p * 4, c.st) != hipSuccess) {
            l4p_set_error("l4p_track_window_forward: hipMemsetAsync failed");
            c.rc = L4P_E_HIP;
        }
    }
    // (the three tokens' MLPs are independent: each stage of the three runs as one grouped launch; the prompt feature for the
    //  next window (sparse_heads.py:650-658: io token 5) rides with the first stage)
    {
        void *h1[3], *h2[3];
        GemmParams st0[4], st1[3], st2[3];
        for (int i = 0; i < 3; ++i) {
            const std::string hk = "hyper" + std::to_string(i);
            h1[i] = c.T(N, Cc);
            h2[i] = c.T(N, Cc);
            st0[i] = c.desc((const char*)hsT + (size_t)i * Cc * c.es, N, Cc, 6ll * Cc, hk + ".0", Cc, true, ACT_RELU, nullptr, 0, nullptr, h1[i], Cc);
            st1[i] = c.desc(h1[i], N, Cc, Cc, hk + ".1", Cc, true, ACT_RELU, nullptr, 0, nullptr, h2[i], Cc);
            st2[i] = c.desc(h2[i], N, Cc, Cc, hk + ".2", d1, true, ACT_NONE, nullptr, 0, c.dry ? nullptr : hyper + (size_t)i * d1p, nullptr, 3ll * d1p);
        }
        st0[3] = c.desc((const char*)hsT + (size_t)5 * Cc * c.es, N, Cc, 6ll * Cc, "prompt_lin", Cc, true, ACT_NONE, nullptr, 0, new_pfeat, nullptr, Cc);
        c.group(st0, 4);
        c.group(st1, 3);
        c.group(st2, 3);
    }

    // --- memory tokens for the next window (sparse_heads.py:406-448,660-665): project the 2nd temporal half of the
    //     processed video tokens into the 1st half of the history, pad the rest with the learned mask token ---
    if (need_history == 3) {
        // the plain single-window forward returns the projection of EVERY processed token (sparse_heads.py:560-569, :658-665:
        // <task>_enc_features_with_track_history_bnpc): hist [N][P][C] receives it, no mask-token rows
        c.gemm(curT, NP, Cc, Cc, "history_proj", Cc, true, ACT_NONE, nullptr, 0, hist, nullptr, Cc);
    } else if (need_history) {
        const int half = P / 2;
        const int a_map[3] = {half, P, half}, c_map[3] = {half, P, 0};
        c.gemm(curT, (long long)N * half, Cc, Cc, "history_proj", Cc, true, ACT_NONE, nullptr, 0, hist, nullptr, Cc, a_map, c_map);
        // (need_history == 2: the caller guarantees that rows [P/2, P) of every track still hold the mask token - nothing but
        //  this fill ever writes them - so the 0.13 ms pass per 128-query window is skipped)
        if (!c.rc && !c.dry && need_history != 2)
            c.rc = launch_fill_rows(hist, c.Wf("history_mask_token"), (long long)N * half, Cc, half, P, half, c.st);
    }

    // --- output up-scaling (mask_decoder.py:58-66,136-137) on channels-last tokens ---
    const int nt = g.nt, nh = g.nh, nw = g.nw;
    const int d0 = (2 * Cc / g.out_dim_factor) < Cc ? 2 * Cc / g.out_dim_factor : Cc;
    const long long M1 = NP * 8;
    void* u0T = c.T(M1, d0);
    if (!c.rc && !c.dry) {
        GemmParams p;
        memset(&p, 0, sizeof(p));
        p.A = curT;
        p.lda = Cc;
        p.W = c.W("up0.w");
        p.ldw = Cc;
        p.M = (int)NP;
        p.N = 8 * d0;
        p.K = Cc;
        p.Ti = nt;
        p.Hi = nh;
        p.Wi = nw;
        p.bias = c.Wf("up0.b");
        p.out_T = u0T;
        p.epi = EPI_CONVT;
        p.kt = p.kh = p.kw = 2;
        p.Cout = d0;
        if (!c.rc) c.rc = launch_gemm(c.dt, 0, p, c.st);
        if (!c.rc) c.rc = launch_layernorm_T(c.dt, u0T, c.Wf("up_ln.g"), c.Wf("up_ln.b"), 1e-6f, u0T, (int)M1, d0, ACT_GELU, c.st);
    }
    // up1 (ConvTranspose (1,2,2) + GELU) fused with the hyper-network mask product (mask_decoder.py:136-139)
    const int Tl = nt * 2, hl = nh * 4, wl = nw * 4;
    if (Tl != T) {
        l4p_set_error("l4p_track_window_forward: decoded masks have %d frames, the window %d", Tl, T);
        return L4P_E_INVALID;
    }
    float* partial = (float*)c.alloc((size_t)4 * cpt * 3 * M1 * 4);
    float* masks = (float*)c.alloc((size_t)N * 3 * Tl * hl * wl * 4);
    if (!c.rc && !c.dry) {
        GemmParams p;
        memset(&p, 0, sizeof(p));
        p.A = u0T;
        p.lda = d0;
        p.W = c.W("up1.w");
        p.ldw = d0;
        p.M = (int)M1;
        p.N = 4 * d1p;
        p.K = d0;
        p.bias = c.Wf("up1.b");
        p.act = ACT_GELU;
        p.out_f32 = partial;
        p.epi = EPI_MASKDOT;
        p.Cout = d1p;
        p.hyper = hyper;
        p.hyper_rows = (int)(M1 / N);
        if (!c.rc) c.rc = launch_gemm(c.dt, 0, p, c.st);
        if (!c.rc) c.rc = launch_mask_gather(partial, masks, N, Tl, nh * 2, nw * 2, cpt, c.st);
        if (!c.rc) c.rc = launch_track_readout(masks, traj, vis, depth, N, T, hl, wl, H, W, c.st);
    }
    return c.rc;
}

bool cfg_ok(const l4p_track_cfg* g, int N) {
    if (!g || N < 1 || g->dim % 64 || g->tokens != g->nt * g->nh * g->nw || g->sam_depth < 1 || g->out_dim_factor < 1 ||
        (long long)N * g->tokens * 8 > 0x7FFFFFFFll) {
        l4p_set_error("l4p_track_window: bad configuration / query count (N=%d)", N);
        return false;
    }
    return true;
}

}  // namespace

extern "C" {

size_t l4p_track_window_workspace_bytes(const l4p_engine* e, const l4p_track_cfg* cfg, int N, int hist_uniform) {
    if (!e || !cfg_ok(cfg, N)) return 0;
    TW c;
    c.e = e;
    c.st = nullptr;
    c.dt = e->dtype;
    c.es = esize_of(e->dtype);
    c.ws = Stack{nullptr, 0, 0, 0, true};
    c.dry = true;
    run(c, *cfg, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, N, 1, hist_uniform, nullptr, nullptr, nullptr, nullptr);
    return c.ws.peak + 256;
}

int l4p_track_window_forward(l4p_engine* e, l4p_stream stream, const l4p_track_cfg* cfg, const float* enc_last, float* hist,
                             const float* q_off, const float* labels, const float* pfeat, const float* plabel, int N,
                             int need_history, int hist_uniform, void* workspace, size_t ws_bytes, float* traj, float* vis,
                             float* depth, float* new_pfeat) {
    if (!e || !cfg_ok(cfg, N)) return L4P_E_INVALID;
    if (!enc_last || !hist || !q_off || !labels || !pfeat || !plabel || !workspace || !traj || !vis || !depth || !new_pfeat) {
        l4p_set_error("l4p_track_window_forward: null argument");
        return L4P_E_INVALID;
    }
    TW c;
    c.e = e;
    c.st = (hipStream_t)stream;
    c.dt = e->dtype;
    c.es = esize_of(e->dtype);
    const size_t mis = (size_t)((uintptr_t)workspace & 255);
    char* base = (char*)workspace + (mis ? 256 - mis : 0);
    c.ws = Stack{base, 0, ws_bytes - (mis ? 256 - mis : 0), 0, false};
    return run(c, *cfg, enc_last, hist, q_off, labels, pfeat, plabel, N, need_history, hist_uniform, traj, vis, depth, new_pfeat);
}

}  // extern "C"
