// C ABI entry points of the tracker kernels (track.hip) and the extended LayerNorm.
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
int launch_mask_product(int dtype, const void* up, const float* hyper, float* masks, int N, long long vox, int Cc,
                        hipStream_t stream);
int launch_track_readout(const float* masks, float* traj, float* vis, float* depth, int N, int T, int h, int w, int H,
                         int W, hipStream_t stream);
int launch_track_prepare(const float* cur_q, const float* orig_q, int start, int ws, float* q_off, float* labels,
                         unsigned char* valid_t, unsigned char* valid_n, int N, hipStream_t stream);
int launch_track_commit(const float* w_traj, const float* w_vis, const float* w_depth, const unsigned char* valid_t,
                        const unsigned char* valid_n, float* traj, float* vis, float* depth, int T, int start, int ws,
                        int next_start, int last_window, float* cur_q, float* plabel, const float* new_pfeat, float* pfeat,
                        int* best_out, int N, int C, hipStream_t stream);

extern "C" {

int l4p_layernorm_res(l4p_stream s, int dtype, const float* x, int x_mod, const void* delta_T, const float* gamma, const float* beta,
                      float eps, void* out_T, float* out_f32, int M, int C, const float* add, int add_mod, void* out_T2,
                      const float* x_shared, int x_period, int x_split, float* out_stats) {
    return launch_layernorm_res(dtype, x, x_mod, delta_T, gamma, beta, eps, out_T, out_f32, M, C, add, add_mod, out_T2, x_shared, x_period,
                                x_split, (hipStream_t)s, nullptr, nullptr, 0, nullptr, out_stats);
}
int l4p_layernorm_chain(l4p_stream s, int dtype, const float* x_shared_rows, int x_mod, const void* delta_prev_T, const float* stats_prev,
                        const float* gamma_prev, const float* beta_prev, const void* delta_T, const float* gamma, const float* beta, float eps,
                        void* out_T, float* out_f32, int M, int C, const float* add, int add_mod, void* out_T2) {
    return launch_layernorm_chain(dtype, x_shared_rows, x_mod, delta_prev_T, stats_prev, gamma_prev, beta_prev, delta_T, gamma, beta, eps, out_T,
                                  out_f32, M, C, add, add_mod, out_T2, (hipStream_t)s);
}
int l4p_layernorm_ex(l4p_stream s, int dtype, const float* x, const float* gamma, const float* beta, float eps,
                     void* out_T, float* out_f32, int M, int C, const float* add, int add_mod, void* out_T2, int act) {
    return launch_layernorm_ex(dtype, x, gamma, beta, eps, out_T, out_f32, M, C, add, add_mod, out_T2, act, (hipStream_t)s);
}
int l4p_layernorm_t(l4p_stream s, int dtype, const void* x_T, const float* gamma, const float* beta, float eps, void* out_T,
                    int M, int C, int act) {
    return launch_layernorm_T(dtype, x_T, gamma, beta, eps, out_T, M, C, act, (hipStream_t)s);
}
int l4p_track_tokens(l4p_stream s, const float* queries, const float* labels, const float* pfeat, const float* plabel,
                     const float* gauss, const float* mask_tokens, const float* point_emb0, const float* point_emb1,
                     const float* not_a_point, const float* feat_emb0, const float* feat_emb1, float* tokens, int N, int C,
                     int T, int H, int W) {
    return launch_track_tokens(queries, labels, pfeat, plabel, gauss, mask_tokens, point_emb0, point_emb1, not_a_point,
                               feat_emb0, feat_emb1, tokens, N, C, T, H, W, (hipStream_t)s);
}
int l4p_track_keys_init(l4p_stream s, int dtype, const float* enc, const float* hist, const float* pos, float* k32,
                        void* kT, void* kP, int N, int P, int C, int shared_from, float* k32_shared) {
    return launch_track_keys_init(dtype, enc, hist, pos, k32, kT, kP, N, P, C, shared_from, k32_shared, (hipStream_t)s);
}
int l4p_fill_rows(l4p_stream s, float* out, const float* v, long long rows, int C, long long group_rows,
                  long long group_stride, long long group_off) {
    return launch_fill_rows(out, v, rows, C, group_rows, group_stride, group_off, (hipStream_t)s);
}
int l4p_broadcast_block(l4p_stream s, void* base, long long off, long long bytes, long long stride, int n) {
    return launch_broadcast_block(base, off, bytes, stride, n, (hipStream_t)s);
}
int l4p_small_attn(l4p_stream s, int dtype, int kind, const void* q, const void* k, const void* v, void* out, int N, int P,
                   int D, int heads) {
    return launch_small_attn(dtype, kind, q, k, v, out, N, P, D, heads, (hipStream_t)s);
}
int l4p_mask_product(l4p_stream s, int dtype, const void* up, const float* hyper, float* masks, int N, long long vox,
                     int C) {
    return launch_mask_product(dtype, up, hyper, masks, N, vox, C, (hipStream_t)s);
}
int l4p_i2t_probs(l4p_stream s, int dtype, const float* scores, long long ld_scores, int pairs, const float* cbias, int rows_per_group,
                  void* probs_T, int ld_probs, long long M, int heads, int tokens) {
    return launch_i2t_probs(dtype, scores, ld_scores, pairs, cbias, rows_per_group, probs_T, ld_probs, M, heads, tokens, (hipStream_t)s);
}
int l4p_t2i_attn_scores(l4p_stream s, int dtype, const float* scores, long long ld_scores, const void* v_T, void* out_T, int N, int P, int D,
                        int heads) {
    return launch_t2i_attn_scores(dtype, scores, ld_scores, v_T, out_T, N, P, D, heads, (hipStream_t)s);
}
int l4p_split_hilo(l4p_stream s, int dtype, const float* in, void* out_T, int G, int R, long long C) {
    return launch_split_hilo(dtype, in, out_T, G, R, C, (hipStream_t)s);
}
int l4p_transpose_pad(l4p_stream s, int dtype, const void* in_T, void* out_T, int G, int R, int C, int Rp) {
    return launch_transpose_pad(dtype, in_T, out_T, G, R, C, Rp, (hipStream_t)s);
}
int l4p_i2t_delta(l4p_stream s, int dtype, const void* probs_T, const void* vt_T, const float* bias, void* delta_T, int N, int P, int C, int K) {
    return launch_i2t_delta(dtype, probs_T, vt_T, bias, delta_T, N, P, C, K, (hipStream_t)s);
}
int l4p_t2i_probs(l4p_stream s, int dtype, const float* scores, long long ld_scores, void* probs_T, float* stats, int N, int P, int HT) {
    return launch_t2i_probs(dtype, scores, ld_scores, probs_T, stats, N, P, HT, (hipStream_t)s);
}
int l4p_t2i_context(l4p_stream s, int dtype, const void* probs_T, const float* stats, const void* keys_T, void* ctx_T, int N, int P, int C,
                    int heads, int tokens, long long Rg, int shared_from) {
    return launch_t2i_context(dtype, probs_T, stats, keys_T, ctx_T, N, P, C, heads, tokens, Rg, shared_from, (hipStream_t)s);
}
int l4p_mask_gather(l4p_stream s, const float* partial, float* masks, int N, int T, int h, int w, int chunks_per_tap) {
    return launch_mask_gather(partial, masks, N, T, h, w, chunks_per_tap, (hipStream_t)s);
}
int l4p_track_readout(l4p_stream s, const float* masks, float* traj, float* vis, float* depth, int N, int T, int h, int w,
                      int H, int W) {
    return launch_track_readout(masks, traj, vis, depth, N, T, h, w, H, W, (hipStream_t)s);
}
int l4p_track_prepare(l4p_stream s, const float* cur_q, const float* orig_q, int start, int ws, float* q_off,
                      float* labels, unsigned char* valid_t, unsigned char* valid_n, int N) {
    return launch_track_prepare(cur_q, orig_q, start, ws, q_off, labels, valid_t, valid_n, N, (hipStream_t)s);
}
int l4p_track_commit(l4p_stream s, const float* w_traj, const float* w_vis, const float* w_depth,
                     const unsigned char* valid_t, const unsigned char* valid_n, float* traj, float* vis, float* depth, int T,
                     int start, int ws, int next_start, int last_window, float* cur_q, float* plabel, const float* new_pfeat,
                     float* pfeat, int* best_out, int N, int C) {
    return launch_track_commit(w_traj, w_vis, w_depth, valid_t, valid_n, traj, vis, depth, T, start, ws, next_start,
                               last_window, cur_q, plabel, new_pfeat, pfeat, best_out, N, C, (hipStream_t)s);
}

}  // extern "C"
