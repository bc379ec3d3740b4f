/*
 * l4p_hip.h — C ABI of libl4p_hip.so: the MI355X (gfx950) engine behind the L4P inference hot path.
 *
 * This is the drop-in boundary.  The reference (NVlabs/L4P) is pure Python/PyTorch with no native
 * layer, so there is no FFI to mirror one-to-one: each entry point below replaces the ATen/cuDNN
 * work done by a specific reference function (cited per entry as file:line relative to the
 * reference tree).  The Python host (package l4p_amd, same class / argument names as the
 * reference's l4p.l4p.L4PLitModule, l4p.models.*) binds these symbols with ctypes; INTEGRATION.md
 * shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer named *dev* / every tensor argument is a DEVICE
 *    pointer owned by the caller (the Python host allocates with torch and passes data_ptr()).
 *  - every function returns 0 on success, a negative L4P_E_* code otherwise; the message is
 *    available through l4p_last_error() (thread-local).  No C++ exception crosses this boundary.
 *  - all work is enqueued asynchronously on the caller-supplied HIP stream (l4p_stream ==
 *    hipStream_t, NULL = default stream).  No hidden synchronisation, no allocation.
 *  - dtype selects the arithmetic/storage type T of activations and weights: L4P_BF16 (bf16 storage,
 *    bf16 MFMA, f32 accumulate; residual stream, LayerNorm and softmax statistics always f32) or
 *    L4P_F32 (f32 storage, exact-f32 MFMA) — the parity mode; L4P_F16: as L4P_BF16 with IEEE half in place of bf16
 *    (3 more mantissa bits, 5-bit exponent: the arithmetic class of the reference's fp16 autocast).
 */
#ifndef L4P_HIP_H
#define L4P_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define L4P_BF16 0
#define L4P_F32 1
#define L4P_F16 2 /* IEEE half storage + f16 MFMA, f32 accumulate: what the reference's shipped "16-mixed" (fp16 autocast) computes in */

#define L4P_OK 0
#define L4P_E_INVALID (-1) /* bad argument / unsupported shape */
#define L4P_E_HIP (-2)     /* a HIP runtime call failed */
#define L4P_E_MISSING (-3) /* a required weight was never bound */

typedef void* l4p_stream; /* hipStream_t */
typedef struct l4p_engine l4p_engine;

const char* l4p_last_error(void);
int l4p_abi_version(void); /* 9: l4p_similarity_prefix, knobs attn64 / probe_kernels; 8: l4p_gemm_desc.ups_hi / ups_wi; 7: L4P_F16; 6: l4p_set_knob / l4p_get_knob; 5: l4p_layernorm_res(out_stats), l4p_layernorm_chain, l4p_stream_create_cu_mask; 4: l4p_gemm_desc.o_gs, l4p_i2t_delta,
                              * l4p_t2i_probs, l4p_t2i_context; 3: l4p_gemm_desc.w_gr / w_gs / b_gs, l4p_i2t_probs,
                              * l4p_t2i_attn_scores, l4p_split_hilo, l4p_transpose_pad */

/* A HIP stream restricted to CUs [first_cu, first_cu + n_cus) (hipExtStreamCreateWithCUMask) / its release.  Plumbing for the sharded
 * long-video path (l4p_amd/parallel.py): the tracker recursion's small dependent kernels on a slice of the chip of their own. */
int l4p_stream_create_cu_mask(int first_cu, int n_cus, l4p_stream* out);
int l4p_stream_destroy(l4p_stream stream);

/* Dispatch knobs of the launchers (A/B and test aids; the defaults are the shipped configuration).  A knob starts from its
 * environment variable (read ONCE, at first use) and can be changed at run time through l4p_set_knob - no getenv on a launch path.
 *   "conv_halo"  (L4P_CONV_HALO, default 1): 0 = implicit-GEMM forms for every 3x3x3 conv, 1 = the LDS-halo kernel where it fits
 *   "gemm_4w"    (L4P_GEMM_4W,   default 0): the two-workgroups-per-CU GEMM: 0 never, 1 on the shapes it won, 2 wherever it fits
 *   "maskdot_mfma" (L4P_MASKDOT_MFMA, default 1): L4P_EPI_MASKDOT of the 16-bit engines contracts the activated row with the
 *                hyper-network vectors on the matrix pipe (the row rounded to T first, as the reference's autocast holds it);
 *                0 = the all-VALU form (float row, float dot products)
 *   "conv_ups"   (L4P_CONV_UPS, default 0): 1 = l4p_dpt_forward forms the up-sampling in front of the head conv inside that conv's
 *                loader (l4p_gemm_desc.ups_hi) where the shape allows - the same result bit for bit, 822 MB per head never written,
 *                but MEASURED SLOWER (round 5: head conv 2190 -> 3404 us against 290 us of up-sampling saved: with one pass of four
 *                taps in flight - all the registers the kernel has left - the taps' latency is not hidden); 0 = up-sample, then convolve
 *   "ln_tracks"  (L4P_LN_TRACKS, default 1): the tracker's key LayerNorms (l4p_layernorm_res / l4p_layernorm_chain with a positional
 *                addend of period P = add_mod over whole tracks) run laid out by TOKEN: a wave owns one token and walks its tracks,
 *                the shared float rows stay in registers and the parameter vectors in LDS (bit-identical to the row kernels;
 *                473 -> 223 us and 485 -> 309 us for 64 tracks x 2048 tokens); 0 = one wave per row
 *   "ln_rows16"  (L4P_LN_ROWS16, default 1): l4p_layernorm_t on short rows (C <= 512, >= 4096 of them: the tracker's LayerNorm3d + GELU)
 *                gives a row to one DPP row of 16 lanes - four rows per wave at a time, statistics by rotations inside the DPP row,
 *                gamma / beta in registers over 16 rows per wave; 0 = one wave per row.  Equal to float rounding, not bit for bit.
 *   "attn64"     (L4P_ATTN64, default 1): l4p_attention of the 16-bit engines on chip-filling launches (>= 256 tiles of 256 query rows,
 *                S % 256 == 0): one wave per SIMD with 64 query rows (csrc/attention64.hip: every K / V^T fragment read feeds two
 *                MFMAs); 0 = the 8-wave form with 32 rows per wave.  Equal to the rounding of P, not bit for bit (the rare rescale of
 *                the deferred maximum is decided per wave).
 *   "gemm_skinny" (L4P_GEMM_SKINNY, default 1): l4p_gemm / l4p_gemm_group of the 16-bit engines on dense problems with M <= 128 rows and
 *                K % 64 == 0 (the tracker's token-side projections of a rank's query shard) run one wave per 16 x 32 output block with
 *                the operands streamed from global memory into MFMA fragment registers (csrc/gemm_skinny.hpp); bit-identical to the
 *                LDS-staged kernels (0)
 *   "readout_wide" (L4P_READOUT_WIDE, default 1): l4p_track_readout with at most 512 (track, frame) pairs runs 1024 threads per pair
 *                (the samples of 32 rows formed by all threads, then summed per column in the order of the 256-thread kernel:
 *                bit-identical; 58 -> ~20 us per launch on a rank's 8-track shard); 0 = the 256-thread kernel
 *   "probe_kernels" (read-only): 1 when the library was built with PROBES=1 and contains the measured-and-not-adopted kernels the knobs
 *                "gemm_4w" and "conv_ups" select; in the shipped build (0) those two knobs stay 0 and setting them is an error.
 * l4p_set_knob returns L4P_E_INVALID for an unknown name; l4p_get_knob returns the current value (or -1). */
int l4p_set_knob(const char* name, int value);
int l4p_get_knob(const char* name);

/* Optional per-kernel-class timing: when enabled every kernel launch is bracketed by a HIP event pair
 * recorded on the launch stream (bench.py's live roofline numbers).  Classes: gemm, conv3d, attention,
 * layernorm, elementwise, track, preprocess.  l4p_prof_read sums the pair durations of one class since the last
 * reset; the caller synchronises the stream first. */
int l4p_prof_enable(int on);
int l4p_prof_reset(void);
int l4p_prof_num_classes(void);
const char* l4p_prof_class_name(int cls);
int l4p_prof_read(int cls, double* total_ms, long long* count);
/* Per-(class, tag) breakdown (tag = GEMM shape, LayerNorm shape, kernel name ...) of the same event pairs as text
 * lines "class\ttag\tcount\ttotal_ms\n", largest first.  Returns the bytes needed incl. the terminator; writes at
 * most cap bytes to buf (buf may be NULL to query the size).  Tuning aid behind tools/prof_detail.py. */
long long l4p_prof_detail(char* buf, long long cap);

/* ------------------------------------------------------------------------------------------------
 * Kernel-level entry points (also the unit-parity surface of tests/).
 * ---------------------------------------------------------------------------------------------- */

#define L4P_EPI_DENSE 0
#define L4P_EPI_QKV 1
#define L4P_EPI_CONVT 2
#define L4P_EPI_MASKDOT 3
#define L4P_ACT_NONE 0
#define L4P_ACT_GELU 1
#define L4P_ACT_RELU 2

/* out[m][n] = epilogue( sum_k A[m][k] * W[n][k] ), both operands k-contiguous.
 * Replaces F.linear / nn.Linear (modeling_finetune.py:57-68,176,188; sam/transformer.py:215-218,
 * 225-228; mask_decoder.py:160-180), nn.Conv3d 1x1x1 (dpt_block.py:447-507,218-227),
 * nn.ConvTranspose3d with kernel == stride (dpt_block.py:255-265; mask_decoder.py:58-66) and,
 * through l4p_conv3d_k3, nn.Conv3d 3x3x3 (dpt_block.py:44-79,110-129,266-276,406-414). */
typedef struct l4p_gemm_desc {
    const void* A; /* [M][lda] T (dense) or channels-last conv input [B][Ti][Hi][Wi][Cin] T */
    long long lda;
    const void* W; /* [ceil(N/128)*128][ldw] T, rows >= N are zero */
    long long ldw;
    int M, N, K;
    /* conv3d k=3 pad=1 (l4p_conv3d_k3 only): M = B*To*Ho*Wo, K = 27*Cin, k = tap*Cin + c */
    int Ti, Hi, Wi, Cin, To, Ho, Wo, st, sh, sw, relu_in;
    /* epilogue: v = acc + bias[n]; v = act(v); v += res1[m][n] (+ res2[m][n]); store */
    const float* bias;
    int act;
    const void* res1;
    const void* res2;
    int res_f32;   /* residual element type: 1 float, 0 T */
    long long ldr; /* residual row stride (elements) */
    int res_mod;   /* > 0: residual row is m % res_mod (broadcast table, e.g. the pos-embed) */
    float* out_f32; /* optional float output  [M][ldc] */
    void* out_T;    /* optional T output      [M][ldc] */
    long long ldc;
    int epi;
    /* L4P_EPI_QKV (Dp = 96): n < H*Dp -> q: out_T[m][n] (ldc = H*Dp); H*Dp <= n < 2*H*Dp -> k: kt, tiled
     * in 8-element groups [b][h][S/KVB][Dp/16][KVB][half ^ ((key>>3)&1)], KVB = 64 (bf16) / 32 (f32);
     * n >= 2*H*Dp -> v transposed: vt[b][h][d][s] */
    void* vt;
    int S, H, Dp;
    /* L4P_EPI_CONVT: n = tap*Cout + co, tap = (dt*kh + dh)*kw + dw; A rows are the (Ti,Hi,Wi) grid;
     * output is channels-last [B][Ti*kt][Hi*kh][Wi*kw][Cout] */
    int kt, kh, kw, Cout;
    /* optional row re-mapping (0 = off): logical row m lives at physical row
     * (m / gr) * gs + go + (m % gr) — used to read / write one temporal half of per-query token blocks
     * (tracker memory tokens, sparse_heads.py:406-448).  a_*: rows of A, c_*: rows of out / residual. */
    int a_gr, a_gs, a_go;
    int c_gr, c_gs, c_go;
    /* optional second T output = relu(v), same addressing as out_T: the pre-activated operand of the next
     * ResidualConvUnit conv (dpt_block.py:139-146), so that conv can stream its input without a fused ReLU */
    void* out_relu_T;
    /* L4P_EPI_QKV: K destination in the attention kernel's tile order (see l4p_attention) */
    void* k_tiled;
    /* split-K (L4P_EPI_DENSE only; for problems with too few output tiles to fill 256 CUs, e.g. the low-resolution
     * DPT convs with K = 27*1024): splitk > 1 slices contract disjoint k ranges into partial[splitk][M][N] (float,
     * caller-provided), a second kernel sums them and applies the epilogue. */
    int splitk;
    float* partial;
    /* L4P_EPI_MASKDOT (the tracker's last up-scaling ConvTranspose fused with the hyper-network mask product,
     * mask_decoder.py:136-139): the activated outputs are not stored; for every 32-column chunk c of row m
     *   out_f32[(c * 3 + i) * M + m] = sum_{n in chunk} act(acc + bias)[m][n] * hyper[((m / hyper_rows) * 3 + i) * Cout + n % Cout]
     * (i = 0..2).  Columns are tap-major with Cout (a multiple of 32, zero padded) columns per tap;
     * l4p_mask_gather sums a tap's chunks and scatters the taps to the up-scaled grid. */
    const float* hyper;
    int hyper_rows;
    /* L4P_EPI_QKV: the q columns (n < H*Dp) are multiplied by q_scale (after the bias, before rounding to T) — the
     * reference's `q = q * self.scale` (modeling_finetune.py:180) folded into the projection; the engine passes
     * head_dim^-0.5 * log2(e) so that the scores leave the attention MFMA in the exp2 domain (see l4p_attention).
     * 0 is treated as 1. */
    float q_scale;
    /* tuning aid, normally 0.  bit 0: use the generic run-time-dispatched epilogue even where a lean specialisation exists
     * (set by the launcher when L4P_EPI_GENERIC=1: in-run A/B of the two forms).  bit 1: a split-K launch leaves its float
     * partials [splitk][M][N] to the caller and runs no finish pass (bias / residual / outputs of the descriptor are ignored):
     * the encoder sums them in the LayerNorm that follows the batch-1 MLP-out projection.  bit 2: L4P_EPI_MASKDOT in its all-VALU
     * form (set by the launcher when the "maskdot_mfma" knob is 0). */
    int tuning;
    /* Row-grouped weights (0 = off; dense GEMM without split-K, w_gr a multiple of 128): rows [g * w_gr, (g + 1) * w_gr) of A
     * multiply their OWN weight matrix W + g * w_gs (elements, same ldw) and add their own bias row bias + g * b_gs (b_gs = 0: one
     * bias for all groups).  The tracker's image -> token attention with the projections folded into the token side
     * (sparse_heads.py / sam/transformer.py:180-185, see l4p_amd/models/task_heads/sparse_heads.py): every track's 2048 key rows
     * meet that track's 48 x 1408 folded key matrix, then its 1408 x 48 folded value matrix.  A group's matrix needs exactly N rows
     * (tile rows past N re-read row N - 1); the plain (w_gr = 0) path reads whole tiles: W padded to a multiple of 128 rows. */
    int w_gr;
    long long w_gs;
    int b_gs;
    /* ... and write to out_T / out_f32 + g * o_gs (elements; with the c_* row map rows of all groups can land on the same physical rows
     * in their own column blocks: the per-head value projection of the folded token -> image attention, l4p_t2i_context) */
    long long o_gs;
    /* l4p_conv3d_k3 with the bilinear up-sampling of its input fused into the loader (ups_hi > 0; 16-bit engines, stride 1, the
     * LDS-halo kernel's shapes: N == 128, Cin % 32 == 0, To % 2 == Ho % 16 == Wo % 16 == 0): A is the LOW-resolution volume
     * [B][Ti][ups_hi][ups_wi][Cin]; the conv reads F.interpolate(A, (Ti, Hi, Wi), trilinear, align_corners=True) - the time axis is
     * not resized - rounded to T exactly as l4p_upsample_trilinear would have stored it, without that tensor ever existing
     * (dpt_head.py:79-84: interpolate -> head conv).  l4p_conv3d_k3 returns L4P_E_INVALID for shapes the fused loader does not take. */
    int ups_hi, ups_wi;
    /* Block-structured weights (dense GEMM, kw_cols > 0): output columns [g * kw_cols, (g + 1) * kw_cols) only meet the kw_len elements
     * [g * kw_len, (g + 1) * kw_len) of the contraction - W is zero everywhere else (the tracker's folded projections, packing.py
     * fold_i2t / fold_t2i: head h's 1408 columns meet head h's 88 inputs) - so a tile walks only the k-tiles that cover its group's
     * window: the same sums bit for bit (the skipped tiles add zeros), 1/4 of the weight bytes.  kw_cols a multiple of 128. */
    int kw_cols, kw_len;
} l4p_gemm_desc;

int l4p_gemm(l4p_stream stream, int dtype, const l4p_gemm_desc* d);
/* n <= L4P_GEMM_GROUP_MAX INDEPENDENT dense GEMMs (no output of one is an input of another).  Equal to n l4p_gemm calls,
 * bit for bit; small bf16 problems run as ONE kernel launch (the tracker's token-side projections, sam/transformer.py:159-185:
 * q / k / v of the self-attention, k / v of the image -> token attention, the hyper-network MLP stages of the three mask
 * tokens, mask_decoder.py:130-133), anything else as separate launches. */
#define L4P_GEMM_GROUP_MAX 4
int l4p_gemm_group(l4p_stream stream, int dtype, const l4p_gemm_desc* d, int n);
int l4p_conv3d_k3(l4p_stream stream, int dtype, const l4p_gemm_desc* d);

/* Row LayerNorm of a float [M][C] stream -> T and/or float.  Replaces nn.LayerNorm
 * (modeling_finetune.py:212,235; l4p_videomae.py:115,177; sam/transformer.py:143-153) and
 * LayerNorm3d over channels-last data (mask_decoder.py:145-157). */
int l4p_layernorm(l4p_stream stream, int dtype, const float* x, const float* gamma, const float* beta, float eps,
                  void* out_T, float* out_f32, int M, int C);

/* Fused softmax(q k^T * scale) v for the encoder (modeling_finetune.py:180-186).
 * q: [B*S][H*96] T, kt: tiled K, vt: [B][H][96][S] T (all three written by L4P_EPI_QKV), out: [B*S][H*Dh] T.
 * Dh in {88, 64} (head dim < 96: the padding carries the softmax denominator (V^T) and the running maximum (Q, K)).
 * scale > 0: q is the plain projection; the bf16 kernel folds scale * log2(e) into its Q fragments (a second bf16
 * rounding of q).  scale == 0 (L4P_ATTN_PRESCALED): q was already multiplied by head_dim^-0.5 * log2(e) when it was
 * produced (l4p_gemm_desc.q_scale) and the weights are exp2(q k^T) — the form the engine uses: one rounding of q. */
#define L4P_ATTN_PRESCALED 0.0f
int l4p_attention(l4p_stream stream, int dtype, const void* q, const void* kt, const void* vt, void* out, int B, int S,
                  int H, int Dh, float scale);

/* Tubelet gather of PatchEmbed's Conv3d(kernel=stride) (modeling_finetune.py:269-283):
 * rgb [B][Cin][T][H][W] float -> out [B*nT*nH*nW][Kp] T, zero-padded columns >= Cin*pt*ph*pw. */
int l4p_patch_gather(l4p_stream stream, int dtype, const float* rgb, void* out, int B, int Cin, int T, int H, int W,
                     int pt, int ph, int pw, int Kp);

int l4p_cast(l4p_stream stream, int dtype, const float* x, void* y, long long n);

/* Trilinear resize of channels-last data [B][Ti][Hi][Wi][C] -> [B][To][Ho][Wo][C] (C % 8 == 0).
 * Replaces F.interpolate(mode="trilinear", align_corners=True) in the DPT decoders
 * (dpt_block.py:229-234; dpt_head.py:79-83) and align_corners=False in the tracker
 * (sparse_heads.py:645-647). */
int l4p_upsample_trilinear(l4p_stream stream, int dtype, const void* x, void* y, int B, int Ti, int Hi, int Wi, int To,
                           int Ho, int Wo, int C, int align_corners);

/* DPT head2[2]: Conv3d 1x1x1 C(=128) -> Cout (<= 8) + optional exp; channels-last T in, NCDHW float out
 * (dpt_block.py:413; dense_heads.py:73,179,215 with apply_fn 'exp' misc.py:23-24).
 * w: float [Cout][C], bias: float [Cout], y: float [B][Cout][vox_per_b]. */
int l4p_head_out(l4p_stream stream, int dtype, const void* x, const float* w, const float* bias, float* y,
                 long long vox_per_b, int B, int C, int Cout, int post_exp);

/* LstSqAffineAligner (aligner.py:29-66): least-squares scale/shift between two overlapping depth
 * windows, in inverse depth (mode bit 0 set: f = safe_inverse, misc.py:48-62) or directly (bit 0 clear).
 * solve: sol[0..1] = argmin_{s,t} || s f(pred) + t - f(target) ||^2 over n floats.
 * Mode bit 1 (L4P_ALIGN_RATIO_MEAN) selects LinearAligner(method="mean") instead (aligner.py:69-118):
 * sol = (mean(f(target) / (f(pred) + 1e-8)), 0).
 * scratch: L4P_AFFINE_SCRATCH_DOUBLES doubles (per-workgroup partial sums, added in a fixed order: the
 * result is bit-reproducible).
 * apply: y = f(s f(x) + t). */
#define L4P_ALIGN_INVERSE 1
#define L4P_ALIGN_RATIO_MEAN 2
#define L4P_AFFINE_SCRATCH_DOUBLES 4096
int l4p_affine_align_solve(l4p_stream stream, const float* pred, const float* target, long long n, int mode,
                           double* scratch, float* sol);
int l4p_affine_align_apply(l4p_stream stream, const float* x, float* y, long long n, int inverse, const float* sol);

/* rays_to_cameras + pose inversion (geometry_utils.py:331-406, :249-328; dense_heads.py:346-348):
 * rays float [B][6][T][h][w] (Pluecker direction|moment), K float [B][4][4][T] pixel intrinsics of an
 * H x W image  ->  out float [B][16][T] = row-major world_T_cam per frame. */
int l4p_rays_to_pose(l4p_stream stream, const float* rays, const float* K, float* out, int B, int T, int h, int w,
                     int H, int W);

/* Intrinsics from the ray map of frame t0 (compute_optimal_rotation_intrinsics, geometry_utils.py:409-456, as
 * used by rays_to_cameras_and_fixed_per_frame_intrinsics :493-579): homography pixel -> direction by
 * normalised DLT re-estimated on its consensus set (reprojection error < reproj_thr), H^-1 = K R, RQ.
 * Deterministic replacement of cv2.findHomography(RANSAC)+cv2.RQDecomp3x3 ("parity unpinned").
 * rays float [B][6][T][h][w], out_K float [B][4][4][T] (pixel units of the H x W image, same K for all
 * frames), diag optional float [B][2] = (consensus size, iterations). */
int l4p_rays_to_intrinsics(l4p_stream stream, const float* rays, float* out_K, float* diag, int B, int T, int h, int w,
                           int H, int W, int t0, float reproj_thr);

/* The per-frame VARIABLE intrinsics branch (fixed_intrinsics = False: rays_to_cameras_and_variable_per_frame_intrinsics,
 * geometry_utils.py:582-654; reached from dense_heads.py:336-344): the same estimator run on EVERY frame's ray map:
 * out_K float [B][4][4][T] = frame t's own K; out_R float [B][9][T] = the rotation R of H^-1 = K R (row-major), which that
 * branch uses as the camera rotation as it is (no Kabsch step); diag optional float [B*T][2].  l4p_rays_to_pose_rot then
 * gives world_T_cam = [R^T | c] with the camera centre c solved from the rays (intersect_skew_lines_high_dim, :249-282). */
int l4p_rays_to_intrinsics_frames(l4p_stream stream, const float* rays, float* out_K, float* out_R, float* diag, int B, int T,
                                  int h, int w, int H, int W, float reproj_thr);
int l4p_rays_to_pose_rot(l4p_stream stream, const float* rays, const float* R, float* out, int B, int T, int h, int w);

/* ------------------------------------------------------------------------------------------------
 * Joint depth + camera seam alignment (KabaschUmeyama3DAligner, aligner.py:121-265;
 * generate_point_map, geometry_utils.py:13-53).  The reference runs numpy + skimage.measure.ransac on
 * the CPU (randomised, unpinned): these are deterministic GPU restatements of the same estimator,
 * validated against synthetic ground truth ("parity unpinned").
 * ---------------------------------------------------------------------------------------------- */

/* EXACT q-quantile of n finite floats (either sign) with torch.quantile's linear interpolation (aligner.py:187): radix
 * select of the order statistics floor(q (n-1)) and the next one on an order-preserving integer image of the float bit
 * patterns, then ATen's lerp.  ws >= L4P_QUANTILE_WS_UINTS uints. */
#define L4P_QUANTILE_WS_UINTS 2052
int l4p_quantile(l4p_stream stream, const float* x, long long n, float q, unsigned* ws, float* out);
/* EXACT order statistic `rank` (0-based, ascending) of n finite floats; torch.median(x) = rank (n - 1) / 2. */
int l4p_select_rank(l4p_stream stream, const float* x, long long n, long long rank, unsigned* ws, float* out);
/* LinearAligner(method="median").solve (aligner.py:96-107): sol[0] = torch.median over f(target) / (f(pred) + 1e-8)
 * (f = safe_inverse when `inverse`, misc.py:48-62; float arithmetic; the LOWER median as torch.median returns it),
 * sol[1] = 0, so that l4p_affine_align_apply applies it.  ratios: n floats of scratch; ws >= L4P_QUANTILE_WS_UINTS uints. */
int l4p_ratio_median_solve(l4p_stream stream, const float* pred, const float* target, long long n, int inverse,
                           float* ratios, unsigned* ws, float* sol);

/* world-space points of a hashed 1/ratio pixel subset of F frames: depth [F][H*W], K and P (world_T_cam)
 * [F][16] row-major 4x4  ->  out [F*(H*W/ratio)][3] */
int l4p_point_map_samples(l4p_stream stream, const float* depth, const float* K, const float* P, float* out, int F,
                          int H, int W, int ratio, unsigned seed);

/* RANSAC similarity dst ~ s R src + t over n correspondences (min_samples per trial, inlier threshold
 * q98[0]*thr_rel, best model re-estimated on its inliers as skimage does).  ws: float[15*trials].
 * out: float[18] = row-major 4x4 [sR|t;0 0 0 1], s, inlier count. */
int l4p_similarity_ransac(l4p_stream stream, const float* src, const float* dst, int n, const float* q98,
                          float thr_rel, int trials, int min_samples, unsigned seed, float* ws, float* out);

/* aligner.apply (aligner.py:239-265): pose [16][T] <- T pose with the 3x3 block / s ; depth[n] *= s */
int l4p_similarity_apply(l4p_stream stream, const float* sim, float* pose, int T, float* depth, long long n);

/* Prefix composition of per-seam similarities (l4p_similarity_ransac records): rel [n][B][18], seam i = window i + 1 aligned to the
 * RAW window i; out [n + 1][B][18], out[0] = identity, out[w] = out[w - 1] o rel[w - 1] = window w in window 0's frame (applying two
 * records in a row = applying the 4x4 product with the product of the scales: l4p_similarity_apply divides the rotation block by
 * the scale after each product).  The sharded long-video path's seam-local exchange (reference loop
 * dense_heads.py:444-467, which aligns every window to the accumulated buffer one after the other). */
int l4p_similarity_prefix(l4p_stream stream, const float* rel, float* out, int n, int B);

/* ------------------------------------------------------------------------------------------------
 * SAM-style point tracker (sparse_heads.py, sam/{prompt_encoder,transformer,mask_decoder}.py).
 * The projections of the two-way transformer run through l4p_gemm; these are the remaining pieces.
 * ---------------------------------------------------------------------------------------------- */

/* LayerNorm with the tracker's fused extras: y = LN(x) [then GELU if act == L4P_ACT_GELU, as in
 * mask_decoder.py:60-62]; out_f32 = y, out_T = T(y), out_T2 = T(y + add[row % add_mod]) — the
 * "queries + query_pe" / "keys + key_pe" operands of sam/transformer.py:166-185. */
int l4p_layernorm_ex(l4p_stream stream, int dtype, const float* x, const float* gamma, const float* beta, float eps,
                     void* out_T, float* out_f32, int M, int C, const float* add, int add_mod, void* out_T2, int act);

/* keys = LayerNorm(keys + attention output) (sam/transformer.py:183-185) with the sum formed in the LayerNorm: the row
 * normalised is x[row % x_mod] (float; x_mod = 0: x[row]) + delta[row] (engine dtype, the out projection's result), outputs as
 * l4p_layernorm_ex.  The float key stream is read once here instead of read + written by the projection's epilogue and read
 * again; out_f32 may alias x when x_mod = 0.  x_shared (may be NULL): rows p = row % x_period with p >= x_split read
 * x_shared[p - x_split] instead - the part of the key stream that is the same for every track of a later window (encoder
 * feature + the learned mask token: l4p_track_keys_init's k32_shared); x_shared is never written, so out_f32 may still alias x. */
int l4p_layernorm_res(l4p_stream stream, int dtype, const float* x, int x_mod, const void* delta_T, const float* gamma,
                      const float* beta, float eps, void* out_T, float* out_f32, int M, int C, const float* add, int add_mod,
                      void* out_T2, const float* x_shared, int x_period, int x_split, float* out_stats);
/* ... out_stats (may be NULL): float [M][2] = (mean, rstd) of every row, and
 * l4p_layernorm_chain: the NEXT layer's keys = LayerNorm(y_prev + delta) where y_prev - the float result of an l4p_layernorm_res launch
 * over x_shared_rows[row % x_mod] + delta_prev[row] that stored only its engine-dtype outputs and out_stats - is re-derived from those
 * inputs, bit for bit (same affine form, same statistics).  The tracker's second layer in a first window: the float key master
 * [N * P][C] is neither written by layer 0 nor read by layer 1 (1.1 GB of 3.7 GB per 64 tracks).  Outputs as l4p_layernorm_ex. */
int l4p_layernorm_chain(l4p_stream stream, int dtype, const float* x_shared_rows, int x_mod, const void* delta_prev_T, const float* stats_prev,
                        const float* gamma_prev, const float* beta_prev, const void* delta_T, const float* gamma, const float* beta, float eps,
                        void* out_T, float* out_f32, int M, int C, const float* add, int add_mod, void* out_T2);

/* The same LayerNorm (+ optional GELU) for rows that are STORED in the engine dtype: x_T, out_T are T [M][C] and may be the
 * same buffer.  Used for LayerNorm3d + GELU after the first up-scaling ConvTranspose (mask_decoder.py:60-62), whose 1M x 352
 * activation per clip is then never held in float (the reference holds it in fp16 under autocast).  L4P_F32: T = float. */
int l4p_layernorm_t(l4p_stream stream, int dtype, const void* x_T, const float* gamma, const float* beta, float eps,
                    void* out_T, int M, int C, int act);

/* PromptEncoder (prompt_encoder.py:78-121,196-203) + token concat (mask_decoder.py:107-113):
 * tokens float [N][6][C] = 3 mask tokens | point PE + label embedding | not-a-point | feature prompt. */
int l4p_track_tokens(l4p_stream stream, const float* queries, const float* labels, const float* pfeat,
                     const float* plabel, const float* gauss, const float* mask_tokens, const float* point_emb0,
                     const float* point_emb1, const float* not_a_point, const float* feat_emb0, const float* feat_emb1,
                     float* tokens, int N, int C, int T, int H, int W);

/* keys = enc_features[-1] (broadcast over queries) + per-query history (sparse_heads.py:341-346):
 * k32 float, kT = T(keys), kP = T(keys + dense_pe), each [N][P][C].
 * shared_from > 0 (later windows of a long clip: the history rows [shared_from, P) of every track are the learned mask token
 * again, sparse_heads.py:418-427): those rows are formed for track 0 only - hist rows past shared_from of the other tracks
 * are not read, their k32 / kT / kP rows not written - and track 0's float rows are also stored in k32_shared
 * [P - shared_from][C] (the residual master l4p_layernorm_res reads for those rows of every track).  shared_from = 0: every
 * row of every track (k32_shared ignored). */
int l4p_track_keys_init(l4p_stream stream, int dtype, const float* enc, const float* hist, const float* pos,
                        float* k32, void* kT, void* kP, int N, int P, int C, int shared_from, float* k32_shared);

/* Broadcast a C-vector into `rows` rows of a float matrix (row map as in l4p_gemm_desc.a_*): the learned
 * mask token of the memory mechanism (sparse_heads.py:262-265,418-427). */
int l4p_fill_rows(l4p_stream stream, float* out, const float* v, long long rows, int C, long long group_rows,
                  long long group_stride, long long group_off);

/* Copies the `bytes` bytes at base + off to the same offset of the following n - 1 groups (group g starts at base + g * stride;
 * all multiples of 16).  Later tracker windows (sparse_heads.py:406-448): the second temporal half of every track's keys is
 * encoder feature + the same learned mask token, so what layer 0 derives from it (rows [P/2, P) of t2i.k, t2i.v, i2t.q) is
 * computed for track 0 and copied to the other tracks' rows. */
int l4p_broadcast_block(l4p_stream stream, void* base, long long off, long long bytes, long long stride, int n);

/* Attention of sam/transformer.py:223-245 after the projections. kind 0: 6 prompt tokens among themselves
 * (q,k,v,out [N][6][D]); kind 1: tokens -> image (q [N][6][D], k,v [N][P][D], out [N][6][D]);
 * kind 2: image -> tokens (q [N][P][D], k,v [N][6][D], out [N][P][D]);
 * kind 3 / 4: kinds 1 / 2 when every track still has the SAME image-side operand (first window, first layer: the
 * keys have not been touched by a query yet): k,v [P][D] resp. q [P][D] are shared, outputs stay per track. */
int l4p_small_attn(l4p_stream stream, int dtype, int kind, const void* q, const void* k, const void* v, void* out,
                   int N, int P, int D, int heads);

/* masks[n][m][vox] = hyper[n][m][:] . up[n][vox][:]  (mask_decoder.py:139); up channels-last T. */
int l4p_mask_product(l4p_stream stream, int dtype, const void* up, const float* hyper, float* masks, int N,
                     long long vox, int C);

/* masks[n][i][t][y][x] (the [N][3][T][2h][2w] logits of mask_decoder.py:139) from the L4P_EPI_MASKDOT partial sums of
 * the (1,2,2) ConvTranspose over M = N*T*h*w rows: row m = ((n*T + t)*h + y/2)*w + x/2, tap = (y%2)*2 + x%2, chunks_per_tap
 * chunks each. */
int l4p_mask_gather(l4p_stream stream, const float* partial, float* masks, int N, int T, int h, int w, int chunks_per_tap);

/* Folded image -> token attention of the tracker (sam/transformer.py:180-185,223-245; the projections of the 2048 x N image
 * tokens are folded into the 6 prompt tokens of each track, l4p_amd/models/task_heads/sparse_heads.py "folded i2t"):
 * l4p_i2t_probs: scores float [M][ld_scores], column t * heads + h = scaled score of image token m against prompt token t in head h
 *   (pairs != 0: columns tokens * heads + t * heads + h hold a second addend - the scores against the low halves of the folded key
 *   matrix; cbias != NULL: + cbias[(m / rows_per_group) * tokens * heads + t * heads + h], the track's query-bias term)
 *   -> probs T [M][ld_probs]: softmax over t for every h, same column order, columns tokens * heads .. ld_probs - 1 zero.
 * l4p_split_hilo: in float [G][R][C] -> out T [G][2 R][C], rows [0, R) = T(x), rows [R, 2 R) = T(x - T(x)): a folded key matrix as a
 *   PAIR of engine-dtype matrices (an option for its 1408-term products with the keys; off by default: no measurable effect).
 * l4p_transpose_pad: in T [G][R][C] -> out T [G][C][Rp] (k index padded with zeros): a track's folded value matrix in the
 *   k-contiguous form l4p_gemm reads weights in. */
int l4p_i2t_probs(l4p_stream stream, int dtype, const float* scores, long long ld_scores, int pairs, const float* cbias, int rows_per_group,
                  void* probs_T, int ld_probs, long long M, int heads, int tokens);
int l4p_split_hilo(l4p_stream stream, int dtype, const float* in, void* out_T, int G, int R, long long C);
/* Token -> image attention (l4p_small_attn kind 1) from scores formed elsewhere: scores float [N * P][ld_scores], column t * heads + h
 * = scaled score of prompt token t against image token p in head h (the keys' projection folded into the tokens: scores = kP x Q'^T,
 * Q' = q W_k per head; the key bias shifts all P scores of a (t, h) alike and drops out of the softmax); v T [N][P][D];
 * out T [N][6][D] = softmax over p, then P.V. */
int l4p_t2i_attn_scores(l4p_stream stream, int dtype, const float* scores, long long ld_scores, const void* v_T, void* out_T, int N, int P,
                        int D, int heads);
int l4p_transpose_pad(l4p_stream stream, int dtype, const void* in_T, void* out_T, int G, int R, int C, int Rp);
/* Token -> image attention with the value projection folded away as well (sam/transformer.py:168-173,223-245):
 *   out[t, head h] = sum_p prob[t,h,p] (keys[p] Wv_h^T + bv_h) = (sum_p prob[t,h,p] keys[p]) Wv_h^T + bv_h.
 * l4p_t2i_probs: scores float [N][P][ld_scores] (column t * heads + h, 48 of them) -> e T [N][P][48] = exp(score - m) with m the column
 *   maximum over the SPLIT of 256 keys the row belongs to, and stats float [N][ceil(P / 256)][2][48]: those maxima, and the splits'
 *   sums of e AS ROUNDED to T (the weights the context product applies then sum to one exactly).  (One coalesced pass over 8 x N workgroups; the softmax over all P keys is assembled by l4p_t2i_context: a split's
 *   terms carry exp(m_split - M) / Z.)
 * l4p_t2i_context: ctx T [(h * Rg + n * tokens + t)][C] = sum_p softmax_p[n][p][t * heads + h] * keys[n][p][C] (keys T [N][P][C] WITHOUT
 *   the positional term; rows grouped by head with Rg >= N * tokens rows per group, a multiple of 128, rows past N * tokens untouched):
 *   the A operand of the row-grouped-weights GEMM against the head blocks of W_v (w_gr = Rg, w_gs = hd * C, b_gs = o_gs = hd, c_gr = Rg).
 *   heads * tokens == 48, C % 64 == 0, P % 32 == 0, 96 <= P <= 4096.  bf16: MFMA kernel bound by one read of the keys.
 *   shared_from (a multiple of 32; >= P: none): key rows p >= shared_from of EVERY track are read from track 0's block - later windows of
 *   the recursion, layer 0: the second temporal half of every track's keys is still the same (l4p_track_keys_init(shared_from)). */
/* delta = P x V' + b of the folded image -> token attention as a streaming kernel (bf16, K == 64, C % 128 == 0, P % 16 == 0): probs T
 * [N][P][64] (l4p_i2t_probs), vt T [N][C][64] (l4p_transpose_pad), bias float [C] or NULL -> delta T [N][P][C].  Bit-identical to
 * l4p_gemm with w_gr = P, w_gs = C * 64 on the same operands, which serves every other shape / dtype. */
int l4p_i2t_delta(l4p_stream stream, int dtype, const void* probs_T, const void* vt_T, const float* bias, void* delta_T, int N, int P, int C,
                  int K);
int l4p_t2i_probs(l4p_stream stream, int dtype, const float* scores, long long ld_scores, void* probs_T, float* stats, int N, int P, int HT);
int l4p_t2i_context(l4p_stream stream, int dtype, const void* probs_T, const float* stats, const void* keys_T, void* ctx_T, int N, int P,
                    int C, int heads, int tokens, long long Rg, int shared_from);

/* Fused read-out (sparse_heads.py:572-589,645-647): trilinear (align_corners=False) resize of masks
 * [N][3][T][h][w] to H x W, soft-argmax of channel 0 (traj [N][2][T]), spatial mean of channel 1
 * (vis [N][T]) and exp(mean) of channel 2 (depth [N][T]); the up-sampled logits are never stored. */
int l4p_track_readout(l4p_stream stream, const float* masks, float* traj, float* vis, float* depth, int N, int T,
                      int h, int w, int H, int W);

/* Sliding-window state of forward_windowed_core (sparse_heads.py:303-335 prepare; :366-393,:455-486 commit).
 * Integer / boolean results (valid masks, labels {0,1,2}, argmax index) are bit-exact w.r.t. the reference. */
int l4p_track_prepare(l4p_stream stream, const float* cur_q, const float* orig_q, int start, int ws, float* q_off,
                      float* labels, unsigned char* valid_t, unsigned char* valid_n, int N);
int l4p_track_commit(l4p_stream stream, const float* w_traj, const float* w_vis, const float* w_depth,
                     const unsigned char* valid_t, const unsigned char* valid_n, float* traj, float* vis, float* depth,
                     int T, int start, int ws, int next_start, int last_window, float* cur_q, float* plabel,
                     const float* new_pfeat, float* pfeat, int* best_out, int N, int C);

/* ------------------------------------------------------------------------------------------------
 * Engine: holds the table of packed device weights and runs whole sub-networks with one call.
 * ---------------------------------------------------------------------------------------------- */

int l4p_create(int device, int dtype, l4p_engine** out);
int l4p_destroy(l4p_engine* e);

/* Register a packed weight that lives in caller-owned device memory (the Python host packs the
 * reference state_dict — models/utils.py:52-53 — into kernel layouts and keeps the arena alive). */
int l4p_bind_weight(l4p_engine* e, const char* name, const void* dev_ptr, long long numel);

typedef struct l4p_encoder_cfg {
    int dim, depth, heads, head_dim, mlp_hidden;
    int in_chans, frames, img_h, img_w, pt, ph, pw; /* tubelet */
    int patch_kp;                                    /* padded gather width (multiple of 64) */
    float ln_eps;
} l4p_encoder_cfg;

int l4p_encoder_configure(l4p_engine* e, const l4p_encoder_cfg* cfg);
size_t l4p_encoder_workspace_bytes(const l4p_engine* e, int B);

/* VideoMAEEncoder.forward (l4p_videomae.py:80-122): patch embed + sinusoid pos + `depth` pre-LN
 * blocks + final norm.  The reference returns all depth+1 features; here the caller names the taps
 * it wants: tap_layer[i] in [0, depth] (0 = embeddings, k = after k blocks, depth = norm(x_depth)),
 * written as float to tap_f32[i] and/or as T to tap_T[i] (either may be NULL), each [B*tokens][dim].
 * Blocks after the highest requested tap are not executed. */
int l4p_encoder_forward(l4p_engine* e, l4p_stream stream, const float* rgb, int B, void* workspace, size_t ws_bytes,
                        int n_taps, const int* tap_layer, float* const* tap_f32, void* const* tap_T);

/* DPTOutputAdapter_fix.forward (dpt_head.py:41-86) of one dense head as a single call: act_postprocess (1x1x1 conv +
 * ConvTranspose / strided conv), layer_rn, four FeatureFusion blocks, head1, trilinear resize, head2, output
 * projection (+exp).  hooks[i]: T [B][nt*nh*nw][dim] features of the head's hook layers; out: float
 * [B][out_ch][out_t][out_h][out_w].  Weights "dpt.<task>.*" must have been bound with l4p_bind_weight.
 * actpost / fusion are the reference's scale-factor tuples (dense_heads.py:30-31, :269-271). */
typedef struct l4p_dpt_cfg {
    int dim, nt, nh, nw;
    int layer_dims[4];
    int feature_dim, last_dim, out_ch;
    int actpost[4][3];
    int fusion[4][3];
    int out_t, out_h, out_w;
    int post_exp;
} l4p_dpt_cfg;

size_t l4p_dpt_workspace_bytes(const l4p_engine* e, const l4p_dpt_cfg* cfg, int B);
int l4p_dpt_forward(l4p_engine* e, l4p_stream stream, const char* task, const l4p_dpt_cfg* cfg, const void* const* hooks,
                    int B, void* workspace, size_t ws_bytes, float* out);

/* One window of the SAM-style tracker for the N queries of ONE clip as a single call (VideoMAETrack2DSamHead.forward /
 * forward_single_batch, sparse_heads.py:497-667, with PromptEncoder, TwoWayTransformer and MaskDecoder.predict_masks):
 * prompt tokens, key initialisation, sam_depth two-way layers + final attention, hyper-network MLPs, prompt feature for
 * the next window, memory tokens (need_history), up-scaling fused with the mask product, fused up-sample + soft-argmax.
 * enc_last float [P][C] (enc_features[-1] of the clip); hist float [N][P][C] per-query history tokens (read; rewritten in
 * place for the next window when need_history: the second temporal half of the processed tokens projected into rows
 * [0, P/2), rows [P/2, P) set to the learned mask token; need_history == 2: the caller guarantees that rows [P/2, P) already
 * hold the mask token - true for a buffer that was filled with it once and only ever passed to this function, which writes
 * nothing else there - and the fill is skipped; need_history == 3: hist [N][P][C] receives the projection of ALL P processed
 * tokens of every track, the reference's <task>_enc_features_with_track_history_bnpc of a plain single-window forward) — or, with hist_uniform (every track still has the same history rows: the
 * first window / the plain single-window forward), only its first P rows are read; hist_uniform == 2: rows [P/2, P) of every
 * track's history are identical (the state need_history leaves behind: the learned mask token), so the layer-0 projections
 * of those rows are computed once and copied (l4p_broadcast_block) — same values, half the key-side projection work of
 * layer 0; hist_uniform == 2 or 4 name a LATER window of a recursion (4: without the half-sharing of 2, every track on its own rows -
 * the equality check of that shortcut): layer 0's token -> image attention then takes the folded form of the later layers
 * (packing.py fold_t2i: scores against the folded prompt tokens, softmax x keys on the matrix pipe, the head blocks of W_v on the 48
 * context rows) instead of projecting N x P key rows twice and attending per (track, head) - L4P_TRACK_FOLD_L0=0: the projected
 * form; hist_uniform == 0 (a window evaluated out of context) keeps the projected form, bit-identical to the first window's shortcut;
 * q_off float [N][3] (t, x, y) relative
 * to the window; labels, plabel float [N]; pfeat float [N][C].  Outputs: traj float [N][2][T], vis, depth float [N][T],
 * new_pfeat float [N][C].  Weights "trk.*" must have been bound with l4p_bind_weight. */
typedef struct l4p_track_cfg {
    int dim, tokens, nt, nh, nw; /* key geometry: tokens = nt * nh * nw */
    int sam_depth, sam_heads, sam_mlp, out_dim_factor;
    int T, H, W; /* window / image size the masks are read out at */
} l4p_track_cfg;
size_t l4p_track_window_workspace_bytes(const l4p_engine* e, const l4p_track_cfg* cfg, int N, int hist_uniform);
int l4p_track_window_forward(l4p_engine* e, l4p_stream stream, const l4p_track_cfg* cfg, const float* enc_last, float* hist,
                             const float* q_off, const float* labels, const float* pfeat, const float* plabel, int N,
                             int need_history, int hist_uniform, void* workspace, size_t ws_bytes, float* traj, float* vis,
                             float* depth, float* new_pfeat);

/* ------------------------------------------------------------------------------------------------
 * Clip preparation (the caller side of the hot path): decoded uint8 frames in HBM ->
 * the network's input tensor.  Replaces VideoDataset.getitem_helper's per-frame PIL resize-blur-resize +
 * to_tensor (l4p/data/video_dataset.py:86-93) and L4PDataset.__getitem__'s mirror-pad / resize / crop /
 * normalise (l4p/data/l4p_dataset_mini.py:543-587, :126-190, :236-288, :290-391).
 * ------------------------------------------------------------------------------------------------ */

/* HOST function (no GPU work): Pillow's coefficient tables for one axis of Image.resize(BILINEAR) over the
 * full axis (libImaging/Resample.c precompute_coeffs + normalize_coeffs_8bpc: triangle filter, support scaled
 * by the down-scale factor, 22-bit fixed point).  bounds: host int[2*out_size] (first tap, tap count);
 * coeffs: host int[out_size * ksize].  With bounds == NULL or coeffs == NULL only *ksize is written. */
int l4p_pil_coeffs(int in_size, int out_size, int* bounds, int* coeffs, int coeffs_cap, int* ksize);

/* One 8-bit pass of Pillow's resampler on n_img images [in_h][in_w][channels] (uint8, device):
 * axis 1 = horizontal (in_w -> out_size), axis 0 = vertical (in_h -> out_size).  bounds / coeffs: the tables of
 * l4p_pil_coeffs copied to the device.  Integer arithmetic, bit-exact with Pillow. */
int l4p_pil_resample_u8(l4p_stream stream, const unsigned char* src, unsigned char* dst, long long n_img, int in_h,
                        int in_w, int channels, int axis, int out_size, const int* bounds, const int* coeffs, int ksize);

/* rgb_out[c][t][y][x] (float, [3][T_out][out_h][out_w]) =
 *   (bilinear(frame[frame_index[t]] / 255 resized to res_h x res_w)[y + crop_i0][x + crop_j0] - mean[c]) / std[c]
 * with F.interpolate(align_corners=False) arithmetic (skipped, as in the reference, when res == in).  frame_index
 * (device int[T_out]) carries stride, temporal mirror-padding and the temporal crop.  mean3 / std3: HOST float[3],
 * read at call time.  frames: uint8 [n][in_h][in_w][3]; OR, when vbounds != NULL, [n][src_h][in_w][3] = the frames
 * before a final vertical Pillow pass src_h -> in_h (tables vbounds / vcoeffs / vksize on the device), which is then
 * evaluated on the fly for the <= 4 pixels an output pixel needs instead of being written out.
 * Compact rows (x_lambda != NULL): an output column j only reads source columns i0[j], i1[j] (l4p_resize_index_table),
 * so the caller may produce just those — frames is then [n][src_h][2*out_w][3] with columns (i0[0], i1[0], i0[1], ...)
 * and x_lambda the device copy of lambda1[out_w]; the previous horizontal pass shrinks accordingly. */
int l4p_clip_resize_normalize(l4p_stream stream, const unsigned char* frames, const int* frame_index, float* rgb_out,
                              int T_out, int in_h, int in_w, int res_h, int res_w, int crop_i0, int crop_j0, int out_h,
                              int out_w, const float* mean3, const float* std3, int src_h, const int* vbounds,
                              const int* vcoeffs, int vksize, const float* x_lambda);

/* HOST function: the source indices / weight of F.interpolate(align_corners=False) along one axis, exactly as the
 * kernel evaluates them (float32, source coordinate = one fma): output j (0 <= j < out_size) of the crop starting at
 * crop0 of the axis resized in_size -> res_size reads i0[j], i1[j] with weights 1 - lambda1[j], lambda1[j]. */
int l4p_resize_index_table(int in_size, int res_size, int crop0, int out_size, int* i0, int* i1, float* lambda1);

#ifdef __cplusplus
}
#endif
#endif /* L4P_HIP_H */
