"""SAM-style 2D/3D point tracker on the MI355X engine — host mirror of
l4p/models/task_heads/sparse_heads.py (+ sam/prompt_encoder.py, sam/transformer.py, sam/mask_decoder.py).

Same class name, constructor arguments, ``forward_windowed`` / ``forward`` signatures and output keys as
the reference.  All arithmetic — projections (MFMA GEMM), the three small-token attention shapes,
LayerNorms, up-scaling ConvTransposes, hyper-network product, the fused up-sample + soft-argmax read-out
and the integer/boolean sliding-window bookkeeping — runs in libl4p_hip.so; the Python below only
sequences kernels over device buffers (no host synchronisation inside a window).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, List, Literal, Optional, Tuple

import torch

from ... import _lib, ops
from ..._lib import ACT_GELU, ACT_NONE, ACT_RELU, EPI_CONVT, EPI_DENSE, EPI_MASKDOT, L4P_BF16, L4P_F32, GemmDesc
from ...ops import _p, _stream


def _gemm(a: torch.Tensor, M: int, K: int, lda: int, w: torch.Tensor, n: int, *, bias=None, act=ACT_NONE, res1=None,
          out_f32: Optional[torch.Tensor] = None, out_T: Optional[torch.Tensor] = None, ldc: Optional[int] = None,
          a_map=None, c_map=None, a_off: int = 0, f32_off: int = 0, res_mod: int = 0, wgroup=None,
          ldw: Optional[int] = None, ogroup: int = 0, kwin=None) -> None:
    """Raw l4p_gemm call with explicit strides / row maps (see include/l4p_hip.h)."""
    d = GemmDesc()
    es = a.element_size()
    d.A, d.lda, d.W, d.ldw = a.data_ptr() + a_off * es, lda, _p(w), K
    d.M, d.N, d.K = M, n, K
    d.bias, d.act = _p(bias), act
    if res1 is not None:
        d.res1, d.res_f32, d.ldr, d.res_mod = _p(res1), 1, n, res_mod
    d.out_f32 = None if out_f32 is None else out_f32.data_ptr() + 4 * f32_off
    d.out_T = _p(out_T)
    d.ldc = n if ldc is None else ldc
    d.epi = EPI_DENSE
    if a_map is not None:
        d.a_gr, d.a_gs, d.a_go = a_map
    if c_map is not None:
        d.c_gr, d.c_gs, d.c_go = c_map
    if wgroup is not None:  # row-grouped weights: (rows per group, W elements between groups, bias elements between groups)
        d.w_gr, d.w_gs, d.b_gs = wgroup
        d.ldw = K if ldw is None else ldw
        d.o_gs = ogroup  # output elements between groups
    if kwin is not None:    # block-structured weights: (output columns per group, contraction elements per group)
        d.kw_cols, d.kw_len = kwin
    _lib.check(_lib.load().l4p_gemm(_stream(), ops.code_of(a.dtype), C.byref(d)), "l4p_gemm")


class VideoMAETrack2DSamHead(torch.nn.Module):
    def __init__(
        self,
        task_name: str = "track_2d",
        prompt_embed_dim: int = 1408,
        image_size: Tuple[int, int, int] = (16, 224, 224),
        patch_size: Tuple[int, int, int] = (2, 14, 14),
        estimate_vis: bool = False,
        estimate_depth: bool = False,
        sam_head_depth: int = 2,
        decoding_out_dim_factor: int = 8,
        num_prompt_points: int = 2,
        num_point_embeddings: int = 2,
        modify_pointlabels_for_windowing: bool = False,
        prompt_using_features: bool = False,
        attend_to_past: bool = False,
        depth_fn: str = "linear",
        vis_fn: str = "linear",
        estimation_directions: List[Literal[1, -1]] = [1, -1],
        max_queries: int = 192,
    ):
        super().__init__()
        # the engine implements the configuration shipped in configs/model.yaml:53-66
        if not (estimate_vis and estimate_depth and prompt_using_features and attend_to_past and
                modify_pointlabels_for_windowing and num_point_embeddings == 2 and num_prompt_points == 2 and
                depth_fn == "exp" and vis_fn == "linear"):
            raise NotImplementedError("tracker options other than those of configs/model.yaml are not built into the engine")
        self.task_name = task_name
        self.prompt_embed_dim = prompt_embed_dim
        self.image_size = tuple(image_size)
        self.patch_size = tuple(patch_size)
        self.sam_head_depth = sam_head_depth
        self.decoding_out_dim_factor = decoding_out_dim_factor
        self.estimation_directions = list(estimation_directions)
        self.max_queries = max_queries
        self.image_embedding_size = tuple(int(image_size[i] / patch_size[i]) for i in range(3))
        self.video_tokens_size = self.image_embedding_size[0] * self.image_embedding_size[1] * self.image_embedding_size[2]
        self.task_suffix = "_track_2d"
        self._rt = None
        self._engine_task = ""
        self.trace: Optional[list] = None  # set to [] to record per-window labels / queries (tests)

    # ------------------------------------------------------------------------------------------------
    def _single_window_history_rows(self, N: int, P: int) -> int:
        return P  # (a single window only reads the shared first P rows; tests override this to exercise the general path)

    def _w(self, k: str) -> torch.Tensor:
        return self._rt.weights["trk." + k]

    def _ln(self, x32: torch.Tensor, key: str, add: Optional[torch.Tensor], add_mod: int, want_T: bool = True,
            want_T2: bool = True, eps: float = 1e-5, act: int = ACT_NONE, out32: Optional[torch.Tensor] = None):
        M, Cc = x32.shape
        td = ops.torch_dtype(self._rt.dtype)
        oT = torch.empty((M, Cc), dtype=td, device=x32.device) if want_T else None
        oT2 = torch.empty((M, Cc), dtype=td, device=x32.device) if (want_T2 and add is not None) else None
        o32 = x32 if out32 is None else out32
        _lib.check(_lib.load().l4p_layernorm_ex(_stream(), self._rt.dtype, _p(x32), _p(self._w(key + ".g")),
                                                _p(self._w(key + ".b")), eps, _p(oT), _p(o32), M, Cc, _p(add), add_mod,
                                                _p(oT2), act), "l4p_layernorm_ex")
        return o32, oT, oT2

    def _proj(self, x: torch.Tensor, key: str, n: int, **kw) -> torch.Tensor:
        M, K = x.shape
        out = torch.empty((M, n), dtype=x.dtype, device=x.device)
        _gemm(x, M, K, K, self._w(key + ".w"), n, bias=self._w(key + ".b"), out_T=out, **kw)
        return out

    def _attn(self, kind: int, q, k, v, N: int, P: int, D: int) -> torch.Tensor:
        out = torch.empty_like(q)
        _lib.check(_lib.load().l4p_small_attn(_stream(), ops.code_of(q.dtype), kind, _p(q), _p(k), _p(v), _p(out), N, P, D,
                                              self._rt.cfg.sam_heads), "l4p_small_attn")
        return out

    # ------------------------------------------------------------------------------------------------
    def _window(self, enc_last: torch.Tensor, hist: torch.Tensor, q_off: torch.Tensor, labels: torch.Tensor,
                pfeat: torch.Tensor, plabel: torch.Tensor, need_history: bool, hist_uniform: int = 0):
        """forward / forward_single_batch (sparse_heads.py:497-667) for N queries of one clip.
        enc_last: float [P,C]; hist: float [N,P,C]; returns window traj [N,2,T], vis [N,T], depth [N,T],
        new prompt features [N,C]; updates ``hist`` in place when ``need_history``.
        ``hist_uniform``: every track has the same history rows (first window: the learned mask token), so until the
        first image->token update the keys are ONE [P,C] set: the first layer's t2i.k / t2i.v / i2t.q projections and
        the key initialisation run once instead of N times (identical rows in, identical rows out)."""
        rt = self._rt
        cfg, dt = rt.cfg, rt.dtype
        eng = getattr(rt, "engine", None)
        if eng is not None and not os.environ.get("L4P_TRACK_PYTHON"):
            # the whole window as ONE native call (csrc/api_trackwin.hip: the same kernels in the same order as below)
            tc = _lib.TrackCfg(dim=cfg.dim, tokens=cfg.tokens, nt=cfg.grid[0], nh=cfg.grid[1], nw=cfg.grid[2],
                               sam_depth=cfg.sam_depth, sam_heads=cfg.sam_heads, sam_mlp=cfg.sam_mlp,
                               out_dim_factor=self.decoding_out_dim_factor, T=self.image_size[0], H=self.image_size[1],
                               W=self.image_size[2])
            return eng.track_window(tc, enc_last, hist, q_off, labels, pfeat, plabel, need_history, hist_uniform,
                                    slot=getattr(self, "_ws_slot", 0))
        lib = _lib.load()
        dev = enc_last.device
        td = ops.torch_dtype(dt)
        N = q_off.shape[0]
        P, Cc = cfg.tokens, cfg.dim
        Dh = Cc // 2
        T, H, W = self.image_size
        f32 = dict(dtype=torch.float32, device=dev)

        tok32 = torch.empty((6 * N, Cc), **f32)
        _lib.check(lib.l4p_track_tokens(_stream(), _p(q_off), _p(labels), _p(pfeat), _p(plabel), _p(self._w("gauss")),
                                        _p(self._w("mask_tokens")), _p(self._w("point_emb0")), _p(self._w("point_emb1")),
                                        _p(self._w("not_a_point")), _p(self._w("feat_emb0")), _p(self._w("feat_emb1")),
                                        _p(tok32), N, Cc, T, H, W), "l4p_track_tokens")
        tokT = torch.empty((6 * N, Cc), dtype=td, device=dev)
        _lib.check(lib.l4p_cast(_stream(), dt, _p(tok32), _p(tokT), tok32.numel()), "l4p_cast")

        pos = self._w("dense_pe")
        # hist_uniform == 2: rows [P/2, P) of every track's keys coincide until the first image -> token update
        half_shared = hist_uniform == 2 and N > 1 and P % 2 == 0
        # hist_uniform 2 / 4: a later window of a recursion - layer 0's token -> image attention in the folded form (csrc/api_trackwin.hip)
        fold_l0 = hist_uniform in (2, 4) and os.environ.get("L4P_TRACK_FOLD_L0", "1") != "0"
        if hist_uniform in (2, 4):
            hist_uniform = 0

        def proj_half_shared(x: torch.Tensor, key: str, n: int) -> torch.Tensor:
            """Projection of per-track keys [N*P, C] whose second temporal half is common to all tracks: the first halves of
            all tracks (row-mapped GEMM), track 0's second half, and a copy of that block to the other tracks."""
            half = P // 2
            o = torch.empty((N * P, n), dtype=x.dtype, device=x.device)
            w, b = self._w(key + ".w"), self._w(key + ".b")
            _gemm(x, N * half, Cc, Cc, w, n, bias=b, out_T=o, a_map=(half, P, 0), c_map=(half, P, 0))
            _gemm(x, half, Cc, Cc, w, n, bias=b, out_T=o, a_map=(half, P, half), c_map=(half, P, half))
            es = o.element_size()
            _lib.check(lib.l4p_broadcast_block(_stream(), _p(o), half * n * es, half * n * es, P * n * es, N), "l4p_broadcast_block")
            return o

        Nk = 1 if hist_uniform else N  # distinct key sets before the first image -> token update
        k32 = torch.empty((Nk * P, Cc), **f32)
        kT = torch.empty((Nk * P, Cc), dtype=td, device=dev)
        kP = torch.empty((Nk * P, Cc), dtype=td, device=dev)
        # half_shared: rows [P/2, P) are formed for track 0 only; their float master (read by layer 0's key LayerNorm for every
        # track) goes to kh32 (csrc/api_trackwin.hip has the same sequence)
        kh32 = torch.empty((P // 2, Cc), **f32) if half_shared else None
        _lib.check(lib.l4p_track_keys_init(_stream(), dt, _p(enc_last), _p(hist), _p(pos), _p(k32), _p(kT), _p(kP), Nk, P, Cc,
                                           P // 2 if half_shared else 0, _p(kh32) if half_shared else None), "l4p_track_keys_init")

        # token -> image attention with the keys' projection folded into the tokens (packing.py fold_t2i; csrc/api_trackwin.hip has
        # the same sequence): Q' = q_tok x kfold^T, scores = kP x Q'^T (row-grouped weights), softmax over the keys + P.V
        HTk = 6 * cfg.sam_heads
        fold_t2i_ok = os.environ.get("L4P_TRACK_FOLD_T2I", "1") != "0" and P % 128 == 0 and HTk <= 64

        # ... and the value projection (l4p_t2i_context): out = (probs x keys) Wv_h^T + bv_h - probs [N][P][HT], the context of every
        # (token, head) against the keys without the positional term, then each head's 6 N context rows through its block of W_v
        fold_v = (fold_t2i_ok and os.environ.get("L4P_TRACK_FOLD_T2I_V", "1") != "0" and HTk == 48 and Cc % 128 == 0 and (Dh // cfg.sam_heads) % 8 == 0 and P % 32 == 0
                  and 96 <= P <= 4096)
        RgT = (6 * N + 127) // 128 * 128
        # the folded weights are block-structured (packing.py: head h's C columns meet head h's inputs only): l4p_gemm_desc.kw_cols
        kwin = (Cc, Dh // cfg.sam_heads) if Cc % 128 == 0 and os.environ.get("L4P_TRACK_KWIN", "1") != "0" else None

        def t2i_folded(tq: torch.Tensor, prefix: str, keysP: torch.Tensor, keysT: torch.Tensor, hs: bool = False) -> torch.Tensor:
            KW = cfg.sam_heads * Cc
            qf = torch.empty((N * HTk + 128, Cc), dtype=td, device=dev)  # Q' [N][HT][C] (+ slack rows under the last tile)
            _gemm(tq, 6 * N, Dh, Dh, self._w(prefix + ".kfold.w"), KW, out_T=qf, ldc=KW, kwin=kwin)
            sc = torch.empty((N * P, HTk), **f32)
            if hs:  # rows [P/2, P) exist for track 0 only: two row-mapped launches over the half blocks
                half = P // 2
                _gemm(keysP, N * half, Cc, Cc, qf, HTk, out_f32=sc, ldc=HTk, wgroup=(half, HTk * Cc, 0), ldw=Cc,
                      a_map=(half, P, 0), c_map=(half, P, 0))
                _gemm(keysP, N * half, Cc, Cc, qf, HTk, out_f32=sc, ldc=HTk, wgroup=(half, HTk * Cc, 0), ldw=Cc,
                      a_map=(half, 0, half), c_map=(half, P, half))
            else:
                _gemm(keysP, N * P, Cc, Cc, qf, HTk, out_f32=sc, ldc=HTk, wgroup=(P, HTk * Cc, 0), ldw=Cc)
            ta = torch.empty((RgT, Dh), dtype=td, device=dev)  # (rows past 6 N: scratch of the head groups' padding rows)
            if fold_v:
                hd = Dh // cfg.sam_heads
                pr = torch.empty((N * P, HTk), dtype=td, device=dev)
                cx = torch.empty((cfg.sam_heads * RgT, Cc), dtype=td, device=dev)
                stt = torch.empty((N * ((P + 255) // 256), 2 * HTk), **f32)  # per 256-key split: column maxima, sums
                _lib.check(lib.l4p_t2i_probs(_stream(), dt, _p(sc), HTk, _p(pr), _p(stt), N, P, HTk), "l4p_t2i_probs")
                _lib.check(lib.l4p_t2i_context(_stream(), dt, _p(pr), _p(stt), _p(keysT), _p(cx), N, P, Cc, cfg.sam_heads, 6, RgT,
                                               P // 2 if hs else P), "l4p_t2i_context")
                _gemm(cx, cfg.sam_heads * RgT, Cc, Cc, self._w(prefix + ".v.w"), hd, bias=self._w(prefix + ".v.b"), out_T=ta, ldc=Dh,
                      wgroup=(RgT, hd * Cc, hd), c_map=(RgT, 0, 0), ogroup=hd)
            else:
                tv = proj_half_shared(keysT, prefix + ".v", Dh) if hs else self._proj(keysT, prefix + ".v", Dh)
                _lib.check(lib.l4p_t2i_attn_scores(_stream(), dt, _p(sc), HTk, _p(tv), _p(ta), N, P, Dh, cfg.sam_heads),
                           "l4p_t2i_attn_scores")
            return ta

        q32: Optional[torch.Tensor] = None
        qT, qP = tokT, tokT
        x32 = torch.empty((6 * N, Cc), **f32)
        # chained key LayerNorm (csrc/api_trackwin.hip has the same sequence): in a first window layer 0 keeps its update, its
        # engine-dtype results and (mean, rstd) per row; layer 1 re-derives layer 0's float result from those - no float key master
        chain_on = os.environ.get("L4P_TRACK_LN_CHAIN", "1") != "0" and Cc <= 1536
        chain = None  # (shared float rows, the previous layer's update, its statistics, its norm's name)
        for l in range(cfg.sam_depth):
            lo = f"l{l}."
            shared = Nk == 1 and N > 1  # keys still common to all tracks (only in layer 0 of a first window)
            # --- self attention of the prompt tokens (transformer.py:159-166) ---
            sq = self._proj(qP, lo + "self.q", Cc)
            sk = self._proj(qP, lo + "self.k", Cc)
            sv = self._proj(qT, lo + "self.v", Cc)
            sa = self._attn(0, sq, sk, sv, N, 6, Cc)
            _gemm(sa, 6 * N, Cc, Cc, self._w(lo + "self.out.w"), Cc, bias=self._w(lo + "self.out.b"), res1=q32, out_f32=x32)
            q32, qT, qP = self._ln(x32, lo + "norm1", tok32, 6 * N, out32=torch.empty_like(x32))
            # --- tokens -> image (transformer.py:168-173) ---
            tq = self._proj(qP, lo + "t2i.q", Dh)
            hs = half_shared and l == 0
            if fold_t2i_ok and (l >= 1 or fold_l0) and not shared and (not hs or P % 256 == 0):  # (see csrc/api_trackwin.hip)
                ta = t2i_folded(tq, lo + "t2i", kP, kT, hs)
            else:
                tv = proj_half_shared(kT, lo + "t2i.v", Dh) if hs else self._proj(kT, lo + "t2i.v", Dh)
                tk = proj_half_shared(kP, lo + "t2i.k", Dh) if hs else self._proj(kP, lo + "t2i.k", Dh)
                ta = self._attn(3 if shared else 1, tq, tk, tv, N, P, Dh)
                del tk, tv
            _gemm(ta, 6 * N, Dh, Dh, self._w(lo + "t2i.out.w"), Cc, bias=self._w(lo + "t2i.out.b"), res1=q32, out_f32=x32)
            q32, qT, qP = self._ln(x32, lo + "norm2", tok32, 6 * N, out32=torch.empty_like(x32))
            # --- MLP (transformer.py:175-178), ReLU ---
            hdn = self._proj(qT, lo + "mlp1", cfg.sam_mlp, act=ACT_RELU)
            _gemm(hdn, 6 * N, cfg.sam_mlp, cfg.sam_mlp, self._w(lo + "mlp2.w"), Cc, bias=self._w(lo + "mlp2.b"), res1=q32,
                  out_f32=x32)
            q32, qT, qP = self._ln(x32, lo + "norm3", tok32, 6 * N, out32=torch.empty_like(x32))
            # --- image -> tokens (transformer.py:180-185): keys are updated in place ---
            ik = self._proj(qP, lo + "i2t.k", Dh)
            iv = self._proj(qT, lo + "i2t.v", Dh)
            # keys = norm4(keys + out_proj(attention)): the update leaves the projection in the engine dtype (delta), the LayerNorm
            # forms the sum (l4p_layernorm_res; csrc/api_trackwin.hip has the same sequence).  While the keys are still common to
            # all tracks the float residual is row m % P of the common set; from here on every track owns its keys.
            delta = torch.empty((N * P, Cc), dtype=td, device=dev)
            heads = cfg.sam_heads
            HT = 6 * heads
            HTp = (HT + 63) // 64 * 64
            fold = os.environ.get("L4P_TRACK_FOLD_I2T", "1") != "0" and P % 256 == 0 and HT <= 64
            if fold:
                # folded form (packing.py fold_i2t; csrc/api_trackwin.hip): the projections of the image tokens are folded into
                # the 6 prompt tokens of each track - scores = kP x K'^T + c, softmax over the tokens of each head, delta = P x V' + b
                KW = heads * Cc
                # optional (bf16 engine): K' as a pair of bf16 matrices (hi, lo; l4p_split_hilo); the two score halves and c are
                # summed in l4p_i2t_probs
                pair = dt != L4P_F32 and os.environ.get("L4P_TRACK_FOLD_PAIR") == "1"  # (measured: no accuracy effect; off)
                NS = 2 * HT if pair else HT
                kf = torch.empty((N * NS + 128, Cc), dtype=td, device=dev)  # K' [N][NS][C] (+ slack rows under the last tile)
                vf = torch.empty((N * HT, Cc), dtype=td, device=dev)        # V' [N][HT][C]
                cf = torch.empty((6 * N, heads), **f32)                     # c  [N][HT]
                if pair:
                    kf32 = torch.empty((6 * N, KW), **f32)
                    _gemm(ik, 6 * N, Dh, Dh, self._w(lo + "i2t.qfold.w"), KW, out_f32=kf32, ldc=KW)
                else:
                    _gemm(ik, 6 * N, Dh, Dh, self._w(lo + "i2t.qfold.w"), KW, out_T=kf, ldc=KW, kwin=kwin)
                _gemm(iv, 6 * N, Dh, Dh, self._w(lo + "i2t.ofold.w"), KW, out_T=vf, ldc=KW, kwin=kwin)
                _gemm(ik, 6 * N, Dh, Dh, self._w(lo + "i2t.cfold.w"), heads, out_f32=cf, ldc=heads)
                if pair:
                    _lib.check(lib.l4p_split_hilo(_stream(), dt, _p(kf32), _p(kf), N, HT, Cc), "l4p_split_hilo")
                    del kf32
                vt = torch.empty((N * Cc + 128, HTp), dtype=td, device=dev)  # V'^T [N][C][HTp]
                _lib.check(lib.l4p_transpose_pad(_stream(), dt, _p(vf), _p(vt), N, HT, Cc, HTp), "l4p_transpose_pad")
                sc = torch.empty((N * P, NS), **f32)
                if shared:    # the keys are still common to all tracks: every track reads row m % P of the common set
                    _gemm(kP, N * P, Cc, Cc, kf, NS, out_f32=sc, ldc=NS, wgroup=(P, NS * Cc, 0), ldw=Cc, a_map=(P, 0, 0))
                elif hs:      # later windows, layer 0: the second temporal half of every track's keys is track 0's
                    half = P // 2
                    _gemm(kP, N * half, Cc, Cc, kf, NS, out_f32=sc, ldc=NS, wgroup=(half, NS * Cc, 0), ldw=Cc,
                          a_map=(half, P, 0), c_map=(half, P, 0))
                    _gemm(kP, N * half, Cc, Cc, kf, NS, out_f32=sc, ldc=NS, wgroup=(half, NS * Cc, 0), ldw=Cc,
                          a_map=(half, 0, half), c_map=(half, P, half))
                else:
                    _gemm(kP, N * P, Cc, Cc, kf, NS, out_f32=sc, ldc=NS, wgroup=(P, NS * Cc, 0), ldw=Cc)
                pr = torch.empty((N * P, HTp), dtype=td, device=dev)
                _lib.check(lib.l4p_i2t_probs(_stream(), dt, _p(sc), NS, 1 if pair else 0, _p(cf), P, _p(pr), HTp, N * P, heads, 6),
                           "l4p_i2t_probs")
                if (os.environ.get("L4P_TRACK_DELTA_KERNEL", "1") != "0" and dt != L4P_F32 and HTp == 64 and Cc % 128 == 0
                        and P % 16 == 0):  # its own streaming kernel, bit-identical to the GEMM (csrc/track.hip i2t_delta_kernel)
                    _lib.check(lib.l4p_i2t_delta(_stream(), dt, _p(pr), _p(vt), _p(self._w(lo + "i2t.out.b")), _p(delta), N, P, Cc, HTp),
                               "l4p_i2t_delta")
                else:
                    _gemm(pr, N * P, HTp, HTp, vt, Cc, bias=self._w(lo + "i2t.out.b"), out_T=delta, ldc=Cc, wgroup=(P, Cc * HTp, 0),
                          ldw=HTp)
                del kf, vf, cf, vt, sc, pr
            else:
                iq = proj_half_shared(kP, lo + "i2t.q", Dh) if hs else self._proj(kP, lo + "i2t.q", Dh)
                ia = torch.empty((N * P, Dh), dtype=td, device=dev)
                _lib.check(lib.l4p_small_attn(_stream(), dt, 4 if shared else 2, _p(iq), _p(ik), _p(iv), _p(ia), N, P, Dh,
                                              cfg.sam_heads), "l4p_small_attn")
                del iq
                _gemm(ia, N * P, Dh, Dh, self._w(lo + "i2t.out.w"), Cc, bias=self._w(lo + "i2t.out.b"), out_T=delta)
                del ia
            k_res = k32
            chain_next = chain_on and shared and l + 1 < cfg.sam_depth
            if shared:
                k32 = None if chain_next else torch.empty((N * P, Cc), **f32)
                kT = torch.empty((N * P, Cc), dtype=td, device=dev)
                kP = torch.empty((N * P, Cc), dtype=td, device=dev)
                Nk = N
            # (after the last layer nothing adds to the float keys any more: only the T copies are written)
            want32 = l + 1 < cfg.sam_depth and not chain_next
            if chain is not None:
                xs, dprev, stats, pnorm = chain
                if want32:
                    k32 = torch.empty((N * P, Cc), **f32)
                _lib.check(lib.l4p_layernorm_chain(_stream(), dt, _p(xs), P, _p(dprev), _p(stats), _p(self._w(pnorm + ".g")),
                                                   _p(self._w(pnorm + ".b")), _p(delta), _p(self._w(lo + "norm4.g")), _p(self._w(lo + "norm4.b")),
                                                   1e-5, _p(kT), _p(k32) if want32 else None, N * P, Cc, _p(pos), P, _p(kP)),
                           "l4p_layernorm_chain")
                chain = None
            else:
                stats = torch.empty((N * P, 2), **f32) if chain_next else None
                _lib.check(lib.l4p_layernorm_res(_stream(), dt, _p(k_res), P if shared else 0, _p(delta), _p(self._w(lo + "norm4.g")),
                                                 _p(self._w(lo + "norm4.b")), 1e-5, _p(kT), _p(k32) if want32 else None,
                                                 N * P, Cc, _p(pos), P, _p(kP), _p(kh32) if (half_shared and l == 0) else None, P, P // 2,
                                                 _p(stats) if chain_next else None), "l4p_layernorm_res")
                if chain_next:
                    chain = (k_res, delta, stats, lo + "norm4")
            del delta, k_res
        # --- final tokens -> image attention (transformer.py:103-109) ---
        fq = self._proj(qP, "final.q", Dh)
        if fold_t2i_ok and Nk == N:
            fa = t2i_folded(fq, "final", kP, kT)
        else:
            fv = self._proj(kT, "final.v", Dh)
            fk = self._proj(kP, "final.k", Dh)
            fa = self._attn(1, fq, fk, fv, N, P, Dh)
            del fk, fv
        del kP
        _gemm(fa, 6 * N, Dh, Dh, self._w("final.out.w"), Cc, bias=self._w("final.out.b"), res1=q32, out_f32=x32)
        _, hsT, _ = self._ln(x32, "norm_final", None, 0, want_T2=False)

        # --- hyper-network MLPs on the 3 mask tokens (mask_decoder.py:130-133,160-180) ---
        d1 = Cc // self.decoding_out_dim_factor
        d1p = (d1 + 31) // 32 * 32  # channels per tap of the padded up1 weight (packing.py)
        cpt = d1p // 32             # 32-column chunks per tap
        hyper = torch.zeros((N, 3, d1p), **f32)
        for i in range(3):
            h1 = torch.empty((N, Cc), dtype=td, device=dev)
            _gemm(hsT, N, Cc, 6 * Cc, self._w(f"hyper{i}.0.w"), Cc, bias=self._w(f"hyper{i}.0.b"), act=ACT_RELU, out_T=h1,
                  a_off=i * Cc)
            h2 = self._proj(h1, f"hyper{i}.1", Cc, act=ACT_RELU)
            _gemm(h2, N, Cc, Cc, self._w(f"hyper{i}.2.w"), d1, bias=self._w(f"hyper{i}.2.b"), out_f32=hyper, ldc=3 * d1p,
                  f32_off=i * d1p)
        # prompt feature for the next window (sparse_heads.py:650-658): io token 5
        new_pfeat = torch.empty((N, Cc), **f32)
        _gemm(hsT, N, Cc, 6 * Cc, self._w("prompt_lin.w"), Cc, bias=self._w("prompt_lin.b"), out_f32=new_pfeat, a_off=5 * Cc)

        # --- memory tokens for the next window (sparse_heads.py:406-448,660-665): project the 2nd temporal half of the
        #     processed video tokens into the 1st half of the history, pad the rest with the learned mask token ---
        if int(need_history) == 3:  # every processed token projected (the single-window forward's ..._with_track_history_bnpc)
            _gemm(kT, N * P, Cc, Cc, self._w("history_proj.w"), Cc, bias=self._w("history_proj.b"), out_f32=hist)
        elif need_history:
            half = P // 2
            _gemm(kT, N * half, Cc, Cc, self._w("history_proj.w"), Cc, bias=self._w("history_proj.b"), out_f32=hist,
                  a_map=(half, P, half), c_map=(half, P, 0))
            if int(need_history) != 2:  # (2: rows [P/2, P) still hold the mask token, see l4p_track_window_forward)
                _lib.check(lib.l4p_fill_rows(_stream(), _p(hist), _p(self._w("history_mask_token")), N * half, Cc, half, P, half),
                           "l4p_fill_rows")

        # --- output up-scaling (mask_decoder.py:58-66,136-137) on channels-last tokens ---
        nt, nh, nw = cfg.grid
        d0 = min(2 * Cc // self.decoding_out_dim_factor, Cc)
        # the ConvTranspose writes its activation in the engine dtype and LayerNorm3d + GELU normalises it in place: the
        # [N,16,32,32,352] tensor is never held in float (1.5 GB per clip less HBM traffic at bf16; float in the f32 engine)
        u0T = torch.empty((N * nt * 2 * nh * 2 * nw * 2, d0), dtype=td, device=dev)
        dsc = GemmDesc()
        dsc.A, dsc.lda, dsc.W, dsc.ldw = _p(kT), Cc, _p(self._w("up0.w")), Cc
        dsc.M, dsc.N, dsc.K = N * P, 8 * d0, Cc
        dsc.Ti, dsc.Hi, dsc.Wi = nt, nh, nw
        dsc.bias = _p(self._w("up0.b"))
        dsc.out_T = _p(u0T)
        dsc.epi, dsc.kt, dsc.kh, dsc.kw, dsc.Cout = EPI_CONVT, 2, 2, 2, d0
        _lib.check(lib.l4p_gemm(_stream(), dt, C.byref(dsc)), "l4p_gemm(up0)")
        del kT, k32
        _lib.check(lib.l4p_layernorm_t(_stream(), dt, _p(u0T), _p(self._w("up_ln.g")), _p(self._w("up_ln.b")), 1e-6, _p(u0T),
                                       u0T.shape[0], d0, ACT_GELU), "l4p_layernorm_t(up)")
        # up1 (ConvTranspose (1,2,2) + GELU) fused with the hyper-network mask product (mask_decoder.py:136-139): the
        # [N,16,64,64,176] activation is never written; the GEMM epilogue leaves 3 partial sums per 32-column chunk
        Tl, hl, wl = nt * 2, nh * 4, nw * 4
        M1 = u0T.shape[0]
        partial = torch.empty((4 * cpt, 3, M1), **f32)
        dsc = GemmDesc()
        dsc.A, dsc.lda, dsc.W, dsc.ldw = _p(u0T), d0, _p(self._w("up1.w")), d0
        dsc.M, dsc.N, dsc.K = M1, 4 * d1p, d0
        dsc.bias, dsc.act = _p(self._w("up1.b")), ACT_GELU
        dsc.out_f32 = _p(partial)
        dsc.epi, dsc.Cout = EPI_MASKDOT, d1p
        dsc.hyper, dsc.hyper_rows = _p(hyper), M1 // N
        _lib.check(lib.l4p_gemm(_stream(), dt, C.byref(dsc)), "l4p_gemm(up1 + mask product)")
        del u0T
        masks = torch.empty((N, 3, Tl, hl, wl), **f32)
        _lib.check(lib.l4p_mask_gather(_stream(), _p(partial), _p(masks), N, Tl, nh * 2, nw * 2, cpt), "l4p_mask_gather")
        del partial
        assert Tl == T, "temporal size of the decoded masks must equal the window length"
        traj = torch.empty((N, 2, T), **f32)
        vis = torch.empty((N, T), **f32)
        dep = torch.empty((N, T), **f32)
        _lib.check(lib.l4p_track_readout(_stream(), _p(masks), _p(traj), _p(vis), _p(dep), N, T, hl, wl, H, W),
                   "l4p_track_readout")
        return traj, vis, dep, new_pfeat

    # ------------------------------------------------------------------------------------------------
    def forward_windowed(self, enc_features_bpc_2dlist, track_2d_pointquerries_bn3: torch.Tensor,
                         track_2d_pointlabels_bn: torch.Tensor, time_strides: Optional[torch.Tensor] = None,
                         **kwargs) -> Dict[str, torch.Tensor]:
        """Chunks of ``max_queries`` (sparse_heads.py:162-211)."""
        N = track_2d_pointquerries_bn3.shape[1]
        if N < self.max_queries:
            return self.forward_windowed_core(enc_features_bpc_2dlist, track_2d_pointquerries_bn3, track_2d_pointlabels_bn,
                                              time_strides, **kwargs)
        outs = []
        for i in range(int(math.ceil(N / self.max_queries))):
            sl = slice(i * self.max_queries, (i + 1) * self.max_queries)
            outs.append(self.forward_windowed_core(enc_features_bpc_2dlist, track_2d_pointquerries_bn3[:, sl],
                                                   track_2d_pointlabels_bn[:, sl], time_strides, **kwargs))
        # The concatenation below reads every chunk's buffers on the launching stream: with a deferred join the clip streams
        # may still be writing them (and the chunk buffers would go back to the allocator while in use).  Join here; the
        # chunks of one clip already ran back to back on that clip's stream.
        self.join_streams()
        return {k: torch.cat([o[k] for o in outs], dim=1) for k in outs[0]}

    def forward_windowed_core(self, enc_features_bpc_2dlist, track_2d_pointquerries_bn3: torch.Tensor,
                              track_2d_pointlabels_bn: torch.Tensor, time_strides: Optional[torch.Tensor] = None,
                              **kwargs) -> Dict[str, torch.Tensor]:
        """Causal sliding-window tracking (sparse_heads.py:213-495), estimation_directions == [1]."""
        if self._rt is None:
            raise RuntimeError("tracker head has no weights: call load_state_dict on the model first")
        assert len(self.estimation_directions) == 1 and self.estimation_directions[0] == 1, (
            "Currently only positive direction estimation is supported for sliding window tracking.")
        if time_strides is None:  # if windowing is not needed just do a forward pass (sparse_heads.py:223-227)
            return self.forward(enc_features_bpc_2dlist[0], track_2d_pointquerries_bn3, track_2d_pointlabels_bn)
        lib = _lib.load()
        cfg = self._rt.cfg
        dev = enc_features_bpc_2dlist[0].f32(-1).device
        ws = self.image_size[0]
        B = track_2d_pointquerries_bn3.shape[0]
        N = track_2d_pointquerries_bn3.shape[1]
        T = int(time_strides[-1]) + ws
        if N == 0:  # no queries: the reference's buffers with an empty query axis (sparse_heads.py:233-239), no kernel to launch
            z = dict(dtype=torch.float32, device=dev)
            return {f"{self.task_name}_traj_est_bn2t": torch.zeros(B, 0, 2, T, **z),
                    f"{self.task_name}_vis_est_bn1t": torch.full((B, 0, 1, T), -10.0, **z),
                    f"{self.task_name}_depth_est_bn1t": torch.zeros(B, 0, 1, T, **z)}
        P, Cc = cfg.tokens, cfg.dim
        f32 = dict(dtype=torch.float32, device=dev)
        def output_buffers():
            return (torch.zeros(B, N, 2, T, **f32), torch.full((B, N, 1, T), -10.0, **f32), torch.zeros(B, N, 1, T, **f32))

        # (start_event: the clip streams wait for that event only, not for what the launching stream has queued since - the fills
        #  of the output buffers must then run on a clip stream too, or they would land behind that queue and wipe the results)
        early = (getattr(self, "start_event", None) is not None and bool(getattr(self, "own_stream", False)) and dev.type == "cuda"
                 and os.environ.get("L4P_TRACK_STREAMS", "1") != "0")
        traj_all = vis_all = dep_all = None
        if not early:
            traj_all, vis_all, dep_all = output_buffers()
        nwin = len(time_strides)
        # Clips are independent (the reference asserts B == 1, sparse_heads.py:241).  Each clip's tracker runs on its own
        # HIP stream: its many token-side launches are tiny (M = 6N rows -> a few dozen workgroups) and leave most CUs
        # idle, so the clips fill each other's gaps.
        def run_clip(b: int) -> None:
            orig_q = track_2d_pointquerries_bn3[b].to(**f32).contiguous()
            cur_q = orig_q.clone()
            pfeat = torch.zeros(N, Cc, **f32)
            plabel = torch.zeros(N, **f32)
            # history tokens: the learned mask token everywhere before the first window; a single window only ever reads
            # the first P rows (shared keys), so the per-track copy is not materialised for it
            hrows = N * P if nwin > 1 else self._single_window_history_rows(N, P)
            hist = torch.empty(hrows, Cc, **f32)
            _lib.check(lib.l4p_fill_rows(_stream(), _p(hist), _p(self._w("history_mask_token")), hrows, Cc, hrows, 0, 0),
                       "l4p_fill_rows")
            q_off = torch.empty(N, 3, **f32)
            labels = torch.empty(N, **f32)
            valid_t = torch.empty(N, ws, dtype=torch.uint8, device=dev)
            valid_n = torch.empty(N, dtype=torch.uint8, device=dev)
            best = torch.zeros(N, dtype=torch.int32, device=dev)
            traj_b, vis_b, dep_b = traj_all[b], vis_all[b, :, 0], dep_all[b, :, 0]  # (late binding: set before any clip runs)
            for wi in range(nwin):
                start = int(time_strides[wi])
                last = wi == nwin - 1
                nxt = int(time_strides[wi + 1]) if not last else start
                _lib.check(lib.l4p_track_prepare(_stream(), _p(cur_q), _p(orig_q), start, ws, _p(q_off), _p(labels),
                                                 _p(valid_t), _p(valid_n), N), "l4p_track_prepare")
                if self.trace is not None:
                    # (recorded on whatever stream the clip runs on: the clones are ordered behind track_prepare there, so
                    # tracing does not change the schedule — the benchmarked stream configuration can be traced as it runs)
                    self.trace.append({"clip": b, "window": wi, "labels": labels.clone(), "queries": q_off.clone(),
                                       "prompt_labels": plabel.clone(), "valid_t": valid_t.clone()})
                enc_last = enc_features_bpc_2dlist[wi].f32(-1)[b].contiguous()
                # first window: the history of every track is the learned mask token (filled above) -> shared keys
                # later windows: the second temporal half of every track's history is the mask token again (written by the
                # previous window's memory update) -> what layer 0 derives from those rows is computed once (L4P_TRACK_HALF_SHARE=0:
                # every track on its own, the A/B and equality check)
                hu = 1 if wi == 0 else (2 if os.environ.get("L4P_TRACK_HALF_SHARE", "1") != "0" else 4)  # (4: a later window, every track on its own rows)
                # (need_history = 2: hist was filled with the mask token above and only this loop writes it - the memory update
                #  of a window rewrites rows [0, P/2) of each track, nothing touches rows [P/2, P) - so the re-fill is skipped)
                w_traj, w_vis, w_dep, new_pfeat = self._window(enc_last, hist, q_off, labels, pfeat, plabel, 0 if last else 2,
                                                               hist_uniform=hu)
                _lib.check(lib.l4p_track_commit(_stream(), _p(w_traj), _p(w_vis), _p(w_dep), _p(valid_t), _p(valid_n),
                                                traj_b.data_ptr(), vis_b.data_ptr(), dep_b.data_ptr(), T, start, ws, nxt,
                                                1 if last else 0, _p(cur_q), _p(plabel), _p(new_pfeat), _p(pfeat), _p(best),
                                                N, Cc), "l4p_track_commit")
                if self.trace is not None and not last:
                    self.trace[-1]["best_vis_id"] = best.clone()

        # (own_stream: a single clip goes to a stream of its own as well - parallel.forward_windows_sharded starts the recursion
        #  of a long video before the dense decoders and joins it before the seam alignment)
        use_streams = ((B > 1 or bool(getattr(self, "own_stream", False))) and dev.type == "cuda"
                       and os.environ.get("L4P_TRACK_STREAMS", "1") != "0")
        if use_streams:
            main = torch.cuda.current_stream()
            pool = getattr(self, "clip_stream_override", None)  # (the caller's streams: parallel.forward_windows_sharded, CU-masked)
            if pool is None or len(pool) < B:
                pool = getattr(self, "_clip_streams", None)
            if pool is None or len(pool) < B:
                # (L4P_TRACK_PRIO=1: high-priority clip streams.  Measured, round 4, same call: c3 832 -> 654 frames/s - at 64 queries
                #  the tracker's kernels are chip-sized themselves and pre-empt the decoders' rounds -, configs[4] 414 -> 415: off)
                prio = -1 if os.environ.get("L4P_TRACK_PRIO", "0") == "1" else 0
                pool = [torch.cuda.Stream(device=dev, priority=prio) for _ in range(B)]
                self._clip_streams = pool
            start = getattr(self, "start_event", None) if early else None  # what the clip streams wait for: an event of the
            if start is not None:                       # launching stream (parallel.forward_windows_sharded), else everything
                pool[0].wait_event(start)               # queued on it so far
                with torch.cuda.stream(pool[0]):
                    traj_all, vis_all, dep_all = output_buffers()
                # allocated on a clip stream, handed to the launching stream's consumers after the join: the allocator must not
                # hand the blocks out again (to pool[0]) while work queued on the launching stream still reads them
                for t in (traj_all, vis_all, dep_all):
                    t.record_stream(main)
            for b in range(B):
                if start is None:
                    pool[b].wait_stream(main)
                elif b > 0:
                    pool[b].wait_stream(pool[0])
                with torch.cuda.stream(pool[b]):
                    self._ws_slot = b + 1  # concurrent clips: one native workspace each
                    run_clip(b)
            self._ws_slot = 0
            if getattr(self, "defer_join", False):
                # the caller (L4P_VideoMAE.stitch_windows) runs the dense heads on the main stream meanwhile and joins the
                # clip streams before it returns: the tracker's ~130 tiny dependent launches per clip fill the gaps of the
                # decoders' large kernels instead of serialising in front of them
                self._pending = (main, pool[:B])
            else:
                for b in range(B):
                    main.wait_stream(pool[b])
        else:
            for b in range(B):
                run_clip(b)
        return {f"{self.task_name}_traj_est_bn2t": traj_all, f"{self.task_name}_vis_est_bn1t": vis_all,
                f"{self.task_name}_depth_est_bn1t": dep_all}

    def join_streams(self) -> None:
        """Make the stream that launched the tracker wait for the clip streams (no-op when nothing is pending)."""
        pend = getattr(self, "_pending", None)
        if pend is not None:
            main, pool = pend
            for st in pool:
                main.wait_stream(st)
            self._pending = None

    def forward(self, enc_features_bpc_list, track_2d_pointquerries_bn3: torch.Tensor, track_2d_pointlabels_bn: torch.Tensor,
                track_2d_promptfeatures_bnc: Optional[torch.Tensor] = None,
                track_2d_promptfeaturelabels_bn: Optional[torch.Tensor] = None, **kwargs) -> Dict[str, torch.Tensor]:
        """Single-window forward (sparse_heads.py:497-600), reached from L4P_VideoMAE.forward_single_window
        (always_use_windowed_version=False and T == 16) and from forward_windowed_core(time_strides=None).  Unlike a window of
        the sliding tracker it attends to the raw enc_features[-1] (NO history / mask-token term), takes the caller's point
        labels as they are, starts from zero prompt features unless they are passed in, and returns the window's estimates
        for every frame (no validity masking, no -10 visibility fill).
        Returned: traj [B,N,2,T], vis [B,N,1,T], depth [B,N,1,T], <task>_prompt_features_bnc [B,N,C] and, as the reference
        (sparse_heads.py:560-569), <task>_enc_features_with_track_history_bnpc [B,N,P,C] float - the projection of every
        processed video token (11.5 MB per query at the full geometry: ``self.return_track_history = False`` skips it)."""
        if self._rt is None:
            raise RuntimeError("tracker head has no weights: call load_state_dict on the model first")
        lib = _lib.load()
        cfg = self._rt.cfg
        enc = enc_features_bpc_list.f32(-1)
        dev = enc.device
        B, N = track_2d_pointquerries_bn3.shape[:2]
        T = self.image_size[0]
        P, Cc = cfg.tokens, cfg.dim
        f32 = dict(dtype=torch.float32, device=dev)
        traj_all = torch.empty(B, N, 2, T, **f32)
        vis_all = torch.empty(B, N, 1, T, **f32)
        dep_all = torch.empty(B, N, 1, T, **f32)
        pf_all = torch.empty(B, N, Cc, **f32)
        zero_hist = torch.zeros(P, Cc, **f32)  # keys = enc_features[-1] + 0: one key set shared by all tracks
        # the reference returns the [B,N,P,C] float history from this entry unconditionally (sparse_heads.py:560-569), so the
        # default keeps the key; it costs 11.5 MB and one P x C x C projection per query (0.74 GB per clip at 64 queries): a
        # caller that only wants the trajectories sets ``head.return_track_history = False``.  The sliding-window path
        # (forward_windowed_core with time strides - what bench.py and the demo run) never materialises it.
        want_hist = bool(getattr(self, "return_track_history", True))
        hist_all = torch.empty(B, N, P, Cc, **f32) if want_hist else None
        for b in range(B):
            q = track_2d_pointquerries_bn3[b].to(**f32).contiguous()
            labels = track_2d_pointlabels_bn[b].to(**f32).contiguous()
            pfeat = (torch.zeros(N, Cc, **f32) if track_2d_promptfeatures_bnc is None
                     else track_2d_promptfeatures_bnc[b].to(**f32).contiguous())
            plabel = (torch.zeros(N, **f32) if track_2d_promptfeaturelabels_bn is None
                      else track_2d_promptfeaturelabels_bn[b].to(**f32).contiguous())
            if want_hist:
                # hist [N*P, C]: its first P rows are the (zero) history the shared keys are built from, the call then
                # overwrites all of it with the projection of the processed tokens (need_history = 3)
                hist = hist_all[b].view(N * P, Cc)
                hist[:P].zero_()
            else:
                hist = zero_hist
            w_traj, w_vis, w_dep, new_pfeat = self._window(enc[b].contiguous(), hist, q, labels, pfeat, plabel,
                                                           3 if want_hist else 0, hist_uniform=True)
            traj_all[b], vis_all[b, :, 0], dep_all[b, :, 0], pf_all[b] = w_traj, w_vis, w_dep, new_pfeat
        out = {f"{self.task_name}_traj_est_bn2t": traj_all, f"{self.task_name}_vis_est_bn1t": vis_all,
               f"{self.task_name}_depth_est_bn1t": dep_all, f"{self.task_name}_prompt_features_bnc": pf_all}
        if want_hist:
            out[f"{self.task_name}_enc_features_with_track_history_bnpc"] = hist_all
        return out
