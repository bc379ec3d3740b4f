"""Python handle on the C-ABI engine (include/l4p_hip.h): owns the packed-weight arena, binds it into
libl4p_hip.so and runs whole sub-networks with one native call."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import L4P_BF16, L4P_F32, EncoderCfg
from .ops import code_of, torch_dtype
from .packing import PackedWeights
from .weights import ModelCfg


def _on_device(fn):
    """The engine's GPU is the current device (and its current stream the launch stream) for the duration of a native call."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **kw):
        with torch.cuda.device(self.device):
            return fn(self, *a, **kw)

    return wrapped


class Engine:
    """One engine per (process, device, dtype)."""

    def __init__(self, cfg: ModelCfg, weights: PackedWeights, dtype: int, device: torch.device):
        if device.type != "cuda":
            raise _lib.L4PHipError("the L4P engine runs on an AMD GPU only (device must be cuda:N); no CPU path exists")
        self.lib = _lib.load()
        self.cfg, self.weights, self.dtype, self.device = cfg, weights, dtype, device
        self.tdtype = torch_dtype(dtype)
        h = C.c_void_p()
        _lib.check(self.lib.l4p_create(device.index or 0, dtype, C.byref(h)), "l4p_create")
        self.handle = h
        for name, t in weights.t.items():
            if name.startswith("enc.") or name.startswith("dpt.") or name.startswith("trk."):
                _lib.check(self.lib.l4p_bind_weight(self.handle, name.encode(), t.data_ptr(), t.numel()), "l4p_bind_weight")
        self._dpt_ws: Dict[Tuple[str, int], torch.Tensor] = {}
        self._trk_ws: Dict[int, torch.Tensor] = {}
        self._trk_need: Dict[Tuple[int, int], int] = {}
        ec = EncoderCfg(dim=cfg.dim, depth=cfg.depth, heads=cfg.heads, head_dim=cfg.head_dim, mlp_hidden=cfg.mlp_hidden,
                        in_chans=cfg.in_chans, frames=cfg.frames, img_h=cfg.img, img_w=cfg.img, pt=cfg.patch[0],
                        ph=cfg.patch[1], pw=cfg.patch[2], patch_kp=int(weights.meta["patch_kp"]), ln_eps=cfg.ln_eps)
        _lib.check(self.lib.l4p_encoder_configure(self.handle, C.byref(ec)), "l4p_encoder_configure")
        self._ws: Optional[torch.Tensor] = None

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.l4p_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def _workspace(self, B: int) -> torch.Tensor:
        need = int(self.lib.l4p_encoder_workspace_bytes(self.handle, B))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    @_on_device
    def dpt_forward(self, task: str, hooks: Sequence[torch.Tensor], out_ch: int, actpost, fusion,
                    out_size: Tuple[int, int, int], post_exp: bool) -> torch.Tensor:
        """DPTOutputAdapter_fix.forward (dpt_head.py:41-86) of head ``task`` as one native call.
        hooks: 4 x [B, P, C] engine dtype; returns float [B, out_ch, T, H, W]."""
        c = self.cfg
        B = hooks[0].shape[0]
        dc = _lib.DptCfg()
        dc.dim = c.dim
        dc.nt, dc.nh, dc.nw = c.grid
        for i in range(4):
            dc.layer_dims[i] = c.layer_dims[i]
            for j in range(3):
                dc.actpost[i][j] = actpost[i][j]
                dc.fusion[i][j] = fusion[i][j]
        dc.feature_dim, dc.last_dim, dc.out_ch = c.feature_dim, c.last_dim, out_ch
        dc.out_t, dc.out_h, dc.out_w = out_size
        dc.post_exp = 1 if post_exp else 0
        key = (task, B)
        ws = self._dpt_ws.get(key)
        if ws is None:
            need = int(self.lib.l4p_dpt_workspace_bytes(self.handle, C.byref(dc), B))
            ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._dpt_ws[key] = ws
        hs = [h.contiguous() for h in hooks]
        hp = (C.c_void_p * 4)(*[h.data_ptr() for h in hs])
        out = torch.empty((B, out_ch) + tuple(out_size), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.l4p_dpt_forward(self.handle, torch.cuda.current_stream().cuda_stream, task.encode(), C.byref(dc),
                                            hp, B, ws.data_ptr(), ws.numel(), out.data_ptr()), "l4p_dpt_forward")
        return out

    @_on_device
    def encoder_forward(self, rgb: torch.Tensor, taps_f32: Iterable[int] = (), taps_T: Iterable[int] = ()) -> Tuple[Dict[int, torch.Tensor], Dict[int, torch.Tensor]]:
        """VideoMAEEncoder.forward (l4p_videomae.py:80-122) for the requested feature indices only.
        Returns ({layer: float [B,P,C]}, {layer: T [B,P,C]})."""
        assert rgb.is_cuda and rgb.dtype == torch.float32 and rgb.is_contiguous()
        B = rgb.shape[0]
        c = self.cfg
        assert tuple(rgb.shape[1:]) == (c.in_chans, c.frames, c.img, c.img), f"bad clip shape {tuple(rgb.shape)}"
        tf, tT = sorted(set(taps_f32)), sorted(set(taps_T))
        layers = sorted(set(tf) | set(tT))
        n = len(layers)
        out_f = {l: torch.empty(B, c.tokens, c.dim, dtype=torch.float32, device=self.device) for l in tf}
        out_T = {l: torch.empty(B, c.tokens, c.dim, dtype=self.tdtype, device=self.device) for l in tT}
        lay = (C.c_int * n)(*layers)
        pf = (C.c_void_p * n)(*[out_f[l].data_ptr() if l in out_f else None for l in layers])
        pT = (C.c_void_p * n)(*[out_T[l].data_ptr() if l in out_T else None for l in layers])
        ws = self._workspace(B)
        _lib.check(self.lib.l4p_encoder_forward(self.handle, torch.cuda.current_stream().cuda_stream, rgb.data_ptr(), B,
                                                ws.data_ptr(), ws.numel(), n, lay, pf, pT), "l4p_encoder_forward")
        return out_f, out_T

    @_on_device
    def track_window(self, tcfg: "_lib.TrackCfg", enc_last: torch.Tensor, hist: torch.Tensor, q_off: torch.Tensor,
                     labels: torch.Tensor, pfeat: torch.Tensor, plabel: torch.Tensor, need_history: bool, hist_uniform: int,
                     slot: int = 0):
        """One tracker window of one clip as ONE native call (l4p_track_window_forward).  ``slot`` selects the workspace:
        clips whose trackers run concurrently on different HIP streams must not share one."""
        N, Cc = q_off.shape[0], self.cfg.dim
        T = tcfg.T
        hu = int(hist_uniform)  # 0 per-track history, 1 uniform (first window), 2 second temporal half uniform (later windows), 4 later window, per-track
        # ONE workspace per slot, sized for the largest of the three history forms at the largest N seen (a clip's first
        # window runs hu = 1, its later ones hu = 2, a last chunk has fewer queries: keyed by shape, every forward freed and
        # re-allocated several hundred MB per clip); it only ever grows
        need = self._trk_need.get((N, hu))
        if need is None:
            for h in (0, 1, 2, 4):
                nb = int(self.lib.l4p_track_window_workspace_bytes(self.handle, C.byref(tcfg), N, h))
                if nb == 0:
                    raise _lib.L4PHipError("l4p_track_window_workspace_bytes: " + self.lib.l4p_last_error().decode())
                self._trk_need[(N, h)] = nb
            need = max(self._trk_need[(N, h)] for h in (0, 1, 2, 4))
            for h in (0, 1, 2, 4):
                self._trk_need[(N, h)] = need
        ws = self._trk_ws.get(slot)
        if ws is None or ws.numel() < need:
            ws = None
            self._trk_ws.pop(slot, None)
            ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._trk_ws[slot] = ws
        f32 = dict(dtype=torch.float32, device=self.device)
        traj = torch.empty((N, 2, T), **f32)
        vis = torch.empty((N, T), **f32)
        dep = torch.empty((N, T), **f32)
        new_pfeat = torch.empty((N, Cc), **f32)
        _lib.check(self.lib.l4p_track_window_forward(
            self.handle, torch.cuda.current_stream().cuda_stream, C.byref(tcfg), enc_last.data_ptr(), hist.data_ptr(),
            q_off.data_ptr(), labels.data_ptr(), pfeat.data_ptr(), plabel.data_ptr(), N, int(need_history),
            hu, ws.data_ptr(), ws.numel(), traj.data_ptr(), vis.data_ptr(), dep.data_ptr(),
            new_pfeat.data_ptr()), "l4p_track_window_forward")
        return traj, vis, dep, new_pfeat
