"""L4P_VideoMAE on the MI355X engine — host mirror of l4p/models/l4p_videomae.py.

Same constructor arguments, ``forward(data, tasks)`` semantics, window slicing and output keys as the
reference (l4p_videomae.py:125-330).  The module owns no torch parameters: ``load_state_dict`` accepts
the reference checkpoint layout, repacks it into kernel layouts on the device
and binds it into libl4p_hip.so.
"""
from __future__ import annotations

import os

from collections import OrderedDict
from typing import Any, Dict, Iterable, List, Optional, Sequence, Tuple

import torch

from .. import _lib
from .._lib import L4P_BF16, L4P_F16, L4P_F32
from ..engine import Engine
from ..packing import PackedWeights, pack_state_dict
from ..weights import ModelCfg, state_dict_schema
from .task_heads.dense_heads import _Runtime, VideoMAEFlowDPTHead, VideoMAETraj3DDPTHead


class EncoderFeatures:
    """Stand-in for the reference's list of depth+1 per-layer features (l4p_videomae.py:106-122).
    Only the layers some head asked for exist; they are held as float and/or engine dtype (T)."""

    def __init__(self, depth: int, f32: Dict[int, torch.Tensor], T: Dict[int, torch.Tensor]):
        self.depth, self._f32, self._T = depth, f32, T

    def __len__(self) -> int:
        return self.depth + 1

    def _norm(self, i: int) -> int:
        return i + self.depth + 1 if i < 0 else i

    def T(self, i: int) -> torch.Tensor:
        i = self._norm(i)
        if i not in self._T:
            raise KeyError(f"encoder feature {i} was not requested in engine dtype (available: {sorted(self._T)})")
        return self._T[i]

    def f32(self, i: int) -> torch.Tensor:
        i = self._norm(i)
        if i in self._f32:
            return self._f32[i]
        if i in self._T:
            return self._T[i].float()
        raise KeyError(f"encoder feature {i} was not requested (available: {sorted(set(self._f32) | set(self._T))})")

    def __getitem__(self, i: int) -> torch.Tensor:
        return self.f32(i)


def _on_own_device(fn):
    """Run a model entry point with the model's GPU as the current device: kernels are launched on
    ``torch.cuda.current_stream()`` and allocations land on the current device, so a model pinned to cuda:3 must not depend on
    what the caller's current device happens to be (round-3 advisor finding)."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **kw):
        dev = getattr(self, "device", None)
        if dev is None or dev.type != "cuda" or self.engine is None:
            return fn(self, *a, **kw)
        with torch.cuda.device(dev):
            return fn(self, *a, **kw)

    return wrapped


def _engine_dtype(name) -> int:
    if name in (L4P_BF16, "bf16", "bf16-mixed", "bf16-true"):
        return L4P_BF16
    # Fabric's "16-mixed" / "16-true" are float16 (demo/demo.py:22-23, l4p/models/utils.py:57-58 of the reference): the half engine -
    # f16 MFMA operands and stored activations, f32 accumulate / residual stream / LayerNorm / softmax statistics
    if name in (L4P_F16, "16-mixed", "16-true", "f16", "fp16", "16"):
        return L4P_F16
    if name in (L4P_F32, "32", "32-true", "fp32", "f32"):
        return L4P_F32
    raise ValueError(f"unsupported precision {name!r}")


class L4P_VideoMAE(torch.nn.Module):
    """Main L4P model: shared video encoder + task heads (l4p_videomae.py:125-330)."""

    def __init__(
        self,
        task_heads: torch.nn.ModuleDict,
        video_encoder_ckpt_path: Optional[str] = None,
        window_size: Tuple[int, int, int] = (16, 224, 224),
        window_stride_T: int = 8,
        freeze_video_encoder: bool = False,
        freeze_heads: Optional[List[str]] = None,
        unfreeze_blocks: Optional[List[int]] = None,
        always_use_windowed_version: bool = False,
        joint_alignment: bool = False,
        cam_emb_placed_at_enc: Optional[str] = None,
        cam_emb_type: str = "add",
        model_cfg: Optional[ModelCfg] = None,
        precision: str = "bf16",
        device: Optional[Any] = None,
    ) -> None:
        super().__init__()
        if cam_emb_placed_at_enc is not None:
            raise NotImplementedError("camera embeddings in the encoder are disabled in the shipped config (configs/model.yaml)")
        if video_encoder_ckpt_path is not None:
            raise NotImplementedError("encoder-only checkpoints are a training-time convenience; load the full state_dict")
        self.cfg = model_cfg or ModelCfg.full()
        self.task_heads = task_heads
        self.window_size = tuple(window_size)
        self.window_stride_T = window_stride_T
        self.always_use_windowed_version = always_use_windowed_version
        self.joint_alignment = joint_alignment
        # Engine option (not a reference argument): windows of a long clip that go through the encoder and the dense
        # decoders TOGETHER as one batch.  1 = one window at a time, the reference's order and bit-identical to it;
        # > 1 trades bit-identity (row counts change GEMM tile / split-K choices: differences at float rounding level,
        # tests/test_sharded_windows_gpu.py) for batch-4 efficiency on long videos (demo/demo.py sets 4).
        self.window_batch = 1
        self.engine_dtype = _engine_dtype(precision)
        self.engine: Optional[Engine] = None
        self.weights: Optional[PackedWeights] = None
        # Engine option: the GPU this model lives on ("cuda:3", 3, torch.device).  None = the current device AT THE TIME THE
        # WEIGHTS ARRIVE (load_state_dict / set_weights), so a rank may build the model before torch.cuda.set_device(local_rank).
        self._device_arg = None if device is None else torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        if self._device_arg is not None and self._device_arg.type != "cuda":
            raise _lib.L4PHipError(f"the L4P engine runs on an AMD GPU only (device={device!r})")
        if self._device_arg is not None and self._device_arg.index is None:
            self._device_arg = None  # plain "cuda": the current device, resolved late like None
        self.device = self._device_arg or (torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available()
                                           else torch.device("cpu"))
        for key, head in self.task_heads.items():
            head._engine_task = key
            # the packed weight layout and the decoder geometry are keyed by the ModuleDict key (weights.actpost_of /
            # fusion_of: the ray head's tables under "camray", the dense tables otherwise): a head whose constructor
            # arguments disagree with its key would decode with the wrong strides — refuse it here, not at the first forward
            if isinstance(head, VideoMAEFlowDPTHead):
                from ..weights import actpost_of, fusion_of

                as_t = lambda x: tuple(tuple(int(v) for v in r) for r in x)  # noqa: E731
                if as_t(head.actpost_scale_factors) != as_t(actpost_of(key)) or as_t(head.fusion_scale_factors) != as_t(fusion_of(key)):
                    raise NotImplementedError(
                        f"task head '{key}': actpost/fusion scale factors {head.actpost_scale_factors} / {head.fusion_scale_factors} "
                        f"differ from the geometry the engine packs for this key ({actpost_of(key)} / {fusion_of(key)}); "
                        "register ray heads under 'camray' and dense heads under any other key, with the default factors")

    # ---- weights ---------------------------------------------------------------------------------
    def expected_keys(self) -> "OrderedDict[str, Tuple[int, ...]]":
        return state_dict_schema(self.cfg, tasks=list(self.task_heads.keys()))

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True, assign: bool = False):
        """Accepts the reference layout (keys relative to ``l4p_model.``; a leading ``l4p_model.`` is stripped),
        checks it strictly against the schema, packs it for the engine and uploads it."""
        sd = {(k[len("l4p_model."):] if k.startswith("l4p_model.") else k): v for k, v in state_dict.items()}
        exp = self.expected_keys()
        missing = [k for k in exp if k not in sd]
        unexpected = [k for k in sd if k not in exp]
        bad = [k for k in exp if k in sd and tuple(sd[k].shape) != tuple(exp[k])]
        # strict=False tolerates UNEXPECTED keys only: the engine packs every expected tensor into its arena, so a missing
        # one cannot be skipped (the reference would keep its random initialisation — never wanted for inference)
        if bad or missing or (strict and unexpected):
            raise RuntimeError(
                f"Error(s) in loading state_dict for L4P_VideoMAE: missing={missing[:5]} ({len(missing)}), "
                f"unexpected={unexpected[:5]} ({len(unexpected)}), shape mismatch={bad[:5]} ({len(bad)})")
        if not torch.cuda.is_available():
            raise _lib.L4PHipError("no AMD GPU visible: the L4P engine has no CPU path")
        # the device is resolved when the weights arrive, not at construction: a rank that builds the model before
        # torch.cuda.set_device(local_rank) must still end up on its own GPU
        self.device = self._device_arg or torch.device("cuda", torch.cuda.current_device())
        self.set_weights(pack_state_dict(sd, self.cfg, {L4P_BF16: torch.bfloat16, L4P_F16: torch.float16, L4P_F32: torch.float32}[self.engine_dtype],
                                         self.device, tasks=list(self.task_heads.keys())))
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def set_weights(self, weights: PackedWeights) -> None:
        """Attach an already packed arena (e.g. received by RCCL broadcast, parallel.py)."""
        self.weights = weights
        if weights.arena.is_cuda:
            self.device = weights.arena.device  # the engine lives where its weights are
        self.engine = Engine(self.cfg, weights, self.engine_dtype, self.device)
        rt = _Runtime(self.cfg, weights, self.engine_dtype)
        rt.engine = self.engine
        for head in self.task_heads.values():
            head._rt = rt

    # ---- encoder ---------------------------------------------------------------------------------
    def _taps(self, tasks: Sequence[str]) -> Tuple[List[int], List[int]]:
        tT, tf = set(), set()
        for t in tasks:
            head = self.task_heads[t]
            if isinstance(head, VideoMAEFlowDPTHead):
                tT.update(head.required_taps())
            else:
                tf.add(self.cfg.depth)  # tracker reads enc_features[-1]  (sparse_heads.py:521)
        return sorted(tf), sorted(tT)

    @_on_own_device
    def video_encoder(self, rgb_b3thw: torch.Tensor, taps_f32: Iterable[int] = (), taps_T: Iterable[int] = ()) -> EncoderFeatures:
        if self.engine is None:
            raise RuntimeError("weights not loaded: call load_state_dict / set_weights first")
        rgb = rgb_b3thw.to(device=self.device, dtype=torch.float32).contiguous()
        f32, T = self.engine.encoder_forward(rgb, taps_f32, taps_T)
        return EncoderFeatures(self.cfg.depth, f32, T)

    def encode_features(self, data: Dict[str, Any], tasks: Optional[Sequence[str]] = None) -> EncoderFeatures:
        tf, tT = self._taps(tasks if tasks is not None else list(self.task_heads.keys()))
        return self.video_encoder(data["rgb_b3thw"], tf, tT)

    @_on_own_device
    def forward_single_window(self, data: Dict[str, Any], tasks: List[str]) -> Dict[str, Any]:
        feats = self.encode_features(data, tasks)
        out: Dict[str, Any] = {"enc_features_bpc_list": feats}
        for task in tasks:
            out.update(self.task_heads[task](enc_features_bpc_list=feats, **data))
        return out

    # ---- main entry ------------------------------------------------------------------------------
    @_on_own_device
    def forward(self, data: Dict[str, Any], tasks: List[str]) -> Dict[str, Any]:
        rgb = data["rgb_b3thw"]
        B, _, T, H, W = rgb.shape
        assert H == self.window_size[1] and W == self.window_size[2], "Supports only fixed spatial size"
        single = (not self.always_use_windowed_version) and (T == self.window_size[0])
        # (argument checks come before any device work, in the reference's order: l4p_videomae.py:260,267-269)
        assert single or T % self.window_stride_T == 0, "Temporal window needs to be a multiple of window stride, for now!"
        data = {k: (v.to(self.device) if torch.is_tensor(v) else v) for k, v in data.items()}
        if single:
            return self.forward_single_window(data, tasks)
        if self.window_batch > 1:
            from ..parallel import forward_windows_sharded  # the same code path the multi-GPU split uses, on one rank

            return forward_windows_sharded(self, data, tasks, 0, 1, group=int(self.window_batch))
        time_strides = self.time_strides(T)
        tf, tT = self._taps(tasks)
        ws = self.window_size[0]
        feats2d = [self.video_encoder(data["rgb_b3thw"][:, :, int(s):int(s) + ws], tf, tT) for s in time_strides]
        return self.stitch_windows(feats2d, data, tasks, time_strides)

    def time_strides(self, T: int) -> torch.Tensor:
        return torch.arange(0, T - self.window_size[0] + 1, self.window_stride_T)

    @_on_own_device
    def stitch_windows(self, feats2d: list, data: Dict[str, Any], tasks: List[str], time_strides: torch.Tensor) -> Dict[str, Any]:
        """Everything after the per-window encoder (l4p_videomae.py:296-329): per-window heads + stitching / alignment /
        track recursion.  ``feats2d`` holds one entry per window: EncoderFeatures, or parallel.DecodedWindow for windows
        whose heavy work already ran (possibly on another GPU)."""
        out: Dict[str, Any] = {"enc_features_bpc_2dlist": feats2d}
        joint_possible = "depth" in tasks and "camray" in tasks
        trk = self.task_heads["track_2d"] if "track_2d" in tasks else None
        if trk is not None and hasattr(trk, "join_streams"):
            # its per-clip streams run beside the dense heads and are joined below (L4P_TRACK_DEFER=0: join at once, A/B aid)
            trk.defer_join = len(tasks) > 1 and os.environ.get("L4P_TRACK_DEFER", "1") != "0"
        try:
            return self._stitch_windows(out, feats2d, data, tasks, time_strides, joint_possible)
        finally:
            self._join_head_streams()
            if trk is not None and hasattr(trk, "join_streams"):
                trk.join_streams()
                trk.defer_join = False

    def _heads_on_streams(self, feats2d) -> bool:
        """dyn_mask / flow decoders beside the depth / camray ones (L4P_HEAD_STREAMS=0: one after the other, A/B and test aid).
        Only where the windows still have to be decoded (EncoderFeatures): windows decoded elsewhere (parallel.DecodedWindow)
        leave nothing but copies.  Measured, round 5, same call x 3: c3 953.8 -> 963.0 frames/s at batch 4, 992.7 -> 994.5 at
        batch 8; the depth decoder beside the camray one as well: no further gain (961.2), not built."""
        return (self.device.type == "cuda" and os.environ.get("L4P_HEAD_STREAMS", "1") != "0"
                and all(isinstance(f, EncoderFeatures) for f in feats2d))

    def _run_heads_on_streams(self, names, out, feats2d, data, time_strides) -> None:
        main = torch.cuda.current_stream()
        pool = getattr(self, "_head_streams", None)
        if pool is None or len(pool) < len(names):
            pool = [torch.cuda.Stream(device=self.device) for _ in names]
            self._head_streams = pool
        ready = torch.cuda.Event()
        ready.record(main)
        # recorded BEFORE anything is queued: if a head (or a native call inside it) raises with an earlier side stream already at
        # work, the join in forward()'s finally block still makes the main stream wait for every stream that may have been touched -
        # the next forward reuses the per-task workspaces and the encoder feature blocks
        self._pending_heads = (main, pool[:len(names)])
        for st, task in zip(pool, names):
            st.wait_event(ready)  # (the encoder features and the batch exist; nothing the main stream queues later is waited for)
            with torch.cuda.stream(st):
                o = self.task_heads[task].forward_windowed(enc_features_bpc_2dlist=feats2d, time_strides=time_strides, **data)
            for v in o.values():
                if torch.is_tensor(v):
                    v.record_stream(main)  # allocated on the side stream, consumed by the caller on the main one after the join
            out.update(o)

    def _join_head_streams(self) -> None:
        pend = getattr(self, "_pending_heads", None)
        if pend is not None:
            main, pool = pend
            for st in pool:
                main.wait_stream(st)
            self._pending_heads = None

    def _stitch_windows(self, out, feats2d, data, tasks, time_strides, joint_possible):
        if self.joint_alignment and joint_possible:
            from .task_heads.dense_heads import joint_windowed_estimation

            if "track_2d" in tasks:
                out.update(self.task_heads["track_2d"].forward_windowed(
                    enc_features_bpc_2dlist=feats2d, time_strides=time_strides, **data))
            side = [t for t in ("dyn_mask", "flow_2d_backward") if t in tasks]
            if len(side) > 0 and self._heads_on_streams(feats2d):
                # the two decoders that take no part in the joint alignment run on streams of their own BESIDE the depth / camray
                # decoders below (joined in stitch_windows): the low-resolution levels of a DPT decoder are launches of 30 - 130
                # workgroups, and four decoders in a row leave the chip to them one at a time
                self._run_heads_on_streams(side, out, feats2d, data, time_strides)
            else:
                for task in side:
                    out.update(self.task_heads[task].forward_windowed(
                        enc_features_bpc_2dlist=feats2d, time_strides=time_strides, **data))
            out.update(joint_windowed_estimation(["depth", "camray"], self.task_heads, enc_features_bpc_2dlist=feats2d,
                                                 time_strides=time_strides, **data))
        else:
            if self.joint_alignment:
                print("Joint alignment is not possible as depth or camray tasks are not present")
            for task in tasks:
                out.update(self.task_heads[task].forward_windowed(
                    enc_features_bpc_2dlist=feats2d, time_strides=time_strides, **data))
        return out
