"""ctypes binding of libl4p_hip.so (C ABI declared in include/l4p_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C l4p_amd/csrc`` into
``l4p_amd/lib/``.  There is no fallback: if the shared object is missing or a symbol cannot be
resolved, importing the engine raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

L4P_BF16 = 0
L4P_F32 = 1
L4P_F16 = 2  # IEEE half storage / f16 MFMA: the arithmetic class of the reference's "16-mixed" (fp16 autocast)

EPI_DENSE, EPI_QKV, EPI_CONVT, EPI_MASKDOT = 0, 1, 2, 3
ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2

_HERE = os.path.dirname(os.path.abspath(__file__))
# L4P_HIP_LIB: another build of the same library (tuning aid: A/B of two builds inside one GPU call)
LIB_PATH = os.environ.get("L4P_HIP_LIB") or os.path.join(_HERE, "lib", "libl4p_hip.so")


class GemmDesc(C.Structure):
    """Mirror of ``l4p_gemm_desc`` (include/l4p_hip.h) — field order and types must match."""

    _fields_ = [
        ("A", C.c_void_p), ("lda", C.c_longlong),
        ("W", C.c_void_p), ("ldw", C.c_longlong),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("Ti", C.c_int), ("Hi", C.c_int), ("Wi", C.c_int), ("Cin", C.c_int),
        ("To", C.c_int), ("Ho", C.c_int), ("Wo", C.c_int),
        ("st", C.c_int), ("sh", C.c_int), ("sw", C.c_int), ("relu_in", C.c_int),
        ("bias", C.c_void_p), ("act", C.c_int),
        ("res1", C.c_void_p), ("res2", C.c_void_p), ("res_f32", C.c_int),
        ("ldr", C.c_longlong), ("res_mod", C.c_int),
        ("out_f32", C.c_void_p), ("out_T", C.c_void_p), ("ldc", C.c_longlong),
        ("epi", C.c_int),
        ("vt", C.c_void_p), ("S", C.c_int), ("H", C.c_int), ("Dp", C.c_int),
        ("kt", C.c_int), ("kh", C.c_int), ("kw", C.c_int), ("Cout", C.c_int),
        ("a_gr", C.c_int), ("a_gs", C.c_int), ("a_go", C.c_int),
        ("c_gr", C.c_int), ("c_gs", C.c_int), ("c_go", C.c_int),
        ("out_relu_T", C.c_void_p),
        ("k_tiled", C.c_void_p),
        ("splitk", C.c_int), ("partial", C.c_void_p),
        ("hyper", C.c_void_p), ("hyper_rows", C.c_int),
        ("q_scale", C.c_float), ("tuning", C.c_int),
        ("w_gr", C.c_int), ("w_gs", C.c_longlong), ("b_gs", C.c_int), ("o_gs", C.c_longlong),
        ("ups_hi", C.c_int), ("ups_wi", C.c_int),
        ("kw_cols", C.c_int), ("kw_len", C.c_int),
    ]


class EncoderCfg(C.Structure):
    """Mirror of ``l4p_encoder_cfg``."""

    _fields_ = [
        ("dim", C.c_int), ("depth", C.c_int), ("heads", C.c_int), ("head_dim", C.c_int), ("mlp_hidden", C.c_int),
        ("in_chans", C.c_int), ("frames", C.c_int), ("img_h", C.c_int), ("img_w", C.c_int),
        ("pt", C.c_int), ("ph", C.c_int), ("pw", C.c_int),
        ("patch_kp", C.c_int), ("ln_eps", C.c_float),
    ]


class DptCfg(C.Structure):
    """Mirror of ``l4p_dpt_cfg``."""

    _fields_ = [
        ("dim", C.c_int), ("nt", C.c_int), ("nh", C.c_int), ("nw", C.c_int),
        ("layer_dims", C.c_int * 4),
        ("feature_dim", C.c_int), ("last_dim", C.c_int), ("out_ch", C.c_int),
        ("actpost", (C.c_int * 3) * 4), ("fusion", (C.c_int * 3) * 4),
        ("out_t", C.c_int), ("out_h", C.c_int), ("out_w", C.c_int),
        ("post_exp", C.c_int),
    ]


class TrackCfg(C.Structure):
    """Mirror of ``l4p_track_cfg``."""

    _fields_ = [(n, C.c_int) for n in ("dim", "tokens", "nt", "nh", "nw", "sam_depth", "sam_heads", "sam_mlp", "out_dim_factor",
                                        "T", "H", "W")]


# name -> (restype, argtypes); every symbol include/l4p_hip.h declares must appear here
_VP, _I, _LL, _F, _SZ = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_size_t
SIGNATURES = {
    "l4p_last_error": (C.c_char_p, []),
    "l4p_abi_version": (_I, []),
    "l4p_stream_create_cu_mask": (_I, [_I, _I, C.POINTER(C.c_void_p)]),
    "l4p_stream_destroy": (_I, [_VP]),
    "l4p_set_knob": (_I, [C.c_char_p, _I]),
    "l4p_get_knob": (_I, [C.c_char_p]),
    "l4p_prof_enable": (_I, [_I]),
    "l4p_prof_reset": (_I, []),
    "l4p_prof_num_classes": (_I, []),
    "l4p_prof_class_name": (C.c_char_p, [_I]),
    "l4p_prof_read": (_I, [_I, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
    "l4p_prof_detail": (_LL, [C.c_char_p, _LL]),
    "l4p_gemm": (_I, [_VP, _I, C.POINTER(GemmDesc)]),
    "l4p_conv3d_k3": (_I, [_VP, _I, C.POINTER(GemmDesc)]),
    "l4p_layernorm": (_I, [_VP, _I, _VP, _VP, _VP, _F, _VP, _VP, _I, _I]),
    "l4p_attention": (_I, [_VP, _I, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _F]),
    "l4p_patch_gather": (_I, [_VP, _I, _VP, _VP, _I, _I, _I, _I, _I, _I, _I, _I, _I]),
    "l4p_cast": (_I, [_VP, _I, _VP, _VP, _LL]),
    "l4p_upsample_trilinear": (_I, [_VP, _I, _VP, _VP, _I, _I, _I, _I, _I, _I, _I, _I, _I]),
    "l4p_head_out": (_I, [_VP, _I, _VP, _VP, _VP, _VP, _LL, _I, _I, _I, _I]),
    "l4p_affine_align_solve": (_I, [_VP, _VP, _VP, _LL, _I, _VP, _VP]),
    "l4p_affine_align_apply": (_I, [_VP, _VP, _VP, _LL, _I, _VP]),
    "l4p_rays_to_pose": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _I, _I, _I]),
    "l4p_rays_to_intrinsics": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _I, _I, _I, _I, _F]),
    "l4p_rays_to_intrinsics_frames": (_I, [_VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _I, _I, _F]),
    "l4p_rays_to_pose_rot": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _I]),
    "l4p_quantile": (_I, [_VP, _VP, _LL, _F, _VP, _VP]),
    "l4p_select_rank": (_I, [_VP, _VP, _LL, _LL, _VP, _VP]),
    "l4p_ratio_median_solve": (_I, [_VP, _VP, _VP, _LL, _I, _VP, _VP, _VP]),
    "l4p_point_map_samples": (_I, [_VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, C.c_uint]),
    "l4p_similarity_ransac": (_I, [_VP, _VP, _VP, _I, _VP, _F, _I, _I, C.c_uint, _VP, _VP]),
    "l4p_similarity_apply": (_I, [_VP, _VP, _VP, _I, _VP, _LL]),
    "l4p_similarity_prefix": (_I, [_VP, _VP, _VP, _I, _I]),
    "l4p_layernorm_ex": (_I, [_VP, _I, _VP, _VP, _VP, _F, _VP, _VP, _I, _I, _VP, _I, _VP, _I]),
    "l4p_gemm_group": (_I, [_VP, _I, _VP, _I]),
    "l4p_layernorm_res": (_I, [_VP, _I, _VP, _I, _VP, _VP, _VP, _F, _VP, _VP, _I, _I, _VP, _I, _VP, _VP, _I, _I, _VP]),
    "l4p_layernorm_chain": (_I, [_VP, _I, _VP, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _F, _VP, _VP, _I, _I, _VP, _I, _VP]),
    "l4p_track_tokens": (_I, [_VP] * 13 + [_I] * 5),
    "l4p_track_keys_init": (_I, [_VP, _I, _VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP]),
    "l4p_fill_rows": (_I, [_VP, _VP, _VP, _LL, _I, _LL, _LL, _LL]),
    "l4p_broadcast_block": (_I, [_VP, _VP, _LL, _LL, _LL, _I]),
    "l4p_small_attn": (_I, [_VP, _I, _I, _VP, _VP, _VP, _VP, _I, _I, _I, _I]),
    "l4p_mask_product": (_I, [_VP, _I, _VP, _VP, _VP, _I, _LL, _I]),
    "l4p_mask_gather": (_I, [_VP, _VP, _VP, _I, _I, _I, _I, _I]),
    "l4p_i2t_probs": (_I, [_VP, _I, _VP, _LL, _I, _VP, _I, _VP, _I, _LL, _I, _I]),
    "l4p_split_hilo": (_I, [_VP, _I, _VP, _VP, _I, _I, _LL]),
    "l4p_t2i_attn_scores": (_I, [_VP, _I, _VP, _LL, _VP, _VP, _I, _I, _I, _I]),
    "l4p_transpose_pad": (_I, [_VP, _I, _VP, _VP, _I, _I, _I, _I]),
    "l4p_i2t_delta": (_I, [_VP, _I, _VP, _VP, _VP, _VP, _I, _I, _I, _I]),
    "l4p_t2i_probs": (_I, [_VP, _I, _VP, _LL, _VP, _VP, _I, _I, _I]),
    "l4p_t2i_context": (_I, [_VP, _I, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _I, _LL, _I]),
    "l4p_layernorm_t": (_I, [_VP, _I, _VP, _VP, _VP, C.c_float, _VP, _I, _I, _I]),
    "l4p_pil_coeffs": (_I, [_I, _I, _VP, _VP, _I, C.POINTER(_I)]),
    "l4p_pil_resample_u8": (_I, [_VP, _VP, _VP, _LL, _I, _I, _I, _I, _I, _VP, _VP, _I]),
    "l4p_clip_resize_normalize": (_I, [_VP, _VP, _VP, _VP] + [_I] * 9 + [_VP, _VP, _I, _VP, _VP, _I, _VP]),
    "l4p_resize_index_table": (_I, [_I, _I, _I, _I, _VP, _VP, _VP]),
    "l4p_track_readout": (_I, [_VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _I, _I]),
    "l4p_track_prepare": (_I, [_VP, _VP, _VP, _I, _I, _VP, _VP, _VP, _VP, _I]),
    "l4p_track_commit": (_I, [_VP] * 9 + [_I] * 5 + [_VP] * 5 + [_I, _I]),
    "l4p_create": (_I, [_I, _I, C.POINTER(_VP)]),
    "l4p_destroy": (_I, [_VP]),
    "l4p_bind_weight": (_I, [_VP, C.c_char_p, _VP, _LL]),
    "l4p_encoder_configure": (_I, [_VP, C.POINTER(EncoderCfg)]),
    "l4p_encoder_workspace_bytes": (_SZ, [_VP, _I]),
    "l4p_encoder_forward": (_I, [_VP, _VP, _VP, _I, _VP, _SZ, _I, C.POINTER(_I), C.POINTER(_VP), C.POINTER(_VP)]),
    "l4p_track_window_workspace_bytes": (_SZ, [_VP, C.POINTER(TrackCfg), _I, _I]),
    "l4p_track_window_forward": (_I, [_VP, _VP, C.POINTER(TrackCfg)] + [_VP] * 6 + [_I, _I, _I, _VP, _SZ] + [_VP] * 4),
    "l4p_dpt_workspace_bytes": (_SZ, [_VP, C.POINTER(DptCfg), _I]),
    "l4p_dpt_forward": (_I, [_VP, _VP, C.c_char_p, C.POINTER(DptCfg), C.POINTER(_VP), _I, _VP, _SZ, _VP]),
}

_lib: Optional[C.CDLL] = None


class L4PHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libl4p_hip.so and bind every declared symbol.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise L4PHipError(
            f"{LIB_PATH} not found — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C l4p_amd/csrc`. There is no CPU/eager fallback."
        )
    # PyTorch-ROCm bundles its own libamdhip64 and must be the FIRST to load one: the library then binds to that runtime
    # by soname.  Loaded the other way round (our /opt/rocm copy first, torch's afterwards) the process holds two HIP
    # runtimes and hipSetDevice in ours reports "no ROCm-capable device" (measured: build() followed by smoke()).
    import torch  # noqa: F401

    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def kernel_tree_hash() -> str:
    """sha1 over the kernel sources and the C header (file names + contents, sorted): ties a measurement artefact (the PMC traffic
    file under profiles/) to the kernels it was taken on - bench.py attaches `roofline.traffic` only when the hashes agree."""
    import hashlib

    root = os.path.dirname(os.path.abspath(__file__))
    files = sorted(os.path.join(root, "csrc", f) for f in os.listdir(os.path.join(root, "csrc"))
                   if f.endswith((".hip", ".hpp", ".inc")))
    files.append(os.path.join(os.path.dirname(root), "include", "l4p_hip.h"))
    h = hashlib.sha1()
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def set_knob(name: str, value: int) -> None:
    """Dispatch knob of the native launchers (include/l4p_hip.h: "conv_halo", "gemm_4w")."""
    check(load().l4p_set_knob(name.encode(), int(value)), f"l4p_set_knob({name})")


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().l4p_last_error().decode("utf-8", "replace")
        raise L4PHipError(f"{what or 'l4p_hip call'} failed (rc={rc}): {msg}")
