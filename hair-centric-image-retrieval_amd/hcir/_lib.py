"""ctypes loader for libhcir.so (the C ABI declared in include/hcir.h).

The product path has no CPU fallback: if the library is missing, or an op is handed
a tensor that is not on a HIP device, it raises.  `build()` compiles the library
in-tree with hipcc (make -C csrc); it does not need a GPU.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

# torch bundles its own HIP runtime (torch/lib/libamdhip64.so).  It MUST be loaded before
# libhcir.so so that the library's libamdhip64 dependency resolves to that same copy: two HIP
# runtimes in one process cannot share streams or device pointers.
import torch  # noqa: F401  (side effect: loads torch's libamdhip64)

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC_DIR = os.path.join(_PKG_DIR, "csrc")
# HCIR_LIB_PATH: load another build of the same sources (A/B and ablation builds under tools/); default in-tree
LIB_PATH = os.environ.get("HCIR_LIB_PATH") or os.path.join(CSRC_DIR, "libhcir.so")

F32, F16, BF16 = 0, 1, 2
EPI_BIAS_F16, EPI_BIAS_GELU_F16, EPI_BIAS_RESID_F32, EPI_BIAS_F32, EPI_AFFINE_RELU_F16, EPI_AFFINE_F32, \
    EPI_BIAS_RESID_F16 = range(7)

_lib = None

c_i32, c_i64, c_f32, c_vp, c_sz, c_int = (
    ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int)

# name -> (restype, argtypes); must list every symbol of include/hcir.h
SIGNATURES = {
    "hcir_version": (c_int, []),
    "hcir_status_string": (ctypes.c_char_p, [c_int]),
    "hcir_build_id": (ctypes.c_char_p, []),
    "hcir_row_invnorm": (c_int, [c_vp, c_i64, c_i32, c_i64, c_int, c_f32, c_vp, c_vp]),
    "hcir_l2_normalize": (c_int, [c_vp, c_i64, c_i32, c_f32, c_vp, c_vp, c_vp]),
    "hcir_sim_topk_workspace_bytes": (c_sz, [c_i64, c_i64, c_i32, c_i32, c_int]),
    "hcir_sim_topk": (c_int, [c_vp, c_i64, c_vp, c_i64, c_i32, c_i32, c_int, c_vp, c_vp, c_i64,
                              c_vp, c_vp, c_vp, c_sz, c_vp]),
    "hcir_topk_refine_f32": (c_int, [c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_vp, c_i32, c_i32, c_i64, c_vp, c_vp,
                                     c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "hcir_topk_merge": (c_int, [c_vp, c_vp, c_i32, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "hcir_ntxent_workspace_bytes": (c_sz, [c_i64, c_i32, c_int]),
    "hcir_ntxent_fwd": (c_int, [c_vp, c_vp, c_i64, c_i32, c_int, c_f32, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "hcir_ntxent_bwd_workspace_bytes": (c_sz, [c_i64, c_i32, c_int]),
    "hcir_ntxent_bwd": (c_int, [c_vp, c_vp, c_i64, c_i32, c_int, c_f32, c_vp, c_f32, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "hcir_layernorm_f16": (c_int, [c_vp, c_int, c_i64, c_i32, c_i64, c_vp, c_vp, c_f32, c_vp, c_i64, c_vp]),
    "hcir_gemm_f16": (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_i32, c_i32, c_int,
                              c_vp, c_i64, c_vp]),
    "hcir_gemm_f16_resid": (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_i32, c_i32, c_int, c_vp,
                                    c_vp, c_i64, c_vp]),
    "hcir_gemm_f16_gelu_dual": (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_i64, c_vp]),
    "hcir_gemm_fused_supported": (c_int, [c_i64, c_i32, c_i32]),
    "hcir_gemm_stats_slices": (c_i32, [c_i32]),
    "hcir_gemm_f16_fused": (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_i32, c_i32, c_int,
                                    c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "hcir_ln_stats_finalize": (c_int, [c_vp, c_i32, c_i64, c_i32, c_f32, c_vp, c_vp]),
    "hcir_patch_embed": (c_int, [c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp, c_vp, c_vp,
                                 c_f32, c_i32, c_vp, c_int, c_vp]),
    "hcir_attn_fwd": (c_int, [c_vp, c_i64, c_i32, c_i32, c_i32, c_f32, c_i32, c_vp, c_vp]),
    "hcir_cls_head": (c_int, [c_vp, c_int, c_i64, c_i32, c_i32, c_vp, c_vp, c_f32, c_int, c_vp, c_vp, c_vp]),
    "hcir_patch_mean": (c_int, [c_vp, c_int, c_i64, c_i32, c_i32, c_vp, c_vp, c_f32, c_vp, c_vp]),
    "hcir_knn_transform_u8": (c_int, [c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "hcir_ema_update": (c_int, [c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_vp]),
    "hcir_positive_masking": (c_int, [c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp, c_vp, c_vp, c_vp,
                                      c_vp]),
    "hcir_convert_f32": (c_int, [c_vp, c_i64, c_int, c_vp, c_vp]),
    "hcir_gelu_fwd_f16": (c_int, [c_vp, c_i64, c_vp, c_vp]),
    "hcir_gelu_bwd_f16": (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp]),
    "hcir_add_f32_f16": (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp]),
    "hcir_layernorm_bwd_blocks": (c_i32, [c_i64]),
    "hcir_layernorm_bwd": (c_int, [c_vp, c_int, c_i64, c_i32, c_i64, c_vp, c_i64, c_vp, c_f32, c_vp, c_vp, c_i64,
                                   c_vp, c_vp, c_int, c_vp, c_sz, c_vp]),
    "hcir_layernorm_bwd_fused": (c_int, [c_vp, c_int, c_i64, c_i32, c_i64, c_vp, c_i64, c_vp, c_f32, c_vp, c_vp,
                                         c_i64, c_vp, c_vp, c_int, c_vp, c_i64, c_vp, c_vp, c_sz, c_vp]),
    "hcir_gelu_bwd_colsum_f16": (c_int, [c_vp, c_vp, c_i64, c_i32, c_i64, c_vp, c_vp, c_int, c_vp, c_sz, c_vp]),
    "hcir_colsum_chunks": (c_i32, [c_i64]),
    "hcir_colsum_f16": (c_int, [c_vp, c_i64, c_i32, c_i64, c_vp, c_int, c_vp, c_sz, c_vp]),
    "hcir_gemm_f16_tn_workspace_bytes": (c_sz, [c_i64, c_i32, c_i32]),
    "hcir_gemm_f16_tn": (c_int, [c_vp, c_i64, c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_i64, c_int, c_vp, c_sz,
                                 c_vp]),
    "hcir_attn_fwd_lse": (c_int, [c_vp, c_i64, c_i32, c_i32, c_i32, c_f32, c_vp, c_vp, c_vp]),
    "hcir_attn_bwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_f32, c_vp, c_vp]),
    "hcir_attn_cls_fwd_lse": (c_int, [c_vp, c_i64, c_i32, c_i32, c_i32, c_f32, c_vp, c_vp, c_vp]),
    "hcir_attn_cls_bwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_f32, c_vp, c_vp]),
    "hcir_triplet_margin_fwd": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i32, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp]),
    "hcir_triplet_margin_bwd": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                        c_vp]),
    "hcir_mse_fwd": (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "hcir_mse_bwd": (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "hcir_knn_vote": (c_int, [c_vp, c_i64, c_i32, c_i64, c_vp, c_i64, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp]),
    "hcir_confusion_matrix": (c_int, [c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp]),
    "hcir_retrieval_metrics": (c_int, [c_vp, c_i64, c_i32, c_vp, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp,
                                       c_vp]),
    "hcir_positive_transform": (c_int, [c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "hcir_bn1d_fwd": (c_int, [c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_f32, c_f32, c_int, c_vp, c_vp, c_vp, c_vp,
                              c_vp, c_i64, c_vp, c_i64, c_vp]),
    "hcir_bn1d_bwd": (c_int, [c_vp, c_i64, c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp,
                              c_i64, c_vp, c_vp, c_vp]),
    "hcir_jpeg_stage_bytes": (c_sz, [c_vp, c_sz]),
    "hcir_jpeg_stage": (c_int, [c_vp, c_sz, c_vp, c_vp, c_sz, c_sz, c_vp]),
    "hcir_jpeg_stage_batch": (c_int, [c_vp, c_vp, c_i64, c_vp, c_sz, c_vp, c_vp, c_i32]),
    "hcir_jpeg_workspace_bytes": (c_sz, [c_vp, c_i64, c_i32, c_i32]),
    "hcir_jpeg_decode_window_u8": (c_int, [c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "hcir_png_stage_bytes": (c_sz, [c_vp, c_sz]),
    "hcir_png_stage": (c_int, [c_vp, c_sz, c_i32, c_vp, c_vp, c_sz, c_sz, c_vp]),
    "hcir_png_stage_batch": (c_int, [c_vp, c_vp, c_i64, c_i32, c_vp, c_sz, c_vp, c_vp, c_i32]),
    "hcir_png_workspace_bytes": (c_sz, [c_vp, c_i64, c_i32, c_i32]),
    "hcir_png_decode_window_u8": (c_int, [c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "hcir_resize_bicubic_ksize": (c_i32, [c_i32, c_i32]),
    "hcir_resize_bicubic_coeffs": (c_int, [c_i32, c_i32, c_vp, c_vp]),
    "hcir_resize_crop_workspace_bytes": (c_sz, [c_vp, c_i64, c_i32, c_i32]),
    "hcir_resize_crop_bicubic_u8": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_sz, c_vp]),
}


class HcirError(RuntimeError):
    pass


def source_hash() -> str:
    """sha256 (first 16 hex digits) over the kernel sources ON DISK (csrc/*.hip, *.h, include/hcir.h), in sorted name
    order.  The same digest is embedded in the binary at build time (`build_id()`)."""
    from ._srchash import source_hash as _h
    return _h(CSRC_DIR, os.path.join(os.path.dirname(_PKG_DIR), "include", "hcir.h"))


def build_id() -> str:
    """Identity of the LOADED binary: "<source hash it was compiled from>[ -D flags]" (hcir_build_id).  Recorded next
    to every PMC summary under profiles/; bench.py reports a summary only when it carries the loaded binary's id, so a
    stale .so or a tools/_libhcir_<tag>.so variant loaded through HCIR_LIB_PATH can never be credited with numbers
    taken on another binary."""
    return lib().hcir_build_id().decode()


def build(verbose: bool = False) -> str:
    """Compile libhcir.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC_DIR, "-j", str(min(8, os.cpu_count() or 1))]
    if not verbose:
        cmd.append("-s")
    subprocess.check_call(cmd)
    return LIB_PATH


def lib() -> ctypes.CDLL:
    """Load libhcir.so; raises HcirError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HcirError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or make -C csrc). "
                "There is no CPU fallback for the hcir hot path.")
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the header and the library disagree
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = lib().hcir_status_string(status).decode()
        raise HcirError(f"{what} failed: {msg} (status {status})")
