"""ctypes binding of ``libhgvae_hip.so`` (the C ABI declared in ``include/hgvae_hip.h``).

The shared library is built in-tree (``hgvae_amd/csrc/Makefile``).  There is no
fallback: if it is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HGVAE_LIB") or os.path.join(_HERE, "csrc", "libhgvae_hip.so")   # HGVAE_LIB: another build of the library

_vp, _i64, _i32, _f32, _u64, _sz = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_uint64, C.c_size_t

# name -> (restype, argtypes); must list every symbol of include/hgvae_hip.h
SIGNATURES = {
    "hgv_version": (C.c_int, []),
    "hgv_last_error": (C.c_char_p, []),
    "hgv_csr_build_ws": (_sz, [_i64, _i64]),
    "hgv_csr_build": (C.c_int, [_vp, _i64, _i64, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "hgv_sort_pairs_ws": (_sz, [_i64]),
    "hgv_sort_pairs_u32": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "hgv_agg_fwd": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp]),
    "hgv_agg_bwd": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp, _i64, _vp, _i64, _vp]),
    "hgv_csr_long_rows": (C.c_int, [_vp, _i64, _i32, _vp, _i32, _vp, _vp]),
    "hgv_agg_fwd_long": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i32, _vp, _vp,
                                   _i32, _i32, _vp]),
    "hgv_bn_act_ws": (_sz, [_i64, _i32]),
    "hgv_bn_act_fwd": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _f32, _vp, _u64, _u64, _vp, _vp, _vp, _f32, _vp, _vp,
                                 _f32, _vp, _vp, _vp, _vp, _i64, _vp, _sz, _vp]),
    "hgv_bn_act_fwd_static": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _f32, _vp, _u64, _u64, _vp, _vp, _vp, _f32, _vp, _vp,
                                        _f32, _vp, _vp, _vp, _vp, _i64, _vp]),
    "hgv_bn_act_eval": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _vp, _vp, _f32, _vp, _vp, _vp, _i64, _vp]),
    "hgv_bn_act_bwd": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _f32, _vp, _u64, _u64, _vp, _vp, _vp, _vp,
                                 _vp, _i64, _vp, _vp, _i32, _vp, _sz, _vp]),
    "hgv_bn_act_fwd2": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _f32, _u64, _u64, _u64, _vp, _vp, _vp, _f32, _vp, _vp,
                                  _f32, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _sz, _vp]),
    "hgv_bn_act_bwd2": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _f32, _u64, _u64, _u64,
                                  _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i32, _vp, _sz, _vp]),
    "hgv_exclusive_scan_ws": (_sz, [_i64]),
    "hgv_exclusive_scan_i32": (C.c_int, [_vp, _vp, _i64, _vp, _sz, _vp]),
    "hgv_gat_fwd": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _i64, _i32, _vp, _i64, _vp, _i64, _f32, _f32, _vp, _u64, _u64, _vp, _vp,
                              _vp, _i64, _vp, _vp, _vp]),
    "hgv_gat_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _vp, _i64, _vp, _i64, _f32, _f32,
                              _vp, _u64, _u64, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp]),
    "hgv_gat_fwd_long": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _i64, _i32, _vp, _i64, _vp, _i64, _f32, _f32, _vp, _u64, _u64, _vp, _vp,
                              _vp, _i64, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _vp]),
    "hgv_gat_bwd_long": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _vp, _i64, _vp, _i64, _f32, _f32,
                              _vp, _u64, _u64, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _vp]),
    "hgv_topk_ws": (_sz, [_i64]),
    "hgv_topk_perm": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _sz, _vp]),
    "hgv_filter_adj_ws": (_sz, [_i64, _i64]),
    "hgv_filter_adj": (C.c_int, [_vp, _i64, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i32, _vp, _sz, _vp]),
    "hgv_gather_rows": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _i32, _vp, _i64, _vp]),
    "hgv_gather_rows_bwd": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _i32, _vp, _i64, _vp, _vp]),
    "hgv_gather_i64": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "hgv_pack_rows": (C.c_int, [_i32]),
    "hgv_pack_bytes": (_sz, [_i32, _i32]),
    "hgv_pack_weight": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _vp, _vp]),
    "hgv_pdb_build": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _i64, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _vp]),
    "hgv_collate_nodes": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i64, _i64, _vp, _i64, _vp, _vp, _vp,
                                    _vp]),
    "hgv_collate_edges": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _i64, _i64, _vp,
                                    _vp, _vp, _vp, _vp, _vp, _vp]),
    "hgv_pack_weights": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hgv_pack_weight_rows": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _vp, _i32, _i32, _vp]),
    "hgv_gemm_nt_packed": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _i64, _i32, _vp, _vp, _i64, _i32, _vp, _i64, _vp]),
    "hgv_gemm_nt_ws": (_sz, [_i32, _i32, _i32]),
    "hgv_gemm_nt": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i64, _vp, _i64, _i32, _vp, _i64, _i64, _vp, _i64, _i32, _vp,
                              _i64, _vp, _sz, _vp]),
    "hgv_gemm_nt16_supported": (C.c_int, [_i32, _i32, _i32]),
    "hgv_pack16_bytes": (_sz, [_i32, _i32, _i32]),
    "hgv_pack16_weights": (C.c_int, [_vp, _i64, _i64, _i32, _vp, _i64, _i64, _i32, _i32, _vp, _vp]),
    "hgv_pack16_weights_multi": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hgv_gemm_nt16": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _vp, _i64, _i32, _vp, _i64, _vp]),
    "hgv_gemm_nt16h_supported": (C.c_int, [_i32, _i32, _i32]),
    "hgv_pack16h_bytes": (_sz, [_i32, _i32, _i32]),
    "hgv_pack16h_weights_multi": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hgv_gemm_nt16h": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _vp, _i64, _i32, _vp, _i64, _vp]),
    "hgv_pack16h_weights_affine": (C.c_int, [_vp, _i64, _i64, _i32, _vp, _vp, _i32, _vp, _i64, _i64, _i32, _vp, _vp, _i32, _vp, _vp,
                                             _vp, _vp]),
    "hgv_agg_fwd_ind": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _i32, _vp, _i64, _i32, _i32, _vp, _vp, _i32, _i32, _vp]),
    "hgv_agg_fwd_affine": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _i32, _i32, _vp]),
    "hgv_bn_stats_act": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _f32, _vp, _u64, _u64, _vp, _vp, _vp, _f32, _vp, _vp, _f32, _vp,
                                   _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "hgv_bn_apply": (C.c_int, [_vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "hgv_bn_act_bwd_pre": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _i64, _vp, _vp,
                                     _i32, _vp, _sz, _vp]),
    "hgv_bn_act_bwd_pre_alg": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _f32, _vp, _vp, _vp, _i32, _vp, _vp, _vp,
                                         _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i32, _vp, _sz, _vp]),
    "hgv_bn_bwd_sums_alg": (C.c_int, [_i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "hgv_gemm_nt16h_bwd_supported": (C.c_int, [_i64, _i32, _i32, _i32, _i32]),
    "hgv_gemm_nt16h_bwd": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _i64, _i32, _vp, _i64, _vp, _i64, _i32, _f32, _vp, _vp, _vp,
                                     _vp, _vp, _i64, _vp]),
    "hgv_bn_stats_act2": (C.c_int, [_vp, _vp, _i64, _i64, _i32, _i32, _f32, _u64, _u64, _u64, _vp, _vp, _vp, _f32, _vp, _vp, _f32, _vp,
                                    _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "hgv_bn_act_bwd2_pre": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _i64, _i32, _i32, _f32, _vp, _vp,
                                      _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i32, _vp, _sz, _vp]),
    "hgv_wgrad_finish": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "hgv_row_range_flag": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _i64, _i32, _vp, _vp]),
    "hgv_gemm_nt16h_gated": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _vp, _i64, _i32, _vp, _i64, _vp, _i32, _vp]),
    "hgv_gemm_nt16h_act_supported": (C.c_int, [_i32, _i32, _i32]),
    "hgv_gemm_nt16h_act": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _vp, _i64, _i32, _vp, _i64, _vp, _i64, _i32, _f32, _vp]),
    "hgv_gemm_nt16h_act_parts": (C.c_int, [_i64, _i32, _i32, _i32]),
    "hgv_gemm_nt16h_act_stats": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _vp, _i64, _i32, _vp, _i64, _vp, _i64, _i32, _f32, _vp,
                                           _vp]),
    "hgv_gemm_nt16h_act2_supported": (C.c_int, [_i64, _i32, _i32, _i32]),
    "hgv_gemm_nt16h_act2_parts": (C.c_int, [_i64]),
    "hgv_gemm_nt16h_act2_stats": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _vp, _i64, _i32, _vp, _vp, _i64, _vp, _vp, _i64, _i32,
                                            _f32, _vp, _vp, _vp]),
    "hgv_bn_stats_merge": (C.c_int, [_vp, _i32, _i32, _i64, _i32, _vp, _vp, _f32, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hgv_dropout_bits": (C.c_int, [_i64, _i32, _f32, _vp, _u64, _u64, _vp, _vp, _i64, _vp]),
    "hgv_bn_stats_pre": (C.c_int, [_vp, _i64, _i64, _i32, _vp, _vp, _f32, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "hgv_gemm_nt16_gated": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _vp, _i64, _i32, _vp, _i64, _vp, _i32, _vp]),
    "hgv_distance_matrix": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _vp, _vp]),
    "hgv_normalized_laplacian": (C.c_int, [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _i64, _vp, _vp]),
    "hgv_adam_step": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _i64, _vp, _vp, _vp]),
    "hgv_grad_norm_ws": (_sz, []),
    "hgv_grad_norm_clip": (C.c_int, [_vp, _i64, _f32, _vp, _vp, _vp, _sz, _vp]),
    "hgv_gemm_tn_ws": (_sz, [_i64, _i32, _i32]),
    "hgv_gemm_tn": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _i64, _vp, _i64, _vp, _i32, _vp, _sz, _vp]),
    "hgv_gemm_tn2_ws": (_sz, [_i64, _i32, _i32, _i32]),
    "hgv_gemm_tn16_supported": (C.c_int, [_i64, _i32, _i32, _i32]),
    "hgv_gemm_tn16_ws": (_sz, [_i64, _i32, _i32, _i32]),
    "hgv_gemm_tn16": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _i64, _vp, _i64, _i32, _vp, _i64, _i64, _vp, _i32,
                                _vp, _sz, _vp]),
    "hgv_gemm_tn16h": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _i64, _vp, _i64, _i32, _vp, _i64, _i64, _vp, _i32,
                                 _vp, _sz, _vp]),
    "hgv_gemm_tn2": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _i64, _vp, _i64, _i32, _vp, _i64, _i64, _vp, _i32,
                               _vp, _sz, _vp]),
    "hgv_gemm_tn_gg_ws": (_sz, [_i64, _i32, _i32]),
    "hgv_gemm_tn_gg": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _vp, _i64, _i32, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i32, _vp, _sz,
                                 _vp]),
    "hgv_gemm_tn16_gg": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _vp, _i64, _i32, _i64, _vp, _i64, _vp, _i32, _vp, _sz, _vp]),
    "hgv_loss_ws": (_sz, []),
    "hgv_reparam_fwd": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _vp, _u64, _u64, _vp, _vp, _i64, _vp, _vp]),
    "hgv_reparam_bwd": (C.c_int, [_vp, _i64, _vp, _i64, _i64, _i32, _vp, _vp, _i64, _vp]),
    "hgv_kl_fwd": (C.c_int, [_vp, _i64, _i64, _i32, _vp, _vp, _sz, _vp]),
    "hgv_kl_bwd": (C.c_int, [_vp, _i64, _i64, _i32, _vp, _vp, _i64, _i32, _vp]),
    "hgv_mse_fwd": (C.c_int, [_vp, _i64, _vp, _i64, _i64, _i32, _vp, _vp, _sz, _vp]),
    "hgv_mse_bwd": (C.c_int, [_vp, _i64, _vp, _i64, _i64, _i32, _vp, _vp, _i64, _vp]),
    "hgv_mse_fwd_grad": (C.c_int, [_vp, _i64, _vp, _i64, _i64, _i32, _vp, _vp, _i64, _vp, _sz, _vp]),
    "hgv_mse_bwd_unless_unit": (C.c_int, [_vp, _i64, _vp, _i64, _i64, _i32, _vp, _vp, _i64, _vp]),
}

_lib = None


class HgvError(RuntimeError):
    pass


def load():
    """Load the library once; raise if the built artefact is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HgvError(
            f"{LIB_PATH} not found: build it with `make -C hgvae_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def call(name: str, *args):
    """Call an int-returning entry point and raise on a non-zero status."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise HgvError(f"{name} failed ({rc}): {lib.hgv_last_error().decode()}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """Handle of torch's current stream on the current device (every launch goes there).  The raw accessors skip the
    Stream object that ``torch.cuda.current_stream()`` builds - ~8 us per call, paid ~150 times per train step."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise HgvError("hgvae_amd operators run on the GPU only (got a CPU tensor); there is no CPU fallback")
