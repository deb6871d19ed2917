"""ctypes binding of the C-ABI shared library ``libagcn_hip.so`` (declared in ``include/agcn_hip.h``).

The library is built in-tree by ``__graft_entry__.build()`` (``hipcc --offload-arch=gfx950``).  There is NO
fallback: if the library is missing or a call fails, a ``RuntimeError`` is raised (the product path never
routes through the CPU oracle or stock PyTorch operators for the hot path).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libagcn_hip.so")

_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float
_D = ctypes.c_double
_Z = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/agcn_hip.h one to one
SIGNATURES = {
    "agcn_version": (_I, []),
    "agcn_arch": (ctypes.c_char_p, []),
    "agcn_last_kernel": (ctypes.c_char_p, []),
    "agcn_gemm_mode": (ctypes.c_char_p, []),
    "agcn_chain_mode": (ctypes.c_char_p, []),
    "agcn_conv_tile_frames": (_I, [_I, _I]),
    "agcn_conv_num_tiles": (_I, [_I, _I]),
    "agcn_conv_stats_tiles": (_I, [_I, _I, _I, _I, _I, _I]),
    "agcn_dadj_num_slots": (_I, [_I, _I, _I]),
    "agcn_scores_num_tiles": (_I, [_I, _I]),
    "agcn_conv_workspace": (_Z, [_I, _I, _I, _I, _I, _I]),
    "agcn_gcn_workspace": (_Z, [_I, _I, _I, _I]),
    "agcn_gcn_stats_tiles": (_I, [_I, _I, _I, _I]),
    "agcn_gcn_stats_slots": (_I, [_I, _I, _I, _I, _I]),
    "agcn_conv_fwd": (_I, [_P, _P, _P, _P, _P, _P, _Z, _I, _I, _I, _I, _I, _I, _I, _P]),
    "agcn_conv_bwd_data": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _P, _Z, _I, _I, _I, _I, _I, _I, _I, _P]),
    "agcn_conv_bwd_weight_workspace": (_Z, [_I, _I, _I, _I, _I, _I, _I]),
    "agcn_conv_bwd_weight": (_I, [_P, _P, _P, _P, _Z, _I, _I, _I, _I, _I, _I, _I, _P]),
    "agcn_gcn_aggregate_project_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _Z, _I, _I, _I, _I, _I, _P]),
    "agcn_gcn_aggregate_project_fwd_ex": (_I, [_P, _P, _P, _P, _P, _P, _P, _Z, _I, _I, _I, _I, _I, _P, _P]),
    "agcn_gcn_aggregate_project_bwd_data_ex": (_I, [_P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _P, _I, _P, _Z, _I, _I, _I, _I, _I, _P, _P, _P]),
    "agcn_gcn_dadj_ex": (_I, [_P, _P, _P, _P, _P, _Z, _I, _I, _I, _I, _I, _P, _P, _P]),
    "agcn_adjacency_bwd_scores_ex": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "agcn_conv_bwd_weight_ex": (_I, [_P, _P, _P, _P, _Z, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "agcn_gcn_project_bwd_weight_ex": (_I, [_P, _P, _P, _P, _P, _Z, _I, _I, _I, _I, _I, _P, _P, _P]),
    "agcn_absmax": (_I, [_P, ctypes.c_long, _P, _P]),
    "agcn_bn_act_fwd_ex": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "agcn_bn_bwd_apply_ex": (_I, [_P, _I, _D, _F, _P, _P, _I] + [_P] * 16 + [_I, _I, _I, _P]),
    "agcn_conv_fwd_ex": (_I, [_P, _P, _P, _P, _P, _P, _Z, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "agcn_conv_bwd_data_ex": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _P, _Z, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "agcn_gcn_first_supported": (_I, [_I, _I, _I]),
    "agcn_gcn_first_tiles": (_I, [_I, _I]),
    "agcn_gcn_first_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "agcn_gcn_unit_infer_workspace": (_Z, [_I, _I, _I, _I, _I]),
    "agcn_gcn_unit_infer": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _Z, _I, _I, _I, _I, _I, _P]),
    "agcn_conv9_infer": (_I, [_P, _P, _P, _P, _I, _P, _P, _Z, _I, _I, _I, _I, _I, _I, _P]),
    "agcn_gcn_aggregate_project_bwd_data": (_I, [_P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _P, _Z, _I, _I, _I, _I, _I, _P]),
    "agcn_gcn_bwd_data_fused_supported": (_I, [_I, _I, _I]),
    "agcn_gcn_aggregate_project_bwd_data_fused": (_I, [_P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _P, _I, _P, _Z, _I, _I, _I, _I, _I, _P]),
    "agcn_gcn_dadj": (_I, [_P, _P, _P, _P, _P, _Z, _I, _I, _I, _I, _I, _P]),
    "agcn_gcn_project_bwd_weight_workspace": (_Z, [_I, _I, _I, _I, _I]),
    "agcn_gcn_project_bwd_weight": (_I, [_P, _P, _P, _P, _P, _Z, _I, _I, _I, _I, _I, _P]),
    "agcn_adjacency_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "agcn_adjacency_fused_supported": (_I, [_I, _I, _I, _I]),
    "agcn_adjacency_fused_workspace": (_Z, [_I, _I]),
    "agcn_adjacency_fused_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _I, _I, _I, _I, _I, _P]),
    "agcn_adjacency_fused_fwd_ex": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _I, _I, _I, _I, _I, _P]),
    "agcn_adjacency_fused_bwd_scores": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _I, _I, _I, _I, _I, _P]),
    "agcn_adjacency_bwd_softmax": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "agcn_adjacency_bwd_scores": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "agcn_colsum_scratch_bytes": (_Z, [_I]),
    "agcn_colsum": (_I, [_P, _I, _I, _P, _P, _P]),
    "agcn_bn_stats_finalize": (_I, [_P, _I, _I, _D, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P]),
    "agcn_bn_eval_coeff": (_I, [_P, _P, _P, _P, _F, _I, _P, _P, _P]),
    "agcn_bn_act_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "agcn_bn_bwd": (_I, [_P, _P, _I] + [_P] * 16 + [_I, _I, _I, _P]),
    "agcn_bn_bwd_reduce": (_I, [_P, _P, _I, _P, _P, _P, _I, _I, _I, _P]),
    "agcn_bn_bwd_apply": (_I, [_P, _I, _D, _F, _P, _P, _I] + [_P] * 15 + [_I, _I, _I, _P]),
    "agcn_stc_row_reduce": (_I, [_P, _P, _P, _P, _I, _P, _P, _F, _F, _I, _I, _I, _I, _P]),
    "agcn_stc_apply": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "agcn_stc_apply_ex": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "agcn_stc_bwd_apply": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "agcn_data_bn_stats": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "agcn_data_bn_apply": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "agcn_data_bn_bwd_reduce": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "agcn_data_bn_bwd_apply": (_I, [_P, _P, _P, _P, _P, _P, _D, _P, _I, _I, _I, _I, _I, _P]),
    "agcn_pool_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "agcn_pool_bwd": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "agcn_linear_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "agcn_linear_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "agcn_gate_conv_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "agcn_gate_conv_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "agcn_sgd_step_workspace": (_Z, [ctypes.c_long]),
    "agcn_sgd_step": (_I, [_P, _P, _P, ctypes.c_long, _F, _F, _F, _I, _F, _F, _I, _P, _Z, _P, _P]),
}

_lib = None


def load():
    """Load the shared library (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"agcn_amd: HIP extension {LIB_PATH} not found -- run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (hipcc --offload-arch=gfx950).  There is no CPU fallback for the hot path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def ptr_bits(t):
    """Device pointer of an int32 sign-bit-mask tensor (None -> NULL)."""
    if t is None:
        return None
    if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.int32):
        raise RuntimeError(f"agcn_amd: expected a contiguous int32 GPU tensor, got {t.dtype} on {t.device}")
    return t.data_ptr()


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  The tensor must be contiguous fp32 on a GPU."""
    if t is None:
        return None
    if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
        raise RuntimeError(f"agcn_amd: expected a contiguous fp32 GPU tensor, got {t.dtype} "
                           f"{tuple(t.shape)} on {t.device} contiguous={t.is_contiguous()}")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"agcn_amd: {what} failed with code {rc}")
