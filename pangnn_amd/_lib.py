"""ctypes binding of libpangnn_hip.so (C ABI declared in include/pangnn_hip.h).

There is NO fallback: if the shared library is missing or a symbol is absent, importing the
operators raises.  Tensors cross the boundary as raw device pointers (`tensor.data_ptr()`), sizes
and the current HIP stream handle; PyTorch is only the allocator / stream owner.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PANGNN_HIP_LIB") or os.path.join(_HERE, "libpangnn_hip.so")   # override: diagnostic builds (tools/)

_i64, _i32, _p, _sz = C.c_int64, C.c_int32, C.c_void_p, C.c_size_t

# name -> (restype, argtypes); mirrors include/pangnn_hip.h one to one
SIGNATURES = {
    "pangnn_abi_version": (C.c_int, []),
    "pangnn_last_error": (C.c_char_p, []),
    "pangnn_csr_build_workspace_bytes": (_sz, [_i64, _i64]),
    "pangnn_csr_build": (C.c_int, [_p, _i64, _i64, _i64, C.c_int, _p, _p, _p, _p, _sz, _p]),
    "pangnn_csr_build_flag_ptr": (_p, [_p, _i64]),
    "pangnn_structure_small_supported": (C.c_int, [_i64, _i64]),
    "pangnn_structure_small": (C.c_int, [_p, _i64, _i64, _i64, _i32] + [_p] * 15 + [_p]),
    "pangnn_collate_subgraphs_padded": (C.c_int, [_p, _i64, _p, _i64, _p, _p, _p, _p, _p, _i64, _p, _i32, _i64, _i64, _i64,
                                                  _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "pangnn_set_i64": (C.c_int, [_p, _p, _i32, _p]),
    "pangnn_collate_subgraphs": (C.c_int, [_p, _i64, _i64, _i64, _p, _i64, _i64, _i64, _p, _i32, _i64, _i64,
                                           _p, _p, _p, _p, _p, _p]),
    "pangnn_embed_linear_supported": (C.c_int, [_i32, _i32]),
    "pangnn_embed_linear_fwd": (C.c_int, [_p, _p, _i64, _p, _p, _p, _p, _i32, _i32, _p, _p, _i32, _p, _i64, _p]),
    "pangnn_embed_linear_bwd_workspace_bytes": (_sz, [_i32, _i32]),
    "pangnn_embed_linear_bwd": (C.c_int, [_p, _i64, _p, _p, _i64, _p, _p, _p, _p, _i32, _i32, _p, _i32, _p, _p, _p, _p, _sz, _p]),
    "pangnn_embed_conv_in_grads_from_sums": (C.c_int, [_p, _p, _p, _p, _i32, _i32, _p, _p, _p, _p, _p]),
    "pangnn_gcn_norm_f32": (C.c_int, [_p, _p, _p, _p, _i64, _i64, _p, _p, _p, _p]),
    "pangnn_gcn_degree_f32": (C.c_int, [_p, _p, _p, _i64, _p, _p]),
    "pangnn_gcn_edge_norm_f32": (C.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _p, _p, _p]),
    "pangnn_permute_f32": (C.c_int, [_p, _p, _p, _i64, _p]),
    "pangnn_spmm_csr_f32": (C.c_int, [_p, _p, _p, _p, _i64, _i64, _p, _p, _i64, _i64, _i64, _i32, C.c_int, _p]),
    "pangnn_spmm_csr_bf16": (C.c_int, [_p, _p, _p, _p, _i64, _i64, _p, _p, _i64, _i64, _i64, _i32, C.c_int, _p]),
    "pangnn_spmm_csr_f16": (C.c_int, [_p, _p, _p, _p, _i64, _i64, _p, _p, _i64, _i64, _i64, _i32, C.c_int, _p]),
    "pangnn_edge_gather_concat_f32": (C.c_int, [_p, _i64, _i64, _p, _i64, _i64, _i64, _p, _p, _i64, _i32, _p]),
    "pangnn_edge_pair_add_f32": (C.c_int, [_p, _p, _i64, _i64, _p, _i64, _i64, _i64, _p, _p, _p, _i64, _i32, _p]),
    "pangnn_segment_sum_rows_f32": (C.c_int, [_p, _p, _p, _i64, _i64, _i64, _p, _i64, _i64, _i32, C.c_int, _p]),
    "pangnn_segment_max_rows_f32": (C.c_int, [_p, _p, _p, _i64, _p, _p, _i64, _i64, _i32, _p]),
    "pangnn_segment_max_bwd_f32": (C.c_int, [_p, _p, _p, _p, _i64, _i64, _i64, _i32, _p]),
    "pangnn_linear_supported": (C.c_int, [_i32, _i32, C.c_int]),
    "pangnn_linear_fwd_f32": (C.c_int, [_p, _i64, _p, _p, _p, _i64, _i64, _i32, _i32, _p]),
    "pangnn_linear_wgrad_workspace_bytes": (_sz, [_i32, _i32]),
    "pangnn_linear_wgrad_f32": (C.c_int, [_p, _i64, _p, _i64, _i64, _i32, _i32, _p, _p, _p, _sz, _p]),
    "pangnn_linear_act_fwd_f32": (C.c_int, [_p, _i64, _p, _p, _p, _i64, _i64, _i32, _i32, _i32, _p, _i64, _p]),
    "pangnn_linear_act_wgrad_f32": (C.c_int, [_p, _i64, _p, _i64, _i64, _i32, _i32, _i32, _p, _p, _p, _sz, _p]),
    # bf16-storage modes: (x, x_dtype, ldx, w, bias, y, y_dtype, ldy, n, K, M, in_act, gate, gate_dtype, ldgate, stream)
    "pangnn_linear_dgrad_mixed": (C.c_int, [_p, _i32, _i64, _p, _p, _i32, _i64, _i64, _i32, _i32, _p, _i32, _i64, _p]),
    "pangnn_node_actions_f32": (C.c_int, [_p, _p, _p, _p, _i64, _p, _p, _p]),
    "pangnn_pq_operands_f32": (C.c_int, [_p, _i64, _p, _i32, C.c_int, _p, _p, _p, _p]),
    "pangnn_colsum_small": (C.c_int, [_p, _i32, _i64, _i64, _i32, _p, _p]),
    "pangnn_linear_act_fwd_mixed": (C.c_int, [_p, _i32, _i64, _p, _p, _p, _i32, _i64, _i64, _i32, _i32, _i32, _p, _i32,
                                              _i64, _p]),
    # (g, g_dtype, ldg, x, x_dtype, ldx, n, K, M, in_act, gw, gb, workspace, bytes, stream)
    "pangnn_linear_act_wgrad_mixed": (C.c_int, [_p, _i32, _i64, _p, _i32, _i64, _i64, _i32, _i32, _i32, _p, _p, _p, _sz,
                                                _p]),
    "pangnn_confusion_update_f32": (C.c_int, [_p, _p, _i64, C.c_float, C.c_int, _p, _p]),
    "pangnn_weighted_colsum_workspace_bytes": (_sz, [_i32]),
    "pangnn_weighted_colsum_f32": (C.c_int, [_p, _i64, _p, _p, _i64, _i32, _p, _p, _sz, _p]),
    "pangnn_band_propagate_workspace_bytes": (_sz, [_i32]),
    "pangnn_band_propagate": (C.c_int, [_p, _i32, _i64, _p, _p, _p, _i64, _i64, _i32, _i32, _p, _p, _sz, _p]),
    "pangnn_embed_conv_in_rows": (C.c_int, [_p, _p, _p, _p, _p, _p, _i32, _p, _i32, _i64, _i64, _i32, _p]),
    "pangnn_embed_conv_in_grads_workspace_bytes": (_sz, [_i32]),
    "pangnn_embed_conv_in_grads": (C.c_int, [_p, _i32, _i64, _p, _p, _i64, _p, _p, _p, _i32, _i32, _p, _p, _p, _p, _p, _sz,
                                             _p]),
    "pangnn_rank2_rows": (C.c_int, [_p, _p, _p, _p, _p, _p, _i32, _i64, _i64, _i32, _p]),
    "pangnn_weighted_colsum3_workspace_bytes": (_sz, [_i32]),
    "pangnn_weighted_colsum3": (C.c_int, [_p, _i32, _i64, _p, _p, _i64, _i32, _p, _p, _sz, _p]),
    "pangnn_bce_logits_workspace_bytes": (_sz, []),
    "pangnn_bce_logits_f32": (C.c_int, [_p, _p, _p, _i64, _i64, _p, _p, _p, _sz, _p]),
    "pangnn_decoder_mlp_fwd_f32": (C.c_int, [_p, _i64, _p, _i64, _i64, _p, _i64, _i64, _p, _p, _p, _p, _p, _p, _i32,
                                             _p, _p]),
    "pangnn_decoder_mlp_infer_f32": (C.c_int, [_p, _i64, _p, _i64, _i64, _p, _i64, _i64, _p, _p, _p, _p, _p, _p, _i32,
                                               _p, _i32, _p]),
    "pangnn_decoder_mlp_bwd_workspace_bytes": (_sz, [_i64]),
    "pangnn_decoder_mlp_bwd_f32": (C.c_int, [_p, _i64, _p, _i64, _i64, _p, _i64, _i64, _p, _p, _p, _p, _p, _p, _i32,
                                             _p, _p, _p, _p, _p, _p, _p, _p, _p,      # g_logits .. part_off
                                             _i32, _p, _sz, _p]),                     # precision, workspace, bytes, stream
    "pangnn_decoder_mlp_loss_f32": (C.c_int, [_p, _i64, _p, _i64, _i64, _p, _i64, _i64, _p, _p, _p, _p, _p, _p, _i32,
                                              _p, _p, _i64,                            # y, pos_weight, denom
                                              _p, _p, _p, _p, _p, _p, _p, _p, _p, _p,  # logits .. part_off
                                              _i32, _p, _sz, _p]),
    "pangnn_softmax_qscore_f64": (C.c_int, [_p, _p, _i64, _i64, C.c_double, C.c_double, C.c_double, _p, _p]),
    # two-wave-per-SIMD training decoder (csrc/decoder16.hip): S kernel (+ by-source run sums, per-edge records)
    "pangnn_decoder_chunk_tiles": (C.c_int, []),
    "pangnn_decoder_chunk_tiles_for": (C.c_int, [_i64]),
    "pangnn_decoder_train_workspace_bytes": (_sz, []),
    "pangnn_decoder_train_f32": (C.c_int, [_p, _i64, _p, _i64, _i64, _p, _i64, _i64, _p, _p, _p, _p, _p, _p, _i32,
                                           _p, _p, _i64, _p,                        # y, pos_weight, denom, g_logits
                                           _p, _p, _p, _p, _p,                      # logits, loss, rec, part_buf, part_off
                                           _p, _p, _p, _p, _p,                      # g_w2, g_w3, g_b3, g_cvec, live_edges
                                           _p, _sz, _p]),
    "pangnn_decoder_train_mixed": (C.c_int, [_p, _i64, _p, _i64, _i32, _i64, _p, _i64, _i64, _p, _p, _p, _p, _p, _p, _i32,
                                             _p, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "pangnn_decoder_mlp_infer_mixed": (C.c_int, [_p, _i64, _p, _i64, _i32, _i64, _p, _i64, _i64, _p, _p, _p, _p, _p, _p,
                                                 _i32, _p, _p]),
    # T kernel: dL/dh1 run sums in a permuted (CSR) edge order from the records
    "pangnn_decoder_dgrad_workspace_bytes": (_sz, []),
    "pangnn_decoder_dgrad_f32": (C.c_int, [_p, _p, _p, _p, _p, _i64, _p, _p, _p, _p, _p, _sz, _p]),
    # (host array of device pointers, host array of element counts, n_tensors, device scalar, stream)
    "pangnn_scale_unless_one_f32": (C.c_int, [_p, _p, _i32, _p, _p]),
}

ABI_VERSION = 3          # PANGNN_ABI_VERSION of include/pangnn_hip.h this binding was written against

_lib = None


class PangnnHipError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle.  Raises if the HIP extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PangnnHipError(
            f"pangnn_amd: HIP extension {LIB_PATH} is missing. Build it with "
            f"`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C pangnn_amd/csrc`). "
            f"There is no CPU / PyTorch fallback for this path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here == ABI mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.pangnn_abi_version() != ABI_VERSION:
        raise PangnnHipError(f"pangnn_amd: {LIB_PATH} has ABI version {lib.pangnn_abi_version()}, this binding needs "
                             f"{ABI_VERSION} — rebuild it from this tree (`make -C pangnn_amd/csrc`); a library selected "
                             f"through PANGNN_HIP_LIB must come from the same tree")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().pangnn_last_error().decode("utf-8", "replace")
        raise PangnnHipError(f"{what or 'pangnn_hip'} failed (rc={rc}): {msg}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr() -> int:
    """hipStream_t of torch's current stream on the current device (the raw handle: `torch.cuda.current_stream()` builds a
    Stream object per call, ~10 us — a tenth of a launch-bound mini-batch step's host time)."""
    if _raw_stream is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


class _NoGuard:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NO_GUARD = _NoGuard()


def device_guard(dev):
    """`with torch.cuda.device(dev)` that costs nothing when `dev` already is the current device (the usual case)"""
    if _cur_device is not None and dev.index is not None and dev.index == _cur_device():
        return _NO_GUARD
    return torch.cuda.device(dev)


def ptr(t):
    """Device pointer of a tensor, or NULL for None."""
    return None if t is None else t.data_ptr()


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise PangnnHipError(
                "pangnn_amd: operands must live on the GPU (HIP device tensors); this package has "
                "no CPU fallback. Got a tensor on " + str(t.device))
