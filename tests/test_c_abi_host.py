"""Host halves of the C ABI on a box WITHOUT a GPU: every entry point that carves a caller-provided workspace must refuse a
workspace one byte short of its `*_workspace_bytes()` query with PANGNN_E_WORKSPACE — before it touches a pointer or launches
anything (the pointers handed over here are fakes) — and the argument checks in front of it must fire first for bad shapes.
These are also the calls `tools/run_sanitized.sh` drives through the AddressSanitizer / UBSan build of the library
(`make -C pangnn_amd/csrc san`): the workspace arithmetic, slab carving and launch geometry are host code."""
import ctypes as C

import pytest
import torch

from pangnn_amd import _lib

pytestmark = pytest.mark.skipif(torch.cuda.is_available(),
                                reason="fake device pointers: only where a missing check cannot reach a GPU")

F = 0x7f0000100000          # a 16-byte aligned address that is never dereferenced
E_BADARG, E_TOOLARGE, E_WORKSPACE, E_ALIGN = -1, -2, -3, -4
# sizes at which every kernel's grid is the whole chip, so that the entry needs the FULL `*_workspace_bytes()` (a smaller
# problem runs a smaller grid and legitimately accepts a smaller workspace: the check is against the slabs it will write)
N, E = 4_000_000, 60_000_000


def lib():
    return _lib.load()


def _decoder_common(dtype_arg=False):
    """(p, ldp, q, ldq, [pq_dtype,] num_nodes, edge_index, ld, num_edges, extra, cvec, w2, b2, w3, b3, D)"""
    head = [F, 64, F, 64] + ([0] if dtype_arg else []) + [N, F, E, E, None, None, F, F, F, F, 64]
    return head


def workspace_calls():
    """name -> (workspace query result, callable(workspace_bytes) -> rc)"""
    L = lib()
    out = {}
    q = L.pangnn_csr_build_workspace_bytes(E, N)
    out["pangnn_csr_build"] = (q, lambda b: L.pangnn_csr_build(F, E, E, N, 1, F, F, F, F, b, None))
    q = L.pangnn_decoder_mlp_bwd_workspace_bytes(E)
    out["pangnn_decoder_mlp_bwd_f32"] = (q, lambda b: L.pangnn_decoder_mlp_bwd_f32(
        *_decoder_common(), F, F, F, F, F, F, None, None, None, 0, F, b, None))
    out["pangnn_decoder_mlp_loss_f32"] = (q, lambda b: L.pangnn_decoder_mlp_loss_f32(
        *_decoder_common(), F, None, E, F, F, F, F, F, F, F, None, None, None, 0, F, b, None))
    q = L.pangnn_decoder_train_workspace_bytes()
    out["pangnn_decoder_train_f32"] = (q, lambda b: L.pangnn_decoder_train_f32(
        *_decoder_common(), F, None, E, None, F, F, F, None, None, F, F, F, None, None, F, b, None))
    out["pangnn_decoder_train_mixed"] = (q, lambda b: L.pangnn_decoder_train_mixed(
        *_decoder_common(True), F, None, E, None, F, F, F, None, None, F, F, F, None, None, F, b, None))
    q = L.pangnn_decoder_dgrad_workspace_bytes()
    out["pangnn_decoder_dgrad_f32"] = (q, lambda b: L.pangnn_decoder_dgrad_f32(F, None, None, F, F, E, None, None, F, None, F, b, None))
    q = L.pangnn_linear_wgrad_workspace_bytes(64, 128)
    out["pangnn_linear_wgrad_f32"] = (q, lambda b: L.pangnn_linear_wgrad_f32(F, 128, F, 64, N, 64, 128, F, F, F, b, None))
    out["pangnn_linear_act_wgrad_f32"] = (q, lambda b: L.pangnn_linear_act_wgrad_f32(F, 128, F, 64, N, 64, 128, 1, F, F, F, b, None))
    out["pangnn_linear_act_wgrad_mixed"] = (q, lambda b: L.pangnn_linear_act_wgrad_mixed(F, 1, 128, F, 1, 64, N, 64, 128, 1, F, F, F, b, None))
    q = L.pangnn_weighted_colsum_workspace_bytes(64)
    out["pangnn_weighted_colsum_f32"] = (q, lambda b: L.pangnn_weighted_colsum_f32(F, 64, F, F, N, 64, F, F, b, None))
    q = L.pangnn_weighted_colsum3_workspace_bytes(128)
    out["pangnn_weighted_colsum3"] = (q, lambda b: L.pangnn_weighted_colsum3(F, 0, 128, F, F, N, 128, F, F, b, None))
    q = L.pangnn_band_propagate_workspace_bytes(64)
    out["pangnn_band_propagate"] = (q, lambda b: L.pangnn_band_propagate(F, 0, 64, F, F, F, 64, N, 64, 1, F, F, b, None))
    q = L.pangnn_embed_conv_in_grads_workspace_bytes(128)
    out["pangnn_embed_conv_in_grads"] = (q, lambda b: L.pangnn_embed_conv_in_grads(F, 0, 128, F, F, N, F, F, F, 64, 128, F, F, F, F, F, b, None))
    q = L.pangnn_embed_linear_bwd_workspace_bytes(128, 64)
    out["pangnn_embed_linear_bwd"] = (q, lambda b: L.pangnn_embed_linear_bwd(F, 64, F, F, N, F, F, F, F, 64, 128, F, 64, F, F, F, F, b, None))
    q = L.pangnn_bce_logits_workspace_bytes()
    out["pangnn_bce_logits_f32"] = (q, lambda b: L.pangnn_bce_logits_f32(F, F, None, E, E, F, F, F, b, None))
    return out


def test_every_workspace_entry_is_covered():
    """the table above names every entry point of the header that takes (workspace, workspace_bytes)"""
    takers = sorted(n for n, (_, args) in _lib.SIGNATURES.items()
                    if any(a is C.c_size_t for a in args) and not n.endswith("_workspace_bytes"))
    assert takers == sorted(workspace_calls())


@pytest.mark.parametrize("name", sorted(workspace_calls()) if not torch.cuda.is_available() else [])
def test_workspace_one_byte_short_is_refused_before_anything_runs(name):
    need, call = workspace_calls()[name]
    if name == "pangnn_csr_build" and need == 0:
        # its query asks rocPRIM for the radix sort's temporary size, which needs a device: without one the query reports
        # failure (0) and the call refuses to run on it
        assert call(1 << 30) == E_BADARG and b"size query" in lib().pangnn_last_error()
        pytest.skip("rocPRIM's size query needs a device")
    assert need > 0, "the workspace query itself failed"
    rc = call(need - 1)
    assert rc == E_WORKSPACE, (name, rc, lib().pangnn_last_error())
    assert b"workspace" in lib().pangnn_last_error().lower()
    assert call(0) == E_WORKSPACE


def test_workspace_queries_scale_as_documented():
    L = lib()
    assert L.pangnn_csr_build_workspace_bytes(0, N) == 256                 # (non-empty lists: rocPRIM's size query needs a device)
    assert L.pangnn_decoder_mlp_bwd_workspace_bytes(2 * E) >= L.pangnn_decoder_mlp_bwd_workspace_bytes(E) > 0
    assert L.pangnn_linear_wgrad_workspace_bytes(128, 128) >= L.pangnn_linear_wgrad_workspace_bytes(64, 64) > 0
    assert L.pangnn_decoder_chunk_tiles_for(10) == 1 and L.pangnn_decoder_chunk_tiles_for(10 ** 8) == L.pangnn_decoder_chunk_tiles()


def test_shape_and_alignment_errors_come_before_the_workspace_check():
    L = lib()
    w = L.pangnn_linear_wgrad_workspace_bytes(64, 128)
    assert L.pangnn_linear_wgrad_f32(F, 128, F, 64, N, 63, 128, F, F, F, w, None) == E_BADARG        # K not covered
    assert L.pangnn_linear_wgrad_f32(F + 4, 128, F, 64, N, 64, 128, F, F, F, w, None) == E_ALIGN     # g not 16-byte aligned
    assert L.pangnn_linear_wgrad_f32(None, 128, F, 64, N, 64, 128, F, F, F, w, None) == E_BADARG     # null pointer
    ws = L.pangnn_decoder_train_workspace_bytes()
    args = _decoder_common()
    args[-1] = 32                                                                                     # D != 64
    assert L.pangnn_decoder_train_f32(*args, F, None, E, None, F, F, F, None, None, F, F, F, None, None, F, ws, None) == E_BADARG
    assert L.pangnn_scale_unless_one_f32(None, None, 0, None, None) == 0                              # nothing to do
    ptrs, counts = (C.c_void_p * 1)(F + 2), (C.c_int64 * 1)(8)
    assert L.pangnn_scale_unless_one_f32(ptrs, counts, 1, F, None) == E_ALIGN
    counts[0] = -1
    ptrs[0] = F
    assert L.pangnn_scale_unless_one_f32(ptrs, counts, 1, F, None) == E_BADARG


def test_sizes_beyond_the_int32_index_range_are_refused():
    L = lib()
    big = 2 ** 31 + 5
    assert L.pangnn_csr_build(None, big, big, 4, 1, C.c_void_p(16), None, None, None, 0, None) == E_TOOLARGE
    assert L.pangnn_decoder_dgrad_f32(F, None, None, F, F, big, None, None, F, None, F, L.pangnn_decoder_dgrad_workspace_bytes(),
                                      None) == E_TOOLARGE
    args = _decoder_common()
    args[6], args[7] = big, big                                                                       # ld, num_edges
    assert L.pangnn_decoder_train_f32(*args, F, None, E, None, F, F, F, None, None, F, F, F, None, None, F,
                                      L.pangnn_decoder_train_workspace_bytes(), None) == E_TOOLARGE


def test_storage_type_codes_are_checked_on_the_host():
    """PANGNN_DTYPE_F16 (2) is a storage type next to bfloat16; the 2-byte operands of one call share a format; an unknown code
    is refused everywhere — all before a pointer is touched"""
    L = lib()
    F32, BF16, F16 = 0, 1, 2
    # dense layer: bfloat16 x with a float16 result / float16 g with bfloat16 x
    assert L.pangnn_linear_act_fwd_mixed(F, BF16, 64, F, None, F, F16, 64, N, 64, 64, 0, None, F32, 0, None) == E_BADARG
    assert b"all bfloat16 or all float16" in L.pangnn_last_error()
    w = L.pangnn_linear_wgrad_workspace_bytes(64, 64)
    assert L.pangnn_linear_act_wgrad_mixed(F, F16, 64, F, BF16, 64, N, 64, 64, 0, F, F, F, w, None) == E_BADARG
    assert L.pangnn_linear_act_fwd_mixed(F, 3, 64, F, None, F, F32, 64, N, 64, 64, 0, None, F32, 0, None) == E_BADARG
    # 2-byte rows start on 8 bytes, not 16 (F + 8 is fine for f16 x, not for f32 x)
    assert L.pangnn_linear_act_fwd_mixed(F + 8, F32, 64, F, None, F, F32, 64, N, 64, 64, 0, None, F32, 0, None) == E_ALIGN
    # the decoder's tables: float32, bfloat16 or float16 (2-byte tables: row strides in multiples of 8 elements)
    args = _decoder_common(dtype_arg=True)
    args[4] = 3
    assert L.pangnn_decoder_mlp_infer_mixed(*args, F, None) == E_BADARG
    assert b"pq_dtype" in L.pangnn_last_error()
    args[4], args[1] = F16, 68
    assert L.pangnn_decoder_mlp_infer_mixed(*args, F, None) == E_BADARG
    # generated rows / column sums / band: 0, 1, 2 are the codes
    assert L.pangnn_rank2_rows(F, F, F, F, None, F, 3, 64, N, 64, None) == E_BADARG
    wsb = L.pangnn_weighted_colsum3_workspace_bytes(64)
    assert L.pangnn_weighted_colsum3(F, 3, 64, F, F, N, 64, F, F, wsb, None) == E_BADARG
    assert L.pangnn_weighted_colsum3(F + 4, F16, 64, F, F, N, 64, F, F, wsb, None) == E_ALIGN        # f16 rows: 8-byte aligned
    assert L.pangnn_band_propagate(F, 3, 64, F, None, F, 64, N, 64, 1, None, None, 0, None) == E_BADARG
    assert L.pangnn_spmm_csr_f16(F, F, F, F + 4, 64, N, None, F, 64, N, E, 64, 0, None) == E_ALIGN
    assert L.pangnn_spmm_csr_f16(F, F, F, F, 64, N, None, F, 64, N, E, 48, 0, None) == E_BADARG      # F not 32 / 64 / 128 / 256
