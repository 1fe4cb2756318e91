"""The C-ABI library loads without a GPU and exports every symbol include/mvnerf_hip.h declares;
the ctypes table in _lib.py lists exactly those symbols.  No compute calls here."""
import ctypes
import os
import re

from thesis_clip_nerf_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, 'include', 'mvnerf_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(mvnerf_[a-z0-9_]+)\s*\(', text)))


def test_header_and_ctypes_table_agree():
    decl = declared_functions()
    assert len(decl) >= 10
    assert decl == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol():
    lib = _lib.lib()
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert lib.mvnerf_abi_version() == 1
    assert lib.mvnerf_packed_net_floats() == 251656
    assert lib.mvnerf_render_workspace_bytes(1, 1, 4096, 64) == (16 * 64 + 128) * 4096 * 4
    assert lib.mvnerf_field_workspace_bytes(2, 3, 10) == 2 * 3 * 10 * 128 * 4


def test_argument_validation_without_gpu():
    lib = _lib.lib()
    assert lib.mvnerf_pack_net(None, None, None) == -1
    assert b'null' in lib.mvnerf_last_error()
    one = ctypes.c_void_p(16)
    assert lib.mvnerf_composite(one, one, 4, 100, one, one, None, None) == -2          # S=100 unsupported
    assert lib.mvnerf_resample(one, one, one, 4, 32, 0, one, None, None, None, None, None) == -2
    fe_tail = [one, None, None, None, None, None, one, None]   # rgbs, tap_idx, pix, embedding, acts x2, workspace, stream
    assert lib.mvnerf_field_eval(one, one, one, one, ctypes.c_void_p(20), one, one, one, 1, 1, 4, 64, 8, 8, *fe_tail) == -3  # misaligned features
    assert lib.mvnerf_field_eval(one, one, one, one, one, one, one, one, 1, 1, 4, 64, 1, 8, *fe_tail) == -2              # H < 2
    fe_tail[6] = None
    assert lib.mvnerf_field_eval(one, one, one, one, one, one, one, one, 1, 1, 4, 64, 8, 8, *fe_tail) == -1              # no workspace
    assert lib.mvnerf_sample_pdf(one, one, one, 4, 33, 64, 0, one, None, None, None) == -2  # only 63 bins is built


def test_argument_validation_of_table_and_query_entry_points():
    lib = _lib.lib()
    one, odd = ctypes.c_void_p(16), ctypes.c_void_p(20)
    assert lib.mvnerf_texel_table_bytes(1, 3, 480, 640) == 3 * 480 * 640 * 128 * 4
    assert lib.mvnerf_texel_table_bytes(0, 1, 8, 8) == 0
    assert lib.mvnerf_project_texels(one, one, 1, 1, 8, 8, None, None) == -1                     # no table
    assert lib.mvnerf_project_texels(one, one, 1, 1, 1, 8, one, None) == -1                      # H < 2
    assert lib.mvnerf_project_texels(one, one, 1, 1, 8, 8, odd, None) == -3                      # misaligned table
    assert lib.mvnerf_project_texels2(one, one, one, 1, 1, 8, 8, one, None, None) == -1          # second net without its table
    assert lib.mvnerf_project_texels_bf16(one, one, None, 1, 1, 8, 8, one, one, None) == -1      # second table without its net
    fe_tail = [one, None, None, None, None, None, one, None]
    assert lib.mvnerf_field_eval_table(one, one, one, one, one, None, one, one, one, 1, 1, 4, 64, 8, 8, *fe_tail) == -1   # null table
    assert b'texel_table' in lib.mvnerf_last_error()
    assert lib.mvnerf_field_eval_table(one, one, one, one, one, odd, one, one, one, 1, 1, 4, 64, 8, 8, *fe_tail) == -3    # misaligned
    # render_fwd: S must be 64 whether or not tables are given
    assert lib.mvnerf_render_fwd(*([one] * 10), 1, 1, 4, 32, 8, 8, 0.3, 1.3, 0, one, one, one, one, one, one, 0, None) == -2
    # query points
    assert lib.mvnerf_query_workspace_bytes(2, 3, 10) == 2 * 2 * 3 * 10 * 128 * 4
    assert lib.mvnerf_query_vjp_scratch_bytes(1, 2, 64) == 3 * 2 * 2 * 4096 * 4
    q_tail = [None, one, one, None]                                    # acts, t_acts, workspace, stream
    assert lib.mvnerf_query_jvp(*([one] * 9), 1, 1, 0, 8, 8, *q_tail) == -1                       # N = 0
    assert lib.mvnerf_query_jvp(*([one] * 9), 1, 1, 8, 8, 1, *q_tail) == -2                       # W < 2
    assert lib.mvnerf_query_jvp(*([one] * 3), None, *([one] * 5), 1, 1, 8, 8, 8, *q_tail) == -1   # no t_dirs
    v_tail = [one, one, one, None]                                     # scratch, d_points, d_dirs, stream
    assert lib.mvnerf_query_vjp(*([one] * 9), 1, 2, 40, 8, 8, *v_tail) == -2                      # V > 1 needs N % 32 == 0
    assert lib.mvnerf_query_vjp(*([one] * 8), odd, 1, 1, 40, 8, 8, *v_tail) == -3                 # misaligned g_acts
    # training backward with the texel table: the table gradient scratch needs the table; both are checked for alignment
    fb = lambda table, tgrad: lib.mvnerf_field_backward_table(one, one, one, one, one, table, tgrad, one, one, one, one, one, one, one,
                                                              1, 1, 4, 64, 8, 8, one, one, None, one, None)
    assert fb(None, one) == -1 and b'texel_grad needs texel_table' in lib.mvnerf_last_error()
    assert fb(odd, None) == -3
    assert fb(one, odd) == -3
    # bf16 entry point: optional table / fused activations are checked for alignment
    b_tail = [one, None, None, odd, one, None]                         # rgbs, tap_idx, embedding, acts_fused, workspace, stream
    assert lib.mvnerf_field_eval_bf16(one, one, one, one, one, None, one, one, one, one, 1, 1, 4, 64, 8, 8, *b_tail) == -3


def test_train_step_entry_points_validate_their_structs():
    lib = _lib.lib()
    assert lib.mvnerf_train_workspace_bytes(1, 1, 0, 64, 8, 8, 0, 0) == 0
    small = lib.mvnerf_train_workspace_bytes(1, 1, 64, 64, 8, 8, 0, 0)
    assert 0 < small < lib.mvnerf_train_workspace_bytes(1, 1, 64, 64, 8, 8, 1, 0) < lib.mvnerf_train_workspace_bytes(1, 1, 64, 64, 8, 8, 1, 1)
    c = _lib.TrainCall()
    assert ctypes.sizeof(c) == 9 * 8 + 6 * 4 + 2 * 8 + 3 * 4 + 4 + 16 * 8 + 8 + 8         # the layout include/mvnerf_hip.h declares (LP64)
    assert lib.mvnerf_loss_and_grads(ctypes.byref(c), None) == -1 and b'null pointer' in lib.mvnerf_last_error()
    for name, _ in _lib.TrainCall._fields_:
        if _ is ctypes.c_void_p and name not in ('split_coarse', 'split_fine', 'd_features', 'fine_grad_event'):
            setattr(c, name, 256)
    c.B, c.V, c.R, c.S, c.H, c.W = 1, 1, 8, 32, 8, 8
    assert lib.mvnerf_loss_and_grads(ctypes.byref(c), None) == -2 and b'n_samples=64' in lib.mvnerf_last_error()
    c.S = 64
    c.workspace = 16
    assert lib.mvnerf_loss_and_grads(ctypes.byref(c), None) == -3
    c.workspace, c.workspace_bytes = 256, 1000
    assert lib.mvnerf_loss_and_grads(ctypes.byref(c), None) == -2 and b'workspace' in lib.mvnerf_last_error()
    c.split_coarse = 256
    assert lib.mvnerf_loss_and_grads(ctypes.byref(c), None) == -1 and b'both' in lib.mvnerf_last_error()
    a = _lib.AdamState()
    assert lib.mvnerf_apply_gradients(ctypes.byref(c), ctypes.byref(a), None) == -1            # no moments
    assert lib.mvnerf_train_step(None, None, None) == -1


def test_missing_library_fails_loudly(monkeypatch):
    import pytest
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libmvnerf_hip.so')
    with pytest.raises(RuntimeError, match='no fallback'):
        _lib.lib()


def test_round3_entry_points_validate_their_arguments():
    lib = _lib.lib()
    assert lib.mvnerf_set_split_kernel(7) == -1 and b'which=7' in lib.mvnerf_last_error()
    prev = lib.mvnerf_set_split_kernel(1)
    assert prev in (0, 1, 2) and lib.mvnerf_set_split_kernel(prev) == 1                 # returns the previous value
    one = ctypes.c_void_p(16)
    q = _lib.GemmTnBatch()
    assert ctypes.sizeof(q) == 4 * 8 + 4 * 8 + 5 * 4 + 4                                  # the layout include/mvnerf_hip.h declares (LP64, padded)
    assert lib.mvnerf_gemm_tn_batched(ctypes.byref(q), one, None, 64, 64, 64, 1, None, None) == -1        # null operands
    q.g, q.a, q.ldg, q.lda = 16, 16, 64, 64
    assert lib.mvnerf_gemm_tn_batched(ctypes.byref(q), one, None, 60, 64, 64, 1, None, None) == -2        # M % 8
    assert lib.mvnerf_gemm_tn_batched(ctypes.byref(q), one, None, 64, 64, 64, 0, None, None) == -2        # batch
    q.ldg = 32
    assert lib.mvnerf_gemm_tn_batched(ctypes.byref(q), one, None, 64, 64, 64, 1, None, None) == -2 and b'row strides' in lib.mvnerf_last_error()
    q.ldg, q.colsum_of = 64, 2
    assert lib.mvnerf_gemm_tn_batched(ctypes.byref(q), one, one, 64, 64, 64, 1, None, None) == -1         # column sums of an absent G2
    q.colsum_of = 1
    assert lib.mvnerf_gemm_tn_batched(ctypes.byref(q), one, None, 64, 64, 64, 1, None, None) == -1        # ... without an output
    assert lib.mvnerf_gemm_tn_batched_scratch_bytes(64512, 64, 128, 4, 1) > 0
    assert lib.mvnerf_gemm_tn_batched_scratch_bytes(60, 64, 128, 4, 1) == 0
