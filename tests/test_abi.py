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
    assert lib.mvnerf_packed_net_floats() == 251144
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


def test_missing_library_fails_loudly(monkeypatch):
    import pytest
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libmvnerf_hip.so')
    with pytest.raises(RuntimeError, match='no fallback'):
        _lib.lib()
