"""The C-ABI library loads and exports every symbol include/tmpnn.h declares (no compute calls here)."""
import ctypes
import os

import pytest

from trackmpnn_amd import _lib


@pytest.fixture(scope='module')
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.load()


def test_header_symbols_exported(lib):
    names = _lib.header_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/tmpnn.h but not exported'


def test_bindings_cover_header(lib):
    assert sorted(_lib._SIGNATURES) == _lib.header_symbols()


def test_abi_version_and_error_string(lib):
    assert lib.tmpnn_abi_version() == _lib.ABI_VERSION
    assert isinstance(lib.tmpnn_last_error(), bytes)


def test_argument_validation_needs_no_gpu(lib):
    # bad arguments are rejected on the host before any launch
    rc = lib.tmpnn_gather_diff_fwd(None, None, 0, None, 0, 64, 0, None)
    assert rc == -1 and b'graph is null' in lib.tmpnn_last_error()
    g = _lib.CGraph(3, 1, 2, None, None, None, None, None, None)
    rc = lib.tmpnn_segsum_fwd(ctypes.byref(g), None, 64, None, 64, 48, 0, 0, None)
    assert rc == -1
    rc = lib.tmpnn_gru_fwd(None, 4, 1, None, None, None, 0, 0, 64, None, 64, 48, None, None, None, None, None, 64,
                           None, 0, None, None, 0, None)
    assert rc == -1 and b'unsupported H' in lib.tmpnn_last_error()
    assert lib.tmpnn_gru_bwd_weights_ws(1000, 64, 64) > 0
    assert lib.tmpnn_heads_bwd_ws(1000, 64) > 0
    assert lib.tmpnn_gru_fwd_head_parts(64, 64, 3) == 2 and lib.tmpnn_gru_fwd_head_parts(256, 256, 1) == 0


def test_cpu_tensors_fail_loudly():
    import torch
    from trackmpnn_amd import TrackMPNN
    m = TrackMPNN('2d', 3, 64, 0, 'diff')
    a = torch.eye(3)
    with pytest.raises(RuntimeError, match='HIP kernels only'):
        m(torch.zeros(3, 8), None, a, a)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(RuntimeError, match='no CPU or torch fallback'):
        _lib.load()
