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
    # the comparison-build section of the header (#ifdef TMPNN_KEEP_VARIANTS) and its optional bindings agree too
    assert sorted(_lib._VARIANT_SIGNATURES) == sorted(set(_lib.header_symbols(variants=True)) - set(_lib.header_symbols()))


def test_shipped_library_exports_exactly_the_header(lib):
    """The boundary is frozen at ABI v4: libtmpnn.so exports the entry points include/tmpnn.h declares for the shipped
    library and nothing else (superseded forms live behind -DTMPNN_KEEP_VARIANTS; helpers other entry points call are
    hidden), and the count INTEGRATION.md / DESIGN.md quote is that number."""
    import re
    import subprocess
    if os.environ.get('TMPNN_LIB_PATH'):
        pytest.skip('a comparison build is loaded')
    out = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(m.group(1) for m in re.finditer(r' T (tmpnn_[a-z0-9_]+)$', out, flags=re.M))
    names = _lib.header_symbols()
    assert exported == names
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for doc in ('INTEGRATION.md', 'DESIGN.md'):
        txt = open(os.path.join(root, doc)).read()
        assert f'{len(names)} entry points' in txt, f'{doc} does not quote the current count ({len(names)} entry points)'


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
    # the det-side backward of the wide cells: unsupported width, inconsistent graph, short workspace
    rc = lib.tmpnn_wide_gru_bwd_diff(None, ctypes.byref(g), None, 64, 64, None, 0, None, 0, None, None, None, 64, None, None,
                                     None, None, None, 0, None)
    assert rc == -1 and b'wide_gru_bwd_diff' in lib.tmpnn_last_error()
    assert lib.tmpnn_wide_gru_bwd_diff_ws(100, 90, 10, 256) >= 4 * (100 * 4 * 256 + 10 * 3 * 256)
    assert lib.tmpnn_wide_gru_bwd_diff_ws(100, 90, 10, 64) == 0          # H = 64 is not a wide cell
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


def test_device_graph_arena_layout_matches_the_library():
    """trackmpnn_amd.graph.DeviceGraph restates tmpnn_dgraph_bind's pointer arithmetic in Python (one ctypes call
    less per forward); it must agree with the library for every capacity."""
    import ctypes as C
    from trackmpnn_amd import _lib
    lib = _lib.load()
    for cap in (0, 1, 3, 4, 5, 17, 375, 1700, 4095, 4096):
        ints = lib.tmpnn_dgraph_ints(cap)
        c = (cap + 1 + 3) & ~3
        nb = (((cap + 3) // 4) + 3) & ~3
        assert ints == 8 + nb + 10 * c
        st = _lib.CDGraph()
        base = 1 << 20
        assert lib.tmpnn_dgraph_bind(base, cap, cap, C.byref(st)) == 0
        o = base + 32 + 4 * nb
        assert (st.meta, st.is_edge, st.pos, st.src, st.dst, st.src_pos, st.dst_pos, st.edge_row, st.det_row, st.rowptr,
                st.inc) == (base, base + 32, o, o + 4 * c, o + 8 * c, o + 12 * c, o + 16 * c, o + 20 * c, o + 24 * c,
                            o + 28 * c, o + 32 * c)


def test_fused_iteration_buffer_sizes_match_the_library():
    """trackmpnn_amd.small restates the save / workspace size formulas of tmpnn_mp_iter_*; they must agree."""
    from trackmpnn_amd import _lib
    from trackmpnn_amd.small import bwd_ws_bytes, save_floats
    lib = _lib.load()
    for N, n in ((1, 0), (1, 1), (17, 5), (375, 60), (1700, 1700), (4096, 0), (4096, 4096)):
        for G in (1, 3):
            for H, IN_e in ((32, 32), (32, 64), (64, 64), (64, 128)):
                assert save_floats(N, n, G, H) == lib.tmpnn_mp_iter_save_floats(N, n, G, H)
                assert bwd_ws_bytes(N, n, G, H, IN_e) == lib.tmpnn_mp_iter_bwd_ws(N, n, G, H, IN_e)
