"""CPU-only: the C-ABI library loads, exports every symbol include/lerf_hip.h
declares, and its host-side helpers agree with the oracle / golden vectors.
No device entry point is exercised here."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import REPO

import lerf_pytorch_amd as L
from lerf_pytorch_amd import _lib


def _declared_functions():
    src = open(os.path.join(REPO, "include", "lerf_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lerf_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    names = _declared_functions()
    assert len(names) >= 14
    lib = ctypes.CDLL(L.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "liblerf_hip.so does not export %s" % n
    assert sorted(_lib.EXPORTS) == names
    assert lib.lerf_abi_version() == 7


def test_error_strings():
    lib = _lib.lib()
    assert lib.lerf_strerror(0) == b"ok"
    assert b"invalid" in lib.lerf_strerror(-1)


@pytest.mark.parametrize("mode", list("sctdy"))
def test_mode_offsets_match_oracle(oracle, mode):
    for r in range(4):
        dy, dx = _lib.mode_offsets(mode, r)
        assert list(zip(dy.tolist(), dx.tolist())) == [tuple(o) for o in oracle.rotated_offsets(mode, r)]


def test_unknown_mode_is_value_error():
    with pytest.raises(ValueError, match="Mode q not implemented."):
        _lib.mode_offsets("q", 0)


@pytest.mark.parametrize("n_in,scale,S", [(32, 2, 2), (40, 2, 4), (30, 1.5, 2), (17, 3, 2), (23, 3, 4), (16, 4, 2),
                                          (5, 2.4, 2), (6, 1.3, 2), (9, 1.0, 2), (1080, 2, 2), (1920, 2.0, 2),
                                          (1080, 1.5, 2), (333, 3.0, 4), (7, 7.77, 8)])
def test_sr_tables_bit_equal_oracle(oracle, n_in, scale, S):
    n_out = oracle.out_size(n_in, scale)
    assert _lib.out_size(n_in, scale) == n_out
    left, d64, d32, pads = _lib.sr_axis_tables(n_in, n_out, scale, S)
    ol, od, plo, phi = oracle.sr_axis_tables(n_in, n_out, scale, S)
    assert np.array_equal(left, ol)
    assert np.array_equal(d64, od)              # bit-equal float64
    assert pads == (plo, phi)
    # float32 table keeps the linear kernel's mask classes of the float64 distances
    cls = lambda x: np.where((x >= -1) & (x < 0), 1, np.where((x >= 0) & (x <= 1), 2, 0))
    assert np.array_equal(cls(d64), cls(d32.astype(np.float64)))
    assert np.max(np.abs(d32 - d64)) < 2.5e-7


@pytest.mark.parametrize("ci", range(8))
def test_sr_tables_match_reference_golden(golden, ci):
    g = golden("g23_sr.npz")
    H, W, sh, sw, S = g["gauss/%d/cfg" % ci]
    H, W, S = int(H), int(W), int(S)
    lx, dx, _, px = _lib.sr_axis_tables(H, _lib.out_size(H, sh), sh, S)
    ly, dy, _, py = _lib.sr_axis_tables(W, _lib.out_size(W, sw), sw, S)
    assert list(px) + list(py) == list(g["gauss/%d/pad" % ci])
    assert np.array_equal(dx, g["gauss/%d/disx" % ci])
    assert np.array_equal(dy, g["gauss/%d/disy" % ci])
    assert np.array_equal(lx[:, None] + px[0] + np.arange(S), g["gauss/%d/fovx" % ci])
    assert np.array_equal(ly[:, None] + py[0] + np.arange(S), g["gauss/%d/fovy" % ci])


@pytest.mark.parametrize("p", ["isc", "osc"])
def test_warp_pads_match_reference_golden(golden, p):
    g = golden("g4_warp.npz")
    minv = np.linalg.inv(g["%s/matrix" % p])
    for S in (2, 4):
        assert list(_lib.warp_pads(minv, (52, 52), (60, 70), S)) == list(g["%s/60x70/S%d/pad" % (p, S)])
    assert list(_lib.warp_pads(minv, (52, 52), (344, 228), 2)) == list(g["%s/344x228/S2/pad" % p])


def test_invert3x3():
    m = np.array([[2.05, 0.12, 15.0], [-0.08, 1.95, 40.0], [1.5e-5, -1.0e-5, 1.0]])
    out = np.zeros(9)
    assert _lib.lib().lerf_invert3x3(np.ascontiguousarray(m).ctypes.data, out.ctypes.data) == 0
    np.testing.assert_allclose(out.reshape(3, 3), np.linalg.inv(m), rtol=1e-12)
    sing = np.zeros(9)
    assert _lib.lib().lerf_invert3x3(sing.ctypes.data, out.ctypes.data) == -1


def test_device_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(L.LerfError, match="no CPU fallback"):
        L.LutSet.shipped("lerf-g")
    with pytest.raises(L.LerfError):
        L.sr(np.zeros((8, 8, 3), np.uint8), 2)


def test_lut_loader_shapes():
    for name, oC in (("lerf-g", 3), ("lerf-l", 1)):
        from lerf_pytorch_amd.luts import ASSET_DIR
        d = L.load_lut_arrays(os.path.join(ASSET_DIR, name), linear=(oC == 1))
        assert sorted(d) == sorted(["s1_%sr0" % m for m in "sct"] + ["s2_%sr%d" % (m, r) for m in "sct" for r in (0, 1)])
        assert d["s1_sr0"].shape == (83521, 1) and d["s2_tr1"].shape == (83521, oC)
        assert all(v.dtype == np.int8 for v in d.values())


# ---- property tests (hypothesis): host geometry == oracle for arbitrary sizes / scales / supports
from hypothesis import given, settings, strategies as st


@settings(max_examples=150, deadline=None)
@given(n_in=st.integers(1, 3000), scale=st.one_of(st.sampled_from([1.0, 1.5, 2.0, 2.4, 3.0, 4.0]),
                                                   st.floats(1.0, 8.0, allow_nan=False, allow_infinity=False)),
       S=st.integers(1, 8))
def test_sr_tables_property(n_in, scale, S):
    from oracle import lerf_oracle as O
    n_out = O.out_size(n_in, scale)
    assert _lib.out_size(n_in, scale) == n_out
    left, d64, d32, pads = _lib.sr_axis_tables(n_in, n_out, scale, S)
    ol, od, plo, phi = O.sr_axis_tables(n_in, n_out, scale, S)
    assert np.array_equal(left, ol) and np.array_equal(d64, od) and pads == (plo, phi)
    assert np.all(np.diff(left) >= 0)                       # monotone: what the tile / strip ownership search relies on
    cls = lambda x: np.where((x >= -1) & (x < 0), 1, np.where((x >= 0) & (x <= 1), 2, 0))
    assert np.array_equal(cls(d64), cls(d32.astype(np.float64)))


@settings(max_examples=60, deadline=None)
@given(H=st.integers(16, 400), world=st.integers(1, 8), S=st.sampled_from([2, 4]),
       scale=st.sampled_from([1.0, 1.5, 2.0, 2.4, 3.0, 4.0]))
def test_strip_plan_property(H, world, S, scale):
    from oracle import lerf_oracle as O
    from lerf_pytorch_amd import dist as ldist
    if H < world * ldist.halo_rows(S):
        return
    left, _, _, _ = O.sr_axis_tables(H, O.out_size(H, scale), scale, S)
    prev = 0
    for r in range(world):
        p = ldist.StripPlan(H, world, r, S, left)
        assert p.check_support(left)
        assert p.out_rows()[0] == prev
        prev = p.out_rows()[1]
    assert prev == len(left)


@pytest.mark.parametrize("n_in,scale,S", [(6, 3, 2), (17, 3, 4), (33, 1.7, 2), (1080, 2, 2), (10, 2.5, 4), (48, 4, 2), (7, 1.0, 2)])
def test_torch32_axis_tables_equal_oracle(oracle, n_in, scale, S):
    """lerf_sr_axis_tables_f32 == the float32 restatement of Resize2dTorch.get_distance (pinned to the reference's
    tensors when the goldens were generated: tests/golden/gen_golden.py g6 / g9 / g11 outputs depend on them)."""
    from lerf_pytorch_amd import _lib
    n_out = oracle.out_size(n_in, scale)
    left, dis64, dis32, pads = _lib.sr_axis_tables_f32(n_in, n_out, scale, S)
    rl, rd, plo, phi = oracle.sr_axis_tables_torch32(n_in, n_out, scale, S)
    assert np.array_equal(left, rl) and np.array_equal(dis64, rd) and pads == (plo, phi)
