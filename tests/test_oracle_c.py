"""Pin the C restatement (oracle/lerf_oracle.c) to the reference-generated
golden vectors and to the numpy oracle.  CPU-only."""
import hashlib
import json
import os

import numpy as np
import pytest
from PIL import Image

from conftest import DATA, GOLDEN


@pytest.fixture(scope="module")
def co():
    from oracle import c_oracle
    c_oracle.lib()
    return c_oracle


@pytest.mark.parametrize("name", ["noise24x20", "noise33x47", "baby64", "tiny5x6", "extremes8x8"])
@pytest.mark.parametrize("model", ["lerf-g", "lerf-l"])
def test_c_lut_stages_golden(co, golden, luts_g, luts_l, model, name):
    g = golden("g1_lut_stages.npz")
    luts, oC = (luts_g, 3) if model == "lerf-g" else (luts_l, 1)
    feat, hq = co.lut_stages(g["%s/%s/img" % (model, name)], luts, oC)
    assert np.array_equal(feat, g["%s/%s/feat" % (model, name)])
    assert np.array_equal(hq, g["%s/%s/hq" % (model, name)])


@pytest.mark.parametrize("ci", range(8))
def test_c_resize_golden(co, golden, ci):
    g = golden("g23_sr.npz")
    H, W, sh, sw, S = g["gauss/%d/cfg" % ci]
    feat = np.ascontiguousarray(g["gauss/%d/feat" % ci].transpose(1, 2, 0))
    hq = np.ascontiguousarray(g["gauss/%d/hq" % ci].transpose(2, 3, 1, 0))
    out = co.resize(feat, hq, sh, sw, int(S), 10, "gauss").transpose(2, 0, 1)
    assert np.max(np.abs(out - g["gauss/%d/out" % ci])) <= 1e-9
    if int(S) == 2:
        out = co.resize(feat, hq[..., :1], sh, sw, 2, 1, "linear").transpose(2, 0, 1)
        np.testing.assert_allclose(out, g["linear/%d/out" % ci], rtol=0, atol=1e-9)


def test_c_matches_numpy_oracle_random(co, oracle, luts_g):
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    f, h, o64, o8 = oracle.sr_pipeline(img, luts_g, 2.4, 1.3, return_all=True)
    cf, ch = co.lut_stages(img, luts_g, 3)
    assert np.array_equal(cf, f) and np.array_equal(ch, h)
    assert np.max(np.abs(co.resize(cf, ch, 2.4, 1.3) - o64)) <= 1e-10
    assert np.array_equal(co.sr_u8(img, luts_g, 2.4, 1.3), o8)


def test_c_set5_md5(co, luts_g, luts_l):
    ref = json.load(open(os.path.join(GOLDEN, "g5_set5.json")))["sr"]
    for model, luts, linear in (("lerf-g", luts_g, False), ("lerf-l", luts_l, True)):
        for scale in (2, 3, 4):
            for n in ("baby", "bird", "butterfly", "head", "woman"):
                lr = np.array(Image.open(os.path.join(DATA, "LR_bicubic/rrLR_X%.2f_%.2f" % (scale, scale), n + ".png")))
                o8 = co.sr_u8(lr, luts, scale, scale, linear=linear)
                assert hashlib.md5(o8.tobytes()).hexdigest() == ref["%s/x%d/%s" % (model, scale, n)]["md5_out"]


@pytest.mark.parametrize("p", ["isc", "osc"])
def test_c_warp_golden(co, golden, p):
    """lerf_oracle_warp against the reference's SteeringGaussian / AmplifiedLinear / Nearest warps (g4)."""
    g = golden("g4_warp.npz")
    M = g["%s/matrix" % p]
    feat = np.ascontiguousarray(g["%s/feat" % p].transpose(1, 2, 0))              # [H,W,C] uint8
    hq = np.ascontiguousarray(g["%s/hq" % p].transpose(2, 3, 1, 0))               # [H,W,C,oC]
    for S in (2, 4):
        assert list(co.warp_pads(M, feat.shape[:2], (60, 70), S)) == list(g["%s/60x70/S%d/pad" % (p, S)])
        out = co.warp(feat, hq, M, (60, 70), S, 10, "gauss").transpose(2, 0, 1)
        np.testing.assert_allclose(out, g["%s/60x70/S%d/gauss" % (p, S)], rtol=0, atol=1e-9, equal_nan=True)
    out = co.warp(feat, hq[..., :1], M, (60, 70), 2, 1, "linear").transpose(2, 0, 1)
    np.testing.assert_allclose(out, g["%s/60x70/linear" % p], rtol=0, atol=1e-9, equal_nan=True)
    out = co.warp(feat * 0 + 7, None, M, (60, 70), 1, 1, "nearest").transpose(2, 0, 1)
    assert np.array_equal(np.isnan(out), np.isnan(g["%s/60x70/nearest" % p]))
    out = co.warp(feat, hq, M, (344, 228), 2, 10, "gauss").transpose(2, 0, 1)
    np.testing.assert_allclose(out, g["%s/344x228/S2/gauss_f32" % p], rtol=1e-6, atol=1e-4, equal_nan=True)


@pytest.mark.parametrize("p", ["isc", "osc"])
def test_c_set5_warp_md5(co, luts_g, luts_l, p):
    """lerf_oracle_warp_u8 reproduces the reference's masked uint8 outputs and masks of eval_lut_warp.py (g5)."""
    ref = json.load(open(os.path.join(GOLDEN, "g5_set5.json")))["warp"]
    for model, luts, linear in (("lerf-g", luts_g, False), ("lerf-l", luts_l, True)):
        for n in ("baby", "bird", "butterfly", "head", "woman"):
            r = ref["%s/%s/%s" % (model, p, n)]
            lr = np.array(Image.open(os.path.join(DATA, p, n + ".png")))
            o8, mask = co.warp_u8(lr, luts, np.array(r["matrix"]), r["shape"][:2], linear=linear)
            assert int(mask.sum()) == r["mask_sum"]
            assert hashlib.md5(mask.astype(np.uint8).tobytes()).hexdigest() == r["md5_mask"]
            assert hashlib.md5((o8 * mask).tobytes()).hexdigest() == r["md5_out_masked"]


def test_c_warp_matches_numpy_oracle_random(co, oracle, luts_g):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (41, 57, 3), dtype=np.uint8)
    M = np.array([[2.05, 0.12, 15.0], [-0.08, 1.95, 40.0], [1.5e-5, -1.0e-5, 1.0]])
    f, h, o64, o8 = oracle.warp_pipeline(img, luts_g, M, (90, 120), return_all=True)
    c64 = co.warp(f, h, M, (90, 120))
    assert np.array_equal(np.isnan(c64), np.isnan(o64))
    assert np.nanmax(np.abs(c64 - o64)) <= 1e-10
    c8, cm = co.warp_u8(img, luts_g, M, (90, 120))
    assert np.array_equal(c8, o8)
    assert np.array_equal(cm, oracle.warp_mask(img.shape[:2], M, (90, 120)))
