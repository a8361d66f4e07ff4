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
