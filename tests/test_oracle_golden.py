"""Pin the CPU oracle (oracle/lerf_oracle.py) to vectors produced by the
reference itself (tests/golden/gen_golden.py).  CPU-only."""
import hashlib
import json
import os

import numpy as np
import pytest
from PIL import Image

from conftest import DATA, GOLDEN

G1_INPUTS = ["noise24x20", "noise33x47", "baby64", "tiny5x6", "extremes8x8"]


@pytest.mark.parametrize("name", G1_INPUTS)
@pytest.mark.parametrize("model", ["lerf-g", "lerf-l"])
def test_lut_stages_bit_exact(oracle, golden, luts_g, luts_l, model, name):
    g = golden("g1_lut_stages.npz")
    luts, oC = (luts_g, 3) if model == "lerf-g" else (luts_l, 1)
    img = g["%s/%s/img" % (model, name)]
    feat, hq = oracle.lut_stages(img, luts, oC)
    assert np.array_equal(feat, g["%s/%s/feat" % (model, name)])
    assert np.array_equal(hq, g["%s/%s/hq" % (model, name)])


def test_single_lut_passes_bit_exact(oracle, golden, luts_g):
    """per-(mode, rotation) raw numerators (x16) == FourSimplexInterpFaster output x16."""
    g = golden("g1_lut_stages.npz")
    img = g["lerf-g/noise24x20/img"]
    feat = g["lerf-g/noise24x20/feat"]
    for mode in "sct":
        for r in range(4):
            got = oracle.lut_interp_numer(luts_g["s1_%sr0" % mode], img, mode, r)
            assert np.array_equal(got, g["lerf-g/noise24x20/raw/s1_%s_r%d" % (mode, r)].astype(np.int32))
            got = oracle.lut_interp_numer(luts_g["s2_%sr%d" % (mode, r & 1)], feat, mode, r)
            assert np.array_equal(got, g["lerf-g/noise24x20/raw/s2_%s_r%d" % (mode, r)].astype(np.int32))


def test_unknown_mode_raises(oracle, luts_g):
    with pytest.raises(ValueError, match="not implemented"):
        oracle.lut_interp_numer(luts_g["s1_sr0"], np.zeros((4, 4, 3), np.uint8), "q", 0)


@pytest.mark.parametrize("ci", range(8))
def test_sr_geometry_tables(oracle, golden, ci):
    g = golden("g23_sr.npz")
    H, W, sh, sw, S = g["gauss/%d/cfg" % ci]
    H, W, S = int(H), int(W), int(S)
    oH, oW = oracle.out_size(H, sh), oracle.out_size(W, sw)
    lx, dx, plx, phx = oracle.sr_axis_tables(H, oH, sh, S)
    ly, dy, ply, phy = oracle.sr_axis_tables(W, oW, sw, S)
    assert [plx, phx, ply, phy] == list(g["gauss/%d/pad" % ci])
    # reference fov is in padded coordinates: left + pad + ordinal
    assert np.array_equal(lx[:, None] + plx + np.arange(S)[None], g["gauss/%d/fovx" % ci])
    assert np.array_equal(ly[:, None] + ply + np.arange(S)[None], g["gauss/%d/fovy" % ci])
    assert np.array_equal(dx, g["gauss/%d/disx" % ci])      # bit-equal float64
    assert np.array_equal(dy, g["gauss/%d/disy" % ci])


@pytest.mark.parametrize("ci", range(8))
def test_sr_gauss_float64(oracle, golden, ci):
    g = golden("g23_sr.npz")
    H, W, sh, sw, S = g["gauss/%d/cfg" % ci]
    feat = g["gauss/%d/feat" % ci].astype(np.float32)
    h = g["gauss/%d/hq" % ci].astype(np.float32) / np.float32(255)
    out = oracle.resize_params_f32(feat, h[0], h[1], h[2], sh, sw, int(S), 10, "gauss")
    ref = g["gauss/%d/out" % ci]
    assert out.shape == ref.shape
    assert np.max(np.abs(out - ref)) <= 1e-9


@pytest.mark.parametrize("ci", [0, 2, 3, 4, 5, 6])
def test_sr_linear_float64(oracle, golden, ci):
    g = golden("g23_sr.npz")
    H, W, sh, sw, S = g["gauss/%d/cfg" % ci]
    feat = g["gauss/%d/feat" % ci].astype(np.float32)
    h = g["gauss/%d/hq" % ci].astype(np.float32) / np.float32(255)
    out = oracle.resize_params_f32(feat, h[0], None, None, sh, sw, 2, 1, "linear")
    ref = g["linear/%d/out" % ci]
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-9, equal_nan=True)


@pytest.mark.parametrize("p", ["isc", "osc"])
def test_warp_float64(oracle, golden, p):
    g = golden("g4_warp.npz")
    M = g["%s/matrix" % p]
    feat = g["%s/feat" % p].astype(np.float32)
    h = g["%s/hq" % p].astype(np.float32) / np.float32(255)
    for S in (2, 4):
        geo = oracle.warp_geometry(M, feat.shape[1:], (60, 70), S)
        assert list(geo["pad"]) == list(g["%s/60x70/S%d/pad" % (p, S)])
        out = oracle.warp_params_f32(feat, h[0], h[1], h[2], M, (60, 70), S, 10, "gauss")
        np.testing.assert_allclose(out, g["%s/60x70/S%d/gauss" % (p, S)], rtol=0, atol=1e-9, equal_nan=True)
    out = oracle.warp_params_f32(feat, h[0], None, None, M, (60, 70), 2, 1, "linear")
    np.testing.assert_allclose(out, g["%s/60x70/linear" % p], rtol=0, atol=1e-9, equal_nan=True)
    out = oracle.warp_params_f32(g["%s/feat" % p].astype(np.float32) * 0 + 7, None, None, None, M, (60, 70), 1, 1, "nearest")
    assert np.array_equal(np.isnan(out), np.isnan(g["%s/60x70/nearest" % p]))
    for hw in ((60, 70), (344, 228)):
        m = oracle.warp_mask(feat.shape[1:], M, hw)
        assert np.array_equal(m.transpose(2, 0, 1), g["%s/%dx%d/mask" % (p, hw[0], hw[1])])
    out = oracle.warp_params_f32(feat, h[0], h[1], h[2], M, (344, 228), 2, 10, "gauss")
    np.testing.assert_allclose(out, g["%s/344x228/S2/gauss_f32" % p], rtol=1e-6, atol=1e-4, equal_nan=True)


def _md5(a):
    return hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()


SET5 = ["baby", "bird", "butterfly", "head", "woman"]


@pytest.mark.parametrize("scale", [2, 3, 4])
@pytest.mark.parametrize("model", ["lerf-g", "lerf-l"])
def test_set5_sr_known_answers(oracle, luts_g, luts_l, model, scale):
    """md5 of the reference's uint8 outputs + the scripts.sh PSNR table."""
    ref = json.load(open(os.path.join(GOLDEN, "g5_set5.json")))["sr"]
    luts, linear = (luts_g, False) if model == "lerf-g" else (luts_l, True)
    published = {"lerf-g": {2: 35.71, 3: 32.02, 4: 30.15}, "lerf-l": {2: 34.84, 3: 30.72, 4: 29.13}}
    names = SET5 if scale == 2 else ["butterfly", "woman"]     # keep the CPU suite short
    ps = []
    for n in names:
        lr = np.array(Image.open(os.path.join(DATA, "LR_bicubic/rrLR_X%.2f_%.2f" % (scale, scale), n + ".png")))
        gt = np.array(Image.open(os.path.join(DATA, "HR", n + ".png")))
        feat, hq, out, o8 = oracle.sr_pipeline(lr, luts, scale, scale, linear=linear, return_all=True)
        r = ref["%s/x%d/%s" % (model, scale, n)]
        assert _md5(feat) == r["md5_feat"]
        assert _md5(np.transpose(hq, (2, 3, 0, 1))) == r["md5_hq"]      # reference layout [C*oC,H,W]
        assert _md5(o8) == r["md5_out"]
        p = oracle.psnr_y(gt, o8, scale)
        assert abs(p - r["psnr_y"]) < 1e-4
        ps.append(p)
    if scale == 2:
        assert "%.2f" % np.mean(ps) == "%.2f" % published[model][scale]


@pytest.mark.parametrize("p", ["isc", "osc"])
def test_set5_warp_known_answers(oracle, luts_g, p):
    ref = json.load(open(os.path.join(GOLDEN, "g5_set5.json")))["warp"]
    published = {"isc": 33.81, "osc": 27.89}
    ms = []
    for n in SET5:
        r = ref["lerf-g/%s/%s" % (p, n)]
        lr = np.array(Image.open(os.path.join(DATA, p, n + ".png")))
        gt = np.array(Image.open(os.path.join(DATA, "HR", n + ".png")))
        M = np.array(r["matrix"])
        o8 = oracle.warp_pipeline(lr, luts_g, M, gt.shape[:2])
        mask = oracle.warp_mask(lr.shape[:2], M, gt.shape[:2])
        assert int(mask.sum()) == r["mask_sum"]
        assert _md5(mask.astype(np.uint8)) == r["md5_mask"]
        assert _md5(o8 * mask) == r["md5_out_masked"]
        m = oracle.mpsnr(o8, gt, mask)
        assert abs(m - r["mpsnr"]) < 1e-3
        ms.append(m)
    assert "%.2f" % np.mean(ms) == "%.2f" % published[p]


@pytest.mark.parametrize("p", ["isc", "osc"])
@pytest.mark.parametrize("kind,S", [("cubic", 4), ("bilinear", 2), ("lanczos2", 4), ("lanczos3", 6)])
def test_fixed_kernel_warps(oracle, golden, p, kind, S):
    g4, g7 = golden("g4_warp.npz"), golden("g7_fixed_warp.npz")
    feat = g4["%s/feat" % p].astype(np.float32)
    out = oracle.warp_params_f32(feat, None, None, None, g4["%s/matrix" % p], (60, 70), S, 1, kind)
    np.testing.assert_allclose(out, g7["%s/%s" % (p, kind)], rtol=0, atol=1e-9, equal_nan=True)


@pytest.mark.parametrize("model,scale", [("lerf-g", 4), ("lerf-l", 3)])
def test_ssim_known_answers(oracle, luts_g, luts_l, model, scale):
    """cal_ssim of the reference (g8_ssim.json) on the md5-pinned Set5 outputs; and the published SSIM column."""
    from oracle import c_oracle
    ref = json.load(open(os.path.join(GOLDEN, "g8_ssim.json")))
    luts, linear = (luts_g, False) if model == "lerf-g" else (luts_l, True)
    for n in ("butterfly", "woman"):
        lr = np.array(Image.open(os.path.join(DATA, "LR_bicubic/rrLR_X%.2f_%.2f" % (scale, scale), n + ".png")))
        gt = np.array(Image.open(os.path.join(DATA, "HR", n + ".png")))
        o8 = c_oracle.sr_u8(lr, luts, scale, scale, linear=linear)
        r = ref["%s/x%d/%s" % (model, scale, n)]
        assert abs(oracle.ssim_y(gt, o8) - r["ssim"]) < 1e-9
        assert abs(oracle.psnr_y(gt, o8, scale) - r["psnr_y"]) < 1e-4
    published = {"lerf-g": ["35.71/0.9475", "32.02/0.8980", "30.15/0.8548"],        # scripts.sh:36-41
                 "lerf-l": ["34.84/0.9432", "30.72/0.8773", "29.13/0.8270"]}
    for m in published:
        assert [ref["%s/x%d/mean" % (m, s)] for s in (2, 3, 4)] == published[m]


@pytest.mark.parametrize("ci", range(5))
def test_bicubic_resize_torch_golden(oracle, golden, ci):
    """BicubicResize2dTorch of the reference (float32 geometry and arithmetic) vs the float64 restatement."""
    g = golden("g9_bicubic_resize.npz")
    x = g["%d/x" % ci].astype(np.float32)
    s = g["%d/scale" % ci]
    B, C, H, W = x.shape
    o = oracle.resize_params_f32(x.reshape(B * C, H, W), None, None, None, s[0], s[1], 4, 1, "cubic")
    ref = g["%d/out" % ci]
    assert o.reshape(ref.shape).shape == ref.shape
    assert np.max(np.abs(o.reshape(ref.shape) - ref)) <= 2e-3


@pytest.mark.parametrize("mi,mode", list(enumerate("sdyct")))
@pytest.mark.parametrize("model", ["lerf-g", "lerf-l"])
def test_swf2lut_interp_restatement(oracle, golden, luts_g, luts_l, model, mi, mode):
    """SWF2LUT.InterpTorchBatch of the reference (forward + autograd gradients) vs oracle.swf2lut_interp."""
    import sys
    sys.path.insert(0, GOLDEN)
    import swf_inputs
    g = golden("g10_swf2lut.npz")
    luts, outC = (luts_g, 3) if model == "lerf-g" else (luts_l, 1)
    key = "s2_%sr0" % (mode if mode in "sct" else "s")
    bd, img, G = swf_inputs.case_inputs(1000 + mi, mode, outC)
    w = swf_inputs.case_weight(luts[key].astype(np.float32).reshape(-1, outC) / np.float32(127.0), 2000 + mi)
    out, gw, gimg = oracle.swf2lut_interp(w, outC, mode, img, bd, G)
    pre = "%s/interp/%s/" % (model, mode)
    assert np.array_equal(out, g[pre + "out"])
    assert np.max(np.abs(gimg - g[pre + "gimg"])) <= 2e-5
    rows = g[pre + "gw_rows"]
    assert np.array_equal(np.nonzero(np.abs(gw).sum(1))[0], rows)
    assert np.max(np.abs(gw[rows] - g[pre + "gw_vals"])) <= 2e-5 * np.abs(g[pre + "gw_vals"]).max()


@pytest.mark.parametrize("ci", range(6))
def test_downscale_antialias_golden(oracle, golden, ci):
    """scale < 1 through the reference's numpy classes (anti-aliasing keyed on the ROW factor only, :51-55, 186-193)."""
    g = golden("g12_downscale.npz")
    Cn, H, W, sh, sw, S, S2 = g["%d/cfg" % ci]
    feat = g["%d/feat" % ci].astype(np.float32)
    hy = g["%d/hq" % ci].astype(np.float32) / np.float32(255)
    o = oracle.resize_params_f32(feat, hy[0], hy[1], hy[2], sh, sw, int(S), 10, "gauss")
    assert np.max(np.abs(o - g["%d/gauss" % ci])) <= 1e-9
    if "%d/linear" % ci in g:
        o = oracle.resize_params_f32(feat, hy[0], None, None, sh, sw, 2, 1, "linear")
        np.testing.assert_allclose(o, g["%d/linear" % ci], rtol=0, atol=1e-9, equal_nan=True)


@pytest.mark.parametrize("p", ["isc", "osc"])
def test_torch_warp_classes_golden(oracle, golden, p):
    """the reference's TORCH warp classes (g13: SteeringGaussian S=2/4, AmplifiedLinear, Nearest, Bicubic; float64
    geometry) agree with the numpy restatement to float64 rounding"""
    g4, g13 = golden("g4_warp.npz"), golden("g13_torch_warp.npz")
    M = g4["%s/matrix" % p]
    feat = g4["%s/feat" % p][:2].astype(np.float32)
    h = g4["%s/hq" % p][:, :2].astype(np.float32) / np.float32(255)
    for hw in ((60, 70), (97, 41)):
        key = "%s/%dx%d" % (p, hw[0], hw[1])
        for S in (2, 4):
            geo = oracle.warp_geometry(M, (52, 52), hw, S)
            plx, phx, ply, phy = geo["pad"]
            assert [ply, phy, plx, phx] == list(g13[key + "/pad_S%d" % S])          # torch order: last dim first
            out = oracle.warp_params_f32(feat, h[0], h[1], h[2], M, hw, S, 10, "gauss")
            np.testing.assert_allclose(out, g13[key + "/gauss_S%d" % S][:, 0], rtol=0, atol=1e-9, equal_nan=True)
        out = oracle.warp_params_f32(feat, h[0], None, None, M, hw, 2, 1, "linear")
        np.testing.assert_allclose(out, g13[key + "/linear"][:, 0], rtol=0, atol=1e-9, equal_nan=True)
        out = oracle.warp_params_f32(feat, None, None, None, M, hw, 1, 1, "nearest")
        np.testing.assert_allclose(out, g13[key + "/nearest"][:, 0], rtol=0, atol=0, equal_nan=True)
        out = oracle.warp_params_f32(feat, None, None, None, M, hw, 4, 1, "cubic")
        np.testing.assert_allclose(out, g13[key + "/cubic"][:, 0], rtol=0, atol=1e-9, equal_nan=True)


@pytest.mark.parametrize("model,keys", [("lerf-g", ["s1_sr0", "s2_cr1", "s2_tr0"]), ("lerf-l", ["s1_tr0", "s2_sr1"])])
def test_net_to_lut_transfer_golden(oracle, model, keys):
    """the float64 restatement of SRNet + the transfer enumeration against the reference's own transfer of its shipped
    checkpoint (g14): identical int8 LUTs except entries whose value sits within 1e-3 LUT units of a rounding boundary
    (the reference evaluates in float32), never more than one step, a handful per LUT"""
    from conftest import ASSETS
    w = dict(np.load(os.path.join(ASSETS, model, "srnets_weights.npz")))
    g = np.load(os.path.join(GOLDEN, "g14_transfer_%s.npz" % model))
    assert np.array_equal(oracle.transfer_inputs(4)[[0, 1, 17, 83520]],
                          np.array([[0, 0, 0, 0], [0, 0, 0, 16], [0, 0, 16, 0], [255, 255, 255, 255]], np.float32) / np.float32(255))
    for key in keys:
        lut, y = oracle.transfer_lut(w, key, return_float=True)
        ref = g[key]
        assert lut.shape == ref.shape
        assert np.max(np.abs(y[g["probe_index"]] - g[key + "/f32"])) <= 1e-5
        d = lut.astype(int) - ref.astype(int)
        bad = d != 0
        assert bad.sum() <= 8 and np.abs(d).max() <= 1
        if bad.any():
            amb = np.abs((y * 127) - np.floor(y * 127) - 0.5)
            assert amb[bad].max() < 1e-3


def test_pad_modes_golden(oracle, golden):
    """non-default pad_mode of the reference's classes (g15): image operand padded with np.pad's edge / reflect /
    symmetric / wrap, hyper maps edge-padded as always; numpy SR (S = 2, 4, linear), numpy warp, torch SR."""
    g, g4 = golden("g15_pad_modes.npz"), golden("g4_warp.npz")
    feat = g["feat"].astype(np.float32)
    h = g["hq"].astype(np.float32) / np.float32(255)
    for mode in ("edge", "reflect", "symmetric", "wrap"):
        for S, sc in ((2, (2.0, 3.0)), (4, (1.5, 2.0))):
            o = oracle.resize_params_f32(feat, h[0], h[1], h[2], sc[0], sc[1], S, 10, "gauss", pad_mode=mode)
            np.testing.assert_allclose(o, g["sr/%s/gauss_S%d" % (mode, S)], rtol=0, atol=1e-9)
        o = oracle.resize_params_f32(feat, h[0], None, None, 3.0, 2.0, 2, 1, "linear", pad_mode=mode)
        np.testing.assert_allclose(o, g["sr/%s/linear" % mode], rtol=0, atol=1e-9, equal_nan=True)
        for p in ("isc", "osc"):
            f52 = g4["%s/feat" % p].astype(np.float32)
            h52 = g4["%s/hq" % p].astype(np.float32) / np.float32(255)
            o = oracle.warp_params_f32(f52, h52[0], h52[1], h52[2], g4["%s/matrix" % p], (60, 70), 2, 10, "gauss", pad_mode=mode)
            np.testing.assert_allclose(o, g["warp/%s/%s" % (mode, p)], rtol=0, atol=1e-9, equal_nan=True)
    for mode in ("replicate", "reflect", "circular"):
        o = oracle.resize_params_f32(feat, h[0], h[1], h[2], 2.0, 2.0, 2, 10, "gauss", geometry="torch32", pad_mode=mode)
        assert np.abs(o - g["torch/%s/gauss" % mode][0]).max() <= 5e-4          # the reference's own float32 arithmetic


@pytest.mark.parametrize("interval", [3, 5, 6, 7])
def test_lut_interp_other_intervals(oracle, golden, interval):
    """the LUT pass at sampling intervals other than the shipped 4 (g17, the reference's FourSimplexInterpFaster)"""
    g = golden("g17_intervals.npz")
    img = g["img"]
    lut = g["lut/%d" % interval]
    for mode in "sct":
        for r in (0, 3):
            num = oracle.lut_interp_numer(lut, img, mode, r, interval)            # [H,W,C,oC], value * 2^interval
            ref = g["out/%d/%s/%d" % (interval, mode, r)]                          # [C*oC, H, W], already rotated back
            mine = num.transpose(2, 3, 0, 1).reshape(ref.shape) / float(2 ** interval)
            assert np.array_equal(mine, ref)


@pytest.mark.parametrize("mode", ["d", "y"])
def test_lut_interp_modes_d_y(oracle, golden, luts_g, mode):
    """sampling patterns 'd' and 'y' (no shipped model uses them) against the reference's FourSimplexInterpFaster (g18)"""
    g = golden("g18_modes_dy.npz")
    for key in ("s1_sr0", "s2_tr1"):
        for r in range(4):
            num = oracle.lut_interp_numer(luts_g[key], g["img"], mode, r)
            ref = g["%s/%s/%d" % (mode, key, r)]
            assert np.array_equal(num.transpose(2, 3, 0, 1).reshape(ref.shape) / 16.0, ref)
