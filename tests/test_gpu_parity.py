"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and
the reference-generated golden vectors.  Run with `-m gpu` on an MI355X.

Tolerances
  * LUT stages (integer):            bit-exact
  * stage 3, float64 outputs:        <= 1e-9 absolute (0..255 scale)
  * stage 3, float32 outputs:        <= 2.55e-2 absolute (= 1e-4 of the 255 range, north_star's fp32 bound);
                                     the observed maximum is also asserted to stay below 5e-4
  * uint8 outputs:                   the Set5 md5s of the reference must match exactly (SR and warp)
"""
import hashlib
import json
import os

import numpy as np
import pytest
from PIL import Image

from conftest import DATA, GOLDEN

pytestmark = pytest.mark.gpu

F32_TOL = 2.55e-2
F32_OBSERVED = 5e-4
# SR uint8 outputs: float32 arithmetic + a float64 re-evaluation of values within 3e-4 of a half-integer
# (tie guard, csrc/lerf_stage3.h) -> byte-identical to the reference on every input tested.
# Warp uint8 outputs carry the same guard (warp_tie_guard) and are byte-identical on the valid region as well.


@pytest.fixture(scope="module")
def torch():
    import torch as t
    assert t.cuda.is_available(), "GPU tests need an MI355X"
    return t


@pytest.fixture(scope="module")
def eng_g(torch):
    import lerf_pytorch_amd as L
    return L.LerfEngine.shipped("lerf-g")


@pytest.fixture(scope="module")
def eng_l(torch):
    import lerf_pytorch_amd as L
    return L.LerfEngine.shipped("lerf-l")


def _md5(a):
    return hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()


G1_INPUTS = ["noise24x20", "noise33x47", "baby64", "tiny5x6", "extremes8x8"]


@pytest.mark.parametrize("name", G1_INPUTS)
@pytest.mark.parametrize("model", ["lerf-g", "lerf-l"])
def test_lut_stages_golden_bit_exact(golden, eng_g, eng_l, model, name):
    g = golden("g1_lut_stages.npz")
    eng = eng_g if model == "lerf-g" else eng_l
    feat, hq = eng.stages(g["%s/%s/img" % (model, name)])
    assert np.array_equal(feat, g["%s/%s/feat" % (model, name)])
    assert np.array_equal(hq, g["%s/%s/hq" % (model, name)])


@pytest.mark.parametrize("shape", [(1, 1, 3), (2, 3, 3), (7, 1, 3), (64, 64, 1), (97, 131, 3), (256, 256, 3)])
def test_lut_stages_random_vs_oracle(oracle, luts_g, luts_l, eng_g, eng_l, shape):
    rng = np.random.default_rng(sum(shape))
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    for eng, luts, oC in ((eng_g, luts_g, 3), (eng_l, luts_l, 1)):
        feat, hq = eng.stages(img)
        of, oh = oracle.lut_stages(img, luts, oC)
        assert np.array_equal(feat, of)
        assert np.array_equal(hq, oh)


def test_four_simplex_interp_mirror(golden, luts_g):
    """FourSimplexInterpFaster drop-in == the reference's raw per-pass outputs."""
    from lerf_pytorch_amd.resample.eval_lut_sr import FourSimplexInterpFaster, mode_pad_dict
    g = golden("g1_lut_stages.npz")
    img = g["lerf-g/noise24x20/img"].astype(np.float32)
    feat = g["lerf-g/noise24x20/feat"].astype(np.float32)
    for stage, src in ((1, img), (2, feat)):
        for mode in "sct":
            pad = mode_pad_dict[mode]
            for r in range(4):
                key = "s1_%sr0" % mode if stage == 1 else "s2_%sr%d" % (mode, r & 1)
                oC = 1 if stage == 1 else 3
                rot = np.rot90(src, r)
                h, w, _ = rot.shape
                img_in = np.pad(rot, ((0, pad), (0, pad), (0, 0)), mode="edge").transpose((2, 0, 1))
                out = FourSimplexInterpFaster(luts_g[key].astype(np.float32), img_in, h, w, 4, 4 - r,
                                              upscale=1, mode=mode, oC=oC)
                assert out.dtype == np.float64 and out.shape == (3 * oC, 24, 20)
                ref = g["lerf-g/noise24x20/raw/s%d_%s_r%d" % (stage, mode, r)]      # [H,W,C,oC] x16
                ref = ref.transpose(2, 3, 0, 1).reshape(3 * oC, 24, 20) / 16.0
                assert np.array_equal(out, ref)
    with pytest.raises(ValueError, match="Mode q not implemented."):
        FourSimplexInterpFaster(luts_g["s1_sr0"], np.zeros((3, 5, 5), np.float32), 4, 4, 4, 0, mode="q")


@pytest.mark.parametrize("ci", range(8))
def test_sr_gauss_numpy_class_vs_golden(golden, ci):
    from lerf_pytorch_amd.resize_right.resize_right2d_numpy import SteeringGaussianResize2dNumpy
    g = golden("g23_sr.npz")
    H, W, sh, sw, S = g["gauss/%d/cfg" % ci]
    feat = g["gauss/%d/feat" % ci].astype(np.float32)
    h = g["gauss/%d/hq" % ci].astype(np.float32) / np.float32(255)
    r = SteeringGaussianResize2dNumpy(support_sz=int(S), max_sigma=10)
    r.set_shape([3, int(H), int(W)], scale_factors=[sh, sw])
    assert [list(p) for p in r.pad_vec[1:]] == [list(g["gauss/%d/pad" % ci][:2]), list(g["gauss/%d/pad" % ci][2:])]
    out = r.resize(feat, h[0], h[1], h[2])
    ref = g["gauss/%d/out" % ci]
    assert out.dtype == np.float64 and out.shape == ref.shape
    assert np.max(np.abs(out - ref)) <= 1e-9


@pytest.mark.parametrize("ci", [0, 2, 3, 4, 5, 6])
def test_sr_linear_numpy_class_vs_golden(golden, ci):
    from lerf_pytorch_amd.resize_right.resize_right2d_numpy import AmplifiedLinearResize2dNumpy
    g = golden("g23_sr.npz")
    H, W, sh, sw, S = g["gauss/%d/cfg" % ci]
    feat = g["gauss/%d/feat" % ci].astype(np.float32)
    h = g["gauss/%d/hq" % ci].astype(np.float32) / np.float32(255)
    r = AmplifiedLinearResize2dNumpy()
    r.set_shape([3, int(H), int(W)], scale_factors=[sh, sw])
    out = r.resize(feat, h[0])
    np.testing.assert_allclose(out, g["linear/%d/out" % ci], rtol=0, atol=1e-9, equal_nan=True)


@pytest.mark.parametrize("ci", range(8))
@pytest.mark.parametrize("kind", ["gauss", "linear"])
def test_sr_float32_and_uint8_vs_golden(torch, golden, ci, kind):
    """the uint8-input production kernels (float32 arithmetic) against the float64 reference."""
    from lerf_pytorch_amd import ops
    g = golden("g23_sr.npz")
    H, W, sh, sw, S = g["gauss/%d/cfg" % ci]
    if kind == "linear" and int(S) != 2:
        pytest.skip("LeRF-L is S=2 only")
    feat = torch.from_numpy(g["gauss/%d/feat" % ci].transpose(1, 2, 0).copy()).cuda()            # HWC
    hq = torch.from_numpy(g["gauss/%d/hq" % ci].transpose(2, 3, 1, 0).copy()).cuda()             # [H,W,C,k]
    geo = ops.SrGeometry((int(H), int(W)), [sh, sw], None, int(S))
    ms = 10 if kind == "gauss" else 1
    ref = (g["gauss/%d/out" % ci] if kind == "gauss" else g["linear/%d/out" % ci]).transpose(1, 2, 0)
    o32 = ops.resize_hwc_u8(feat, hq, geo, kind, ms, out="f32").cpu().numpy()
    err = np.max(np.abs(o32 - ref))
    assert err <= F32_TOL and err <= F32_OBSERVED, err
    o8 = ops.resize_hwc_u8(feat, hq, geo, kind, ms, out="u8").cpu().numpy()
    r8 = np.clip(np.round(ref), 0, 255).astype(np.uint8)
    assert np.max(np.abs(o8.astype(int) - r8.astype(int))) <= 1
    o64 = ops.resize_hwc_u8(feat, hq, geo, kind, ms, out="f64").cpu().numpy()
    assert np.max(np.abs(o64 - ref)) <= 1e-9


def test_sr_torch_class_vs_reference_torch_path(torch, golden):
    """secondary oracle: fp32 outputs of the reference's SteeringGaussianResize2dTorch."""
    from lerf_pytorch_amd.resize_right.resize_right2d_torch import SteeringGaussianResize2dTorch
    g = golden("g6_torch.npz")
    for ci in range(3):
        H, W, s = g["%d/cfg" % ci]
        feat = torch.from_numpy(g["%d/feat" % ci].astype(np.float32)).cuda()
        h = torch.from_numpy(g["%d/hq" % ci].astype(np.float32) / np.float32(255)).cuda()
        r = SteeringGaussianResize2dTorch(support_sz=2, device=torch.device("cuda"), max_sigma=10)
        r.set_shape([2, 1, int(H), int(W)], scale_factors=[s, s])
        out = r.resize(feat, h[0], h[1], h[2])
        assert out.dtype == torch.float32 and out.is_cuda
        ref = g["%d/out" % ci]
        assert out.shape == ref.shape
        # the torch classes carry the reference's float32 geometry (lerf_sr_axis_tables_f32, bit-equal tables),
        # so what is left is float32 arithmetic on both sides
        assert np.max(np.abs(out.cpu().numpy() - ref)) <= F32_OBSERVED


@pytest.mark.parametrize("p", ["isc", "osc"])
def test_warp_classes_vs_golden(golden, p):
    from lerf_pytorch_amd.resize_right.resize_right2d_numpy import (
        AmplifiedLinearWarp2dNumpy, NearestWarp2dNumpy, SteeringGaussianWarp2dNumpy)
    g = golden("g4_warp.npz")
    M = g["%s/matrix" % p]
    feat = g["%s/feat" % p].astype(np.float32)
    h = g["%s/hq" % p].astype(np.float32) / np.float32(255)
    for S in (2, 4):
        w = SteeringGaussianWarp2dNumpy(support_sz=S, max_sigma=10)
        w.set_shape([3, 52, 52], M, [3, 60, 70])
        assert [w.pad_vec[1][0], w.pad_vec[1][1], w.pad_vec[2][0], w.pad_vec[2][1]] == list(g["%s/60x70/S%d/pad" % (p, S)])
        out = w.warp(feat, h[0], h[1], h[2])
        np.testing.assert_allclose(out, g["%s/60x70/S%d/gauss" % (p, S)], rtol=0, atol=1e-9, equal_nan=True)
    wl = AmplifiedLinearWarp2dNumpy()
    wl.set_shape([3, 52, 52], M, [3, 60, 70])
    np.testing.assert_allclose(wl.warp(feat, h[0]), g["%s/60x70/linear" % p], rtol=0, atol=1e-9, equal_nan=True)
    nn = NearestWarp2dNumpy()
    white = np.zeros((3, 52, 52), np.float32)
    white[:, 4:48, 4:48] = 255
    for hw in ((60, 70), (344, 228)):
        nn.set_shape([3, 52, 52], M, [3, hw[0], hw[1]])
        mo = nn.warp(white)
        assert np.array_equal(mo == 255, g["%s/%dx%d/mask" % (p, hw[0], hw[1])])
    nn.set_shape([3, 52, 52], M, [3, 60, 70])
    np.testing.assert_allclose(nn.warp(white), g["%s/60x70/nearest" % p], rtol=0, atol=0, equal_nan=True)


@pytest.mark.parametrize("p", ["isc", "osc"])
def test_warp_float32_vs_golden(torch, golden, p):
    from lerf_pytorch_amd import ops
    g = golden("g4_warp.npz")
    M = g["%s/matrix" % p]
    feat = torch.from_numpy(g["%s/feat" % p].transpose(1, 2, 0).copy()).cuda()
    hq = torch.from_numpy(g["%s/hq" % p].transpose(2, 3, 1, 0).copy()).cuda()
    geo = ops.WarpGeometry((52, 52), M, (344, 228), 2)
    o = ops.warp_hwc_u8(feat, hq, geo, "gauss", 10, out="f32").cpu().numpy()
    ref = g["%s/344x228/S2/gauss_f32" % p].transpose(1, 2, 0)
    assert np.array_equal(np.isnan(o), np.isnan(ref))
    ok = ~np.isnan(ref)
    # random hyper-parameters with distances up to 2 px: exponents reach 1e3, float32 rounding of the
    # quadratic form shows up at the 1e-3 level (0..255 scale); still 30x inside the 1e-4-of-range bound
    err = np.max(np.abs(o[ok] - ref[ok]))
    assert err <= F32_TOL and err <= 2e-3, err


SET5 = ["baby", "bird", "butterfly", "head", "woman"]


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("scale", [2, 3, 4])
@pytest.mark.parametrize("model", ["lerf-g", "lerf-l"])
def test_set5_sr_md5_and_psnr(oracle, luts_g, luts_l, eng_g, eng_l, model, scale, fused):
    """End-to-end known answers: md5 of the reference's uint8 outputs and the
    scripts.sh PSNR table (35.71/32.02/30.15 and 34.84/30.72/29.13)."""
    ref = json.load(open(os.path.join(GOLDEN, "g5_set5.json")))["sr"]
    published = {"lerf-g": {2: 35.71, 3: 32.02, 4: 30.15}, "lerf-l": {2: 34.84, 3: 30.72, 4: 29.13}}
    from oracle import c_oracle
    eng = eng_g if model == "lerf-g" else eng_l
    luts = luts_g if model == "lerf-g" else luts_l
    ps = []
    for n in SET5:
        lr = np.array(Image.open(os.path.join(DATA, "LR_bicubic/rrLR_X%.2f_%.2f" % (scale, scale), n + ".png")))
        gt = np.array(Image.open(os.path.join(DATA, "HR", n + ".png")))
        o8 = eng.sr(lr, scale, fused=fused)
        r = ref["%s/x%d/%s" % (model, scale, n)]
        assert list(o8.shape) == r["shape"]
        ref8 = c_oracle.sr_u8(lr, luts, scale, scale, linear=(model == "lerf-l"))
        assert _md5(ref8) == r["md5_out"]              # the checker reproduces the reference's bytes
        d = np.abs(o8.astype(int) - ref8.astype(int))
        assert d.max() <= 1                            # <= 1 LSB (north-star bound)
        # with the float64 tie guard the bytes are the reference's bytes
        assert _md5(o8) == r["md5_out"], "%d byte(s) differ from the reference for %s" % ((d != 0).sum(), n)
        p = oracle.psnr_y(gt, o8, scale)
        assert abs(p - r["psnr_y"]) <= 0.01
        ps.append(p)
    assert "%.2f" % np.mean(ps) == "%.2f" % published[model][scale]


@pytest.mark.parametrize("p", ["isc", "osc"])
@pytest.mark.parametrize("model", ["lerf-g", "lerf-l"])
def test_set5_warp_md5_and_mpsnr(oracle, luts_g, luts_l, eng_g, eng_l, model, p):
    ref = json.load(open(os.path.join(GOLDEN, "g5_set5.json")))["warp"]
    published = {"lerf-g": {"isc": 33.81, "osc": 27.89}, "lerf-l": {"isc": 32.90, "osc": 27.13}}
    eng = eng_g if model == "lerf-g" else eng_l
    ms = []
    for n in SET5:
        r = ref["%s/%s/%s" % (model, p, n)]
        lr = np.array(Image.open(os.path.join(DATA, p, n + ".png")))
        gt = np.array(Image.open(os.path.join(DATA, "HR", n + ".png")))
        o8, mask = eng.warp(lr, np.array(r["matrix"]), gt.shape[:2])
        assert int(mask.sum()) == r["mask_sum"] and _md5(mask.astype(np.uint8)) == r["md5_mask"]
        assert o8.shape == gt.shape
        ref8 = oracle.warp_pipeline(lr, luts_g if model == "lerf-g" else luts_l, np.array(r["matrix"]), gt.shape[:2],
                                    linear=(model == "lerf-l"))
        assert _md5(ref8 * mask) == r["md5_out_masked"]
        assert _md5(o8 * mask) == r["md5_out_masked"]   # the reference's own bytes (float32 + float64 tie guard)
        m = oracle.mpsnr(o8, gt, mask)
        assert abs(m - r["mpsnr"]) <= 0.01
        ms.append(m)
    assert "%.2f" % np.mean(ms) == "%.2f" % published[model][p]


def test_sr_full_size_properties(torch, eng_g):
    """BASELINE config 2 size (1920x1080 -> 3840x2160): size-independent properties.
    (a) tiling invariance: any interior crop, processed alone with a 7-px halo, reproduces the
        full-frame output exactly (stage radii 3+3+1);
    (b) a constant image stays constant away from the zero-padded border;
    (c) fused and unfused paths agree bit for bit."""
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
    x = torch.from_numpy(img).cuda()
    full = eng_g.sr(x, 2)
    assert tuple(full.shape) == (2160, 3840, 3)
    assert torch.equal(full, eng_g.sr(x, 2, fused=False))
    for (y0, x0, h, w) in ((100, 200, 64, 96), (500, 1000, 33, 47), (1000, 1800, 60, 100)):
        crop = x[y0 - 7:y0 + h + 7, x0 - 7:x0 + w + 7].contiguous()
        oc = eng_g.sr(crop, 2)
        assert torch.equal(oc[14:-14, 14:-14], full[2 * y0:2 * (y0 + h), 2 * x0:2 * (x0 + w)])
    const = torch.full((1080, 1920, 3), 77, dtype=torch.uint8, device="cuda")
    oc = eng_g.sr(const, 2)
    inner = oc[2:-2, 2:-2]
    assert int(inner.min()) == int(inner.max())


def test_batched_frames(torch, eng_g):
    rng = np.random.default_rng(5)
    imgs = torch.from_numpy(rng.integers(0, 256, (3, 40, 56, 3), dtype=np.uint8)).cuda()
    out = eng_g.sr(imgs, 2)
    for b in range(3):
        assert torch.equal(out[b], eng_g.sr(imgs[b], 2))


def test_error_behaviour(torch, eng_g):
    from lerf_pytorch_amd import ops
    with pytest.raises(ValueError):
        eng_g.sr(np.zeros((8, 8, 3), np.float32), 2)
    with pytest.raises(NotImplementedError):
        ops.SrGeometry((8, 8), [0.2, 0.2], None, 10)          # anti-aliased support beyond the kernels' maximum (8)
    with pytest.raises(ValueError):
        ops.SrGeometry((8, 8), [0.0, 1.0], None, 2)
    with pytest.raises(ValueError):
        ops.lut_stages(torch.zeros((8, 8, 3), dtype=torch.float32, device="cuda"), eng_g.luts)


@pytest.mark.parametrize("H,W,scale,world", [(200, 96, 2, 4), (150, 64, 1.5, 3), (128, 70, 3, 2), (97, 64, 2.4, 2)])
def test_strip_partition_single_gpu(torch, eng_g, H, W, scale, world):
    """multi-GPU strip partition, ranks emulated one after the other on one GPU: every rank computes
    its output rows from its LR rows + 7-row halo with the global geometry rebased to the strip;
    the stitched result must equal the full-frame result bit for bit (any scale, not just integer)."""
    from lerf_pytorch_amd import dist as ldist
    rng = np.random.default_rng(H + W)
    x = torch.from_numpy(rng.integers(0, 256, (H, W, 3), dtype=np.uint8)).cuda()
    full = eng_g.sr(x, scale)
    geo = eng_g.sr_geometry((H, W), scale)
    parts = []
    for r in range(world):
        plan = ldist.StripPlan(H, world, r, eng_g.support, geo.host["left_r"])
        assert plan.check_support(geo.host["left_r"])
        parts.append(ldist.sr_strip(eng_g, x[plan.ylo:plan.yhi].contiguous(), plan, geo))
    assert torch.equal(torch.cat(parts, dim=0), full)


def test_torch_custom_ops(torch, eng_g, oracle, luts_g):
    """torch.ops.lerf.* (C++ TORCH_LIBRARY registration) against the ORACLE, not against the engine"""
    from lerf_pytorch_amd import torch_ops
    s1, s2, pack = torch_ops.lut_args(eng_g.luts)
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (48, 40, 3), dtype=np.uint8)
    x = torch.from_numpy(img).cuda()
    of, oh, _, o8 = oracle.sr_pipeline(img, luts_g, 2, 2, return_all=True)
    out = torch.ops.lerf.sr_fused(x, s1, s2, pack, 2.0, 2.0, 2, 10.0)
    assert np.array_equal(out.cpu().numpy(), o8)
    assert np.array_equal(torch.ops.lerf.sr_fused(x, s1, s2, None, 2.0, 2.0, 2, 10.0).cpu().numpy(), o8)      # no pack: direct kernels
    outb = torch.ops.lerf.sr_fused(torch.stack([x, x]), s1, s2, pack, 2.0, 2.0, 2, 10.0)
    assert tuple(outb.shape) == (2, 96, 80, 3) and torch.equal(outb[1], out)
    feat, hq = torch.ops.lerf.lut_stages(x, s1, s2)
    assert np.array_equal(feat.cpu().numpy(), of) and np.array_equal(hq.cpu().numpy(), oh)
    fe = feat.permute(2, 0, 1).float().unsqueeze(0)
    hy = (hq.float() / 255).permute(3, 2, 0, 1).unsqueeze(1)
    o = torch.ops.lerf.resize_gauss(fe, hy[0], hy[1], hy[2], 2.0, 2.0, 2, 10.0)
    assert o.shape == (1, 3, 96, 80)
    ref = oracle.resize_params_f32(fe[0].cpu().numpy(), hy[0, 0].cpu().numpy(), hy[1, 0].cpu().numpy(), hy[2, 0].cpu().numpy(),
                                   2.0, 2.0, 2, 10, "gauss", geometry="torch32")
    assert np.abs(o[0].cpu().numpy() - ref).max() <= 1e-4
    M = np.array([[2.05, 0.12, 3.0], [-0.08, 1.95, 4.0], [1.5e-4, -1.0e-4, 1.0]])
    w = torch.ops.lerf.warp_fused(x, s1, s2, pack, torch.tensor(M, dtype=torch.float64), 90, 70, 2, 10.0)
    assert np.array_equal(w.cpu().numpy(), oracle.warp_pipeline(img, luts_g, M, (90, 70)))
    with pytest.raises((RuntimeError, NotImplementedError)):
        torch.ops.lerf.sr_fused(x.cpu(), [t.cpu() for t in s1], [t.cpu() for t in s2], None, 2.0, 2.0, 2, 10.0)   # no CPU kernel
    with pytest.raises(RuntimeError):
        torch.ops.lerf.sr_fused(x.float(), s1, s2, pack, 2.0, 2.0, 2, 10.0)


def test_torch_custom_ops_linear(torch, eng_l, oracle, luts_l):
    from lerf_pytorch_amd import torch_ops
    s1, s2, pack = torch_ops.lut_args(eng_l.luts)
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (30, 44, 3), dtype=np.uint8)
    x = torch.from_numpy(img).cuda()
    out = torch.ops.lerf.sr_fused(x, s1, s2, pack, 1.5, 2.0, 2, 10.0)
    assert np.array_equal(out.cpu().numpy(), oracle.sr_pipeline(img, luts_l, 1.5, 2.0, linear=True))
    feat, hq = torch.ops.lerf.lut_stages(x, s1, s2)
    fe = feat.permute(2, 0, 1).float().unsqueeze(0)
    al = (hq.float() / 255).permute(3, 2, 0, 1)[0].unsqueeze(0)
    o = torch.ops.lerf.resize_linear(fe, al, 1.5, 2.0, 1.0)
    ref = oracle.resize_params_f32(fe[0].cpu().numpy(), al[0].cpu().numpy(), None, None, 1.5, 2.0, 2, 1, "linear", geometry="torch32")
    assert np.abs(o[0].cpu().numpy() - ref).max() <= 1e-4


def test_torch_custom_ops_autograd(torch):
    """gradients of torch.ops.lerf.resize_{gauss,linear} == the class path, which tests/test_gpu_train.py pins to the
    reference's autograd (g11); schema / fake-kernel consistency through torch.library.opcheck"""
    from lerf_pytorch_amd import torch_ops  # noqa: F401
    dev = torch.device("cuda")
    rng = np.random.default_rng(21)
    feat = torch.tensor(rng.random((2, 1, 9, 8)) * 255, dtype=torch.float32, device=dev, requires_grad=True)
    hs = [torch.tensor(rng.random((2, 1, 9, 8)), dtype=torch.float32, device=dev, requires_grad=True) for _ in range(3)]
    from lerf_pytorch_amd.resize_right.resize_right2d_torch import AmplifiedLinearResize2dTorch, SteeringGaussianResize2dTorch
    wgt = torch.tensor(rng.random((2, 1, 18, 24)), dtype=torch.float32, device=dev)
    # op path
    (torch.ops.lerf.resize_gauss(feat, hs[0], hs[1], hs[2], 2.0, 3.0, 2, 10.0) * wgt).sum().backward()
    got = [feat.grad.clone()] + [h.grad.clone() for h in hs]
    for t in [feat] + hs:
        t.grad = None
    # class path (golden-tested against the reference's autograd in tests/test_gpu_train.py)
    r = SteeringGaussianResize2dTorch(support_sz=2, device=dev, max_sigma=10)
    r.set_shape([2, 1, 9, 8], scale_factors=[2.0, 3.0])
    (r.resize(feat, hs[0], hs[1], hs[2]) * wgt).sum().backward()
    want = [feat.grad.clone()] + [h.grad.clone() for h in hs]
    for a, b in zip(got, want):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-4 * float(b.abs().max()))
    for t in [feat] + hs:
        t.grad = None
    (torch.ops.lerf.resize_linear(feat, hs[0], 2.0, 3.0, 1.0) * wgt).sum().backward()
    got = [feat.grad.clone(), hs[0].grad.clone()]
    feat.grad = None
    hs[0].grad = None
    rl = AmplifiedLinearResize2dTorch(device=dev)
    rl.set_shape([2, 1, 9, 8], scale_factors=[2.0, 3.0])
    (rl.resize(feat, hs[0]) * wgt).sum().backward()
    for a, b in zip(got, [feat.grad, hs[0].grad]):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-4 * float(b.abs().max()))
    # finite-difference spot check of the op's gradient w.r.t. one hyper-parameter entry
    torch.library.opcheck(torch.ops.lerf.resize_linear, (feat.detach(), hs[0].detach(), 2.0, 3.0, 1.0), test_utils=("test_schema", "test_faketensor"))


def test_full_frame_bytes_equal_cpu_oracle(torch, eng_g, luts_g):
    """1080p -> 4K, uniform noise (every LUT entry and simplex ordering exercised): the uint8 frame from the fused
    kernel equals the float64 C oracle byte for byte (tie guard), not just within 1 LSB."""
    from oracle import c_oracle
    rng = np.random.default_rng(2024)
    img = rng.integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
    out = eng_g.sr(img, 2)
    ref = c_oracle.sr_u8(img, luts_g, 2, 2)
    d = np.abs(out.astype(int) - ref.astype(int))
    assert d.max() <= 1
    assert (d != 0).sum() == 0, "%d of %d bytes differ" % ((d != 0).sum(), d.size)


@pytest.mark.parametrize("shape", [(1, 1, 3), (5, 7, 3), (64, 64, 3), (65, 130, 3), (200, 75, 3)])
def test_direct_and_fused_stage_kernels_agree(torch, eng_g, eng_l, shape):
    """lerf_lut_stages_u8 (direct kernels, global-memory LUT gathers) and lerf_stages_packed_u8 (tile-fused,
    LDS-resident LUT pieces) are independent implementations of stages 1+2: they must agree bit for bit."""
    from lerf_pytorch_amd import ops
    rng = np.random.default_rng(sum(shape) + 1)
    x = torch.from_numpy(rng.integers(0, 256, shape, dtype=np.uint8)).cuda()
    for eng in (eng_g, eng_l):
        f1, h1 = ops.lut_stages(x, eng.luts)
        f2, h2 = ops.unpack_stages(ops.stages_packed(x, eng.luts), eng.luts.oC)
        assert torch.equal(f1, f2) and torch.equal(h1, h2)
    # batched frames
    xb = torch.from_numpy(rng.integers(0, 256, (3,) + shape, dtype=np.uint8)).cuda()
    pb = ops.stages_packed(xb, eng_g.luts)
    for b in range(3):
        assert torch.equal(pb[b], ops.stages_packed(xb[b], eng_g.luts))


@pytest.mark.parametrize("p", ["isc", "osc"])
def test_warp_packed_equals_direct_warp(torch, eng_g, p):
    from lerf_pytorch_amd import ops
    ref = json.load(open(os.path.join(GOLDEN, "g5_set5.json")))["warp"]
    r = ref["lerf-g/%s/woman" % p]
    lr = torch.from_numpy(np.array(Image.open(os.path.join(DATA, p, "woman.png")))).cuda()
    geo = ops.WarpGeometry(lr.shape[:2], np.array(r["matrix"]), (344, 228), 2)
    feat, hq = ops.lut_stages(lr, eng_g.luts)
    a = ops.warp_hwc_u8(feat, hq, geo, "gauss", 10.0, out="f32")
    b = ops.warp_packed(ops.stages_packed(lr, eng_g.luts), geo, "gauss", 10.0, out="f32")
    assert torch.equal(torch.isnan(a), torch.isnan(b))
    ok = ~torch.isnan(a)
    assert float((a[ok] - b[ok]).abs().max()) <= 1e-3


@pytest.mark.parametrize("H,W,scale,S", [
    (1, 1, 2, 2), (2, 3, 2, 2), (7, 5, 3, 2), (63, 65, 2, 2), (64, 64, 1, 2), (64, 128, 4, 2), (65, 64, 2.5, 2),
    (33, 200, (1.5, 2.0), 2), (130, 70, (1.0, 3.0), 2), (100, 90, 2, 4), (40, 300, 1.25, 4), (129, 129, 4, 4)])
def test_fused_equals_unfused_and_oracle_shape_sweep(torch, oracle, luts_g, luts_l, eng_g, eng_l, H, W, scale, S):
    """edge shapes: single pixels, tiles cut by the frame, scale 1 and 4, anisotropic scales, odd row pitches,
    S = 2 and 4, both models.  fused == unfused bit for bit, and == the float64 oracle on the small ones."""
    import lerf_pytorch_amd as L
    rng = np.random.default_rng(H * 1000 + W)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    sc = scale if isinstance(scale, tuple) else (scale, scale)
    for model, luts in (("lerf-g", luts_g), ("lerf-l", luts_l)):
        if model == "lerf-l" and S != 2:
            continue
        eng = (eng_g if model == "lerf-g" else eng_l) if S == 2 else L.LerfEngine(eng_g.luts, support=S)
        a = eng.sr(img, sc, fused=True)
        b = eng.sr(img, sc, fused=False)
        assert a.shape == (oracle.out_size(H, sc[0]), oracle.out_size(W, sc[1]), 3)
        assert np.array_equal(a, b)
        if H * W <= 130 * 70:
            ref = oracle.sr_pipeline(img, luts, sc[0], sc[1], S=S, linear=(model == "lerf-l"))
            assert np.array_equal(a, ref)


@pytest.mark.parametrize("p", ["isc", "osc"])
def test_fixed_kernel_warp_classes_vs_golden(golden, p):
    """the reference's non-learned warps (resize_right2d_numpy.py:451-494) through the same warp kernel"""
    from lerf_pytorch_amd.resize_right import resize_right2d_numpy as rn
    g4, g7 = golden("g4_warp.npz"), golden("g7_fixed_warp.npz")
    feat = g4["%s/feat" % p].astype(np.float32)
    for name, cls in (("cubic", rn.BicubicWarp2dNumpy), ("bilinear", rn.BilinearWarp2dNumpy),
                      ("lanczos2", rn.Lanczos2Warp2dNumpy), ("lanczos3", rn.Lanczos3Warp2dNumpy)):
        w = cls()
        w.set_shape([3, 52, 52], g4["%s/matrix" % p], [3, 60, 70])
        assert [w.pad_vec[1][0], w.pad_vec[1][1], w.pad_vec[2][0], w.pad_vec[2][1]] == list(g7["%s/%s/pad" % (p, name)])
        np.testing.assert_allclose(w.warp(feat), g7["%s/%s" % (p, name)], rtol=0, atol=1e-9, equal_nan=True)


# ---------------------------------------------------------------- fixed-kernel SR
@pytest.mark.parametrize("kind,S", [("cubic", 4), ("bilinear", 2), ("lanczos2", 4), ("lanczos3", 6)])
@pytest.mark.parametrize("shape,scale", [((3, 24, 20), (2, 2)), ((2, 17, 23), (3, 3)), ((1, 9, 7), (1.5, 2.4))])
def test_fixed_kernel_resize_vs_oracle(torch, oracle, kind, S, shape, scale):
    from lerf_pytorch_amd import ops
    rng = np.random.default_rng(S * 100 + shape[1])
    x = rng.integers(0, 256, shape).astype(np.float32)
    ref = oracle.resize_params_f32(x, None, None, None, scale[0], scale[1], S, 1, kind)
    geo = ops.SrGeometry(shape[1:], list(scale), None, S)
    xt = torch.from_numpy(x).cuda()
    o64 = ops.resize_planar(xt, [], geo, kind, 1.0, out="f64").cpu().numpy()
    assert np.max(np.abs(o64 - ref)) <= 1e-9
    o32 = ops.resize_planar(xt, [], geo, kind, 1.0, out="f32").cpu().numpy()
    assert np.max(np.abs(o32 - ref)) <= F32_OBSERVED
    # uint8 HWC frames in and out
    x8 = torch.from_numpy(np.ascontiguousarray(x.astype(np.uint8).transpose(1, 2, 0))).cuda()
    o8 = ops.resize_hwc_u8(x8, None, geo, kind, 1.0, out="u8").cpu().numpy().transpose(2, 0, 1)
    want = np.clip(np.round(ref), 0, 255)
    near_tie = np.abs(ref - np.floor(ref) - 0.5) < 1e-3
    assert np.all((o8 == want) | near_tie)
    assert np.max(np.abs(o8.astype(int) - want)) <= 1


@pytest.mark.parametrize("ci", range(5))
def test_bicubic_resize2d_torch_class(torch, golden, oracle, ci):
    from lerf_pytorch_amd.resize_right.resize_right2d_torch import BicubicResize2dTorch
    g = golden("g9_bicubic_resize.npz")
    x = g["%d/x" % ci].astype(np.float32)
    s = [float(v) for v in g["%d/scale" % ci]]
    r = BicubicResize2dTorch(support_sz=4, device=torch.device("cuda"))
    r.set_shape(list(x.shape), scale_factors=s)
    o = r.resize(torch.from_numpy(x).cuda())
    ref = g["%d/out" % ci]
    assert list(o.shape) == list(ref.shape) and o.dtype == torch.float32
    assert np.max(np.abs(o.cpu().numpy() - ref)) <= F32_OBSERVED   # same float32 geometry, float32 arithmetic on both sides
    B, C, H, W = x.shape
    f64 = oracle.resize_params_f32(x.reshape(B * C, H, W), None, None, None, s[0], s[1], 4, 1, "cubic",
                                   geometry="torch32").reshape(ref.shape)
    assert np.max(np.abs(o.cpu().numpy() - f64)) <= F32_OBSERVED


def test_resize2d_torch_base_class_needs_a_kernel(torch):
    from lerf_pytorch_amd.resize_right.resize_right2d_torch import Resize2dTorch
    r = Resize2dTorch()
    r.set_shape([1, 1, 8, 8], scale_factors=2)
    with pytest.raises(NotImplementedError):
        r.resize(torch.zeros(1, 1, 8, 8, device="cuda"))


# ---------------------------------------------------------------- down-sampling
@pytest.mark.parametrize("ci", range(6))
def test_downscale_numpy_classes_golden(golden, ci):
    from lerf_pytorch_amd.resize_right.resize_right2d_numpy import AmplifiedLinearResize2dNumpy, SteeringGaussianResize2dNumpy
    g = golden("g12_downscale.npz")
    Cn, H, W, sh, sw, S, S2 = g["%d/cfg" % ci]
    feat = g["%d/feat" % ci].astype(np.float32)
    hy = g["%d/hq" % ci].astype(np.float32) / np.float32(255)
    r = SteeringGaussianResize2dNumpy(support_sz=int(S), max_sigma=10)
    r.set_shape([int(Cn), int(H), int(W)], scale_factors=[float(sh), float(sw)])
    assert r.support_sz == int(S2) and r.antialias == (sh < 1.0)
    assert [r.pad_vec[1][0], r.pad_vec[1][1], r.pad_vec[2][0], r.pad_vec[2][1]] == list(g["%d/pad" % ci])
    assert np.max(np.abs(r.resize(feat, hy[0], hy[1], hy[2]) - g["%d/gauss" % ci])) <= 1e-9
    if "%d/linear" % ci in g:
        l = AmplifiedLinearResize2dNumpy()
        l.set_shape([int(Cn), int(H), int(W)], scale_factors=[float(sh), float(sw)])
        np.testing.assert_allclose(l.resize(feat, hy[0]), g["%d/linear" % ci], rtol=0, atol=1e-9, equal_nan=True)


@pytest.mark.parametrize("ci", range(6))
def test_downscale_torch_class_golden(torch, golden, ci):
    """the torch classes do not anti-alias (`if False:`, resize_right2d_torch.py:38-44)."""
    from lerf_pytorch_amd.resize_right.resize_right2d_torch import SteeringGaussianResize2dTorch
    g = golden("g12_downscale.npz")
    Cn, H, W, sh, sw, S, S2 = g["%d/cfg" % ci]
    feat = torch.tensor(g["%d/feat" % ci].astype(np.float32)[None], device="cuda")
    hy = torch.tensor((g["%d/hq" % ci].astype(np.float32) / np.float32(255))[:, None], device="cuda")
    t = SteeringGaussianResize2dTorch(support_sz=int(S), device=torch.device("cuda"), max_sigma=10)
    if ci in (3, 5):  # row pads != column pads there: the reference mis-pads (F.pad order), not reproduced
        with pytest.raises(NotImplementedError, match="mis-pads"):
            t.set_shape([1, int(Cn), int(H), int(W)], scale_factors=[float(sh), float(sw)])
        return
    t.set_shape([1, int(Cn), int(H), int(W)], scale_factors=[float(sh), float(sw)])
    o = t.resize(feat, hy[0], hy[1], hy[2]).cpu().numpy()
    assert np.max(np.abs(o - g["%d/torch" % ci])) <= F32_OBSERVED


@pytest.mark.parametrize("model,scale", [("lerf-g", (0.5, 0.5)), ("lerf-g", (0.75, 1.25)), ("lerf-l", (0.6, 0.6))])
def test_engine_downscale_vs_oracle(eng_g, eng_l, oracle, luts_g, luts_l, model, scale):
    """LerfEngine.sr with a scale < 1 (falls back from the tile-fused kernel to the direct kernels)."""
    eng, luts = (eng_g, luts_g) if model == "lerf-g" else (eng_l, luts_l)
    rng = np.random.default_rng(int(scale[0] * 100))
    img = rng.integers(0, 256, (40, 52, 3), dtype=np.uint8)
    from lerf_pytorch_amd.resize_right.resize_right2d_numpy import AmplifiedLinearResize2dNumpy, SteeringGaussianResize2dNumpy
    feat, hq = eng.stages(img)
    fc = feat.transpose(2, 0, 1).astype(np.float32)
    hy = hq.astype(np.float32) / np.float32(255)
    if model == "lerf-g":
        want = oracle.resize_params_f32(fc, hy[..., 0].transpose(2, 0, 1), hy[..., 1].transpose(2, 0, 1), hy[..., 2].transpose(2, 0, 1),
                                        scale[0], scale[1], 2, 10, "gauss")
        r = SteeringGaussianResize2dNumpy(support_sz=2, max_sigma=10)
        r.set_shape(list(fc.shape), scale_factors=list(scale))
        got = r.resize(fc, hy[..., 0].transpose(2, 0, 1), hy[..., 1].transpose(2, 0, 1), hy[..., 2].transpose(2, 0, 1))
    else:
        want = oracle.resize_params_f32(fc, hy[..., 0].transpose(2, 0, 1), None, None, scale[0], scale[1], 2, 1, "linear")
        r = AmplifiedLinearResize2dNumpy()
        r.set_shape(list(fc.shape), scale_factors=list(scale))
        got = r.resize(fc, hy[..., 0].transpose(2, 0, 1))
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-9, equal_nan=True)


@pytest.mark.parametrize("transport", ["dma", "zero_copy"])
def test_streaming_sr_matches_engine(torch, eng_g, transport):
    """host -> host pipeline (copy engines beside the kernel on three streams, or kernel I/O on pinned host memory) returns
    the same bytes as the synchronous engine, batch after batch through the reused slots."""
    from lerf_pytorch_amd.stream import StreamingSR
    rng = np.random.default_rng(12)
    batches = [rng.integers(0, 256, (2, 40, 52, 3), dtype=np.uint8) for _ in range(7)]
    st = StreamingSR(eng_g, (40, 52), 2, frames_per_batch=2, depth=2, transport=transport)
    outs = [o.copy() for o in st.run(batches)]
    assert len(outs) == 7
    st3 = StreamingSR(eng_g, (40, 52), 2, frames_per_batch=2, transport=transport)         # default depth
    assert st3.depth == (3 if transport == "dma" else 2)
    assert all(np.array_equal(a, b) for a, b in zip(outs, [o.copy() for o in st3.run(batches)]))
    with pytest.raises(ValueError):
        StreamingSR(eng_g, (40, 52), 2, transport="carrier pigeon")
    for b, o in zip(batches, outs):
        assert np.array_equal(o, eng_g.sr(torch.from_numpy(b).cuda(), 2).cpu().numpy())
    st.input()[:] = batches[1]                     # producer fills the pinned buffer itself
    i = st.submit()
    st.submit(batches[2])
    with pytest.raises(RuntimeError):
        st.submit(batches[3])                      # every slot holds an uncollected result
    assert np.array_equal(st.result(i), outs[1])
    with pytest.raises(ValueError):
        st.submit(batches[0][:1])


@pytest.mark.parametrize("p", ["isc", "osc"])
def test_torch_warp_classes_vs_reference_torch_path(torch, golden, p):
    """A9: all seven *Warp2dTorch mirrors.  SteeringGaussian (S = 2, 4), AmplifiedLinear, Nearest and Bicubic against
    the outputs of the reference's own torch classes (g13, resize_right2d_torch.py:346-487; float64 like theirs);
    Bilinear / Lanczos2 / Lanczos3 exist only as numpy classes upstream and are held to those (g7)."""
    from lerf_pytorch_amd.resize_right import resize_right2d_torch as T
    g4, g7, g13 = golden("g4_warp.npz"), golden("g7_fixed_warp.npz"), golden("g13_torch_warp.npz")
    dev = torch.device("cuda")
    M = torch.tensor(g4["%s/matrix" % p], dtype=torch.float64, device=dev)
    feat = torch.from_numpy(g4["%s/feat" % p][:2].astype(np.float32)).unsqueeze(1).to(dev)                    # [2,1,52,52]
    hy = torch.from_numpy(g4["%s/hq" % p][:, :2].astype(np.float32) / np.float32(255)).unsqueeze(2).to(dev)   # [3,2,1,52,52]

    def check(out, ref, tol):
        assert out.dtype == torch.float64 and out.is_cuda and tuple(out.shape) == ref.shape
        np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=0, atol=tol, equal_nan=True)

    for hw in ((60, 70), (97, 41)):
        key = "%s/%dx%d" % (p, hw[0], hw[1])
        for S in (2, 4):
            w = T.SteeringGaussianWarp2dTorch(support_sz=S, device=dev, max_sigma=10)
            w.set_shape([2, 1, 52, 52], M, [2, 1, hw[0], hw[1]])
            assert list(w.pad_vec) == list(g13[key + "/pad_S%d" % S])
            assert w.out_shape == [2, 1, hw[0], hw[1]] and w.in_sz == [52, 52] and w.out_sz == [hw[0], hw[1]]
            check(w.warp(feat, hy[0], hy[1], hy[2]), g13[key + "/gauss_S%d" % S], 1e-9)
        wl = T.AmplifiedLinearWarp2dTorch(device=dev)
        wl.set_shape([2, 1, 52, 52], M, [2, 1, hw[0], hw[1]])
        check(wl.warp(feat, hy[0]), g13[key + "/linear"], 1e-9)
        nn = T.NearestWarp2dTorch(device=dev)
        nn.set_shape([2, 1, 52, 52], M, [2, 1, hw[0], hw[1]])
        check(nn.warp(feat), g13[key + "/nearest"], 0)
        white = torch.zeros((2, 1, 52, 52), device=dev)
        white[:, :, 4:48, 4:48] = 255
        check(nn.warp(white), g13[key + "/nearest_white"], 0)
        bc = T.BicubicWarp2dTorch(device=dev)
        bc.set_shape([2, 1, 52, 52], M, [2, 1, hw[0], hw[1]])
        check(bc.warp(feat), g13[key + "/cubic"], 1e-9)
    feat3 = torch.from_numpy(g4["%s/feat" % p].astype(np.float32)).unsqueeze(0).to(dev)                       # [1,3,52,52]
    for name, cls in (("bilinear", T.BilinearWarp2dTorch), ("lanczos2", T.Lanczos2Warp2dTorch), ("lanczos3", T.Lanczos3Warp2dTorch)):
        w = cls(device=dev)
        w.set_shape([1, 3, 52, 52], M, [1, 3, 60, 70])
        check(w.warp(feat3), g7["%s/%s" % (p, name)][None], 1e-9)
    with pytest.raises(ValueError):
        bc.warp(feat3[:, :, :50])                                                # shape other than set_shape's


def test_pad_modes_of_the_classes(torch, golden):
    """pad_mode != 'constant' through the class API (the reference hands it to np.pad / F.pad for the IMAGE operand,
    resize_right2d_numpy.py:143,208,560; resize_right2d_torch.py:189): every index-remapping mode of both libraries."""
    from lerf_pytorch_amd.resize_right import resize_right2d_numpy as N
    from lerf_pytorch_amd.resize_right import resize_right2d_torch as T
    g, g4 = golden("g15_pad_modes.npz"), golden("g4_warp.npz")
    feat = g["feat"].astype(np.float32)
    h = g["hq"].astype(np.float32) / np.float32(255)
    for mode in ("edge", "reflect", "symmetric", "wrap"):
        for S, sc in ((2, (2.0, 3.0)), (4, (1.5, 2.0))):
            r = N.SteeringGaussianResize2dNumpy(support_sz=S, max_sigma=10, pad_mode=mode)
            r.set_shape([2, 11, 9], scale_factors=list(sc))
            np.testing.assert_allclose(r.resize(feat, h[0], h[1], h[2]), g["sr/%s/gauss_S%d" % (mode, S)], rtol=0, atol=1e-9)
        rl = N.AmplifiedLinearResize2dNumpy(pad_mode=mode)
        rl.set_shape([2, 11, 9], scale_factors=[3.0, 2.0])
        np.testing.assert_allclose(rl.resize(feat, h[0]), g["sr/%s/linear" % mode], rtol=0, atol=1e-9, equal_nan=True)
        for p in ("isc", "osc"):
            f52 = g4["%s/feat" % p].astype(np.float32)
            h52 = g4["%s/hq" % p].astype(np.float32) / np.float32(255)
            w = N.SteeringGaussianWarp2dNumpy(support_sz=2, max_sigma=10, pad_mode=mode)
            w.set_shape([3, 52, 52], g4["%s/matrix" % p], [3, 60, 70])
            np.testing.assert_allclose(w.warp(f52, h52[0], h52[1], h52[2]), g["warp/%s/%s" % (mode, p)], rtol=0, atol=1e-9,
                                       equal_nan=True)
    dev = torch.device("cuda")
    ft = torch.from_numpy(feat)[None].to(dev)
    ht = torch.from_numpy(h)[:, None].to(dev)
    for mode in ("replicate", "reflect", "circular"):
        r = T.SteeringGaussianResize2dTorch(support_sz=2, device=dev, max_sigma=10, pad_mode=mode)
        r.set_shape([1, 2, 11, 9], scale_factors=[2.0, 2.0])
        out = r.resize(ft, ht[0], ht[1], ht[2])
        assert np.abs(out.cpu().numpy() - g["torch/%s/gauss" % mode]).max() <= F32_OBSERVED
    with pytest.raises(NotImplementedError):
        N.SteeringGaussianResize2dNumpy(pad_mode="linear_ramp")
    with pytest.raises(NotImplementedError):
        T.SteeringGaussianResize2dTorch(pad_mode="symmetric")              # not an F.pad mode


def test_set_shape_dense_geometry_attributes(torch, golden):
    """field_of_view_x/y and dis_x/y as the reference's set_shape leaves them (g16), numpy and torch classes"""
    from lerf_pytorch_amd.resize_right import resize_right2d_numpy as N
    from lerf_pytorch_amd.resize_right import resize_right2d_torch as T
    g = golden("g16_geometry_attrs.npz")
    for ci in (0, 1):
        H, W, sh, sw, S = g["%d/cfg" % ci]
        r = N.SteeringGaussianResize2dNumpy(support_sz=int(S), max_sigma=10)
        r.set_shape([3, int(H), int(W)], scale_factors=[sh, sw])
        t = T.SteeringGaussianResize2dTorch(support_sz=int(S), device=torch.device("cuda"), max_sigma=10)
        t.set_shape([2, 3, int(H), int(W)], scale_factors=[sh, sw])
        for nm in ("field_of_view_x", "field_of_view_y", "dis_x", "dis_y"):
            a, ref = getattr(r, nm), g["%d/numpy/%s" % (ci, nm)]
            assert a.dtype == ref.dtype and np.array_equal(a, ref), nm
            b, ref = getattr(t, nm), g["%d/torch/%s" % (ci, nm)]
            assert b.is_cuda and tuple(b.shape) == ref.shape and np.array_equal(b.cpu().numpy(), ref), nm


@pytest.mark.parametrize("interval", [3, 5, 6, 7])
def test_four_simplex_interp_other_intervals(golden, interval):
    """FourSimplexInterpFaster drop-in with interval != 4 (q = 2^interval, L = 2^(8-interval)+1, eval_lut_sr.py:27-28)"""
    from lerf_pytorch_amd.resample.eval_lut_sr import FourSimplexInterpFaster, mode_pad_dict
    g = golden("g17_intervals.npz")
    img = g["img"].astype(np.float32)
    lut = g["lut/%d" % interval]
    oC = lut.shape[1]
    for mode in "sct":
        pad = mode_pad_dict[mode]
        for r in (0, 3):
            rot = np.rot90(img, r)
            h, w, _ = rot.shape
            img_in = np.pad(rot, ((0, pad), (0, pad), (0, 0)), mode="edge").transpose((2, 0, 1))
            out = FourSimplexInterpFaster(lut.astype(np.float32), img_in, h, w, interval, 4 - r, upscale=1, mode=mode, oC=oC)
            ref = g["out/%d/%s/%d" % (interval, mode, r)]
            assert out.dtype == np.float64 and np.array_equal(out, ref)
    with pytest.raises(ValueError):
        FourSimplexInterpFaster(lut.astype(np.float32), img_in, h, w, 9, 0, mode="s", oC=oC)


@pytest.mark.parametrize("mode", ["d", "y"])
def test_four_simplex_interp_modes_d_y(torch, golden, oracle, luts_g, mode):
    """the 'd' / 'y' sampling patterns: the function mirror against the reference (g18), and a whole LUT stage pair with
    modes "sdy" through the general-mode kernels against the oracle"""
    from lerf_pytorch_amd.resample.eval_lut_sr import FourSimplexInterpFaster, mode_pad_dict
    import lerf_pytorch_amd as L
    g = golden("g18_modes_dy.npz")
    img = g["img"].astype(np.float32)
    pad = mode_pad_dict[mode]
    for key, oC in (("s1_sr0", 1), ("s2_tr1", 3)):
        for r in range(4):
            rot = np.rot90(img, r)
            h, w, _ = rot.shape
            img_in = np.pad(rot, ((0, pad), (0, pad), (0, 0)), mode="edge").transpose((2, 0, 1))
            out = FourSimplexInterpFaster(luts_g[key].astype(np.float32), img_in, h, w, 4, 4 - r, upscale=1, mode=mode, oC=oC)
            assert np.array_equal(out, g["%s/%s/%d" % (mode, key, r)])
    # a model whose mode set contains the pattern (LUT contents borrowed from the shipped tables)
    arrays = {"s1_sr0": luts_g["s1_sr0"], "s1_%sr0" % mode: luts_g["s1_cr0"], "s2_sr0": luts_g["s2_sr0"], "s2_sr1": luts_g["s2_sr1"],
              "s2_%sr0" % mode: luts_g["s2_tr0"], "s2_%sr1" % mode: luts_g["s2_tr1"]}
    modes = "s" + mode
    eng = L.LerfEngine(L.LutSet(arrays, 3, modes, modes))
    rng = np.random.default_rng(181)
    x = rng.integers(0, 256, (37, 29, 3), dtype=np.uint8)
    feat, hq = eng.stages(x)
    of, oh = oracle.lut_stages(x, arrays, 3, modes, modes)
    assert np.array_equal(feat, of) and np.array_equal(hq, oh)
    assert np.array_equal(eng.sr(x, 2), oracle.sr_pipeline(x, arrays, 2, 2, modes=modes, modes2=modes))


@pytest.mark.parametrize("oC,interval", [(1, 4), (3, 4), (3, 3)])
def test_lut_interp_abi6_forms_agree(torch, oracle, luts_g, oC, interval):
    """lerf_lut_interp (ABI 6): uint8 and float32 images, int16 numerators and float32 / float64 values, the four rotations
    written through the strides of the rotated view -- all equal to the int16 entry point of ABI 5 followed by np.rot90 and /q
    (resample/eval_lut_sr.py:464-469), which the golden passes above pin to the reference."""
    from lerf_pytorch_amd import _lib, ops
    rng = np.random.default_rng(oC * 10 + interval)
    L = 2 ** (8 - interval) + 1
    lut = torch.from_numpy(rng.integers(-128, 128, (L ** 4, oC), dtype=np.int8)).cuda() if interval != 4 else \
        torch.from_numpy(np.ascontiguousarray(luts_g["s2_cr1" if oC == 3 else "s1_tr0"].reshape(-1, oC))).cuda()
    h, w = 37, 53
    img8 = rng.integers(0, 256, (3, h + 3, w + 3), dtype=np.uint8)
    x8 = torch.from_numpy(img8).cuda()
    xf = x8.to(torch.float32)
    xv = torch.from_numpy(np.ascontiguousarray(img8.transpose(1, 2, 0))).cuda().to(torch.float32).permute(2, 0, 1)     # an HWC buffer viewed as CHW
    for mode in ("s", "c", "t"):
        dy, dx = _lib.mode_offsets(mode, 0)
        want = ops.lut_interp_i16(x8, h, w, dy, dx, lut, interval).reshape(3 * oC, h, w).cpu().numpy()
        for rot in (0, 1, 2, 3, 4, -1):
            ref = np.rot90(want, rot, [1, 2])
            for x in (x8, xf, xv):
                got = ops.lut_interp(x, h, w, dy, dx, lut, interval, rot=rot, out_dtype=torch.float64).cpu().numpy()
                assert got.dtype == np.float64 and got.shape == ref.shape and np.array_equal(got, ref / float(2 ** interval))
            n16 = ops.lut_interp(x8, h, w, dy, dx, lut, interval, rot=rot, out_dtype=torch.int16).cpu().numpy()
            f32 = ops.lut_interp(xf, h, w, dy, dx, lut, interval, rot=rot, out_dtype=torch.float32).cpu().numpy()
            assert np.array_equal(n16, ref) and f32.dtype == np.float32 and np.array_equal(f32, (ref / float(2 ** interval)).astype(np.float32))
    # non-integer / out-of-range float pixels: rounded half-to-even and clipped, as `.round().clamp(0, 255)` did for the uint8 kernels
    odd = torch.tensor([[[-3.0, 0.5, 1.5, 2.5, 254.5, 255.5, 300.0, 7.49]]], device="cuda").expand(1, 4, 8).contiguous()
    dy, dx = _lib.mode_offsets("s", 0)
    a = ops.lut_interp(odd, 3, 7, dy, dx, lut, interval, out_dtype=torch.int16)
    b = ops.lut_interp(odd.round().clamp(0, 255).to(torch.uint8), 3, 7, dy, dx, lut, interval, out_dtype=torch.int16)
    assert torch.equal(a, b)
    with pytest.raises(ValueError):
        ops.lut_interp(x8.to(torch.float64), h, w, dy, dx, lut, interval)
