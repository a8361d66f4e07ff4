"""GPU tests of the fine-tuning path: the SWF2LUT mirror against vectors generated from the
reference's own SWF2LUT (tests/golden/g10_swf2lut.npz: forward values and autograd gradients) and against the numpy
restatement in oracle/.

Tolerances: forward values bit-exact (integer arithmetic carried in float32); gradients 2e-5 relative to the largest
gradient entry (float32 products accumulated with atomics in a different order than autograd's)."""
import os
import sys
import types

import numpy as np
import pytest

from conftest import ASSETS, GOLDEN

sys.path.insert(0, GOLDEN)
import swf_inputs  # noqa: E402

pytestmark = pytest.mark.gpu
GRAD_RTOL = 2e-5


@pytest.fixture(scope="module")
def torch():
    import torch as t
    assert t.cuda.is_available(), "GPU tests need an MI355X"
    return t


def _model(torch, name):
    from lerf_pytorch_amd.resample.model import SWF2LUT
    opt = types.SimpleNamespace(modes="sct", modes2="sct", stages=2, norm=255, interval=4,
                                expDir=os.path.join(ASSETS, name), lutName="LUTft")
    return SWF2LUT(opt, inC=1, outC=3 if name == "lerf-g" else 1).cuda()


@pytest.fixture(scope="module")
def models(torch):
    return {n: _model(torch, n) for n in ("lerf-g", "lerf-l")}


def _close(a, b, scale=None):
    scale = np.abs(b).max() if scale is None else scale
    return np.max(np.abs(a - b)) <= GRAD_RTOL * max(scale, 1.0)


@pytest.mark.parametrize("mi,mode", list(enumerate("sdyct")))
@pytest.mark.parametrize("name", ["lerf-g", "lerf-l"])
def test_interp_torch_batch_golden(torch, golden, models, name, mi, mode):
    g = golden("g10_swf2lut.npz")
    m = models[name]
    outC = m.outC
    key = "weight_s2_%sr0" % (mode if mode in "sct" else "s")
    bd, img, G = swf_inputs.case_inputs(1000 + mi, mode, outC)
    w = torch.tensor(swf_inputs.case_weight(getattr(m, key).detach().cpu().numpy(), 2000 + mi), device="cuda", requires_grad=True)
    x = torch.tensor(img, device="cuda", requires_grad=True)
    o = m.InterpTorchBatch(w, outC, mode, x, bd)
    pre = "%s/interp/%s/" % (name, mode)
    assert np.array_equal(o.detach().cpu().numpy(), g[pre + "out"])                 # bit-exact
    (o * torch.tensor(G, device="cuda")).sum().backward()
    assert _close(x.grad.cpu().numpy(), g[pre + "gimg"])
    gw = w.grad.cpu().numpy()
    rows = np.nonzero(np.abs(gw).sum(1))[0]
    assert np.array_equal(rows, g[pre + "gw_rows"])                                 # same LUT rows touched, same clamp gate
    assert _close(gw[rows], g[pre + "gw_vals"])


@pytest.mark.parametrize("mode,outC,shape", [("s", 3, (1, 3, 20, 33)), ("c", 1, (3, 1, 5, 70)), ("t", 3, (2, 2, 9, 4)),
                                             ("y", 1, (1, 1, 1, 1)), ("d", 3, (1, 1, 64, 300))])
def test_interp_vs_oracle_random(torch, oracle, models, mode, outC, shape):
    from lerf_pytorch_amd.resample.model import _InterpFn
    rng = np.random.default_rng(sum(shape) + ord(mode))
    bd = swf_inputs.MODE_PAD[mode]
    B, Cn, h, w = shape
    img = rng.integers(0, 256, (B, Cn, h + bd, w + bd)).astype(np.float32)
    img[0, 0, :, : min(4, w + bd)] = 255                                                # top of the range, many ties
    wt = np.clip(rng.standard_normal((17 ** 4, outC)).astype(np.float32) * 0.5, -1.3, 1.3)
    G = rng.standard_normal((B, Cn * outC, h, w)).astype(np.float32)
    ro, rgw, rgi = oracle.swf2lut_interp(wt, outC, mode, img, bd, G)
    wv = torch.tensor(wt, device="cuda", requires_grad=True)
    xv = torch.tensor(img, device="cuda", requires_grad=True)
    o = _InterpFn.apply(wv, xv, outC, mode, bd)
    assert np.array_equal(o.detach().cpu().numpy(), ro)
    (o * torch.tensor(G, device="cuda")).sum().backward()
    assert _close(xv.grad.cpu().numpy(), rgi)
    assert _close(wv.grad.cpu().numpy(), rgw)


@pytest.mark.parametrize("name", ["lerf-g", "lerf-l"])
def test_predict_both_stages_golden(torch, golden, models, name):
    g = golden("g10_swf2lut.npz")
    m = models[name]
    x = torch.tensor(g["%s/predict/x" % name], device="cuda")
    with torch.no_grad():
        feat = m.predict(x, stage=1)
        hyper = m.predict(feat / 255.0, stage=2)
    assert np.array_equal(feat.cpu().numpy(), g["%s/predict/feat" % name])
    # the hyper maps are numerator / 255: torch's GPU division by a scalar multiplies by the reciprocal, which may
    # differ from the CPU quotient in the last bit -- compare the uint8 numerators exactly, the quotients to 1 ulp
    h, ref = hyper.cpu().numpy(), g["%s/predict/hyper" % name]
    assert np.array_equal(np.round(h * 255.0), np.round(ref * 255.0))
    assert np.max(np.abs(h - ref)) <= 6e-8


def test_mode_and_device_errors(torch, models):
    m = models["lerf-g"]
    x = torch.zeros(1, 1, 8, 8, device="cuda")
    with pytest.raises(ValueError, match="not implemented"):
        m.InterpTorchBatch(m.weight_s2_sr0, 3, "q", x, 1)
    with pytest.raises(ValueError):
        m.InterpTorchBatch(m.weight_s2_sr0.cpu(), 3, "s", x.cpu(), 1)
    with pytest.raises(ValueError):
        m.InterpTorchBatch(m.weight_s2_sr0, 3, "c", x, 1)            # pad smaller than the pattern reach


# ---------------------------------------------------------------- resampler backward + one training step
def _resizer(torch, kind, S):
    from lerf_pytorch_amd.resize_right.resize_right2d_torch import AmplifiedLinearResize2dTorch, SteeringGaussianResize2dTorch
    if kind == "gauss":
        return SteeringGaussianResize2dTorch(support_sz=S, device=torch.device("cuda"), max_sigma=10)
    return AmplifiedLinearResize2dTorch(support_sz=2, device=torch.device("cuda"))


@pytest.mark.parametrize("kind,ci", [("gauss", 0), ("gauss", 1), ("gauss", 2), ("gauss", 3), ("linear", 0), ("linear", 1), ("linear", 3)])
def test_resizer_gradients_golden(torch, golden, kind, ci):
    """autograd gradients of the reference's torch resamplers (same float32 geometry; float32 arithmetic on both sides)."""
    g = golden("g11_resize_grads.npz")
    pre = "%s/%d/" % (kind, ci)
    B, Cn, H, W, s, S = g[pre + "cfg"]
    B, Cn, H, W, S = int(B), int(Cn), int(H), int(W), int(S)
    r = _resizer(torch, kind, S)
    r.set_shape([B, Cn, H, W], scale_factors=[float(s), float(s)])
    x = torch.tensor(g[pre + "x"].astype(np.float32), device="cuda", requires_grad=True)
    hs = [torch.tensor(g[pre + "hy"][k], device="cuda", requires_grad=True) for k in range(3 if kind == "gauss" else 1)]
    o = r.resize(x, *hs)
    ref = g[pre + "out"]
    assert np.max(np.abs(o.detach().cpu().numpy() - ref)) <= 5e-4
    (o * torch.tensor(g[pre + "G"], device="cuda")).sum().backward()
    for got, want in [(x.grad, g[pre + "gx"])] + [(hs[k].grad, g[pre + "gh%d" % k]) for k in range(len(hs))]:
        want = np.asarray(want)
        assert np.max(np.abs(got.cpu().numpy() - want)) <= 1e-3 * max(np.abs(want).max(), 1.0)


@pytest.mark.parametrize("kind,S,shape,scale", [("gauss", 2, (3, 9, 7), 2.0), ("gauss", 4, (1, 8, 8), 3.0), ("linear", 2, (2, 6, 10), 1.7)])
def test_resizer_gradients_finite_differences(torch, kind, S, shape, scale):
    """the HIP backward against central differences of the float64 HIP forward (same geometry on both sides)."""
    from lerf_pytorch_amd import ops
    rng = np.random.default_rng(S + shape[1])
    N, H, W = shape
    geo = ops.SrGeometry((H, W), [scale, scale], None, S)
    x = rng.integers(0, 256, shape).astype(np.float32)
    nh = 3 if kind == "gauss" else 1
    hy = (0.2 + 0.6 * rng.random((nh,) + shape)).astype(np.float32)
    G = rng.standard_normal((N,) + geo.out_hw).astype(np.float32)
    ms = 10.0 if kind == "gauss" else 1.0

    def f64(xv, hv):
        o = ops.resize_planar(torch.tensor(xv, device="cuda"), [torch.tensor(h, device="cuda") for h in hv], geo, kind, ms, out="f64")
        return float((o * torch.tensor(G, device="cuda").double()).sum())

    from lerf_pytorch_amd.resize_right.resize_right2d_torch import _ResizeFn
    xt = torch.tensor(x, device="cuda", requires_grad=True)
    ht = [torch.tensor(hy[k], device="cuda", requires_grad=True) for k in range(nh)]
    (_ResizeFn.apply(geo, kind, ms, xt, *ht) * torch.tensor(G, device="cuda")).sum().backward()
    eps = 1e-3
    for _ in range(12):
        n, y, xx = rng.integers(0, N), rng.integers(0, H), rng.integers(0, W)
        for k in range(nh):
            hp, hm = hy.copy(), hy.copy()
            hp[k, n, y, xx] += eps
            hm[k, n, y, xx] -= eps
            fd = (f64(x, hp) - f64(x, hm)) / (2 * eps)
            an = float(ht[k].grad[n, y, xx])
            assert abs(fd - an) <= 2e-2 * max(1.0, abs(fd)), (kind, k, fd, an)
        xp, xm = x.copy(), x.copy()
        xp[n, y, xx] += 1.0
        xm[n, y, xx] -= 1.0
        fd = (f64(xp, hy) - f64(xm, hy)) / 2.0
        assert abs(fd - float(xt.grad[n, y, xx])) <= 1e-3 * max(1.0, abs(fd))


@pytest.mark.parametrize("name", ["lerf-g", "lerf-l"])
def test_training_step_gradients_golden(torch, golden, name):
    """train_model.py:416-441 (x2): predict stage 1 -> stage 2 -> resize -> clamp -> MSE; LUT gradients of the reference."""
    import torch.nn.functional as F
    g = golden("g10_swf2lut.npz")
    m = _model(torch, name)
    x = torch.tensor(g["%s/predict/x" % name], device="cuda")
    lb = torch.tensor(g["%s/step/lb" % name], device="cuda")
    feat = m.predict(x, stage=1)
    hyper = m.predict(feat / 255.0, stage=2)
    r = _resizer(torch, "gauss" if name == "lerf-g" else "linear", 2)
    r.set_shape([2, 1, 12, 10], scale_factors=2)
    pred = r.resize(feat, hyper[:, :1], hyper[:, 1:2], hyper[:, 2:]) if name == "lerf-g" else r.resize(feat, hyper)
    pred = torch.clamp(pred, 0, 255) / 255.0
    loss = F.mse_loss(pred, lb)
    loss.backward()
    assert abs(loss.item() - float(g["%s/step/loss" % name][0])) <= 1e-6
    assert np.max(np.abs(pred.detach().cpu().numpy() - g["%s/step/pred" % name])) <= 2e-5
    for key in ("weight_s1_sr0", "weight_s1_tr0", "weight_s2_cr1", "weight_s2_tr0"):
        gw = getattr(m, key).grad.cpu().numpy()
        rows, vals = g["%s/step/%s/rows" % (name, key)], g["%s/step/%s/vals" % (name, key)]
        scale = max(np.abs(vals).max(), 1e-12)
        assert np.max(np.abs(gw[rows] - vals)) <= 2e-3 * scale, key
        other = np.ones(gw.shape[0], bool)
        other[rows] = False
        assert np.max(np.abs(gw[other])) <= 1e-6 * scale, key


def test_finetune_steps_reduce_loss_and_export(torch, tmp_path, oracle):
    """A few Adam steps of lutft_step lower the loss on a fixed batch; export_luts writes int8 files the eval
    engine loads, and whose stage outputs equal the oracle's on the same files (train_model.py:416-442, 481-497)."""
    from lerf_pytorch_amd.resample.model import export_luts, lutft_step
    from lerf_pytorch_amd.resize_right.resize_right2d_torch import SteeringGaussianResize2dTorch
    import lerf_pytorch_amd as L
    m = _model(torch, "lerf-g")
    rng = np.random.default_rng(77)
    # smooth synthetic HR patches, LR = 2x2 box average
    hr = rng.random((4, 1, 6, 6)).astype(np.float32)
    hr = np.kron(hr, np.ones((1, 1, 8, 8), np.float32))
    hr = 0.5 * hr + 0.5 * np.roll(hr, 3, axis=3)
    lr = hr.reshape(4, 1, 24, 2, 24, 2).mean(axis=(3, 5))
    im, lb = torch.tensor(lr, device="cuda"), torch.tensor(hr, device="cuda")
    r = SteeringGaussianResize2dTorch(support_sz=2, device=torch.device("cuda"), max_sigma=10)
    r.set_shape([4, 1, 24, 24], scale_factors=2)
    opt_G = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    losses = [float(lutft_step(m, r, im, lb, opt_G).detach()) for _ in range(12)]
    assert all(np.isfinite(losses))
    assert losses[-1] < losses[0] * 0.98, losses
    paths = export_luts(m, str(tmp_path), "LUTft")
    assert len(paths) == 9 and all(np.load(p).dtype == np.int8 for p in paths)
    eng = L.LerfEngine(L.LutSet.from_dir(str(tmp_path), linear=False))
    img = rng.integers(0, 256, (20, 24, 3), dtype=np.uint8)
    feat, hq = eng.stages(img)
    rf, rh = oracle.lut_stages(img, oracle.load_luts(str(tmp_path), linear=False), 3)
    assert np.array_equal(feat, rf) and np.array_equal(hq, rh)
