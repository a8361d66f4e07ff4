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
