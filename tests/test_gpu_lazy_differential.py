"""GPU: lazy / deferred device results (lerf_pytorch_amd.lazy) against plain numpy results of the SAME calls -- usages the
reference's workers do not make but a caller could: results used twice, mutated after something was derived from them, views
taken before a later `+=`, operands overwritten while a deferred pass still wants to read them, sums started from host arrays,
results dropped unread.  Every scenario runs twice, lazy on (DeviceArray / LazyArray) and lazy off (float64 ndarrays, the
section-8(b) contract to the letter), and must give the same values and dtypes."""
import os
import sys

import numpy as np
import pytest

from conftest import ASSETS

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(oracle):
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import callsite_driver as cd
    luts = cd.float_luts(oracle.load_luts(os.path.join(ASSETS, "lerf-g"), linear=False))
    interp, pads, resizer = cd.mirror_api(linear=False)
    return interp, pads, resizer, luts


HW = [41, 57]          # set per test: small frames take the direct kernel, 260 x 330 the LDS-resident one (>= 65 536 positions)


def _img(seed):
    return np.random.default_rng(seed).integers(0, 256, (HW[0], HW[1], 3)).astype(np.float32)


def _pass(interp, pads, luts, img, key="s1_sr0", mode="s", r=0, oC=1):
    p = pads[mode]
    rot = np.rot90(img, r)
    h, w, _ = rot.shape
    chw = np.pad(rot, ((0, p), (0, p), (0, 0)), mode="edge").transpose((2, 0, 1))
    return interp(luts[key], chw, h, w, 4, 4 - r, upscale=1, mode=mode, oC=oC)


def s_used_twice(interp, pads, luts, dev):
    x = _pass(interp, pads, luts, dev(_img(1)))
    a = x + 1.5
    b = x * 2
    return [a, b, x]


def s_mutated_after_derivation(interp, pads, luts, dev):
    x = _pass(interp, pads, luts, dev(_img(2)))
    y = x + 1                      # derived first ...
    x += _pass(interp, pads, luts, dev(_img(2)), mode="c", key="s1_cr0")    # ... then the source changes
    x *= 0.5
    return [y, x]


def s_view_before_iadd(interp, pads, luts, dev):
    pred = _pass(interp, pads, luts, dev(_img(3)))
    v = pred[1]                    # a view in numpy: sees the later +=
    t = pred.transpose((1, 2, 0))
    pred += _pass(interp, pads, luts, dev(_img(3)), r=2)
    pred += 3
    return [v, t, pred]


def s_operand_overwritten_while_pending(interp, pads, luts, dev):
    img = dev(_img(4))
    x = _pass(interp, pads, luts, img)           # may still be pending ...
    img[:, :, 0] = 7.0                           # ... when its operand changes: the pass must have read the OLD pixels
    y = _pass(interp, pads, luts, img)
    return [x, y]


def s_sum_started_on_the_host(interp, pads, luts, dev):
    pred = np.zeros((3, HW[0], HW[1]))
    pred += _pass(interp, pads, luts, dev(_img(5)))
    pred += _pass(interp, pads, luts, dev(_img(5)), r=1)
    return [pred]


def s_int_start_and_reversed_operands(interp, pads, luts, dev):
    total = 0
    for r in (0, 1, 2, 3):
        total = _pass(interp, pads, luts, dev(_img(6)), r=r) + total      # not +=: a fresh sum each time
    z = 10 - total
    return [total, z, -total, abs(z)]


def s_stage_epilogue_variants(interp, pads, luts, dev):
    s = 0
    for m, k in (("s", "s1_sr0"), ("c", "s1_cr0"), ("t", "s1_tr0")):
        for r in (0, 1, 2, 3):
            s += _pass(interp, pads, luts, dev(_img(7)), key=k, mode=m, r=r)
    a = np.round(np.clip(s / 3 + 0, 0, 255)).astype(np.float32)
    b = np.clip(np.round(s / 3), 0, 255)                 # the other order
    c = (s / 3).astype(np.float32)
    d = np.round(s / 3 + 0.5, 1)                         # decimals: not the recognised chain
    return [a, b, c, d, s]


def s_results_dropped_unread(interp, pads, luts, dev):
    for r in (0, 1):
        _pass(interp, pads, luts, dev(_img(8)), r=r)     # never used
    x = _pass(interp, pads, luts, dev(_img(8)), r=3)
    del x
    return [_pass(interp, pads, luts, dev(_img(8)), r=2)]


def s_three_channel_lut_and_slices(interp, pads, luts, dev):
    feat = dev(_img(9))
    s2 = _pass(interp, pads, luts, feat, key="s2_sr0", oC=3)
    s2 += _pass(interp, pads, luts, feat, key="s2_sr1", r=1, oC=3)
    hyper = np.round(np.clip(s2 / 8 + 127, 0, 255)).astype(np.float32) / 255.0
    n = hyper.shape[0]
    return [hyper[list(range(0, n, 3)), :, :], hyper[1::3], hyper[:, 3:9, ::2], hyper.max(), hyper.sum(axis=0)]


def s_copy_semantics(interp, pads, luts, dev):
    x = _pass(interp, pads, luts, dev(_img(10)))
    c = x.copy()
    a = x.astype(np.float64)          # a copy even for the same dtype
    x -= 100
    return [c, a, x, np.array(x), np.asarray(x).dtype.str]


def _stages(interp, pads, luts, dev, seed):
    import callsite_driver as cd
    img = dev(_img(seed))
    s1 = cd.lut_ensemble(interp, pads, luts, img, 1, "sct", 1, 4, lambda r: "r0")
    feat = np.round(np.clip(s1 / 3 + 0, 0, 255)).astype(np.float32).transpose((1, 2, 0))
    s2 = cd.lut_ensemble(interp, pads, luts, feat, 2, "sct", 3, 4, lambda r: "r%d" % (r & 1))
    hyper = np.round(np.clip(s2 / 12 + 127, 0, 255)).astype(np.float32) / 255.0
    return feat.transpose((2, 0, 1)), hyper


def s_resize_tails(interp, pads, luts, dev, resizer):
    chw, hyper = _stages(interp, pads, luts, dev, 11)
    resizer.set_shape(chw.shape, scale_factors=[2.0, 1.5])
    n = hyper.shape[0]
    args = (hyper[list(range(0, n, 3)), :, :], hyper[list(range(1, n + 1, 3)), :, :], hyper[list(range(2, n + 2, 3)), :, :])
    a = np.clip(np.round(resizer.resize(chw, *args)).transpose((1, 2, 0)), 0, 255).astype(np.uint8)     # the worker's tail
    b = np.clip(np.round(resizer.resize(chw, *args)), 0, 255).astype(np.uint8)                          # no transpose
    c = resizer.resize(chw, *args)                                                                      # the float64 values
    d = resizer.resize(chw, *args).astype(np.uint8)                                                     # truncation, not rounding
    e = np.round(resizer.resize(chw, *args) * 0.5)                                                      # something else in between
    return [a, b, c, d, e]


def s_resize_after_operands_changed(interp, pads, luts, dev, resizer):
    chw, hyper = _stages(interp, pads, luts, dev, 12)
    resizer.set_shape(chw.shape, scale_factors=[2.0, 2.0])
    n = hyper.shape[0]
    out = resizer.resize(chw, hyper[0:n:3], hyper[1:n:3], hyper[2:n:3])          # maybe pending ...
    hyper *= 0.5                                                                   # ... when its operands change (and lose their
    chw += 1                                                                       #     exact uint8 form)
    out2 = resizer.resize(chw, hyper[0:n:3], hyper[1:n:3], hyper[2:n:3])
    t = lambda o: np.clip(np.round(o).transpose((1, 2, 0)), 0, 255).astype(np.uint8)
    return [t(out), t(out2)]


SCENARIOS = [s_used_twice, s_mutated_after_derivation, s_view_before_iadd, s_operand_overwritten_while_pending,
             s_sum_started_on_the_host, s_int_start_and_reversed_operands, s_stage_epilogue_variants,
             s_results_dropped_unread, s_three_channel_lut_and_slices, s_copy_semantics]
RESIZE_SCENARIOS = [s_resize_tails, s_resize_after_operands_changed]


@pytest.mark.parametrize("hw", [(41, 57), (260, 330)])
@pytest.mark.parametrize("device_image", [True, False])
@pytest.mark.parametrize("scenario", SCENARIOS, ids=lambda f: f.__name__[2:])
def test_lazy_results_equal_numpy_results(env, scenario, device_image, hw):
    from lerf_pytorch_amd import lazy
    interp, pads, resizer, luts = env
    HW[:] = hw
    dev = (lambda a: lazy.asdevice(a)) if device_image else (lambda a: a)
    lazy.set_enabled(False)
    try:
        want = scenario(interp, pads, luts, lambda a: a)                 # plain numpy all the way
    finally:
        lazy.set_enabled(True)
    got = scenario(interp, pads, luts, dev)
    assert len(got) == len(want)
    for k, (g, w) in enumerate(zip(got, want)):
        if isinstance(w, str):
            assert g == w, (k, g, w)
            continue
        gn, wn = np.asarray(g), np.asarray(w)
        assert gn.dtype == wn.dtype and gn.shape == wn.shape, (scenario.__name__, k, gn.dtype, wn.dtype, gn.shape, wn.shape)
        assert np.array_equal(gn, wn), (scenario.__name__, k, float(np.max(np.abs(gn.astype(np.float64) - wn.astype(np.float64)))))


@pytest.mark.parametrize("device_image", [True, False])
@pytest.mark.parametrize("scenario", RESIZE_SCENARIOS, ids=lambda f: f.__name__[2:])
def test_lazy_resize_chains_equal_numpy_results(env, scenario, device_image):
    from lerf_pytorch_amd import lazy
    interp, pads, resizer, luts = env
    HW[:] = (41, 57)
    lazy.set_enabled(False)
    try:
        want = scenario(interp, pads, luts, lambda a: a, resizer)
    finally:
        lazy.set_enabled(True)
    got = scenario(interp, pads, luts, (lambda a: lazy.asdevice(a)) if device_image else (lambda a: a), resizer)
    for k, (g, w) in enumerate(zip(got, want)):
        gn, wn = np.asarray(g), np.asarray(w)
        assert gn.dtype == wn.dtype and gn.shape == wn.shape, (scenario.__name__, k, gn.dtype, wn.dtype, gn.shape, wn.shape)
        if gn.dtype == np.float64:                               # float64 values of the resampler: the class contract is 1e-9
            np.testing.assert_allclose(gn, wn, rtol=0, atol=1e-9, err_msg="%s %d" % (scenario.__name__, k))
        else:
            assert np.array_equal(gn, wn), (scenario.__name__, k, int((gn != wn).sum()))
