"""The unchanged-call-site path: the reference's _worker protocol (tools/callsite_driver.py:
24 FourSimplexInterpFaster calls + set_shape + resize, numpy calls in between, resample/eval_lut_sr.py:541-665) driven
against the mirrored names.  The bytes must be the reference's own (g5 md5s of its Set5 outputs), with the device-backed
lazy arrays (lerf_pytorch_amd.lazy) and with plain numpy results alike; and DeviceArray must answer the numpy operations
of the call sites exactly as numpy does."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest
from PIL import Image

from conftest import ASSETS, DATA, GOLDEN

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch as t
    assert t.cuda.is_available(), "GPU tests need an MI355X"
    return t


def _md5(a):
    return hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("lazy_on", [True, False])
@pytest.mark.parametrize("model,scale", [("lerf-g", 2), ("lerf-g", 3), ("lerf-l", 2), ("lerf-l", 4)])
def test_worker_protocol_gives_the_reference_bytes(torch, oracle, model, scale, lazy_on):
    import callsite_driver as cd
    from lerf_pytorch_amd import lazy
    ref = json.load(open(os.path.join(GOLDEN, "g5_set5.json")))["sr"]
    linear = model == "lerf-l"
    luts = cd.float_luts(oracle.load_luts(os.path.join(ASSETS, model), linear=linear))
    interp, pads, resizer = cd.mirror_api(linear=linear)
    lazy.set_enabled(lazy_on)
    try:
        for n in ("baby", "butterfly", "woman"):
            lr = np.array(Image.open(os.path.join(DATA, "LR_bicubic/rrLR_X%.2f_%.2f" % (scale, scale), n + ".png"))).astype(np.float32)
            out = cd.worker_sr(interp, pads, resizer, luts, lr, (scale, scale), out_c=1 if linear else 3, linear=linear)
            assert isinstance(out, lazy.DeviceArray) == lazy_on
            o8 = np.asarray(out)
            assert o8.dtype == np.uint8 and list(o8.shape) == ref["%s/x%d/%s" % (model, scale, n)]["shape"]
            assert _md5(o8) == ref["%s/x%d/%s" % (model, scale, n)]["md5_out"]
            assert np.array_equal(np.array(Image.fromarray(out)), o8)            # PIL takes it through __array_interface__
    finally:
        lazy.set_enabled(True)


@pytest.mark.parametrize("lazy_on", [True, False])
@pytest.mark.parametrize("model,p", [("lerf-g", "isc"), ("lerf-l", "osc")])
def test_warp_worker_protocol_gives_the_reference_bytes(torch, oracle, model, p, lazy_on):
    """the warp harness' call sequence (resample/eval_lut_warp.py:100-233) against the mirrors: masked output md5 and mask md5 of
    the reference's own Set5 runs"""
    import callsite_driver as cd
    from lerf_pytorch_amd import lazy
    ref = json.load(open(os.path.join(GOLDEN, "g5_set5.json")))["warp"]
    linear = model == "lerf-l"
    luts = cd.float_luts(oracle.load_luts(os.path.join(ASSETS, model), linear=linear))
    interp, pads, warper, nn = cd.mirror_warp_api(linear=linear)
    lazy.set_enabled(lazy_on)
    try:
        for n in ("bird", "head"):
            r = ref["%s/%s/%s" % (model, p, n)]
            lr = np.array(Image.open(os.path.join(DATA, p, n + ".png"))).astype(np.float32)
            gt = np.array(Image.open(os.path.join(DATA, "HR", n + ".png")))
            out8, mask = cd.worker_warp(interp, pads, warper, nn, luts, lr, np.array(r["matrix"]), gt.shape[:2],
                                        out_c=1 if linear else 3, linear=linear)
            o8, mk = np.asarray(out8), np.asarray(mask)
            assert int(mk.sum()) == r["mask_sum"] and _md5(mk.astype(np.uint8)) == r["md5_mask"]
            assert _md5(o8 * mk) == r["md5_out_masked"]
    finally:
        lazy.set_enabled(True)


@pytest.mark.parametrize("lazy_on", [True, False])
@pytest.mark.parametrize("model,scale", [("lerf-g", 2), ("lerf-g", 3), ("lerf-l", 4)])
def test_whole_sr_worker_to_its_last_statement(torch, oracle, tmp_path, model, scale, lazy_on):
    """the SR worker to its END (resample/eval_lut_sr.py:514-744: the library calls, then Image.fromarray(...).save,
    np.save of the hyper maps, the crop, the colour transform with its item assignment, PSNR, SSIM), lazy results on and off:
    the reference's md5 of the output, its PSNR (float32 arithmetic: equal) and SSIM, and files that read back as the values."""
    import callsite_driver as cd
    from lerf_pytorch_amd import lazy
    ref = json.load(open(os.path.join(GOLDEN, "g5_set5.json")))["sr"]
    ref8 = json.load(open(os.path.join(GOLDEN, "g8_ssim.json")))
    linear = model == "lerf-l"
    luts = cd.float_luts(oracle.load_luts(os.path.join(ASSETS, model), linear=linear))
    interp, pads, resizer = cd.mirror_api(linear=linear)
    lazy.set_enabled(lazy_on)
    try:
        for n in ("bird", "head"):
            key = "%s/x%d/%s" % (model, scale, n)
            lr = np.array(Image.open(os.path.join(DATA, "LR_bicubic/rrLR_X%.2f_%.2f" % (scale, scale), n + ".png"))).astype(np.float32)
            gt = np.array(Image.open(os.path.join(DATA, "HR", n + ".png")))
            keep = {}
            out = cd.worker_sr(interp, pads, resizer, luts, lr, (scale, scale), out_c=1 if linear else 3, linear=linear, keep=keep)
            assert isinstance(out, lazy.DeviceArray) == lazy_on
            ps, ss = cd.tail_sr(out, keep["feat_chw"], keep["hyper"], gt, (scale, scale), str(tmp_path), n)
            assert float(ps) == pytest.approx(ref[key]["psnr_y"], abs=1e-5) and float(ss) == pytest.approx(ref8[key]["ssim"], abs=1e-9)
            saved = np.array(Image.open(os.path.join(str(tmp_path), n + "_LUTft.png")))
            assert _md5(saved) == ref[key]["md5_out"]
            assert _md5(np.array(Image.open(os.path.join(str(tmp_path), n + "_lr.png")))) == ref[key]["md5_feat"]
            hy = np.load(os.path.join(str(tmp_path), n + "_LUTft_hyper.npy"))
            assert hy.dtype == np.float32 and _md5(np.round(hy * 255).astype(np.uint8)) == ref[key]["md5_hq"]
    finally:
        lazy.set_enabled(True)


@pytest.mark.parametrize("lazy_on", [True, False])
@pytest.mark.parametrize("model,p", [("lerf-g", "isc"), ("lerf-g", "osc"), ("lerf-l", "isc")])
def test_whole_warp_worker_to_its_last_statement(torch, oracle, tmp_path, model, p, lazy_on):
    """the warp worker to its END (resample/eval_lut_warp.py:70-302): torch.Tensor(img_out) -- the statement that raised on the
    round-4 DeviceArray --, np.array(mask_output == 255), mPSNR on float32 tensors, the white fill by boolean arithmetic and the
    PNGs.  mPSNR equals the reference's number (its float32 sum runs in the same memory order: lazy.download keeps numpy's layout)."""
    import callsite_driver as cd
    from lerf_pytorch_amd import lazy
    ref = json.load(open(os.path.join(GOLDEN, "g5_set5.json")))["warp"]
    linear = model == "lerf-l"
    luts = cd.float_luts(oracle.load_luts(os.path.join(ASSETS, model), linear=linear))
    interp, pads, warper, nn = cd.mirror_warp_api(linear=linear)
    lazy.set_enabled(lazy_on)
    try:
        for n in ("baby", "woman"):
            r = ref["%s/%s/%s" % (model, p, n)]
            lr = np.array(Image.open(os.path.join(DATA, p, n + ".png"))).astype(np.float32)
            gt = np.array(Image.open(os.path.join(DATA, "HR", n + ".png")))
            keep = {}
            out8, mask = cd.worker_warp(interp, pads, warper, nn, luts, lr, np.array(r["matrix"]), gt.shape[:2],
                                        out_c=1 if linear else 3, linear=linear, keep=keep)
            assert isinstance(out8, lazy.DeviceArray) == lazy_on and isinstance(keep["mask_output"], lazy.DeviceArray) == lazy_on
            (mp,) = cd.tail_warp(out8, keep["mask_output"], keep["feat_chw"], gt, str(tmp_path), n)
            assert float(mp) == pytest.approx(r["mpsnr"], abs=2e-5)
            mk = np.array(Image.open(os.path.join(str(tmp_path), n + "_mask.png")))
            assert _md5((mk == 255).astype(np.uint8)) == r["md5_mask"] and int((mk == 255).sum()) == r["mask_sum"]
            saved = np.array(Image.open(os.path.join(str(tmp_path), n + "_LUTft.png")))
            assert _md5(saved * (mk == 255)) == r["md5_out_masked"] and bool((saved[mk != 255] == 255).all())
    finally:
        lazy.set_enabled(True)


def test_device_results_take_everything_the_callers_do(torch):
    """the conversions and writes of tests/test_lazy_cpu.py on real device tensors"""
    from lerf_pytorch_amd import lazy
    rng = np.random.default_rng(2)
    a = rng.integers(0, 256, (270, 480, 3), dtype=np.uint8)
    A = lazy.asdevice(a)
    assert A.t.is_cuda
    x = torch.Tensor(A)
    assert x.device.type == "cpu" and x.dtype == torch.float32 and torch.equal(x, torch.Tensor(a))
    assert torch.equal(torch.tensor(A), torch.tensor(a)) and torch.equal(torch.as_tensor(A), torch.as_tensor(a))
    f = rng.standard_normal((9, 270, 480))
    F = lazy.asdevice(f)
    f = f.copy()
    F[f > 1.0] = 0.0
    f[f > 1.0] = 0.0
    F[:, 5:9, ::2] = 3.0
    f[:, 5:9, ::2] = 3.0
    v, vn = F.transpose((1, 2, 0)), f.transpose((1, 2, 0))
    v[:, 0] += 16.0
    vn[:, 0] += 16.0
    F *= 0.5
    f *= 0.5
    assert np.array_equal(np.asarray(F), f) and np.array_equal(np.asarray(v), vn)
    O = lazy.asdevice(np.zeros_like(f))
    np.clip(F, -1, 1, out=O)
    assert np.array_equal(O.t.cpu().numpy(), np.clip(f, -1, 1))
    i = lazy.asdevice(np.arange(12, dtype=np.int32).reshape(3, 4))
    assert (i * 0.5).dtype == np.float64 and np.array_equal(np.asarray(i * 0.5), np.arange(12).reshape(3, 4) * 0.5)
    m = np.asarray(v == 3.0)
    assert m.strides == (vn == 3.0).strides                 # numpy's memory layout for results of transposed views


def test_device_array_answers_like_numpy(torch):
    """every operation the call sites apply between the library calls, DeviceArray against the same numpy expression"""
    from lerf_pytorch_amd import lazy
    rng = np.random.default_rng(0)
    a64 = rng.integers(-2032, 2033, (9, 37, 41)).astype(np.float64) / 16.0
    b64 = rng.integers(-2032, 2033, (9, 37, 41)).astype(np.float64) / 16.0
    A, B = lazy.asdevice(a64), lazy.asdevice(b64)
    eq = lambda d, n: isinstance(d, lazy.DeviceArray) and d.dtype == n.dtype and d.shape == n.shape and np.array_equal(np.asarray(d), n)
    p_d, p_n = 0, 0
    p_d += A
    p_n += a64
    p_d += B
    p_n = p_n + b64
    assert eq(p_d, p_n)
    for avg, bias in ((3, 0), (12, 127)):
        d = np.round(np.clip((p_d / avg) + bias, 0, 255)).astype(np.float32).transpose((1, 2, 0))
        n = np.round(np.clip((p_n / avg) + bias, 0, 255)).astype(np.float32).transpose((1, 2, 0))
        assert eq(d, n)
        assert eq(d / float(255), n / float(255))
        for r in range(4):
            dr, nr = np.rot90(d, r), np.rot90(n, r)
            assert eq(dr, nr)
            assert eq(np.pad(dr, ((0, 3), (0, 3), (0, 0)), mode="edge").transpose((2, 0, 1)),
                      np.pad(nr, ((0, 3), (0, 3), (0, 0)), mode="edge").transpose((2, 0, 1)))
    idx = list(range(1, 10, 3))
    assert eq(A[idx, :, :], a64[idx, :, :])
    assert eq(np.clip(np.round(A).transpose((1, 2, 0)), 0, 255).astype(np.uint8), np.clip(np.round(a64).transpose((1, 2, 0)), 0, 255).astype(np.uint8))
    # IEEE division, not a multiplication by the reciprocal: N / 48 at the ties of the round that follows, x / 255 in float32
    n = (np.arange(0, 4000, dtype=np.float64) * 24.0 - 24000.0) / 16.0
    assert eq(np.round(lazy.asdevice(n) / 3), np.round(n / 3)) and eq(lazy.asdevice(n) / 12 + 127, n / 12 + 127)
    q = np.arange(256, dtype=np.float32)
    assert eq(lazy.asdevice(q) / float(255), q / float(255)) and eq(lazy.asdevice(q) / np.float32(255), q / np.float32(255))
    # ties round to even, like np.round
    t = np.array([0.5, 1.5, 2.5, -0.5, 254.5, 255.5])
    assert eq(np.round(lazy.asdevice(t)), np.round(t))
    # everything else happens on the host copy, with numpy's own result
    assert float(A.max()) == a64.max() and np.allclose(np.mean(A), a64.mean()) and np.array_equal(np.abs(A), np.abs(a64))
    assert np.array_equal(np.dot(A[0], np.ones(41)), np.dot(a64[0], np.ones(41)))


def test_interp_accepts_device_arrays_and_caches_luts(torch, oracle):
    """a lazy result fed back as img_in (stage 2 of the call sites) == the numpy route; a LUT array mutated in place is
    uploaded afresh"""
    from lerf_pytorch_amd import lazy
    from lerf_pytorch_amd.resample.eval_lut_sr import FourSimplexInterpFaster
    luts = oracle.load_luts(os.path.join(ASSETS, "lerf-g"))
    w = luts["s2_cr1"].astype(np.float32)
    img = np.random.default_rng(3).integers(0, 256, (3, 30 + 3, 44 + 3)).astype(np.float32)
    lazy.set_enabled(False)
    try:
        want = FourSimplexInterpFaster(w, img, 30, 44, 4, 1, upscale=1, mode="c", oC=3)
    finally:
        lazy.set_enabled(True)
    assert isinstance(want, np.ndarray)
    got = FourSimplexInterpFaster(w, lazy.asdevice(img), 30, 44, 4, 1, upscale=1, mode="c", oC=3)
    assert isinstance(got, lazy.DeviceArray) and np.array_equal(np.asarray(got), want)
    assert np.array_equal(np.asarray(FourSimplexInterpFaster(w, img, 30, 44, 4, 1, upscale=1, mode="c", oC=3)), want)
    w[::7] = 5.0                                          # same buffer, new values: the cached device copy must not be used
    changed = FourSimplexInterpFaster(w, img, 30, 44, 4, 1, upscale=1, mode="c", oC=3)
    fresh = FourSimplexInterpFaster(w.copy(), img, 30, 44, 4, 1, upscale=1, mode="c", oC=3)     # another buffer: no cache hit possible
    assert np.array_equal(np.asarray(changed), np.asarray(fresh)) and not np.array_equal(np.asarray(changed), want)
    w[1000:1100] = 7.0                                    # a SPARSE change between the points any sample grid would look at
    changed = FourSimplexInterpFaster(w, img, 30, 44, 4, 1, upscale=1, mode="c", oC=3)
    fresh = FourSimplexInterpFaster(w.copy(), img, 30, 44, 4, 1, upscale=1, mode="c", oC=3)
    assert np.array_equal(np.asarray(changed), np.asarray(fresh))
    w[40000, 1] = -9.0                                    # one element
    changed = FourSimplexInterpFaster(w, img, 30, 44, 4, 1, upscale=1, mode="c", oC=3)
    fresh = FourSimplexInterpFaster(w.copy(), img, 30, 44, 4, 1, upscale=1, mode="c", oC=3)
    assert np.array_equal(np.asarray(changed), np.asarray(fresh))


def test_upload_ring_reuses_its_pinned_slots_safely(torch):
    """lazy.upload: five large arrays in a row through the two-slot pinned ring (a slot is rewritten only after its copy has
    finished), other dtypes and sizes, and the small-array path."""
    from lerf_pytorch_amd import lazy
    rng = np.random.default_rng(5)
    arrs = [rng.integers(0, 256, (3, 700, 900)).astype(np.float32) for _ in range(5)]
    devs = [lazy.upload(a) for a in arrs]                       # no synchronisation in between
    for a, d in zip(arrs, devs):
        assert d.dtype == torch.float32 and tuple(d.shape) == a.shape
        assert np.array_equal(d.cpu().numpy(), a)
    for dt in (np.uint8, np.int16, np.float64, np.int64):
        a = rng.integers(0, 100, (1200, 1100)).astype(dt)
        assert np.array_equal(lazy.upload(a).cpu().numpy(), a)
    v = arrs[0].transpose(1, 2, 0)                              # not contiguous: gathered on the host first
    assert np.array_equal(lazy.upload(v).cpu().numpy(), v)
    s = rng.integers(0, 9, (7, 5)).astype(np.float32)
    assert np.array_equal(lazy.upload(s).cpu().numpy(), s)
