"""GPU: the LDS-resident form of the single LUT pass (csrc/lerf_lut_interp.hip, lerf_lut_interp_ex of ABI 7) against the oracle
(oracle.lut_interp_numer, pinned to the reference's FourSimplexInterpFaster by tests/test_oracle_golden.py), against the
reference's own raw per-pass outputs (g1) and against the direct kernel -- every operand layout, rotation, pattern, channel
count, frame sizes with partial tiles, and the accumulate form."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch


def _lut(torch, luts_g, oC):
    return torch.from_numpy(np.ascontiguousarray(luts_g["s2_cr1" if oC == 3 else "s1_tr0"].reshape(-1, oC))).cuda()


def _operands(torch, img8):
    """the image [C,H,W] as uint8 planar, float32 planar, float32 HWC buffer viewed as CHW (what the call sites hand over,
    resample/eval_lut_sr.py:551-553), and two odd views (a column-sliced plane, a channel-padded HWC buffer)"""
    x8 = torch.from_numpy(img8).cuda()
    xf = x8.to(torch.float32)
    xv = torch.from_numpy(np.ascontiguousarray(img8.transpose(1, 2, 0))).cuda().to(torch.float32).permute(2, 0, 1)
    C, H, W = img8.shape
    wide = torch.zeros((C, H, 2 * W), dtype=torch.float32, device="cuda")
    wide[:, :, ::2] = xf
    xs = wide[:, :, ::2]                                                     # sx = 2
    padc = torch.zeros((H, W, C + 1), dtype=torch.uint8, device="cuda")
    padc[:, :, :C] = x8.permute(1, 2, 0)
    xp = padc[:, :, :C].permute(2, 0, 1)                                      # sc = 1, sx = C + 1
    return {"u8": x8, "f32": xf, "hwc": xv, "sliced": xs, "padc": xp}


@pytest.mark.parametrize("oC", [1, 3])
@pytest.mark.parametrize("hw", [(37, 53), (130, 200), (64, 64), (1, 300), (257, 65)])
def test_lds_kernel_vs_oracle_all_layouts(torch, oracle, luts_g, oC, hw):
    from lerf_pytorch_amd import _lib, ops
    h, w = hw
    rng = np.random.default_rng(h * 1000 + w + oC)
    lut = _lut(torch, luts_g, oC)
    lut_np = lut.cpu().numpy()
    img8 = rng.integers(0, 256, (3, h + 3, w + 3), dtype=np.uint8)
    ops_in = _operands(torch, img8)
    for mode in "sctdy":
        dy, dx = _lib.mode_offsets(mode, 0)
        # the oracle on the padded frame: positions (y, x) < (h, w) never clamp, like the kernel's
        want = oracle.lut_interp_numer(lut_np, img8.transpose(1, 2, 0), mode, 0)[:h, :w]        # [h,w,C,oC]
        want = want.transpose(2, 3, 0, 1).reshape(3 * oC, h, w)
        for name, x in ops_in.items():
            got = ops.lut_interp(x, h, w, dy, dx, lut, 4, out_dtype=torch.int16, kernel="lds").cpu().numpy()
            assert np.array_equal(got, want), (mode, name)
        for rot in (1, 2, 3):
            ref = np.rot90(want, rot, [1, 2]) / 16.0
            got = ops.lut_interp(ops_in["hwc"], h, w, dy, dx, lut, 4, rot=rot, kernel="lds").cpu().numpy()
            assert got.dtype == np.float64 and np.array_equal(got, ref), (mode, rot)
            got32 = ops.lut_interp(ops_in["u8"], h, w, dy, dx, lut, 4, rot=rot, out_dtype=torch.float32, kernel="lds").cpu().numpy()
            assert np.array_equal(got32, ref.astype(np.float32))


@pytest.mark.parametrize("C", [1, 2, 4])
def test_lds_kernel_channel_counts_and_rotated_patterns(torch, oracle, luts_g, C):
    """C = 1, 2, 4 and the ROTATED patterns (negative offsets, clamped at the frame's border like the fused stages)"""
    from lerf_pytorch_amd import _lib, ops
    rng = np.random.default_rng(C)
    h, w = 150, 131
    img8 = rng.integers(0, 256, (C, h, w), dtype=np.uint8)
    ops_in = _operands(torch, img8)
    for oC in (1, 3):
        lut = _lut(torch, luts_g, oC)
        lut_np = lut.cpu().numpy()
        for mode, r in (("s", 1), ("c", 2), ("t", 3), ("c", 1)):
            dy, dx = _lib.mode_offsets(mode, r)
            want = oracle.lut_interp_numer(lut_np, img8.transpose(1, 2, 0), mode, r).transpose(2, 3, 0, 1).reshape(C * oC, h, w)
            for name in ("u8", "hwc", "f32"):
                got = ops.lut_interp(ops_in[name], h, w, dy, dx, lut, 4, out_dtype=torch.int16, kernel="lds").cpu().numpy()
                assert np.array_equal(got, want), (oC, mode, r, name)


def test_lds_kernel_reference_raw_passes(torch, golden, luts_g):
    """the reference's own raw per-pass outputs (g1, FourSimplexInterpFaster of resample/eval_lut_sr.py:24-470) through the
    LDS kernel, operands built like the call sites build them (:549-553)"""
    from lerf_pytorch_amd import _lib, ops
    from lerf_pytorch_amd.resample.eval_lut_sr import mode_pad_dict
    g = golden("g1_lut_stages.npz")
    img = g["lerf-g/noise24x20/img"].astype(np.float32)
    feat = g["lerf-g/noise24x20/feat"].astype(np.float32)
    for stage, src in ((1, img), (2, feat)):
        for mode in "sct":
            pad = mode_pad_dict[mode]
            dy, dx = _lib.mode_offsets(mode, 0)
            for r in range(4):
                key = "s1_%sr0" % mode if stage == 1 else "s2_%sr%d" % (mode, r & 1)
                oC = 1 if stage == 1 else 3
                rot = np.rot90(src, r)
                h, w, _ = rot.shape
                img_in = np.pad(rot, ((0, pad), (0, pad), (0, 0)), mode="edge")
                x = torch.from_numpy(np.ascontiguousarray(img_in)).cuda().permute(2, 0, 1)
                lut = torch.from_numpy(np.ascontiguousarray(luts_g[key].reshape(-1, oC))).cuda()
                out = ops.lut_interp(x, h, w, dy, dx, lut, 4, rot=4 - r, kernel="lds").cpu().numpy()
                ref = g["lerf-g/noise24x20/raw/s%d_%s_r%d" % (stage, mode, r)].transpose(2, 3, 0, 1).reshape(3 * oC, 24, 20) / 16.0
                assert np.array_equal(out, ref)


@pytest.mark.parametrize("oC", [1, 3])
def test_lds_kernel_equals_direct_at_1080p_and_accumulates(torch, luts_g, oC):
    """full size (what the call sites run 24 times per frame): LDS kernel == direct kernel; accumulate == a separate add; the
    library picks the LDS kernel by itself at this size (same values either way)"""
    from lerf_pytorch_amd import _lib, ops
    rng = np.random.default_rng(5 + oC)
    h, w = 1080, 1920
    lut = _lut(torch, luts_g, oC)
    hwc = torch.from_numpy(rng.integers(0, 256, (h + 3, w + 3, 3), dtype=np.uint8)).cuda().to(torch.float32)
    x = hwc.permute(2, 0, 1)
    for mode, rot in (("s", 0), ("c", 1), ("t", 2), ("s", 3)):
        dy, dx = _lib.mode_offsets(mode, 0)
        a = ops.lut_interp(x, h, w, dy, dx, lut, 4, rot=rot, kernel="lds")
        b = ops.lut_interp(x, h, w, dy, dx, lut, 4, rot=rot, kernel="direct")
        c = ops.lut_interp(x, h, w, dy, dx, lut, 4, rot=rot)
        assert torch.equal(a, b) and torch.equal(a, c)
        acc = a.clone()
        for kern in ("lds", "direct"):
            ops.lut_interp(x, h, w, dy, dx, lut, 4, rot=rot, out=acc, accumulate=True, kernel=kern)
        assert torch.equal(acc, a * 3)
    with pytest.raises(ValueError):
        ops.lut_interp(x, h, w, dy, dx, lut, 4, accumulate=True)
    # what the LDS kernel does not cover is refused when it is insisted on, and served by the direct kernel otherwise
    far_dy, far_dx = np.array([0, 0, 4, 4], np.int8), np.array([0, 4, 0, 4], np.int8)
    with pytest.raises(_lib.LerfError):
        ops.lut_interp(x, h - 1, w - 1, far_dy, far_dx, lut, 4, kernel="lds")
    ops.lut_interp(x, h - 1, w - 1, far_dy, far_dx, lut, 4)


def test_lds_kernel_rounds_and_clips_float_pixels_like_the_direct_kernel(torch, luts_g):
    """non-integer / out-of-range float pixels: rounded half-to-even and clipped to 0..255 in the tile staging (v_rndne +
    the saturating v_cvt_pk_u8_f32) exactly like pixel_value<float> of the direct kernel"""
    from lerf_pytorch_amd import _lib, ops
    lut = _lut(torch, luts_g, 1)
    vals = torch.tensor([-3.0, 0.5, 1.5, 2.5, 254.5, 255.5, 300.0, 7.49, 1e9, -1e9, 127.5, 128.5, -0.4, 0.49999, 255.49, 1e-30],
                        device="cuda")
    g = torch.Generator(device="cuda").manual_seed(3)
    idx = torch.randint(0, vals.numel(), (3, 140, 200), device="cuda", generator=g)
    planar = vals[idx].contiguous()
    hwc = planar.permute(1, 2, 0).contiguous().permute(2, 0, 1)
    for x in (planar, hwc):
        for mode in "sct":
            dy, dx = _lib.mode_offsets(mode, 0)
            a = ops.lut_interp(x, 137, 197, dy, dx, lut, 4, out_dtype=torch.int16, kernel="lds")
            b = ops.lut_interp(x, 137, 197, dy, dx, lut, 4, out_dtype=torch.int16, kernel="direct")
            c = ops.lut_interp(x.round().clamp(0, 255).to(torch.uint8), 137, 197, dy, dx, lut, 4, out_dtype=torch.int16, kernel="lds")
            assert torch.equal(a, b) and torch.equal(a, c)


def test_planar_lut_form(torch, luts_g):
    """LERF_INTERP_LUT_PLANAR: the LDS kernel fed its own plane layout == the interleaved table; the direct kernel (small
    launch, library's choice) is served from the interleaved table as before"""
    from lerf_pytorch_amd import _lib, ops
    lut = _lut(torch, luts_g, 3)
    planes = ops.lut_planes(lut)
    assert tuple(planes.shape) == (3, 83584)
    rng = np.random.default_rng(11)
    for (h, w) in ((300, 500), (20, 30)):
        x = torch.from_numpy(rng.integers(0, 256, (h + 3, w + 3, 3), dtype=np.uint8)).cuda().to(torch.float32).permute(2, 0, 1)
        for mode, rot in (("s", 0), ("t", 1)):
            dy, dx = _lib.mode_offsets(mode, 0)
            want = ops.lut_interp(x, h, w, dy, dx, lut, 4, rot=rot, kernel="direct")
            assert torch.equal(ops.lut_interp(x, h, w, dy, dx, lut, 4, rot=rot, planes=planes), want)
            assert torch.equal(ops.lut_interp(x, h, w, dy, dx, lut, 4, rot=rot, planes=planes, kernel="lds"), want)
