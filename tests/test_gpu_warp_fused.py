"""GPU: the tile-fused warp (lerf_warp_fused_u8: s1_kernel, then stage 2 + the warp of each source tile's own output pixels from
LDS) against the packed-map path (lerf_stages_packed_u8 + lerf_warp_packed, itself pinned to the reference by the g4 goldens,
the Set5 warp md5s and the oracle) -- same bytes for in-scale and out-of-scale homographies, rotations, batches, both models --
and the host's tile boxes against a brute-force ownership map."""
import json
import os

import numpy as np
import pytest

from conftest import DATA, GOLDEN

pytestmark = pytest.mark.gpu

M_ISC = [[2.05, 0.12, 15.0], [-0.08, 1.95, 40.0], [1.5e-5, -1.0e-5, 1.0]]
M_OSC = [[4.1, 0.4, 30.0], [0.5, 3.8, 25.0], [8e-5, 1.2e-4, 1.0]]
M_ROT = [[0.9, -1.7, 700.0], [1.6, 1.1, -60.0], [2e-5, 4e-5, 1.0]]          # a rotation by ~60 degrees with perspective
M_SHRINK = [[1.2, 0.0, -40.0], [0.05, 1.1, 30.0], [0.0, 0.0, 1.0]]


@pytest.fixture(scope="module")
def torch():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch


@pytest.mark.parametrize("model", ["lerf-g", "lerf-l"])
@pytest.mark.parametrize("case", [("isc", M_ISC, (300, 420), (600, 840)), ("osc", M_OSC, (300, 420), (600, 840)),
                                  ("rot", M_ROT, (257, 391), (500, 700)), ("shrink", M_SHRINK, (130, 190), (180, 260))])
def test_fused_warp_equals_packed_path(torch, model, case):
    import lerf_pytorch_amd as L
    from lerf_pytorch_amd import ops
    name, M, (H, W), out_hw = case
    eng = L.LerfEngine.shipped(model)
    rng = np.random.default_rng(len(name) + H)
    x = torch.from_numpy(rng.integers(0, 256, (3, H, W, 3), dtype=np.uint8)).cuda()
    geo = ops.WarpGeometry((H, W), np.array(M), out_hw, 2)
    assert ops.warp_fused_supported(x, eng.luts, geo, eng.kind, eng.max_sigma)
    want = ops.warp_packed(ops.stages_packed(x, eng.luts), geo, eng.kind, eng.max_sigma, out="u8")
    got = ops.warp_fused_u8(x, eng.luts, geo, eng.kind, eng.max_sigma)
    assert got.shape == want.shape and torch.equal(got, want), (name, int((got != want).sum()))
    one = ops.warp_fused_u8(x[1], eng.luts, geo, eng.kind, eng.max_sigma)              # a single frame
    assert torch.equal(one, want[1])


def test_fused_warp_set5_md5(torch):
    """the reference's own Set5 warp outputs (md5 of the masked uint8 image, tests/golden/g5_set5.json) through the engine,
    which takes the tile-fused kernel"""
    import hashlib
    from PIL import Image
    import lerf_pytorch_amd as L
    ref = json.load(open(os.path.join(GOLDEN, "g5_set5.json")))["warp"]
    eng = L.LerfEngine.shipped("lerf-g")
    eng.fused_warp = True
    for p in ("isc", "osc"):
        for n in ("baby", "butterfly", "woman"):
            lr = np.array(Image.open(os.path.join(DATA, p, n + ".png")))
            gt = np.array(Image.open(os.path.join(DATA, "HR", n + ".png")))
            e = ref["lerf-g/%s/%s" % (p, n)]
            o, mask = eng.warp(lr, np.array(e["matrix"]), gt.shape[:2])
            eng.fused_warp = False
            o2, mask2 = eng.warp(lr, np.array(e["matrix"]), gt.shape[:2])
            eng.fused_warp = True
            assert np.array_equal(o, o2) and np.array_equal(mask, mask2)
            assert hashlib.md5(np.ascontiguousarray(o * mask).tobytes()).hexdigest() == e["md5_out_masked"]


def test_fused_warp_full_size_config4(torch):
    """BASELINE config 4 at its real size (1080p -> 4K, 2 frames): tile-fused == packed path, both matrices"""
    import lerf_pytorch_amd as L
    from lerf_pytorch_amd import ops
    eng = L.LerfEngine.shipped("lerf-g")
    rng = np.random.default_rng(4)
    x = torch.from_numpy(rng.integers(0, 256, (2, 1080, 1920, 3), dtype=np.uint8)).cuda()
    for M in (M_ISC, M_OSC):
        geo = ops.WarpGeometry((1080, 1920), np.array(M), (2160, 3840), 2)
        want = ops.warp_packed(ops.stages_packed(x, eng.luts), geo, "gauss", 10.0, out="u8")
        got = ops.warp_fused_u8(x, eng.luts, geo, "gauss", 10.0)
        assert torch.equal(got, want)


def test_tile_boxes_bound_every_owner(oracle):
    """lerf_warp_tile_boxes: every output pixel lies inside the box of the tile that owns it (ownership restated in numpy:
    key = min(first tap + 1, n - 1) of the clamped taps, resize_right2d_numpy.py:338-339, 396-398)"""
    from lerf_pytorch_amd import ops
    for M, (H, W), (oH, oW) in ((M_ISC, (200, 300), (400, 600)), (M_OSC, (200, 300), (400, 600)), (M_ROT, (257, 391), (300, 333))):
        geo = ops.WarpGeometry((H, W), np.array(M), (oH, oW), 2)
        nt_x = (W + 63) // 64
        b = np.zeros((((H + 63) // 64) * nt_x, 4), np.int32)
        from lerf_pytorch_amd import _lib
        _lib.check(_lib.lib().lerf_warp_tile_boxes(geo.ref(), H, W, b.ctypes.data))
        ii, jj = np.meshgrid(np.arange(oH), np.arange(oW), indexing="ij")
        mi = geo.minv
        X = mi[0, 0] * jj + mi[0, 1] * ii + mi[0, 2]
        Y = mi[1, 0] * jj + mi[1, 1] * ii + mi[1, 2]
        Wh = mi[2, 0] * jj + mi[2, 1] * ii + mi[2, 2]
        gr, gc = np.clip(Y / Wh, 0, H), np.clip(X / Wh, 0, W)
        eps = np.finfo(np.float32).eps
        pr, pc = geo.pad_vec[1][0], geo.pad_vec[2][0]
        lr = np.ceil(gr - 1.0 - eps).astype(int) + pr
        lc = np.ceil(gc - 1.0 - eps).astype(int) + pc
        r0 = np.clip(np.clip(lr, 0, H - 1) - pr, 0, H - 1)
        c0 = np.clip(np.clip(lc, 0, W - 1) - pc, 0, W - 1)
        t = (np.minimum(r0 + 1, H - 1) // 64) * nt_x + np.minimum(c0 + 1, W - 1) // 64
        bb = b[t]
        assert ((ii >= bb[..., 0]) & (ii < bb[..., 1]) & (jj >= bb[..., 2]) & (jj < bb[..., 3])).all()


@pytest.mark.parametrize("case", [("isc", M_ISC, (300, 420), (600, 840), 4), ("osc", M_OSC, (300, 420), (600, 840), 8), ("rot", M_ROT, (257, 391), (500, 700), 3)])
def test_warp_row_partition_single_gpu(torch, case):
    """dist.WarpRowPlan / warp_rows with the product kernels, every rank's launches in turn on one device: each rank sees only
    its band of the input (lerf_warp_geo_t out_y0 / src_y0); the stitched rows and masks are the whole-frame warp's, byte for byte"""
    import lerf_pytorch_amd as L
    from lerf_pytorch_amd import dist as ldist
    name, M, (H, W), out_hw, world = case
    eng = L.LerfEngine.shipped("lerf-g")
    rng = np.random.default_rng(world)
    img = torch.from_numpy(rng.integers(0, 256, (H, W, 3), dtype=np.uint8)).cuda()
    want, wmask = eng.warp(img, np.array(M), out_hw)
    rows, masks = [], []
    for r in range(world):
        plan = ldist.WarpRowPlan(H, W, M, out_hw, world, r, 2)
        o, m = ldist.warp_rows(eng, img[plan.b0:plan.b1].contiguous(), plan)
        assert o.shape[0] == plan.i1 - plan.i0
        rows.append(o)
        masks.append(m)
    assert torch.equal(torch.cat(rows, 0), want) and torch.equal(torch.cat(masks, 0), wmask)
