"""GPU parity of the GENERAL tile-fused kernels (round 3): channel counts 1 / 3 / 4, any 1..4 sampling patterns per stage
(incl. 'd' and 'y'), scale factors up to x8, non-constant image padding, frames of different sizes in one launch, a region
of interest (2-D block partition), the batched packed warp and the ABI-4 argument checks.  Every case: tile-fused ==
direct kernels == the float64 oracle (resample/eval_lut_sr.py:24-470, 541-665; resize_right2d_numpy.py:18-282)."""
import os

import numpy as np
import pytest

from conftest import ASSETS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch as t
    assert t.cuda.is_available(), "GPU tests need an MI355X"
    return t


def _arrays(oracle, model, modes, modes2, seed=0):
    """a LUT dictionary for arbitrary pattern strings: the shipped s/c/t tables dealt out to the requested patterns
    (the pattern decides WHERE the four pixels are sampled, any table can be looked up with them)"""
    linear = model == "lerf-l"
    base = oracle.load_luts(os.path.join(ASSETS, model), linear=linear)
    src = "sct"
    d = {}
    for i, m in enumerate(modes):
        d["s1_%sr0" % m] = base["s1_%sr0" % src[(i + seed) % 3]]
    for i, m in enumerate(modes2):
        for r in (0, 1):
            d["s2_%sr%d" % (m, r)] = base["s2_%sr%d" % (src[(i + seed + 1) % 3], r)]
    return d


def _engine(model, arrays, modes, modes2, S=2):
    import lerf_pytorch_amd as L
    luts = L.LutSet.from_arrays(arrays, modes, modes2)
    assert luts.struct.fused_pack is not None
    return L.LerfEngine(luts, support=S)


CASES = [
    # C, model, modes, modes2, S, H, W, scale
    (1, "lerf-g", "sct", "sct", 2, 70, 200, 2),             # grey: 64 x 192 tiles
    (1, "lerf-g", "sct", "sct", 2, 64, 192, (3.0, 1.5)),
    (4, "lerf-g", "sct", "sct", 2, 70, 100, 2),             # RGBA: 64 x 48 tiles
    (4, "lerf-l", "sct", "sct", 2, 66, 50, (1.5, 2.0)),
    (4, "lerf-g", "sct", "sct", 4, 40, 60, 2),
    (3, "lerf-g", "sdy", "yct", 2, 70, 75, 2),              # patterns d and y in both stages
    (3, "lerf-l", "dy", "yd", 2, 65, 66, 3),
    (3, "lerf-g", "s", "c", 2, 33, 47, 2),                  # one pattern per stage
    (3, "lerf-g", "sdyc", "sdyt", 2, 30, 70, 2),            # four patterns per stage (the 16-bit sums' limit)
    (3, "lerf-g", "sct", "sct", 2, 70, 66, 6),              # scale > 4.9: large geometry tables
    (3, "lerf-g", "sct", "sct", 2, 20, 70, 8),
    (3, "lerf-g", "sct", "sct", 4, 66, 20, (7.3, 5.0)),
    (3, "lerf-l", "sct", "sct", 2, 64, 64, (8.0, 1.0)),
    (1, "lerf-l", "tcs", "y", 2, 9, 200, 5),
    # frames large enough for INTERIOR tiles (region + halo inside the frame): the vector-memory pixel path of the general
    # kernels (byte_phase_vmem_rt), stage 1 and the LeRF-L stage 2
    (3, "lerf-g", "sdy", "yct", 2, 210, 220, 2),
    (3, "lerf-l", "dyc", "ts", 2, 215, 205, (1.5, 2.0)),
    (1, "lerf-g", "sct", "sct", 2, 150, 600, 2),
    (1, "lerf-l", "yd", "cs", 2, 140, 620, 3),
    (4, "lerf-g", "tys", "dc", 2, 160, 170, 2),
    (4, "lerf-l", "sct", "sct", 2, 141, 163, 2),
    (3, "lerf-g", "sct", "sct", 2, 150, 215, 5),
]


@pytest.mark.parametrize("C,model,modes,modes2,S,H,W,scale", CASES)
def test_general_fused_equals_direct_and_oracle(torch, oracle, C, model, modes, modes2, S, H, W, scale):
    from lerf_pytorch_amd import ops
    arrays = _arrays(oracle, model, modes, modes2)
    eng = _engine(model, arrays, modes, modes2, S)
    rng = np.random.default_rng(H * 131 + W * 7 + C)
    img = rng.integers(0, 256, (H, W, C), dtype=np.uint8)
    sc = scale if isinstance(scale, tuple) else (scale, scale)
    geo = eng.sr_geometry((H, W), sc)
    assert ops.sr_fused_supported(C, eng.luts, geo, eng.kind), "this configuration must take the tile-fused kernels"
    a = eng.sr(img, sc, fused=True)
    b = eng.sr(img, sc, fused=False)
    assert np.array_equal(a, b), "fused != direct kernels"
    ref = oracle.sr_pipeline(img, arrays, sc[0], sc[1], S=S, linear=(model == "lerf-l"), modes=modes, modes2=modes2)
    assert a.shape == ref.shape
    assert np.array_equal(a, ref), "fused != oracle (%d bytes differ)" % int((a != ref).sum())
    # single launch (stage 1 recomputed per tile) == two launches
    x = torch.from_numpy(img).cuda()
    one = ops.sr_fused_u8(x, eng.luts, geo, eng.kind, eng.max_sigma, workspace=False).cpu().numpy()
    assert np.array_equal(one, a)
    # the packed stage outputs of the general EMIT kernels == the direct stage kernels
    f1, h1 = ops.unpack_stages(ops.stages_packed(x, eng.luts), eng.luts.oC)
    f2, h2 = ops.lut_stages(x, eng.luts)
    assert torch.equal(f1, f2) and torch.equal(h1, h2)


def test_five_patterns_fall_back_to_the_direct_kernels(torch, oracle):
    import lerf_pytorch_amd as L
    from lerf_pytorch_amd import ops
    arrays = _arrays(oracle, "lerf-g", "sdyct", "sct")
    luts = L.LutSet.from_arrays(arrays, "sdyct", "sct")
    assert luts.struct.fused_pack is None                     # five patterns overflow the packed 16-bit sums: no fused kernel
    eng = L.LerfEngine(luts)
    img = np.random.default_rng(3).integers(0, 256, (40, 50, 3), dtype=np.uint8)
    assert not ops.sr_fused_supported(3, luts, eng.sr_geometry((40, 50), 2), "gauss")
    out = eng.sr(img, 2)                                      # lerf_sr_fused_u8 routes it through the workspace itself
    assert np.array_equal(out, oracle.sr_pipeline(img, arrays, 2, 2, modes="sdyct", modes2="sct"))


@pytest.mark.parametrize("pad", ["edge", "reflect", "symmetric", "wrap"])
@pytest.mark.parametrize("model,C,H,W,scale,S", [("lerf-g", 3, 70, 80, 2, 2), ("lerf-l", 3, 33, 150, (1.5, 2.0), 2),
                                                 ("lerf-g", 3, 130, 66, 3, 4), ("lerf-g", 1, 5, 6, 2, 2)])
def test_pad_modes_on_the_fused_uint8_path(torch, oracle, luts_g, luts_l, pad, model, C, H, W, scale, S):
    """pad_mode of the image operand (np.pad(input, pad_vec, mode=...), resize_right2d_numpy.py:143, 208) through the
    tile-fused kernel, the direct kernel and the oracle; wrap reads the far side of the frame from the stage-1 output"""
    import lerf_pytorch_amd as L
    from lerf_pytorch_amd import _lib, ops
    linear = model == "lerf-l"
    arrays = luts_l if linear else luts_g
    eng = L.LerfEngine(L.LutSet.from_arrays(arrays), support=S)
    sc = scale if isinstance(scale, tuple) else (scale, scale)
    img = np.random.default_rng(H + W).integers(0, 256, (H, W, C), dtype=np.uint8)
    x = torch.from_numpy(img).cuda()
    geo = ops.SrGeometry((H, W), list(sc), None, eng.support, "cuda", pad_mode=_lib.PAD_MODES[pad])
    assert ops.sr_fused_supported(C, eng.luts, geo, eng.kind)
    a = ops.sr_fused_u8(x, eng.luts, geo, eng.kind, eng.max_sigma).cpu().numpy()
    feat, hq = ops.lut_stages(x, eng.luts)
    b = ops.resize_hwc_u8(feat, hq, geo, eng.kind, eng.max_sigma, out="u8").cpu().numpy()
    assert np.array_equal(a, b), "fused != direct with pad_mode=%s" % pad
    fo, ho = oracle.lut_stages(img, arrays, 1 if linear else 3)
    p0, p1, p2 = oracle._split_hq(ho, eng.kind)
    ref = oracle.resize_params_f32(np.transpose(fo.astype(np.float32), (2, 0, 1)), p0, p1, p2, sc[0], sc[1], eng.support,
                                   eng.max_sigma, eng.kind, pad_mode=pad)
    ref = oracle.to_u8(np.transpose(ref, (1, 2, 0)))
    assert np.array_equal(a, ref), "fused != oracle with pad_mode=%s (%d bytes)" % (pad, int((a != ref).sum()))
    if pad != "wrap":                                         # single launch: every source pixel is inside the tile
        one = ops.sr_fused_u8(x, eng.luts, geo, eng.kind, eng.max_sigma, workspace=False).cpu().numpy()
        assert np.array_equal(one, a)
    else:
        # a workspace the caller tells the launch not to use (LERF_GEO_SINGLE_LAUNCH) must not put wrap padding on
        # the single-launch kernel (it cannot serve the far side of the frame): the call takes the direct kernels, same bytes
        geo.struct.flags |= _lib.GEO_SINGLE_LAUNCH
        forced = ops.sr_fused_u8(x, eng.luts, geo, eng.kind, eng.max_sigma).cpu().numpy()
        assert np.array_equal(forced, a)


@pytest.mark.parametrize("max_sigma", [13.0, 14.0, 20.0, 40.0])
def test_large_max_sigma_takes_the_shifted_sums(torch, oracle, luts_g, max_sigma):
    """max_sigma is a free constructor argument (resize_right2d_numpy.py:143).  Beyond 13.2 the unshifted exp2 sums of
    the fast stage 3 can underflow to 0/0 (every tap beyond 2^-126); the kernels then take the minimum-shifted sums,
    like the reference's float64 arithmetic never underflows.  Saturated hyper-parameters make the worst case."""
    import lerf_pytorch_amd as L
    from lerf_pytorch_amd import ops
    eng = L.LerfEngine(L.LutSet.from_arrays(luts_g), support=2, max_sigma=max_sigma)
    rng = np.random.default_rng(11)
    H, W = 48, 70
    feat = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    hq = rng.integers(0, 256, (H, W, 3, 3), dtype=np.uint8)
    hq[: H // 2, :, :, 1:] = 255                              # sigma_x = sigma_y = max_sigma: the largest exponents
    geo = eng.sr_geometry((H, W), 2)
    out = ops.resize_hwc_u8(torch.from_numpy(feat).cuda(), torch.from_numpy(hq).cuda(), geo, "gauss", max_sigma, out="u8").cpu().numpy()
    ref = oracle.to_u8(oracle.resize_u8(feat, hq, 2, 2, 2, max_sigma, "gauss"))
    assert np.array_equal(out, ref), "%d bytes differ at max_sigma=%g" % (int((out != ref).sum()), max_sigma)
    # whole path through the tile-fused kernel (its hyper-parameters come from the LUTs)
    img = rng.integers(0, 256, (70, 66, 3), dtype=np.uint8)
    a = eng.sr(img, 2)
    assert np.array_equal(a, eng.sr(img, 2, fused=False))
    assert np.array_equal(a, oracle.sr_pipeline(img, luts_g, 2, 2, S=2, max_sigma=max_sigma))


def test_ragged_launch_equals_frame_by_frame(torch, oracle, luts_g, luts_l):
    """frames of different sizes AND scale factors through one launch pair (the benchmark-folder loop of eltr.run,
    eval_lut_sr.py:489-512); 20 frames = two descriptor chunks"""
    import lerf_pytorch_amd as L
    from lerf_pytorch_amd import ops
    rng = np.random.default_rng(5)
    shapes = [(64, 64, 2), (100, 37, 3), (1, 1, 4), (129, 65, 2), (30, 200, (1.5, 2.0)), (70, 70, 4), (5, 300, 2), (66, 130, 3)]
    shapes = shapes + [(h + 3, w + 1, s) for h, w, s in shapes] + shapes[:4]
    for model, arrays in (("lerf-g", luts_g), ("lerf-l", luts_l)):
        eng = L.LerfEngine(L.LutSet.from_arrays(arrays))
        imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w, _ in shapes]
        xs = [torch.from_numpy(i).cuda() for i in imgs]
        geos = [eng.sr_geometry((h, w), s if isinstance(s, tuple) else (s, s)) for h, w, s in shapes]
        outs = ops.sr_fused_ragged_u8(xs, eng.luts, geos, eng.kind, eng.max_sigma)
        for img, (h, w, s), o in zip(imgs, shapes, outs):
            assert np.array_equal(o.cpu().numpy(), eng.sr(img, s)), "ragged != single launch for %dx%d x%s" % (h, w, s)
        sc = shapes[1][2]
        assert np.array_equal(outs[1].cpu().numpy(), oracle.sr_pipeline(imgs[1], arrays, sc, sc, linear=(model == "lerf-l")))
        packed = ops.stages_packed_ragged(xs, eng.luts)
        for x, pk in zip(xs[:6], packed[:6]):
            assert torch.equal(pk, ops.stages_packed(x, eng.luts))


@pytest.mark.parametrize("H,W,scale,grid,S", [(200, 260, 2, (2, 4), 2), (150, 131, 1.5, (2, 2), 2), (140, 150, 3, (2, 3), 4),
                                              (97, 128, 2.4, (1, 2), 2)])
def test_blocks_emulated_equal_full_frame(torch, H, W, scale, grid, S):
    """what every rank of a 2-D block partition runs (block + halo as the frame, tiles over the owned block, one launch),
    rank by rank on one GPU, stitched == the full frame bit for bit"""
    import lerf_pytorch_amd as L
    from lerf_pytorch_amd import dist as ldist
    eng = L.LerfEngine.shipped("lerf-g", support=S)
    img = np.random.default_rng(H + W).integers(0, 256, (H, W, 3), dtype=np.uint8)
    x = torch.from_numpy(img).cuda()
    full = eng.sr(x, scale)
    geo = eng.sr_geometry((H, W), scale)
    lr, lc = geo.host["left_r"], geo.host["left_c"]
    out = torch.zeros_like(full)
    for r in range(grid[0] * grid[1]):
        p = ldist.BlockPlan(H, W, grid, r, S, lr, lc)
        ext = x[p.ylo:p.yhi, p.xlo:p.xhi].contiguous().unsqueeze(0)
        i0, i1, j0, j1 = p.out_rect()
        out[i0:i1, j0:j1] = ldist.sr_block(eng, ext, p, geo)[0]
    assert torch.equal(out, full)


@pytest.mark.parametrize("H,W,scale,grid,S,model", [(200, 260, 2, (2, 4), 2, "lerf-g"), (150, 131, 1.5, (2, 2), 2, "lerf-g"),
                                                    (140, 150, 3, (2, 3), 4, "lerf-g"), (97, 128, 2.4, (1, 2), 2, "lerf-g"),
                                                    (300, 280, 2, (2, 2), 2, "lerf-l"), (280, 304, (1.5, 2.0), (2, 2), 2, "lerf-l")])
def test_blocks_batched_two_launch_roi_equal_full_frame(torch, H, W, scale, grid, S, model):
    """round 4: a BATCH of blocks takes the two-launch path over the region of interest -- stage 1 once per pixel over the
    owned block widened by 3 + S/2 pixels, stages 2+3 over the owned block -- rank by rank on one GPU, stitched == the full
    frames bit for bit; the workspace starts out poisoned (nothing outside the stage-1 region may be read for an owned output)"""
    import lerf_pytorch_amd as L
    from lerf_pytorch_amd import dist as ldist, ops
    eng = L.LerfEngine.shipped(model, support=S)
    N = 3
    x = torch.from_numpy(np.random.default_rng(H * W).integers(0, 256, (N, H, W, 3), dtype=np.uint8)).cuda()
    geo = eng.sr_geometry((H, W), scale)
    full = ops.sr_fused_u8(x, eng.luts, geo, eng.kind, eng.max_sigma)
    lr, lc = geo.host["left_r"], geo.host["left_c"]
    out = torch.zeros_like(full)
    for r in range(grid[0] * grid[1]):
        p = ldist.BlockPlan(H, W, grid, r, S, lr, lc)
        ext = x[:, p.ylo:p.yhi, p.xlo:p.xhi].contiguous()
        lh, lw = p.local_hw
        ws = torch.full((int(_ws_bytes(lh, lw, 3, N)),), 0xA5 if r % 2 else 0x00, dtype=torch.uint8, device="cuda")
        i0, i1, j0, j1 = p.out_rect()
        got = ldist.sr_block(eng, ext, p, geo, workspace=ws)
        assert torch.equal(got, ldist.sr_block(eng, ext, p, geo, workspace=False))     # two launches == one launch
        out[:, i0:i1, j0:j1] = got
    assert torch.equal(out, full)


def _ws_bytes(H, W, C, N):
    from lerf_pytorch_amd import _lib
    return _lib.lib().lerf_sr_fused_workspace_bytes(H, W, C, N)


def test_ragged_launch_of_more_than_64_frames_and_its_argument_checks(torch, oracle):
    """the outer loop of the ragged API (64 items per FusedArgs, 16 per launch pair); items that disagree on the per-call
    knobs (tie_queue_cap, flags) are refused instead of silently taking the first item's"""
    import lerf_pytorch_amd as L
    from lerf_pytorch_amd import ops
    eng = L.LerfEngine.shipped("lerf-g")
    rng = np.random.default_rng(70)
    shapes = [(8 + (3 * i) % 23, 10 + (5 * i) % 31) for i in range(70)]
    imgs = [torch.from_numpy(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).cuda() for h, w in shapes]
    geos = {}
    for h, w in shapes:
        geos.setdefault((h, w), eng.sr_geometry((h, w), 2))
    outs = ops.sr_fused_ragged_u8(imgs, eng.luts, [geos[s] for s in shapes], eng.kind, eng.max_sigma)
    for i in (0, 15, 16, 63, 64, 69):
        assert torch.equal(outs[i], eng.sr(imgs[i], 2))
    gl = [geos[s] for s in shapes[:3]]
    with pytest.raises(ValueError):
        ops.sr_fused_ragged_u8(imgs[:3], eng.luts, [gl[0], gl[1].with_tie_queue_cap(8), gl[2]], eng.kind, eng.max_sigma)
    with pytest.raises(ValueError):
        ops.sr_fused_ragged_u8(imgs[:3], eng.luts, [gl[0], gl[1], gl[2].with_flags(1)], eng.kind, eng.max_sigma)


def test_pitched_output_rows(torch):
    """lerf_sr_geo_t.out_row_pitch: the output as a [:, :, :w] view of wider rows -- same bytes, nothing written past the w-th pixel
    of a row; specialised, general and fallback kernels"""
    import lerf_pytorch_amd as L
    from lerf_pytorch_amd import ops
    # (70 x 65 and 66 x 129: a last tile column of ONE LR pixel -- a single column pair / a one-chunk row of the block tasks,
    #  whose index divisions by 1 overflowed their 32-bit magic before round 4; reachable only with pitched rows)
    for model, S, shape, scale in (("lerf-g", 2, (2, 70, 131, 3), 2), ("lerf-l", 2, (1, 66, 90, 3), (1.5, 2.0)), ("lerf-g", 4, (1, 40, 77, 3), 3),
                                   ("lerf-g", 2, (1, 50, 60, 1), 2), ("lerf-g", 2, (2, 70, 65, 3), 2), ("lerf-l", 2, (1, 66, 129, 3), 2),
                                   ("lerf-g", 2, (1, 65, 64, 3), 2)):
        eng = L.LerfEngine.shipped(model, support=S)
        x = torch.from_numpy(np.random.default_rng(shape[2]).integers(0, 256, shape, dtype=np.uint8)).cuda()
        geo = eng.sr_geometry(shape[1:3], scale)
        dense = ops.sr_fused_u8(x, eng.luts, geo, eng.kind, eng.max_sigma)
        oh, ow = geo.out_hw
        wide = torch.full((shape[0], oh, ow + 21, shape[3]), 0x5A, dtype=torch.uint8, device="cuda")
        got = ops.sr_fused_u8(x, eng.luts, geo, eng.kind, eng.max_sigma, out=wide[:, :, :ow])
        assert torch.equal(got, dense) and bool((wide[:, :, ow:] == 0x5A).all())
        one = ops.sr_fused_u8(x, eng.luts, geo, eng.kind, eng.max_sigma, out=wide[:, :, :ow], workspace=False)
        assert torch.equal(one, dense) and bool((wide[:, :, ow:] == 0x5A).all())


def test_force_general_flag_and_partition_pad_guard(torch):
    """lerf_sr_geo_t.flags replaces the LERF_FORCE_GENERAL environment variable of round 3 (the ABI reads no environment):
    the general kernels give the specialised kernels' bytes; strips / blocks refuse wrap padding"""
    import lerf_pytorch_amd as L
    from lerf_pytorch_amd import _lib, ops
    eng = L.LerfEngine.shipped("lerf-g")
    x = torch.from_numpy(np.random.default_rng(5).integers(0, 256, (2, 150, 210, 3), dtype=np.uint8)).cuda()
    geo = eng.sr_geometry((150, 210), 2)
    a = ops.sr_fused_u8(x, eng.luts, geo, eng.kind, eng.max_sigma)
    b = ops.sr_fused_u8(x, eng.luts, geo.with_flags(_lib.GEO_FORCE_GENERAL), eng.kind, eng.max_sigma)
    assert torch.equal(a, b)
    gw = ops.SrGeometry((150, 210), 2, support=2, pad_mode=_lib.PAD_MODES["wrap"])
    with pytest.raises(ValueError, match="wrap"):
        gw.row_slice(0, 80, 0, 150)
    with pytest.raises(ValueError, match="wrap"):
        gw.block_slice(0, 80, 0, 150, 0, 100, 0, 190)


def test_rect_copy_round_trip(torch):
    from lerf_pytorch_amd import ops
    g = torch.Generator(device="cpu").manual_seed(1)
    fr = torch.randint(0, 256, (3, 40, 50, 3), dtype=torch.uint8, generator=g).cuda()
    rects, off = [], 0
    for (y, x, h, w) in [(0, 0, 7, 50), (33, 0, 7, 50), (7, 0, 26, 8), (7, 42, 26, 8), (0, 0, 1, 1), (39, 49, 1, 1), (10, 10, 3, 5), (5, 6, 7, 8)]:
        rects.append((y, x, h, w, off))
        off += 3 * h * w * 3
    st = torch.zeros(off, dtype=torch.uint8, device="cuda")
    ops.rect_copy(fr, st, rects, True)
    for y, x, h, w, o in rects:
        assert torch.equal(st[o:o + 3 * h * w * 3].view(3, h, w, 3), fr[:, y:y + h, x:x + w])
    fr2 = torch.zeros_like(fr)
    ops.rect_copy(fr2, st, rects, False)
    for y, x, h, w, o in rects:
        assert torch.equal(fr2[:, y:y + h, x:x + w], fr[:, y:y + h, x:x + w])


def test_batched_packed_warp_equals_frame_by_frame(torch):
    import lerf_pytorch_amd as L
    from lerf_pytorch_amd import ops
    eng = L.LerfEngine.shipped("lerf-g")
    M = np.array([[2.05, 0.12, 15.0], [-0.08, 1.95, 40.0], [1.5e-5, -1.0e-5, 1.0]])
    x = torch.from_numpy(np.random.default_rng(2).integers(0, 256, (3, 90, 120, 3), dtype=np.uint8)).cuda()
    geo = ops.WarpGeometry((90, 120), M, (180, 240), 2)
    packed = ops.stages_packed(x, eng.luts)
    for out in ("u8", "f32"):
        a = ops.warp_packed(packed, geo, "gauss", 10.0, out=out)
        for b in range(3):
            one = ops.warp_packed(packed[b], geo, "gauss", 10.0, out=out)
            assert torch.equal(torch.nan_to_num(a[b].float()), torch.nan_to_num(one.float()))


def test_abi_argument_checks(torch):
    import ctypes as C
    import lerf_pytorch_amd as L
    from lerf_pytorch_amd import _lib, ops
    eng = L.LerfEngine.shipped("lerf-g")
    x = torch.zeros((2, 70, 70, 3), dtype=torch.uint8, device="cuda")
    geo = eng.sr_geometry((70, 70), 2)
    out = torch.empty((2, 140, 140, 3), dtype=torch.uint8, device="cuda")
    need = _lib.lib().lerf_sr_fused_workspace_bytes(70, 70, 3, 2)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    lib = _lib.lib()
    call = lambda nbytes: lib.lerf_sr_fused_u8(x.data_ptr(), x.stride(0), 2, 70, 70, 3, eng.luts.ref(), geo.ref(), 0, 10.0,
                                               out.data_ptr(), out.stride(0), ws.data_ptr(), nbytes, _lib.current_stream())
    assert call(need) == 0
    assert call(2 * 14704 - 1) == -1                          # shorter than the stage-1 output of the batch: LERF_EINVAL, no launch
    pk = torch.empty((2, 70, 70, 3), dtype=torch.int32, device="cuda")
    assert lib.lerf_stages_packed_u8(x.data_ptr(), x.stride(0), 2, 70, 70, 3, eng.luts.ref(), pk.data_ptr(), pk.stride(0),
                                     ws.data_ptr(), 100, _lib.current_stream()) == -1
    with pytest.raises(ValueError):
        ops.sr_fused_u8(x, eng.luts, geo, workspace=torch.empty(10, dtype=torch.uint8, device="cuda"))
    g2 = _lib.SrGeo()
    C.memmove(C.byref(g2), geo.ref(), C.sizeof(g2))
    g2.pad_mode = 9
    assert lib.lerf_sr_fused_u8(x.data_ptr(), x.stride(0), 2, 70, 70, 3, eng.luts.ref(), C.byref(g2), 0, 10.0, out.data_ptr(),
                                out.stride(0), ws.data_ptr(), need, _lib.current_stream()) == -1
    g2.pad_mode = 0
    g2.roi_y, g2.roi_x, g2.roi_h, g2.roi_w = 10, 10, 70, 70  # outside the frame
    assert lib.lerf_sr_fused_u8(x.data_ptr(), x.stride(0), 2, 70, 70, 3, eng.luts.ref(), C.byref(g2), 0, 10.0, out.data_ptr(),
                                out.stride(0), ws.data_ptr(), need, _lib.current_stream()) == -1


def test_halo_buffers_pack_and_unpack_on_the_device(torch, oracle):
    """the pack / unpack launches of StripBuffer and BlockBuffer (lerf_rect_copy_u8) without a process group: what the ranks
    send is the right rectangles of their own pixels, what they receive lands in the right halo rectangles"""
    from lerf_pytorch_amd import dist as ldist
    H, W, N, C = 200, 260, 2, 3
    lr, _, _, _ = oracle.sr_axis_tables(H, 2 * H, 2.0, 2)
    lc, _, _, _ = oracle.sr_axis_tables(W, 2 * W, 2.0, 2)
    bufs = [ldist.StripBuffer(ldist.StripPlan(H, 3, 1, 2, lr), N, W, C, torch.uint8, torch.device("cuda")),
            ldist.BlockBuffer(ldist.BlockPlan(H, W, (3, 3), 4, 2, lr, lc), N, C, torch.uint8, torch.device("cuda"), lr, lc)]   # the centre block: 8 neighbours
    g = torch.Generator(device="cpu").manual_seed(3)
    for buf in bufs:
        assert len(buf.sends) == len(buf.recvs) and len(buf.sends) in (2, 8)
        buf.ext.copy_(torch.randint(0, 256, tuple(buf.ext.shape), dtype=torch.uint8, generator=g))
        buf._copy(buf.sends, buf.send_buf, True)
        for _, (y, x, h, w), off, nb in buf.sends:
            assert torch.equal(buf.send_buf[off:off + nb].view(N, h, w, C), buf.ext[:, y:y + h, x:x + w])
        buf.recv_buf.copy_(torch.randint(0, 256, tuple(buf.recv_buf.shape), dtype=torch.uint8, generator=g))
        before = buf.own.clone()
        buf._copy(buf.recvs, buf.recv_buf, False)
        for _, (y, x, h, w), off, nb in buf.recvs:
            assert torch.equal(buf.ext[:, y:y + h, x:x + w], buf.recv_buf[off:off + nb].view(N, h, w, C))
        assert torch.equal(buf.own, before)                    # the halo writes never touch the owned pixels


@pytest.mark.parametrize("model,S,scale,shapes", [("lerf-g", 2, 2, [(256, 256), (40, 56), (97, 131)]), ("lerf-g", 4, 2, [(70, 90)]),
                                                  ("lerf-l", 2, (1.5, 2.0), [(130, 67)]), ("lerf-g", 2, 3, [(33, 200)])])
def test_tile_rows_give_the_same_bytes(torch, model, S, scale, shapes):
    """round 4: the 32- and 16-row tile instances (small launches) against the 64-row tiles, fused SR, packed stages and a
    ragged launch; the default rule picks small tiles for a launch that cannot fill the chip"""
    import lerf_pytorch_amd as L
    from lerf_pytorch_amd import _lib, ops
    eng = L.LerfEngine.shipped(model, support=S)
    rng = np.random.default_rng(len(shapes) + S)
    xs = [torch.from_numpy(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).cuda() for h, w in shapes]
    geos = [eng.sr_geometry(x.shape[:2], scale) for x in xs]
    for x, g in zip(xs, geos):
        want = ops.sr_fused_u8(x, eng.luts, g.with_flags(_lib.GEO_TILE_ROWS_64), eng.kind, eng.max_sigma)
        assert torch.equal(want, eng.sr(x, scale, fused=False))
        for fl in (_lib.GEO_TILE_ROWS_32, _lib.GEO_TILE_ROWS_16, 0):
            assert torch.equal(ops.sr_fused_u8(x, eng.luts, g.with_flags(fl), eng.kind, eng.max_sigma), want)
            assert torch.equal(ops.sr_fused_u8(x, eng.luts, g.with_flags(fl), eng.kind, eng.max_sigma, workspace=False), want)
    want = [ops.sr_fused_u8(x, eng.luts, g.with_flags(_lib.GEO_TILE_ROWS_64), eng.kind, eng.max_sigma) for x, g in zip(xs, geos)]
    for fl in (_lib.GEO_TILE_ROWS_32, _lib.GEO_TILE_ROWS_16, 0):
        got = ops.sr_fused_ragged_u8(xs, eng.luts, [g.with_flags(fl) for g in geos], eng.kind, eng.max_sigma)
        assert all(torch.equal(a, b) for a, b in zip(got, want))
    feat, hq = eng.stages(xs[0])
    f2, h2 = ops.lut_stages(xs[0], eng.luts)
    assert torch.equal(torch.as_tensor(feat).cuda(), f2) and torch.equal(torch.as_tensor(hq).cuda(), h2)     # small launch: small tiles by default


@pytest.mark.parametrize("model,S,N,H,W,scale", [("lerf-g", 2, 1, 256, 256, 2), ("lerf-g", 4, 2, 70, 90, 2), ("lerf-l", 2, 1, 130, 67, (1.5, 2.0)),
                                                 ("lerf-g", 2, 2, 300, 520, 2)])
def test_fused_path_is_capturable_in_a_hip_graph(torch, model, S, N, H, W, scale):
    """The C ABI issues only stream-ordered work (two kernel launches + the tie-queue reset; no allocation, no synchronisation, no
    pointer query once the workspace exists): a caller may capture it (torch.cuda.CUDAGraph = hipGraph) and replay it on new
    input bytes.  Replay must give the bytes of an eager call."""
    import lerf_pytorch_amd as L
    eng = L.LerfEngine.shipped(model, support=S)
    rng = np.random.default_rng(N * H + W)
    x = torch.from_numpy(rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)).cuda()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):                              # warm-up off the default stream: workspace and tables exist before the capture
        eng.sr(x, scale)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = eng.sr(x, scale)
    for k in range(3):
        x2 = torch.from_numpy(rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)).cuda()
        x.copy_(x2)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, eng.sr(x2, scale))
