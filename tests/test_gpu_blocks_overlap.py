"""GPU (one device runs every rank's launches in turn): dist.OverlappedBlock -- a block's interior launched before the halo is
in, its border rectangles after -- stitches to the bytes of the whole-frame launch, single frames and batches."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("H,W,grid,scale,N", [(1080, 1920, (2, 4), 2.0, 1), (1080, 1920, (2, 2), 2.0, 2), (700, 900, (2, 2), 1.5, 1)])
def test_overlapped_block_launches_equal_the_whole_frame(H, W, grid, scale, N):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import lerf_pytorch_amd as L
    from lerf_pytorch_amd import dist as ldist, ops
    eng = L.LerfEngine.shipped("lerf-g")
    rng = np.random.default_rng(H + N)
    frames = torch.from_numpy(rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)).cuda()
    geo = eng.sr_geometry((H, W), [scale, scale])
    whole = ops.sr_fused_u8(frames, eng.luts, geo, eng.kind, eng.max_sigma)
    lr, lc = geo.host["left_r"], geo.host["left_c"]
    stitched = torch.zeros_like(whole)
    for rank in range(grid[0] * grid[1]):
        plan = ldist.BlockPlan(H, W, grid, rank, eng.support, lr, lc)
        buf = ldist.BlockBuffer(plan, N, 3, torch.uint8, torch.device("cuda"), lr, lc)
        ovl = ldist.OverlappedBlock(eng, plan, geo)
        assert len(ovl.parts) > 1 and ovl.parts[0]["interior"]
        out = ldist.block_output(plan, N, 3, torch.device("cuda"))
        out.fill_(7)
        # the interior first, with the halo still EMPTY (poisoned): it must not read it
        buf.ext.fill_(255)
        buf.own.copy_(frames[:, plan.y0:plan.y1, plan.x0:plan.x1])
        ws = None if N > 1 else False
        ovl._launch(0, buf.ext, out, ws)
        torch.cuda.synchronize()
        buf.ext.copy_(frames[:, plan.ylo:plan.yhi, plan.xlo:plan.xhi])          # "the halo arrives"
        for k in range(1, len(ovl.parts)):
            ovl._launch(k, buf.ext, out, ws)
        stitched[:, plan.i0:plan.i1, plan.j0:plan.j1] = out
    assert torch.equal(stitched, whole)
