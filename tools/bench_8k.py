#!/usr/bin/env python3
"""BASELINE config 5 shape on one GPU: 2160x3840 -> 4320x7680 frames (fused == unfused check + throughput)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import lerf_pytorch_amd as L
from lerf_pytorch_amd import ops, dist as ldist
eng = L.LerfEngine.shipped("lerf-g")
rng = np.random.default_rng(9)
B = 4
x = torch.from_numpy(rng.integers(0, 256, (B, 2160, 3840, 3), dtype=np.uint8)).cuda()
geo = eng.sr_geometry((2160, 3840), 2)
out = torch.empty((B, 4320, 7680, 3), dtype=torch.uint8, device="cuda")
ops.sr_fused_u8(x, eng.luts, geo, "gauss", 10.0, out=out)
ref = eng.sr(x[1], 2, fused=False)
print("fused == unfused on a 4K->8K frame:", bool(torch.equal(out[1], ref)))
torch.cuda.synchronize(); t = time.perf_counter(); n = 5
for _ in range(n): ops.sr_fused_u8(x, eng.luts, geo, "gauss", 10.0, out=out)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
print("4K->8K, %d frames/launch: %.3f ms/frame, %.1f Mpix/s" % (B, dt * 1e3 / B, B * 4320 * 7680 / dt / 1e6))
# 8 strips of one frame, emulated on one GPU (per-strip kernel time = what each of 8 GPUs would run)
parts = []; ts = []
for r in range(8):
    plan = ldist.StripPlan(2160, 8, r, 2, geo.host["left_r"])
    ext = x[0, plan.ylo:plan.yhi].contiguous()
    lg = geo.row_slice(plan.ylo, plan.yhi - plan.ylo, plan.i0, plan.i1)
    o = ops.sr_fused_u8(ext, eng.luts, lg, "gauss", 10.0)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): ops.sr_fused_u8(ext, eng.luts, lg, "gauss", 10.0, out=o)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t) / 5)
    parts.append(o)
print("8 strips stitched == full frame:", bool(torch.equal(torch.cat(parts, 0), out[0])))
print("per-strip kernel time (270 LR rows + halo): %.3f ms max, %.3f ms mean -> one frame over 8 GPUs ~ %.1f Mpix/s + halo exchange"
      % (max(ts) * 1e3, np.mean(ts) * 1e3, 4320 * 7680 / max(ts) / 1e6))

# the same with the 8 frames of a bench step in ONE launch per rank (bench.py --config 5 --mode strips): rank 3's strips
x8 = torch.from_numpy(rng.integers(0, 256, (8, 2160, 3840, 3), dtype=np.uint8)).cuda()
plan = ldist.StripPlan(2160, 8, 3, 2, geo.host["left_r"])
ext8 = x8[:, plan.ylo:plan.yhi].contiguous()
lg = geo.row_slice(plan.ylo, plan.yhi - plan.ylo, plan.i0, plan.i1)
o8 = ops.sr_fused_u8(ext8, eng.luts, lg, "gauss", 10.0)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): ops.sr_fused_u8(ext8, eng.luts, lg, "gauss", 10.0, out=o8)
torch.cuda.synchronize(); d8 = (time.perf_counter() - t) / 5
full8 = 8 * dt / B
print("8 frames, one rank's strips in one launch: %.3f ms vs 1/8 of the whole-frame batch %.3f ms -> strong-scaling efficiency of the "
      "compute part %.0f %% (%.1f Gpix/s over 8 GPUs before the halo exchange)" % (d8 * 1e3, full8 / 8 * 1e3, 100 * full8 / 8 / d8, 8 * 4320 * 7680 / d8 / 1e9))

# ---- 2 x 4 blocks of one frame (round 3), emulated rank by rank: block + halo as the frame, tiles over the owned block
#      (255 per rank = one round of workgroups), ONE launch per rank (stage 1 recomputed on the tile halos)
lr_, lc_ = geo.host["left_r"], geo.host["left_c"]
outb = torch.zeros_like(out[0]); tb = []
for r in range(8):
    plan = ldist.BlockPlan(2160, 3840, (2, 4), r, 2, lr_, lc_)
    ext = x[0, plan.ylo:plan.yhi, plan.xlo:plan.xhi].contiguous().unsqueeze(0)
    lg = ldist.block_geometry(geo, plan)
    i0, i1, j0, j1 = plan.out_rect()
    o = ops.sr_fused_u8(ext, eng.luts, lg, "gauss", 10.0, workspace=False, out=ldist.block_output(plan, 1, 3, ext.device))
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): ops.sr_fused_u8(ext, eng.luts, lg, "gauss", 10.0, out=o, workspace=False)
    torch.cuda.synchronize(); tb.append((time.perf_counter() - t) / 10)
    outb[i0:i1, j0:j1] = o[0]
print("8 blocks (2 x 4) stitched == full frame:", bool(torch.equal(outb, out[0])))
print("per-block kernel time (1080 x 960 LR pixels + halo, 255 tiles, one launch): %.3f ms max, %.3f ms mean -> one frame over 8 GPUs ~ %.1f Mpix/s + halo "
      "exchange; strips: %.3f ms max (300 tiles, two launches)" % (max(tb) * 1e3, np.mean(tb) * 1e3, 4320 * 7680 / max(tb) / 1e6, max(ts) * 1e3))
print("single-frame efficiency against 1/8 of a whole frame in a batch (%.3f ms): blocks %.0f %%, strips %.0f %%"
      % (dt / B / 8 * 1e3, 100 * dt / B / 8 / max(tb), 100 * dt / B / 8 / max(ts)))
# the two-launch variant on a block (288 + 272 workgroups) for comparison
plan = ldist.BlockPlan(2160, 3840, (2, 4), 5, 2, lr_, lc_)
ext = x[0, plan.ylo:plan.yhi, plan.xlo:plan.xhi].contiguous().unsqueeze(0)
lgw = geo.block_slice(plan.ylo, plan.local_hw[0], plan.i0, plan.i1, plan.xlo, plan.local_hw[1], plan.j0, plan.j1)    # no roi: tiles over block + halo
o = ops.sr_fused_u8(ext, eng.luts, lgw, "gauss", 10.0)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): ops.sr_fused_u8(ext, eng.luts, lgw, "gauss", 10.0, out=o)
torch.cuda.synchronize()
print("block 5 without the region of interest (tiles over block + halo: 18 x 16 = 288, two launches): %.3f ms" % ((time.perf_counter() - t) / 10 * 1e3))

# ---- round 4: the 8 frames of a bench step as 2 x 4 blocks, one rank's blocks in ONE launch pair over the region of interest
#      (stage 1 once per pixel over block + 4 px: 8 x 272 workgroups, stages 2+3 over the owned block: 8 x 255) against the
#      single-launch form (stage 1 recomputed on every tile's halo) and against 1/8 of the whole-frame batch
full8 = 8 * dt / B
tb2, tb1 = [], []
stitched = torch.zeros((8, 4320, 7680, 3), dtype=torch.uint8, device="cuda")
for r in range(8):
    plan = ldist.BlockPlan(2160, 3840, (2, 4), r, 2, lr_, lc_)
    ext8 = x8[:, plan.ylo:plan.yhi, plan.xlo:plan.xhi].contiguous()
    lg = ldist.block_geometry(geo, plan)
    o2 = ops.sr_fused_u8(ext8, eng.luts, lg, "gauss", 10.0, out=ldist.block_output(plan, 8, 3, ext8.device))   # workspace from the cache: two launches; rows padded to 16 B
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): ops.sr_fused_u8(ext8, eng.luts, lg, "gauss", 10.0, out=o2)
    torch.cuda.synchronize(); tb2.append((time.perf_counter() - t) / 5)
    if r in (0, 5):
        o1 = ops.sr_fused_u8(ext8, eng.luts, lg, "gauss", 10.0, workspace=False, out=ldist.block_output(plan, 8, 3, ext8.device))
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): ops.sr_fused_u8(ext8, eng.luts, lg, "gauss", 10.0, out=o1, workspace=False)
        torch.cuda.synchronize(); tb1.append((time.perf_counter() - t) / 5)
        assert torch.equal(o1, o2)
    i0, i1, j0, j1 = plan.out_rect()
    stitched[:, i0:i1, j0:j1] = o2
whole8 = torch.empty((8, 4320, 7680, 3), dtype=torch.uint8, device="cuda")
for h in range(2): ops.sr_fused_u8(x8[4 * h:4 * h + 4], eng.luts, geo, "gauss", 10.0, out=whole8[4 * h:4 * h + 4])
print("per-rank ms (two-launch ROI, 8 frames):", " ".join("%.3f" % (t * 1e3) for t in tb2))
print("8 frames x 8 blocks (two-launch ROI) stitched == whole frames:", bool(torch.equal(stitched, whole8)))
print("8 frames, one rank's BLOCKS, two launches over the region of interest: %.3f ms max, %.3f ms mean (single launch, stage 1 per tile halo: %.3f ms) "
      "vs 1/8 of the whole-frame batch %.3f ms -> strong-scaling efficiency of the compute part %.0f %% (single launch %.0f %%; strips %.0f %%)"
      % (max(tb2) * 1e3, np.mean(tb2) * 1e3, max(tb1) * 1e3, full8 / 8 * 1e3, 100 * full8 / 8 / max(tb2), 100 * full8 / 8 / max(tb1), 100 * full8 / 8 / d8))

# ---- round 6: the same 8-frame blocks as interior + border parts (dist.OverlappedBlock): the interior pair of launches does not
#      need the halo and runs UNDER the exchange, the (up to four) border parts after it -- emulated per-rank compute time of all
#      parts against the single region-of-interest pair above (what the overlap costs in launches), and the interior's share
#      (the time available to hide the exchange in); one frame per step likewise
tov, tin = [], []
tov1, tin1 = [], []
st2 = torch.zeros((8, 4320, 7680, 3), dtype=torch.uint8, device="cuda")
for r in range(8):
    plan = ldist.BlockPlan(2160, 3840, (2, 4), r, 2, lr_, lc_)
    ext8 = x8[:, plan.ylo:plan.yhi, plan.xlo:plan.xhi].contiguous()
    ovl = ldist.OverlappedBlock(eng, plan, geo)
    o = ldist.block_output(plan, 8, 3, ext8.device)
    ws = torch.empty(max(1, L._lib.lib().lerf_sr_fused_workspace_bytes(ext8.shape[1], ext8.shape[2], 3, 8)), dtype=torch.uint8, device="cuda")

    def run(e, oo, w):
        for k in range(len(ovl.parts)): ovl._launch(k, e, oo, w)
    run(ext8, o, ws)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): run(ext8, o, ws)
    torch.cuda.synchronize(); tov.append((time.perf_counter() - t) / 5)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): ovl._launch(0, ext8, o, ws)
    torch.cuda.synchronize(); tin.append((time.perf_counter() - t) / 5)
    i0, i1, j0, j1 = plan.out_rect()
    st2[:, i0:i1, j0:j1] = o
    o1 = ldist.block_output(plan, 1, 3, ext8.device)
    run(ext8[:1], o1, False)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): run(ext8[:1], o1, False)
    torch.cuda.synchronize(); tov1.append((time.perf_counter() - t) / 10)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): ovl._launch(0, ext8[:1], o1, False)
    torch.cuda.synchronize(); tin1.append((time.perf_counter() - t) / 10)
print("8 frames x 8 blocks as interior + border parts stitched == whole frames:", bool(torch.equal(st2, whole8)))
print("8 frames per rank, interior + border parts (%d launches pairs): %.3f ms max (region-of-interest pair: %.3f ms); the interior alone %.3f ms = the window the "
      "halo exchange hides in" % (len(ovl.parts), max(tov) * 1e3, max(tb2) * 1e3, max(tin) * 1e3))
print("1 frame per rank, interior + border parts: %.3f ms max (one region-of-interest launch: %.3f ms); the interior alone %.3f ms"
      % (max(tov1) * 1e3, max(tb) * 1e3, max(tin1) * 1e3))

# ---- round 6: the warp over 8 ranks (dist.WarpRowPlan): config 4 (1080p -> 4K, isc matrix, 8 frames sharing the homography), every
#      rank computes an eighth of the OUTPUT rows from its band of the source; nothing is exchanged.  Emulated per rank on one GPU:
#      band sizes, per-rank time (LUT stages on the band + the warp of its rows), stitched == the whole-frame warp
del st2, x8, whole8
torch.cuda.empty_cache()
M = np.array([[2.05, 0.12, 15.0], [-0.08, 1.95, 40.0], [1.5e-5, -1.0e-5, 1.0]])      # bench.py's M_ISC
Hs, Ws, ohw = 1080, 1920, (2160, 3840)
fr = torch.from_numpy(rng.integers(0, 256, (8, Hs, Ws, 3), dtype=np.uint8)).cuda()
geo_w = ops.WarpGeometry((Hs, Ws), M, ohw, eng.support)


def whole_warp():
    return ops.warp_packed(ops.stages_packed(fr, eng.luts), geo_w, eng.kind, eng.max_sigma, out="u8")


ref_w = whole_warp()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): whole_warp()
torch.cuda.synchronize(); t_whole = (time.perf_counter() - t) / 5
tw, bands = [], []
st_w = torch.zeros_like(ref_w)
for r in range(8):
    wp = ldist.WarpRowPlan(Hs, Ws, M, ohw, 8, r, eng.support)
    band = fr[:, wp.b0:wp.b1].contiguous()
    gw = wp.geometry()

    def run_w():
        return ops.warp_packed(ops.stages_packed(band, eng.luts), gw, eng.kind, eng.max_sigma, out="u8")
    o = run_w()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): run_w()
    torch.cuda.synchronize(); tw.append((time.perf_counter() - t) / 5)
    st_w[:, wp.i0:wp.i1] = o
    bands.append(wp.b1 - wp.b0)
print("warp 1080p -> 4K over 8 ranks by output rows: stitched == whole-frame warp:", bool(torch.equal(st_w, ref_w)))
print("source bands (rows of 1080): %s; per-rank ms (8 frames): %s" % (" ".join(str(b) for b in bands), " ".join("%.3f" % (t * 1e3) for t in tw)))
print("8 frames, one rank's output rows: %.3f ms max, %.3f ms mean vs 1/8 of the whole-frame batch %.3f ms -> strong-scaling efficiency of the "
      "compute part %.0f %% (the bands overlap by the taps' and the LUT stages' reach: %.0f source rows are computed in all, %.2fx the frame)"
      % (max(tw) * 1e3, np.mean(tw) * 1e3, t_whole / 8 * 1e3, 100 * t_whole / 8 / max(tw), sum(bands), sum(bands) / Hs))
