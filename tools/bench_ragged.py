#!/usr/bin/env python3
"""Round 3: small frames.  (a) Set5 LR images at x2 / x3 / x4 (15 frames of 5 sizes x 3 scales, LeRF-G and LeRF-L): image by
image (one launch pair each, what eltr.run does: eval_lut_sr.py:489-512) against ONE ragged launch pair; md5 of every output
against tests/golden/g5_set5.json.  (b) the 256 x 256 tile of BASELINE config 1: latency of one call, two launches vs one."""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from PIL import Image
import lerf_pytorch_amd as L
from lerf_pytorch_amd import ops

DATA = os.path.join(ROOT, "tests", "data", "Set5")
NAMES = ["baby", "bird", "butterfly", "head", "woman"]
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "g5_set5.json")))


def timeit(fn, n=30):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


for model in ("lerf-g", "lerf-l"):
    eng = L.LerfEngine.shipped(model)
    jobs = [(s, n) for s in (2, 3, 4) for n in NAMES]
    xs = [torch.from_numpy(np.array(Image.open(os.path.join(DATA, "LR_bicubic/rrLR_X%.2f_%.2f" % (s, s), n + ".png")))).cuda() for s, n in jobs]
    scs = [(float(s), float(s)) for s, _ in jobs]
    geos = [eng.sr_geometry(x.shape[:2], sc) for x, sc in zip(xs, scs)]
    one = lambda: [ops.sr_fused_u8(x, eng.luts, g, eng.kind, eng.max_sigma) for x, g in zip(xs, geos)]
    rag = lambda: ops.sr_fused_ragged_u8(xs, eng.luts, geos, eng.kind, eng.max_sigma)
    a, b = one(), rag()
    same = all(torch.equal(p, q) for p, q in zip(a, b))
    md5ok = True
    sr = gold.get("sr", {})
    for (s, n), o in zip(jobs, b):
        ent = sr["%s/x%d/%s" % (model, s, n)]
        md5ok = md5ok and hashlib.md5(np.ascontiguousarray(o.cpu().numpy()).tobytes()).hexdigest() == ent["md5_out"]
    t1, t2 = timeit(one), timeit(rag)
    lrpx = sum(x.shape[0] * x.shape[1] for x in xs)
    print("%s Set5 x2/x3/x4, 15 frames (%.2f M LR px, %.2f M output px): image by image %.3f ms (15 launch pairs), ragged %.3f ms (1 launch pair): x%.2f; "
          "ragged == image by image: %s; md5 == reference: %s"
          % (model, lrpx / 1e6, sum(g.out_hw[0] * g.out_hw[1] for g in geos) / 1e6, t1 * 1e3, t2 * 1e3, t1 / t2, same, md5ok))

eng = L.LerfEngine.shipped("lerf-g")
tile = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (256, 256, 3), dtype=np.uint8)).cuda()
geo = eng.sr_geometry((256, 256), 2)
o = torch.empty((512, 512, 3), dtype=torch.uint8, device="cuda")
ta = timeit(lambda: ops.sr_fused_u8(tile, eng.luts, geo, "gauss", 10.0, out=o), 200)
tb = timeit(lambda: ops.sr_fused_u8(tile, eng.luts, geo, "gauss", 10.0, out=o, workspace=False), 200)
print("config-1 tile 256x256 -> 512x512 (16 tiles on 256 CUs): two launches %.4f ms, one launch %.4f ms per call (host-timed, back to back)" % (ta * 1e3, tb * 1e3))

# ---- round 4: tile rows.  64-row tiles are the throughput tile; launches too small to fill the chip take 32- or 16-row tiles
#      (lerf_fused.hip tile_rows_for); LERF_GEO_TILE_ROWS_* force a height
from lerf_pytorch_amd import _lib
ref = ops.sr_fused_u8(tile, eng.luts, geo.with_flags(_lib.GEO_TILE_ROWS_64), "gauss", 10.0)
for nm, fl in (("64 rows (16 tiles)", _lib.GEO_TILE_ROWS_64), ("32 rows (32 tiles)", _lib.GEO_TILE_ROWS_32), ("16 rows (64 tiles)", _lib.GEO_TILE_ROWS_16), ("default", 0)):
    g = geo.with_flags(fl)
    same = torch.equal(ops.sr_fused_u8(tile, eng.luts, g, "gauss", 10.0), ref)
    t2 = timeit(lambda: ops.sr_fused_u8(tile, eng.luts, g, "gauss", 10.0, out=o), 200)
    t1 = timeit(lambda: ops.sr_fused_u8(tile, eng.luts, g, "gauss", 10.0, out=o, workspace=False), 200)
    print("config-1 tile 256x256 -> 512x512, tiles of %-18s: two launches %.4f ms, one launch %.4f ms; bytes == 64-row tiles: %s" % (nm, t2 * 1e3, t1 * 1e3, same))
for model in ("lerf-g",):
    eng = L.LerfEngine.shipped(model)
    jobs = [(s, n) for s in (2, 3, 4) for n in NAMES]
    xs = [torch.from_numpy(np.array(Image.open(os.path.join(DATA, "LR_bicubic/rrLR_X%.2f_%.2f" % (s, s), n + ".png")))).cuda() for s, n in jobs]
    geos0 = [eng.sr_geometry(x.shape[:2], (float(s), float(s))) for x, (s, _) in zip(xs, jobs)]
    base = None
    for nm, fl in (("64", _lib.GEO_TILE_ROWS_64), ("32", _lib.GEO_TILE_ROWS_32), ("16", _lib.GEO_TILE_ROWS_16), ("default", 0)):
        geos = [g.with_flags(fl) for g in geos0]
        outs = ops.sr_fused_ragged_u8(xs, eng.luts, geos, eng.kind, eng.max_sigma)
        base = base or outs
        same = all(torch.equal(p, q) for p, q in zip(outs, base))
        t = timeit(lambda: ops.sr_fused_ragged_u8(xs, eng.luts, geos, eng.kind, eng.max_sigma))
        print("%s Set5 ragged launch, tile rows %-7s: %.3f ms; bytes == 64-row tiles: %s" % (model, nm, t * 1e3, same))
    for k, (x, g0) in enumerate(list(zip(xs, geos0))[:5]):
        ts = []
        for fl in (_lib.GEO_TILE_ROWS_64, 0):
            g = g0.with_flags(fl)
            ts.append(timeit(lambda: ops.sr_fused_u8(x, eng.luts, g, eng.kind, eng.max_sigma), 100))
        print("   %s x2 LR %dx%d alone: 64-row tiles %.4f ms, default rule %.4f ms" % (NAMES[k], x.shape[1], x.shape[0], ts[0] * 1e3, ts[1] * 1e3))
