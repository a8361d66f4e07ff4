"""The reference's evaluation harness on the MI355X path: the counterpart of the `eltr` classes and `__main__`
blocks of resample/eval_lut_sr.py:472-812 (arbitrary-scale SR, Y-PSNR / SSIM) and resample/eval_lut_warp.py:27-330
(homographic warp, masked mPSNR).  Same directory layout, option names, result files and printed tables:

    python -m lerf_pytorch_amd.resample.eval_harness sr   --testDir data/rrBenchmark  --resultRoot results/sr  -e models/lerf-g
    python -m lerf_pytorch_amd.resample.eval_harness warp --testDir data/WarpBenchmark --resultRoot results/warp -e models/lerf-l --linear

<testDir>/<dataset>/HR/*.png, <testDir>/<dataset>/LR_bicubic/rrLR_X{sh:.2f}_{sw:.2f}/*.png (SR) or
<testDir>/<dataset>/{isc,osc}/*.png + *.pth (warp; the 3x3 matrices may also be given as *.npy or *.json).
LUTs, resampling and metrics run on the GPU; PNG decoding/encoding stays on the host.
"""
from __future__ import annotations

import argparse
import json
import os

import numpy as np

from .. import metrics
from ..luts import LutSet
from ..pipeline import LerfEngine


def _load_rgb(path):
    from PIL import Image
    a = np.array(Image.open(path))
    if a.ndim == 2:                                   # eval_lut_sr.py:527-529: grey -> 3 equal channels
        a = np.stack([a, a, a], axis=2)
    return np.ascontiguousarray(a[:, :, :3])


def _load_matrix(stem):
    """3x3 homography next to the image: .npy / .json, or the reference's torch-saved .pth."""
    if os.path.exists(stem + ".npy"):
        return np.load(stem + ".npy").astype(np.float64)
    if os.path.exists(stem + ".json"):
        return np.array(json.load(open(stem + ".json")), dtype=np.float64)
    import torch
    return torch.load(stem + ".pth").numpy().astype(np.float64)


class Eltr:
    """eltr of eval_lut_sr.py:472 / eval_lut_warp.py:27 with the per-image work on the GPU."""

    def __init__(self, opt, luts: LutSet):
        self.opt = opt
        self.engine = LerfEngine(luts, support=opt.suppSize, max_sigma=opt.maxSigma)

    def _result_dir(self, *parts):
        if not self.opt.resultRoot:
            return None
        p = os.path.join(self.opt.resultRoot, os.path.basename(os.path.normpath(self.opt.expDir)), *parts)
        os.makedirs(p, exist_ok=True)
        return p

    def _files(self, dataset):
        folder = os.path.join(self.opt.testDir, dataset, "HR")
        return sorted(f for f in os.listdir(folder) if "png" in f)

    def run_sr_many(self, dataset, scales):
        """{(sh, sw): [[psnr, ssim], ...]}: every image of the dataset at every scale through ONE ragged launch pair
        (the reference walks them one by one: eval_lut_sr.py:487-512, 514-744)."""
        from PIL import Image
        files = self._files(dataset)
        jobs = [(sh, sw, f) for sh, sw in scales for f in files]
        lrs = [_load_rgb(os.path.join(self.opt.testDir, dataset, "LR_bicubic/rrLR_X{:.2f}_{:.2f}".format(sh, sw), f)) for sh, sw, f in jobs]
        outs = self.engine.sr_many([self.engine._dev(a)[0] for a in lrs], [(float(sh), float(sw)) for sh, sw, _ in jobs])
        res = {tuple(sc): [] for sc in scales}
        gts = {f: _load_rgb(os.path.join(self.opt.testDir, dataset, "HR", f)) for f in files}
        for (sh, sw, f), out in zip(jobs, outs):
            gt = gts[f]
            rdir = self._result_dir("X{:.2f}_{:.2f}".format(sh, sw), dataset)
            if rdir:
                Image.fromarray(out.cpu().numpy()).save(os.path.join(rdir, "{}_{}.png".format(f[:-4], self.opt.lutName)))
                Image.fromarray(gt).save(os.path.join(rdir, "{}_gt.png".format(f[:-4])))
            shave = max(int(sh), int(sw))
            res[(sh, sw)].append([metrics.psnr_y(gt, out, shave), metrics.ssim_y(gt, out)])
        return res

    def run_sr(self, dataset, scale_h, scale_w):
        """[[psnr, ssim], ...] per image (eval_lut_sr.py:487-512, 514-744)."""
        return self.run_sr_many(dataset, [(scale_h, scale_w)])[(scale_h, scale_w)]

    def run_warp(self, dataset, mode):
        """[[mpsnr], ...] per image (eval_lut_warp.py:42-68, 70-302); the LUT stages of all images run in one ragged launch"""
        from PIL import Image
        rdir = self._result_dir(dataset, mode)          # eval_lut_warp.py:51-56
        files = self._files(dataset)
        lrs = [_load_rgb(os.path.join(self.opt.testDir, dataset, mode, f)) for f in files]
        gts = [_load_rgb(os.path.join(self.opt.testDir, dataset, "HR", f)) for f in files]
        Ms = [_load_matrix(os.path.join(self.opt.testDir, dataset, mode, f[:-4])) for f in files]
        outs = self.engine.warp_many([self.engine._dev(a)[0] for a in lrs], Ms, [g.shape[:2] for g in gts])
        res = []
        for f, gt, (out, mask) in zip(files, gts, outs):
            res.append([metrics.mpsnr(out, gt, mask)])
            if rdir:
                m = mask.cpu().numpy()
                o = out.cpu().numpy()
                white = np.full_like(gt, 255)
                Image.fromarray(np.where(m, o, white)).save(os.path.join(rdir, "{}_{}.png".format(f[:-4], self.opt.lutName)))
                Image.fromarray(np.where(m, gt, white)).save(os.path.join(rdir, "{}_gt.png".format(f[:-4])))
                Image.fromarray((m * 255).astype(np.uint8)).save(os.path.join(rdir, "{}_mask.png".format(f[:-4])))
        return res


def sr_table(etr, datasets=("Set5",), scales=((2, 2), (3, 3), (4, 4))):
    """The printed table of eval_lut_sr.py:789-812 as a list of lines."""
    lines = ["\t".join(["Scale".ljust(15, " ")] + ["{:.1f}x{:.1f}\t".format(float(a), float(b)) for a, b in scales])]
    for ds in datasets:
        row = [ds.ljust(15, " ")]
        allr = etr.run_sr_many(ds, [tuple(sc) for sc in scales])     # every image x every scale: one ragged launch pair
        for a, b in scales:
            r = np.asarray(allr[(a, b)])
            row.append("{:.2f}/{:.4f}".format(np.mean(r[:, 0]), np.mean(r[:, 1])))
        lines.append("\t".join(row))
    return lines


def warp_table(etr, datasets=("Set5",), modes=("isc", "osc")):
    """The printed table of eval_lut_warp.py:305-330."""
    lines = ["\t".join(["Scale".ljust(15, " ")] + ["{}\t".format(m) for m in modes])]
    for ds in datasets:
        row = [ds.ljust(15, " ")]
        for m in modes:
            r = np.asarray(etr.run_warp(ds, m))
            row.append("{:.2f}".format(np.mean(r[:, 0])))
        lines.append("\t".join(row))
    return lines


def parse(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("task", choices=["sr", "warp"])
    # names and defaults of common/option.py:21-35, 214-217
    ap.add_argument("--expDir", "-e", type=str, default="", help="directory holding {lutName}_s{1,2}_{mode}r{0,1}.npy")
    ap.add_argument("--lutName", type=str, default="LUTft")
    ap.add_argument("--modes", type=str, default="sct")
    ap.add_argument("--modes2", type=str, default="sct")
    ap.add_argument("--suppSize", type=int, default=2)
    ap.add_argument("--maxSigma", type=int, default=10)
    ap.add_argument("--linear", action="store_true", default=False)
    ap.add_argument("--testDir", type=str, default=None)
    ap.add_argument("--resultRoot", type=str, default="")
    ap.add_argument("--datasets", type=str, nargs="+", default=["Set5"])
    opt = ap.parse_args(argv)
    if opt.testDir is None:
        opt.testDir = "./data/rrBenchmark" if opt.task == "sr" else "./data/WarpBenchmark"
    return opt


def main(argv=None):
    opt = parse(argv)
    luts = LutSet.from_dir(opt.expDir, linear=opt.linear, lut_name=opt.lutName, modes=opt.modes, modes2=opt.modes2)
    etr = Eltr(opt, luts)
    for line in (sr_table(etr, opt.datasets) if opt.task == "sr" else warp_table(etr, opt.datasets)):
        print(line)


if __name__ == "__main__":
    main()
