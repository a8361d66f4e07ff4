"""GPU tests of the evaluation harness: device metrics against the oracle / the reference's
golden values, and the Set5 tables of the reference's scripts.sh:33-47 reproduced end to end on the MI355X path.

Tolerances: PSNR / mPSNR 1e-4 dB (float32 summation order differs from numpy's pairwise mean), SSIM 1e-9."""
import json
import os

import numpy as np
import pytest
from PIL import Image

from conftest import ASSETS, DATA, GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def M():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from lerf_pytorch_amd import metrics
    return metrics


@pytest.mark.parametrize("shape,shave", [((40, 52), 2), ((64, 33), 4), ((11, 11), 0), ((300, 257), 3)])
def test_metrics_vs_oracle_random(M, oracle, shape, shave):
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    gt = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
    out = np.clip(gt.astype(int) + rng.integers(-12, 13, gt.shape), 0, 255).astype(np.uint8)
    assert abs(M.psnr_y(gt, out, shave) - oracle.psnr_y(gt, out, shave)) < 1e-4
    assert abs(M.ssim_y(gt, out) - oracle.ssim_y(gt, out)) < 1e-9
    mask = rng.integers(0, 2, gt.shape).astype(bool)
    assert abs(M.mpsnr(out, gt, mask) - oracle.mpsnr(out, gt, mask)) < 1e-4


def test_metrics_crop_rule_and_views(M, oracle):
    """eval_lut_sr.py:735-739: prediction larger than GT in one axis, smaller in the other."""
    rng = np.random.default_rng(5)
    gt = rng.integers(0, 256, (50, 60, 3), dtype=np.uint8)
    out = rng.integers(0, 256, (52, 57, 3), dtype=np.uint8)
    assert abs(M.psnr_y(gt, out, 2) - oracle.psnr_y(gt, out, 2)) < 1e-4
    assert abs(M.ssim_y(gt, out) - oracle.ssim_y(gt, out)) < 1e-9


def test_metrics_errors(M):
    a = np.zeros((8, 8, 3), np.uint8)
    with pytest.raises(ValueError):
        M.ssim_y(a, a)                       # smaller than the 11x11 window
    with pytest.raises(ValueError):
        M.psnr_y(a, a, 4)                    # nothing left after the shave
    with pytest.raises(ValueError):
        M.psnr_y(a.astype(np.float32), a, 0)
    with pytest.raises(ValueError):
        M.mpsnr(a, a, np.zeros((8, 8), bool))


def test_set5_metric_known_answers(M):
    """PSNR / SSIM of the reference functions (g5 / g8 goldens) on the GPU outputs of two scales."""
    import lerf_pytorch_amd as L
    ref = json.load(open(os.path.join(GOLDEN, "g8_ssim.json")))
    for model, scale in (("lerf-g", 2), ("lerf-l", 4)):
        eng = L.LerfEngine.shipped(model)
        for n in ("baby", "butterfly"):
            lr = np.array(Image.open(os.path.join(DATA, "LR_bicubic/rrLR_X%.2f_%.2f" % (scale, scale), n + ".png")))
            gt = np.array(Image.open(os.path.join(DATA, "HR", n + ".png")))
            out = eng.sr(eng._dev(lr)[0], scale)
            r = ref["%s/x%d/%s" % (model, scale, n)]
            assert abs(M.psnr_y(gt, out, scale) - r["psnr_y"]) < 1e-4
            assert abs(M.ssim_y(gt, out) - r["ssim"]) < 1e-9


@pytest.mark.parametrize("model,linear,row", [
    ("lerf-g", False, ["35.71/0.9475", "32.02/0.8980", "30.15/0.8548"]),          # scripts.sh:39-41
    ("lerf-l", True, ["34.84/0.9432", "30.72/0.8773", "29.13/0.8270"]),           # scripts.sh:35-37
])
def test_sr_table_of_scripts_sh(model, linear, row, tmp_path):
    from lerf_pytorch_amd.resample import eval_harness as EH
    argv = ["sr", "--testDir", os.path.dirname(DATA), "--resultRoot", str(tmp_path), "-e", os.path.join(ASSETS, model)]
    opt = EH.parse(argv + (["--linear"] if linear else []))
    etr = EH.Eltr(opt, EH.LutSet.from_dir(opt.expDir, linear=opt.linear))
    lines = EH.sr_table(etr)
    assert lines[0].split() == ["Scale", "2.0x2.0", "3.0x3.0", "4.0x4.0"]
    assert lines[1].split() == ["Set5"] + row
    rdir = tmp_path / model / "X2.00_2.00" / "Set5"
    assert sorted(os.listdir(rdir))[:2] == ["baby_LUTft.png", "baby_gt.png"]
    ref = json.load(open(os.path.join(GOLDEN, "g5_set5.json")))["sr"]
    import hashlib
    saved = np.array(Image.open(rdir / "woman_LUTft.png"))
    assert hashlib.md5(saved.tobytes()).hexdigest() == ref["%s/x2/woman" % model]["md5_out"]


@pytest.mark.parametrize("model,linear,row", [
    ("lerf-g", False, ["33.81", "27.89"]),                                        # scripts.sh:46-47
    ("lerf-l", True, ["32.90", "27.13"]),                                         # scripts.sh:43-45
])
def test_warp_table_of_scripts_sh(model, linear, row, tmp_path):
    from lerf_pytorch_amd.resample import eval_harness as EH
    argv = ["warp", "--testDir", os.path.dirname(DATA), "--resultRoot", str(tmp_path), "-e", os.path.join(ASSETS, model)]
    opt = EH.parse(argv + (["--linear"] if linear else []))
    etr = EH.Eltr(opt, EH.LutSet.from_dir(opt.expDir, linear=opt.linear))
    lines = EH.warp_table(etr)
    assert lines[0].split() == ["Scale", "isc", "osc"]
    assert lines[1].split() == ["Set5"] + row
    assert os.path.exists(tmp_path / model / "Set5" / "isc" / "head_mask.png")
