"""The trainable LUT model of the reference, on the MI355X path: mirror of `SWF2LUT` in resample/model.py:130-431
(`InterpTorchBatch`, `forward`, `predict`) -- LUT fine-tuning, scripts.sh:28-30.

The LUT pass runs in liblerf_hip.so (lerf_swf2lut_interp_f32 / _bwd_f32) behind a torch.autograd.Function whose
backward is the gradient autograd derives for the reference code: into the LUT parameters (straight-through round,
clamp gate) and into the input through the LSB terms.  Everything else in `predict` (rot90, replicate pad, the
straight-through rounding between passes) is the reference's own torch glue.

Like the reference, modes "c" and "t" of this twin read their LSBs at the 'y' pattern pixels (model.py:229-232,
240-243); the deploy-time numpy pass (eval_lut_sr.FourSimplexInterpFaster) does not.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib

mode_pad_dict = {"s": 1, "d": 2, "y": 2, "c": 3, "t": 3, "e": 3, "l": 3, "f": 4, "m": 4, "g": 5, "n": 5}


def round_func(input):
    """Backward-pass differentiable approximation of round (model.py:16-22)."""
    forward_value = torch.round(input)
    out = input.clone()
    out.data = forward_value.data
    return out


def _mode_char(mode):
    if not isinstance(mode, str) or len(mode) != 1 or mode not in "sdyct":
        raise ValueError("Mode {} not implemented.".format(mode))          # model.py:246
    return mode.encode()


class _InterpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, img_in, outC, mode, bd):
        if not (weight.is_cuda and img_in.is_cuda):
            raise ValueError("SWF2LUT runs on the GPU (there is no CPU path)")
        w = weight.detach().contiguous().float()
        x = img_in.detach().contiguous().float()
        B, Cn, hp, wp = x.shape
        h, wd = hp - bd, wp - bd
        if w.shape != (_lib.LERF_LUT_ENTRIES, outC):
            raise ValueError("weight must be [17^4, outC]")
        out = torch.empty((B, Cn * outC, h, wd), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().lerf_swf2lut_interp_f32(C.c_void_p(w.data_ptr()), int(outC), _mode_char(mode),
                                                      C.c_void_p(x.data_ptr()), B * Cn, h, wd, int(bd),
                                                      C.c_void_p(out.data_ptr()), _lib.current_stream()),
                   "lerf_swf2lut_interp_f32")
        ctx.save_for_backward(w, x)
        ctx.meta = (int(outC), mode, int(bd))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        w, x = ctx.saved_tensors
        outC, mode, bd = ctx.meta
        B, Cn, hp, wp = x.shape
        g = grad_out.contiguous().float()
        gw = torch.zeros_like(w) if ctx.needs_input_grad[0] else None
        gx = torch.zeros_like(x) if ctx.needs_input_grad[1] else None
        _lib.check(_lib.lib().lerf_swf2lut_interp_bwd_f32(
            C.c_void_p(w.data_ptr()), outC, _mode_char(mode), C.c_void_p(x.data_ptr()), C.c_void_p(g.data_ptr()),
            B * Cn, hp - bd, wp - bd, bd, C.c_void_p(gw.data_ptr() if gw is not None else None),
            C.c_void_p(gx.data_ptr() if gx is not None else None), _lib.current_stream()), "lerf_swf2lut_interp_bwd_f32")
        return gw, gx, None, None, None


class SWF2LUT(nn.Module):
    """model.py:130-431.  `opt` carries modes, modes2, stages, norm, interval, expDir (common/option.py:21-35);
    the LUT files are `<expDir>/<lutName>_s{stage}_{mode}r{r}.npy` with lutName = "LUT" like the reference
    (`opt.lutName` selects another prefix, e.g. the shipped "LUTft")."""

    def __init__(self, opt, inC=1, outC=3):
        super(SWF2LUT, self).__init__()
        self.modes2 = opt.modes2
        self.modes = opt.modes
        self.stages = opt.stages
        self.norm = opt.norm
        self.interval = opt.interval
        if self.interval != 4 or self.stages != 2:
            raise NotImplementedError("only interval=4, stages=2 (the shipped models) are implemented")
        name = getattr(opt, "lutName", None) or "LUT"
        self.outC = outC
        for mode in self.modes2:                                # hyper stage (:140-149)
            for r in [0, 1]:
                key = "s{}_{}r{}".format(2, mode, r)
                arr = np.load(os.path.join(opt.expDir, "{}_{}.npy".format(name, key))).reshape(-1, outC).astype(np.float32) / 127.0
                self.register_parameter(name="weight_" + key, param=torch.nn.Parameter(torch.Tensor(arr)))
        for mode in self.modes:                                 # stage 1 (:151-158)
            key = "s{}_{}r{}".format(1, mode, 0)
            arr = np.load(os.path.join(opt.expDir, "{}_{}.npy".format(name, key))).reshape(-1, 1).astype(np.float32) / 127.0
            self.register_parameter(name="weight_" + key, param=torch.nn.Parameter(torch.Tensor(arr)))

    round_func = staticmethod(round_func)

    def InterpTorchBatch(self, weight, outC, mode, img_in, bd):
        """[B, C, h+bd, w+bd] (integer-valued float32) -> [B, C*outC, h, w] (:172-385)."""
        _mode_char(mode)
        return _InterpFn.apply(weight, img_in, outC, mode, bd)

    def forward(self, x, stage, mode, r):
        key = "s{}_{}r{}".format(str(stage), mode, r)
        pad = mode_pad_dict[mode]
        outC = 1 if stage == 1 else self.outC
        return self.InterpTorchBatch(getattr(self, "weight_" + key), outC, mode, x, pad)

    def _rotation_ensemble(self, x, modes, stage, lut_of_rotation):
        """Sum over modes and the four quarter-turns of one stage: rotate, replicate-pad bottom/right by the mode's
        reach, LUT pass, rotate back, straight-through round (model.py:405-412, 419-424)."""
        total = 0
        for mode in modes:
            reach = mode_pad_dict[mode]
            for quarter_turns in range(4):
                rotated = F.pad(torch.rot90(x, quarter_turns, [2, 3]), (0, reach, 0, reach), mode="replicate")
                passed = self.forward(rotated, stage=stage, mode=mode, r=lut_of_rotation(quarter_turns))
                total = total + round_func(torch.rot90(passed, (4 - quarter_turns) % 4, [2, 3]))
        return total

    def predict(self, x, stage=None):
        """x in [0, 1]; stage 2 -> hyper-parameter maps in [0, 1] ([B, outC, H, W] per input channel), otherwise the
        pre-filtered image in 0..255 (model.py:398-431)."""
        x = round_func(x * 255.0)                                             # 8-bit input
        if stage == 2:
            # hyper stage: LUT r0 serves rotations 0 and 2, LUT r1 rotations 1 and 3; mean over 4 x modes, + 127
            pred = self._rotation_ensemble(x, self.modes2, self.stages, lambda q: q & 1)
            return torch.clamp(round_func(pred / (len(self.modes2) * 4) + self.norm // 2), 0, self.norm) / float(self.norm)
        # feature stage(s): one LUT per mode for all four rotations
        for s in range(self.stages - 1):
            pred = self._rotation_ensemble(x, self.modes, s + 1, lambda q: 0)
            if s + 1 == self.stages - 1:        # the last feature stage keeps the 0..255 range (divide by the modes only)
                x = torch.clamp(round_func(pred / len(self.modes)), 0, self.norm)
            else:
                x = torch.clamp(round_func(pred / (len(self.modes) * 4)) + self.norm // 2, 0, self.norm) / float(self.norm)
        return x


def export_luts(model: SWF2LUT, exp_dir: str, lut_name: str = "LUTft"):
    """Write the fine-tuned int8 LUTs the way train_model.py:481-497 does:
    `<exp_dir>/<lut_name>_s{stage}_{mode}r{r}.npy = round(clip(weight, -1, 1) * 127).astype(int8)` -- the files
    eval_lut_sr / eval_lut_warp (here: LutSet.from_dir, LerfEngine) load."""
    paths = []
    os.makedirs(exp_dir, exist_ok=True)
    keys = ["s2_{}r{}".format(m, r) for m in model.modes2 for r in (0, 1)] + ["s1_{}r0".format(m) for m in model.modes]
    for key in keys:
        w = getattr(model, "weight_" + key).detach().cpu().numpy()
        p = os.path.join(exp_dir, "{}_{}.npy".format(lut_name, key))
        np.save(p, np.round(np.clip(w, -1, 1) * 127).astype(np.int8))
        paths.append(p)
    return paths


def mulut_predict(model_G, x, stage=1, inC=1):
    """train_model.py:38-46: the model sees one channel at a time when inC == 1."""
    if inC == 1:
        return torch.cat([model_G.predict(x[:, i:i + 1, :, :], stage=stage) for i in range(x.shape[1])], dim=1)
    return model_G.predict(x, stage=stage)


def lutft_step(model_G, resizer, im, lb, opt_G=None, linear=False, norm=255, featC=1, reduce_grads=None):
    """One LUT fine-tuning iteration, train_model.py:416-442 (`--lutft --twoStage`): stage 1 -> stage 2 -> spatially
    varying resize -> clamp -> MSE against the HR patch; backward; optimiser step.  `resizer` is a torch-facing
    resampler with set_shape already called for im's shape.  `reduce_grads(model)` (e.g. dist.allreduce_grads) runs
    between backward and step for data-parallel training.  Returns the loss tensor."""
    if opt_G is not None:
        opt_G.zero_grad()
    feat_im = mulut_predict(model_G, im, 1)
    hyper_in = feat_im / float(norm)
    pred_hyper = mulut_predict(model_G, hyper_in, 2)
    if linear:
        pred = resizer.resize(feat_im, pred_hyper)
    else:
        pred = resizer.resize(feat_im, pred_hyper[:, :1 * featC, :, :], pred_hyper[:, 1 * featC:2 * featC, :, :],
                              pred_hyper[:, 2 * featC:, :, :])
    pred = torch.clamp(pred, 0, norm) / float(norm)
    loss_G = F.mse_loss(pred, lb)
    if loss_G.requires_grad:
        loss_G.backward()
        if reduce_grads is not None:
            reduce_grads(model_G)
        if opt_G is not None:
            opt_G.step()
    return loss_G
