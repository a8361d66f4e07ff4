"""Net -> LUT transfer: drop-in for the reference's resample/transfer_to_lut.py.

The reference script (:85-170) loads the pickled `SRNetsSWF2` module (nine SRNet MLPs, resample/model.py:81-99),
enumerates the 17^4 sampled pixel tuples (`get_input_tensor`, :12-42), places them in each mode's receptive field
(`get_mode_input_tensor`, :45-81), runs the module and saves `round(clamp(y, -1, 1) * 127)` as int8 `LUT_<key>.npy`.
Here the nine MLPs run in one HIP kernel each (`lerf_srnet_to_lut`, float32-input MFMA for the hidden layers); the
CNN classes themselves are not mirrored -- the weights come as plain arrays (`srnets_weights.npz`, exported from the
reference checkpoint by tests/golden/gen_transfer_golden.py, or any mapping with the module's state_dict keys).

    python -m lerf_pytorch_amd.resample.transfer_to_lut -e <expDir> [--modes sct --modes2 sct --outC 3 --interval 4]

writes <expDir>/LUT_s{1,2}_<mode>r{0,1}.npy with the reference's shapes: (83521, outC, 1, 1) for stage 2,
(83521, 1, 1, 1) for stage 1 (scripts.sh:19-24).
"""
from __future__ import annotations

import argparse
import os

import numpy as np

from .. import _lib

_LAYERS = ["conv1.conv", "conv2.conv1.conv", "conv3.conv1.conv", "conv4.conv1.conv", "conv5.conv1.conv", "conv6.conv"]


def get_input_tensor(opt):
    """[L^4, 1, 2, 2] float32 device tensor of the sampled pixel tuples / 255 (transfer_to_lut.py:12-42): base values
    0, 16, ..., 240, 255 for interval 4; pixel (0,0) is the slowest-varying one."""
    torch = _lib.require_gpu()
    interval = int(getattr(opt, "interval", opt))
    base = np.arange(0, 257, 2 ** interval)
    base[-1] -= 1
    L = len(base)
    grid = np.stack(np.meshgrid(base, base, base, base, indexing="ij"), axis=-1).reshape(L ** 4, 4)
    # divided on the host: a correctly rounded float32 division like the reference's CPU tensors (the device would
    # multiply by the reciprocal, one ulp off for some values)
    x = (grid.astype(np.float32) / np.float32(255.0)).reshape(-1, 1, 2, 2)
    return torch.from_numpy(x).cuda()


_PLACEMENT = {          # mode -> (field size, positions of the four values (0,0),(0,1),(1,0),(1,1) of the 2x2 input)
    "d": (3, [(0, 0), (0, 2), (2, 0), (2, 2)]),
    "y": (3, [(0, 0), (1, 1), (1, 2), (2, 1)]),
    "c": (4, [(0, 0), (0, 1), (0, 2), (0, 3)]),
    "t": (4, [(0, 0), (1, 1), (2, 2), (3, 3)]),
}


def get_mode_input_tensor(input_tensor, mode):
    """the 2x2 tuple scattered into the mode's receptive field (transfer_to_lut.py:45-81); mode 's' is the 2x2 itself"""
    torch = _lib.require_gpu()
    if mode not in _PLACEMENT:
        raise ValueError("Mode {} not implemented.".format(mode))
    K, pos = _PLACEMENT[mode]
    out = torch.zeros((input_tensor.shape[0], input_tensor.shape[1], K, K), dtype=input_tensor.dtype, device=input_tensor.device)
    for (sy, sx), (dy, dx) in zip([(0, 0), (0, 1), (1, 0), (1, 1)], pos):
        out[:, :, dy, dx] = input_tensor[:, :, sy, sx]
    return out


def pack_srnet_weights(weights, key):
    """state_dict arrays of one SRNet ('<key>.model.conv1.conv.weight', ...) -> the flat float32 layout of
    lerf_srnet_to_lut, and outC."""
    parts = []
    outC = None
    for li, name in enumerate(_LAYERS):
        w = np.asarray(weights["%s.model.%s.weight" % (key, name)], dtype=np.float32)
        b = np.asarray(weights["%s.model.%s.bias" % (key, name)], dtype=np.float32)
        w = w.reshape(w.shape[0], -1)
        want_in = 4 if li == 0 else li * 64
        if w.shape[1] != want_in or (li < 5 and w.shape[0] != 64):
            raise ValueError("unexpected shape %s for %s.%s (nf = 64, dense 1x1 layers)" % (w.shape, key, name))
        if li == 5:
            outC = w.shape[0]
        parts += [w.reshape(-1), b.reshape(-1)]
    flat = np.concatenate(parts)
    assert flat.size == _lib.lib().lerf_srnet_weight_floats(outC)
    return flat, outC


def srnet_to_lut(weights, key, interval=4, return_float=False):
    """int8 [L^4, outC] LUT of the network `key` (e.g. 's2_cr1'); with return_float also its float32 outputs."""
    torch = _lib.require_gpu()
    flat, outC = pack_srnet_weights(weights, key)
    L = 2 ** (8 - interval) + 1
    wd = torch.from_numpy(flat).cuda()
    lut = torch.empty((L ** 4, outC), dtype=torch.int8, device="cuda")
    y = torch.empty((L ** 4, outC), dtype=torch.float32, device="cuda") if return_float else None
    _lib.check(_lib.lib().lerf_srnet_to_lut(wd.data_ptr(), outC, int(interval), lut.data_ptr(),
                                            y.data_ptr() if y is not None else None, _lib.current_stream()), "lerf_srnet_to_lut")
    return (lut, y) if return_float else lut


def transfer(weights, modes="sct", modes2="sct", interval=4):
    """all LUTs of a model: {key: int8 ndarray [L^4, outC]} in the reference's order (stage 2 first, :96-170)"""
    out = {}
    for mode in modes2:
        for r in (0, 1):
            key = "s2_%sr%d" % (mode, r)
            out[key] = srnet_to_lut(weights, key, interval).cpu().numpy()
    for mode in modes:
        key = "s1_%sr0" % mode
        out[key] = srnet_to_lut(weights, key, interval).cpu().numpy()
    return out


def load_weights(path):
    """srnets_weights.npz (or a directory holding it)"""
    if os.path.isdir(path):
        path = os.path.join(path, "srnets_weights.npz")
    return dict(np.load(path))


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("-e", "--expDir", required=True, help="directory with srnets_weights.npz; LUT_*.npy are written there")
    ap.add_argument("--modes", default="sct")
    ap.add_argument("--modes2", default="sct")
    ap.add_argument("--interval", type=int, default=4)
    ap.add_argument("--outDir", default=None)
    a = ap.parse_args(argv)
    luts = transfer(load_weights(a.expDir), a.modes, a.modes2, a.interval)
    dst = a.outDir or a.expDir
    for key, lut in luts.items():
        res = lut.reshape(lut.shape[0], lut.shape[1], 1, 1)                   # the reference's saved shape
        path = os.path.join(dst, "LUT_%s.npy" % key)
        np.save(path, res)
        print("Resulting LUT size: ", res.shape, "Saved to", path)


if __name__ == "__main__":
    main()
