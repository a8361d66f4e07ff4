"""LUT sets: loading the reference's int8 `.npy` tables and keeping them
resident in HBM (resample/eval_lut_sr.py:750-775 is the loader this mirrors)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib

ASSET_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "models")


def load_lut_arrays(model_dir, linear=False, lut_name="LUTft", modes="sct", modes2="sct", stages=2):
    """{'s1_sr0': int8 [17^4,1], ..., 's2_tr1': int8 [17^4,oC]} (host, numpy).

    Same keys and file naming as the reference's lutDict; values stay int8
    (the reference widens them to float32, the arithmetic is integer either way).
    """
    if stages != 2:
        raise NotImplementedError("only the shipped 2-stage models are supported")
    oC = 1 if linear else 3
    d = {}
    for mode in modes:
        a = np.load(os.path.join(model_dir, "{}_s1_{}r0.npy".format(lut_name, mode)))
        d["s1_{}r0".format(mode)] = np.ascontiguousarray(a.astype(np.int8).reshape(-1, 1))
    for mode in modes2:
        for r in (0, 1):
            a = np.load(os.path.join(model_dir, "{}_s2_{}r{}.npy".format(lut_name, mode, r)))
            d["s2_{}r{}".format(mode, r)] = np.ascontiguousarray(a.astype(np.int8).reshape(-1, oC))
    for k, v in d.items():
        if v.shape[0] != _lib.LERF_LUT_ENTRIES:
            raise ValueError("LUT {} has {} entries, expected 17^4 (interval=4)".format(k, v.shape[0]))
    return d


class LutSet:
    """Device-resident LUT set of one model + the `lerf_luts_t` descriptor."""

    def __init__(self, arrays: dict, oC: int, modes="sct", modes2="sct", device=None):
        torch = _lib.require_gpu()
        self.device = torch.device(device if device is not None else "cuda")
        self.oC = int(oC)
        self.modes, self.modes2 = modes, modes2
        if not (1 <= len(modes) <= _lib.LERF_MAX_MODES and 1 <= len(modes2) <= _lib.LERF_MAX_MODES):
            raise ValueError("1..5 modes per stage")
        for m in modes + modes2:
            _lib.mode_offsets(m, 0)          # ValueError("Mode x not implemented.")
        self.tensors = {}
        st = _lib.Luts()
        st.n_modes1, st.n_modes2, st.oC = len(modes), len(modes2), self.oC
        st.modes1 = modes.encode()
        st.modes2 = modes2.encode()
        for i, m in enumerate(modes):
            t = self._up(arrays["s1_{}r0".format(m)], 1)
            self.tensors["s1_{}r0".format(m)] = t
            st.s1[i] = t.data_ptr()
        for i, m in enumerate(modes2):
            for r in (0, 1):
                key = "s2_{}r{}".format(m, r)
                t = self._up(arrays[key], self.oC)
                self.tensors[key] = t
                st.s2[i][r] = t.data_ptr()
        self.struct = st
        st.fused_pack = None
        nb = int(_lib.lib().lerf_fused_lutpack_bytes(C.byref(st)))       # 0: no tile-fused kernel for this set (5 modes in a stage)
        if nb > 0:
            pack = torch.empty(nb, dtype=torch.uint8, device=self.device)
            _lib.check(_lib.lib().lerf_fused_lutpack_build(C.byref(st), pack.data_ptr(), _lib.current_stream()),
                       "lerf_fused_lutpack_build")
            self.tensors["fused_pack"] = pack
            st.fused_pack = pack.data_ptr()
        self.nbytes = sum(v.numel() for k, v in self.tensors.items() if k != "fused_pack")

    def _up(self, a, oC):
        import torch
        a = np.ascontiguousarray(np.asarray(a).astype(np.int8).reshape(-1, oC))
        if a.shape[0] != _lib.LERF_LUT_ENTRIES:
            raise ValueError("LUT must have 17^4 entries")
        return torch.from_numpy(a).to(self.device)

    @classmethod
    def from_dir(cls, model_dir, linear=False, lut_name="LUTft", modes="sct", modes2="sct", device=None):
        arrays = load_lut_arrays(model_dir, linear, lut_name, modes, modes2)
        return cls(arrays, 1 if linear else 3, modes, modes2, device)

    @classmethod
    def from_arrays(cls, arrays, modes="sct", modes2="sct", device=None):
        """From a {key: int8 [17^4, oC]} dict, e.g. the output of resample.transfer_to_lut.transfer(); the number of
        hyper-parameter channels is read off the stage-2 tables (3 = LeRF-G, 1 = LeRF-L)."""
        a = np.asarray(arrays["s2_{}r0".format(modes2[0])])
        oC = int(a.reshape(a.shape[0], -1).shape[1])
        return cls(arrays, oC, modes, modes2, device)

    @classmethod
    def shipped(cls, name="lerf-g", device=None):
        """The LUTs shipped with the reference (models/lerf-g, models/lerf-l)."""
        return cls.from_dir(os.path.join(ASSET_DIR, name), linear=(name == "lerf-l"), device=device)

    def ref(self):
        return C.byref(self.struct)
