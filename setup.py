"""Packaging of the MI355X LeRF path: `pip install --no-build-isolation [-e] .` makes `import lerf_pytorch_amd` work from any
directory (the reference's scripts import their resamplers the same way, resample/eval_lut_sr.py:10, eval_lut_warp.py:16-17).

The package lives in the directory `lerf-pytorch_amd/` (the repository layout); `package_dir` maps it to the importable name.
The two shared libraries are built by `make -C lerf-pytorch_amd/csrc` (hipcc --offload-arch=gfx950, g++ against the installed
PyTorch-ROCm) before the files are collected when they are missing; they and the shipped LUTs travel as package data.
--no-build-isolation: the build needs the interpreter's own torch (the torch.ops.lerf.* library links against it) and the
image has no package index."""
import os
import subprocess

from setuptools import setup
from setuptools.command.build_py import build_py

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "lerf-pytorch_amd")


def build_native():
    if not (os.path.exists(os.path.join(PKG, "liblerf_hip.so")) and os.path.exists(os.path.join(PKG, "liblerf_torch.so"))):
        subprocess.check_call(["make", "-j4", "-C", os.path.join(PKG, "csrc"), "ARCH=gfx950"])


class BuildPy(build_py):
    def run(self):
        build_native()
        build_py.run(self)


try:                                   # editable installs (PEP 660) go through `develop` on this setuptools
    from setuptools.command.develop import develop

    class Develop(develop):
        def run(self):
            build_native()
            develop.run(self)
    extra_cmds = {"develop": Develop}
except ImportError:                    # pragma: no cover
    extra_cmds = {}

setup(
    name="lerf-pytorch-amd",
    version="0.6.0",
    description="LeRF LUT resampling hot path on MI355X (gfx950): hand-written HIP behind the reference's Python API",
    packages=["lerf_pytorch_amd", "lerf_pytorch_amd.resample", "lerf_pytorch_amd.resize_right"],
    package_dir={"lerf_pytorch_amd": "lerf-pytorch_amd"},
    package_data={"lerf_pytorch_amd": ["*.so", "assets/models/*/*.npy", "assets/models/*/*.npz"]},
    include_package_data=False,
    python_requires=">=3.8",
    install_requires=["numpy"],
    cmdclass=dict(build_py=BuildPy, **extra_cmds),
    zip_safe=False,
)
