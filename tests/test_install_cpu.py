"""CPU-only: the package installs (`pip install --no-build-isolation .`, setup.py / pyproject.toml) and then imports from any
directory WITHOUT the repository on sys.path -- the import sites of the reference (resample/eval_lut_sr.py:10,
resample/eval_lut_warp.py:16-17) swapped for the mirrors of INTEGRATION.md section 2 -- with the libraries and the shipped LUTs
travelling as package data."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHECK = r"""
import os, sys
assert not any(os.path.abspath(p) == %r for p in sys.path if p), sys.path
import lerf_pytorch_amd as L
from lerf_pytorch_amd.resize_right.resize_right2d_numpy import SteeringGaussianResize2dNumpy, AmplifiedLinearResize2dNumpy
from lerf_pytorch_amd.resize_right.resize_right2d_torch import SteeringGaussianResize2dTorch
from lerf_pytorch_amd.resample.eval_lut_sr import FourSimplexInterpFaster, mode_pad_dict
here = os.path.dirname(L.__file__)
assert os.path.abspath(here).startswith(os.path.abspath(sys.argv[1])), here
assert L._lib.lib().lerf_abi_version() == 7
assert os.path.dirname(L._lib.LIB_PATH) == here
luts = L.load_lut_arrays(os.path.join(here, "assets", "models", "lerf-g"))
assert luts["s1_sr0"].shape[0] == 17 ** 4 and mode_pad_dict["t"] == 3
print("installed ok")
"""


def test_pip_install_then_import_from_anywhere(tmp_path):
    target = tmp_path / "site"
    r = subprocess.run([sys.executable, "-m", "pip", "install", "--no-build-isolation", "--no-deps", "--quiet", "--target", str(target), REPO],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    env["PYTHONPATH"] = str(target)
    r = subprocess.run([sys.executable, "-c", CHECK % REPO, str(target)], capture_output=True, text=True, cwd=str(tmp_path), env=env, timeout=300)
    assert r.returncode == 0 and "installed ok" in r.stdout, r.stdout + r.stderr
