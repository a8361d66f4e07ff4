"""CPU-only: AddressSanitizer + UndefinedBehaviorSanitizer builds of the host-side C / C++ (GPU sanitizers are not available on
this pool): the C restatement of the oracle through its golden checks, and the host-only entry points of the C ABI (geometry
tables, pads, 3x3 inverse, pattern offsets) against the product library."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _asan_env():
    lib = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(lib) or not os.path.exists(lib):
        pytest.skip("no libasan for this gcc")
    env = dict(os.environ)
    env["LD_PRELOAD"] = lib
    env["ASAN_OPTIONS"] = "detect_leaks=0:halt_on_error=1:abort_on_error=0"          # (the interpreter itself is not leak-clean)
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    return env


def test_host_abi_functions_under_asan_ubsan():
    subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "lerf-pytorch_amd", "csrc"), "asan"])
    so = os.path.join(REPO, "lerf-pytorch_amd", "csrc", "build_asan", "liblerf_host_asan.so")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "sanitize_host_check.py"), so], env=_asan_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "sanitized host functions ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_oracle_c_golden_checks_under_asan_ubsan():
    subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "oracle"), "asan"])
    env = _asan_env()
    env["LERF_ORACLE_LIB"] = os.path.join(REPO, "oracle", "_asan", "liblerf_oracle_asan.so")
    # the golden-vector checks of the C restatement (stages, resize, warp, Set5 md5s of the small images)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(REPO, "tests", "test_oracle_c.py"), "-x", "-q", "-p", "no:cacheprovider",
                        "-k", "golden or warp or md5"], env=env, capture_output=True, text=True, timeout=1500, cwd=REPO)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.skipif(os.environ.get("LERF_TEST_SPILLS") != "1", reason="1.5 minutes of hipcc: opt in with LERF_TEST_SPILLS=1")
def test_no_kernel_spills_or_uses_scratch():
    """tools/check_spills.sh: every kernel of the library compiles for gfx950 without scratch memory (round 6 found a stage-3
    variant that had started to spill through its doubled WRITE_SIZE counter, not through its timing)"""
    out = subprocess.run(["bash", os.path.join(REPO, "tools", "check_spills.sh")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "0 kernel(s) with scratch or spills" in out.stdout, out.stdout[-2000:]
