"""bench.py --gpus N started bare must spawn its own ranks from a parent that never touches the GPU stack, hand every
rank its RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, and exit non-zero when a rank fails.
No GPU here: the ranks cannot initialise CUDA and fail, which is exactly the behaviour under test; the happy path runs on
the GPU box (tests/test_gpu_fullsize.py::test_rccl_two_rank_halo_exchange_and_stitch, bench.py itself)."""
import json
import os
import subprocess
import sys
import textwrap

from conftest import REPO


def _fake_torch(tmp_path, body):
    d = tmp_path / "fake"
    d.mkdir()
    (d / "torch.py").write_text(textwrap.dedent(body))
    return str(d)


def test_parent_stays_off_torch_and_reports_failed_ranks(tmp_path):
    # a torch that records who imported it (and then fails like a box without devices)
    log = tmp_path / "imports.log"
    fake = _fake_torch(tmp_path, """
        import os
        with open(%r, "a") as f:
            f.write("%%s %%s %%s %%s %%s\\n" %% (os.environ.get("RANK"), os.environ.get("LOCAL_RANK"), os.environ.get("WORLD_SIZE"),
                                            os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT")))
        raise ImportError("no GPU stack in this test")
        """ % str(log))
    env = dict(os.environ, PYTHONPATH=fake + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0, "a failed rank must fail the launcher"
    assert "rank(s) failed" in r.stderr
    rows = [l.split() for l in log.read_text().splitlines()]
    # only the two ranks imported torch, never the parent (whose RANK is unset -> "None")
    assert sorted(x[0] for x in rows) == ["0", "1"], rows
    for rank, local, world, addr, port in rows:
        assert local == rank and world == "2" and addr == "127.0.0.1" and port.isdigit()
    assert len({x[4] for x in rows}) == 1, "one rendezvous port for all ranks"
    # no JSON line claims a result
    assert not any(l.startswith("{") and "n_gpus" in l for l in r.stdout.splitlines())


def test_single_rank_line_is_not_spawned(tmp_path):
    # --gpus 1 runs in-process: with the failing torch the process itself fails, and no child is started
    log = tmp_path / "imports.log"
    fake = _fake_torch(tmp_path, """
        import os
        open(%r, "a").write("%%s\\n" %% os.environ.get("LERF_BENCH_SPAWNED"))
        raise ImportError("no GPU stack in this test")
        """ % str(log))
    env = dict(os.environ, PYTHONPATH=fake + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LERF_BENCH_SPAWNED"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert log.read_text().split() == ["None"], "exactly one import, by the un-spawned process"
