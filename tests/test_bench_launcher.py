"""bench.py's N > 1 path without a GPU: the launcher (child count, per-rank environment, rank -> device) and the
statistics path (per-image numbers through StatsAccumulator and a gloo all-reduce), run for real with 2 ranks."""
import json
import os
import subprocess
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def test_rank_environment_and_device_mapping():
    env = bench.child_env(3, 8, 29511, base={"PATH": "/bin"})
    assert (env["RANK"], env["LOCAL_RANK"], env["WORLD_SIZE"], env["LOCAL_WORLD_SIZE"]) == ("3", "3", "8", "8")
    assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "29511" and env["PATH"] == "/bin"
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert bench.device_for(5, dry_run=False) == torch.device("cuda", 5)
    assert bench.device_for(5, dry_run=True) == torch.device("cpu")


def test_launcher_starts_one_child_per_gpu_before_any_gpu_call(monkeypatch):
    started = []

    class FakeProc:
        def __init__(self, cmd, env=None, stdout=None, text=None):
            started.append((cmd, env, stdout))
            self.returncode = 0
            if stdout is not subprocess.DEVNULL:            # rank 0's output goes to a file of the launcher's
                stdout.write('{"metric": "x", "n_gpus": 4}\n')

        def wait(self, timeout=None):
            return 0

        def poll(self):
            return 0

    monkeypatch.setattr(bench.subprocess, "Popen", FakeProc)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    argv = ["--gpus", "4", "--dry-run", "--steps", "1"]
    assert bench.main(argv) == 0
    assert len(started) == 4
    assert [e["RANK"] for _, e, _ in started] == ["0", "1", "2", "3"]
    assert all(e["WORLD_SIZE"] == "4" and e["LOCAL_RANK"] == e["RANK"] for _, e, _ in started)
    assert len({e["MASTER_PORT"] for _, e, _ in started}) == 1
    assert all(cmd[1].endswith("bench.py") and cmd[2:] == argv for cmd, _, _ in started)
    assert started[0][2] is not subprocess.DEVNULL and all(s == subprocess.DEVNULL for _, _, s in started[1:])
    assert not torch.cuda.is_initialized()                  # the parent has not touched a GPU


def test_launched_by_a_distributed_runner_does_not_spawn(monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "2")
    called = []
    monkeypatch.setattr(bench, "spawn_ranks", lambda *a: called.append("spawn") or 0)
    monkeypatch.setattr(bench, "run_rank", lambda a: called.append("rank") or 0)
    assert bench.main(["--gpus", "2"]) == 0 and called == ["rank"]


def test_two_rank_dry_run_reduces_the_real_statistics():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-run", "--batch", "3",
                        "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["backend"] == "gloo" and line["device"] == "cpu"
    assert line["env"]["WORLD_SIZE"] == "2" and line["env"]["MASTER_ADDR"] == "127.0.0.1"
    from compression_amd.distributed import StatsAccumulator
    single = StatsAccumulator()
    for g in range(6):                                       # rank r, image i -> global image r*3 + i
        single.add_image(bits=1000.0 * (g + 1), pixels=512 * 768, sq_err=0.5 * (g + 1), nbytes=125 * (g + 1), seconds=0.01,
                         msssim=0.9 + 0.001 * g)
    want = single.result()
    for k, v in want.items():
        assert line["stats"][k] == pytest.approx(v, rel=1e-12), k
    assert line["stats"]["images"] == 6


def test_a_rank_that_dies_before_the_rendezvous_ends_the_job():
    """Rank 1 exits with an error before it joins the process group; rank 0 would wait in the rendezvous for ever.  The
    launcher polls all ranks, ends the survivors at the first failure and returns the failing rank's code."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-run", "--fail-rank", "1",
                        "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 3, (p.returncode, p.stderr[-2000:])
    assert "rank exit codes" in p.stderr and not [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert time.monotonic() - t0 < 60


def test_ranks_that_never_finish_are_ended_at_the_deadline(monkeypatch):
    killed = []

    class Hung:
        def __init__(self, cmd, env=None, stdout=None, text=None):
            self.returncode = None

        def poll(self):
            return self.returncode

        def terminate(self):
            killed.append(self)
            self.returncode = -15

        def wait(self, timeout=None):
            return self.returncode

    monkeypatch.setattr(bench.subprocess, "Popen", Hung)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench.main(["--gpus", "2", "--dry-run", "--launch-timeout", "0.2"]) == 1
    assert len(killed) == 2


def test_counter_table_is_tied_to_the_kernel_sources(tmp_path, monkeypatch):
    """roofline.traffic comes from committed counter passes: it is reported only for the kernel sources they were
    taken on."""
    table, meta = tmp_path / "pmc.csv", tmp_path / "meta.json"
    table.write_text('Kernel,FETCH_SIZE,WRITE_SIZE\n"conv_igemm_x3_kernel<6, 1, false, 8>",1000,500\n')
    monkeypatch.setattr(bench, "PMC_TABLE", str(table))
    monkeypatch.setattr(bench, "PMC_META", str(meta))
    meta.write_text(json.dumps({"commit": "abc", "sources_sha256_16": bench.sources_digest()}))
    assert bench.pmc_traffic("conv_igemm_x3_kernel<6, 1, false, 8>") == ((2 * 1000 + 500) * 1024.0, "abc")
    meta.write_text(json.dumps({"commit": "abc", "sources_sha256_16": "0" * 16}))
    assert bench.pmc_traffic("conv_igemm_x3_kernel<6, 1, false, 8>") == (None, "abc")
