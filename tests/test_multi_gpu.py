"""The multi-GPU leg started the way a user (and the driver) starts it.

* `python bench.py --gpus N` without torchrun around it: the parent counts the
  devices in a child, then starts the N ranks as child processes -- or says
  "N GPUs requested, M visible" within seconds (CPU test, runs here).
* On a box with at least two GPUs (skipped otherwise): `cuking --num_gpus=N` in
  both schedules over RCCL, `python -m cuking_amd.run` under torchrun and
  `bench.py --gpus N`, each compared with the one-GPU output of the same job,
  with zero allocations / host waits after the workspace reservation on every
  rank.  Replaces the reference's one-VM-per-shard fan-out
  (cloud_batch_submit.py:45,73; cuking.cu:129-152).
* The phase watchdog of the C++ host (`--phase_timeout_seconds`): a rank that
  never comes back turns into exit code 1 with every rank's phase, not a hang
  (three rank threads on one GPU over the test-only loopback collectives).
"""
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import pytest

import cuking_amd

ROOT = Path(__file__).resolve().parent.parent
CLI = ROOT / "cuking_amd" / "bin" / "cuking"


def gpu_count():
    return cuking_amd.device_count()


def run(cmd, timeout=900, env=None):
    e = dict(os.environ)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e.update(env or {})
    return subprocess.run([str(c) for c in cmd], capture_output=True, text=True, timeout=timeout,
                          cwd=str(ROOT), env=e)


def last_json(text):
    for line in reversed(text.strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise AssertionError("no JSON line in:\n" + text[-2000:])


# ------------------------------------------------------------------ CPU ----
def test_bench_gpus_n_fails_fast_when_the_gpus_are_not_there():
    """`python3 bench.py --gpus N` with fewer than N GPUs visible: one line on
    stderr, non-zero exit, within 10 s -- not a traceback, not a hang, and
    nothing of torch imported on the way."""
    seen = gpu_count()
    n = max(2, seen + 1)
    t0 = time.perf_counter()
    p = run([sys.executable, "bench.py", "--gpus", n], timeout=60)
    dt = time.perf_counter() - t0
    assert p.returncode != 0
    assert f"{n} GPUs requested, {seen} visible" in p.stderr, p.stderr
    assert "Traceback" not in p.stderr and p.stdout.strip() == ""
    assert dt < 10.0, dt
    # the rehearsal takes the same self-launch and needs one GPU
    if seen == 0:
        p = run([sys.executable, "bench.py", "--gpus", 3], timeout=60,
                env={"CUKING_BENCH_REHEARSAL": "1"})
        assert p.returncode != 0 and "3 GPUs requested, 0 visible" in p.stderr


def test_bench_refuses_a_world_size_that_does_not_match():
    p = run([sys.executable, "bench.py", "--gpus", 2], timeout=300,
            env={"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "--gpus 2 but WORLD_SIZE=3" in p.stderr
    assert "Traceback" not in p.stderr


# --------------------------------------------------- one GPU (rehearsal) ----
SMALL = ["--samples", "3000", "--sites", "20000", "--steps", "2", "--warmup", "1",
         "--extra-configs", "none", "--cpu-seconds", "0", "--no-clock-pass"]


@pytest.mark.gpu
def test_bench_self_launch_rehearsal_on_one_gpu():
    """CUKING_BENCH_REHEARSAL=1 python bench.py --gpus 3: the parent starts three
    ranks itself (all on cuda:0, gloo), rank 0 prints the JSON line, exit 0."""
    p = run([sys.executable, "bench.py", "--gpus", 3, *SMALL], timeout=900,
            env={"CUKING_BENCH_REHEARSAL": "1"})
    assert p.returncode == 0, p.stderr[-3000:]
    out = last_json(p.stdout)
    assert out["n_gpus"] == 3 and out["config"]["backend"] == "gloo"
    assert out["single_gpu_same_run"] is not None and out["speedup"] > 0
    assert out["config"]["samples"] == 3000


@pytest.mark.gpu
@pytest.mark.parametrize("phase,where", [("hang_compute", "exchange 2: counts"),
                                         ("hang_gather", "exchange 2: records")])
def test_phase_watchdog_turns_a_hang_into_exit_1(tmp_path, phase, where):
    t0 = time.perf_counter()
    p = run([CLI, "--synthetic=700,3000,5", "--output_uri", tmp_path / "out", "--num_gpus=3",
             "--collectives=loopback", f"--inject_failure=1:{phase}",
             "--phase_timeout_seconds=3"], timeout=300)
    dt = time.perf_counter() - t0
    assert p.returncode == 1, (p.stdout, p.stderr)
    assert "Error: DEADLINE_EXCEEDED: rank " in p.stderr, p.stderr
    assert f"[1] {where}" in p.stderr, p.stderr
    assert not (tmp_path / "out" / "part-00000.snappy.parquet").exists()
    assert dt < 120, dt
    # ... and a run that is merely given a limit is not disturbed by it
    p = run([CLI, "--synthetic=700,3000,5", "--output_uri", tmp_path / "ok", "--num_gpus=3",
             "--collectives=loopback", "--phase_timeout_seconds=300"], timeout=300)
    assert p.returncode == 0, p.stderr


# ------------------------------------------------------- two or more GPUs ----
needs_two = pytest.mark.skipif(gpu_count() < 2, reason="needs at least two GPUs")
SYNTH = "6000,20000,5"


@pytest.fixture(scope="module")
def one_gpu_file(tmp_path_factory):
    d = tmp_path_factory.mktemp("mg_ref")
    p = run([CLI, f"--synthetic={SYNTH}", "--output_uri", d, "--kin_threshold=0.05"])
    assert p.returncode == 0, p.stderr
    return (d / "part-00000.snappy.parquet").read_bytes()


@pytest.mark.gpu
@needs_two
@pytest.mark.parametrize("extra", [["--multi_gpu_mode=staged"], ["--multi_gpu_mode=simple"],
                                   ["--multi_gpu_mode=simple", "--calibration_tiles=2"],
                                   ["--multi_gpu_mode=staged", "--bcast_chunks=3",
                                    "--kin_threshold=0.05"]])
def test_cuking_num_gpus_over_rccl(tmp_path, one_gpu_file, extra):
    """`cuking --num_gpus=min(count, 8)` over RCCL (ncclCommInitAll, chunked
    ncclBroadcast, ncclAllGather, grouped send/recv): the one-GPU file, byte for
    byte, nothing allocated or waited for after the reservation on any rank."""
    g = min(gpu_count(), 8)
    p = run([CLI, f"--synthetic={SYNTH}", "--output_uri", tmp_path / "out", f"--num_gpus={g}",
             "--kin_threshold=0.05", "--phase_timeout_seconds=300", *extra], timeout=900)
    assert p.returncode == 0, p.stderr
    assert (tmp_path / "out" / "part-00000.snappy.parquet").read_bytes() == one_gpu_file
    s = last_json(p.stdout)
    assert s["gpus"] == g and s["collectives"] == "rccl"
    assert sum(s["rank_results"]) == s["results"] and len(s["rank_results"]) == g
    assert s["allocations_after_reserve"] == [0] * g, s
    assert s["host_syncs_after_reserve"] == [0] * g, s
    assert all(ms > 0 for ms in s["rank_kernel_ms"])


@pytest.mark.gpu
@needs_two
def test_cuking_num_gpus_failures_over_rccl(tmp_path):
    """A rank that fails -- or never comes back -- ends the job on all ranks."""
    g = min(gpu_count(), 8)
    for phase in ("setup", "compute", "gather"):
        p = run([CLI, f"--synthetic={SYNTH}", "--output_uri", tmp_path / phase, f"--num_gpus={g}",
                 f"--inject_failure=1:{phase}", "--phase_timeout_seconds=120"], timeout=600)
        assert p.returncode == 1 and f"rank 1: injected failure in phase {phase}" in p.stderr
    p = run([CLI, f"--synthetic={SYNTH}", "--output_uri", tmp_path / "hang", f"--num_gpus={g}",
             "--inject_failure=1:hang_compute", "--phase_timeout_seconds=10"], timeout=600)
    assert p.returncode == 1 and "DEADLINE_EXCEEDED" in p.stderr


@pytest.mark.gpu
@needs_two
@pytest.mark.parametrize("extra", [[], ["--split-factor=2", "--shard-index=1"]])
def test_python_driver_under_torchrun(tmp_path, extra):
    """`python -m cuking_amd.run` with one process per GPU (torch.distributed over
    RCCL): staged broadcast for the diagonal block, broadcast + tile ranges for an
    off-diagonal shard; the one-process output, byte for byte."""
    import socket
    g = min(gpu_count(), 8)
    shard = 1 if extra else 0
    base = ["-m", "cuking_amd.run", "--synthetic", SYNTH, "--kin-threshold=0.05", *extra]
    p = run([sys.executable, *base, "--output-uri", tmp_path / "one"])
    assert p.returncode == 0, p.stderr
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    p = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
             f"--nproc-per-node={g}", "--master-addr", "127.0.0.1", "--master-port", port,
             *base, "--output-uri", tmp_path / "many"], timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    name = f"part-{shard:05d}.snappy.parquet"
    assert (tmp_path / "many" / name).read_bytes() == (tmp_path / "one" / name).read_bytes()
    s = last_json(p.stdout)
    assert s["gpus"] == g
    assert s["allocations_after_reserve"] == [0] * g, s
    assert s["host_syncs_after_reserve"] == [0] * g, s


@pytest.mark.gpu
@needs_two
def test_bench_gpus_n_over_rccl():
    """`python bench.py --gpus N` as the driver starts `--gpus 1`: self-launched
    ranks over RCCL, records identical to the single-GPU pass of the same run and
    to both broadcast-inclusive passes (the bench checks and raises otherwise)."""
    g = min(gpu_count(), 8)
    p = run([sys.executable, "bench.py", "--gpus", g, *SMALL], timeout=1200)
    assert p.returncode == 0, p.stderr[-3000:]
    out = last_json(p.stdout)
    assert out["n_gpus"] == g and out["config"]["backend"] == "nccl"
    assert out["single_gpu_same_run"] is not None and out["speedup"] > 0
    assert set(out["with_broadcast"]) == {"staged", "simple"}
