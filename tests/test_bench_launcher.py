"""`python bench.py --gpus N` must launch its own ranks (the driver runs it plainly) and rank 0 must print ONE JSON
line with the driver's keys.  Here on CPU: world 2 over gloo, the per-rank engine supplied by the oracle
(tests/bench_engine.py) through bench.py's test hook; on the GPU box tests/test_gpu_chains.py runs the real
`--gpus 1 --force-lockstep` line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config"}


def _run(extra, world):
    env = dict(os.environ, HTM_BENCH_TEST_ENGINE="tests.bench_engine:make", PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1",
           "--iters-per-step", "10", "--events", "40", "--stations", "8", "--chains", "2", "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("world", [1, 2])
def test_bench_launches_its_own_ranks_and_prints_one_json_line(world):
    out = _run([], world)
    assert KEYS <= set(out)
    assert out["n_gpus"] == world and out["steps"] == 3 and out["warmup"] == 1
    assert out["config"]["iterations_per_step"] == 10 and out["config"]["iterations_timed"] == 30
    assert out["value"] > 0 and out["scaling"] == "weak"
    # value = all ranks' proposal steps / the max-over-ranks time of the region
    assert out["value"] == pytest.approx(world * 2 * 30 / (out["ms_per_step"] * 3e-3), rel=1e-9)
    assert "test double" in out["engine"]


@pytest.mark.gpu
def test_bench_two_ranks_with_the_real_engine_sharing_the_gpu():
    """`python bench.py --gpus 2` end to end with the HIP engine: the launcher starts two ranks, both on device 0
    (HTM_BENCH_ONE_GPU=1: a rehearsal, RCCL refuses two ranks on one device so the rendezvous is gloo), the ranks map each
    other's inboxes over IPC and run the persistent lock-step loop; rank 0 prints the one JSON line."""
    env = dict(os.environ, PYTHONPATH=ROOT, HTM_BENCH_ONE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HTM_BENCH_TEST_ENGINE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--iters-per-step", "512", "--events", "200", "--stations", "32", "--chains", "4", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert KEYS <= set(out) and out["n_gpus"] == 2 and out["value"] > 0
    assert out["config"]["iterations_timed"] == 1024
    assert "persistent lock-step" in out["config"]["swap_transport"]
    assert "rehearsal" in out["data"] and "engine" not in out
    assert out["roofline"]["full_evals"] > 0 and out["roofline"]["partial_evals"] > 0
