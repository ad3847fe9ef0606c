"""`python bench.py --gpus N` must launch its own N ranks (the driver's command form has no torch.distributed.run around it) and rank 0
must print exactly one JSON line with the contract's fields.  Runs the real bench.py end to end on the host: Mantis-tiny, 2 ranks over
gloo, the oracle's CPU operators standing in for the HIP backend (tests/bench_cpu_harness.py)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "tests", "bench_cpu_harness.py")


def _run(args, timeout=900):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["OMP_NUM_THREADS"] = "2"
    r = subprocess.run([sys.executable, HARNESS, *args], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"expected ONE JSON line from rank 0, got {len(lines)}:\n{r.stdout[-2000:]}"
    return json.loads(lines[0])


def test_bench_gpus2_self_launches_and_prints_one_line():
    out = _run(["--gpus", "2", "--steps", "1", "--warmup", "1", "--config", "mantis_tiny", "--no-cpu-baseline"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config"):
        assert k in out, k
    assert out["n_gpus"] == 2 and out["steps"] == 1 and out["warmup"] == 1
    assert out["scaling"] == "weak" and out["higher_is_better"] is True and out["vs_baseline"] is None
    assert out["config"]["global_batch"] == 4 and out["config"]["parallelism"] == "dp2"
    assert out["value"] > 0 and abs(out["value"] - 4 / (out["ms_per_step"] * 1e-3)) < 1e-2 * out["value"]
    # the gradient exchange ran: every bucket of the arena once per step, on both ranks (GradReducer over gloo)
    assert out["dp"] is not None and out["dp"]["buckets_per_step"] >= 4 and out["dp"]["bytes_per_step"] > 0
    assert "HOST-ONLY" in out["data"]
    assert 5.0 < out["loss"] < 15.0          # ~ ln V for random-init weights


def test_bench_gpus1_host_plumbing():
    out = _run(["--gpus", "1", "--steps", "1", "--warmup", "0", "--config", "mantis_tiny", "--no-cpu-baseline", "--no-optimizer"])
    assert out["n_gpus"] == 1 and out["dp"] is None and out["config"]["global_batch"] == 2
