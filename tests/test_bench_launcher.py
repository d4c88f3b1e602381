"""`python bench.py --gpus N` must run N ranks: the launcher, the rendezvous and the
one-JSON-line contract are driven here on CPU ranks (gloo) through bench.py's own entry point."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, cwd=ROOT, env=e,
                          capture_output=True, text=True, timeout=300)


def test_gpus_2_spawns_two_ranks_and_prints_one_line():
    r = _run(["--gpus", "2", "--steps", "5", "--warmup", "1", "--batches", "2",
              "--workload", "tests.bench_cpu_workload:CpuWorkload"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 5 and line["warmup"] == 1
    assert line["scaling"] == "weak" and line["config"]["parallelism"] == "dp2"
    # whole-job aggregate: 2 ranks x 2 graphs x 5 steps over the (max-over-ranks) time
    assert abs(line["value"] - 2 * 2 * 5 / (line["ms_per_step"] * 5e-3)) <= 1e-2 * line["value"]
    assert line["grad_checksum"] > 0


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "tests.bench_cpu_workload:CpuWorkload"],
             env={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_failing_rank_fails_the_launch():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "tests.no_such_module:X"])
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
