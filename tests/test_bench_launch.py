"""bench.py's own multi-rank launch path on CPU (gloo, stub step): `python bench.py --gpus 2` without a launcher must start two
ranks through torch.distributed.run, report n_gpus from the process group, time with barrier + max over ranks and print ONE JSON line;
a process group smaller than --gpus must fail loudly (VERDICT r01 missing #1: it used to run one rank and print n_gpus 1)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=600):
    env = dict(os.environ, GSL_BENCH_STUB="1", **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        if k not in env_extra:
            env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_spawns_its_own_ranks():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "8"], {})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                       # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["parallelism"] == "dp2" and out["config"]["global_batch"] == 2 * 2 * 8
    assert out["value"] > 0 and out["ms_per_step"] > 0 and out["higher_is_better"] is True


def test_bench_strong_scaling_splits_the_global_batch():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8", "--scaling", "strong"], {})
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["scaling"] == "strong" and out["config"]["global_batch"] == 2 * 8 and out["n_gpus"] == 2


def test_bench_fails_when_the_group_is_smaller_than_requested():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "process group has 1 rank" in (r.stderr + r.stdout)
