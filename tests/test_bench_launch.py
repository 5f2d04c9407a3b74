"""bench.py --gpus N must work WITHOUT a wrapper: with no torch.distributed environment it launches its own N ranks
(re-executes itself under torch.distributed.run on 127.0.0.1 - the reference self-spawns too,
pointcept/engines/launch.py:35-135).  CPU test of that path: --dry-run does the launcher + process-group plumbing
(gloo here, RCCL on a GPU box: barrier, MAX / SUM all-reduce, rank-0 JSON line) and skips the model."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout  # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks():
    res = _run(["--gpus", "2", "--dry-run"])
    assert res["n_gpus"] == 2 and res["ranks_seen"] == 2
    assert res["max_over_ranks"] == 2.0  # MAX over ranks of (1 + rank)


def test_bench_single_rank_needs_no_launcher():
    res = _run(["--dry-run"])
    assert res["n_gpus"] == 1 and res["ranks_seen"] == 1


def test_bench_under_an_external_launcher_reads_the_environment():
    """The driver's form: python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                          "127.0.0.1", "--master-port", "29631", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"],
                         env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["ranks_seen"] == 2


def test_bench_shard_mode_plumbing_two_ranks():
    """--shard (BASELINE config 4's shape: 64 LiDAR sweeps over the GPUs of a node): ONE global scene list, LPT-sharded,
    counters all-reduced - the plumbing on two gloo ranks (the model itself needs a GPU)."""
    res = _run(["--gpus", "2", "--dry-run", "--shard", "64", "--dataset", "nuscenes", "--points", "40000"])
    assert res["n_gpus"] == 2 and res["shard_scenes"] == 64
    assert res["every_scene_on_exactly_one_rank"]
    a, b = res["points_per_rank"]
    assert a + b == res["points_total"] == res["points_counted"]
    assert abs(a - b) <= 0.02 * res["points_total"]  # LPT balance
