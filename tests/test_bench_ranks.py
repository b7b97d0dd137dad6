"""bench.py's own launcher: --gpus N without a torchrun environment must start N
rank processes and report n_gpus = N (rehearsed with the gloo backend on the one
GPU of the test box; RCCL needs one GPU per rank)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_gpus_2_starts_two_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
                        "--groups", "65536", "--steps", "16", "--warmup", "4", "--no-cpu-baseline", "--consume"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["rccl_ranks"] == 2 and len(r["config"]["devices"]) == 2
    assert r["config"]["summary_allgather"] and r["config"]["summary_consumed_on_device"]
    assert r["config"]["summary_escapes"] == 0
    assert r["value"] > 0 and r["steps"] == 16 and r["warmup"] == 4


@pytest.mark.gpu
def test_bench_refuses_more_rccl_ranks_than_gpus():
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--groups", "4096",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0 and not p.stdout.strip()          # never a line with n_gpus != N


def test_launcher_never_imports_torch_in_the_parent():
    """The parent of a self-launched multi-rank run must not initialise the GPU:
    everything before launch_ranks() is stdlib-only."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[: src.index("def main():")]
    assert "import torch" not in head
    body = src[src.index("def main():"):]
    assert body.index("launch_ranks(args)") < body.index("import torch")


@pytest.mark.gpu
def test_bench_single_rank_line_carries_every_secondary_figure():
    """The default (one-rank) line at a small group count: one JSON line with the roofline
    object and every secondary figure under `extra`, none of them failed."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--groups", "65536", "--steps", "8", "--warmup", "2",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=400, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 1 and r["steps"] == 8 and r["value"] > 0 and r["dtype"] == "int32"
    assert r["roofline"]["bound"] == "hbm" and 0 < r["roofline"]["frac"] < 1 and r["roofline"]["kernel_ms"] > 0
    assert "error" not in r["extra"], r["extra"].get("error")
    for k in ("arena_i32", "three_streams", "term_start", "planes", "commit_only", "vote_only", "append_check",
              "append_check_commit", "replicate_commit", "i64", "i64_narrow_arena", "mixed", "mixed_term_start"):
        assert k in r["extra"], k
        assert r["extra"][k].get("kernel_ms", r["extra"][k].get("ms_per_round", 0)) > 0, k
