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
                        "--groups", "65536", "--steps", "16", "--warmup", "4", "--no-cpu-baseline", "--consume",
                        "--strong-total", "262144"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["rccl_ranks"] == 2 and len(r["config"]["devices"]) == 2
    assert r["config"]["summary_allgather"] and r["config"]["summary_consumed_on_device"]
    assert r["config"]["summary_escapes"] == 0
    assert r["value"] > 0 and r["steps"] == 16 and r["warmup"] == 4
    assert r["verified"]["ok"] and r["verified"]["groups"] == 2 * 65536
    # one invocation separates kernel scaling from exchange cost, and carries the sharded configuration
    assert r["config"]["total_groups"] == 2 * 65536 and "weak scaling" in r["config"]["scaling_note"]
    co = r["extra"]["compute_only"]
    assert co["ms_per_step"] > 0 and co["value"] > 0 and co["kernel_ms"]["max"] >= co["kernel_ms"]["min"] > 0
    kr = r["roofline"]["kernel_ms_ranks"]
    assert kr["max"] >= kr["min"] > 0
    s8 = r["extra"]["strong_8m"]
    assert s8["total_groups"] == 262144 and s8["groups_per_gpu"] == 131072 and s8["value"] > 0 and s8["verified"]["ok"]


@pytest.mark.gpu
def test_a_rank_that_stops_making_progress_ends_the_run_with_a_diagnosis():
    """Rank 1 never reaches step 6 of the timed steps with the exchange (a wedged fence or collective
    looks like this to its peers): within the watchdog's limit a rank reports where it was and what
    its exchange was doing and exits with code 3; the launcher ends the other rank and fails.  The
    steps without the exchange had been timed by then: rank 0 leaves their figure behind in a line
    marked invalid."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["RQ_BENCH_INJECT_HANG"] = "1:6"
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
                        "--groups", "65536", "--steps", "16", "--warmup", "4", "--no-cpu-baseline", "--no-extras",
                        "--watchdog-seconds", "6"], capture_output=True, text=True, timeout=280, env=env)
    assert p.returncode != 0
    assert time.time() - t0 < 200
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["valid"] is False and "WITHOUT the summary exchange" in r["degraded"] and "main:timed" in r["degraded"]
    assert r["n_gpus"] == 2 and r["steps"] == 16 and r["value"] > 0 and 0 < r["roofline"]["frac"] < 1
    assert r["config"]["total_groups"] == 131072
    assert "watchdog: rank" in p.stderr and "made no progress" in p.stderr, p.stderr[-3000:]
    assert '"exchange"' in p.stderr and "launched_with_signal" in p.stderr or "exchange_pending" in p.stderr
    assert "rank(s) failed" in p.stderr


@pytest.mark.gpu
def test_a_stuck_additional_leg_does_not_cost_the_line():
    """Rank 1 stops inside the strong-scaling leg that follows the headline's measurement at 2 and 4
    ranks: the watchdog ends the run with a non-zero code, and rank 0's line still goes out complete
    (verified figures of the headline), with the leg's place saying what happened."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["RQ_BENCH_INJECT_HANG"] = "1:6:strong_8m"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
                        "--groups", "65536", "--strong-total", "131072", "--steps", "16", "--warmup", "4",
                        "--no-cpu-baseline", "--watchdog-seconds", "6"], capture_output=True, text=True, timeout=280, env=env)
    assert p.returncode != 0
    assert "watchdog: rank" in p.stderr and "strong_8m:timed" in p.stderr, p.stderr[-3000:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert "valid" not in r and r["verified"]["ok"] and r["n_gpus"] == 2 and r["value"] > 0
    assert r["extra"]["compute_only"]["value"] > 0
    assert "stopped making progress" in r["extra"]["strong_8m"]["error"]


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
              "append_check_commit", "replicate_commit", "i64", "i64_narrow_arena", "mixed", "mixed_term_start",
              "mixed_masks_only", "mixed_images", "rounds_in_one_launch"):
        assert k in r["extra"], k
        assert r["extra"][k].get("kernel_ms", r["extra"][k].get("ms_per_round", 0)) > 0, k


def _bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_supervise_ends_every_rank_when_one_fails(tmp_path):
    """The launcher's supervisor on stand-in ranks: rank 1 exits with 3 while rank 0 would run
    for minutes -- the run fails at once and no child is left behind; a run whose ranks all
    succeed relays rank 0's stdout; a run past its deadline is ended."""
    import io
    import time
    bench = _bench()
    py = sys.executable

    def start(code, pipe=False):
        return subprocess.Popen([py, "-c", code], stdout=subprocess.PIPE if pipe else None, start_new_session=True)

    t0 = time.time()
    procs = [start("import time; time.sleep(300)", pipe=True), start("import sys, time; time.sleep(0.5); sys.exit(3)")]
    err = io.StringIO()
    rc, out = bench.supervise(procs, 120, out=err)
    assert rc == 1 and time.time() - t0 < 30 and "rank(s) failed: [(1, 3)]" in err.getvalue()
    assert all(p.poll() is not None for p in procs)

    procs = [start("print('the line')", pipe=True), start("pass")]
    rc, out = bench.supervise(procs, 60)
    assert rc == 0 and out.decode().strip() == "the line"

    t0 = time.time()
    procs = [start("import time; time.sleep(300)", pipe=True), start("import time; time.sleep(300)")]
    err = io.StringIO()
    rc, out = bench.supervise(procs, 1.0, out=err)
    assert rc == 1 and time.time() - t0 < 30 and "no result after" in err.getvalue()
    assert all(p.poll() is not None for p in procs)


def test_watchdog_reports_and_exits_only_while_armed():
    import io
    import time
    bench = _bench()
    out, codes = io.StringIO(), []
    dog = bench.Watchdog(0.4, describe=lambda: {"step": 7, "exchange": {"filled": [True, False]}}, out=out, exit_fn=codes.append)
    time.sleep(1.0)
    assert not codes                       # not armed: setup and the CPU legs may take as long as they like
    dog.arm("preheat")
    for _ in range(10):                    # beats keep it quiet
        time.sleep(0.1)
        dog.beat("step")
    assert not codes
    time.sleep(1.5)                        # no beat: it fires once, with the state and the stacks
    assert codes == [3]
    text = out.getvalue()
    assert "made no progress" in text and '"step": 7' in text and "filled" in text and "Thread" in text
    dog2 = bench.Watchdog(0.3, out=io.StringIO(), exit_fn=codes.append)
    dog2.arm("x"); dog2.disarm()
    time.sleep(1.0)
    assert codes == [3]
