"""CPU rehearsal of the multi-GPU path: world_size 2 over gloo."""
import os
import socket
import subprocess
import sys

from raft_amd.sharded import shard_bounds, summary_bytes

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shard_bounds_cover_everything_once():
    for total in (0, 1, 7, 8, 1000, (1 << 23) + 5):
        for world in (1, 2, 3, 4, 8):
            cuts = [shard_bounds(total, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == total
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_summary_layout_keeps_sections_aligned():
    for mode, width in (("u16", 2), ("i32", 4)):
        for g in (1, 2, 3, 63, 64, 65, 1037, 1 << 20):
            n = summary_bytes(g, mode)
            commit = (g * width + 7) // 8 * 8
            assert n == commit + 8 * ((g + 63) // 64) and n % 8 == 0


def test_two_ranks_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_gloo_worker.py")],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "GLOO_SHARDED_OK" in outs[0]
