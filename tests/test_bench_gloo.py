"""The N>1 path of bench.py ("replicas only": no data-path collective, barrier + max-over-ranks timing) on CPU with
gloo, world_size 2."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_aggregation():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_gloo_worker.py")]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1]
    r = json.loads(line[len("RESULT "):])
    assert r["world"] == 2
    assert abs(r["wall"] - 0.020) < 1e-12 and abs(r["kern_ms"] - 0.10) < 1e-12        # the slowest rank's times
    ops = 2.0 * 4096 ** 3
    assert abs(r["value"] - 2 * ops * 100 / 0.020 / 1e12) < 1e-6                        # whole-job aggregate
    assert abs(r["achieved"] - ops / 0.10e-3 / 1e12) < 1e-6                             # per-launch roofline number


def test_algorithmic_bytes_match_baseline_md():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.algorithmic_bytes(4096, 4096, 4096) == 51380224
    assert bench.algorithmic_bytes(1, 4096, 4096) == 8923264           # config 2
