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


def test_committed_traffic_profile_is_of_these_kernel_sources():
    """bench.py quotes roofline.traffic only from a PMC pass taken on the kernel sources of THIS tree (the summaries under
    profiles/rNN carry the hash of atom_amd/csrc): a kernel edit without a fresh tools/profile_bench.sh pass would silently turn the
    field into null.  The headline format's traffic must be there, and above the algorithmic bytes (eight L2s fetch the operands)."""
    sys.path.insert(0, ROOT)
    import bench
    t = bench.measured_traffic(4096, 4096, 4096, "f6")
    assert t is not None, "profiles/rNN/bench_hbm_traffic_f6.json is not of kernel sources " + bench.kernel_source_sha()[:12]
    assert bench.algorithmic_bytes(4096, 4096, 4096) < t < 4 * bench.algorithmic_bytes(4096, 4096, 4096)


def test_bench_gpus_flag_starts_the_ranks():
    """`python bench.py --gpus 2` (no torch.distributed environment) must itself start 2 ranks and report n_gpus = 2; the
    launch / rendezvous / barrier / timed-region / aggregation code is the product's, only the step is a CPU stand-in."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "3",
                          "--stub-step"], capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                                   # ONE JSON line, from rank 0
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 20 and r["warmup"] == 3
    assert r["steps_run"] == 23                                          # W untimed + exactly K timed steps
    assert r["ms_per_step"] > 0 and r["value"] > 0


def test_bench_gpus_flag_must_match_world_size():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub-step"], capture_output=True,
                         text=True, timeout=120, env=env, cwd=ROOT)
    assert out.returncode != 0 and "WORLD_SIZE" in (out.stderr + out.stdout)
