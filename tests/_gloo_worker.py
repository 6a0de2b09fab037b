"""Worker for tests/test_bench_gloo.py: exercises bench.py's multi-rank aggregation over a gloo process group."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
# rank r pretends to be slower the higher its rank: the job time is the slowest rank's
wall, kern_ms = 0.010 * (1 + rank), 0.05 * (1 + rank)
dist.barrier()
wall_m, kern_m = bench.max_over_ranks([wall, kern_ms], dist, torch.device("cpu"))
agg = bench.aggregate(world, 100, wall_m, kern_m, 4096, 4096, 4096)
if rank == 0:
    print("RESULT " + json.dumps({"world": world, "wall": wall_m, "kern_ms": kern_m, **agg}), flush=True)
dist.barrier()
dist.destroy_process_group()
