"""Worker of tests/test_gpu_sharded_ntt.py: one rank of ONE transform spread over the ranks
(zk_ntt_sharded).  Every rank derives the same input from a seed, keeps its residue class, runs the
sharded transform (forward, then inverse on the result) and writes what it holds to
<out_dir>/fwd_<rank>.npy and inv_<rank>.npy.  The ranks share cuda:0 and exchange over gloo on the
one-GPU test box; with backend nccl the same code runs the all-to-all over RCCL / xGMI."""
import datetime
import os
import sys

import numpy as np
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import zkevm_circuits_amd as z  # noqa: E402
from zkevm_circuits_amd import sharding  # noqa: E402


def main():
    out_dir, log_n = sys.argv[1], int(sys.argv[2])
    dist.init_process_group(backend=os.environ.get("ZK_TEST_BACKEND", "gloo"), timeout=datetime.timedelta(seconds=150))      # a collective that never completes is an error, not a wait
    rank, world = dist.get_rank(), dist.get_world_size()
    ctx = z.Context(int(os.environ.get("ZK_TEST_DEVICE", "0")))
    x = np.load(os.path.join(out_dir, "input.npy"))
    cb = sharding.make_alltoall_dev()
    buf = ctx.to_device(sharding.ntt_shard_input(x, rank, world))
    m = (1 << log_n) // world
    ctx.ntt_sharded(buf, log_n, rank, world, cb)
    fwd = buf.download((m, 4))
    np.save(os.path.join(out_dir, f"fwd_{rank}.npy"), fwd)
    # inverse: the input layout is again a residue class, so re-shard the forward result through the host
    full = np.empty((1 << log_n, 4), dtype=np.uint64)
    parts = [None] * world
    dist.all_gather_object(parts, fwd)
    for r in range(world):
        full[sharding.ntt_shard_output_index(log_n, r, world)] = parts[r]
    buf.upload(sharding.ntt_shard_input(full, rank, world))
    ctx.ntt_sharded(buf, log_n, rank, world, cb, inverse=True)
    np.save(os.path.join(out_dir, f"inv_{rank}.npy"), buf.download((m, 4)))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
