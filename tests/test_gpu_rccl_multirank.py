"""GPU: the library's own RCCL communicator (csrc/comm.hip) with ONE RANK PER GPU -- raw collectives, the sharded
proving session (zk_proof_set_sharding_comm: commitments all-gathered per transcript round, advice columns and
quotient cosets device to device) and the sharded NTT with the in-library all-to-all (zk_ntt_sharded, NULL callback).

world = 1 always runs (the whole worker through librccl with a one-rank communicator: what a 1-GPU box can execute);
world = 2 / 4 / 8 run whenever the box has that many GPUs -- on the 8-GPU node `pytest -m gpu` exercises comm.hip
over xGMI without edits, and requires byte-identical proofs and a bit-exact transform.  The gloo-based tests
(test_gpu_sharded_proof.py, test_gpu_sharded_ntt.py: ranks sharing one GPU) stay as the 1-GPU fallback for the
session logic; RCCL itself cannot put two ranks on one device."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import bn254

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _gpus() -> int:
    import torch
    return torch.cuda.device_count()


def launch_ranks(world, argv, out_dir, timeout=300):
    """A launcher that is a plain loop: what the rendezvous documents as sufficient (no torch.distributed.run)."""
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   ZK_COMM_ID_FILE=os.path.join(str(out_dir), "comm_id"), OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable] + argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()                     # the exact processes started above
    return [p.returncode for p in procs], outs


@pytest.mark.parametrize("world,k,multiopen,log_n", [(1, 7, 1, 10), (2, 7, 1, 11), (2, 11, 0, 20), (4, 8, 1, 14), (8, 8, 0, 16), (8, 12, 1, 20)])
def test_one_rank_per_gpu_through_the_library_communicator(tmp_path, cref, world, k, multiopen, log_n):
    have = _gpus()
    if have < world:
        pytest.skip(f"{world} ranks need {world} GPUs (RCCL cannot share a device); this box has {have}")
    from zkevm_circuits_amd import sharding
    n = 1 << log_n
    x = cref.rand_fr_stream(777 + log_n, n)
    np.save(tmp_path / "input.npy", x)
    codes, outs = launch_ranks(world, [os.path.join(HERE, "_rccl_worker.py"), str(tmp_path), str(k), str(multiopen), str(log_n)], tmp_path)
    assert codes == [0] * world, "\n".join(o[-3000:] for o in outs)
    single = open(tmp_path / "proof_single.bin", "rb").read()
    assert len(single) > 500
    for r in range(world):
        assert open(tmp_path / f"proof_{r}.bin", "rb").read() == single, f"rank {r}: sharded proof differs from the single-GPU proof"
    want = cref.best_fft(x, bn254.omega_for_k(log_n), log_n)
    got, back = np.empty_like(want), np.empty_like(x)
    for r in range(world):
        idx = sharding.ntt_shard_output_index(log_n, r, world)
        got[idx] = np.load(tmp_path / f"fwd_{r}.npy")
        back[idx] = np.load(tmp_path / f"inv_{r}.npy")
    assert np.array_equal(got, want), "sharded NTT (in-library all-to-all) differs from best_fft"
    assert np.array_equal(back, x)
    assert not os.path.exists(tmp_path / "comm_id") or world == 1, "the id file must be retired once every rank has joined"
