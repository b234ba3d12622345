"""GPU: a proving session sharded over ranks (commitments and quotient cosets split
over the ranks, 64-byte points and finished cosets all-gathered) yields, on every rank, exactly the
bytes of the unsharded session.  The box has one GPU, so the ranks share cuda:0 and exchange over
gloo; the code path is the one an 8-GPU node runs with backend nccl (RCCL).  Cases: commitments
split by column, advice columns all-gathered between devices (devgather), and -- at k >= 11, where a
batch has fewer columns than ranks -- single MSMs split by points with host-summed partial results."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


@pytest.mark.parametrize("world,k,multiopen,devgather", [(2, 7, 0, 0), (3, 8, 1, 0), (2, 7, 1, 1), (3, 7, 0, 1), (2, 11, 1, 0), (3, 12, 0, 1)])
def test_sharded_session_matches_single(tmp_path, world, k, multiopen, devgather):
    from _launch import run_ranks
    # devgather: the advice columns are uploaded by their owning rank only and all-gathered between devices
    env = dict(os.environ, ZK_TEST_BACKEND="gloo", OMP_NUM_THREADS="1", ZK_TEST_DEVGATHER=str(devgather))
    res = run_ranks(world, os.path.join(HERE, "_sharded_proof_worker.py"), [tmp_path, k, multiopen], env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    single = open(tmp_path / "proof_single.bin", "rb").read()
    assert len(single) > 500
    for r in range(world):
        assert open(tmp_path / f"proof_{r}.bin", "rb").read() == single, f"rank {r} produced a different proof"
