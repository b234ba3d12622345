"""GPU: ONE NTT spread over several ranks (SURVEY 8e "domain halves": local transforms of the
residue classes, one all-to-all, world-point butterflies) equals best_fft of the C oracle bit for
bit, and the sharded inverse brings the input back.  The box has one GPU: the ranks share cuda:0
and exchange over gloo; an 8-GPU node runs the same code with backend nccl (RCCL)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import bn254

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


@pytest.mark.parametrize("world,log_n", [(2, 2), (2, 11), (4, 4), (4, 14), (2, 20), (8, 16)])
def test_sharded_ntt_matches_best_fft(tmp_path, cref, world, log_n):
    from zkevm_circuits_amd import sharding
    n = 1 << log_n
    x = cref.rand_fr_stream(4242 + log_n, n)
    np.save(tmp_path / "input.npy", x)
    from _launch import run_ranks
    env = dict(os.environ, ZK_TEST_BACKEND="gloo", OMP_NUM_THREADS="1")
    res = run_ranks(world, os.path.join(HERE, "_sharded_ntt_worker.py"), [tmp_path, log_n], env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    want = cref.best_fft(x, bn254.omega_for_k(log_n), log_n)
    got = np.empty_like(want)
    back = np.empty_like(x)
    for r in range(world):
        idx = sharding.ntt_shard_output_index(log_n, r, world)
        got[idx] = np.load(tmp_path / f"fwd_{r}.npy")
        back[idx] = np.load(tmp_path / f"inv_{r}.npy")
    assert np.array_equal(got, want)
    assert np.array_equal(back, x)
