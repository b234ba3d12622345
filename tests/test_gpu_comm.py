"""GPU: the library's own RCCL collectives (csrc/comm.hip) executed on hardware.  The test box has
one GPU, so the communicator has one rank: every call goes through librccl (unique id,
ncclCommInitRank, ncclAllGather, grouped ncclSend / ncclRecv) and must behave as the identity
exchange; the multi-rank data paths are covered by the gloo tests (tests/test_sharding_gloo.py,
test_gpu_sharded_proof.py), which drive the same session code through the callback interface."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from plonk_fixtures import build_circuit  # noqa: E402
from zkevm_circuits_amd import plonk  # noqa: E402

pytestmark = pytest.mark.gpu


def test_single_rank_communicator(zk, ctx, cref):
    uid = ctx.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    ctx.comm_init(uid, 0, 1)
    try:
        a = cref.rand_fr_stream(9, 1 << 12)
        src, dst = ctx.to_device(a), ctx.alloc(a.nbytes)
        ctx.comm_allgather(src, a.nbytes, dst)
        ctx.sync()
        assert np.array_equal(dst.download(a.shape), a)
        dst2 = ctx.alloc(a.nbytes)
        ctx.comm_alltoall(src, a.nbytes, dst2)
        ctx.sync()
        assert np.array_equal(dst2.download(a.shape), a)
        # a session sharded over the (one-rank) communicator produces the plain session's proof
        circ, adv, inst = build_circuit(6, seed=21, wide=False)
        srs = ctx.srs_setup_with_s(circ.k, cref.fr_const(0x5EC2E7))
        pk = ctx.pk_create(srs, circ.blob())
        proofs = []
        for with_comm in (False, True):
            sess = ctx.proof_session(pk, [plonk.column_to_mont(c) for c in inst], bytes(16))
            if with_comm:
                sess.set_sharding_comm()
            sess.set_multiopen(1)
            sess.advice_phase({i: plonk.column_to_mont(c) for i, c in enumerate(adv)})
            proofs.append(sess.finish())
        assert proofs[0] == proofs[1] and len(proofs[0]) > 500
        pk.destroy()
        srs.destroy()
    finally:
        ctx.comm_destroy()
    with pytest.raises(zk.ZkError):            # no communicator any more
        ctx.comm_allgather(src, a.nbytes, dst)
    # a rank outside the world is refused before RCCL is touched
    with pytest.raises(zk.ZkError):
        ctx.comm_init(uid, 3, 2)


def test_communicator_joined_from_the_launcher_environment(zk, ctx, monkeypatch, tmp_path):
    """rendezvous.comm_init_from_env: what `bench_proof.py --rccl` does under a launcher, here with the one rank the box has"""
    from zkevm_circuits_amd import rendezvous

    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("ZK_COMM_ID_FILE", str(tmp_path / "uid"))
    rank, world = rendezvous.comm_init_from_env(ctx)
    try:
        assert (rank, world) == (0, 1)
        rendezvous.comm_barrier(ctx, rank, world)
    finally:
        ctx.comm_destroy()
