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


@pytest.mark.parametrize("world,k,multiopen,devgather,knobs", [(2, 7, 0, 0, {}), (3, 8, 1, 0, {}), (2, 7, 1, 1, {}), (3, 7, 0, 1, {}), (2, 11, 1, 0, {}), (3, 12, 0, 1, {}),
                                                                (3, 8, 1, 1, {"ZK_QUOTIENT_COSTGATE": "0"}), (2, 8, 1, 0, {"ZK_QUOTIENT_DAG": "0"}),
                                                                (3, 8, 1, 1, {"ZK_SHARD_EXCHANGE_GROUPS": "1"}), (2, 8, 0, 1, {"ZK_SHARD_EXCHANGE_GROUPS": "3"}),
                                                                (2, 8, 1, 1, {"ZK_TEST_CIRCUIT": "lookups"}), (3, 8, 0, 0, {"ZK_TEST_CIRCUIT": "lookups"}), (3, 9, 1, 1, {"ZK_TEST_CIRCUIT": "lookups", "ZK_SHARD_EXCHANGE_GROUPS": "1"}),
                                                                (2, 8, 1, 1, {"ZK_TEST_CIRCUIT": "lookups", "ZK_SHARD_LOOKUPS": "0"}), (3, 8, 1, 1, {"ZK_TEST_CIRCUIT": "lookups", "ZK_SHARD_COEFF": "0"}), (2, 7, 0, 1, {"ZK_SHARD_COEFF": "0"})])
def test_sharded_session_matches_single(tmp_path, world, k, multiopen, devgather, knobs):
    """knobs: ZK_QUOTIENT_COSTGATE=0 forces the additive split (remainder polynomials travelling between degree classes, (class, coset)
    pairs of three classes dealt over the ranks); ZK_QUOTIENT_DAG=0 the class programs as round 5 assembled them; ZK_SHARD_EXCHANGE_GROUPS: how many
    groups of `world` advice columns one device all-gather moves (default 16: here every column of a phase in one exchange; 1: a collective per group; 3: a ragged last chunk);
    ZK_TEST_CIRCUIT=lookups: five lookup arguments, consecutive ones into one table -- the arguments are split over the ranks (each rank compresses, counts and sums its own,
    m and phi all-gathered device to device or through the host), ZK_SHARD_LOOKUPS=0 keeps them replicated; in the device-gather cases the owner ships the COEFFICIENT form
    of every advice column that no lookup, permutation or remainder program reads (no inverse transform on the peers), ZK_SHARD_COEFF=0 ships Lagrange values for all"""
    from _launch import run_ranks
    # devgather: the advice columns are uploaded by their owning rank only and all-gathered between devices
    env = dict(os.environ, ZK_TEST_BACKEND="gloo", OMP_NUM_THREADS="1", ZK_TEST_DEVGATHER=str(devgather), **knobs)
    res = run_ranks(world, os.path.join(HERE, "_sharded_proof_worker.py"), [tmp_path, k, multiopen], env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    single = open(tmp_path / "proof_single.bin", "rb").read()
    assert len(single) > 500
    for r in range(world):
        assert open(tmp_path / f"proof_{r}.bin", "rb").read() == single, f"rank {r} produced a different proof"


@pytest.mark.parametrize("world,rank", [(2, 0), (2, 1), (4, 3), (8, 0)])
def test_emulated_rank_runs_its_share_on_one_gpu(world, rank):
    """sharding.EmulatedRank (what `bench.py` times for `extra.projected_rank_device_s`): rank r of N on ONE GPU with its peers'
    columns served from the resident witness.  The session must take the owner-only hand-over (NULL for the columns of other ranks),
    run every stage, and return a proof of the unsharded proof's length; with the peers' commitments and quotient pairs replaced by
    stand-ins its bytes differ from the real proof (world > 1) -- it is a timing vehicle, and says so."""
    import numpy as np
    import zkevm_circuits_amd as z
    from zkevm_circuits_amd import plonk, sharding
    from plonk_fixtures import build_circuit
    ctx = z.Context(0)
    circ, adv, inst = build_circuit(9, seed=6, wide=True)
    srs = ctx.srs_setup_with_s(9, np.frombuffer(plonk.fr_mont_bytes(0x5EC2E7), dtype=np.uint64).copy())
    pk = ctx.pk_create(srs, circ.blob())
    inst_m = [plonk.column_to_mont(c) for c in inst]
    adv_dev = {i: ctx.to_device(plonk.column_to_mont(c)) for i, c in enumerate(adv)}
    try:
        sess = ctx.proof_session(pk, inst_m, bytes(range(16)))
        sess.set_multiopen(1)
        sess.advice_phase_dev(dict(adv_dev))
        real = sess.finish()
        owned = set(range(circ.A)[rank::world])
        emu = sharding.EmulatedRank(ctx, circ, adv_dev, rank, world)
        sess = ctx.proof_session(pk, inst_m, bytes(range(16)))
        sess.set_multiopen(1)
        emu.attach(sess)
        emu.begin_phase(0)
        sess.advice_phase_dev({i: (adv_dev[i] if i in owned else None) for i in range(circ.A)})
        got = sess.finish()
        assert len(got) == len(real) and got != real
        assert emu.phase_cols == []                        # every group of the phase went through the device gather
    finally:
        for b_ in adv_dev.values():
            b_.free()
        pk.destroy()
        srs.destroy()
        ctx.close()
