"""Worker of tests/test_gpu_rccl_multirank.py: one rank = one process = one GPU, NO torch in the process.

Started by a plain launcher loop with RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT / ZK_COMM_ID_FILE in the
environment; joins the LIBRARY'S OWN RCCL communicator (csrc/comm.hip through rendezvous.comm_init_from_env) and
runs, through it and nothing else:
  * raw collectives (all-gather, all-to-all of device buffers) against their definition,
  * a sharded proving session (zk_proof_set_sharding_comm): proof bytes of every rank -> <out>/proof_<rank>.bin,
    rank 0 also writes the unsharded session's proof,
  * one NTT spread over the ranks (zk_ntt_sharded with the NULL callback = grouped ncclSend / ncclRecv): forward
    parts -> <out>/fwd_<rank>.npy, the sharded inverse of the forward result -> <out>/inv_<rank>.npy.
The test process compares the files with the single-GPU proof and the oracle's best_fft."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import zkevm_circuits_amd as z  # noqa: E402
from zkevm_circuits_amd import plonk, rendezvous, sharding  # noqa: E402
from plonk_fixtures import build_circuit  # noqa: E402


def main():
    out_dir, k, multiopen, log_n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    assert "torch" not in sys.modules
    ctx = z.Context(int(os.environ.get("LOCAL_RANK", "0")))
    rank, world = rendezvous.comm_init_from_env(ctx, timeout=300.0)

    # ---- raw collectives: rank r contributes a block filled with r + 1
    blk = 4096
    mine = np.full(blk, rank + 1, dtype=np.uint8)
    send, recv = ctx.to_device(mine), ctx.alloc(blk * world)
    ctx.comm_allgather(send, blk, recv)
    ctx.sync()
    got = recv.download((world, blk), dtype=np.uint8)
    assert all((got[r] == r + 1).all() for r in range(world)), "all-gather: wrong block order or content"
    a2a = np.concatenate([np.full(blk, 16 * (rank + 1) + p, dtype=np.uint8) for p in range(world)])      # block p goes to rank p
    send2, recv2 = ctx.to_device(a2a), ctx.alloc(blk * world)
    ctx.comm_alltoall(send2, blk, recv2)
    ctx.sync()
    got2 = recv2.download((world, blk), dtype=np.uint8)
    assert all((got2[p] == 16 * (p + 1) + rank).all() for p in range(world)), "all-to-all: block p must come from rank p"

    # ---- sharded proving session through the communicator
    circ, adv, inst = build_circuit(k, seed=5, wide=True)
    s_mont = np.frombuffer(plonk.fr_mont_bytes(0x5EC2E7), dtype=np.uint64).copy()
    srs = ctx.srs_setup_with_s(k, s_mont)
    pk = ctx.pk_create(srs, circ.blob())
    adv_m = [plonk.column_to_mont(c) for c in adv]
    inst_m = [plonk.column_to_mont(c) for c in inst]

    def prove(sharded: bool) -> bytes:
        sess = ctx.proof_session(pk, inst_m, bytes(range(16)))
        sess.set_multiopen(multiopen)
        if sharded:
            sess.set_sharding_comm()
        sess.advice_phase({i: c for i, c in enumerate(adv_m)})
        return sess.finish()

    open(os.path.join(out_dir, f"proof_{rank}.bin"), "wb").write(prove(True))
    if rank == 0:
        open(os.path.join(out_dir, "proof_single.bin"), "wb").write(prove(False))
    pk.destroy()
    srs.destroy()

    # ---- one transform spread over the ranks, the exchange inside the library (callback = NULL)
    x = np.load(os.path.join(out_dir, "input.npy"))
    m = (1 << log_n) // world
    buf = ctx.to_device(sharding.ntt_shard_input(x, rank, world))
    ctx.ntt_sharded(buf, log_n, rank, world, None)
    ctx.sync()
    fwd = buf.download((m, 4))
    np.save(os.path.join(out_dir, f"fwd_{rank}.tmp.npy"), fwd)
    os.replace(os.path.join(out_dir, f"fwd_{rank}.tmp.npy"), os.path.join(out_dir, f"fwd_{rank}.npy"))
    rendezvous.comm_barrier(ctx, rank, world)                     # every rank's forward part is on disk
    full = np.empty((1 << log_n, 4), dtype=np.uint64)
    for r in range(world):
        full[sharding.ntt_shard_output_index(log_n, r, world)] = np.load(os.path.join(out_dir, f"fwd_{r}.npy"))
    buf.upload(sharding.ntt_shard_input(full, rank, world))
    ctx.ntt_sharded(buf, log_n, rank, world, None, inverse=True)
    ctx.sync()
    np.save(os.path.join(out_dir, f"inv_{rank}.npy"), buf.download((m, 4)))
    rendezvous.comm_barrier(ctx, rank, world)
    ctx.comm_destroy()
    ctx.close()
    print(f"rank {rank}/{world} ok", flush=True)


if __name__ == "__main__":
    main()
