"""Worker of tests/test_gpu_sharded_proof.py: one rank of a sharded proving session.

Launched with torch.distributed.run; every rank builds the same circuit / witness, opens a session
on the GPU it is given (the test puts all ranks on cuda:0 -- the box has one GPU -- and exchanges
over gloo; on an 8-GPU node the same code runs with backend nccl = RCCL) and writes its proof to
<out_dir>/proof_<rank>.bin.  Rank 0 also writes the proof of an unsharded session: the two must be
byte-identical, because a sharded session produces the same transcript."""
import datetime
import os
import sys

import numpy as np
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import zkevm_circuits_amd as z  # noqa: E402
from zkevm_circuits_amd import plonk, sharding  # noqa: E402
from plonk_fixtures import build_circuit  # noqa: E402


def main():
    out_dir, k, multiopen = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    dist.init_process_group(backend=os.environ.get("ZK_TEST_BACKEND", "gloo"), timeout=datetime.timedelta(seconds=150))      # a collective that never completes is an error, not a wait
    rank, world = dist.get_rank(), dist.get_world_size()
    ctx = z.Context(int(os.environ.get("ZK_TEST_DEVICE", "0")))
    if os.environ.get("ZK_TEST_CIRCUIT") == "lookups":      # several lookup arguments, consecutive ones into ONE table (arguments are split over the ranks: round 6)
        from plonk_fixtures import build_multi_lookup_circuit
        circ, adv, inst = build_multi_lookup_circuit(k, seed=5, n_inputs=7, gate_degree=3)
    else:
        circ, adv, inst = build_circuit(k, seed=5, wide=True)
    s_mont = np.frombuffer(plonk.fr_mont_bytes(0x5EC2E7), dtype=np.uint64).copy()
    srs = ctx.srs_setup_with_s(k, s_mont)
    pk = ctx.pk_create(srs, circ.blob())
    adv_m = [plonk.column_to_mont(c) for c in adv]
    inst_m = [plonk.column_to_mont(c) for c in inst]

    def prove(sharded: bool) -> bytes:
        sess = ctx.proof_session(pk, inst_m, bytes(range(16)))
        sess.set_multiopen(multiopen)
        keep = None
        if sharded:
            keep = sharding.shard_session_device(sess) if os.environ.get("ZK_TEST_DEVGATHER") == "1" else sharding.shard_session(sess)
        sess.advice_phase({i: c for i, c in enumerate(adv_m)})
        proof = sess.finish()
        del keep
        return proof

    proof = prove(True)
    open(os.path.join(out_dir, f"proof_{rank}.bin"), "wb").write(proof)
    if rank == 0:
        open(os.path.join(out_dir, "proof_single.bin"), "wb").write(prove(False))
    dist.barrier()
    pk.destroy()
    srs.destroy()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
