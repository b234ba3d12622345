"""CPU, world_size 2, gloo: the N > 1 path of the prover.  The per-rank MSMs are
stood in by the oracle (no GPU here); what is under test is the sharding + exchange + host
combine code that runs unchanged over RCCL on the GPU box."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import bn254 as b, cref
    from zkevm_circuits_amd import sharding
    try:
        # --- column sharding: 5 columns over 2 ranks, commitments gathered in column order
        n, ncols = 64, 5
        bases = cref.srs_powers(77, n)
        cols = [cref.rand_fr_stream(100 + c, n) for c in range(ncols)]
        mine = {c: cref.best_multiexp(cols[c], bases, 1) for c in sharding.columns_of_rank(ncols, rank, world)}
        allc = sharding.all_gather_commitments(mine, ncols)
        want = np.stack([cref.best_multiexp(cols[c], bases, 1) for c in range(ncols)])
        assert np.array_equal(allc, want)
        # --- point sharding of one MSM: partial results all-gathered as bytes and added on the host
        n = 101
        bases, sc = cref.srs_powers(5, n), cref.rand_fr_stream(9, n)
        sl = sharding.point_slice(n, rank, world)
        partial = cref.best_multiexp(sc[sl], bases[sl], 1)
        total = sharding.all_reduce_g1(partial)
        assert np.array_equal(total, cref.best_multiexp(sc, bases, 1))
        # slices tile [0, n)
        sizes = [sharding.point_slice(n, r, world) for r in range(world)]
        assert sizes[0].start == 0 and sizes[-1].stop == n and all(a.stop == b_.start for a, b_ in zip(sizes, sizes[1:]))
        # --- the all-gather callback a sharded proving session calls from C (zk_proof_set_sharding)
        import ctypes
        cb = sharding.make_allgather()
        nbytes = 96
        send = (ctypes.c_uint8 * nbytes)(*[(7 * rank + i) & 0xFF for i in range(nbytes)])
        recv = (ctypes.c_uint8 * (nbytes * world))()
        assert cb(None, ctypes.addressof(send), nbytes, ctypes.addressof(recv)) == 0
        for r in range(world):
            assert list(recv[r * nbytes:(r + 1) * nbytes]) == [(7 * r + i) & 0xFF for i in range(nbytes)]
        # --- one NTT over the ranks (zk_ntt_sharded's 4-step split), the device steps stood in by the oracle:
        # local transform of the residue class, twiddle, all-to-all, world-point transform across the rows
        log_n = 6
        n, m = 1 << log_n, (1 << log_n) // world
        cols_ = m // world
        x = cref.rand_fr_stream(31, n)
        w_n = b.omega_for_k(log_n)
        local = cref.from_mont(cref.best_fft(sharding.ntt_shard_input(x, rank, world), b.omega_for_k(log_n - 1), log_n - 1))
        local = [int(v) * pow(w_n, rank * j2, b.R_MOD) % b.R_MOD for j2, v in enumerate(local)]
        rows = [None] * world
        dist.all_gather_object(rows, local)
        recv = [rows[p][rank * cols_:(rank + 1) * cols_] for p in range(world)]          # block p comes from rank p
        w_w = pow(w_n, m, b.R_MOD)
        out = [sum(recv[i1][c] * pow(w_w, i1 * j1, b.R_MOD) for i1 in range(world)) % b.R_MOD for j1 in range(world) for c in range(cols_)]
        full = cref.from_mont(cref.best_fft(x, w_n, log_n))
        idx = sharding.ntt_shard_output_index(log_n, rank, world)
        assert out == [int(full[i]) for i in idx]
        every = np.sort(np.concatenate([sharding.ntt_shard_output_index(log_n, r, world) for r in range(world)]))
        assert np.array_equal(every, np.arange(n))
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_two_rank_column_and_point_sharding():
    from oracle import cref
    cref.build()
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
