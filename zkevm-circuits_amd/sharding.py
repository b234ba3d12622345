"""Multi-GPU layer of the prover hot path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on MI355X nodes; "gloo" in the CPU tests).

The path shards by independent units:
  * column sharding (production mode): a proof commits ~10^3 columns; rank r owns columns
    {c : c % world == r}, runs MSM / NTT on them with no data-path collective, and the only
    exchange is one all-gather of the 64-byte commitments per transcript round;
  * point sharding of ONE MSM (few-column circuits, BASELINE config 5): rank r computes the MSM of
    its contiguous slice of the points, the 64-byte partial results are all-gathered as raw bytes
    (RCCL reduce ops are numeric: no elliptic-curve sum) and every rank adds them on the host.
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Sequence

import numpy as np


class _LazyModule:
    """torch is plumbing for the callback-based exchanges only; the pure index helpers below (ntt_shard_input ...) and
    the in-library RCCL path must work in a prover rank that never loads torch (its bundled HIP runtime next to the
    library's slows the witness uploads, tools/upload_order.py) -- so it is imported on first use, not at import."""

    def __init__(self, name):
        self._name, self._mod = name, None

    def __getattr__(self, attr):
        if self._mod is None:
            import importlib
            self._mod = importlib.import_module(self._name)
        return getattr(self._mod, attr)


torch = _LazyModule("torch")
dist = _LazyModule("torch.distributed")

from . import binding

G1_BYTES = 64


def columns_of_rank(num_columns: int, rank: int, world: int) -> List[int]:
    """Round-robin column ownership."""
    return list(range(rank, num_columns, world))


def point_slice(n: int, rank: int, world: int) -> slice:
    """Contiguous slice of an n-point MSM owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return slice(lo, lo + base + (1 if rank < rem else 0))


def g1_sum_host(points: np.ndarray) -> np.ndarray:
    """Sum of affine points (n, 8) u64 on the host via the C ABI (zk_g1_sum_host)."""
    pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 8)
    out = np.empty(8, dtype=np.uint64)
    rc = binding.lib().zk_g1_sum_host(pts.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(pts.shape[0]), out.ctypes.data_as(ctypes.c_void_p))
    if rc != 0:
        raise binding.ZkError(f"zk_g1_sum_host failed: {rc}")
    return out


def _device_for_backend(group=None) -> torch.device:
    backend = dist.get_backend(group)
    return torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")


def all_gather_commitments(local: Dict[int, np.ndarray], num_columns: int, group=None) -> np.ndarray:
    """Every rank contributes the commitments of the columns it owns; returns (num_columns, 8) u64
    in column order on every rank.  One all_gather of world * ceil(C/world) * 64 bytes."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = (num_columns + world - 1) // world
    dev = _device_for_backend(group)
    send = np.zeros((per, 8), dtype=np.uint64)
    for slot, col in enumerate(columns_of_rank(num_columns, rank, world)):
        send[slot] = local[col]
    t = torch.from_numpy(send.view(np.uint8).reshape(-1)).to(dev)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    res = np.zeros((num_columns, 8), dtype=np.uint64)
    for r in range(world):
        got = out[r].cpu().numpy().view(np.uint64).reshape(per, 8)
        for slot, col in enumerate(columns_of_rank(num_columns, r, world)):
            res[col] = got[slot]
    return res


def all_reduce_g1(partial: np.ndarray, group=None) -> np.ndarray:
    """'All-reduce' of one G1 point: all_gather the raw 64 bytes, add on the host."""
    world = dist.get_world_size(group)
    dev = _device_for_backend(group)
    t = torch.from_numpy(np.ascontiguousarray(partial, dtype=np.uint64).view(np.uint8).reshape(-1)).to(dev)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    pts = np.stack([o.cpu().numpy().view(np.uint64) for o in out])
    return g1_sum_host(pts)


def msm_point_sharded(ctx: "binding.Context", d_scalars_ptr: int, d_bases_ptr: int, n_local: int, group=None) -> np.ndarray:
    """One MSM split over the ranks: the pointers address THIS rank's slice (see point_slice)."""
    partial = ctx.msm(d_scalars_ptr, d_bases_ptr, n_local)
    return all_reduce_g1(partial, group)


# ---------------------------------------------------------------- sharded proof sessions
ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)


def make_allgather(group=None):
    """ctypes callback for zk_proof_set_sharding: all-gathers `bytes` host bytes from every rank
    (rank-major) with torch.distributed -- over RCCL (tensors staged through the rank's GPU) when
    the backend is nccl, over gloo on the CPU otherwise.  Keep the returned object alive for the
    lifetime of the session."""
    dev = _device_for_backend(group)
    world = dist.get_world_size(group)

    def gather(_user, send_ptr, nbytes, recv_ptr):
        try:
            src = np.ctypeslib.as_array((ctypes.c_uint8 * nbytes).from_address(send_ptr))
            dst = np.ctypeslib.as_array((ctypes.c_uint8 * (nbytes * world)).from_address(recv_ptr))
            t_in = torch.from_numpy(src).to(dev)
            t_out = torch.empty(nbytes * world, dtype=torch.uint8, device=dev)
            dist.all_gather_into_tensor(t_out, t_in, group=group)
            dst[:] = t_out.cpu().numpy()
            return 0
        except Exception as e:   # never let an exception cross the C boundary
            print(f"[zkmi355 sharding] all-gather failed: {e!r}", flush=True)
            return 1
    return ALLGATHER_FN(gather)


def shard_session(session: "binding.ProofSession", group=None):
    """Turns `session` into this rank's share of a multi-GPU proof (same calls on every rank)."""
    cb = make_allgather(group)
    session.set_sharding(dist.get_rank(group), dist.get_world_size(group), cb)
    return cb


class _DevBytes:
    """A raw device pointer as a 1-D uint8 array (CUDA array interface), so torch can wrap it."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def make_allgather_dev(group=None):
    """ctypes callback for zk_proof_set_device_gather: all-gathers DEVICE buffers.  Backend nccl:
    torch tensors over the library's own buffers go straight into RCCL (xGMI, no host copy);
    backend gloo (single-GPU test boxes): staged through host memory."""
    backend = dist.get_backend(group)
    world = dist.get_world_size(group)

    def gather(_user, send_ptr, nbytes, recv_ptr):
        try:
            t_in = torch.as_tensor(_DevBytes(send_ptr, nbytes), device="cuda")
            t_out = torch.as_tensor(_DevBytes(recv_ptr, nbytes * world), device="cuda")
            if backend == "nccl":
                dist.all_gather_into_tensor(t_out, t_in, group=group)
            else:
                h_out = torch.empty(nbytes * world, dtype=torch.uint8)
                dist.all_gather_into_tensor(h_out, t_in.cpu(), group=group)
                t_out.copy_(h_out)
            torch.cuda.synchronize()
            return 0
        except Exception as e:   # never let an exception cross the C boundary
            print(f"[zkmi355 sharding] device all-gather failed: {e!r}", flush=True)
            return 1
    return ALLGATHER_FN(gather)


def shard_session_device(session: "binding.ProofSession", group=None):
    """shard_session plus the device all-gather of the advice columns (each rank uploads 1/world)."""
    cb = shard_session(session, group)
    cb_dev = make_allgather_dev(group)
    session.set_device_gather(cb_dev)
    return cb, cb_dev


class EmulatedRank:
    """Rank `rank` of a `world`-rank sharded session on ONE GPU, its peers played by the local witness: what a rank of a real
    multi-GPU run computes -- its own commitments, the transforms of every column, the (degree class, coset) pairs it is dealt,
    everything the session replicates -- with every exchange served locally:
      * the device all-gather of a phase's advice columns hands back the TRUE columns of the peers (all of them are resident here:
        the lookups and the permutation argument must see a satisfied witness), device to device;
      * the all-gather of commitments and of finished quotient pairs hands back copies of this rank's own contribution.
    The proof that comes out is NOT valid (peers' commitments and quotient values are stand-ins); its wall-clock is this rank's
    device time without any communication -- `bench.py`'s `extra.projected_rank_device_s`, a projection, not a measurement of N GPUs.
    Usage: emu = EmulatedRank(ctx, circ, adv_dev, rank, world); emu.attach(sess); emu.begin_phase(p) before every advice phase p."""

    def __init__(self, ctx, circ, adv_dev: dict, rank: int, world: int):
        self.ctx, self.circ, self.adv, self.rank, self.world = ctx, circ, adv_dev, rank, world
        self.phase_cols: List[int] = []
        self.group = 0
        lib = binding.lib()

        def host_gather(_user, send_ptr, nbytes, recv_ptr):
            try:
                for q in range(world):
                    ctypes.memmove(recv_ptr + q * nbytes, send_ptr, nbytes)
                return 0
            except Exception as e:
                print(f"[zkmi355 emulated rank] gather failed: {e!r}", flush=True)
                return 1

        def dev_gather(_user, send_ptr, nbytes, recv_ptr):
            try:
                cols = self.phase_cols
                colbytes = (1 << self.circ.k) * 32
                per = nbytes // colbytes if cols else 1                     # a phase's columns travel several groups of `world` at a time (ZK_SHARD_EXCHANGE_GROUPS)
                blk = colbytes if cols else nbytes
                for q in range(world):
                    for g in range(per):
                        src = send_ptr + g * blk
                        j = (self.group + g) * world + q
                        if cols and q != rank and j < len(cols) and self.adv.get(cols[j]) is not None:
                            src = self.adv[cols[j]].ptr                   # the peer's column
                        if lib.zk_d2d(ctx.h, ctypes.c_void_p(recv_ptr + (q * per + g) * blk), ctypes.c_void_p(src), ctypes.c_size_t(blk)) != 0:
                            return 1
                if cols:
                    self.group += per
                    if self.group * world >= len(cols):
                        self.phase_cols = []                                # the phase's columns are through: later calls are quotient pairs
                ctx.sync()
                return 0
            except Exception as e:
                print(f"[zkmi355 emulated rank] device gather failed: {e!r}", flush=True)
                return 1
        self._cb, self._cb_dev = ALLGATHER_FN(host_gather), ALLGATHER_FN(dev_gather)

    def attach(self, session):
        session.set_sharding(self.rank, self.world, self._cb)
        session.set_device_gather(self._cb_dev)

    def begin_phase(self, phase: int):
        self.phase_cols = [i for i in range(self.circ.A) if self.circ.advice_phase[i] == phase]
        self.group = 0


# ---------------------------------------------------------------- one NTT over several GPUs
ALLTOALL_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)


def make_alltoall_dev(group=None):
    """ctypes callback for zk_ntt_sharded: all-to-all of DEVICE buffers, bytes_per_peer bytes to and
    from every rank.  Backend nccl: all_to_all_single over the library's own buffers (RCCL posts the
    world - 1 sends/receives as one group, so all xGMI links carry traffic at once -- the transfer
    is per-link bound, not a ring); backend gloo (single-GPU test boxes): staged through the host."""
    backend = dist.get_backend(group)
    world = dist.get_world_size(group)

    def exchange(_user, send_ptr, bytes_per_peer, recv_ptr):
        try:
            total = bytes_per_peer * world
            t_in = torch.as_tensor(_DevBytes(send_ptr, total), device="cuda")
            t_out = torch.as_tensor(_DevBytes(recv_ptr, total), device="cuda")
            if backend == "nccl":
                dist.all_to_all_single(t_out, t_in, group=group)
            else:
                h_in = t_in.cpu()
                blocks = [torch.empty(total, dtype=torch.uint8) for _ in range(world)]
                dist.all_gather(blocks, h_in, group=group)          # gloo has no all_to_all on every build
                rank = dist.get_rank(group)
                h_out = torch.cat([b[rank * bytes_per_peer:(rank + 1) * bytes_per_peer] for b in blocks])
                t_out.copy_(h_out)
            torch.cuda.synchronize()
            return 0
        except Exception as e:   # never let an exception cross the C boundary
            print(f"[zkmi355 sharding] device all-to-all failed: {e!r}", flush=True)
            return 1
    return ALLTOALL_FN(exchange)


def ntt_shard_input(x: np.ndarray, rank: int, world: int) -> np.ndarray:
    """Rows of x (n, 4) this rank holds on entry to the sharded transform: x[rank + world * i]."""
    return np.ascontiguousarray(x[rank::world])


def ntt_shard_output_index(log_n: int, rank: int, world: int) -> np.ndarray:
    """Global output index of every element this rank holds after the sharded transform, in the
    order they sit in its buffer ([j1][c], c < m / world)."""
    m = (1 << log_n) // world
    cols = m // world
    j1 = np.arange(world, dtype=np.int64)[:, None]
    c = np.arange(cols, dtype=np.int64)[None, :]
    return (rank * cols + c + m * j1).reshape(-1)


def comm_init_from_torch(ctx):
    """Joins the library's OWN RCCL communicator (zk_comm_init, csrc/comm.hip) using torch.distributed only
    as the out-of-band channel for rank 0's 128-byte unique id -- what a Rust caller does with a file or
    an environment variable.  Afterwards `session.set_sharding_comm()` / `zk_ntt_sharded(.., NULL, NULL)`
    exchange their data inside the library, stream-ordered, without these Python callbacks."""
    import torch.distributed as dist

    rank, world = dist.get_rank(), dist.get_world_size()
    box = [ctx.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    ctx.comm_init(box[0], rank, world)
    return rank, world

