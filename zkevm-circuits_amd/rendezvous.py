"""Joining the library's RCCL communicator without torch in the process (csrc/comm.hip: zk_comm_unique_id / zk_comm_init).

A prover rank started by any launcher that exports RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT (torch.distributed.run,
mpirun wrappers, a shell loop) needs one out-of-band step -- rank 0's 128-byte id must reach the other ranks -- and a
file does it.  Keeping torch out matters for measurements: torch bundles its own HIP runtime, and with two runtimes in
one process the witness uploads of a proving session ran 30 % slower (tools/upload_order.py)."""
import os
def _alive(pid: int) -> bool:
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    except PermissionError:
        return True
    return True


def _host_tag() -> int:
    import hashlib
    import socket
    return int.from_bytes(hashlib.blake2b(socket.gethostname().encode(), digest_size=8).digest(), "little")


def _launch_nonce() -> bytes:
    """16 bytes that every rank of ONE launch derives alike and another launch does not: ZK_COMM_NONCE when the launcher
    exports one (any string), else zeros (the pid check below is then the only guard against a stale file)."""
    import hashlib
    v = os.environ.get("ZK_COMM_NONCE", "")
    return hashlib.blake2b(v.encode(), digest_size=16).digest() if v else bytes(16)


ID_FILE_LEN = 128 + 8 + 8 + 16


def exchange_unique_id(make_id, rank: int, world: int, path: str, timeout: float = 120.0) -> bytes:
    """Rank 0's 128-byte RCCL id to every rank through a file: the whole out-of-band channel a launcher needs to
    provide (no torch, no MPI).  `make_id` is only called on rank 0.  The file appears atomically (write + rename)
    and carries the id followed by rank 0's pid, a tag of its host name and the launch nonce.  A file whose nonce
    differs is another launch's; with equal nonces, a reader ON RANK 0's HOST also requires the writer to be alive (a
    leftover of an earlier launch from the same shell / port / fixed ZK_COMM_ID_FILE is never accepted), while a reader
    on ANOTHER host (a path on a shared file system) cannot see that pid and relies on the nonce: there ZK_COMM_NONCE is
    REQUIRED -- a file from a foreign host with the all-zero nonce is refused.  Rank 0 removes
    whatever is at `path` before it even creates its id, and `retire_unique_id` removes the file once every rank has joined."""
    import struct
    import time

    if rank == 0:
        try:
            os.unlink(path)
        except FileNotFoundError:
            pass
        uid = make_id()
        assert len(uid) == 128
        if world > 1:
            tmp = f"{path}.{os.getpid()}.tmp"
            with open(tmp, "wb") as f:
                f.write(uid + struct.pack("<QQ", os.getpid(), _host_tag()) + _launch_nonce())
            os.replace(tmp, path)
        return uid
    deadline = time.monotonic() + timeout
    warned = False
    while True:
        try:
            with open(path, "rb") as f:
                blob = f.read()
            if len(blob) == ID_FILE_LEN and blob[144:] == _launch_nonce():
                pid, host = struct.unpack("<QQ", blob[128:144])
                if host == _host_tag():
                    if _alive(pid):
                        return blob[:128]
                elif blob[144:] != bytes(16):
                    return blob[:128]
                elif not warned:
                    # a writer on another host cannot be checked for liveness: without a launch nonce a leftover file of an
                    # earlier launch would be joined (and hang until the timeout) -- refused, and said so once
                    warned = True
                    print(f"[zkmi355 rendezvous] rank {rank}: the id file at {path} was written on another host and carries no launch nonce: "
                          "export ZK_COMM_NONCE (any string, the same on every rank of this launch) for ranks spread over hosts", flush=True)
        except FileNotFoundError:
            pass
        if time.monotonic() > deadline:
            raise TimeoutError(f"rank {rank}: no communicator id at {path} after {timeout:.0f} s")
        time.sleep(0.01)


def retire_unique_id(ctx, rank: int, world: int, path: str):
    """After zk_comm_init: every rank has read the id (first barrier), rank 0 removes the file, and nobody leaves
    before it is gone (second barrier) -- a second rendezvous in the same launch cannot pick up this one's id."""
    if world <= 1:
        return
    comm_barrier(ctx, rank, world)
    if rank == 0:
        try:
            os.unlink(path)
        except FileNotFoundError:
            pass
    comm_barrier(ctx, rank, world)


def comm_init_from_env(ctx, timeout: float = 120.0):
    """Joins the library's RCCL communicator from the launcher's environment alone (RANK, WORLD_SIZE, MASTER_PORT): rank 0
    creates the unique id and publishes it in a file named after the launch (port + the launcher's pid, the parent of
    every rank); stale files are refused (see exchange_unique_id) and the file is removed once all ranks have joined.
    No torch in the process: a prover rank holds the library and nothing else."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost", "::1"):
        # one node, rendezvous over loopback: RCCL's bootstrap must not go looking for another interface (the containers this runs
        # in may have none, or one that does not route between the ranks)
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    path = os.environ.get("ZK_COMM_ID_FILE") or os.path.join(
        os.environ.get("TMPDIR", "/tmp"), f"zkmi355_comm_{os.environ.get('MASTER_PORT', '29500')}_{os.getppid()}")
    uid = exchange_unique_id(ctx.comm_unique_id, rank, world, path, timeout)
    ctx.comm_init(uid, rank, world)
    retire_unique_id(ctx, rank, world, path)
    return rank, world


def comm_barrier(ctx, rank: int, world: int):
    """All ranks of the library's communicator meet here: a one-byte-per-rank all-gather, then the stream is drained."""
    send, recv = ctx.alloc(256), ctx.alloc(256 * max(world, 1))
    ctx.comm_allgather(send, 256, recv)
    ctx.sync()
    send.free()
    recv.free()

