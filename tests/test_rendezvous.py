"""File rendezvous of the library's RCCL communicator id (zkevm-circuits_amd/rendezvous.py): CPU test of the channel
itself -- rank 0 publishes atomically, the others wait for exactly 128 bytes, no torch gets imported."""
import multiprocessing as mp
import os
import sys
import time


def _rank(rank, world, path, q):
    from zkevm_circuits_amd import rendezvous

    if rank == 0:
        time.sleep(0.3)                      # the others must wait, not fail
    uid = rendezvous.exchange_unique_id(lambda: bytes(range(128)), rank, world, path, timeout=20.0)
    q.put((rank, uid, "torch" in sys.modules))


def test_unique_id_reaches_every_rank_without_torch(tmp_path):
    path = str(tmp_path / "uid")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(r, 3, path, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=60) for _ in procs)
    for p in procs:
        p.join(timeout=30)
    assert [g[0] for g in got] == [0, 1, 2]
    assert all(g[1] == bytes(range(128)) for g in got)
    assert not any(g[2] for g in got), "the rendezvous must not pull torch into a prover rank"
    assert os.path.getsize(path) == 160          # the id + rank 0's pid, host tag and the launch nonce (stale-file checks)


def test_missing_id_times_out(tmp_path):
    from zkevm_circuits_amd import rendezvous
    import pytest

    with pytest.raises(TimeoutError):
        rendezvous.exchange_unique_id(lambda: b"", 1, 2, str(tmp_path / "never"), timeout=0.2)


def test_stale_id_of_a_finished_launch_is_refused(tmp_path):
    """a shell-loop launcher reuses port and parent pid, so the id file of the previous launch may still be
    there when the ranks of the next one start.  A peer must not take it: its writer (rank 0 of the old launch) is gone."""
    import struct
    import subprocess
    import pytest
    from zkevm_circuits_amd import rendezvous

    path = str(tmp_path / "uid")
    dead = subprocess.Popen([sys.executable, "-c", "pass"])
    dead.wait()
    with open(path, "wb") as f:
        f.write(bytes(128) + struct.pack("<QQ", dead.pid, rendezvous._host_tag()) + bytes(16))           # left behind by a rank 0 of this host that has exited
    with pytest.raises(TimeoutError):
        rendezvous.exchange_unique_id(lambda: b"", 1, 2, path, timeout=0.3)
    # written on ANOTHER host (shared file system): its pid cannot be looked up here.  Without a launch nonce (all zeros) it could be
    # the leftover of any earlier launch: refused ...
    with open(path, "wb") as f:
        f.write(bytes(range(128)) + struct.pack("<QQ", dead.pid, rendezvous._host_tag() ^ 1) + bytes(16))
    with pytest.raises(TimeoutError):
        rendezvous.exchange_unique_id(lambda: b"", 1, 2, path, timeout=0.3)
    # ... accepted on the nonce alone when the launch has one, and refused when the nonce is another launch's, whoever wrote it
    os.environ["ZK_COMM_NONCE"] = "launch-1"
    try:
        with open(path, "wb") as f:
            f.write(bytes(range(128)) + struct.pack("<QQ", dead.pid, rendezvous._host_tag() ^ 1) + rendezvous._launch_nonce())
        assert rendezvous.exchange_unique_id(lambda: b"", 1, 2, path, timeout=0.3) == bytes(range(128))
        os.environ["ZK_COMM_NONCE"] = "launch-2"
        with pytest.raises(TimeoutError):
            rendezvous.exchange_unique_id(lambda: b"", 1, 2, path, timeout=0.3)
    finally:
        del os.environ["ZK_COMM_NONCE"]
    with open(path, "wb") as f:
        f.write(bytes(128))                                          # the old format (no pid) is not accepted either
    with pytest.raises(TimeoutError):
        rendezvous.exchange_unique_id(lambda: b"", 1, 2, path, timeout=0.3)
    # rank 0 of the new launch replaces whatever is there
    assert rendezvous.exchange_unique_id(lambda: bytes(range(128)), 0, 2, path) == bytes(range(128))
    assert rendezvous.exchange_unique_id(lambda: b"", 1, 2, path, timeout=5.0) == bytes(range(128))
