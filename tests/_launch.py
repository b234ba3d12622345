"""Launcher shared by the multi-process GPU tests (torch.distributed.run over gloo, ranks sharing one device): a bounded wait
and ONE fresh attempt when the first one does not finish.

Why: on 2026-09-24 a full `pytest -m gpu` run on a freshly leased box sat in `test_sharded_session_matches_single[2-7-1-1]` until
the lease ran out, with a library that was byte-for-byte the one that had passed the same case in every earlier run -- a rendezvous
or collective that never completed, not a wrong result.  A case normally takes seconds; waiting ten minutes for it proves nothing.
A retry is reported as a warning, so a recurring hang stays visible; two attempts that both run into the limit fail the test."""
import socket
import subprocess
import sys
import warnings


def free_port() -> int:
    with socket.socket() as sock:          # a port that is free right now (cases run back to back)
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def run_ranks(world: int, worker: str, args, env, timeout: float = 240.0, attempts: int = 2, launcher=None):
    """`python -m torch.distributed.run --nproc-per-node world worker args...` on 127.0.0.1; returns the CompletedProcess of the
    attempt that finished.  `launcher` overrides the command prefix (tests of this helper)."""
    last = None
    for attempt in range(attempts):
        prefix = launcher or [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                              "--master-port", str(free_port())]
        proc = subprocess.Popen(prefix + [worker] + [str(a) for a in args], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
        try:
            out, err = proc.communicate(timeout=timeout)
            return subprocess.CompletedProcess(proc.args, proc.returncode, out, err)
        except subprocess.TimeoutExpired as e:
            import os
            import signal
            try:
                os.killpg(proc.pid, signal.SIGKILL)        # the launcher and the ranks it started: exactly this process group
            except ProcessLookupError:
                pass
            out, err = proc.communicate()
            last = (out, err)
            if attempt + 1 < attempts:
                warnings.warn(f"{worker}: {world} ranks did not finish within {timeout:.0f} s (attempt {attempt + 1}); trying once more on a new port")
    raise AssertionError(f"{worker}: {world} ranks did not finish within {timeout:.0f} s, {attempts} times\n" + (last[0] or "")[-2000:] + (last[1] or "")[-4000:])
