"""CPU: the launcher of the multi-process GPU tests (tests/_launch.py) -- a worker that does not finish is killed with its whole
process group and tried once more; two misses fail; a worker that finishes is returned as it ran."""
import os
import sys
import textwrap
import warnings

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _launch import run_ranks  # noqa: E402


def _worker(tmp_path, body):
    p = tmp_path / "w.py"
    p.write_text(textwrap.dedent(body))
    return str(p)


def test_a_finished_worker_is_returned(tmp_path):
    w = _worker(tmp_path, """
        import sys
        print("args", sys.argv[1:])
        sys.exit(3)
    """)
    res = run_ranks(1, w, ["a", 7], dict(os.environ), timeout=30, launcher=[sys.executable])
    assert res.returncode == 3 and "args ['a', '7']" in res.stdout


def test_a_hung_first_attempt_is_killed_and_retried(tmp_path):
    # first run: leaves a marker, starts a child and sleeps; second run: sees the marker and finishes
    w = _worker(tmp_path, """
        import os, subprocess, sys, time
        marker = sys.argv[1]
        if not os.path.exists(marker):
            open(marker, "w").write(str(os.getpid()))
            child = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(600)"])
            open(marker + ".child", "w").write(str(child.pid))
            time.sleep(600)
        print("second attempt")
    """)
    marker = str(tmp_path / "marker")
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        res = run_ranks(2, w, [marker], dict(os.environ), timeout=3, launcher=[sys.executable])
    assert res.returncode == 0 and "second attempt" in res.stdout
    assert any("did not finish" in str(c.message) for c in caught)
    # the first attempt's process group is gone: the launcher and the child it started
    for f in (marker, marker + ".child"):
        pid = int(open(f).read())
        try:
            state = open(f"/proc/{pid}/stat").read().rsplit(")", 1)[1].split()[0]
        except FileNotFoundError:
            state = "gone"
        assert state in ("gone", "Z"), (f, state)        # killed; an orphan may wait as a zombie for the container's init to collect it


def test_two_misses_fail(tmp_path):
    w = _worker(tmp_path, """
        import time
        print("started", flush=True)
        time.sleep(600)
    """)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with pytest.raises(AssertionError, match="did not finish within 2 s, 2 times"):
            run_ranks(2, w, [], dict(os.environ), timeout=2, launcher=[sys.executable])
