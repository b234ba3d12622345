"""CPU: tools/trace_stages.py turns a ZK_PROVER_TRACE log into one proof's stage table (sub-marks summed per label, the plan line
and other non-mark lines skipped, several proofs in one log told apart by their multi-open mark)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import trace_stages  # noqa: E402

LOG = """\
[zk prover] advice upload + commits         10.00 ms
[zk prover] multiopen (shplonk)              1.00 ms
[zk prover]   advice: columns sampled        2.50 ms
[zk prover] advice coset plan: free 278.2 + pooled 3.2 GiB, reserve 18.4 GiB -> chosen 0xff
[zk prover] advice upload + commits         20.00 ms
[zk prover]   quotient: program              1.50 ms
[zk prover]   quotient: program              0.50 ms
[zk prover]   quotient: cosets of the columns     0.25 ms
[zk prover] quotient eval + ifft             0.75 ms
[zk prover] multiopen                        4.00 ms
some other stderr line
"""


def test_the_last_proof_is_tabulated():
    ps = trace_stages.proofs(LOG.splitlines())
    assert len(ps) == 2 and len(ps[0]) == 2
    text = trace_stages.table(ps[-1])
    import re
    rows = {re.split(r"\s{2,}", ln.strip())[0]: ln for ln in text.splitlines()}
    assert "22.50" in rows["advice upload + commits"]                      # 20.00 + its sub-mark
    assert "3.00" in rows["quotient eval + ifft"] and "2.00" in rows["quotient: program"]
    assert "29.50" in rows["sum of the marks"]
