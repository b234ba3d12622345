"""CPU: the two Abel-summation identities csrc/runs.hip commits with, checked with the oracle's big-int group law on a small basis
(independent of any device code): a piecewise-constant column by its run ends, and a running sum through its first differences,
both against prefix sums P_e = L_0 + ... + L_e of the basis.  [REF: halo2_proofs commit_lagrange, reached from create_proof's
permutation and lookup arguments -- external crate; the identities are plain algebra over the MSM's definition]"""
import random

from oracle import bn254 as o

R = o.R_MOD


def prefix_points(basis):
    out, acc = [], None
    for p in basis:
        acc = o.g1_add(acc, p)
        out.append(acc)
    return out


def test_piecewise_constant_column_by_its_run_ends():
    rng = random.Random(5)
    n = 48
    basis = [o.g1_mul(o.G1_GEN, rng.randrange(1, R)) for _ in range(n)]
    pfx = prefix_points(basis)
    z, cuts = [], sorted(rng.sample(range(1, n), 6))
    val = rng.randrange(R)
    for i in range(n):
        if i in cuts:
            val = 0 if i == cuts[2] else rng.randrange(R)          # one run of zeros in the middle
        z.append(val)
    want = o.msm_naive(z, basis)
    pairs = [((z[i] - (z[i + 1] if i + 1 < n else 0)) % R, pfx[i]) for i in range(n) if z[i] != (z[i + 1] if i + 1 < n else 0)]
    assert len(pairs) <= len(cuts) + 1
    assert o.msm_naive([s for s, _ in pairs], [p for _, p in pairs]) == want


def test_running_sum_through_its_first_differences():
    rng = random.Random(6)
    n = 48
    basis = [o.g1_mul(o.G1_GEN, rng.randrange(1, R)) for _ in range(n)]
    pfx = prefix_points(basis)
    c = rng.randrange(R)
    active = set(rng.sample(range(n - 1), 5))
    phi, acc = [], rng.randrange(R)                                  # phi_0 need not be zero
    for j in range(n):
        phi.append(acc)
        acc = (acc + (rng.randrange(R) if j in active else c)) % R
    phi[n - 2], phi[n - 1] = rng.randrange(R), rng.randrange(R)      # blinding rows
    want = o.msm_naive(phi, basis)
    s = [(c - (phi[j + 1] - phi[j])) % R for j in range(n - 1)] + [phi[n - 1]]
    assert sum(1 for v in s[:-1] if v) <= len(active) + 3           # zero wherever the increment is the common one
    neg_total = None
    for j in range(n - 1):
        neg_total = o.g1_add(neg_total, pfx[j])
    neg_total = o.g1_neg(neg_total)
    got = o.g1_add(o.msm_naive(s, pfx), o.g1_mul(neg_total, c))
    assert got == want
