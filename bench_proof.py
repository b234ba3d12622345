#!/usr/bin/env python
"""bench_proof.py -- wall-clock of a FULL proof (keygen_pk + create_proof) on one MI355X for a
synthetic-shape PLONKish circuit, verified afterwards by the oracle's pairing-based verifier.

BASELINE.json's metric is "SuperCircuit proof-gen wall-clock (s) at k"; a real SuperCircuit /
Keccak witness needs the reference's Rust + Go toolchain (absent here), so this uses a
synthetic circuit with the same ingredients (custom gates with rotations up to degree 5, a chunked
permutation argument over every advice column, logUp lookups) and labels every number
"synthetic-shape".  Not the driver's bench (that is bench.py); prints one JSON line.

usage: python bench_proof.py [--k 16] [--groups 10] [--no-verify]
"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import zkevm_circuits_amd as z  # noqa: E402
from zkevm_circuits_amd import plonk  # noqa: E402

R = plonk.R_MOD


def build(k: int, groups: int, seed: int = 1):
    """groups x (a, b, c) advice triples; gates per group: q_mul (a b - c), q_add (a + b - c.next);
    one degree-5 gate on group 0; one 2-column lookup on group 0; every advice column is in the
    permutation, with a chain of copy constraints per group."""
    rng = random.Random(seed)
    A, F = 3 * groups, 6
    c = plonk.Circuit(k, num_fixed=F, num_advice=A, num_instance=1, blinding_factors=5)
    n, u = c.n, c.u
    q_mul, q_add, q_cube, q_lk, t_a, t_b = (c.fixed_col(i) for i in range(F))
    for g in range(groups):
        a, b_, cc = (c.advice_col(3 * g + i) for i in range(3))
        c.add_gate(q_mul * (a * b_ - cc))
        c.add_gate(q_add * (a + b_ - cc.rot(1)))
    a0, b0, c0 = c.advice_col(0), c.advice_col(1), c.advice_col(2)
    c.add_gate(q_cube * (a0 * a0 * a0 * b0 + 7 - c0.rot(-1)))
    c.add_lookup([q_lk * a0, q_lk * b0], [t_a, t_b])
    tab_n = min(4096, u)
    for row in range(u):
        if 0 < row < tab_n:
            c.fixed[4][row], c.fixed[5][row] = row, (row * row + 3) % R
    adv = [[0] * n for _ in range(A)]
    inst = [[0] * n]
    kinds = [rng.randrange(4) for _ in range(u // 3)]
    for ri, kind in enumerate(kinds):
        row = 1 + 3 * ri
        if row + 3 >= u:
            break
        if kind == 0:
            c.fixed[0][row] = 1
            for g in range(groups):
                x, y = rng.getrandbits(60), rng.getrandbits(60)
                adv[3 * g][row], adv[3 * g + 1][row], adv[3 * g + 2][row] = x, y, x * y % R
        elif kind == 1:
            c.fixed[1][row] = 1
            for g in range(groups):
                x, y = rng.getrandbits(60), rng.getrandbits(60)
                adv[3 * g][row], adv[3 * g + 1][row], adv[3 * g + 2][row + 1] = x, y, x + y
        elif kind == 2:
            c.fixed[2][row + 1] = 1
            x, y = rng.getrandbits(60), rng.getrandbits(60)
            adv[0][row + 1], adv[1][row + 1] = x, y
            adv[2][row] = (x * x * x * y + 7) % R
        else:
            c.fixed[3][row] = 1
            i = rng.randrange(1, tab_n)
            adv[0][row], adv[1][row] = i, (i * i + 3) % R
    for col in range(A):
        c.enable_equality(plonk.ADVICE, col)
    # copy constraints: c of one mul row feeds a of a later mul row (same group); a few public inputs
    mul_rows = [1 + 3 * ri for ri, kd in enumerate(kinds) if kd == 0 and 1 + 3 * ri + 3 < u]
    for g in range(groups):
        for r0, r1 in list(zip(mul_rows[:-1:2], mul_rows[1::2]))[:256]:
            x = adv[3 * g + 2][r0]
            adv[3 * g][r1] = x
            adv[3 * g + 2][r1] = x * adv[3 * g + 1][r1] % R
            c.copy((plonk.ADVICE, 3 * g + 2, r0), (plonk.ADVICE, 3 * g, r1))
    for j, r0 in enumerate(mul_rows[:4]):
        inst[0][j] = adv[1][r0]
        c.copy((plonk.ADVICE, 1, r0), (plonk.INSTANCE, 0, j))
    return c, adv, inst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--groups", type=int, default=10)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--no-verify", action="store_true")
    args = ap.parse_args()

    t0 = time.perf_counter()
    circ, adv, inst = build(args.k, args.groups)
    blob = circ.blob()
    adv_m = [plonk.column_to_mont(c) for c in adv]
    inst_m = [plonk.column_to_mont(c) for c in inst]
    t_build = time.perf_counter() - t0

    ctx = z.Context(0)
    S = 0x5EC2E7
    s_mont = np.frombuffer(plonk.fr_mont_bytes(S), dtype=np.uint64).copy()
    t0 = time.perf_counter()
    srs = ctx.srs_setup_with_s(args.k, s_mont)
    ctx.sync()
    t_srs = time.perf_counter() - t0
    t0 = time.perf_counter()
    pk = ctx.pk_create(srs, blob)
    ctx.sync()
    t_keygen = time.perf_counter() - t0
    times = []
    proof = b""
    for _ in range(args.repeat):
        t0 = time.perf_counter()
        proof = ctx.create_proof(pk, adv_m, inst_m, bytes(16))
        times.append(time.perf_counter() - t0)
    ok = None
    if not args.no_verify:
        from oracle import cref, pairing as pr, plonk_verifier as pv
        com, rep = pk.vk(circ.F + len(circ.perm_cols))
        ok = bool(pv.verify(circ, cref.affine_from_mont(com), cref.from_mont(rep.reshape(1, 4))[0], inst, proof, pr.ec_mul(pr.G2_GEN, S)))
    d = circ.degree()
    out = {
        "metric": "synthetic-shape full proof wall-clock (s), 1x MI355X",
        "value": round(min(times), 4), "unit": "s", "higher_is_better": False,
        "k": args.k, "advice": circ.A, "fixed": circ.F, "instance": circ.I, "permutation_columns": len(circ.perm_cols),
        "lookups": len(circ.lookups), "gates": len(circ.gates), "degree": d, "extended_k": circ.extended_k(),
        "proof_bytes": len(proof), "create_proof_s": [round(t, 4) for t in times], "keygen_pk_s": round(t_keygen, 4),
        "srs_setup_s": round(t_srs, 4), "host_circuit_build_s": round(t_build, 2), "verified_by_oracle": ok,
        "msm_count": circ.A + 2 * len(circ.lookups) + (len(circ.perm_cols) + d - 3) // (d - 2) + 1 + (d - 1),
        "data": "synthetic-shape",
    }
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
