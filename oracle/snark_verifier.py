"""CPU oracle (TEST INFRASTRUCTURE ONLY) -- a `PlonkProtocol`-driven verifier: snark-verifier's
`PlonkSuccinctVerifier<KzgAs<Bn256, Bdfg21>>::verify` + `KzgDecidingKey` decider, restated for the native
loader and the Poseidon transcript.

This is the verifier the reference runs on every chunk / batch snark:
  verify_snark_shplonk                        [REF prover/src/common/verifier.rs:35]
  extract_accumulators_and_proof              [REF aggregator/src/core.rs:48-107] (Poseidon transcript at :57-58,
                                              `PlonkSuccinctVerifier::read_proof` + `verify` at :60-75)
  extract_proof_and_instances_with_pairing_check [REF aggregator/src/core.rs:111-147] (limbs -> accumulator, decider)
The code itself lives in snark-verifier @ 572ef69 (scroll-tech/snark-verifier, branch develop) which is NOT on disk
[REF Cargo.lock: snark-verifier-sdk]; what is restated here is its published algorithm (`verifier/plonk.rs`,
`verifier/plonk/protocol.rs`, `pcs/kzg/multiopen/bdfg21.rs`, `pcs/kzg/decider.rs`, `system/halo2/transcript/halo2.rs`).

PINNED: unlike the rest of the protocol-level oracle, this module is checked against a proof the reference itself
produced and ships -- `aggregator/data/batch-task.json: chunk_proofs[0]` (struct [REF prover/src/proof/chunk.rs:10-19],
[REF prover/src/proof.rs:25-35]) with `s_g2` from [REF prover/src/utils.rs:36]: tests/test_reference_chunk_proof.py
requires `verify_snark` to accept it and to reject single-bit flips (fixture extracted by
tests/golden/make_reference_vectors.py into tests/golden/reference_chunk_proof.json).

The protocol object is the serde_json image of snark-verifier's `PlonkProtocol<G1Affine>` (field elements as four
u64 Montgomery limbs, little-endian; points as {x, y} of such limbs).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

from . import bn254 as b
from . import pairing as pr
from .transcripts import Poseidon, decompress_g1

R, P = b.R_MOD, b.P_MOD
LIMBS, BITS = 3, 88          # [REF aggregator/src/constants.rs:80-82]


# ------------------------------------------------------------------------------------ serde images
def fe_from_limbs(limbs: Sequence[int], mod: int) -> int:
    """four u64 Montgomery limbs (what serde derives for halo2curves' `Fr([u64; 4])`) -> canonical integer"""
    v = sum(int(l) << (64 * i) for i, l in enumerate(limbs))
    return b.from_mont(v, mod)


def point_from_json(o) -> Optional[Tuple[int, int]]:
    x, y = fe_from_limbs(o["x"], P), fe_from_limbs(o["y"], P)
    if x == 0 and y == 0:
        return None
    assert b.g1_is_on_curve((x, y))
    return (x, y)


class Protocol:
    """snark-verifier `PlonkProtocol` (util/protocol.rs? -> verifier/plonk/protocol.rs)"""

    def __init__(self, j: dict):
        d = j["domain"]
        self.k, self.n = d["k"], d["n"]
        self.n_inv = fe_from_limbs(d["n_inv"], R)
        self.gen = fe_from_limbs(d["gen"], R)
        self.gen_inv = fe_from_limbs(d["gen_inv"], R)
        assert self.n == 1 << self.k and self.n_inv * self.n % R == 1 and self.gen * self.gen_inv % R == 1
        assert self.gen == b.omega_for_k(self.k), "domain generator is not ROOT_OF_UNITY^(2^(28-k))"
        self.preprocessed = [point_from_json(p) for p in j["preprocessed"]]
        self.num_instance: List[int] = j["num_instance"]
        self.num_witness: List[int] = j["num_witness"]
        self.num_challenge: List[int] = j["num_challenge"]
        self.evaluations = [(q["poly"], q["rotation"]) for q in j["evaluations"]]
        self.queries = [(q["poly"], q["rotation"]) for q in j["queries"]]
        self.quotient = j["quotient"]
        tis = j.get("transcript_initial_state")
        self.transcript_initial_state = None if tis is None else fe_from_limbs(tis, R)
        assert j.get("instance_committing_key") is None, "committed instances (IPA-style) are not what the reference uses"
        assert j.get("linearization") is None, "halo2 protocols carry no linearization"
        self.accumulator_indices = [[tuple(ix) for ix in acc] for acc in j.get("accumulator_indices", [])]

    # polynomial numbering: preprocessed | instance | witness | quotient (one combined polynomial)
    def instance_offset(self) -> int: return len(self.preprocessed)
    def witness_offset(self) -> int: return self.instance_offset() + len(self.num_instance)
    def quotient_poly(self) -> int: return self.witness_offset() + sum(self.num_witness)

    def rotate(self, rot: int) -> int:
        """domain.rotate_scalar(1, rotation)"""
        return pow(self.gen, rot, R) if rot >= 0 else pow(self.gen_inv, -rot, R)

    def lagranges(self) -> List[int]:
        """the Lagrange indices the protocol needs: those of the numerator plus one per instance row"""
        out = set()

        def walk(e):
            if isinstance(e, dict):
                for k_, v in e.items():
                    if k_ == "CommonPolynomial" and isinstance(v, dict) and "Lagrange" in v:
                        out.add(v["Lagrange"])
                    else:
                        walk(v)
            elif isinstance(e, list):
                for x in e:
                    walk(x)
        walk(self.quotient["numerator"])
        for i in range(max(self.num_instance + [0])):
            out.add(i)
        return sorted(out)


# ------------------------------------------------------------------------------------ expression evaluation
def evaluate_expression(e, poly_eval, challenge, common, constant=lambda limbs: fe_from_limbs(limbs, R)) -> int:
    """snark-verifier `Expression::evaluate`; DistributePowers(exprs, s) is Horner: acc * s + next."""
    if isinstance(e, str):
        raise ValueError(f"unexpected bare variant {e}")
    (kind, v), = e.items()
    ev = lambda x: evaluate_expression(x, poly_eval, challenge, common, constant)
    if kind == "Constant":
        return constant(v)
    if kind == "CommonPolynomial":
        return common(v)
    if kind == "Polynomial":
        return poly_eval(v["poly"], v["rotation"])
    if kind == "Challenge":
        return challenge(v)
    if kind == "Negated":
        return (-ev(v)) % R
    if kind == "Sum":
        return (ev(v[0]) + ev(v[1])) % R
    if kind == "Product":
        return ev(v[0]) * ev(v[1]) % R
    if kind == "Scaled":
        return ev(v[0]) * constant(v[1]) % R
    if kind == "DistributePowers":
        exprs, scalar = v
        assert exprs
        acc = ev(exprs[0])
        if len(exprs) == 1:
            return acc
        s = ev(scalar)
        for x in exprs[1:]:
            acc = (acc * s + ev(x)) % R
        return acc
    raise ValueError(f"unknown expression variant {kind}")


# ------------------------------------------------------------------------------------ Bdfg21 (SHPLONK) succinct verification
def query_sets(queries):
    """bdfg21.rs `query_sets`: polynomials in order of first appearance, each with its shifts in order of
    appearance; sets in order of first appearance, a polynomial joins the set whose shift SET equals its own and takes
    that set's shift order.  queries: (poly, shift (rotation), eval) -> [(shifts, polys, evals[poly][shift position])]"""
    poly_shifts: List[Tuple[int, List[int], List[int]]] = []
    for poly, shift, ev in queries:
        for p_, shifts, evals in poly_shifts:
            if p_ == poly:
                if shift not in shifts:
                    shifts.append(shift)
                    evals.append(ev)
                break
        else:
            poly_shifts.append((poly, [shift], [ev]))
    sets: List[Tuple[List[int], List[int], List[List[int]]]] = []
    for poly, shifts, evals in poly_shifts:
        for s_shifts, s_polys, s_evals in sets:
            if set(s_shifts) == set(shifts):
                if poly not in s_polys:
                    s_polys.append(poly)
                    s_evals.append([evals[shifts.index(s)] for s in s_shifts])
                break
        else:
            sets.append((list(shifts), [poly], [list(evals)]))
    return sets


def _interp_eval(xs: Sequence[int], ys: Sequence[int], at: int) -> int:
    """value at `at` of the polynomial through (xs[j], ys[j]) -- what the barycentric coefficients of
    `QuerySetCoeff` compute"""
    tot = 0
    for j, (xj, yj) in enumerate(zip(xs, ys)):
        num, den = 1, 1
        for i, xi in enumerate(xs):
            if i != j:
                num = num * (at - xi) % R
                den = den * (xj - xi) % R
        tot = (tot + yj * num % R * b.fr_inv(den)) % R
    return tot


def bdfg21_verify(protocol: Protocol, commitments: Dict[int, object], z: int, queries, mu: int, gamma: int, w, z_prime: int, w_prime):
    """bdfg21.rs `Bdfg21::verify` -> the KZG accumulator (lhs, rhs) with e(lhs, g2) = e(rhs, s_g2) iff the openings hold.
    queries: (poly, rotation, eval)."""
    sets = query_sets(queries)
    z_s = []
    for shifts, _, _ in sets:
        acc = 1
        for s in shifts:
            acc = acc * (z_prime - z * protocol.rotate(s)) % R
        z_s.append(acc)
    z_s_1 = z_s[0]
    f, const = None, 0
    gpow = 1
    for i, (shifts, polys, evals) in enumerate(sets):
        coeff = 1 if i == 0 else z_s_1 * b.fr_inv(z_s[i]) % R
        xs = [z * protocol.rotate(s) % R for s in shifts]
        mpow = 1
        for poly, ev in zip(polys, evals):
            sc = gpow * mpow % R * coeff % R
            f = b.g1_add(f, b.g1_mul(commitments[poly], sc))
            const = (const + sc * _interp_eval(xs, ev, z_prime)) % R
            mpow = mpow * mu % R
        gpow = gpow * gamma % R
    f = b.g1_add(f, b.g1_neg(b.g1_mul(b.G1_GEN, const)))
    f = b.g1_add(f, b.g1_neg(b.g1_mul(w, z_s_1)))
    lhs = b.g1_add(f, b.g1_mul(w_prime, z_prime))
    return lhs, w_prime


# ------------------------------------------------------------------------------------ accumulators
def limbs_to_fq(limbs: Sequence[int]) -> int:
    """fe_from_limbs::<_, _, LIMBS, BITS>: little-endian limbs of BITS bits"""
    v = 0
    for i, l in enumerate(limbs):
        assert l < (1 << BITS), "accumulator limb out of range"
        v |= l << (BITS * i)
    assert v < P
    return v


def accumulator_from_limbs(limbs: Sequence[int]):
    """LimbsEncoding<LIMBS, BITS>::from_repr: [lhs.x, lhs.y, rhs.x, rhs.y] x LIMBS [REF aggregator/src/core.rs:120-135]"""
    assert len(limbs) == 4 * LIMBS
    c = [limbs_to_fq(limbs[i * LIMBS:(i + 1) * LIMBS]) for i in range(4)]
    lhs, rhs = (c[0], c[1]), (c[2], c[3])
    assert b.g1_is_on_curve(lhs) and b.g1_is_on_curve(rhs), "accumulator limbs are not curve points"
    return lhs, rhs


def decide(acc, g2, s_g2) -> bool:
    """KzgDecidingKey: e(lhs, g2) == e(rhs, s_g2) [REF aggregator/src/core.rs:137-146]"""
    lhs, rhs = acc
    return pr.pairing_check([(lhs, g2), (b.g1_neg(rhs), s_g2)])


# ------------------------------------------------------------------------------------ the verifier
class Transcribed:
    """what PlonkProof::read takes off the transcript, kept for tests that replay it elsewhere"""
    witnesses: List
    challenges: List[int]
    quotients: List
    z: int
    evaluations: List[int]
    mu: int
    gamma: int
    w: object
    z_prime: int
    w_prime: object


def read_proof(protocol: Protocol, instances: Sequence[Sequence[int]], proof: bytes, transcript=None) -> Transcribed:
    """plonk.rs `PlonkProof::read` + bdfg21.rs `Bdfg21Proof::read`"""
    tr = transcript if transcript is not None else Poseidon(proof)
    t = Transcribed()
    if protocol.transcript_initial_state is not None:
        tr.common_scalar(protocol.transcript_initial_state)
    assert [len(c) for c in instances] == protocol.num_instance, "instance shape differs from the protocol's"
    for col in instances:
        for v in col:
            tr.common_scalar(v)
    t.witnesses, t.challenges = [], []
    for nw, nc in zip(protocol.num_witness, protocol.num_challenge):
        t.witnesses += [tr.read_point() for _ in range(nw)]
        t.challenges += [tr.squeeze() for _ in range(nc)]
    t.quotients = [tr.read_point() for _ in range(protocol.quotient["num_chunk"])]
    t.z = tr.squeeze()
    t.evaluations = [tr.read_scalar() for _ in protocol.evaluations]
    t.mu, t.gamma = tr.squeeze(), tr.squeeze()
    t.w = tr.read_point()
    t.z_prime = tr.squeeze()
    t.w_prime = tr.read_point()
    assert tr.exhausted(), "trailing bytes after the proof"
    return t


def succinct_verify(protocol: Protocol, instances: Sequence[Sequence[int]], proof: bytes):
    """plonk.rs `PlonkSuccinctVerifier::verify`: the accumulators a decider has to check -- the new one of this proof first,
    then the ones carried in the instances (`accumulator_indices`)."""
    t = read_proof(protocol, instances, proof)
    n, z = protocol.n, t.z
    zn = pow(z, n, R)
    zn_minus_one = (zn - 1) % R
    numer = zn_minus_one * protocol.n_inv % R

    def lagrange(i: int) -> int:
        wi = protocol.rotate(i)
        return numer * wi % R * b.fr_inv((z - wi) % R) % R

    def common(v):
        if v == "Identity":
            return z
        return lagrange(v["Lagrange"])

    evals: Dict[Tuple[int, int], int] = {}
    for c, col in enumerate(instances):
        evals[(protocol.instance_offset() + c, 0)] = sum(v * lagrange(i) for i, v in enumerate(col)) % R
    for q, e in zip(protocol.evaluations, t.evaluations):
        evals[q] = e
    numerator = evaluate_expression(protocol.quotient["numerator"], lambda p_, r_: evals[(p_, r_)], lambda i: t.challenges[i], common)
    qpoly = protocol.quotient_poly()
    evals[(qpoly, 0)] = numerator * b.fr_inv(zn_minus_one) % R

    commitments: Dict[int, object] = {i: pt for i, pt in enumerate(protocol.preprocessed)}
    for i, pt in enumerate(t.witnesses):
        commitments[protocol.witness_offset() + i] = pt
    chunk_pow = pow(z, n * protocol.quotient["chunk_degree"], R)
    hc = None
    for pt in reversed(t.quotients):
        hc = b.g1_add(b.g1_mul(hc, chunk_pow) if hc is not None else None, pt)
    commitments[qpoly] = hc

    queries = [(p_, r_, evals[(p_, r_)]) for p_, r_ in protocol.queries]
    accs = [bdfg21_verify(protocol, commitments, z, queries, t.mu, t.gamma, t.w, t.z_prime, t.w_prime)]
    for indices in protocol.accumulator_indices:
        accs.append(accumulator_from_limbs([instances[c][r] for c, r in indices]))
    return accs


def verify_snark(protocol: Protocol, instances: Sequence[Sequence[int]], proof: bytes, g2, s_g2) -> bool:
    """verify_snark_shplonk [REF prover/src/common/verifier.rs:35]: succinct verification + decide_all"""
    try:
        accs = succinct_verify(protocol, instances, proof)
    except (AssertionError, ZeroDivisionError, ValueError, KeyError):
        return False
    return all(decide(a, g2, s_g2) for a in accs)


def instances_from_bytes(raw: bytes, num_instance: Sequence[int]) -> List[List[int]]:
    """`Proof::instances` bytes: 32-byte BIG-endian words, columns concatenated [REF prover/src/proof.rs:77-85,126-138]"""
    assert len(raw) == 32 * sum(num_instance)
    words = [int.from_bytes(raw[i:i + 32], "big") for i in range(0, len(raw), 32)]
    assert all(w < R for w in words)
    out, pos = [], 0
    for m in num_instance:
        out.append(words[pos:pos + m])
        pos += m
    return out


def g2_from_debug_hex(xc0: int, xc1: int, yc0: int, yc1: int):
    """halo2curves prints `Fq2 { c0, c1 }`; oracle/pairing.py keeps FQ2([c0, c1])"""
    pt = (pr.FQ2([xc0, xc1]), pr.FQ2([yc0, yc1]))
    assert pr.is_on_curve(pt, pr.B2), "s_g2 is not on the twist"
    return pt
