#!/usr/bin/env python
"""T1 kit: everything a box with cargo needs to put zkmi355 proofs in front of UPSTREAM halo2's `verify_proof`
(scroll-tech/halo2 @ e5ddf67, the revision the reference pins [REF Cargo.lock:2214-2216]) -- without the in-fork shim.

    python tools/t1_kit.py make  <dir> [--gpu]     circuits as text (desc.txt), params{k} (RawBytes), instances, our vk commitments;
                                                   a self-check proof under OUR stand-in vk.transcript_repr
    cargo run --release -- repr <dir>              (shim/t1_standalone) keygen_vk per case with upstream halo2 -> vk_repr.hex
    python tools/t1_kit.py prove <dir> [--gpu]     proofs (SHPLONK and GWC, Blake2b) under upstream's vk.transcript_repr
    cargo run --release -- verify <dir>            upstream verify_proof on every proof: THE T1 statement
    python tools/t1_kit.py check <dir>             reads ONLY the files back (desc.txt -> circuit, params.bin -> [s]G2, proofs,
                                                   instances) and verifies with oracle/plonk_verifier.py

--gpu proves through libzkmi355.so on an MI355X; without it the oracle's big-int prover stands in (the GPU session is
byte-equal to it, tests/test_gpu_proof.py), so the kit can be assembled and self-checked anywhere.
This tool is test infrastructure (it imports oracle/)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

S_SECRET = 0x5EC2E7
SEED = bytes(range(16))
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


def two_phase_case(k=6):
    """columns a, b first phase; c = a + r b and d = c r2 + a a b second phase, r / r2 squeezed after phase 0
    (the SuperCircuit's phase structure [REF zkevm-circuits/src/util.rs:120-133])"""
    import random
    from zkevm_circuits_amd import plonk
    circ = plonk.Circuit(k, num_fixed=1, num_advice=4, num_instance=0, blinding_factors=5)
    q, a, b_, c_, d_ = circ.fixed_col(0), circ.advice_col(0), circ.advice_col(1), circ.advice_col(2), circ.advice_col(3)
    circ.advice_phase = [0, 0, 1, 1]
    r = circ.challenge_usable_after(0)
    r2 = circ.challenge_usable_after(0)
    circ.add_gate(q * (a + r * b_ - c_))
    circ.add_gate(q * (c_ * r2 + a * a * b_ - d_))
    circ.enable_equality(plonk.ADVICE, 2)
    rng = random.Random(3)
    n, u = circ.n, circ.u
    av, bv = [0] * n, [0] * n
    for row in range(u):
        circ.fixed[0][row] = 1
        av[row], bv[row] = rng.randrange(R), rng.randrange(R)

    def phase_witness(phase, challenges):
        if phase == 0:
            return {0: av, 1: bv}
        rv, r2v = challenges[0], challenges[1]
        cv = [(av[i] + rv * bv[i]) % R if i < u else 0 for i in range(n)]
        dv = [(cv[i] * r2v + av[i] * av[i] * bv[i]) % R if i < u else 0 for i in range(n)]
        return {2: cv, 3: dv}
    return circ, phase_witness, []


def three_phase_case(k=6):
    """The SuperCircuit's challenge structure [REF zkevm-circuits/src/util.rs:120-133] as the headline bench builds it
    (bench_proof.build_shape, phases=True): `evm_word` and `keccak_input` usable after the first phase, `lookup_input` after the
    second; w = q (a + evm_word b) is a second-phase column, t = q (w + lookup_input b) a third-phase one, next to a first-phase
    multiplication gate, a lookup of (a, b) and copy constraints."""
    import random
    from zkevm_circuits_amd import plonk
    circ = plonk.Circuit(k, num_fixed=4, num_advice=5, num_instance=0, blinding_factors=5)
    q, q_lk, t_a, t_b = (circ.fixed_col(i) for i in range(4))
    a, b_, c_, w_, t_ = (circ.advice_col(i) for i in range(5))
    circ.advice_phase = [0, 0, 0, 1, 2]
    evm_word, keccak_input = circ.challenge_usable_after(0), circ.challenge_usable_after(0)
    lookup_input = circ.challenge_usable_after(1)
    circ.add_gate(q * (a * b_ - c_))
    circ.add_gate(q_lk * (a + evm_word * b_ - w_))
    circ.add_gate(q_lk * (w_ + lookup_input * b_ + keccak_input * 0 - t_))
    circ.add_lookup([q_lk * a, q_lk * b_], [t_a, t_b])
    for col in range(3):
        circ.enable_equality(plonk.ADVICE, col)
    rng = random.Random(7)
    n, u = circ.n, circ.u
    av, bv, cv = [0] * n, [0] * n, [0] * n
    for row in range(u):
        if 0 < row < 16:
            circ.fixed[2][row], circ.fixed[3][row] = row, (row * row + 3) % R
    for row in range(1, u - 1):
        if row % 2:
            circ.fixed[0][row] = 1
            av[row], bv[row] = rng.randrange(R), rng.randrange(R)
            cv[row] = av[row] * bv[row] % R
        else:
            circ.fixed[1][row] = 1
            i = rng.randrange(1, 16)
            av[row], bv[row] = i, (i * i + 3) % R
    av[3] = cv[1]                                           # the product of row 1 feeds row 3
    cv[3] = av[3] * bv[3] % R
    circ.copy((plonk.ADVICE, 2, 1), (plonk.ADVICE, 0, 3))

    def phase_witness(phase, challenges):
        if phase == 0:
            return {0: av, 1: bv, 2: cv}
        sel = circ.fixed[1]
        if phase == 1:
            return {3: [(av[i] + challenges[0] * bv[i]) % R if sel[i] else 0 for i in range(n)]}
        wv = [(av[i] + challenges[0] * bv[i]) % R if sel[i] else 0 for i in range(n)]
        return {4: [(wv[i] + challenges[2] * bv[i]) % R if sel[i] else 0 for i in range(n)]}
    return circ, phase_witness, []


def cases():
    from plonk_fixtures import build_circuit, build_evm_circuit, build_multi_lookup_circuit, build_rotation_circuit

    def static(build):
        circ, adv, inst = build
        return circ, (lambda phase, ch: {i: col for i, col in enumerate(adv)} if phase == 0 else {}), inst
    return {
        "plain_k6": static(build_circuit(6, seed=1, wide=False)),                 # gates with rotations, one lookup, permutation over advice / fixed / instance
        "wide_k7": static(build_circuit(7, seed=2, wide=True)),                   # seven permutation columns at degree 5: three chunks of the grand product
        "rotations_k7": static(build_rotation_circuit(7, seed=1, blinding_factors=15)),   # 13 rotations of one column => 15 blinding rows, as upstream derives them (Keccak-like query pattern)
        "lookup_1x_k6": static(build_multi_lookup_circuit(6, 1, 1, 2, 3)),        # mv-lookup, one input tuple of degree 2
        "lookup_2x_k6": static(build_multi_lookup_circuit(6, 1, 2, 1, 3)),        # two tuples merged into one argument
        "lookup_5x_k7": static(build_multi_lookup_circuit(7, 1, 5, 1, 9)),        # five tuples, degree 9: chunk_lookups splits and packs
        "evm_style_k7": static(build_evm_circuit(7, seed=7, states=6, per_state=16)),   # round 6: the EVM-style shape of the bench -- 113 gates q_usable * q_step * state_selector * (constraint * condition) of degree <= 9 at rotations 0 / 1 / 2, a 4-column lookup
        "two_phase_k6": two_phase_case(6),                                        # second-phase columns behind two challenges
        "three_phase_k6": three_phase_case(6),                                    # the SuperCircuit's three phases and three challenges, with a two-column lookup and a copy constraint
    }


def hexfr(v):
    return f"{v % R:064x}"


def write_case_inputs(d, circ, inst):
    from oracle import params_file, plonk_prover as pp, bn254 as b
    os.makedirs(d, exist_ok=True)
    open(os.path.join(d, "desc.txt"), "w").write(circ.kit_desc())
    g, lag, g2, s_g2 = params_file.setup_with_s(circ.k, S_SECRET)
    open(os.path.join(d, "params.bin"), "wb").write(params_file.write(circ.k, g, lag, params_file.g2_raw_bytes(g2), params_file.g2_raw_bytes(s_g2), params_file.RAW))
    with open(os.path.join(d, "instances.txt"), "w") as f:          # one line per instance column: the values halo2 is handed (&[&[Fr]]), canonical hex
        for col in inst:
            used = max([i + 1 for i, v in enumerate(col) if v % R] + [0])
            f.write(" ".join(hexfr(v) for v in col[:used]) + "\n")
    vk = pp.vk_commitments(circ, pp.Srs(circ.k, S_SECRET))
    with open(os.path.join(d, "expect_vk_commitments.hex"), "w") as f:   # fixed then sigma, compressed as halo2 writes points: Rust compares its keygen_vk with these
        for pt in vk:
            f.write(b.g1_compress(pt).hex() + "\n")
    return vk


def read_instances(d):
    return [[int(x, 16) for x in ln.split()] for ln in open(os.path.join(d, "instances.txt")).read().splitlines()]


def prove(circ, phase_witness, inst, vk_repr, multiopen, gpu, transcript="blake2b"):
    from oracle import plonk_prover as pp
    inst_cols = [list(c) for c in inst]
    if not gpu:
        return pp.create_proof(circ, pp.Srs(circ.k, S_SECRET), [[0] * circ.n for _ in range(circ.A)], inst_cols, vk_repr, SEED, multiopen, transcript=transcript, phase_witness=phase_witness)
    import numpy as np
    import zkevm_circuits_amd as z
    from zkevm_circuits_amd import plonk
    from oracle import cref
    ctx = z.Context(0)
    srs = ctx.srs_setup_with_s(circ.k, cref.fr_const(S_SECRET))
    pk = ctx.pk_create(srs, circ.blob())
    pk.set_transcript_repr(cref.to_mont([vk_repr])[0])
    sess = ctx.proof_session(pk, [plonk.column_to_mont(c) for c in inst_cols], SEED, instance_slices=True)
    sess.set_multiopen(1 if multiopen == "shplonk" else 0)
    if transcript != "blake2b":
        sess.set_transcript_kind({"poseidon": 1, "evm": 2}[transcript])
    challenges = []
    for phase in range(circ.num_phases()):
        cols = phase_witness(phase, challenges)
        ch = sess.advice_phase({i: plonk.column_to_mont(c) for i, c in cols.items()})
        challenges += cref.from_mont(np.asarray(ch).reshape(-1, 4)) if len(ch) else []
    proof = sess.finish()
    pk.destroy()
    srs.destroy()
    ctx.close()
    return proof


def cmd_make(args):
    from oracle import plonk_verifier as pv
    for name, (circ, phase_witness, inst) in cases().items():
        d = os.path.join(args.dir, name)
        vk = write_case_inputs(d, circ, inst)
        inst_vals = read_instances(d)
        repr_ = pv.default_vk_repr(circ, vk)
        for mo in ("shplonk", "gwc"):
            open(os.path.join(d, f"selfcheck_{mo}.bin"), "wb").write(prove(circ, phase_witness, inst_vals, repr_, mo, args.gpu))
        # the same proof under the Poseidon transcript of gen_snark_shplonk [REF prover/src/common/prover/utils.rs:31]: not read by the Rust
        # program (upstream halo2 alone has no Poseidon transcript; snark-verifier's verify_snark_shplonk is its consumer), checked by `check`
        open(os.path.join(d, "selfcheck_poseidon_shplonk.bin"), "wb").write(prove(circ, phase_witness, inst_vals, repr_, "shplonk", args.gpu, transcript="poseidon"))
        open(os.path.join(d, "selfcheck_vk_repr.hex"), "w").write(hexfr(repr_) + "\n")
        print(f"{name}: k = {circ.k}, degree {circ.degree()}, {circ.A} advice / {circ.F} fixed / {len(circ.perm_cols)} permutation columns, {len(circ.lookups)} lookup arguments")
    print(f"kit inputs written to {args.dir}; next: (cd shim/t1_standalone && cargo run --release -- repr {os.path.abspath(args.dir)})")


def cmd_prove(args):
    done = 0
    for name, (circ, phase_witness, inst) in cases().items():
        d = os.path.join(args.dir, name)
        f = os.path.join(d, "vk_repr.hex")
        if not os.path.exists(f):
            print(f"{name}: no vk_repr.hex yet (run the Rust `repr` step first)")
            continue
        repr_ = int(open(f).read().split()[0], 16)          # canonical value, big-endian hex (what `{:?}` of Fr prints, without 0x)
        for mo in ("shplonk", "gwc"):
            open(os.path.join(d, f"proof_{mo}.bin"), "wb").write(prove(circ, phase_witness, read_instances(d), repr_, mo, args.gpu))
        done += 1
        print(f"{name}: proofs written under vk.transcript_repr = 0x{repr_:064x}")
    if not done:
        raise SystemExit(1)


def cmd_check(args):
    """every proof in the kit against the oracle's verifier, reading nothing but the files"""
    from zkevm_circuits_amd import plonk
    from oracle import params_file, plonk_verifier as pv, bn254 as b, pairing
    ok = True
    for name in sorted(os.listdir(args.dir)):
        d = os.path.join(args.dir, name)
        if not os.path.isfile(os.path.join(d, "desc.txt")):
            continue
        circ = plonk.Circuit.from_kit_desc(open(os.path.join(d, "desc.txt")).read())
        k, g, lag, g2_blob, s_g2_blob = params_file.read(open(os.path.join(d, "params.bin"), "rb").read(), params_file.RAW)
        assert k == circ.k
        co = [b.from_mont_bytes(s_g2_blob[i * 32:(i + 1) * 32], b.P_MOD) for i in range(4)]
        s_g2 = (pairing.FQ2([co[0], co[1]]), pairing.FQ2([co[2], co[3]]))
        vk = [params_file.g1_decompress(bytes.fromhex(ln)) for ln in open(os.path.join(d, "expect_vk_commitments.hex")).read().split()]
        inst = read_instances(d)
        for tag, repr_file in (("selfcheck", "selfcheck_vk_repr.hex"), ("proof", "vk_repr.hex")):
            if not os.path.exists(os.path.join(d, repr_file)):
                continue
            repr_ = int(open(os.path.join(d, repr_file)).read().split()[0], 16)
            for mo in ("shplonk", "gwc"):
                pf = os.path.join(d, f"{tag}_{mo}.bin")
                if not os.path.exists(pf):
                    continue
                good = pv.verify(circ, vk, repr_, inst, open(pf, "rb").read(), s_g2, multiopen=mo)
                print(f"{name}: {tag}_{mo}.bin ({os.path.getsize(pf)} B) {'accepted' if good else 'REJECTED'} by the oracle verifier")
                ok &= good
            pf = os.path.join(d, f"{tag}_poseidon_shplonk.bin")
            if os.path.exists(pf):
                good = pv.verify(circ, vk, repr_, inst, open(pf, "rb").read(), s_g2, multiopen="shplonk", transcript="poseidon")
                print(f"{name}: {tag}_poseidon_shplonk.bin ({os.path.getsize(pf)} B) {'accepted' if good else 'REJECTED'} by the oracle verifier (Poseidon transcript)")
                ok &= good
    if not ok:
        raise SystemExit(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cmd", choices=["make", "prove", "check"])
    ap.add_argument("dir")
    ap.add_argument("--gpu", action="store_true", help="prove through libzkmi355.so (MI355X); default: the oracle's big-int prover (byte-equal)")
    args = ap.parse_args()
    {"make": cmd_make, "prove": cmd_prove, "check": cmd_check}[args.cmd](args)


if __name__ == "__main__":
    main()
