"""CPU: the library's host-only wire formats and aggregation arithmetic (csrc/wire.hip,
csrc/host_pairing.hpp) against the oracle and the reference-held vectors.  No GPU needed.

  instances as 32-byte big-endian words      [REF prover/src/proof.rs:77-85,126-138]
  VerifyingKey::write / read                 [REF prover/src/io.rs:97-106]
  accumulators, decider, 88-bit limbs        [REF aggregator/src/core.rs:48-147], [REF aggregator/src/constants.rs:77-82]
  pairing                                    golden G6 [REF bus-mapping/src/evm/opcodes/callop.rs:925-936]
"""
import ctypes
import random
import struct

import numpy as np
import pytest

import zkevm_circuits_amd as z
from oracle import bn254 as b
from oracle import cref, hashes, pairing as pr, params_file

R, P = b.R_MOD, b.P_MOD
lib = z.lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _g2_bytes(pt):
    return np.frombuffer(params_file.g2_raw_bytes(pt), dtype=np.uint64).copy()


def test_instances_big_endian_words():
    rng = random.Random(1)
    vals = [0, 1, R - 1, 1 << 200] + [rng.randrange(R) for _ in range(12)]
    m = cref.to_mont(vals)
    out = np.zeros(32 * len(vals), dtype=np.uint8)
    assert lib().zk_host_instances_encode(_ptr(m), ctypes.c_size_t(len(vals)), _ptr(out)) == 0
    assert bytes(out) == b"".join(v.to_bytes(32, "big") for v in vals)          # serialize_instance: serialize_fr(value) reversed
    back = np.zeros_like(m)
    assert lib().zk_host_instances_decode(_ptr(out), ctypes.c_size_t(len(vals)), _ptr(back)) == 0
    assert np.array_equal(back, m)
    bad = np.frombuffer(R.to_bytes(32, "big"), dtype=np.uint8).copy()            # r itself is not a canonical word
    assert lib().zk_host_instances_decode(_ptr(bad), ctypes.c_size_t(1), _ptr(back)) != 0


@pytest.mark.parametrize("fmt", [0, 1, 2])
def test_g1_serde_formats(fmt):
    rng = random.Random(fmt)
    pts = [None] + [b.g1_mul(b.G1_GEN, rng.randrange(1, R)) for _ in range(9)]
    m = cref.affine_to_mont(pts)
    plen = 32 if fmt == 0 else 64
    out = np.zeros(plen * len(pts), dtype=np.uint8)
    assert lib().zk_host_g1_encode(_ptr(m), ctypes.c_size_t(len(pts)), fmt, _ptr(out)) == 0
    want = b"".join(b.g1_compress(pt) if fmt == 0 else b.g1_affine_bytes_raw(pt) for pt in pts)
    assert bytes(out) == want
    back = np.zeros_like(m)
    assert lib().zk_host_g1_decode(_ptr(out), ctypes.c_size_t(len(pts)), fmt, _ptr(back)) == 0
    assert np.array_equal(back, m)
    # a point off the curve is refused by the checked formats
    broken = bytearray(want[plen:2 * plen])
    broken[0] ^= 1
    arr = np.frombuffer(bytes(broken), dtype=np.uint8).copy()
    rc = lib().zk_host_g1_decode(_ptr(arr), ctypes.c_size_t(1), fmt, _ptr(back))
    assert (rc == 0) == (fmt == 2) or fmt == 0        # compressed: another x may still be on the curve
    if fmt == 1:
        assert rc != 0


def test_vk_write_and_read():
    rng = random.Random(7)
    k, nf, npm, nsel = 6, 3, 4, 2
    fixed = [b.g1_mul(b.G1_GEN, rng.randrange(1, R)) for _ in range(nf)]
    perm = [b.g1_mul(b.G1_GEN, rng.randrange(1, R)) for _ in range(npm)]
    sel = bytes(rng.randrange(256) for _ in range(nsel * (1 << k) // 8))
    fm, pm = cref.affine_to_mont(fixed), cref.affine_to_mont(perm)
    sel_a = np.frombuffer(sel, dtype=np.uint8).copy()
    for fmt in (0, 1):
        n = ctypes.c_size_t()
        buf = np.zeros(4096, dtype=np.uint8)
        assert lib().zk_host_vk_write(k, _ptr(fm), nf, _ptr(pm), npm, _ptr(sel_a), nsel, fmt, _ptr(buf), ctypes.c_size_t(buf.size), ctypes.byref(n)) == 0
        enc = (lambda pt: b.g1_compress(pt)) if fmt == 0 else (lambda pt: b.g1_affine_bytes_raw(pt))
        want = struct.pack(">II", k, nf) + b"".join(enc(pt) for pt in fixed + perm) + sel
        assert bytes(buf[:n.value]) == want
        k2, nf2 = ctypes.c_uint32(), ctypes.c_uint32()
        f2, p2, s2 = np.zeros_like(fm), np.zeros_like(pm), np.zeros_like(sel_a)
        assert lib().zk_host_vk_read(_ptr(buf), ctypes.c_size_t(n.value), fmt, npm, nsel, ctypes.byref(k2), ctypes.byref(nf2), _ptr(f2), ctypes.c_size_t(nf), _ptr(p2), _ptr(s2)) == 0
        assert (k2.value, nf2.value) == (k, nf) and np.array_equal(f2, fm) and np.array_equal(p2, pm) and bytes(s2) == sel
        assert lib().zk_host_vk_read(_ptr(buf), ctypes.c_size_t(n.value - 1), fmt, npm, nsel, ctypes.byref(k2), ctypes.byref(nf2), _ptr(f2), ctypes.c_size_t(nf), _ptr(p2), _ptr(s2)) != 0
    small = ctypes.c_size_t()
    assert lib().zk_host_vk_write(k, _ptr(fm), nf, _ptr(pm), npm, _ptr(sel_a), nsel, 0, _ptr(buf), ctypes.c_size_t(8), ctypes.byref(small)) != 0 and small.value > 8


def _check(pairs):
    g1 = cref.affine_to_mont([p_ for p_, _ in pairs]) if pairs else np.zeros((1, 8), dtype=np.uint64)
    g2 = np.concatenate([_g2_bytes(q_) for _, q_ in pairs] + [np.zeros(0, dtype=np.uint64)])
    ok = ctypes.c_int(-1)
    assert lib().zk_host_pairing_check(_ptr(g1), _ptr(g2), ctypes.c_size_t(len(pairs)), ctypes.byref(ok)) == 0
    return bool(ok.value)


def test_pairing_golden_g6_and_bilinearity():
    # golden G6: the ecPairing call data the reference holds [REF bus-mapping/src/evm/opcodes/callop.rs:925-936]
    pushed = ["23a8eb0b0996252cb548a4487da97b02422ebc0e834613f954de6c7e0afdc1fc", "2a23af9a5ce2ba2796c1f4e453a370eb0af8c212d9dc9acd8fc02c2e907baea2",
              "091058a3141822985733cbdddfed0fd8d6c104e9e9eff40bf5abfef9ab163bc7", "1971ff0471b09fa93caaf13cbf443c1aede09cc4328f5a62aad45f40ec133eb4",
              "30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd45", "0000000000000000000000000000000000000000000000000000000000000001",
              "2fe02e47887507adf0ff1743cbac6ba291e66f59be6bd763950bb16041a0a85e", "2bd368e28381e8eccb5fa81fc26cf3f048eea9abfdd85d7ed3ab3698d63e4f90",
              "22606845ff186793914e03e21df544c34ffe2f2f3504de8a79d9159eca2d98d9", "1fb19bb476f6b9e44e2a32234da8212f61cd63919354bc06aef31e3cfaff3ebc",
              "2c0f001f52110ccfe69108924926e45f0b0c868df0e7bde1fe16d3242dc715f6", "2cf44499d5d27bb186308b7af7af02ac5bc9eeb6a3d147c186b21fb1b76e18da"]
    ws = [int(x, 16) for x in reversed(pushed)]
    pairs = []
    for i in range(2):
        x1, y1, x2i, x2r, y2i, y2r = ws[6 * i:6 * i + 6]
        pairs.append(((x1, y1), (pr.FQ2([x2r, x2i]), pr.FQ2([y2r, y2i]))))
    assert _check(pairs)
    broken = [(b.g1_add(pairs[0][0], b.G1_GEN), pairs[0][1]), pairs[1]]
    assert not _check(broken)
    # bilinearity: e(a P, b Q) * e(-(a b) P, Q) == 1, and the identity contributes nothing
    rng = random.Random(3)
    a_, b_ = rng.randrange(1, R), rng.randrange(1, R)
    Pa, Qb = b.g1_mul(b.G1_GEN, a_), pr.ec_mul(pr.G2_GEN, b_)
    assert _check([(Pa, Qb), (b.g1_neg(b.g1_mul(b.G1_GEN, a_ * b_ % R)), pr.G2_GEN)])
    assert _check([(Pa, Qb), (b.g1_neg(b.g1_mul(b.G1_GEN, a_ * b_ % R)), pr.G2_GEN), (None, pr.G2_GEN)])
    assert not _check([(Pa, Qb), (b.g1_neg(b.g1_mul(b.G1_GEN, (a_ * b_ + 1) % R)), pr.G2_GEN)])
    assert _check([])
    # the reference's own valid case [REF zkevm-circuits/src/ecc_circuit/test.rs:239-266]: alpha = 0x102030, beta = 0x413121
    alpha, beta = 0x102030, 0x413121
    p_neg, q_b, s_ab = b.g1_neg(b.g1_mul(b.G1_GEN, alpha)), pr.ec_mul(pr.G2_GEN, beta), b.g1_mul(b.G1_GEN, alpha * beta)
    assert _check([(p_neg, q_b), (s_ab, pr.G2_GEN)]) and not _check([(b.g1_neg(p_neg), q_b), (s_ab, pr.G2_GEN)])


def test_accumulate_decide_and_limbs():
    """extract_accumulators_and_proof's host arithmetic: accumulators (lhs_i, rhs_i) with e(lhs_i, g2) ==
    e(rhs_i, s g2), i.e. lhs_i = s * rhs_i; their Poseidon-challenge combination must satisfy the same."""
    rng = random.Random(11)
    s = 0x1234567
    rhs = [b.g1_mul(b.G1_GEN, rng.randrange(1, R)) for _ in range(4)]
    lhs = [b.g1_mul(pt, s) for pt in rhs]
    lm, rm = cref.affine_to_mont(lhs), cref.affine_to_mont(rhs)
    lo, ro, r_out = np.zeros(8, dtype=np.uint64), np.zeros(8, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
    assert lib().zk_host_accumulate(_ptr(lm), _ptr(rm), ctypes.c_size_t(4), _ptr(lo), _ptr(ro), _ptr(r_out)) == 0
    sponge = hashes.PoseidonSponge()
    for l_, r_ in zip(lhs, rhs):
        sponge.update([l_[0] % R, l_[1] % R, r_[0] % R, r_[1] % R])
    ch = sponge.squeeze()
    assert int(cref.from_mont(r_out.reshape(1, 4))[0]) == ch
    want_l = want_r = None
    for i, (l_, r_) in enumerate(zip(lhs, rhs)):
        want_l = b.g1_add(want_l, b.g1_mul(l_, pow(ch, i, R)))
        want_r = b.g1_add(want_r, b.g1_mul(r_, pow(ch, i, R)))
    got_l, got_r = cref.affine_from_mont(lo.reshape(1, 8))[0], cref.affine_from_mont(ro.reshape(1, 8))[0]
    assert (got_l, got_r) == (want_l, want_r)
    g2, sg2 = _g2_bytes(pr.G2_GEN), _g2_bytes(pr.ec_mul(pr.G2_GEN, s))
    ok = ctypes.c_int(-1)
    assert lib().zk_host_accumulator_check(_ptr(lo), _ptr(ro), _ptr(g2), _ptr(sg2), ctypes.byref(ok)) == 0 and ok.value == 1
    wrong = _g2_bytes(pr.ec_mul(pr.G2_GEN, s + 1))
    assert lib().zk_host_accumulator_check(_ptr(lo), _ptr(ro), _ptr(g2), _ptr(wrong), ctypes.byref(ok)) == 0 and ok.value == 0
    # 4 coordinates x 3 limbs of 88 bits, least significant limb first [REF aggregator/src/constants.rs:77-82]
    limbs = np.zeros((12, 4), dtype=np.uint64)
    assert lib().zk_host_accumulator_limbs(_ptr(lo), _ptr(ro), _ptr(limbs)) == 0
    want = []
    for coord in (got_l[0], got_l[1], got_r[0], got_r[1]):
        want += [(coord >> (88 * i)) & ((1 << 88) - 1) for i in range(3)]
    assert [int(v) for v in cref.from_mont(limbs)] == want


def _proof_json(proof: bytes, inst: bytes, vk: bytes, git):
    n = ctypes.c_size_t()
    g = None if git is None else git.encode()
    assert lib().zk_host_proof_json_write(proof, ctypes.c_size_t(len(proof)), inst, ctypes.c_size_t(len(inst)), vk, ctypes.c_size_t(len(vk)), g, None, ctypes.c_size_t(0), ctypes.byref(n)) == 0
    buf = ctypes.create_string_buffer(n.value)
    assert lib().zk_host_proof_json_write(proof, ctypes.c_size_t(len(proof)), inst, ctypes.c_size_t(len(inst)), vk, ctypes.c_size_t(len(vk)), g, buf, ctypes.c_size_t(n.value), ctypes.byref(n)) == 0
    return buf.raw[:n.value]


def _proof_json_read(js: bytes):
    lens = [ctypes.c_size_t(0) for _ in range(3)]
    has = ctypes.c_int(-1)
    rc = lib().zk_host_proof_json_read(js, ctypes.c_size_t(len(js)), None, ctypes.byref(lens[0]), None, ctypes.byref(lens[1]), None, ctypes.byref(lens[2]), None, ctypes.c_size_t(0), ctypes.byref(has))
    if rc:
        return rc
    bufs = [ctypes.create_string_buffer(max(ln.value, 1)) for ln in lens]
    caps = [ctypes.c_size_t(ln.value) for ln in lens]
    git = ctypes.create_string_buffer(64)
    assert lib().zk_host_proof_json_read(js, ctypes.c_size_t(len(js)), bufs[0], ctypes.byref(caps[0]), bufs[1], ctypes.byref(caps[1]), bufs[2], ctypes.byref(caps[2]),
                                         git, ctypes.c_size_t(64), ctypes.byref(has)) == 0
    return tuple(bf.raw[:c.value] for bf, c in zip(bufs, caps)) + ((git.value.decode() if has.value else None),)


def test_proof_wire_object_is_serde_jsons():
    """`Proof` [REF prover/src/proof.rs:25-35] as dump_as_json writes it: compact serde_json, struct field order, base64 of the
    `base64` crate (standard alphabet, padded) [REF eth-types/src/lib.rs:71-91] -- byte for byte what Python's json / base64 give
    for the same object, for every padding length, an empty vk (`Proof::new` without a pk) and no git version."""
    import base64
    import json
    rng = random.Random(7)
    for plen in (0, 1, 2, 3, 4, 1630, 1631, 1632):
        proof = bytes(rng.randrange(256) for _ in range(plen))
        inst = b"".join(rng.randrange(R).to_bytes(32, "big") for _ in range(plen % 5))
        vk = bytes(rng.randrange(256) for _ in range((plen * 7) % 11))
        for git in ("a1b2c3d", None, 'odd"ver\\sion'):
            got = _proof_json(proof, inst, vk, git)
            want = json.dumps({"proof": base64.b64encode(proof).decode(), "instances": base64.b64encode(inst).decode(), "vk": base64.b64encode(vk).decode(),
                               "git_version": git}, separators=(",", ":")).encode()
            assert got == want
            assert _proof_json_read(got) == (proof, inst, vk, git)
            # pretty-printed, keys in another order: what a hand-edited or re-serialised file looks like
            obj = json.loads(got)
            pretty = json.dumps({k_: obj[k_] for k_ in ("vk", "git_version", "proof", "instances")}, indent=2).encode()
            assert _proof_json_read(pretty) == (proof, inst, vk, git)
    # `git_version` is an Option: absent is read as None
    js = json.dumps({"proof": "", "instances": "", "vk": ""}).encode()
    assert _proof_json_read(js) == (b"", b"", b"", None)


def test_proof_wire_object_refuses_malformed_input():
    import base64
    import json
    good = {"proof": base64.b64encode(b"abc").decode(), "instances": base64.b64encode(bytes(32)).decode(), "vk": "", "git_version": None}
    assert _proof_json_read(json.dumps(good).encode())[0] == b"abc"
    bad = [
        dict(good, proof="YWJj="),                       # length not a multiple of four
        dict(good, proof="YW=j"),                        # padding in the middle
        dict(good, proof="YWJ*"),                        # a character outside the alphabet
        dict(good, proof="YR=="),                        # non-canonical trailing bits ("a" is YQ==)
        dict(good, instances=base64.b64encode(bytes(31)).decode()),     # not whole 32-byte words
        {k_: v for k_, v in good.items() if k_ != "vk"},                # a missing field
    ]
    for obj in bad:
        assert _proof_json_read(json.dumps(obj).encode()) == -1, obj          # ZK_ERR_INVALID_ARG
    # keys the struct does not have are parsed and dropped, as serde does without deny_unknown_fields (the reference flattens `Proof`
    # into ChunkProof / BatchProof [REF prover/src/proof/chunk.rs:10-19]) -- whatever their value; a malformed value is still refused
    for extra in ("x", 1, -2.5e3, None, True, [1, [2, {"a": "\u00e9\n"}]], {"k": {"l": []}}):
        assert _proof_json_read(json.dumps(dict(good, extra=extra)).encode())[0] == b"abc", extra
    for tail in (b'"extra":[1,}', b'"extra":"\\x"', b'"extra":01', b'"extra":tru', b'"extra":"\\u00g0"'):
        assert _proof_json_read(b'{"proof":"","instances":"","vk":"",' + tail + b'}') == -1, tail
    assert _proof_json_read(b'{"proof":"","instances":"","vk":"","git_version":"a\\b\\f\\u0041"}')[3] == "a\b\fA"     # serde_json's short escapes
    assert _proof_json_read(b'{"proof":"","instances":"","vk":"","git_version":"\\u00zz"}') == -1             # not hex digits
    assert _proof_json_read(b'{"proof":"","instances":"","vk":"","proof":""}') == -1      # a repeated key
    assert _proof_json_read(b'{"proof":"","instances":"","vk":""} x') == -1               # trailing garbage
    n = ctypes.c_size_t()
    assert lib().zk_host_proof_json_write(b"", ctypes.c_size_t(0), bytes(31), ctypes.c_size_t(31), b"", ctypes.c_size_t(0), None, None, ctypes.c_size_t(0), ctypes.byref(n)) == -1


def test_instances_json_matrix_is_serde_jsons():
    """`serialize_instance` [REF prover/src/io.rs:52-56]: serde_json of Vec<Vec<Vec<u8>>> -- the same bytes as Python's json of the
    little-endian byte lists; reading brings the Montgomery values back, column lengths included (an empty column, no column)."""
    import json
    rng = random.Random(3)
    for shape in ([4], [1, 0, 3], [], [0]):
        cols = [[rng.choice([0, 1, R - 1, rng.randrange(R)]) for _ in range(ln)] for ln in shape]
        monts = [cref.to_mont(c) if c else np.zeros((0, 4), dtype=np.uint64) for c in cols]
        ptrs = (ctypes.c_void_p * max(len(cols), 1))(*[m.ctypes.data for m in monts])
        lens = (ctypes.c_size_t * max(len(cols), 1))(*[len(c) for c in cols])
        n = ctypes.c_size_t()
        assert lib().zk_host_instances_json_write(ptrs, lens, ctypes.c_size_t(len(cols)), None, ctypes.c_size_t(0), ctypes.byref(n)) == 0
        buf = ctypes.create_string_buffer(max(n.value, 1))
        assert lib().zk_host_instances_json_write(ptrs, lens, ctypes.c_size_t(len(cols)), buf, ctypes.c_size_t(n.value), ctypes.byref(n)) == 0
        got = buf.raw[:n.value]
        want = json.dumps([[list(v.to_bytes(32, "little")) for v in c] for c in cols], separators=(",", ":")).encode()
        assert got == want
        for text in (got, json.dumps(json.loads(got), indent=1).encode()):
            ncols, total = ctypes.c_size_t(), ctypes.c_size_t()
            assert lib().zk_host_instances_json_read(text, ctypes.c_size_t(len(text)), ctypes.byref(ncols), None, ctypes.c_size_t(0), None, ctypes.c_size_t(0), ctypes.byref(total)) == 0
            assert ncols.value == len(cols) and total.value == sum(shape)
            lens_out = (ctypes.c_size_t * max(ncols.value, 1))()
            vals = np.zeros((max(total.value, 1), 4), dtype=np.uint64)
            assert lib().zk_host_instances_json_read(text, ctypes.c_size_t(len(text)), ctypes.byref(ncols), lens_out, ctypes.c_size_t(ncols.value), _ptr(vals), ctypes.c_size_t(total.value),
                                                     ctypes.byref(total)) == 0
            assert list(lens_out)[:ncols.value] == shape
            assert [int(v) for v in cref.from_mont(vals[:total.value])] == [v for c in cols for v in c]
    for bad in (b"[[[1,2,3]]]", b"[[[" + b",".join([b"256"] + [b"0"] * 31) + b"]]]", b"[[[" + b",".join([b"01"] + [b"0"] * 31) + b"]]]", json.dumps([[list(R.to_bytes(32, "little"))]]).encode(), b"[[[", b"[] x"):
        ncols, total = ctypes.c_size_t(), ctypes.c_size_t()
        assert lib().zk_host_instances_json_read(bad, ctypes.c_size_t(len(bad)), ctypes.byref(ncols), None, ctypes.c_size_t(0), None, ctypes.c_size_t(0), ctypes.byref(total)) == -1, bad
