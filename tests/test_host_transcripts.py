"""CPU: the library's host-side transcripts (csrc/host_hash.hpp behind zk_transcript_* /
zk_proof_set_transcript_kind) against the oracle's restatement (oracle/transcripts.py) and the
known-answer vectors that pin both.  No GPU: these entry points are host-only.

  Blake2b  [REF circuit-benchmarks/src/super_circuit.rs:112]       create_proof of the benches
  Poseidon [REF aggregator/src/core.rs:57-58,91-92]                  gen_snark_shplonk
  EVM      [REF prover/src/common/prover/evm.rs:67]                  gen_evm_proof_shplonk
"""
import random

import numpy as np
import pytest

import zkevm_circuits_amd as z
from oracle import bn254 as b
from oracle import cref, hashes, transcripts

# keccak256("") as the reference holds it [REF eth-types/src/lib.rs:274]
KECCAK_EMPTY = bytes.fromhex("c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470")
# Poseidon reference implementation, poseidonperm_x5_254_5: permutation of (0, 1, 2, 3, 4)
POSEIDON_KAT = [0x299c867db6c1fdd79dcefa40e4510b9837e60ebb1ce0663dbaa525df65250465, 0x1148aaef609aa338b27dafd89bb98862d8bb2b429aceac47d86206154ffe053d,
                0x24febb87fed7462e23f6665ff9a0111f4044c38ee1672c1ac6b0637d34f24907, 0x0eb08f6d809668a981c186beaf6110060707059576406b248e5d9cf6e78b3d3e,
                0x07748bc6877c9b82c8b98666ee9d0626ec7f5be4205f79ee8528ef1c4a376fc7]


def _golden(name):
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name)))


def test_keccak256():
    assert KECCAK_EMPTY.hex() == _golden("reference_vectors.json")["keccak256_empty"]["value"]
    assert hashes.keccak256(b"") == KECCAK_EMPTY == z.binding.host_keccak256(b"")
    rng = random.Random(5)
    for ln in (1, 31, 32, 33, 64, 135, 136, 137, 271, 272, 273, 1000):
        data = bytes(rng.randrange(256) for _ in range(ln))
        assert z.binding.host_keccak256(data) == hashes.keccak256(data)


def test_poseidon_generator_is_pinned_by_the_reference():
    """POSEIDON_CODE_HASH_EMPTY [REF eth-types/src/lib.rs:278] = first word of the width-3 permutation of (0, 0, 0): a vector the
    reference itself holds, reproduced by the oracle's and by the product's constant generation."""
    v = _golden("reference_vectors.json")["G7_poseidon_code_hash_empty"]
    assert v["ref"] == "eth-types/src/lib.rs:278"
    want = int(v["value"], 16)
    assert hashes.poseidon_spec(3, 8, 57).permute([0, 0, 0])[0] == want
    got = z.binding.host_poseidon_permute_width3(cref.to_mont([0, 0, 0]))
    assert int(cref.from_mont(got)[0]) == want
    rng = random.Random(17)
    st = [rng.randrange(b.R_MOD) for _ in range(3)]
    assert [int(x) for x in cref.from_mont(z.binding.host_poseidon_permute_width3(cref.to_mont(st)))] == hashes.poseidon_spec(3, 8, 57).permute(st)


def test_poseidon_permutation():
    kat = _golden("published_vectors.json")["poseidonperm_x5_254_5"]
    assert kat["input"] == [0, 1, 2, 3, 4] and [int(v, 16) for v in kat["output"]] == POSEIDON_KAT
    assert hashes.poseidon_spec().permute([0, 1, 2, 3, 4]) == POSEIDON_KAT
    got = z.binding.host_poseidon_permute(cref.to_mont([0, 1, 2, 3, 4]))
    assert [int(v) for v in cref.from_mont(got)] == POSEIDON_KAT
    rng = random.Random(9)
    st = [rng.randrange(b.R_MOD) for _ in range(5)]
    assert [int(v) for v in cref.from_mont(z.binding.host_poseidon_permute(cref.to_mont(st)))] == hashes.poseidon_spec().permute(st)


@pytest.mark.parametrize("kind,name", [(0, "blake2b"), (1, "poseidon"), (2, "evm")])
def test_transcripts_agree_with_the_oracle(kind, name):
    rng = random.Random(100 + kind)
    t = z.binding.HostTranscript(kind)
    o = transcripts.make(name)
    pts = [b.g1_mul(b.G1_GEN, rng.randrange(1, b.R_MOD)) for _ in range(6)]
    for step in range(60):
        op = rng.choice(["cp", "cs", "wp", "ws", "sq", "sq"])
        if op in ("cp", "wp"):
            pt = rng.choice(pts)
            raw = cref.affine_to_mont([pt]).tobytes()
            (t.common_point if op == "cp" else t.write_point)(raw)
            (o.common_point if op == "cp" else o.write_point)(pt)
        elif op in ("cs", "ws"):
            s = rng.randrange(b.R_MOD) if rng.random() < 0.8 else rng.randrange(4)
            raw = cref.to_mont([s]).tobytes()
            (t.common_scalar if op == "cs" else t.write_scalar)(raw)
            (o.common_scalar if op == "cs" else o.write_scalar)(s)
        else:
            got = int(cref.from_mont(np.frombuffer(t.squeeze_challenge(), dtype=np.uint64).reshape(1, 4))[0])
            assert got == o.squeeze(), f"challenge {step} differs"
    assert t.proof() == bytes(o.proof)
    # what was written reads back on the verifier side of the oracle
    rd = transcripts.make(name, bytes(o.proof))
    assert len(o.proof) > 0 and rd._take(len(o.proof)) == bytes(o.proof)
    t.close()


def test_identity_point_handling():
    ident = bytes(64)
    t = z.binding.HostTranscript(0)
    t.write_point(ident)                       # Blake2b: 64 zero bytes absorbed; written as halo2curves' identity image (bit 255 on a zero x)
    assert t.proof() == bytes(31) + b"\x80"
    for kind in (1, 2):                        # snark-verifier: the identity has no coordinates -> Error::Transcript
        with pytest.raises(z.ZkError):
            z.binding.HostTranscript(kind).write_point(ident)
    with pytest.raises(z.ZkError):
        z.binding.HostTranscript(7)
