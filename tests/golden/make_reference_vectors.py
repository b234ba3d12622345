#!/usr/bin/env python
"""Extracts the golden vectors the reference's own sources hold for the hot path (G3-G6 and keccak256(""))
from /root/reference and writes tests/golden/reference_vectors.json.  Runs only where /root/reference exists (the build
container); the tests read the committed JSON.  Nothing is copied but the constants themselves, each with the file and
line it was found on.

usage: python tests/golden/make_reference_vectors.py [/root/reference]
"""
import base64
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def find(path, pattern, nth=0):
    """(match, 1-based line) of the nth occurrence of `pattern` in a reference file"""
    text = open(os.path.join(REF, path)).read()
    ms = list(re.finditer(pattern, text))
    if len(ms) <= nth:
        raise SystemExit(f"{path}: pattern {pattern!r} not found ({len(ms)} matches)")
    m = ms[nth]
    return m, text.count("\n", 0, m.start()) + 1


out = {}
m, ln = find("zkevm-circuits/src/super_circuit.rs", r"Value \{ inner: Some\((0x[0-9a-f]{64})\) \}")
out["G3_mockprover_third_challenge"] = {"value": m.group(1), "ref": f"zkevm-circuits/src/super_circuit.rs:{ln}",
                                        "pins": "Fr::from_uniform_bytes (64-byte little-endian mod r) and halo2's MockProver challenge derivation"}
m, ln = find("zkevm-circuits/src/ecc_circuit/test.rs", r'word!\("(0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd45)"\)')
out["G4_fq_modulus_minus_two"] = {"value": m.group(1), "ref": f"zkevm-circuits/src/ecc_circuit/test.rs:{ln}", "pins": "the Fq modulus; (1, p - 2) = -G"}
callop = "bus-mapping/src/evm/opcodes/callop.rs"
mx, lx = find(callop, r'word!\("(30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd3)"\)')
my, ly = find(callop, r'word!\("(15ed738c0e0a7c92e7845f96b2ae9c0a68a6a449e3538fc7ff3ebf7a5a18a2c4)"\)')
n_add = len(re.findall(r'word!\("30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd3"\)', open(os.path.join(REF, callop)).read()))
out["G5_two_G"] = {"x": "0x0" + mx.group(1), "y": "0x" + my.group(1), "ref": f"{callop}:{lx},{ly}", "occurrences": n_add,
                   "pins": "G1 group law: ecAdd((1,2),(1,2)) and ecMul((1,2),2) return this point"}
text = open(os.path.join(REF, callop)).read()
start = text.index('name: "ecPairing"')
block = text[start:text.index("PUSH1(12)", start)]
words = re.findall(r'PUSH32\(word!\("0x([0-9a-f]{64})"\)\)', block)
assert len(words) == 12, len(words)
out["G6_ecpairing_pushed_words"] = {"words": words, "ref": f"{callop}:{text.count(chr(10), 0, start) + 1}",
                                    "pins": "Fq2 / Fq12 tower, G2, the ate pairing: two pairs whose product is one (words are PUSHed last-first)"}
m, ln = find("eth-types/src/lib.rs", r'Hash::from_str\("0x(c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470)"\)')
out["keccak256_empty"] = {"value": m.group(1), "ref": f"eth-types/src/lib.rs:{ln}", "pins": "Keccak-256 (the EVM transcript's hash)"}
m, ln = find("eth-types/src/lib.rs", r'Hash::from_str\("0x(2098f5fb9e239eab3ceac3f27b81e481dc3124d55ffed523a839ee8446b64864)"\)')
out["G7_poseidon_code_hash_empty"] = {"value": m.group(1), "ref": f"eth-types/src/lib.rs:{ln}",
                                      "pins": "the Poseidon constant generation (Grain LFSR, Cauchy MDS) and round schedule: the value is permute(0, 0, 0)[0] at T = 3, R_F = 8, "
                                              "R_P = 57 -- the transcript's POSEIDON_SPEC is the same generator at T = 5, R_P = 60"}
with open(os.path.join(HERE, "reference_vectors.json"), "w") as f:
    json.dump(out, f, indent=1)
    f.write("\n")
print("wrote", os.path.join(HERE, "reference_vectors.json"), "with", len(out), "vectors")

# ---- the reference-produced proof of the hot path: a production ChunkProof (layer-2 compression snark, k = 25, SHPLONK,
# Poseidon transcript) held as test data for the batch prover, and s*G2 of the SRS it was made with.
task = "aggregator/data/batch-task.json"
cp = json.load(open(os.path.join(REF, task)))["chunk_proofs"][0]
m, ln = find("prover/src/utils.rs", r'PARAMS_G2_SECRET_POWER: &str = "\(Fq2 \{ c0: (0x[0-9a-f]{64}), c1: (0x[0-9a-f]{64}) \}, Fq2 \{ c0: (0x[0-9a-f]{64}), c1: (0x[0-9a-f]{64}) \}\)"')
chunk = {
    "ref": f"{task}: chunk_proofs[0] (struct prover/src/proof/chunk.rs:10-19 flattening prover/src/proof.rs:25-35)",
    "pins": "compressed G1 encoding, vk Processed layout, BE instance words, PoseidonTranscript<NativeLoader> framing, evaluation and query "
            "order, blinding-row / l_last conventions, the logUp identity, single-chunk permutation, SHPLONK sets / powers / normalisation, "
            "accumulator limbs + decider",
    "protocol": json.loads(base64.b64decode(cp["protocol"])),      # snark-verifier PlonkProtocol, serde_json image
    "proof": cp["proof"], "instances": cp["instances"], "vk": cp["vk"],          # base64, as the reference's `Proof` holds them
    "git_version": cp["git_version"],
    # the flattened ChunkProof object with the bulky out-of-path members dropped: what zk_host_proof_json_read must accept (serde ignores
    # unknown keys; `proof`, `instances`, `vk`, `git_version` sit beside `protocol`, `chunk_info`, `row_usages`)
    "flattened_object": json.dumps({"protocol": cp["protocol"], "proof": cp["proof"], "instances": cp["instances"], "vk": cp["vk"],
                                    "chunk_info": {k_: v for k_, v in cp["chunk_info"].items() if k_ != "tx_bytes"},
                                    "git_version": cp["git_version"], "row_usages": cp["row_usages"]}, separators=(",", ":")),
    "s_g2": {"x_c0": m.group(1), "x_c1": m.group(2), "y_c0": m.group(3), "y_c1": m.group(4), "ref": f"prover/src/utils.rs:{ln}",
             "pins": "s*G2 of Scroll's production SRS (Debug string of a G2Affine), checked by load_params at prover/src/utils.rs:78"},
}
with open(os.path.join(HERE, "reference_chunk_proof.json"), "w") as f:
    json.dump(chunk, f, indent=1)
    f.write("\n")
print("wrote", os.path.join(HERE, "reference_chunk_proof.json"))
