"""CPU oracle (TEST INFRASTRUCTURE ONLY) -- the three transcripts the reference's call sites plug
into `create_proof` / `verify_proof`, prover side (`*Write`) and verifier side (`*Read`):

  Blake2b   halo2_proofs::transcript::{Blake2bWrite, Blake2bRead} + Challenge255
            [REF circuit-benchmarks/src/super_circuit.rs:112,138]
  Poseidon  snark-verifier `system::halo2::transcript::halo2::PoseidonTranscript<G1Affine, NativeLoader, _>`
            with POSEIDON_SPEC (T 5, RATE 4, R_F 8, R_P 60) [REF aggregator/src/core.rs:57-58,91-92],
            [REF aggregator/src/recursion/util.rs:97] -- what gen_snark_shplonk uses
            [REF prover/src/common/prover/utils.rs:31]
  Evm       snark-verifier `system::halo2::transcript::evm::EvmTranscript<G1Affine, NativeLoader, _, _>`
            (Keccak-256) -- what gen_evm_proof_shplonk uses [REF prover/src/common/prover/evm.rs:67]

Encodings (external crates, restated): Blake2b and Poseidon write a point as halo2curves'
32-byte compressed form and a scalar as its 32-byte little-endian repr; the EVM transcript writes
x || y as 32-byte big-endian words (64 bytes, identity not representable) and scalars big-endian.
Points are affine int tuples (None = identity), scalars ints mod r.
"""
from __future__ import annotations

import hashlib

from . import bn254 as b
from .hashes import PoseidonSponge, keccak256

R, P = b.R_MOD, b.P_MOD


def decompress_g1(raw: bytes):
    """halo2curves `G1Affine::from_bytes` (bn254.g1_decompress), failing with AssertionError as the readers here do"""
    try:
        return b.g1_decompress(raw)
    except ValueError as e:
        raise AssertionError(str(e))


class _Base:
    """write_* = common_* + append to the proof; read_* = take from the proof + common_*"""
    point_len, scalar_len = 32, 32

    def __init__(self, proof: bytes = b""):
        self.proof = bytearray(proof)
        self.pos = 0

    # encodings (compressed LE by default)
    def _enc_point(self, pt) -> bytes: return b.g1_compress(pt)
    def _dec_point(self, raw: bytes): return decompress_g1(raw)
    def _enc_scalar(self, s: int) -> bytes: return (s % R).to_bytes(32, "little")
    def _dec_scalar(self, raw: bytes) -> int: return int.from_bytes(raw, "little")

    def write_point(self, pt):
        self.common_point(pt)
        self.proof += self._enc_point(pt)

    def write_scalar(self, s: int):
        self.common_scalar(s)
        self.proof += self._enc_scalar(s)

    def _take(self, n: int) -> bytes:
        raw = bytes(self.proof[self.pos:self.pos + n])
        assert len(raw) == n, "proof truncated"
        self.pos += n
        return raw

    def read_point(self):
        pt = self._dec_point(self._take(self.point_len))
        self.common_point(pt)
        return pt

    def read_scalar(self) -> int:
        s = self._dec_scalar(self._take(self.scalar_len))
        assert s < R, "non-canonical scalar"
        self.common_scalar(s)
        return s

    def exhausted(self) -> bool:
        return self.pos == len(self.proof)


class Blake2b(_Base):
    def __init__(self, proof: bytes = b""):
        super().__init__(proof)
        self.h = hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript")

    def common_point(self, pt):
        self.h.update(b"\x01" + (bytes(64) if pt is None else pt[0].to_bytes(32, "little") + pt[1].to_bytes(32, "little")))

    def common_scalar(self, s: int):
        self.h.update(b"\x02" + (s % R).to_bytes(32, "little"))

    def squeeze(self) -> int:
        self.h.update(b"\x00")
        return b.fr_from_uniform_bytes(self.h.copy().digest())


class Poseidon(_Base):
    """state stays on the host; coordinates enter as Fq integers reduced mod r (`fe_to_fe`)"""

    def __init__(self, proof: bytes = b""):
        super().__init__(proof)
        self.sponge = PoseidonSponge()

    def common_point(self, pt):
        assert pt is not None, "the identity has no coordinates to absorb (snark-verifier: Error::Transcript)"
        self.sponge.update([pt[0] % R, pt[1] % R])

    def common_scalar(self, s: int):
        self.sponge.update([s % R])

    def squeeze(self) -> int:
        return self.sponge.squeeze()


class Evm(_Base):
    point_len = 64

    def __init__(self, proof: bytes = b""):
        super().__init__(proof)
        self.buf = bytearray()

    def _enc_point(self, pt) -> bytes:
        assert pt is not None
        return pt[0].to_bytes(32, "big") + pt[1].to_bytes(32, "big")

    def _dec_point(self, raw: bytes):
        x, y = int.from_bytes(raw[:32], "big"), int.from_bytes(raw[32:], "big")
        assert x < P and y < P and (y * y - x * x * x - 3) % P == 0, "point not on curve"
        return (x, y)

    def _enc_scalar(self, s: int) -> bytes: return (s % R).to_bytes(32, "big")
    def _dec_scalar(self, raw: bytes) -> int: return int.from_bytes(raw, "big")

    def common_point(self, pt):
        assert pt is not None, "the identity has no coordinates to absorb"
        self.buf += pt[0].to_bytes(32, "big") + pt[1].to_bytes(32, "big")

    def common_scalar(self, s: int):
        self.buf += (s % R).to_bytes(32, "big")

    def squeeze(self) -> int:
        data = bytes(self.buf) + (b"\x01" if len(self.buf) == 32 else b"")
        h = keccak256(data)
        self.buf = bytearray(h)
        return int.from_bytes(h, "big") % R


KINDS = {"blake2b": Blake2b, "poseidon": Poseidon, "evm": Evm}


def make(kind: str, proof: bytes = b""):
    return KINDS[kind](proof)
