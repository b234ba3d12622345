"""CPU oracle (TEST INFRASTRUCTURE ONLY) -- the two hash functions behind the reference's non-Blake2b
transcripts, restated from their published specifications.

  * Keccak-256 (original Keccak padding 0x01, as the EVM uses it) -- `EvmTranscript` of
    snark-verifier @ 572ef69 (`system/halo2/transcript/evm.rs`), reached from
    [REF prover/src/common/prover/evm.rs:67] (gen_evm_proof_shplonk) and [REF aggregator/src/tests.rs:118].
    Pinned by the reference-held constant keccak256("") = c5d2460186f7...85a470
    [REF eth-types/src/lib.rs: KECCAK_CODE_HASH_EMPTY / bus-mapping EMPTY hash] (tests/test_oracle_golden.py).
  * Poseidon over BN254 Fr, x^5 S-box, T = 5, RATE = 4, R_F = 8, R_P = 60 -- `POSEIDON_SPEC` /
    `PoseidonTranscript<NativeLoader, _>` of snark-verifier-sdk [REF aggregator/src/core.rs:25-28,57-58],
    [REF prover/src/common/prover/utils.rs:31] (gen_snark_shplonk).  Constants come from the Grain
    LFSR of the Poseidon paper (eprint 2019/458, supplementary material F) as the `poseidon 0.2.0`
    crate @ 5787dd3 derives them [REF Cargo.lock:3402-3404]; the plain (un-optimised) round
    schedule computed here is mathematically the permutation that crate evaluates with its
    pre-/sparse-MDS optimisation.  Pinned by the Poseidon reference implementation's published
    known-answer vector `poseidonperm_x5_254_5` (input [0,1,2,3,4]), which is also that crate's own
    cross-test.
"""
from __future__ import annotations

from typing import List

R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001

# ------------------------------------------------------------------------------------ Keccak-256
_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001,
       0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
       0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
       0x000000000000800A, 0x800000008000000A, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
_M64 = (1 << 64) - 1


def _rol(x: int, n: int) -> int:
    n %= 64
    return ((x << n) | (x >> (64 - n))) & _M64 if n else x


def keccak_f1600(a: List[List[int]]) -> None:
    """a[x][y], 5 x 5 lanes of 64 bits, in place (FIPS 202 section 3.3 round function)."""
    for rc in _RC:
        c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ _rol(c[(x + 1) % 5], 1) for x in range(5)]
        for x in range(5):
            for y in range(5):
                a[x][y] ^= d[x]
        b = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                b[y][(2 * x + 3 * y) % 5] = _rol(a[x][y], _ROT[x][y])
        for x in range(5):
            for y in range(5):
                a[x][y] = b[x][y] ^ (~b[(x + 1) % 5][y] & b[(x + 2) % 5][y] & _M64)
        a[0][0] ^= rc


def keccak256(data: bytes) -> bytes:
    rate = 136
    msg = bytearray(data)
    msg.append(0x01)                       # Keccak (pre-FIPS) domain padding, the EVM's KECCAK256
    while len(msg) % rate:
        msg.append(0)
    msg[-1] |= 0x80
    a = [[0] * 5 for _ in range(5)]
    for off in range(0, len(msg), rate):
        for i in range(rate // 8):
            a[i % 5][i // 5] ^= int.from_bytes(msg[off + 8 * i:off + 8 * i + 8], "little")
        keccak_f1600(a)
    out = b"".join(a[i % 5][i // 5].to_bytes(8, "little") for i in range(4))
    return out


# ------------------------------------------------------------------------------------ Poseidon
class Grain:
    """80-bit Grain LFSR of the Poseidon parameter generation (eprint 2019/458, suppl. F)."""

    def __init__(self, field_bits: int, t: int, r_f: int, r_p: int):
        bits: List[int] = []

        def app(n, v):
            bits.extend((v >> (n - 1 - i)) & 1 for i in range(n))
        app(2, 1)            # prime field
        app(4, 0)            # x^alpha S-box
        app(12, field_bits)
        app(12, t)
        app(10, r_f)
        app(10, r_p)
        app(30, (1 << 30) - 1)
        assert len(bits) == 80
        self.s = bits
        for _ in range(160):
            self._new_bit()

    def _new_bit(self) -> int:
        s = self.s
        nb = s[62] ^ s[51] ^ s[38] ^ s[23] ^ s[13] ^ s[0]
        s.pop(0)
        s.append(nb)
        return nb

    def next_bit(self) -> int:
        while not self._new_bit():          # a 0 discards the following bit
            self._new_bit()
        return self._new_bit()

    def next_bits_int(self, n: int) -> int:
        v = 0
        for _ in range(n):                  # most significant bit first, as the reference implementation
            v = (v << 1) | self.next_bit()
        return v

    def next_field_element(self, field_bits: int) -> int:
        while True:
            v = self.next_bits_int(field_bits)
            if v < R_MOD:
                return v

    def next_field_element_without_rejection(self, field_bits: int) -> int:
        return self.next_bits_int(field_bits) % R_MOD


class PoseidonSpec:
    def __init__(self, t: int = 5, r_f: int = 8, r_p: int = 60):
        self.t, self.r_f, self.r_p = t, r_f, r_p
        g = Grain(254, t, r_f, r_p)
        self.constants = [[g.next_field_element(254) for _ in range(t)] for _ in range(r_f + r_p)]
        vals = [g.next_field_element_without_rejection(254) for _ in range(2 * t)]
        xs, ys = vals[:t], vals[t:]          # Cauchy matrix 1 / (x_i + y_j), as generate_parameters_grain.sage
        self.mds = [[pow((xs[i] + ys[j]) % R_MOD, -1, R_MOD) for j in range(t)] for i in range(t)]

    def permute(self, state: List[int]) -> List[int]:
        t, half = self.t, self.r_f // 2
        s = list(state)
        for rnd in range(self.r_f + self.r_p):
            s = [(v + c) % R_MOD for v, c in zip(s, self.constants[rnd])]
            if rnd < half or rnd >= half + self.r_p:
                s = [pow(v, 5, R_MOD) for v in s]
            else:
                s[0] = pow(s[0], 5, R_MOD)
            s = [sum(self.mds[i][j] * s[j] for j in range(t)) % R_MOD for i in range(t)]
        return s


_SPEC_CACHE = {}


def poseidon_spec(t: int = 5, r_f: int = 8, r_p: int = 60) -> PoseidonSpec:
    key = (t, r_f, r_p)
    if key not in _SPEC_CACHE:
        _SPEC_CACHE[key] = PoseidonSpec(t, r_f, r_p)
    return _SPEC_CACHE[key]


class PoseidonSponge:
    """snark-verifier `util::hash::Poseidon<F, L, T, RATE>`: `update` buffers elements; `squeeze`
    absorbs the buffer RATE elements at a time -- a chunk shorter than RATE (and, when the buffer
    length is a multiple of RATE, one extra empty chunk) gets the padding element 1 right behind
    its last input -- and returns state[1].  Initial state = (2^64, 0, ..., 0)."""

    def __init__(self, t: int = 5, rate: int = 4, r_f: int = 8, r_p: int = 60):
        assert t == rate + 1
        self.spec, self.rate = poseidon_spec(t, r_f, r_p), rate
        self.state = [1 << 64] + [0] * rate
        self.buf: List[int] = []

    def update(self, elems):
        self.buf.extend(e % R_MOD for e in elems)

    def _absorb(self, chunk):
        s = self.state
        for i, v in enumerate(chunk):
            s[1 + i] = (s[1 + i] + v) % R_MOD
        if len(chunk) < self.rate:
            s[1 + len(chunk)] = (s[1 + len(chunk)] + 1) % R_MOD
        self.state = self.spec.permute(s)

    def squeeze(self) -> int:
        buf, self.buf = self.buf, []
        exact = len(buf) % self.rate == 0
        for off in range(0, len(buf), self.rate):
            self._absorb(buf[off:off + self.rate])
        if exact:
            self._absorb([])
        return self.state[1]
