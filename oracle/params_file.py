"""TEST INFRASTRUCTURE (oracle): the `params{k}` file format of halo2 ParamsKZG::write_custom /
read_custom restated over big-int points.  Reference: prover::utils::load_params checks
`4 + 2 * 2^k * g1 + 2 * g2` bytes with g1 = 32 (SerdeFormat::Processed) or 64 (RawBytes /
RawBytesUnchecked, the default) and g2 = 2 * g1  [REF prover/src/utils.rs:32,39-84].

    file = u32 k (LE) | g[0..n) | g_lagrange[0..n) | g2 | s_g2

RawBytes G1 = x | y as Montgomery limbs (oracle/bn254.g1_affine_bytes_raw); Processed G1 = x
canonical LE with the parity of y in bit 254 and the identity flag in bit 255 (bn254.g1_compress); RawBytes G2 = x.c0 | x.c1 |
y.c0 | y.c1, Montgomery limbs.  Only tests may import this module.
"""
from typing import List, Tuple

from . import bn254 as b
from . import pairing

PROCESSED, RAW, RAW_UNCHECKED = 0, 1, 2


def file_len(k: int, fmt: int) -> int:
    g1 = 32 if fmt == PROCESSED else 64
    return 4 + 2 * (1 << k) * g1 + 2 * 2 * g1


def g2_raw_bytes(pt) -> bytes:
    if pt is None:
        return bytes(128)
    x, y = pt
    return b"".join(b.mont_bytes(int(c), b.P_MOD) for c in (x.c[0], x.c[1], y.c[0], y.c[1]))


def setup_with_s(k: int, s: int):
    """ParamsKZG::unsafe_setup_with_s: g[i] = s^i G, g_lagrange[i] = L_i(s) G, g2, s g2."""
    n = 1 << k
    g = [b.g1_mul(b.G1_GEN, pow(s, i, b.R_MOD)) for i in range(n)]
    w = b.omega_for_k(k)
    zn = (pow(s, n, b.R_MOD) - 1) % b.R_MOD
    lag = []
    for i in range(n):
        wi = pow(w, i, b.R_MOD)
        li = wi * zn % b.R_MOD * pow(n * (s - wi) % b.R_MOD, -1, b.R_MOD) % b.R_MOD
        lag.append(b.g1_mul(b.G1_GEN, li))
    return g, lag, pairing.G2_GEN, pairing.ec_mul(pairing.G2_GEN, s)


def write(k: int, g: List, g_lagrange: List, g2_blob: bytes, s_g2_blob: bytes, fmt: int) -> bytes:
    enc = b.g1_compress if fmt == PROCESSED else b.g1_affine_bytes_raw
    out = k.to_bytes(4, "little") + b"".join(enc(p) for p in g) + b"".join(enc(p) for p in g_lagrange) + g2_blob + s_g2_blob
    assert len(out) == file_len(k, fmt)
    return out


def g1_decompress(data: bytes):
    try:
        return b.g1_decompress(data)
    except ValueError as e:
        raise AssertionError(str(e))


def read(data: bytes, fmt: int) -> Tuple[int, List, List, bytes, bytes]:
    k = int.from_bytes(data[:4], "little")
    if len(data) != file_len(k, fmt):
        raise ValueError(f"invalid params file len {len(data)} for degree {k}")
    n, g1 = 1 << k, 32 if fmt == PROCESSED else 64
    dec = g1_decompress if fmt == PROCESSED else b.g1_affine_from_bytes_raw
    pts = [dec(data[4 + i * g1:4 + (i + 1) * g1]) for i in range(2 * n)]
    if fmt == RAW:
        assert all(b.g1_is_on_curve(p) for p in pts if p is not None)
    tail = data[4 + 2 * n * g1:]
    return k, pts[:n], pts[n:], tail[:2 * g1], tail[2 * g1:]
