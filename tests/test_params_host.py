"""CPU: the host-only parts of the SRS-file API -- zk_params_file_len (the length
prover::utils::load_params insists on, [REF prover/src/utils.rs:56-75]) and zk_g2_setup (the G2 half
of unsafe_setup_with_s: generator and s * generator) -- against the oracle.  No GPU work is issued."""
import numpy as np
import pytest

from oracle import bn254 as b
from oracle import pairing, params_file


def _lib():
    import zkevm_circuits_amd as z
    return z


@pytest.mark.parametrize("k", [0, 1, 14, 20, 26])
def test_file_len_matches_load_params_rule(k):
    z = _lib()
    for fmt, g1 in ((0, 32), (1, 64), (2, 64)):
        assert z.binding.params_file_len(k, fmt) == 4 + 2 * (1 << k) * g1 + 2 * 2 * g1 == params_file.file_len(k, fmt)
    assert z.binding.params_file_len(29, 2) == 0 and z.binding.params_file_len(4, 3) == 0


@pytest.mark.parametrize("s", [1, 2, 1234, b.R_MOD - 1, 0x5EC2E7 ** 9 % b.R_MOD])
def test_g2_setup_matches_oracle(s):
    z = _lib()
    s_mont = np.frombuffer(b.mont_bytes(s, b.R_MOD), dtype=np.uint64).copy()
    g2, s_g2 = z.binding.g2_setup(s_mont)
    assert g2 == params_file.g2_raw_bytes(pairing.G2_GEN)
    assert s_g2 == params_file.g2_raw_bytes(pairing.ec_mul(pairing.G2_GEN, s))


def test_g2_setup_zero_scalar_is_identity():
    z = _lib()
    g2, s_g2 = z.binding.g2_setup(np.zeros(4, dtype=np.uint64))
    assert s_g2 == bytes(128) and g2 == params_file.g2_raw_bytes(pairing.G2_GEN)


def test_oracle_file_round_trip():
    g, lag, g2, sg2 = params_file.setup_with_s(3, 1234)
    for fmt in (params_file.PROCESSED, params_file.RAW, params_file.RAW_UNCHECKED):
        blob = params_file.g2_raw_bytes(g2), params_file.g2_raw_bytes(sg2)
        if fmt == params_file.PROCESSED:
            blob = blob[0][:64], blob[1][:64]           # opaque to the prover: any 64 bytes
        data = params_file.write(3, g, lag, blob[0], blob[1], fmt)
        k, g_, lag_, a, c = params_file.read(data, fmt)
        assert (k, g_, lag_, a, c) == (3, g, lag, blob[0], blob[1])
        with pytest.raises(ValueError):
            params_file.read(data + b"\0", fmt)
