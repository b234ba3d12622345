"""GPU: SRS files (ParamsKZG::read_custom / write_custom, what prover::utils::load_params drives,
[REF prover/src/utils.rs:39-84]) through the C ABI against the big-int restatement of the format:
files written from a device SRS equal the oracle's byte for byte in all three serde formats, read
back they give the same SRS, a wrong length is refused before parsing, RawBytes rejects points off
the curve and RawBytesUnchecked does not look."""
import numpy as np
import pytest

from oracle import bn254 as b
from oracle import params_file

pytestmark = pytest.mark.gpu


def _s_mont(s):
    return np.frombuffer(b.mont_bytes(s, b.R_MOD), dtype=np.uint64).copy()


@pytest.mark.parametrize("k,s", [(0, 7), (1, 1234), (5, 0x5EC2E7)])
@pytest.mark.parametrize("fmt", [0, 1, 2])
def test_write_matches_oracle_and_reads_back(zk, ctx, k, s, fmt):
    srs = ctx.srs_setup_with_s(k, _s_mont(s))
    g2, s_g2 = zk.binding.g2_setup(_s_mont(s))
    if fmt == 0:
        g2, s_g2 = g2[:64], s_g2[:64]                   # the G2 blobs are opaque to the prover
    data = ctx.params_write(srs, g2, s_g2, fmt)
    g, lag, _, _ = params_file.setup_with_s(k, s)
    assert data == params_file.write(k, g, lag, g2, s_g2, fmt)
    back, g2_b, s_g2_b = ctx.params_read(data, fmt)
    assert back.k == k and (g2_b, s_g2_b) == (g2, s_g2)
    assert np.array_equal(back.download_g(), srs.download_g())
    assert np.array_equal(back.download_g_lagrange(), srs.download_g_lagrange())
    back.destroy()
    srs.destroy()


def test_processed_round_trip_2_16_and_commit(zk, ctx, cref):
    """A 2^16 SRS through the compressed format: 2^17 square roots on the device; the SRS read back
    commits like the original."""
    k, s = 16, 987654321
    srs = ctx.srs_setup_with_s(k, _s_mont(s))
    g2, s_g2 = zk.binding.g2_setup(_s_mont(s))
    data = ctx.params_write(srs, g2[:64], s_g2[:64], 0)
    assert len(data) == zk.binding.params_file_len(k, 0)
    back, _, _ = ctx.params_read(data, 0)
    assert np.array_equal(back.download_g(), srs.download_g())
    assert np.array_equal(back.download_g_lagrange(), srs.download_g_lagrange())
    col = cref.rand_fr_stream(3, 1 << k)
    d = ctx.to_device(col)
    assert np.array_equal(ctx.commit(back, d, 1 << k, lagrange=True), ctx.commit(srs, d, 1 << k, lagrange=True))
    back.destroy()
    srs.destroy()


def test_wrong_length_is_refused(zk, ctx):
    srs = ctx.srs_setup_with_s(4, _s_mont(5))
    g2, s_g2 = zk.binding.g2_setup(_s_mont(5))
    data = ctx.params_write(srs, g2, s_g2, 2)
    for bad in (data[:-1], data + b"\0", data[:3], (5).to_bytes(4, "little") + data[4:]):
        with pytest.raises(zk.ZkError, match="params file"):
            ctx.params_read(bad, 2)
    with pytest.raises(zk.ZkError, match="params file"):
        ctx.params_read(data, 0)                        # right file, wrong serde format
    srs.destroy()


def test_raw_checks_points_and_unchecked_does_not(zk, ctx):
    k = 4
    srs = ctx.srs_setup_with_s(k, _s_mont(11))
    g2, s_g2 = zk.binding.g2_setup(_s_mont(11))
    data = bytearray(ctx.params_write(srs, g2, s_g2, 1))
    data[4 + 64 * 3] ^= 1                               # g[3].x no longer on the curve
    with pytest.raises(zk.ZkError, match="not points of the curve"):
        ctx.params_read(bytes(data), 1)
    loose, _, _ = ctx.params_read(bytes(data), 2)       # RawBytesUnchecked takes it as it is
    assert loose.k == k
    loose.destroy()
    # a non-canonical limb image (x + p) is refused by RawBytes too
    good = bytearray(ctx.params_write(srs, g2, s_g2, 1))
    x = int.from_bytes(good[4:36], "little") + b.P_MOD
    if x < 1 << 256:
        good[4:36] = x.to_bytes(32, "little")
        with pytest.raises(zk.ZkError, match="not points of the curve"):
            ctx.params_read(bytes(good), 1)
    # Processed: an x with no square root behind it
    comp = bytearray(ctx.params_write(srs, g2[:64], s_g2[:64], 0))
    for delta in range(1, 50):
        trial = bytearray(comp)
        xx = (int.from_bytes(trial[4:36], "little") & ((1 << 254) - 1)) + delta          # both flag bits cleared
        if pow((xx ** 3 + 3) % b.P_MOD, (b.P_MOD - 1) // 2, b.P_MOD) != 1:
            trial[4:36] = xx.to_bytes(32, "little")
            with pytest.raises(zk.ZkError, match="not points of the curve"):
                ctx.params_read(bytes(trial), 0)
            break
    srs.destroy()
