"""CPU: pins the Python oracle against every golden vector the reference holds for this path
 and re-derives the constants both the oracle and the HIP code hard-wire."""
import hashlib
import json
import os

from oracle import bn254 as b
from oracle import pairing as pr

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF = json.load(open(os.path.join(GOLDEN, "reference_vectors.json")))        # extracted from the reference's sources (golden/make_reference_vectors.py)
PUB = json.load(open(os.path.join(GOLDEN, "published_vectors.json")))


def test_g3_mockprover_third_challenge():
    # [REF zkevm-circuits/src/super_circuit.rs:729]
    assert REF["G3_mockprover_third_challenge"]["ref"] == "zkevm-circuits/src/super_circuit.rs:729"
    assert b.mock_prover_challenge(3) == int(REF["G3_mockprover_third_challenge"]["value"], 16) == 0x207A52BA34E1ED068BE1E33B0BC39C8EDE030835F549FE5C0DBE91DCE97D17D2


def test_g4_fq_modulus_minus_two_word():
    # [REF zkevm-circuits/src/ecc_circuit/test.rs:208]  y = p - 2 is -2 mod p, i.e. -(G.y)
    assert b.P_MOD - 2 == int(REF["G4_fq_modulus_minus_two"]["value"], 16) == 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD45
    assert b.g1_add(b.G1_GEN, (1, b.P_MOD - 2)) is None
    assert not b.g1_is_on_curve((2, 3))
    assert not b.g1_is_on_curve((b.P_MOD + 1, b.P_MOD + 2))


def test_g5_ecadd_ecmul_precompile_vectors():
    # [REF bus-mapping/src/evm/opcodes/callop.rs:883-917]
    two_g = (0x030644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD3,
             0x15ED738C0E0A7C92E7845F96B2AE9C0A68A6A449E3538FC7FF3EBF7A5A18A2C4)
    assert two_g == (int(REF["G5_two_G"]["x"], 16), int(REF["G5_two_G"]["y"], 16)) and REF["G5_two_G"]["occurrences"] == 2     # ecAdd and ecMul
    assert b.g1_add(b.G1_GEN, b.G1_GEN) == two_g
    assert b.g1_mul(b.G1_GEN, 2) == two_g
    assert b.g1_is_on_curve(two_g)


def test_g6_ecpairing_precompile_vector():
    # [REF bus-mapping/src/evm/opcodes/callop.rs:925-936]: words are PUSHed last-first.
    pushed = ["23a8eb0b0996252cb548a4487da97b02422ebc0e834613f954de6c7e0afdc1fc", "2a23af9a5ce2ba2796c1f4e453a370eb0af8c212d9dc9acd8fc02c2e907baea2",
              "091058a3141822985733cbdddfed0fd8d6c104e9e9eff40bf5abfef9ab163bc7", "1971ff0471b09fa93caaf13cbf443c1aede09cc4328f5a62aad45f40ec133eb4",
              "30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd45", "0000000000000000000000000000000000000000000000000000000000000001",
              "2fe02e47887507adf0ff1743cbac6ba291e66f59be6bd763950bb16041a0a85e", "2bd368e28381e8eccb5fa81fc26cf3f048eea9abfdd85d7ed3ab3698d63e4f90",
              "22606845ff186793914e03e21df544c34ffe2f2f3504de8a79d9159eca2d98d9", "1fb19bb476f6b9e44e2a32234da8212f61cd63919354bc06aef31e3cfaff3ebc",
              "2c0f001f52110ccfe69108924926e45f0b0c868df0e7bde1fe16d3242dc715f6", "2cf44499d5d27bb186308b7af7af02ac5bc9eeb6a3d147c186b21fb1b76e18da"]
    assert pushed == REF["G6_ecpairing_pushed_words"]["words"]
    ws = [int(x, 16) for x in reversed(pushed)]
    pairs = []
    for i in range(2):
        x1, y1, x2i, x2r, y2i, y2r = ws[6 * i:6 * i + 6]
        P = (x1, y1)
        Q = (pr.FQ2([x2r, x2i]), pr.FQ2([y2r, y2i]))
        assert b.g1_is_on_curve(P) and pr.is_on_curve(Q, pr.B2)
        pairs.append((P, Q))
    assert pr.pairing_check(pairs)
    # and a negative control: break bilinearity
    pairs[0] = (b.g1_add(pairs[0][0], b.G1_GEN), pairs[0][1])
    assert not pr.pairing_check(pairs)


def test_field_constants():
    R, P = b.R_MOD, b.P_MOD
    assert R.bit_length() == 254 and P.bit_length() == 254
    assert (R - 1) % (1 << 28) == 0 and (R - 1) % (1 << 29) != 0
    assert b.FR_ROOT_OF_UNITY == 0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C
    assert pow(b.FR_ROOT_OF_UNITY, 1 << 28, R) == 1 and pow(b.FR_ROOT_OF_UNITY, 1 << 27, R) == R - 1
    assert b.FR_DELTA == 0x09226B6E22C6F0CA64EC26AAD4C86E715B5F898E5E963F25870E56BBE533E9A2
    assert pow(b.FR_ZETA, 3, R) == 1 and b.FR_ZETA != 1
    assert b.FR_MONT_R == 0x0E0A77C19A07DF2F666EA36F7879462E36FC76959F60CD29AC96341C4FFFFFFB
    assert b.FR_MONT_R2 == 0x0216D0B17F4E44A58C49833D53BB808553FE3AB1E35C59E31BB8E645AE216DA7
    assert b.FR_INV64 == 0xC2E1F593EFFFFFFF and b.FQ_INV64 == 0x87D20782E4866389
    assert b.FQ_MONT_R == 0x0E0A77C19A07DF2F666EA36F7879462C0A78EB28F5C70B3DD35D438DC58F0D9D
    assert (R * b.FR_INV64 + 1) % (1 << 64) == 0 and (P * b.FQ_INV64 + 1) % (1 << 64) == 0
    # 29-bit radix constants used by ff29.hip.hpp
    assert (-pow(P, -1, 1 << 29)) % (1 << 29) == 0x4866389 and (-pow(R, -1, 1 << 29)) % (1 << 29) == 0xFFFFFFF


def test_mont_mul_cios_matches_definition():
    import random
    rng = random.Random(5)
    for mod, inv in ((b.R_MOD, b.FR_INV64), (b.P_MOD, b.FQ_INV64)):
        rinv = pow(1 << 256, -1, mod)
        for _ in range(200):
            x, y = rng.randrange(mod), rng.randrange(mod)
            assert b.mont_mul_cios(x, y, mod, inv) == x * y * rinv % mod
        assert b.mont_mul_cios(mod - 1, mod - 1, mod, inv) == (mod - 1) * (mod - 1) * rinv % mod


def test_best_fft_is_the_dft_and_domain_round_trips():
    import random
    rng = random.Random(9)
    for k in (1, 2, 5, 7):
        a = [rng.randrange(b.R_MOD) for _ in range(1 << k)]
        w = list(a)
        b.best_fft(w, b.omega_for_k(k), k)
        assert w == b.ntt_naive(a, b.omega_for_k(k))
    dom = b.EvaluationDomain(j=5, k=4)   # degree 5 -> extended_k = 6
    assert dom.extended_k == 6 and len(dom.t_evaluations) == 4
    a = [rng.randrange(b.R_MOD) for _ in range(16)]
    assert dom.coeff_to_lagrange(dom.lagrange_to_coeff(a)) == a
    ext = dom.coeff_to_extended(a)
    # extended evaluation i is f(zeta * w_ext^i)
    for i in (0, 1, 17, 63):
        x = b.FR_ZETA * pow(dom.extended_omega, i, b.R_MOD) % b.R_MOD
        assert ext[i] == b.eval_polynomial(a, x)
    assert dom.extended_to_coeff(ext)[:16] == a
    x = rng.randrange(b.R_MOD)
    q = b.kate_division(a, x)
    fx = b.eval_polynomial(a, x)
    y = rng.randrange(b.R_MOD)
    assert (b.eval_polynomial(q, y) * (y - x) + fx) % b.R_MOD == b.eval_polynomial(a, y)


def test_pippenger_restatement_matches_naive_sum():
    import random
    rng = random.Random(11)
    for n in (1, 3, 5, 33, 70):
        pts = [b.g1_mul(b.G1_GEN, rng.randrange(1, b.R_MOD)) for _ in range(n)]
        sc = [rng.randrange(b.R_MOD) for _ in range(n)]
        if n > 3:
            pts[2] = None
            sc[1] = 0
        assert b.msm_pippenger_halo2(sc, pts) == b.msm_naive(sc, pts)


def test_kzg_identity_with_pairing():
    """commit(f) - f(z) G = (s - z) * commit(q): checked the way a KZG verifier does it."""
    import random
    rng = random.Random(13)
    s, n = 0xABCDEF, 8
    g = b.srs_powers(s, n)
    f = [rng.randrange(b.R_MOD) for _ in range(n)]
    z = rng.randrange(b.R_MOD)
    C, W = b.msm_naive(f, g), b.msm_naive(b.kate_division(f, z), g[:n - 1])
    lhs = b.g1_add(C, b.g1_neg(b.g1_mul(b.G1_GEN, b.eval_polynomial(f, z))))
    s_g2 = pr.ec_mul(pr.G2_GEN, s)
    rhs_q = pr.ec_add(s_g2, pr.ec_neg(pr.ec_mul(pr.G2_GEN, z)))
    assert pr.pairing_check([(lhs, pr.ec_neg(pr.G2_GEN)), (W, rhs_q)])


def test_chacha20_block_rfc7539_vector():
    """RFC 7539 section 2.3.2 known-answer test for the block function behind zk_fr_random."""
    key = bytes(range(32))
    blk = b.chacha20_block(key, 1 | (0x09000000 << 32), 0x4A000000)
    assert blk.hex() == PUB["chacha20_block_rfc7539_2_3_2"]["keystream_block"] == ("10f1e7e4d13b5915500fdd1fa32071c4c7d1f4c733c068030422aa9ac3d46c4e"
                                                                                "d2826446079faa0914c2d705d98b02a2b5129cd1de164eb9cbd083e8a2503c4e")
    vals = b.fr_random_chacha(key, 7, 0, 4)
    assert len(set(vals)) == 4 and all(0 <= v < b.R_MOD for v in vals)


def test_reference_pairing_case_alpha_beta():
    """The reference's own valid ecPairing case [REF zkevm-circuits/src/ecc_circuit/test.rs:239-266]: alpha = 0x102030, beta = 0x413121,
    e(-alpha G1, beta G2) * e(alpha beta G1, G2) = 1 -- for the oracle's pairing and for the product's host pairing
    (csrc/host_pairing.hpp behind zk_host_pairing_check)."""
    import numpy as np

    alpha, beta = 0x102030, 0x413121
    p_neg = b.g1_neg(b.g1_mul(b.G1_GEN, alpha))
    q = pr.ec_mul(pr.G2_GEN, beta)
    s = b.g1_mul(b.G1_GEN, alpha * beta % b.R_MOD)
    assert pr.pairing_check([(p_neg, q), (s, pr.G2_GEN)])
    assert not pr.pairing_check([(b.g1_neg(p_neg), q), (s, pr.G2_GEN)])            # the reference's "invalid" variants break the relation
