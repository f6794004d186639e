"""Oracle C arithmetic vs Python big integers (an independent implementation) and vs definitions."""
import random

import numpy as np

from oracle import oracle_lib as ol
from oracle.oracle_lib import R_MOD, Q_MOD

rng = random.Random(0xB254)
EDGE = [0, 1, 2, R_MOD - 1, R_MOD - 2, (1 << 253), (1 << 128) - 1]


def _mont(xs, p):
    return ol.ints_to_array([x * ol.MONT_R % p for x in xs])


def _unmont(a, p):
    return [v * pow(ol.MONT_R, -1, p) % p for v in ol.array_to_ints(a)]


def test_constants():
    lib = ol.lib()
    for name, p in (("fr", R_MOD), ("fq", Q_MOD)):
        one = np.zeros((1, 4), dtype=np.uint64)
        getattr(lib, "orc_%s_from_canonical" % name)(ol._p(one), ol._p(ol.ints_to_array([1])), 1)
        assert ol.array_to_ints(one)[0] == ol.MONT_R % p


def test_field_ops_match_python():
    lib = ol.lib()
    for name, p in (("fr", R_MOD), ("fq", Q_MOD)):
        xs = [e % p for e in EDGE] + [rng.randrange(p) for _ in range(200)]
        ys = [rng.randrange(p) for _ in xs[:-3]] + [0, 1, p - 1]
        for x, y in zip(xs, ys):
            a, b = _mont([x], p), _mont([y], p)
            out = np.zeros((1, 4), dtype=np.uint64)
            for op, ref in (("mul", x * y % p), ("add", (x + y) % p), ("sub", (x - y) % p)):
                getattr(lib, "orc_%s_%s" % (name, op))(ol._p(out), ol._p(a), ol._p(b))
                assert _unmont(out, p)[0] == ref, (name, op, x, y)
            getattr(lib, "orc_%s_inv" % name)(ol._p(out), ol._p(a))
            assert _unmont(out, p)[0] == (pow(x, -1, p) if x else 0)


def test_omega_matches_survey():
    assert ol.omega(28) == 0x03ddb9f5166d18b798865ea93dd31f743215cf6dd39329c8d34f1ed960c37c9c
    assert ol.omega(3) == 0x2b337de1c8c14f22ec9b9e2f96afef3652627366f8170a0a948dad4ac1bd5e80
    assert ol.omega(20) == 0x2a14464f1ff42de3856402b62520e670745e39fada049d5b2f0e1e3182673378
    assert ol.omega(22) == 0x18c95f1ae6514e11a1b30fd7923947c5ffcec5347f16e91b4dd654168326bede
    assert ol.omega(28) == pow(7, (R_MOD - 1) >> 28, R_MOD)


def test_ntt_matches_definition_and_roundtrips():
    for log_n in (1, 3, 6, 9):
        n = 1 << log_n
        xs = [rng.randrange(R_MOD) for _ in range(n)]
        a = ol.fr_vec(xs)
        for threads in (1, 4):
            f = ol.ntt(a, log_n, threads=threads)
            assert np.array_equal(f, ol.dft_naive(a, log_n))
            assert np.array_equal(ol.ntt(f, log_n, inverse=True, threads=threads), a)
        w = ol.omega(log_n)
        k = 5 % n
        assert ol.fr_ints(f)[k] == sum(x * pow(w, i * k, R_MOD) for i, x in enumerate(xs)) % R_MOD
        fc = ol.ntt(a, log_n, coset=7)
        assert ol.fr_ints(fc)[k] == ol.poly_eval(a, 7 * pow(w, k, R_MOD) % R_MOD)
        assert np.array_equal(ol.ntt(fc, log_n, inverse=True, coset=7), a)


def test_vector_ops():
    n = 64
    xs = [rng.randrange(R_MOD) for _ in range(n)]
    ys = [rng.randrange(R_MOD) for _ in range(n)]
    xs[3] = 0
    a, b = ol.fr_vec(xs), ol.fr_vec(ys)
    s = rng.randrange(R_MOD)
    assert ol.fr_ints(ol.vmul(a, b)) == [x * y % R_MOD for x, y in zip(xs, ys)]
    assert ol.fr_ints(ol.vaxpy(a, s, b)) == [(x + s * y) % R_MOD for x, y in zip(xs, ys)]
    assert ol.fr_ints(ol.vbatch_inv(a)) == [pow(x, -1, R_MOD) if x else 0 for x in xs]
    pp, acc = [], 1
    for y in ys:
        pp.append(acc)
        acc = acc * y % R_MOD
    assert ol.fr_ints(ol.vshifted_prefix_product(b)) == pp
    z = rng.randrange(R_MOD)
    q = ol.fr_ints(ol.poly_div_linear(a, z))
    pz = ol.poly_eval(a, z)
    # (x - z) * q(x) + p(z) == p(x)
    back = [(0 if i == 0 else q[i - 1]) - z * q[i] for i in range(n)]
    back[0] += pz
    assert [v % R_MOD for v in back] == xs and q[-1] == 0


def test_keccak_kat():
    assert ol.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert ol.keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    assert ol.keccak256(b"a" * 200).hex() == ol.keccak256(b"a" * 136 + b"a" * 64).hex()


def test_g1_group_law_and_crs42(golden_crs):
    G = ol.g1_generator()
    assert ol.g1_to_ints(G) == (1, 2) and ol.g1_on_curve(G)
    assert np.array_equal(ol.g1_mul(G, 42), golden_crs.g1[1])
    assert np.array_equal(ol.g1_add(golden_crs.g1[1], ol.g1_neg(golden_crs.g1[1])), np.zeros(8, dtype=np.uint64))
    assert np.array_equal(ol.g1_add(G, G), ol.g1_mul(G, 2))
    assert ol.g1_is_inf(ol.g1_mul(G, R_MOD))
    assert np.array_equal(ol.crs42(1024, threads=3), golden_crs.g1)      # Crs::crs_42 == committed key


def test_msm_trapdoor_and_edges(golden_crs):
    G = ol.g1_generator()
    for n in (1, 5, 31, 32, 200, 1024):
        ks = [rng.randrange(R_MOD) for _ in range(n)]
        for i in range(0, n, 7):
            ks[i] = rng.choice([0, 1, R_MOD - 1, 2])
        want = ol.g1_mul(G, sum(k * pow(42, i, R_MOD) for i, k in enumerate(ks)) % R_MOD)
        got = ol.msm(golden_crs.g1[:n], ol.fr_vec(ks), threads=3)
        assert np.array_equal(got, want), n
        if n <= 32:
            assert np.array_equal(ol.msm(golden_crs.g1[:n], ol.fr_vec(ks), naive=True), want)
    assert ol.g1_is_inf(ol.msm(golden_crs.g1[:16], ol.fr_vec([0] * 16)))
    # duplicates hitting add == double, and P + (-P)
    bases = np.stack([golden_crs.g1[3]] * 4 + [ol.g1_neg(golden_crs.g1[3])] * 4)
    got = ol.msm(bases, ol.fr_vec([5, 5, 5, 5, 5, 5, 5, 5]))
    assert ol.g1_is_inf(got)
    got = ol.msm(bases, ol.fr_vec([9, 9, 9, 9, 1, 1, 1, 1]))
    assert np.array_equal(got, ol.g1_mul(golden_crs.g1[3], 32))


def test_g1_intt_is_lagrange_basis(golden_crs):
    log_n, tau = 5, 42
    n = 1 << log_n
    out = ol.g1_intt(golden_crs.g1[:n], log_n)
    w = ol.omega(log_n)
    G = ol.g1_generator()
    zh = (pow(tau, n, R_MOD) - 1) % R_MOD
    for i in (0, 1, 7, n - 1):
        wi = pow(w, i, R_MOD)
        li = wi * zh % R_MOD * pow(n * (tau - wi) % R_MOD, -1, R_MOD) % R_MOD
        assert np.array_equal(out[i], ol.g1_mul(G, li))
