"""-m gpu: the pieces of the prover rounds on their own, so that a wrong proof is localised (rows a7, a8, a9 were
only covered through whole-proof byte parity).

  a7  permutation grand product (round 2)            plk_permutation_grand_product_dev  vs  the oracle's vector ops
  a8  quotient (round 3: fused numerator / Z_H, coset iNTT(4N))   plk_prove_trace(5)    vs  the oracle prover's t(x)
  a9  evaluations, linearisation, openings (4, 5)    plk_poly_evaluate_at_dev, plk_poly_divide_by_linear_dev, plk_prove_trace(6..8)
The reference counterparts live in bellman_ce's better_cs prover behind prove_by_steps (src/plonk.rs:152-159); the formulas
are those of SURVEY.md Appendix A.4, restated in oracle/plonk_oracle.py (pinned by the golden proof)."""
import json

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle_lib as ol, plonk_oracle as po
from oracle.oracle_lib import R_MOD

NON_RESIDUES = (1, 5, 7, 10)


@pytest.fixture(scope="module")
def ctx():
    import plonkit_amd as pa
    c = pa.Context(0)
    yield c
    c.close()


def _rand_fr(n, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)
    return a


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to("cuda:0")


def _host(t):
    return t.cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("log_n", [3, 8, 11, 12, 15])
def test_permutation_grand_product(ctx, log_n):
    import torch
    n = 1 << log_n
    w = [_rand_fr(n, 10 * log_n + j) for j in range(4)]
    sig = [_rand_fr(n, 100 * log_n + j) for j in range(4)]
    beta, gamma = 0x1234567890abcdef1234567890abcdef % R_MOD, (R_MOD - 12345)
    dom = ol.vpowers(ol.omega(log_n), n)
    num = den = None
    for j in range(4):
        nj = ol.vadd_scalar(ol.vaxpy(w[j], beta * NON_RESIDUES[j] % R_MOD, dom), gamma)
        dj = ol.vadd_scalar(ol.vaxpy(w[j], beta, sig[j]), gamma)
        num = nj if num is None else ol.vmul(num, nj)
        den = dj if den is None else ol.vmul(den, dj)
    want = ol.vshifted_prefix_product(ol.vmul(num, ol.vbatch_inv(den)))        # z_0 = 1, z_{i+1} = z_i * num_i / den_i
    out = torch.zeros((n, 4), dtype=torch.int64, device="cuda:0")
    ctx.permutation_grand_product_dev([_dev(x) for x in w], [_dev(x) for x in sig], ol.fr_mont(beta), ol.fr_mont(gamma), log_n, out)
    assert np.array_equal(_host(out), want)


@pytest.mark.parametrize("n", [1, 2, 7, 2047, 2048, 2049, (1 << 15) + 3, 1 << 17])
def test_evaluate_at_and_divide_by_linear(ctx, n):
    import torch
    p = _rand_fr(n, n)
    z = (0xdeadbeefcafebabe1122334455667788 * (n + 1)) % R_MOD
    d = _dev(p)
    assert ol.fr_ints(ctx.poly_evaluate_at_dev(d, n, ol.fr_mont(z)).reshape(1, 4))[0] == ol.poly_eval(p, z)
    q = torch.zeros((n, 4), dtype=torch.int64, device="cuda:0")
    ctx.poly_divide_by_linear_dev(d, n, ol.fr_mont(z), q)
    assert np.array_equal(_host(q), ol.poly_div_linear(p, z))
    # (x - z) * q(x) + p(z) == p(x) at a second point: independent of the oracle's own division
    y = 0x55aa55aa55aa55aa % R_MOD
    lhs = ((y - z) * ol.poly_eval(_host(q), y) + ol.poly_eval(p, z)) % R_MOD
    assert lhs == ol.poly_eval(p, y)


@pytest.mark.parametrize("n_cons,log_srs", [(300, 10), (3000, 13)])
def test_round_by_round_against_the_oracle_prover(ctx, n_cons, log_srs):
    """every vector between the rounds (wire polynomials, z, the 4N quotient coefficients, the linearisation polynomial,
    the two opening quotients) equals the oracle prover's on the same circuit and challenges"""
    import plonkit_amd as pa
    from tests.test_gpu_prove import _chain, _circuit_json
    r1cs, wit = _chain(n_cons, 77 + n_cons)
    js = _circuit_json(r1cs, 1)
    r_o = po.load_r1cs_json(json.loads(js))
    srs = ol.crs42(1 << log_srs)
    crs = po.Crs(srs, b"\x00" * 256)
    ctx.srs_upload(srs)
    circ = pa.Circuit(js, True, json.dumps([str(v) for v in wit]).encode(), True)
    setup = pa.SetupForProver(ctx, circ)
    S = po.setup(r_o)
    P, dbg = po.prove(r_o, wit, crs, S, return_debug=True)
    assert setup.prove(circ) == po.write_proof(P)
    for j in range(4):
        assert np.array_equal(ctx.prove_trace(j), dbg["w_coef"][j]), "round 1: wire polynomial %d" % j
    assert np.array_equal(ctx.prove_trace(4), dbg["z_coef"]), "round 2: grand product polynomial"
    assert np.array_equal(ctx.prove_trace(5), dbg["t_coef"]), "round 3: quotient polynomial (4N coefficients)"
    assert np.array_equal(ctx.prove_trace(6), dbg["r"]), "round 4: linearisation polynomial"
    assert np.array_equal(ctx.prove_trace(7), dbg["W_z"]), "round 5: opening quotient at z"
    assert np.array_equal(ctx.prove_trace(8), dbg["W_zw"]), "round 5: opening quotient at z*omega"
    setup.close(); circ.close()


def test_trace_needs_a_finished_proof():
    import plonkit_amd as pa
    c = pa.Context(0)
    with pytest.raises(pa.PlkError):
        c.prove_trace(0)
    c.close()
