"""-m gpu: the BASELINE.json configurations above 2^20, checked on the device the driver gives the test tier.

  configs[2]  "Synthetic R1CS 2^24 constraints, SRS 2^24": one 2^24-term commitment against the tau = 42 trapdoor
              answer (uniform and witness-like scalars), a whole prove at the 2^22 and 2^24 domains accepted by the
              host verifier (real pairing) and rejected after tampering; two ranks with a sliced SRS at the 2^20 domain
              (tests/test_gpu_sharded_prove.py holds the small sizes).
  configs[3]  dump-lagrange at 2^20: out_i = L_i(42) * G at 64 indices and sum_i out_i = G.
  configs[4]  the recursive prover's kernel shapes: NTT 2^24 and 2^26 (round trip, linearity, evaluation at four
              points), MSM 2^24 (the first item).
  Reference sizes: SETUP_MAX_POW2 = 26 (src/plonk.rs:26-27), the 2^24 key of test/test_poseidon_plonk_recursive.sh:8,28.

The oracle cannot re-run these sizes in test time, so each check is a size-independent property computed on the host
with the oracle's serial C arithmetic (Horner evaluation, one scalar multiplication) — never a second GPU result."""
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle_lib as ol
from oracle.oracle_lib import R_MOD


def _rand_fr(n, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)          # < 2^252 < r: valid Montgomery residues
    return a


def _trapdoor(s):
    """MSM(s, crs_42) = s(42) * G: one Horner pass in the oracle's C arithmetic + one scalar multiplication"""
    return ol.g1_mul(ol.g1_generator(), ol.poly_eval(s, 42))


@pytest.fixture(scope="module")
def ctx24():
    """one context holding the 2^24-point tau = 42 key (1 GiB) and its fixed-base table (15 GiB), shared by the
    commitment and the prove tests of this module"""
    import plonkit_amd as pa
    c = pa.Context(0)
    c.srs_generate(1 << 24, 0, 42)
    yield c
    c.close()


@pytest.mark.parametrize("kind", ["uniform", "witness_like", "all_r_minus_1", "all_ones"])
def test_msm_2pow24_trapdoor(ctx24, kind):
    import torch
    n = 1 << 24
    s = _rand_fr(n, 2400)
    if kind == "all_r_minus_1":                   # every digit of every scalar lands in ONE bucket per window
        s[:] = ol.fr_vec([R_MOD - 1])[0]
    if kind == "all_ones":                        # bellman adds exp == 1 bases directly; here a single bucket of 2^24 entries
        s[:] = ol.fr_vec([1])[0]
    if kind == "witness_like":                    # 50 % zero, 25 % below 2^16, 25 % uniform
        rng = np.random.default_rng(7)
        sel = rng.integers(0, 4, size=n)
        s[sel < 2] = 0
        small = ol.fr_vec(list(range(1 << 16)))   # Montgomery forms of 0 .. 65535
        idx = np.nonzero(sel == 2)[0]
        s[idx] = small[rng.integers(0, 1 << 16, size=idx.shape[0])]
    want = _trapdoor(s)
    d = torch.from_numpy(s.view(np.int64)).to("cuda:0")
    torch.cuda.synchronize()
    assert np.array_equal(ctx24.msm_dev(d, n), want)
    # the same vector through the host entry point (staging copy) and as the sum of two half-length commitments
    half = n // 2
    lo = ctx24.msm_partial_dev(d, half)
    hi = ctx24.msm_partial_dev(d[half:], half, base_offset=half)
    import plonkit_amd as pa
    assert np.array_equal(pa.g1_sum_jacobian(np.stack([lo, hi])), want)


@pytest.mark.parametrize("log_n", [27, 28])
def test_ntt_up_to_the_two_adicity_of_fr(log_n):
    """plk_ntt's largest sizes (2^28 = the 2-adicity of Fr; error code 2 above it): a sparse polynomial against its
    closed form at eight indices, plain and on the coset 7 * <omega>, and the round trip of a dense random vector compared on the device"""
    import torch
    import plonkit_amd as pa
    ctx = pa.Context(0)
    dev = torch.device("cuda:0")
    n = 1 << log_n
    w = ol.omega(log_n)
    pos = [0, 1, 12345, n // 2 + 3, n - 1]
    coef = [5, R_MOD - 2, 0x1234567890abcdef, 7, R_MOD - 1]
    t = torch.zeros((n, 4), dtype=torch.int64, device=dev)
    for p_, c in zip(pos, coef):
        t[p_] = torch.from_numpy(np.asarray(ol.fr_mont(c), dtype=np.uint64).view(np.int64).reshape(4).copy()).to(dev)
    for coset in (None, 7):
        x = t.clone()
        torch.cuda.synchronize()                                   # the library runs on its own stream
        g = ol.fr_mont(coset) if coset else None
        ctx.ntt_dev(x, log_n, coset=g)
        ctx.synchronize()
        for k in (0, 1, 2, 977, n // 3, n // 2, n - 2, n - 1):
            pt = (coset or 1) * pow(w, k, R_MOD) % R_MOD
            want = sum(c * pow(pt, p_, R_MOD) for p_, c in zip(pos, coef)) % R_MOD
            assert ol.fr_ints(x[k:k + 1].cpu().numpy().view(np.uint64))[0] == want, (k, coset)
        ctx.ntt_dev(x, log_n, inverse=True, coset=g)
        ctx.synchronize()
        assert torch.equal(x, t), coset
        del x
    gen = torch.Generator(device=dev)
    gen.manual_seed(log_n)
    d = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device=dev, generator=gen)
    d[:, 3] &= (1 << 60) - 1                                       # < 2^252: valid residues, every other limb full range
    e = d.clone()
    torch.cuda.synchronize()
    ctx.ntt_dev(e, log_n)
    ctx.synchronize()
    assert not torch.equal(e, d)
    ctx.ntt_dev(e, log_n, inverse=True)
    ctx.synchronize()
    assert torch.equal(e, d)
    with pytest.raises(Exception):
        ctx.ntt_dev(e, 29)                                         # beyond the 2-adicity: refused, not attempted
    del t, d, e
    ctx.close()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("log_n", [24, 26])
def test_ntt_recursive_prover_shapes(log_n):
    """NTT 2^24 (N) and 2^26 (4N) of the recursive circuit: round trip, linearity, evaluation at four points"""
    import torch
    import plonkit_amd as pa
    ctx = pa.Context(0)
    n = 1 << log_n
    a, b = _rand_fr(n, 10 + log_n), _rand_fr(n, 20 + log_n)
    dev = torch.device("cuda:0")
    ta = torch.from_numpy(a.view(np.int64)).to(dev)
    ctx.ntt_dev(ta, log_n)
    ctx.synchronize()
    fa = ta.cpu().numpy().view(np.uint64)
    w = ol.omega(log_n)
    for k in (0, 1, 123456789 % n, n - 1):
        assert ol.fr_ints(fa[k:k + 1])[0] == ol.poly_eval(a, pow(w, k, R_MOD)), k
    ctx.ntt_dev(ta, log_n, inverse=True)
    ctx.synchronize()
    assert np.array_equal(ta.cpu().numpy().view(np.uint64), a)
    tb = torch.from_numpy(b.view(np.int64)).to(dev)
    ctx.ntt_dev(tb, log_n)
    ts = torch.from_numpy(ol.vadd(a, b).view(np.int64)).to(dev)
    ctx.ntt_dev(ts, log_n)
    ctx.synchronize()
    assert np.array_equal(ts.cpu().numpy().view(np.uint64), ol.vadd(fa, tb.cpu().numpy().view(np.uint64)))
    # coset variant (the 4N evaluation domain of round 3): f(7 * w^k) at two points, and back
    g = ol.fr_mont(7)
    ctx.ntt_dev(ta, log_n, coset=g)
    ctx.synchronize()
    fc = ta.cpu().numpy().view(np.uint64)
    for k in (5, n - 2):
        assert ol.fr_ints(fc[k:k + 1])[0] == ol.poly_eval(a, 7 * pow(w, k, R_MOD) % R_MOD), k
    ctx.ntt_dev(ta, log_n, inverse=True, coset=g)
    ctx.synchronize()
    assert np.array_equal(ta.cpu().numpy().view(np.uint64), a)
    del ta, tb, ts
    ctx.close()


@pytest.mark.parametrize("log_n", [22, 24])
def test_prove_large_domain_verifies(ctx24, log_n):
    """whole prove (setup_prepare + 5 rounds) at the 2^22 and 2^24 domains on the 2^24 key: the library's host verifier
    (real pairing) and the oracle's restatement of contrib/template.sol accept the proof against the GPU-made
    verification key, both reject it after tampering with an evaluation or a commitment, proving twice gives the same
    bytes."""
    import plonkit_amd as pa
    from oracle import plonk_oracle as po
    circ = pa.Circuit.synthetic((1 << log_n) - 2)
    setup = pa.SetupForProver(ctx24, circ)
    assert setup.domain_size == 1 << log_n
    vk_bytes = setup.verification_key_bytes(pa.crs42_g2_bytes())
    proof = setup.prove(circ)
    assert setup.prove(circ) == proof
    assert pa.verify(vk_bytes, proof)
    vk, P = po.read_vk(vk_bytes), po.read_proof(proof)
    assert P.n == (1 << log_n) - 1 and len(P.inputs) == 1
    assert po.verify(vk, P, tau=42)
    P.linearization_polynomial_at_z = (P.linearization_polynomial_at_z + 1) % R_MOD
    assert not pa.verify(vk_bytes, po.write_proof(P)) and not po.verify(vk, P, tau=42)
    bad = bytearray(proof)
    other = pa.g1_to_bytes(ol.g1_generator())
    off = 8 + 8 + 32 + 8                                  # n, inputs (len + 1 value), wire-commitment count
    bad[off:off + 64] = other                             # first wire commitment replaced by G
    assert not pa.verify(vk_bytes, bytes(bad))
    setup.close(); circ.close()


def test_dump_lagrange_2pow20():
    """configs[3]: SRS 2^20 monomial -> Lagrange on one GPU (Crs::<Lagrange>::from_powers, src/plonk.rs:179-185)"""
    import torch
    import plonkit_amd as pa
    log_n = 20
    n = 1 << log_n
    ctx = pa.Context(0)
    ctx.srs_generate(n, 0, 42)
    out = torch.empty((n, 8), dtype=torch.int64, device="cuda:0")
    ctx.g1_intt_srs_dev(log_n, out)
    ctx.synchronize()
    res = out.cpu().numpy().view(np.uint64)
    w, G = ol.omega(log_n), ol.g1_generator()
    zh = (pow(42, n, R_MOD) - 1) % R_MOD
    rnd = random.Random(log_n)
    for i in [0, 1, n - 1] + [rnd.randrange(n) for _ in range(61)]:
        wi = pow(w, i, R_MOD)
        li = wi * zh % R_MOD * pow(n * (42 - wi) % R_MOD, -1, R_MOD) % R_MOD
        assert np.array_equal(res[i], ol.g1_mul(G, li)), i
    one = np.array([0xd35d438dc58f0d9d, 0x0a78eb28f5c70b3d, 0x666ea36f7879462c, 0x0e0a77c19a07df2f], dtype=np.uint64)
    jac = np.concatenate([res, np.tile(one, (n, 1))], axis=1)
    jac[~res.any(axis=1), 8:] = 0
    assert np.array_equal(pa.g1_sum_jacobian(jac), G)              # sum_i L_i(tau) = 1
    # and the Lagrange key commits evaluations to what the monomial key commits coefficients to
    vals = _rand_fr(n, 77)
    coeffs = torch.from_numpy(vals.view(np.int64)).to("cuda:0")
    ctx.ntt_dev(coeffs, log_n, inverse=True)
    ctx.synchronize()
    mono = ctx.msm_dev(coeffs, n)
    lag = pa.Context(0)
    lag.srs_set_dev(out, n)
    dv = torch.from_numpy(vals.view(np.int64)).to("cuda:0")
    torch.cuda.synchronize()
    assert np.array_equal(lag.msm_dev(dv, n), mono)
    lag.close(); ctx.close()


def test_prove_with_lagrange_key_2pow20():
    """`prove -l` at the headline domain (src/plonk.rs:138-146, commit_using_values): the Lagrange-form key made by the
    G1 iNTT of dump-lagrange is resident next to the monomial one; the wire and grand-product commitments come from
    evaluations and the proof bytes are those of the monomial-only path; the host verifier accepts them"""
    import torch
    import plonkit_amd as pa
    log_n = 20
    n = 1 << log_n
    ctx = pa.Context(0)
    ctx.srs_generate(n, 0, 42)
    circ = pa.Circuit.synthetic(n - 2)
    setup = pa.SetupForProver(ctx, circ)
    want = setup.prove(circ)
    lag = torch.zeros((n, 8), dtype=torch.int64, device="cuda:0")
    ctx.g1_intt_srs_dev(log_n, lag.data_ptr())
    ctx.synchronize()
    ctx.srs_lagrange_set_dev(lag.data_ptr(), n)
    got = setup.prove(circ)
    assert got == want
    assert pa.verify(setup.verification_key_bytes(pa.crs42_g2_bytes()), got)
    ctx.srs_lagrange_clear()
    setup.close(); circ.close(); ctx.close()


def test_cli_binary_round_trip_2pow20(tmp_path):
    """the reference's file path at the headline size (src/r1cs_file.rs:100-154, src/reader.rs:124-218,
    src/bin/main.rs:334-343,384-437,484-504): `.r1cs` + `.wtns` written in circom's binary formats (115 MB + 33 MB),
    `plonkit setup -p 20`, `export-verification-key`, `prove`, `verify` — all through the binary; the proof file equals the
    bytes the library produces in-process from the same circuit object, and plk_verify accepts it"""
    import subprocess
    import plonkit_amd as pa
    cli = os.path.join(os.path.dirname(pa.lib_path()), "plonkit")
    log_n = 20
    circ = pa.Circuit.synthetic((1 << log_n) - 2)
    f = lambda name: str(tmp_path / name)
    r1cs_b, wtns_b = circ.export("r1cs"), circ.export("wtns")
    open(f("c.r1cs"), "wb").write(r1cs_b)
    open(f("w.wtns"), "wb").write(wtns_b)
    # the exported files parse back to a circuit that exports the same bytes (loader <-> writer round trip)
    back = pa.Circuit(r1cs_b, False, wtns_b, False)
    assert back.export("r1cs") == r1cs_b and back.export("wtns") == wtns_b
    run = lambda *a: subprocess.run([cli] + list(a), stderr=subprocess.PIPE, timeout=600)
    assert run("setup", "-p", str(log_n), "-m", f("key.bin")).returncode == 0
    assert os.path.getsize(f("key.bin")) == 8 + 64 * (1 << log_n) + 8 + 256
    assert run("export-verification-key", "-m", f("key.bin"), "-c", f("c.r1cs"), "-v", f("vk.bin")).returncode == 0
    r = run("prove", "-m", f("key.bin"), "-c", f("c.r1cs"), "-w", f("w.wtns"), "-p", f("proof.bin"), "-j", f("proof.json"), "-i", f("public.json"))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert run("verify", "-p", f("proof.bin"), "-v", f("vk.bin")).returncode == 0
    vk, proof = open(f("vk.bin"), "rb").read(), open(f("proof.bin"), "rb").read()
    assert pa.verify(vk, proof)
    ctx = pa.Context(0)
    ctx.srs_generate(1 << log_n, 0, 42)
    setup = pa.SetupForProver(ctx, back)
    assert setup.prove(back) == proof and setup.verification_key_bytes(pa.crs42_g2_bytes()) == vk
    from oracle import plonk_oracle as po
    P = po.read_proof(proof)
    P.linearization_polynomial_at_z = (P.linearization_polynomial_at_z + 1) % R_MOD
    open(f("bad.bin"), "wb").write(po.write_proof(P))
    assert run("verify", "-p", f("bad.bin"), "-v", f("vk.bin")).returncode == 144            # exit(400) truncated (src/bin/main.rs:432-437)
    setup.close(); back.close(); circ.close(); ctx.close()


def test_two_ranks_sliced_srs_2pow20():
    """configs[2] at the size one test GPU can host twice: two ranks (sharing device 0, gloo) with half of the 2^20
    key each reproduce the single-GPU verification key and proof bytes"""
    from tests.test_gpu_sharded_prove import test_two_ranks_produce_the_single_gpu_proof as run
    run(20, False)


def test_dense_circuit_at_the_2pow22_domain(ctx24):
    """the dense synthetic circuit (12-term linear combinations: chains of up to four partial sums per constraint, 3.4 M
    temporaries evaluated run by run on the device) at the 2^22 domain: all 11 commitments non-trivial, the host verifier
    (real pairing) accepts, a tampered evaluation is rejected, and the host evaluation of the temporaries
    (PLK_WITNESS_TMP_HOST is read once per process, so: the same proof at 2^16 in a subprocess) gives the same bytes.
    PARITY UNPINNED (chaining rule)."""
    import subprocess
    import sys
    import plonkit_amd as pa
    from plonkit_amd.prover_bench import nonempty_commitments
    log_n = 22
    ctx = ctx24                                                          # (the 2^24-point key covers the 2^22 domain)
    ctx.srs_lagrange_clear()
    circ = pa.Circuit.synthetic_ex((1 << log_n) - 2, lc_terms=12)
    setup = pa.SetupForProver(ctx, circ)
    assert setup.domain_size == 1 << log_n
    vk, proof = setup.verification_key_bytes(pa.crs42_g2_bytes()), setup.prove(circ)
    assert nonempty_commitments(proof) == 11 and pa.verify(vk, proof)
    bad = bytearray(proof); bad[-300] ^= 1
    assert not pa.verify(vk, bytes(bad))
    setup.close(); circ.close()
    code = r"""
import hashlib, plonkit_amd as pa
ctx = pa.Context(0); ctx.srs_generate(1 << 16, 0, 42)
circ = pa.Circuit.synthetic_ex((1 << 16) - 2, lc_terms=12)
print("H", hashlib.sha256(pa.SetupForProver(ctx, circ).prove(circ)).hexdigest())
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for env in ({}, {"PLK_WITNESS_TMP_HOST": "1"}):
        r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
        assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
        outs.append([ln for ln in r.stdout.splitlines() if ln.startswith("H ")][0])
    assert outs[0] == outs[1]
