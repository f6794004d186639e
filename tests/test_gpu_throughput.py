"""-m gpu: several proofs in flight on ONE GPU (one context per host thread, one shared setup, the key and its MSM table
shared between the contexts: plk_ctx_share_srs) and the dense synthetic circuit (long linear combinations folded through
the d column: all 11 commitments of a proof non-trivial).

Reference: SetupForProver::prove takes &self (src/plonk.rs:132-176) — re-entrant on the reference's side too; the CI flow
proves the same circuit several times (.github/workflows/integration-test.yml:105-154); long linear combinations are what
every circom circuit feeds the transpiler (src/circom_circuit.rs:114-131, test/test_poseidon_plonk.sh:47-80)."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import plonk_oracle as po
from oracle.oracle_lib import R_MOD


@pytest.fixture(scope="module")
def ctx():
    import plonkit_amd as pa
    c = pa.Context(0)
    yield c
    c.close()


def _oracle_proof(ctx, circ, log_n):
    import plonkit_amd as pa
    r1cs, wit = po.load_r1cs_bin(circ.export("r1cs")), po.parse_wtns(circ.export("wtns"))
    crs = po.Crs(ctx.srs_download(0, 1 << log_n), pa.crs42_g2_bytes())
    S = po.setup(r1cs)
    return po.write_vk(po.make_verification_key(S, crs)), po.write_proof(po.prove(r1cs, wit, crs, S))


@pytest.mark.parametrize("lc_terms", [0, 7])
def test_two_proofs_in_flight_equal_the_oracle(ctx, lc_terms):
    """two host threads, two contexts on device 0 (the second borrows the first's key and MSM table), ONE setup, two
    different witnesses of the same circuit, six proofs each at the 2^12 domain: every proof equals the ORACLE's bytes for
    its witness (so also the sequential ones), for the pinned-subset circuit and for the dense one (parity unpinned)."""
    import plonkit_amd as pa
    log_n = 12
    ctx.srs_generate(1 << log_n, 0, 42)
    ctx.srs_lagrange_clear()
    circs = [pa.Circuit.synthetic_ex((1 << log_n) - 2, witness_seed=ws, lc_terms=lc_terms) for ws in (0, 4242)]
    assert circs[0].export("r1cs") == circs[1].export("r1cs") and circs[0].export("wtns") != circs[1].export("wtns")
    setup = pa.SetupForProver(ctx, circs[0])
    want = []
    for c in circs:
        vk, pr = _oracle_proof(ctx, c, log_n)
        want.append(pr)
    assert want[0] != want[1]
    assert setup.verification_key_bytes(pa.crs42_g2_bytes()) == vk
    other = pa.Context(0)
    other.share_srs_from(ctx)
    ctxs = [ctx, other]
    got = [[], []]
    errs = []
    gate = threading.Barrier(2)

    def worker(k):
        try:
            gate.wait()
            for _ in range(6):
                got[k].append(setup.prove(circs[k], ctx=ctxs[k]))
        except Exception as exc:                                       # noqa: BLE001
            errs.append(repr(exc))
    th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for k in range(2):
        assert len(got[k]) == 6 and all(p == want[k] for p in got[k]), "thread %d produced a proof that differs from the oracle's" % k
    other.close()
    setup.close()
    for c in circs:
        c.close()


def test_first_proofs_of_a_fresh_setup_race_for_the_cached_extensions(ctx):
    """the setup's constant extensions are computed by whichever proof gets there first, under a lock: three threads start
    their FIRST proof of a fresh setup at the same moment (three contexts, the same witness) — identical, verifying bytes"""
    import plonkit_amd as pa
    log_n = 13
    ctx.srs_generate(1 << log_n, 0, 42)
    ctx.srs_lagrange_clear()
    circ = pa.Circuit.synthetic_ex((1 << log_n) - 2, lc_terms=6)
    setup = pa.SetupForProver(ctx, circ)
    others = [pa.Context(0) for _ in range(2)]
    for o in others:
        o.share_srs_from(ctx)
    ctxs = [ctx] + others
    out = [None] * 3
    gate = threading.Barrier(3)

    def worker(k):
        gate.wait()
        out[k] = setup.prove(circ, ctx=ctxs[k])
    th = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert out[0] is not None and out[0] == out[1] == out[2]
    assert pa.verify(setup.verification_key_bytes(pa.crs42_g2_bytes()), out[0])
    for o in others:
        o.close()
    setup.close(); circ.close()


def test_share_srs_ownership_rules(ctx):
    """the lender refuses to replace its key while it is on loan; a borrower that gets a key of its own stops borrowing;
    commitments through a borrowed table equal the owner's"""
    import torch
    import plonkit_amd as pa
    n = 1 << 12
    ctx.srs_generate(n, 0, 42)
    ctx.srs_lagrange_clear()
    b = pa.Context(0)
    b.share_srs_from(ctx)
    assert b.srs_size() == n
    rng = np.random.default_rng(5)
    s = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    s[:, 3] &= np.uint64((1 << 60) - 1)
    assert np.array_equal(b.msm(s), ctx.msm(s))
    with pytest.raises(pa.PlkError) as e:
        ctx.srs_generate(n, 0, 42)                                     # on loan
    assert e.value.code == 1 and "shared" in str(e.value)
    with pytest.raises(pa.PlkError):
        pa.Context(0).share_srs_from(b)                                # a borrower cannot lend
    b.srs_generate(n // 2, 0, 42)                                      # a key of its own: the loan ends
    ctx.srs_generate(n, 0, 42)                                         # ... and the owner is free again
    assert np.array_equal(b.msm(s[: n // 2]), ctx.msm(s[: n // 2]))
    b2 = pa.Context(0)
    b2.share_srs_from(ctx)
    b2.close()                                                         # destroying a borrower returns the loan too
    ctx.srs_generate(n, 0, 42)
    b.close()


def test_lender_destroyed_before_its_borrowers():
    """plk_destroy of a lender while borrowers are alive must not pull the key from under them (round-4 advisor finding): the key
    and the MSM tables outlive the lender until the last borrower returns its loan"""
    import plonkit_amd as pa
    n = 1 << 12
    owner = pa.Context(0)
    owner.srs_generate(n, 0, 42)
    rng = np.random.default_rng(11)
    s = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    s[:, 3] &= np.uint64((1 << 60) - 1)
    want = owner.msm(s)
    b1, b2 = pa.Context(0), pa.Context(0)
    b1.share_srs_from(owner); b2.share_srs_from(owner)
    owner.close()                                                      # the lender goes first
    assert np.array_equal(b1.msm(s), want) and np.array_equal(b2.msm(s), want)
    b1.close()
    assert np.array_equal(b2.msm(s), want)                             # one loan left: the key is still there
    b2.srs_generate(n, 0, 42)                                          # the last loan ends (a key of its own): the orphaned key is freed
    assert np.array_equal(b2.msm(s), want)
    b2.close()


def test_lender_may_install_a_lagrange_key_nobody_borrowed(ctx):
    """the Lagrange-form key of a lender is only frozen while a borrower holds it (round-4 advisor finding): `prove -l` can be
    started on the owner after its throughput contexts were created; a borrowed Lagrange key stays frozen"""
    import plonkit_amd as pa
    from oracle import oracle_lib as ol
    n = 1 << 10
    ctx.srs_generate(n, 0, 42)
    ctx.srs_lagrange_clear()
    b = pa.Context(0)
    b.share_srs_from(ctx)                                              # no Lagrange key on loan
    lag = ol.g1_intt(ol.crs42(n), 10)
    ctx.srs_lagrange_upload(lag)                                       # allowed: nobody holds the (absent) Lagrange key
    ctx.srs_lagrange_clear()
    ctx.srs_lagrange_upload(lag)
    b.close()
    b = pa.Context(0)
    b.share_srs_from(ctx)                                              # now the Lagrange key is on loan too
    with pytest.raises(pa.PlkError):
        ctx.srs_lagrange_clear()
    b.srs_lagrange_clear()                                             # the borrower drops only its Lagrange loan ...
    ctx.srs_lagrange_clear()                                           # ... which frees the owner's Lagrange slot, not its monomial key
    with pytest.raises(pa.PlkError):
        ctx.srs_generate(n, 0, 42)
    b.close()
    ctx.srs_generate(n, 0, 42)


def test_dense_circuit_proves_all_eleven_commitments_at_2pow20(ctx):
    """the dense synthetic circuit at the headline domain (2^20): d, q_d_next and t_3 live — 11 of 11 commitments are
    non-trivial (the pinned-subset circuit: 9) —, the host verifier (real pairing) accepts, tampering with an evaluation
    and with the d-wire commitment is rejected, an unsatisfying witness is refused.  PARITY UNPINNED (chaining rule)."""
    import plonkit_amd as pa
    from plonkit_amd.prover_bench import nonempty_commitments
    log_n = 20
    ctx.srs_generate(1 << log_n, 0, 42)
    ctx.srs_lagrange_clear()
    circ = pa.Circuit.synthetic_ex((1 << log_n) - 2, lc_terms=7)
    setup = pa.SetupForProver(ctx, circ)
    assert setup.domain_size == 1 << log_n
    vk = setup.verification_key_bytes(pa.crs42_g2_bytes())
    proof = setup.prove(circ)
    assert nonempty_commitments(proof) == 11
    assert pa.verify(vk, proof)
    P = po.read_proof(proof)
    P.wire_values_at_z[3] = (P.wire_values_at_z[3] + 1) % R_MOD
    assert not pa.verify(vk, po.write_proof(P))
    P = po.read_proof(proof)
    P.wire_commitments[3] = P.wire_commitments[0]
    assert not pa.verify(vk, po.write_proof(P))
    plain = pa.Circuit.synthetic((1 << log_n) - 2)
    s2 = pa.SetupForProver(ctx, plain)
    assert nonempty_commitments(s2.prove(plain)) == 9
    s2.close(); plain.close()
    setup.close(); circ.close()


@pytest.mark.parametrize("lc_terms,log_n", [(5, 10), (9, 14), (12, 16)])
def test_dense_circuit_matches_oracle(ctx, lc_terms, log_n):
    """dense synthetic circuits at the 2^10 / 2^14 / 2^16 domains: verification key and proof bytes equal the oracle's
    (PARITY UNPINNED for the chaining rule: product and oracle share one recollection of bellman's adaptor)"""
    import plonkit_amd as pa
    from plonkit_amd.prover_bench import nonempty_commitments
    ctx.srs_generate(1 << log_n, 0, 42)
    ctx.srs_lagrange_clear()
    circ = pa.Circuit.synthetic_ex((1 << log_n) - 2, seed=99 + lc_terms, lc_terms=lc_terms)
    setup = pa.SetupForProver(ctx, circ)
    vk, proof = setup.verification_key_bytes(pa.crs42_g2_bytes()), setup.prove(circ)
    assert nonempty_commitments(proof) == 11
    assert pa.verify(vk, proof)
    if log_n <= 14:
        vk_o, proof_o = _oracle_proof(ctx, circ, log_n)
        assert vk == vk_o and proof == proof_o
    else:
        r1cs = po.load_r1cs_bin(circ.export("r1cs"))
        assert circ.analyse() == po.analyse(r1cs)
        assert po.verify(po.read_vk(vk), po.read_proof(proof), tau=42)
    setup.close(); circ.close()


def test_throughput_leg_of_the_bench(ctx):
    """plonkit_amd.prover_bench.throughput (the `prove.throughput` object of the bench line) at a small domain: its own
    byte-identity assertions hold, the figures are present"""
    from plonkit_amd import prover_bench
    ctx.srs_generate(1 << 14, 0, 42)
    ctx.srs_lagrange_clear()
    r = prover_bench.throughput(ctx, 14, in_flight=2, proofs_each=4)
    assert r["in_flight"] == 2 and r["proofs"] == 8 and r["byte_identical_to_sequential"] and r["proofs_per_s"] > 0
    r = prover_bench.throughput(ctx, 13, in_flight=3, proofs_each=3, lc_terms=7)
    assert r["in_flight"] == 3 and r["proofs"] == 9


def test_two_provers_through_the_c_abi_alone(tmp_path):
    """tests/host/two_provers.c: a C program (pthreads, include/plonkit_amd.h, no Python in the proving process) runs two
    contexts on device 0 against one setup with two witnesses — every concurrent proof byte-identical to the sequential one,
    and the lender refuses to replace its key while it is on loan"""
    import os
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not on PATH")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "two_provers")
    libdir = os.path.join(root, "plonkit_amd", "lib")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Werror", "-pthread", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "host", "two_provers.c"), "-o", exe, "-L", libdir, "-lplonkit_amd", "-Wl,-rpath," + libdir])
    r = subprocess.run([exe, "14", "6"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    assert r.stdout.startswith("OK 12 "), r.stdout
