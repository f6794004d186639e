"""-m gpu: parity of the HIP hot path (through the C ABI) against the CPU oracle, bit-exact.

Covers the reference's kernels on this path: Polynomial::{fft,ifft,coset_fft,icoset_fft}, coset_lde(4)
and commit_using_monomials/dense_multiexp (call sites src/plonk.rs:104,122-124,132-176)."""
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle_lib as ol
from oracle.oracle_lib import R_MOD


@pytest.fixture(scope="module")
def ctx():
    import plonkit_amd as pa
    c = pa.Context(0)
    yield c
    c.close()


def _rand_fr(n, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)          # < 2^252 < r: valid Montgomery residues
    return a


@pytest.mark.parametrize("log_n", [1, 2, 3, 5, 8, 10, 11, 12, 13, 14, 16, 18])
def test_ntt_matches_oracle(ctx, log_n):
    a = _rand_fr(1 << log_n, 100 + log_n)
    assert np.array_equal(ctx.ntt(a, log_n), ol.ntt(a, log_n))
    assert np.array_equal(ctx.ntt(a, log_n, inverse=True), ol.ntt(a, log_n, inverse=True))


@pytest.mark.parametrize("log_n", [11, 13, 14, 15, 16, 18, 20])
def test_ntt_extreme_residues(ctx, log_n):
    """the lazy bounds of the butterflies at their worst inputs (random vectors of `_rand_fr` stay below 2^252): every element r - 1, r - 1
    alternating with 0 at stride 1 and at the stride of the first pass, residues within 2^16 of r, and all zeros — plain, inverse and
    coset transforms against the oracle.  The sizes cover every pass shape: the old kernels (2^11, 6-bit passes of 2^13) and the wave-owned
    ones at 7 + 7, 8 + 7, 8 + 8, 9 + 9 and 10 + 10 bits."""
    n = 1 << log_n
    top = np.array(ol.int_to_limbs(R_MOD - 1), dtype=np.uint64)
    cases = [np.tile(top, (n, 1))]
    alt = np.zeros((n, 4), dtype=np.uint64); alt[::2] = top
    cases.append(alt)
    blk = np.zeros((n, 4), dtype=np.uint64); blk[(np.arange(n) >> (log_n // 2)) & 1 == 1] = top
    cases.append(blk)
    rng = np.random.default_rng(log_n)
    near = ol.ints_to_array([R_MOD - 1 - int(x) for x in rng.integers(0, 1 << 16, size=min(n, 1 << 12))])
    cases.append(np.tile(near, (n // near.shape[0], 1)))
    cases.append(np.zeros((n, 4), dtype=np.uint64))
    for a in cases:
        a = np.ascontiguousarray(a)
        assert np.array_equal(ctx.ntt(a, log_n), ol.ntt(a, log_n))
        assert np.array_equal(ctx.ntt(a, log_n, inverse=True), ol.ntt(a, log_n, inverse=True))
        assert np.array_equal(ctx.ntt(a, log_n, coset=ol.fr_mont(7)), ol.ntt(a, log_n, coset=7))


@pytest.mark.parametrize("log_n", [3, 9, 12, 15, 17])
def test_coset_ntt_and_roundtrip(ctx, log_n):
    a = _rand_fr(1 << log_n, 7 + log_n)
    g = ol.fr_mont(7)
    f = ctx.ntt(a, log_n, coset=g)
    assert np.array_equal(f, ol.ntt(a, log_n, coset=7))
    assert np.array_equal(ctx.ntt(f, log_n, inverse=True, coset=g), a)
    other = 0x123456789abcdef
    assert np.array_equal(ctx.ntt(a, log_n, inverse=True, coset=ol.fr_mont(other)), ol.ntt(a, log_n, inverse=True, coset=other))


@pytest.mark.parametrize("log_n", [20, 22])
def test_ntt_large_properties(ctx, log_n):
    """full BASELINE size: oracle parity at 2^20, plus size-independent properties (round trip,
    linearity, evaluation at a point)."""
    n = 1 << log_n
    a, b = _rand_fr(n, 1), _rand_fr(n, 2)
    fa = ctx.ntt(a, log_n)
    if log_n <= 20:
        assert np.array_equal(fa, ol.ntt(a, log_n))
    assert np.array_equal(ctx.ntt(fa, log_n, inverse=True), a)
    fb = ctx.ntt(b, log_n)
    assert np.array_equal(ctx.ntt(ol.vadd(a, b), log_n), ol.vadd(fa, fb))
    w = ol.omega(log_n)
    for k in (0, 1, 12345, n - 1):
        assert ol.fr_ints(fa[k:k + 1])[0] == ol.poly_eval(a, pow(w, k, R_MOD))


@pytest.mark.parametrize("log_n", [3, 10, 14])
def test_lde4(ctx, log_n):
    a = _rand_fr(1 << log_n, 40 + log_n)
    ext = np.zeros((4 << log_n, 4), dtype=np.uint64)
    ext[:1 << log_n] = a
    assert np.array_equal(ctx.lde4(a, log_n), ol.ntt(ext, log_n + 2, coset=7))


@pytest.mark.parametrize("log_n,count", [(1, 1), (3, 2), (10, 4), (11, 1), (12, 5), (16, 4), (20, 4)])
def test_lde4_coset_major(ctx, log_n, count):
    """the layout of the prover's round 3 (lde4cm_batch_dev: four n-point coset transforms per polynomial, up to four
    polynomials per launch): out[k*n + r] must be the natural-order extension's element 4r + k — against the oracle's
    coset NTT of the zero-padded vector up to 2^16 (bit-exact), against plk_lde4_dev at 2^20"""
    import torch
    n = 1 << log_n
    polys = [_rand_fr(n, 100 * log_n + p) for p in range(count)]
    d_in = [torch.from_numpy(a.view(np.int64)).to("cuda:0") for a in polys]
    d_out = [torch.zeros((4 * n, 4), dtype=torch.int64, device="cuda:0") for _ in range(count)]
    torch.cuda.synchronize()
    ctx.lde4_coset_major_dev(d_in, log_n, d_out)
    ctx.synchronize()
    for p in range(count):
        got = d_out[p].cpu().numpy().view(np.uint64).reshape(4, n, 4)
        if log_n <= 16:
            ext = np.zeros((4 * n, 4), dtype=np.uint64)
            ext[:n] = polys[p]
            nat = ol.ntt(ext, log_n + 2, coset=7)
        else:
            d_nat = torch.zeros((4 * n, 4), dtype=torch.int64, device="cuda:0")
            ctx.lde4_dev(d_in[p], log_n, d_nat)
            ctx.synchronize()
            nat = d_nat.cpu().numpy().view(np.uint64)
        nat = nat.reshape(n, 4, 4)                                   # [r][k][limb]
        for k in range(4):
            assert np.array_equal(got[k], nat[:, k, :]), (p, k)


@pytest.mark.parametrize("log_n", [1, 3, 9, 10, 11, 12, 16, 20])
def test_icoset4_coset_major(ctx, log_n):
    """the coset iNTT of the prover's round 3 (icoset4cm_dev + k_icoset_combine): a random polynomial of 4n coefficients is
    evaluated on 7*<omega_4n> by the natural-order coset NTT (pinned to the oracle above), its values are put in coset-major
    order on the host (position k*n + r <- natural index 4r + k), and the in-place transform must give the coefficients
    back, canonical — bit-exact"""
    import torch
    n = 1 << log_n
    coef = _rand_fr(4 * n, 700 + log_n)
    d = torch.from_numpy(coef.view(np.int64)).to("cuda:0")
    g = ol.fr_vec([7])[0]
    ctx.ntt_dev(d, log_n + 2, inverse=False, coset=g)
    ctx.synchronize()
    nat = d.cpu().numpy().view(np.uint64).reshape(n, 4, 4)                # [r][k][limb]
    if log_n <= 14:
        assert np.array_equal(nat.reshape(4 * n, 4), ol.ntt(coef, log_n + 2, coset=7))
    cm = np.ascontiguousarray(nat.transpose(1, 0, 2)).reshape(4 * n, 4)    # [k][r]
    dcm = torch.from_numpy(cm.view(np.int64)).to("cuda:0")
    torch.cuda.synchronize()
    ctx.icoset4_coset_major_dev(dcm, log_n)
    ctx.synchronize()
    # the input was a raw 252-bit residue < r: already canonical, so equality is exact
    assert np.array_equal(dcm.cpu().numpy().view(np.uint64), coef)


def test_ntt_errors(ctx):
    import plonkit_amd as pa
    with pytest.raises(pa.PlkError) as e:
        pa._lib._check(pa.lib().plk_ntt_dev(ctx._h, None, 29, 0, None, None))
    assert e.value.code in (1, 2)


# ------------------------------------------------------------------------------------ MSM
@pytest.fixture(scope="module")
def srs16():
    return ol.crs42(1 << 16)


def _trapdoor(ks):
    """MSM(s, crs_42) = (sum s_i 42^i) * G  (the tau = 42 known-answer oracle)."""
    acc, p = 0, 1
    for k in ks:
        acc = (acc + k * p) % R_MOD
        p = p * 42 % R_MOD
    return ol.g1_mul(ol.g1_generator(), acc)


@pytest.mark.parametrize("n", [1, 2, 8, 33, 255, 256, 1000, 4095, 4096, 5000, 1 << 15, 40000, 1 << 16])
def test_msm_matches_oracle(ctx, srs16, n):
    ctx.srs_upload(srs16)
    rng = random.Random(n)
    ks = [rng.randrange(R_MOD) for _ in range(n)]
    got = ctx.msm(ol.fr_vec(ks))
    assert np.array_equal(got, _trapdoor(ks))
    if n <= 5000:
        assert np.array_equal(got, ol.msm(srs16[:n], ol.fr_vec(ks)))


def test_msm_base_offset(ctx, srs16):
    ctx.srs_upload(srs16)
    rng = random.Random(5)
    ks = [rng.randrange(R_MOD) for _ in range(6000)]
    assert np.array_equal(ctx.msm(ol.fr_vec(ks), base_offset=1000), ol.msm(srs16[1000:7000], ol.fr_vec(ks)))


@pytest.mark.parametrize("kind", ["zeros", "ones", "minus_one", "witness_like", "small", "one_hot"])
@pytest.mark.parametrize("n", [300, 1 << 14, 50000])
def test_msm_scalar_distributions(ctx, srs16, kind, n):
    """the four distributions of SURVEY.md §8(d) + degenerate ones; hot buckets and all-zero output (300 / 2^14 terms: the short path of msm_small.hip;
    50000: the 2^20-shaped pipeline with bucket-owning lanes for evenly filled tasks, equal pieces for the hot ones, and the quad reductions)"""
    ctx.srs_upload(srs16)
    rng = random.Random(11)
    if kind == "zeros":
        ks = [0] * n
    elif kind == "ones":
        ks = [1] * n
    elif kind == "minus_one":
        ks = [R_MOD - 1] * n
    elif kind == "witness_like":
        ks = [0 if rng.random() < 0.5 else (rng.randrange(1 << 16) if rng.random() < 0.5 else rng.randrange(R_MOD)) for _ in range(n)]
    elif kind == "small":
        ks = [rng.randrange(4) for _ in range(n)]
    else:
        ks = [0] * n
        ks[n // 3] = R_MOD - 2
    got = ctx.msm(ol.fr_vec(ks))
    assert np.array_equal(got, _trapdoor(ks))
    if kind == "zeros":
        import plonkit_amd as pa
        assert pa.g1_to_bytes(got) == b"\x40" + b"\x00" * 63


@pytest.mark.parametrize("n", [64, 300, 4096, 8192, 10000])
def test_msm_short_path_list_overflow(ctx, srs16, n):
    """short commitments (msm_small.hip, <= 2^15 terms): a bucket list holds as many entries as its workgroup has terms (64 .. 256); a constant column whose scalar carries
    the SAME digit in several windows (3 * (1 + 2^17 + .. + 2^68): five windows put five entries each into lo bucket 3, and 2^8-multiples
    do the same to a hi bucket) overflows it — the overflow flag makes msm_finish_batch run the ordinary pipeline (or, below 4096 terms,
    the per-term double-and-add) on the same inputs.  Also a mix: half the column constant, half uniform."""
    ctx.srs_upload(srs16)
    rep = sum(3 << (17 * w) for w in range(5)) + sum((7 << 8) << (17 * w) for w in range(6, 11))
    ks = [rep] * n
    assert np.array_equal(ctx.msm(ol.fr_vec(ks)), _trapdoor(ks))
    rng = random.Random(n)
    ks = [rep if i % 2 else rng.randrange(R_MOD) for i in range(n)]
    assert np.array_equal(ctx.msm(ol.fr_vec(ks)), _trapdoor(ks))
    ks = [1] * n                                                    # one entry per term, all in lo bucket 1: exactly the capacity of its list
    assert np.array_equal(ctx.msm(ol.fr_vec(ks)), _trapdoor(ks))


def test_msm_duplicate_and_opposite_bases(ctx, srs16):
    """add == double and P + (-P) inside one bucket; infinity bases are skipped"""
    p = srs16[3]
    bases = np.stack([p] * 3000 + [ol.g1_neg(p)] * 3000 + [np.zeros(8, dtype=np.uint64)] * 144)
    ctx.srs_upload(bases)
    ks = [5] * 3000 + [5] * 3000 + [77] * 144
    assert ol.g1_is_inf(ctx.msm(ol.fr_vec(ks)))
    ks = [9] * 3000 + [1] * 3000 + [77] * 144
    assert np.array_equal(ctx.msm(ol.fr_vec(ks)), ol.g1_mul(p, 8 * 3000))
    ctx.srs_upload(bases[:300])
    assert np.array_equal(ctx.msm(ol.fr_vec([7] * 300)), ol.g1_mul(p, 7 * 300))


@pytest.mark.parametrize("n,count", [(100, 3), (5000, 4), (1 << 15, 11), (50000, 2), (1 << 16, 4)])
def test_msm_batch_matches_single(ctx, srs16, n, count):
    """batched commitments (one pass of the kernels) == the same commitments one by one == trapdoor"""
    import torch
    ctx.srs_upload(srs16)
    ts, want = [], []
    for k in range(count):
        s = _rand_fr(n, 1000 + k)
        if k == 1:
            s[: n // 2] = 0                                  # a sparse vector inside the batch
        ts.append(torch.from_numpy(s.view(np.int64)).to("cuda:0"))
        want.append(_trapdoor(ol.fr_ints(s)))
    got = ctx.msm_batch_dev(ts, n)
    for k in range(count):
        assert np.array_equal(got[k], want[k])
        assert np.array_equal(got[k], ctx.msm_dev(ts[k], n))


def test_msm_errors(ctx, srs16):
    import plonkit_amd as pa
    ctx.srs_upload(srs16[:100])
    with pytest.raises(pa.PlkError) as e:
        ctx.msm(ol.fr_vec([1] * 101))
    assert e.value.code == 3


def test_msm_2pow20_trapdoor(ctx):
    """BASELINE config 2 size: 2^20 uniform scalars against the O(N) trapdoor answer."""
    n = 1 << 20
    srs = ol.crs42(n)
    ctx.srs_upload(srs)
    s = _rand_fr(n, 99)
    ks = ol.fr_ints(s)
    assert np.array_equal(ctx.msm(s), _trapdoor(ks))


@pytest.mark.parametrize("log_n", [21, 22])
def test_msm_above_2pow20_on_a_fresh_context(log_n):
    """above 2^20 terms an entry can address only 8 / 4 of the 16 shifted SRS copies (24-bit field), so the
    commitment uses every 2nd / 4th copy and 2 / 4 bucket sets; first MSM of a fresh context (the table is
    built for this size), SRS generated on the GPU, checked against the tau = 42 trapdoor answer."""
    import plonkit_amd as pa
    c = pa.Context(0)
    n = 1 << log_n
    c.srs_generate(n, 0, 42)
    s = _rand_fr(n, 1000 + log_n)
    s[::3] = 0                                                   # witness-like: a third of the scalars vanish
    assert np.array_equal(c.msm(s), _trapdoor(ol.fr_ints(s)))
    assert np.array_equal(c.msm(s[: n // 2 + 7], base_offset=5), ol.g1_mul(_trapdoor(ol.fr_ints(s[: n // 2 + 7])), pow(42, 5, R_MOD)))
    c.close()


def test_commitments_in_flight(ctx, srs16):
    """plk_msm_g1_enqueue_dev / plk_msm_g1_finish are a FIFO of depth three (three scratch sets, three streams): results
    come back in order and equal the one-at-a-time results; a fourth enqueue is refused until something is finished;
    slots are reused in any interleaving of enqueue and finish; every pipeline depth of commit_stream gives the same."""
    import torch
    import plonkit_amd as pa
    from plonkit_amd.sharded import ShardedMsm
    ctx.srs_upload(srs16)
    n = 1 << 16
    dev = torch.device("cuda:0")
    vecs = [torch.from_numpy(_rand_fr(n, 500 + k).view(np.int64)).to(dev) for k in range(7)]
    torch.cuda.synchronize()
    single = [np.asarray(ctx.msm_dev(v, n)) for v in vecs]
    fin = lambda: pa.g1_sum_jacobian(ctx.msm_finish())
    ctx.msm_enqueue_dev(vecs[0], n)
    ctx.msm_enqueue_dev(vecs[1], n)
    ctx.msm_enqueue_dev(vecs[2], n)
    with pytest.raises(pa.PlkError):
        ctx.msm_enqueue_dev(vecs[3], n)
    a = fin()
    ctx.msm_enqueue_dev(vecs[3], n)                               # takes the slot commitment 0 left
    b = fin()
    c = fin()
    ctx.msm_enqueue_dev(vecs[4], n)
    ctx.msm_enqueue_dev(vecs[5], n)
    d, e = fin(), fin()
    ctx.msm_enqueue_dev(vecs[6], n)
    f, g = fin(), fin()
    for got, want in zip((a, b, c, d, e, f, g), single):
        assert np.array_equal(got, want)
    with pytest.raises(pa.PlkError):
        ctx.msm_finish()                                          # nothing in flight
    ctx.msm_enqueue_dev(vecs[0], n)                               # g1_intt borrows a slot's scratch: refused while one is in flight
    with pytest.raises(pa.PlkError):
        ctx.g1_intt(srs16[:8], 3)
    assert np.array_equal(fin(), single[0])
    for depth in (1, 2, 3):
        got = list(ShardedMsm(ctx, None, dev).commit_stream(iter(vecs), n, depth=depth))
        assert len(got) == 7 and all(np.array_equal(g, s) for g, s in zip(got, single)), depth
    assert all(np.array_equal(np.asarray(g), s) for g, s in zip(ctx.msm_batch_dev(vecs, n), single))


@pytest.fixture(scope="module")
def srs19_ctx():
    import plonkit_amd as pa
    c = pa.Context(0)
    c.srs_generate(1 << 19, 0, 42)
    yield c
    c.close()


@pytest.mark.parametrize("kind", ["ones", "minus_one", "witness_like", "small", "top_bits"])
def test_msm_17bit_windows_scalar_distributions(srs19_ctx, kind):
    """the same distributions on the path commitments of >= 2^19 terms take: 15 windows of 17 bits against the 15
    shifted table copies, one bucket set, hot buckets folded by msm_fold_hot (repeated scalars), unsigned top window
    (scalars just below r), two commitments (a batch of two) sharing the kernels"""
    n = 1 << 19
    rng = random.Random(23)
    if kind == "ones":
        ks = [1] * n
    elif kind == "minus_one":
        ks = [R_MOD - 1] * n
    elif kind == "witness_like":
        ks = [0 if rng.random() < 0.5 else (rng.randrange(1 << 16) if rng.random() < 0.5 else rng.randrange(R_MOD)) for _ in range(n)]
    elif kind == "small":
        ks = [rng.randrange(4) for _ in range(n)]
    else:
        ks = [R_MOD - 1 - rng.randrange(1 << 20) for _ in range(n)]           # top window at its maximum, carries everywhere
    s = ol.fr_vec(ks)
    want = _trapdoor(ks)
    assert np.array_equal(srs19_ctx.msm(s), want)
    import torch
    d = torch.from_numpy(s.view(np.int64)).to("cuda:0")
    torch.cuda.synchronize()
    pair = srs19_ctx.msm_batch_dev([d, d], n)
    assert np.array_equal(np.asarray(pair[0]), want) and np.array_equal(np.asarray(pair[1]), want)
    # a batch of three or more takes the 16-lanes-per-task shape of the bucket reduction (msm_fold_hot / msm_task_reduce,
    # four buckets per lane): the same vector, the vector reversed and the same vector again
    rev = ks[::-1]
    d2 = torch.from_numpy(ol.fr_vec(rev).view(np.int64)).to("cuda:0")
    torch.cuda.synchronize()
    trio = srs19_ctx.msm_batch_dev([d, d2, d], n)
    assert np.array_equal(np.asarray(trio[0]), want) and np.array_equal(np.asarray(trio[2]), want)
    assert np.array_equal(np.asarray(trio[1]), _trapdoor(rev))


def test_msm_differential_fuzz():
    """tools/msm_fuzz.py: random lengths (1 .. 2^17 - 3, around the 4096 threshold), SRS offsets, batch sizes, scalar
    distributions and call styles (single, batched, two in flight) against the tau = 42 trapdoor answer"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "msm_fuzz.py"), "24", "11"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "mismatches: 0" in r.stdout


def test_batches_in_flight(ctx, srs16):
    """plk_msm_g1_enqueue_batch_dev / plk_msm_g1_finish_batch: the two-slot FIFO with a batch of vectors per slot (the shape
    of the prover's rounds); results equal the one-at-a-time commitments, a finish with the wrong batch size is refused and
    leaves the batch in flight"""
    import torch
    import plonkit_amd as pa
    from plonkit_amd.sharded import ShardedMsm
    ctx.srs_upload(srs16)
    n = 1 << 15
    dev = torch.device("cuda:0")
    vecs = [torch.from_numpy(_rand_fr(n, 900 + k).view(np.int64)).to(dev) for k in range(6)]
    torch.cuda.synchronize()
    single = [np.asarray(ctx.msm_dev(v, n)) for v in vecs]
    ctx.msm_enqueue_batch_dev(vecs[:4], n)
    ctx.msm_enqueue_batch_dev(vecs[4:], n)
    with pytest.raises(pa.PlkError):
        ctx.msm_finish_batch(3)                                   # the batch in flight holds four
    a = ctx.msm_finish_batch(4)
    b = ctx.msm_finish_batch_sharded(2)                           # no combiner installed: plain affine results
    assert all(np.array_equal(pa.g1_sum_jacobian(a[k]), single[k]) for k in range(4))
    assert all(np.array_equal(b[k], single[4 + k]) for k in range(2))
    msm = ShardedMsm(ctx, None, dev)
    outs = list(msm.commit_batches([vecs[:3], vecs[3:], vecs[:1]], n))
    assert [o.shape[0] for o in outs] == [3, 3, 1]
    got = [o[k] for o in outs for k in range(o.shape[0])]
    assert all(np.array_equal(g, s) for g, s in zip(got, single + single[:1]))


def test_ntt_twiddle_table_cap_evicts_idle_tables_and_composes_when_full():
    """ntt.hip keeps the inter-pass twiddles of a (direction, digit plan) as a table, under a cap (PLK_NTT_DIRECT_CAP_MB).  With a
    64 MB cap: 2^20 transforms (32 MB per table, forward + inverse fill the cap) stay bit-exact against the oracle when a 2^21
    transform (64 MB: does not fit beside tables in recent use -> its passes compose their twiddles), then >= 96 further requests
    (the idle 2^20 tables become evictable and are dropped for the 2^21 one), then 2^20 again (rebuilt) are interleaved; and
    PLK_NTT_DIRECT=0 (no tables at all) gives the same bits."""
    import subprocess
    import sys
    code = r"""
import os, sys
import numpy as np, torch
import plonkit_amd as pa
from oracle import oracle_lib as ol
ctx = pa.Context(0)
def run(log_n, inverse, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 1 << 62, size=(1 << log_n, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 60) - 1)
    t = torch.from_numpy(a.view(np.int64)).to("cuda:0")
    ctx.ntt_dev(t, log_n, inverse=inverse); ctx.synchronize()
    return a, t.cpu().numpy().view(np.uint64)
checks = []
for step, (log_n, inv) in enumerate([(20, False), (20, True), (21, False), (20, False)] + [(21, True)] * 100 + [(21, False), (20, True), (20, False)]):
    a, got = run(log_n, inv, step)
    if step < 6 or step > 100:
        checks.append(bool(np.array_equal(got, ol.ntt(a, log_n, inverse=inv))))
print("CHECKS", all(checks), len(checks))
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in ({"PLK_NTT_DIRECT_CAP_MB": "64"}, {"PLK_NTT_DIRECT": "0"}):
        r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=600, env=dict(os.environ, **extra))
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        assert "CHECKS True" in r.stdout, r.stdout[-500:]
