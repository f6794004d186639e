"""Pins the oracle (oracle/plonk_oracle.py + oracle/c) to the reference's own golden vectors:
src/tests.rs:14 (analyse), :31-46 (vk.bin), :49-73 (proof.bin), :76-81 (verify),
src/r1cs_file.rs:162-243 (binary r1cs sample).  No GPU."""
import os
import struct

import pytest

from oracle import oracle_lib as ol, plonk_oracle as po
from oracle.oracle_lib import R_MOD


def _g(golden_dir, name, mode="rb"):
    return open(os.path.join(golden_dir, name), mode).read()


@pytest.fixture(scope="module")
def simple(golden_dir):
    r1cs = po.load_r1cs_json(os.path.join(golden_dir, "circuit.r1cs.json"))
    w = po.load_witness_json(os.path.join(golden_dir, "witness.json"))
    return r1cs, w


def test_crs_roundtrip(golden_dir, golden_crs):
    data = _g(golden_dir, "setup_2pow10.key")
    assert len(data) == 8 + 64 * 1024 + 8 + 256
    assert po.write_crs(golden_crs) == data
    assert all(ol.g1_on_curve(golden_crs.g1[i]) for i in (0, 1, 500, 1023))


def test_analyse_matches_reference_string(golden_dir, simple):
    assert po.analyse(simple[0]) == _g(golden_dir, "analyse.json", "r")


def test_vk_bytes(golden_dir, golden_crs, simple):
    S = po.setup(simple[0])
    assert (S.n, S.N) == (7, 8)
    assert po.write_vk(po.make_verification_key(S, golden_crs)) == _g(golden_dir, "vk.bin")


def test_proof_bytes_and_challenges(golden_dir, golden_crs, simple):
    P, dbg = po.prove(simple[0], simple[1], golden_crs, return_debug=True)
    assert po.write_proof(P) == _g(golden_dir, "proof.bin")
    assert dbg["beta"] == 0x0f72cf563829c88d02442b32aa5bc8b0aff226697faa846756e813710804a058
    assert dbg["gamma"] == 0x19f776d072bc5715a7fb2a727344f31eda1aa1b214c0b8b7f0bf9a8fed192264
    assert dbg["alpha"] == 0x04dcc892670ebc7d73ec75daf729667f55ab1fb7119660e76fc1230207b6c9b5
    assert dbg["z"] == 0x0913d2eba66540a79bf6ea941e38f856105c5cfe6dadb5738a2b895b337dc63e
    assert dbg["v"] == 0x1b49fbb2ccfc097e7d0a05e499dcb39e9861c0240726f81beed8fa082c33e916
    # the d wire is identically zero in this circuit => commitment is the point at infinity (0x40 00..)
    assert ol.g1_is_inf(P.wire_commitments[3]) and ol.g1_is_inf(P.quotient_poly_commitments[3])


def test_verify_golden_and_reject_tampering(golden_dir):
    vk = po.read_vk(_g(golden_dir, "vk.bin"))
    data = _g(golden_dir, "proof.bin")
    P = po.read_proof(data)
    assert po.write_proof(P) == data and po.write_vk(vk) == _g(golden_dir, "vk.bin")
    assert po.verify(vk, P)
    P.wire_values_at_z[1] = (P.wire_values_at_z[1] + 1) % R_MOD
    assert not po.verify(vk, P)
    P = po.read_proof(data)
    P.opening_at_z_proof = ol.g1_add(P.opening_at_z_proof, ol.g1_generator())
    assert not po.verify(vk, P)
    P = po.read_proof(data)
    P.inputs[0] = 36
    assert not po.verify(vk, P)


def test_r1cs_bin_sample(golden_dir):
    data = _g(golden_dir, "r1cs_sample.bin")
    hdr, cons, wmap = po.parse_r1cs_bin(data)
    assert hdr == dict(field_size=32, n_wires=7, n_pub_out=1, n_pub_in=2, n_prv_in=3, n_labels=0x03e8, n_constraints=3)
    assert len(cons) == 3 and len(cons[0][0]) == 2
    assert cons[0][0][0] == (5, 3) and cons[2][1][0] == (0, 6) and len(cons[1][2]) == 0
    assert len(wmap) == 7 and wmap[1] == 3
    r = po.load_r1cs_bin(data)
    assert (r.num_inputs, r.num_aux, r.num_variables) == (4, 3, 7)
    bad = bytearray(data)
    struct.pack_into("<Q", bad, 16, 0x41)          # header section size (src/r1cs_file.rs:245-252)
    with pytest.raises(ValueError):
        po.parse_r1cs_bin(bytes(bad))
    with pytest.raises(ValueError):
        po.parse_r1cs_bin(b"r2cs" + data[4:])


def test_wtns_roundtrip():
    w = [1, 35, 3, 9]
    body = b"".join(v.to_bytes(32, "little") for v in w)
    data = (b"wtns" + struct.pack("<II", 2, 2) + struct.pack("<IQ", 1, 40) + struct.pack("<I", 32) + po.BN254_PRIME_LE
            + struct.pack("<I", len(w)) + struct.pack("<IQ", 2, 32 * len(w)) + body)
    assert po.parse_wtns(data) == w
    with pytest.raises(ValueError):
        po.parse_wtns(b"wtnx" + data[4:])


def _chain_circuit(n_cons, seed=0x706c6f6e6b6974):
    """Synthetic R1CS inside the pinned transpilation subset: alternating
    1-gate (cA*u * cB*v = cC*w) and 2-gate (... = k + c1*p + c2*q) multiplication constraints."""
    rng = po.Xoshiro256ss(seed)
    wit = [1, 0, rng.fr(), rng.fr()]        # ONE, public input (filled last), two seeds
    cons = []
    while len(cons) < n_cons:
        u, v = len(wit) - 1, len(wit) - 2
        ca, cb, cc = rng.fr() or 1, rng.fr() or 1, rng.fr() or 1
        prod = ca * wit[u] % R_MOD * cb % R_MOD * wit[v] % R_MOD
        if len(cons) % 2 == 0:
            wit.append(prod * pow(cc, -1, R_MOD) % R_MOD)
            cons.append(([(u, ca)], [(v, cb)], [(len(wit) - 1, cc)]))
        else:
            k, c1 = rng.fr(), rng.fr() or 1
            p = v
            # prod = k + c1*p + c2*q  with q a fresh wire
            c2 = rng.fr() or 1
            q = (prod - k - c1 * wit[p]) % R_MOD * pow(c2, -1, R_MOD) % R_MOD
            wit.append(q)
            cons.append(([(u, ca)], [(v, cb)], [(0, k), (p, c1), (len(wit) - 1, c2)]))
    wit[1] = wit[-1]
    cons.append(([(1, 1)], [(0, 1)], [(len(wit) - 1, 1)]))     # 1*pub = last wire (constant*LC merge, unpinned shape)
    return po.R1CS(2, len(wit) - 2, len(wit), cons), wit


@pytest.mark.parametrize("n_cons", [5, 40, 300])
def test_synthetic_prove_verify(golden_crs, n_cons):
    r1cs, wit = _chain_circuit(n_cons)
    S = po.setup(r1cs)
    vk = po.make_verification_key(S, golden_crs)
    P = po.prove(r1cs, wit, golden_crs, S)
    assert not ol.g1_is_inf(P.quotient_poly_commitments[2])
    assert po.verify(vk, po.read_proof(po.write_proof(P)))
    P.grand_product_at_z_omega ^= 1
    assert not po.verify(vk, P)


def test_long_lc_chain_proves(golden_crs):
    """d_next chains (UNPINNED transpilation, from recollection) at least yield a sound proof."""
    rng = po.Xoshiro256ss(7)
    wit = [1, 0] + [rng.fr() for _ in range(12)]
    lc = [(i, rng.fr()) for i in range(2, 12)]
    s = sum(c * wit[i] for i, c in lc) % R_MOD
    wit.append(s * wit[13] % R_MOD)
    wit[1] = wit[-1]
    cons = [(lc + [(0, 5)], [(13, 1)], [(14, 1), (13, 5)]), ([(1, 1)], [(0, 1)], [(14, 1)]),
            ([(3, 1), (0, R_MOD - 1)], [(3, 1), (0, 2)], [(15, 1)])]
    wit.append((wit[3] - 1) * (wit[3] + 2) % R_MOD)
    r1cs = po.R1CS(2, len(wit) - 2, len(wit), cons)
    S = po.setup(r1cs)
    P = po.prove(r1cs, wit, golden_crs, S)
    assert po.verify(po.make_verification_key(S, golden_crs), P)


def test_crosscheck_manifest_simple_case_is_the_reference_golden(golden_dir):
    """tools/make_crosscheck_bundle.py (run on an MI355X; its MANIFEST.json is committed as tools/crosscheck_MANIFEST.json) writes
    seven circuits' artefacts for a future comparison with a real `plonkit`.  Six of them are PARITY UNPINNED; the seventh, the
    reference's own `simple` circuit taken through circom's BINARY formats this time, must hash to the reference's committed
    files (src/tests.rs:31-73): key, verification key and proof."""
    import hashlib
    import json
    root = os.path.dirname(golden_dir.rstrip("/"))
    m = json.load(open(os.path.join(os.path.dirname(root), "tools", "crosscheck_MANIFEST.json")))
    sha = lambda name: hashlib.sha256(open(os.path.join(golden_dir, name), "rb").read()).hexdigest()
    assert m["simple"]["vk.bin"] == sha("vk.bin") and m["simple"]["proof.bin"] == sha("proof.bin") and m["simple"]["setup.key"] == sha("setup_2pow10.key")
    assert set(m) == {"simple", "poseidon_12", "poseidon_14", "poseidon_16", "long_lc", "dense_14", "zero_inputs"}


def test_compiled_front_end_equals_the_python_one(golden_dir, golden_crs):
    """the C front end of the oracle (r1cs / wtns parsers, gate synthesis with the witness, gate check, permutation by one
    stable sort: oracle/c/oracle.c + setup_flat) — what bench.py's cpu_baseline proves from — against the Python restatement
    pinned above: same setup polynomials and same proof bytes on the reference's `simple` circuit (its r1cs re-encoded in the
    binary format by the product's exporter), on pinned-subset and dense synthetic circuits and on a Poseidon-shaped one
    (constant x LC merges, 60-term chains); and the two work splits of the MSM give one group element."""
    import json
    import numpy as np
    import plonkit_amd as pa
    from tests.gen import poseidon_like as pl
    cases = []
    c = pa.Circuit.from_files(os.path.join(golden_dir, "circuit.r1cs.json"), os.path.join(golden_dir, "witness.json"))
    cases.append((c.export("r1cs"), c.export("wtns")))
    for lc in (0, 6, 12):
        c = pa.Circuit.synthetic_ex(510, 5 + lc, 0, lc)
        cases.append((c.export("r1cs"), c.export("wtns")))
    ni, nv, cons, wit = pl.build(1, 1056, rp=56)
    js = pl.as_circom_json(ni, nv, cons)
    c = pa.Circuit(json.dumps(js).encode(), True, json.dumps([str(x) for x in wit]).encode(), True)
    cases.append((c.export("r1cs"), c.export("wtns")))
    crs = po.Crs(ol.crs42(1 << 12), golden_crs.g2_raw)
    for k, (raw, wt) in enumerate(cases):
        r1, w = po.load_r1cs_bin(raw), po.parse_wtns(wt)
        rf, wf = po.load_r1cs_flat(raw), ol.wtns_parse(wt)
        assert ol.fr_ints(wf) == [x % ol.R_MOD for x in w]
        S1, S2 = po.setup(r1), po.setup_flat(rf)
        assert (S1.N, S1.num_inputs) == (S2.N, S2.num_inputs)
        for a, b in zip(S1.selectors + S1.sigmas, S2.selectors + S2.sigmas):
            assert np.array_equal(a, b)
        p1 = po.write_proof(po.prove(r1, w, crs, S1))
        ol.MSM_SPLIT[0] = "windows"
        try:
            p2 = po.write_proof(po.prove(rf, wf, crs, S2))
        finally:
            ol.MSM_SPLIT[0] = "chunks"
        assert p1 == p2
        if k == 0:
            assert p1 == open(os.path.join(golden_dir, "proof.bin"), "rb").read()
    bad = bytearray(cases[1][1]); bad[76 + 32 * 5] ^= 1                 # a broken witness entry: the compiled gate check refuses
    with pytest.raises(AssertionError):
        po.prove(po.load_r1cs_flat(cases[1][0]), ol.wtns_parse(bytes(bad)), crs)
    with pytest.raises(ValueError):
        po.load_r1cs_flat(cases[1][0][:200])
