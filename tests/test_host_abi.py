"""CPU-only checks of the product's C ABI library: it loads, exports every symbol the header declares,
and its host half (loaders, transpiler/analyse, transcript, serialisation) agrees with the reference's
golden data and with the oracle.  No compute entry point is called (there is no GPU here)."""
import os
import re
import struct
import subprocess

import numpy as np
import pytest

import plonkit_amd as pa
from oracle import oracle_lib as ol, plonk_oracle as po
from oracle.oracle_lib import R_MOD

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "plonkit_amd.h")).read()
    declared = sorted(set(re.findall(r"\b(plk_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 40
    out = subprocess.check_output(["nm", "-D", "--defined-only", pa.lib_path()]).decode()
    exported = set(re.findall(r" T (plk_[a-z0-9_]+)", out))
    missing = [s for s in declared if s not in exported]
    assert not missing, missing
    L = pa.lib()
    for s in declared:
        getattr(L, s)
    assert b"gfx950" in L.plk_version()


def test_no_cpu_fallback_without_gpu():
    if pa.have_gpu():
        pytest.skip("a GPU is visible")
    with pytest.raises(pa.PlkError) as e:
        pa.Context(0)
    assert e.value.code == 4 and "no CPU fallback" in str(e.value)


def test_keccak_and_transcript_match_oracle():
    for msg in (b"", b"abc", b"x" * 135, b"y" * 136, b"z" * 1000):
        assert pa.keccak256(msg) == ol.keccak256(msg)
    t, o = pa.Transcript(), po.Transcript()
    G = ol.g1_generator()
    for v in (35, 0, R_MOD - 1):
        t.absorb_fr(ol.fr_mont(v)); o.absorb_fr(v)
    t.absorb_g1(G); o.absorb_g1(G)
    t.absorb_g1(np.zeros(8, dtype=np.uint64)); o.absorb_g1(np.zeros(8, dtype=np.uint64))
    for _ in range(3):
        assert ol.fr_ints(t.challenge())[0] == o.challenge()


def test_point_and_scalar_encoding(golden_dir, golden_crs):
    raw = open(os.path.join(golden_dir, "setup_2pow10.key"), "rb").read()
    for i in (0, 1, 777, 1023):
        b = raw[8 + 64 * i: 8 + 64 * i + 64]
        p = pa.g1_from_bytes(b)
        assert np.array_equal(p, golden_crs.g1[i]) and pa.g1_to_bytes(p) == b
    inf = b"\x40" + b"\x00" * 63
    assert not pa.g1_from_bytes(inf).any() and pa.g1_to_bytes(np.zeros(8, dtype=np.uint64)) == inf
    with pytest.raises(pa.PlkError):
        pa.g1_from_bytes((1).to_bytes(32, "big") + (3).to_bytes(32, "big"))     # not on the curve
    with pytest.raises(pa.PlkError):
        pa.fr_from_bytes(R_MOD.to_bytes(32, "big"))
    assert pa.fr_to_bytes(pa.fr_from_bytes((R_MOD - 5).to_bytes(32, "big"))) == (R_MOD - 5).to_bytes(32, "big")
    # partial sums: (P) + (P) + (-2P) = O through the host combiner used by the multi-GPU path
    P = golden_crs.g1[5]
    jac = lambda a: np.concatenate([a, ol.fq_mont(1)]) if a.any() else np.concatenate([ol.fq_mont(1), ol.fq_mont(1), np.zeros(4, dtype=np.uint64)])
    parts = np.stack([jac(P), jac(P), jac(ol.g1_neg(ol.g1_mul(P, 2)))])
    assert not pa.g1_sum_jacobian(parts).any()
    assert np.array_equal(pa.g1_sum_jacobian(parts[:2]), ol.g1_mul(P, 2))


def test_analyse_golden_string(golden_dir):
    c = pa.Circuit.from_files(os.path.join(golden_dir, "circuit.r1cs.json"))
    assert c.analyse() == open(os.path.join(golden_dir, "analyse.json")).read()
    c2 = pa.Circuit.from_files(os.path.join(golden_dir, "circuit.r1cs.json"), os.path.join(golden_dir, "witness.json"))
    assert c2.analyse() == c.analyse()


def test_r1cs_bin_loader_matches_oracle_and_rejects_bad_input(golden_dir):
    data = open(os.path.join(golden_dir, "r1cs_sample.bin"), "rb").read()
    c = pa.Circuit(data, False)
    assert c.analyse() == po.analyse(po.load_r1cs_bin(data))
    for bad in (b"r2cs" + data[4:], data[:100], data[:4] + struct.pack("<I", 2) + data[8:]):
        with pytest.raises(pa.PlkError) as e:
            pa.Circuit(bad, False)
        assert e.value.code == 6
    hdr_bad = bytearray(data)
    hdr_bad[28] ^= 1                                    # prime byte
    with pytest.raises(pa.PlkError):
        pa.Circuit(bytes(hdr_bad), False)


def test_malformed_circuit_files_are_refused(golden_dir):
    """wire ids outside [0, n_wires) and size fields that exceed the file: the reference indexes its
    variable table with the wire id and panics (src/circom_circuit.rs:107-113); here the loaders return PLK_ERR_FORMAT
    instead of writing past the ends of the setup's tables, and nothing unwinds through the C ABI"""
    import json
    data = bytearray(open(os.path.join(golden_dir, "r1cs_sample.bin"), "rb").read())
    r1 = po.load_r1cs_bin(bytes(data))
    n_wires = r1.num_variables
    # locate the first wire id of the constraint section: header (12) + section table walk
    off, secs = 12, {}
    for _ in range(struct.unpack_from("<I", data, 8)[0]):
        t, sz = struct.unpack_from("<IQ", data, off)
        secs[t] = (off + 12, sz)
        off += 12 + sz
    c0 = secs[2][0]
    assert struct.unpack_from("<I", data, c0)[0] >= 1                 # first LC has at least one term
    for wire in (n_wires, 0x7fffffff, 0xffffffff):
        bad = bytearray(data)
        struct.pack_into("<I", bad, c0 + 4, wire)
        with pytest.raises(pa.PlkError) as e:
            pa.Circuit(bytes(bad), False)
        assert e.value.code == 6 and "wire index" in str(e.value)
    ok = bytearray(data)
    struct.pack_into("<I", ok, c0 + 4, n_wires - 1)                   # the largest valid id still loads
    pa.Circuit(bytes(ok), False)
    # header announcing 2^32 - 1 constraints in a 200-byte file: refused before anything is allocated
    h0 = secs[1][0]
    huge = bytearray(data)
    struct.pack_into("<I", huge, h0 + 4 + 32 + 16 + 8, 0xffffffff)    # field_size, prime, 4 x u32, u64 labels, then n_constraints
    with pytest.raises(pa.PlkError) as e:
        pa.Circuit(bytes(huge), False)
    assert e.value.code == 6
    # JSON: wire id >= nVars, and one that only fits 64 bits (it used to be truncated to 32)
    js = json.loads(open(os.path.join(golden_dir, "circuit.r1cs.json")).read())
    for wire in (str(js["nVars"]), str((1 << 32) + 1), str(1 << 70)):
        bad = json.loads(json.dumps(js))
        bad["constraints"][0][0] = {wire: "1"}
        with pytest.raises(pa.PlkError) as e:
            pa.Circuit(json.dumps(bad).encode(), True)
        assert e.value.code == 6
    bad = json.loads(json.dumps(js)); bad["nVars"] = 1 << 40
    with pytest.raises(pa.PlkError):
        pa.Circuit(json.dumps(bad).encode(), True)


def test_point_encoding_is_canonical():
    """pairing_ce refuses an infinity flag with a non-zero remainder and the unflagged (0, 0): so do
    plk_g1_from_bytes, the key parser and the verifier's readers"""
    inf = b"\x40" + b"\x00" * 63
    assert not pa.g1_from_bytes(inf).any()
    for bad in (b"\x40" + b"\x00" * 62 + b"\x01", b"\x41" + b"\x00" * 63, b"\x00" * 64, b"\xc0" + b"\x00" * 63):
        with pytest.raises(pa.PlkError):
            pa.g1_from_bytes(bad)
    g = (1).to_bytes(32, "big") + (2).to_bytes(32, "big")
    assert pa.g1_to_bytes(pa.g1_from_bytes(g)) == g
    with pytest.raises(pa.PlkError):
        pa.g1_from_bytes(b"\x80" + g[1:])                             # compression flag on an uncompressed encoding


def test_witness_loaders(golden_dir):
    r = open(os.path.join(golden_dir, "circuit.r1cs.json"), "rb").read()
    w = [1, 35, 3, 9]
    wt = (b"wtns" + struct.pack("<II", 2, 2) + struct.pack("<IQ", 1, 40) + struct.pack("<I", 32) + po.BN254_PRIME_LE
          + struct.pack("<I", len(w)) + struct.pack("<IQ", 2, 32 * len(w)) + b"".join(v.to_bytes(32, "little") for v in w))
    pa.Circuit(r, True, wt, False)
    pa.Circuit(r, True, b'["1","35","3","9"]', True)
    with pytest.raises(pa.PlkError):
        pa.Circuit(r, True, wt[:-1], False)
    with pytest.raises(pa.PlkError):
        pa.Circuit(r, True, b'["1","35"]', True)        # shorter than nVars
    with pytest.raises(pa.PlkError):
        pa.Circuit(r, True, b"wtnx" + wt[4:], False)


def test_transpiler_matches_oracle_on_synthetic_shapes():
    """same gate counts per constraint as the oracle transpiler, incl. the unpinned shapes"""
    import json
    rng = po.Xoshiro256ss(3)
    cons = []
    nvars = 24
    for k in range(40):
        def lc(nterms, with_const):
            d = {}
            for _ in range(nterms):
                d[str(1 + rng.next() % (nvars - 1))] = str(rng.fr())
            if with_const:
                d["0"] = str(rng.fr())
            return d
        shape = k % 8
        a = lc([1, 1, 3, 6, 0, 1, 1, 9][shape], shape in (2, 6))
        b = lc([1, 1, 1, 2, 2, 0, 1, 5][shape], shape == 3)
        c = lc([1, 3, 2, 7, 3, 5, 0, 1][shape], shape in (1, 4))
        if shape == 6:
            b = {list(a.keys())[0]: "5"} if list(a.keys())[0] != "0" else b
        cons.append([a, b, c])
    obj = {"nPubInputs": 2, "nOutputs": 1, "nVars": nvars, "constraints": cons}
    c = pa.Circuit(json.dumps(obj).encode(), True)
    assert c.analyse() == po.analyse(po.load_r1cs_json(obj))


@pytest.mark.parametrize("perms,rp", [(2, 20), (1, 56)])
def test_poseidon_shaped_transpile_matches_oracle(perms, rp):
    """PARITY UNPINNED: circom-Poseidon-shaped constraints (S-box inputs that are linear combinations of up
    to 60 signals, constant x LC output constraints; tests/gen/poseidon_like.py) take the d / d_next chains and constant
    merges of the transpiler, which no reference fixture reaches.  Checked here: the product's transpiler (multi-threaded,
    flat storage) and the oracle's independent Python restatement agree on every per-constraint gate count, through the
    JSON loader (terms ordered by string key) and through the binary .r1cs loader (file order)."""
    import json
    from tests.gen import poseidon_like as pl
    ni, nv, cons, wit = pl.build(perms, 1000 + rp, rp=rp)
    assert max(len(lc) for c in cons for lc in c) >= (20 if rp == 20 else 55)
    js = pl.as_circom_json(ni, nv, cons)
    c = pa.Circuit(json.dumps(js).encode(), True, json.dumps([str(x) for x in wit]).encode(), True)
    assert c.analyse() == po.analyse(po.load_r1cs_json(js))
    raw = c.export("r1cs")
    c2 = pa.Circuit(raw, False)
    assert c2.analyse() == po.analyse(po.load_r1cs_bin(raw))


def test_key_file_codec_roundtrip(golden_dir, golden_crs):
    """Crs::read / Crs::write through the ABI: the committed 2^10 key parses (every point curve-checked),
    re-serialises to the same bytes, and its G2 section is the crs_42 constant."""
    import ctypes
    raw = open(os.path.join(golden_dir, "setup_2pow10.key"), "rb").read()
    L = pa.lib()
    n = ctypes.c_uint64(0)
    g2 = ctypes.create_string_buffer(256)
    assert L.plk_key_parse(raw, ctypes.c_uint64(len(raw)), None, ctypes.c_uint64(0), ctypes.byref(n), g2) == 0
    assert n.value == 1024
    pts = np.zeros((1024, 8), dtype=np.uint64)
    assert L.plk_key_parse(raw, ctypes.c_uint64(len(raw)), pts.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(1024), ctypes.byref(n), g2) == 0
    assert np.array_equal(pts, golden_crs.g1)
    c42 = ctypes.create_string_buffer(256)
    L.plk_crs42_g2_bytes(c42)
    assert c42.raw == g2.raw == raw[-256:]
    ln = ctypes.c_uint64(0)
    out = ctypes.create_string_buffer(len(raw))
    assert L.plk_key_serialize(pts.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(1024), g2, out, ctypes.c_uint64(len(raw)), ctypes.byref(ln)) == 0
    assert out.raw == raw
    bad = bytearray(raw)
    bad[8 + 64 * 5 + 40] ^= 1                              # y of point 5: no longer on the curve
    assert L.plk_key_parse(bytes(bad), ctypes.c_uint64(len(bad)), pts.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(1024), ctypes.byref(n), g2) == 6
    assert L.plk_key_parse(raw[:1000], ctypes.c_uint64(1000), None, ctypes.c_uint64(0), ctypes.byref(n), g2) == 6


def test_cli_analyse_and_flag_surface(golden_dir, tmp_path):
    cli = os.path.join(os.path.dirname(pa.lib_path()), "plonkit")
    assert os.path.exists(cli)
    out = tmp_path / "analyse.json"
    subprocess.check_call([cli, "analyse", "-c", os.path.join(golden_dir, "circuit.r1cs.json"), "-o", str(out)], stderr=subprocess.DEVNULL)
    assert out.read_text() == open(os.path.join(golden_dir, "analyse.json")).read()
    # unknown flags / missing required arguments are usage errors, like clap's exit code 2
    assert subprocess.call([cli, "setup", "-p", "10"], stderr=subprocess.DEVNULL) == 2
    assert subprocess.call([cli, "prove", "--bogus", "1"], stderr=subprocess.DEVNULL) == 2
    assert subprocess.call([cli, "frobnicate"], stderr=subprocess.DEVNULL) == 2
    # verify (pure CPU): the key option is -v / --verification_key as in the reference (src/bin/main.rs:130-134); --vk stays an alias
    vk, proof = os.path.join(golden_dir, "vk.bin"), os.path.join(golden_dir, "proof.bin")
    for flag in ("-v", "--verification_key", "--vk"):
        assert subprocess.call([cli, "verify", "-p", proof, flag, vk], stderr=subprocess.DEVNULL) == 0, flag
    assert subprocess.call([cli, "verify", "--proof", proof, "--verification_key", vk, "-t", "keccak"], stderr=subprocess.DEVNULL) == 0
    bad = tmp_path / "bad_proof.bin"
    raw = bytearray(open(proof, "rb").read()); raw[-70] ^= 1
    bad.write_bytes(bytes(raw))
    assert subprocess.call([cli, "verify", "-p", str(bad), "--verification_key", vk], stderr=subprocess.DEVNULL) != 0
    assert subprocess.call([cli, "export-verification-key", "--verification_key", "x"], stderr=subprocess.DEVNULL) == 2   # that command spells it --vk


def test_synthetic_circuit_generator_is_stable():
    """the bench / test circuits (xoshiro256** seeded "plonkit", alternating 1-gate and 2-gate constraints)
    are pinned by the hash of their exported .r1cs / .wtns bytes: speed-ups of the generator must not change them"""
    import hashlib
    import plonkit_amd as pa
    want = {10: ("c1191a9442e55454", "a850a41ba6eaf55c"), 1000: ("4a6613f2f4da1a8a", "268bb47514331dc6"),
            (1 << 16) - 2: ("3e937a93121ddf50", "8d3846310dbc619f")}
    for n, (h_r1cs, h_wtns) in want.items():
        c = pa.Circuit.synthetic(n)
        assert hashlib.sha1(c.export("r1cs")).hexdigest()[:16] == h_r1cs, n
        assert hashlib.sha1(c.export("wtns")).hexdigest()[:16] == h_wtns, n
        c.close()


def test_input_generator_has_the_oracles_random_numbers():
    """tests/gen/poseidon_like.py carries its own xoshiro256** (so that bench.py's GPU-timed by-domain leg imports nothing of oracle/): same stream and same
    modulus as the oracle's, i.e. the circuits it builds did not change"""
    from tests.gen import poseidon_like as pl
    assert pl.R_MOD == R_MOD
    a, b = pl.Xoshiro256ss(77), po.Xoshiro256ss(77)
    assert [a.next() for _ in range(50)] == [b.next() for _ in range(50)] and [a.fr() for _ in range(20)] == [b.fr() for _ in range(20)]


def test_domain_size_without_a_setup(golden_dir):
    """plk_circuit_domain_size (round 6: what `dump-lagrange` needs of prepare_setup_for_prover — src/bin/main.rs:360-381 reads setup.n): transpile only,
    no GPU; N = next power of two above (public inputs + gates), as the oracle's setup computes it; long linear combinations count their chained gates"""
    import plonkit_amd as pa
    c = pa.Circuit.from_files(os.path.join(golden_dir, "circuit.r1cs.json"), os.path.join(golden_dir, "witness.json"))
    assert c.domain_size() == 8                                   # the reference's `simple` circuit: vk.bin says n = 7
    c.close()
    for gates, want in ((5, 8), (6, 8), (7, 16), (4094, 4096), (4095, 8192), (40000, 1 << 16)):
        c = pa.Circuit.synthetic(gates)                           # `gates` gates + one public input
        assert c.domain_size() == want, gates
        c.close()
    c = pa.Circuit.synthetic_ex(4094, lc_terms=7)                 # the dense generator hits its gate target with chained linear combinations
    assert c.domain_size() == 4096
    c.close()


def test_header_is_plain_c_and_links(tmp_path):
    """include/plonkit_amd.h is the boundary a cgo / Rust-bindgen / C caller sees: it must compile as C99 (no C++ in it) and
    a C program linked against the shared library must resolve the symbols (the calls made here need no GPU)"""
    import shutil
    if shutil.which("gcc") is None:
        pytest.skip("gcc not on PATH")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "abi.c"
    src.write_text('#include "plonkit_amd.h"\n#include <stdio.h>\n#include <string.h>\n'
                   'int main(void) {\n'
                   '  plk_transcript t; plk_fr c; unsigned char h[32];\n'
                   '  plk_transcript_init(&t); plk_transcript_challenge(&t, &c);\n'
                   '  plk_keccak256((const unsigned char *)"", 0, h);\n'
                   '  plk_ctx *ctx = 0; int rc = plk_create(0, &ctx);            /* PLK_ERR_HIP without a GPU: no CPU fallback */\n'
                   '  printf("%s %02x%02x %d %d\\n", plk_version(), h[0], h[1], rc == PLK_OK || rc == PLK_ERR_HIP, (int)sizeof(plk_comm_id));\n'
                   '  if (ctx) plk_destroy(ctx);\n  return 0;\n}\n')
    exe = str(tmp_path / "abi")
    libdir = os.path.dirname(pa.lib_path())
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(root, "include"), str(src), "-o", exe,
                           "-L" + libdir, "-lplonkit_amd", "-Wl,-rpath," + libdir])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120).stdout.split()
    assert out[0] == "plonkit_amd" and out[-3] == "c5d2" and out[-2] == "1" and out[-1] == "128", out   # keccak256("") = c5d2..., ncclUniqueId is 128 bytes


def test_synthetic_ex_generator_invariants():
    """plk_circuit_synthetic_ex (bench / test input): exactly `target_gates` gates for every linear-combination width and
    size, the product's and the oracle's transpilers agree gate for gate, the generated witness satisfies every gate,
    witness_seed changes the witness and not the R1CS, lc_terms = 0 is plk_circuit_synthetic, bad arguments are refused"""
    import json
    for lc in (5, 6, 7, 8, 12, 33, 64):
        for tg in (4, 5, 9, 30, 126):
            c = pa.Circuit.synthetic_ex(tg, 1000 + lc, 7, lc)
            raw, wt = c.export("r1cs"), c.export("wtns")
            r1, w = po.load_r1cs_bin(raw), po.parse_wtns(wt)
            assert json.loads(c.analyse())["num_gates"] == tg and c.analyse() == po.analyse(r1), (lc, tg)
            T = po.transpile(r1, w)
            assert po.is_satisfied(r1, T, po.setup(r1, T)), (lc, tg)
            c2 = pa.Circuit.synthetic_ex(tg, 1000 + lc, 8, lc)
            assert c2.export("r1cs") == raw and c2.export("wtns") != wt
    a, b = pa.Circuit.synthetic(510), pa.Circuit.synthetic_ex(510)
    assert a.export("r1cs") == b.export("r1cs") and a.export("wtns") == b.export("wtns")
    for args in ((3, 1, 0, 0), (10, 1, 0, 4), (10, 1, 0, 65), (1 << 28, 1, 0, 0)):
        with pytest.raises(pa.PlkError) as e:
            pa.Circuit.synthetic_ex(*args)
        assert e.value.code == 1


def test_verify_strict_inputs_flag(golden_dir, golden_crs, monkeypatch):
    """a key with num_inputs = 0 (legal circom, src/reader.rs:197): plk_verify accepts it by default, refuses it under
    PLK_VERIFY_STRICT_INPUTS=1 / plk_verify_ex(PLK_VERIFY_STRICT_INPUTS) — the Solidity verifier's rule (contrib/template.sol:697).
    The golden one-input proof is untouched by the flag.  Key and proof of the zero-input circuit come from the oracle (CPU)."""
    import json
    import plonkit_amd as pa
    from oracle import plonk_oracle as po
    from oracle.oracle_lib import R_MOD
    u, v = 3, 5
    wit = [1, u, v, u * v % R_MOD]
    cons = [({"1": "1"}, {"2": "1"}, {"3": "1"})]
    for _ in range(4):
        wit.append(wit[-1] * v % R_MOD)
        cons.append(({str(len(wit) - 2): "1"}, {"2": "1"}, {str(len(wit) - 1): "1"}))
    js = {"n8": 32, "prime": str(R_MOD), "nVars": len(wit), "nOutputs": 0, "nPubInputs": 0, "nPrvInputs": 2,
          "nLabels": len(wit), "nConstraints": len(cons), "constraints": [list(c) for c in cons]}
    r1cs = po.load_r1cs_json(js)
    S = po.setup(r1cs)
    proof = po.write_proof(po.prove(r1cs, wit, golden_crs, S))
    vk = po.write_vk(po.make_verification_key(S, golden_crs))
    assert len(po.read_proof(proof).inputs) == 0
    monkeypatch.delenv("PLK_VERIFY_STRICT_INPUTS", raising=False)
    assert pa.verify(vk, proof) and pa.verify(vk, proof, strict_inputs=False)
    assert not pa.verify(vk, proof, strict_inputs=True)
    monkeypatch.setenv("PLK_VERIFY_STRICT_INPUTS", "1")
    assert not pa.verify(vk, proof)                                     # plk_verify reads the switch per call
    assert pa.verify(vk, proof, strict_inputs=False)                    # the explicit flag wins over the environment
    monkeypatch.setenv("PLK_VERIFY_STRICT_INPUTS", "0")
    assert pa.verify(vk, proof)
    gvk, gproof = open(golden_dir + "/vk.bin", "rb").read(), open(golden_dir + "/proof.bin", "rb").read()
    assert pa.verify(gvk, gproof, strict_inputs=True) and pa.verify(gvk, gproof, strict_inputs=False)
    import ctypes
    valid = ctypes.c_int32(0)
    rc = pa.lib().plk_verify_ex(gvk, ctypes.c_uint64(len(gvk)), gproof, ctypes.c_uint64(len(gproof)), ctypes.c_uint32(2), ctypes.byref(valid))
    assert rc == 1                                                      # PLK_ERR_ARG: unknown flag bit
