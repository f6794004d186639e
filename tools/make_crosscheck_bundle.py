#!/usr/bin/env python3
"""Cross-check bundle for whoever has a real `plonkit` (fluidex/plonkit, Rust) at hand — PARITY UNPINNED material.

This image has no Rust toolchain, so the transpilation of long linear combinations (bellman's IntoMultipleGates adaptor,
/root/reference/src/transpile.rs:127-139) has only ever been compared between this package's C++ transpiler and its own
Python oracle: two restatements of one recollection.  This tool writes, for seven circuits, exactly the files the
reference's test flow produces (/root/reference/test/test_poseidon_plonk.sh:47-80):

    <case>/circuit.r1cs  witness.wtns  setup.key  vk.bin  proof.bin  proof.json  public.json  analyse.json

all made by THIS package's `plonkit` binary on an MI355X, plus MANIFEST.json (sha256 of every file) and compare.sh,
which re-runs the same commands with the reference binary ($PLONKIT_REF_BIN) and `cmp`s the outputs.  One run of
compare.sh pins or refutes the transpiler, the setup polynomials, the prover and the serialisers for these shapes:

    simple         the reference's own golden circuit (pinned already: tests/golden/)
    poseidon_12    circom-Poseidon-shaped hash chains (tests/gen/poseidon_like.py): S-box inputs that are linear
    poseidon_14      combinations of up to 24 / 60 signals, constant x LC outputs — domains 2^12, 2^14, 2^16
    poseidon_16
    long_lc        one 11-term LC x signal = 2-term LC, one LC x LC with constants (d / d_next chains, merges)
    zero_inputs    a circuit without public signals: vk.num_inputs = 0 — does the reference's verifier accept it? (this package: yes by default,
                   no under PLK_VERIFY_STRICT_INPUTS=1, the Solidity template's rule)
    dense_14       the bench's DENSE synthetic circuit at the 2^14 domain (plk_circuit_synthetic_ex, lc_terms = 7: every constraint
                   a 7-term linear combination folded through the d column) — the shape `prove.dense` of bench.py times

usage (on a GPU box):  python tools/make_crosscheck_bundle.py <outdir>
Every artefact here is labelled UNPINNED until such a run has happened."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import plonkit_amd as pa                                             # noqa: E402
from oracle.oracle_lib import R_MOD                                   # noqa: E402  (test infrastructure: circuit generators only)
from oracle import plonk_oracle as po                                 # noqa: E402
from tests.gen import poseidon_like as pl                             # noqa: E402

CLI = os.path.join(os.path.dirname(pa.lib_path()), "plonkit")
GOLD = os.path.join(ROOT, "tests", "golden")


def long_lc_case():
    rng = po.Xoshiro256ss(7)
    wit = [1, 0] + [rng.fr() for _ in range(12)]
    lc = [(i, rng.fr()) for i in range(2, 12)]
    s = sum(c * wit[i] for i, c in lc) % R_MOD
    wit.append(s * wit[13] % R_MOD)
    wit[1] = wit[-1]
    cons = [(lc + [(0, 5)], [(13, 1)], [(14, 1), (13, 5)]), ([(1, 1)], [(0, 1)], [(14, 1)]),
            ([(3, 1), (0, R_MOD - 1)], [(3, 1), (0, 2)], [(15, 1)])]
    wit.append((wit[3] - 1) * (wit[3] + 2) % R_MOD)
    js = {"n8": 32, "prime": str(R_MOD), "nVars": len(wit), "nOutputs": 0, "nPubInputs": 1, "nPrvInputs": len(wit) - 2,
          "nLabels": len(wit), "nConstraints": len(cons),
          "constraints": [[{str(i): str(c) for i, c in lcx} for lcx in con] for con in cons]}
    return js, wit


def zero_inputs_case():
    """no public signal at all (nPubInputs = 0, legal circom: src/reader.rs:197): vk.num_inputs = 0 and a proof with an empty input
    list.  This package's verifier accepts it (PLK_VERIFY_STRICT_INPUTS=1 refuses, like contrib/template.sol:697); whether the
    reference's `plonkit verify` does is UNPINNED — compare.sh's last line for this case answers it."""
    u, v = 3, 5
    wit = [1, u, v, u * v % R_MOD]
    cons = [({"1": "1"}, {"2": "1"}, {"3": "1"})]
    for _ in range(4):
        wit.append(wit[-1] * v % R_MOD)
        cons.append(({str(len(wit) - 2): "1"}, {"2": "1"}, {str(len(wit) - 1): "1"}))
    js = {"n8": 32, "prime": str(R_MOD), "nVars": len(wit), "nOutputs": 0, "nPubInputs": 0, "nPrvInputs": 2,
          "nLabels": len(wit), "nConstraints": len(cons), "constraints": [list(c) for c in cons]}
    return js, wit


def cases():
    yield "simple", json.load(open(os.path.join(GOLD, "circuit.r1cs.json"))), [int(x) for x in json.load(open(os.path.join(GOLD, "witness.json")))], 10
    for perms, rp, log_n in ((7, 20, 12), (6, 56, 14), (120, 20, 16)):
        ni, nv, cons, wit = pl.build(perms, 77 + perms, rp=rp)
        yield "poseidon_%d" % log_n, pl.as_circom_json(ni, nv, cons), wit, log_n
    js, wit = long_lc_case()
    yield "long_lc", js, wit, 10
    js, wit = zero_inputs_case()
    yield "zero_inputs", js, wit, 10


def run(*args):
    subprocess.run([CLI] + list(args), check=True, stderr=subprocess.DEVNULL)


def circuits():
    for name, js, wit, log_n in cases():
        yield name, pa.Circuit(json.dumps(js).encode(), True, json.dumps([str(x) for x in wit]).encode(), True), log_n
    yield "dense_14", pa.Circuit.synthetic_ex((1 << 14) - 2, lc_terms=7), 14


def main(out):
    manifest = {}
    for name, circ, log_n in circuits():
        d = os.path.join(out, name)
        os.makedirs(d, exist_ok=True)
        f = lambda x: os.path.join(d, x)                               # noqa: E731
        open(f("circuit.r1cs"), "wb").write(circ.export("r1cs"))       # circom's binary formats (src/r1cs_file.rs, src/reader.rs)
        open(f("witness.wtns"), "wb").write(circ.export("wtns"))
        circ.close()
        run("setup", "-p", str(log_n), "-m", f("setup.key"), "--overwrite")
        run("analyse", "-c", f("circuit.r1cs"), "-o", f("analyse.json"))
        run("export-verification-key", "-m", f("setup.key"), "-c", f("circuit.r1cs"), "-v", f("vk.bin"), "--overwrite")
        run("prove", "-m", f("setup.key"), "-c", f("circuit.r1cs"), "-w", f("witness.wtns"), "-p", f("proof.bin"),
            "-j", f("proof.json"), "-i", f("public.json"), "--overwrite")
        run("verify", "-p", f("proof.bin"), "-v", f("vk.bin"))
        manifest[name] = {x: hashlib.sha256(open(f(x), "rb").read()).hexdigest() for x in sorted(os.listdir(d))}
        manifest[name]["domain_log2"] = log_n
        print(name, "ok: proof", manifest[name]["proof.bin"][:16], "vk", manifest[name]["vk.bin"][:16], flush=True)
    json.dump(manifest, open(os.path.join(out, "MANIFEST.json"), "w"), indent=1, sort_keys=True)
    open(os.path.join(out, "compare.sh"), "w").write(COMPARE)
    os.chmod(os.path.join(out, "compare.sh"), 0o755)
    print("bundle written to", out)


COMPARE = r"""#!/bin/bash
# PLONKIT_REF_BIN=/path/to/the/reference/plonkit ./compare.sh     (run inside the bundle directory)
# For every case: the reference regenerates key, vk, proof and analysis from the SAME circuit.r1cs / witness.wtns and each
# file is compared byte for byte with the one this package made; then the reference verifies this package's proof.
# setup.key: both sides use the insecure tau = 42 generator (src/plonk.rs:30-48), so the keys must be identical too.
# proof.json / public.json formats are unpinned (bellman_vk_codegen is not in the reference tree): reported, not fatal.
set -u
R=${PLONKIT_REF_BIN:?set PLONKIT_REF_BIN}
bad=0
for d in */; do
  d=${d%/}; [ -f "$d/circuit.r1cs" ] || continue
  p=$(python3 -c "import json; print(json.load(open('MANIFEST.json'))['$d']['domain_log2'])")
  t=$(mktemp -d)
  "$R" setup -p "$p" -m "$t/setup.key" --overwrite >/dev/null 2>&1
  "$R" analyse -c "$d/circuit.r1cs" -o "$t/analyse.json" >/dev/null 2>&1
  "$R" export-verification-key -m "$t/setup.key" -c "$d/circuit.r1cs" -v "$t/vk.bin" --overwrite >/dev/null 2>&1
  "$R" prove -m "$t/setup.key" -c "$d/circuit.r1cs" -w "$d/witness.wtns" -p "$t/proof.bin" -j "$t/proof.json" -i "$t/public.json" --overwrite >/dev/null 2>&1
  for f in setup.key analyse.json vk.bin proof.bin; do
    if cmp -s "$d/$f" "$t/$f"; then echo "$d/$f identical"; else echo "$d/$f DIFFERS"; bad=1; fi
  done
  for f in proof.json public.json; do cmp -s "$d/$f" "$t/$f" && echo "$d/$f identical" || echo "$d/$f differs (format unpinned)"; done
  "$R" verify -p "$d/proof.bin" -v "$d/vk.bin" >/dev/null 2>&1 && echo "$d: reference verifies this package's proof" || { echo "$d: reference REJECTS this package's proof"; bad=1; }
  rm -rf "$t"
done
exit $bad
"""

if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "crosscheck_bundle")
