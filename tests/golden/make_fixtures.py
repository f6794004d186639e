#!/usr/bin/env python3
"""Regenerates tests/golden/ from the reference checkout (run in the build container only;
/root/reference does not exist on the GPU box, tests never read it).

Every file is DATA the reference's own tests hold:
  setup_2pow10.key   <- keys/setup/setup_2^10.key           (src/tests.rs:10 MONOMIAL_KEY_FILE)
  circuit.r1cs.json  <- test/circuits/simple/circuit.r1cs.json  (src/tests.rs:6)
  witness.json       <- test/circuits/simple/witness.json   (src/tests.rs:7)
  vk.bin, proof.bin  <- test/circuits/simple/{vk,proof}.bin (src/tests.rs:8-9, golden outputs)
  r1cs_sample.bin    <- the hex test vector of src/r1cs_file.rs:164-214 decoded to bytes
  analyse.json       <- the expected string of src/tests.rs:14
"""
import os, re, shutil
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
for src, dst in [("keys/setup/setup_2^10.key", "setup_2pow10.key"),
                 ("test/circuits/simple/circuit.r1cs.json", "circuit.r1cs.json"),
                 ("test/circuits/simple/witness.json", "witness.json"),
                 ("test/circuits/simple/vk.bin", "vk.bin"),
                 ("test/circuits/simple/proof.bin", "proof.bin")]:
    shutil.copyfile(os.path.join(REF, src), os.path.join(HERE, dst))
rs = open(os.path.join(REF, "src/r1cs_file.rs")).read()
m = re.search(r'let data = hex!\(\s*"(.*?)"\s*\);', rs, re.S)
open(os.path.join(HERE, "r1cs_sample.bin"), "wb").write(bytes.fromhex(re.sub(r"\s+", "", m.group(1))))
ts = open(os.path.join(REF, "src/tests.rs")).read()
m = re.search(r'CIRCUIT_ANALYZE_RESULT: &\'static str = r#"(.*?)"#;', ts)
open(os.path.join(HERE, "analyse.json"), "w").write(m.group(1))
print("fixtures written to", HERE)
