"""differential fuzz of setup + prove against the oracle: random synthetic circuits inside the pinned transpilation
subset, random sizes and seeds; proof and verification-key bytes must match and the host verifier must accept.
python tools/prove_fuzz.py [cases] [seed]"""
import json, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import plonkit_amd as pa
from oracle import oracle_lib as ol, plonk_oracle as po
from test_oracle_golden import _chain_circuit
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = pa.Context(0)
srs = ol.crs42(1 << 12)
ctx.srs_upload(srs)
crs = po.Crs(srs, pa.crs42_g2_bytes())
bad = 0
for c in range(cases):
    n_cons = rng.choice([1, 2, 3, 4, 5, 7, 8, 20, 63, 64, 100, 333, 1000, 2000])
    seed = rng.randrange(1 << 60)
    r1cs, wit = _chain_circuit(n_cons, seed)
    as_json = {"n8": 32, "prime": str(ol.R_MOD), "nVars": r1cs.num_variables, "nOutputs": 0, "nPubInputs": r1cs.num_inputs - 1,
               "nPrvInputs": r1cs.num_variables - r1cs.num_inputs, "nLabels": r1cs.num_variables, "nConstraints": len(r1cs.constraints),
               "constraints": [[{str(i): str(v) for i, v in lc} for lc in con] for con in r1cs.constraints]}
    circ = pa.Circuit(json.dumps(as_json).encode(), True, json.dumps([str(x) for x in wit]).encode(), True)
    setup = pa.SetupForProver(ctx, circ)
    r1cs_o = po.load_r1cs_json(as_json)                      # the JSON loader orders the terms of a combination by string key
    S = po.setup(r1cs_o)
    vk, proof = setup.verification_key_bytes(crs.g2_raw), setup.prove(circ)
    ok_vk = vk == po.write_vk(po.make_verification_key(S, crs))
    ok_pr = proof == po.write_proof(po.prove(r1cs_o, wit, crs, S))
    ok_vf = pa.verify(vk, proof)
    ok = ok_vk and ok_pr and ok_vf
    if not ok: print("   vk %s proof %s verify %s" % (ok_vk, ok_pr, ok_vf))
    bad += 0 if ok else 1
    print("case %3d constraints=%5d domain=%5d %s" % (c, n_cons, setup.domain_size, "ok" if ok else "MISMATCH"), flush=True)
    setup.close(); circ.close()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
