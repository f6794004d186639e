"""determinism soak: the same 2^log_n proof N times (byte-identical every time) — a cheap race detector for the prover"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.environ.get("PLK_AB_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # PLK_AB_ROOT=ab_old: tools/ab_build.sh
import plonkit_amd as pa
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 18
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
ctx = pa.Context(0); ctx.srs_generate(1 << log_n, 0, 42)
circ = pa.Circuit.synthetic((1 << log_n) - 2)
setup = pa.SetupForProver(ctx, circ)
first = setup.prove(circ)
bad = sum(1 for _ in range(reps) if setup.prove(circ) != first)
ok = pa.verify(setup.verification_key_bytes(pa.crs42_g2_bytes()), first)
print("2^%d: %d proofs, %d differ from the first, verifier accepts: %s" % (log_n, reps, bad, ok))
sys.exit(1 if bad or not ok else 0)
