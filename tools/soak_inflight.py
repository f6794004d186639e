"""concurrency soak: K provers (one context + host thread each, shared setup, borrowed key) prove their own witness N times at the
2^log_n domain, pinned-subset and dense circuits; every proof must equal the one made alone — a race detector for the
several-proofs-in-flight path.  python tools/soak_inflight.py [log_n=16] [in_flight=3] [proofs_each=60]"""
import os, sys
sys.path.insert(0, os.path.abspath(os.environ.get("PLK_AB_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import plonkit_amd as pa
from plonkit_amd import prover_bench
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
each = int(sys.argv[3]) if len(sys.argv) > 3 else 60
ctx = pa.Context(0); ctx.srs_generate(1 << log_n, 0, 42)
for lc in (0, 9):
    r = prover_bench.throughput(ctx, log_n, in_flight=k, proofs_each=each, lc_terms=lc)     # raises on any differing proof
    print("2^%d lc_terms %d: %d provers x %d proofs, all byte-identical to the sequential ones; %.2f ms per proof (sequential %.2f)"
          % (log_n, lc, k, each, r["ms_per_proof"], r["sequential"]["ms_per_proof"]), flush=True)
