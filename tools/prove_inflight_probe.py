"""prove throughput with 1 / 2 / 3 ... proofs in flight on one GPU (plonkit_amd.prover_bench.throughput):
python tools/prove_inflight_probe.py [log_n=20] [in_flight list, e.g. 1,2,3] [proofs_each=10] [lc_terms=0]
GPU_MAX_HW_QUEUES in the environment is honoured (default 8, the library's own setting)."""
import json, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.abspath(os.environ.get("PLK_AB_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import plonkit_amd as pa
from plonkit_amd import prover_bench
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
flights = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,3").split(",")]
each = int(sys.argv[3]) if len(sys.argv) > 3 else 10
lc = int(sys.argv[4]) if len(sys.argv) > 4 else 0
ctx = pa.Context(0)
ctx.srs_generate(1 << log_n, 0, 42)
print("GPU_MAX_HW_QUEUES=%s  domain 2^%d  lc_terms %d" % (os.environ.get("GPU_MAX_HW_QUEUES"), log_n, lc), flush=True)
for k in flights:
    r = prover_bench.throughput(ctx, log_n, in_flight=k, proofs_each=each, lc_terms=lc)
    r.pop("what")
    print(json.dumps(r), flush=True)
