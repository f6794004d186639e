"""whole `plonkit prove` process at the 2^20 domain, N runs (prover_bench.cli_whole): python tools/cli_whole_ab.py [runs]"""
import json, os, sys
sys.path.insert(0, os.path.abspath(os.environ.get("PLK_AB_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from plonkit_amd import prover_bench
r = prover_bench.cli_whole(20, runs=int(sys.argv[1]) if len(sys.argv) > 1 else 7)
r.pop("what")
print(json.dumps(r))
