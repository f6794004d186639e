"""NTT timing / profiling probe: python tools/ntt_probe.py <log_n> [reps]"""
import sys, time
import os; sys.path.insert(0, os.path.abspath(os.environ.get("PLK_AB_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # PLK_AB_ROOT=ab_old: tools/ab_build.sh
import torch
import plonkit_amd as pa
ctx = pa.Context(0); dev = torch.device("cuda:0")
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n = 1 << log_n
x = torch.randint(0, 1 << 60, (n, 4), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
ctx.ntt_dev(x.data_ptr(), log_n); ctx.synchronize()
t0 = time.time()
for _ in range(reps): ctx.ntt_dev(x.data_ptr(), log_n)
ctx.synchronize()
print("ntt 2^%d: %.4f ms" % (log_n, (time.time() - t0) / reps * 1e3))
