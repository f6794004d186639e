"""per-kernel picture of ONE scalar distribution (for `rocprofv3 --kernel-trace --stats`): python tools/msm_dist_kernels.py <uniform|r_minus_1|ones|witness> [reps]"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.environ.get("PLK_AB_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import plonkit_amd as pa
from oracle import oracle_lib as ol
kind = sys.argv[1] if len(sys.argv) > 1 else "uniform"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = pa.Context(0); dev = torch.device("cuda:0")
n = 1 << 20
ctx.srs_generate(n, 0, 42)
rng = np.random.default_rng(1)
uni = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); uni[:, 3] &= np.uint64((1 << 60) - 1)
if kind == "uniform": a = uni
elif kind == "r_minus_1": a = np.tile(ol.fr_vec([ol.R_MOD - 1]), (n, 1))
elif kind == "ones": a = np.tile(ol.fr_vec([1]), (n, 1))
else:
    small = ol.fr_vec(list(range(1 << 16)))
    a = uni.copy(); m = rng.random(n); a[m < 0.5] = 0; idx = (m >= 0.5) & (m < 0.75); a[idx] = small[rng.integers(0, 1 << 16, size=int(idx.sum()))]
s = torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(dev)
torch.cuda.synchronize()
ctx.msm_dev(s, n); ctx.msm_dev(s, n)
t0 = time.time()
for _ in range(reps): ctx.msm_dev(s, n)
print("%-12s %.3f ms per commitment" % (kind, (time.time() - t0) / reps * 1e3), flush=True)
