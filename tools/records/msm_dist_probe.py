"""MSM time at 2^20 for the scalar distributions of SURVEY.md §8(d): python tools/records/msm_dist_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.environ.get("PLK_AB_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))   # PLK_AB_ROOT=ab_old: tools/ab_build.sh
import numpy as np, torch
import plonkit_amd as pa
from oracle import oracle_lib as ol
ctx = pa.Context(0); dev = torch.device("cuda:0")
n = 1 << 20
ctx.srs_generate(n, 0, 42)
rng = np.random.default_rng(1)
R = ol.R_MOD
def mont(vals):        # python ints -> Montgomery limbs
    return ol.fr_vec(vals)
uni = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); uni[:, 3] &= np.uint64((1 << 60) - 1)
cases = {"uniform": uni}
cases["all ones"] = np.tile(mont([1]), (n, 1))
cases["all r-1"] = np.tile(mont([R - 1]), (n, 1))
small = mont(list(range(1 << 16)))
w = uni.copy(); m = rng.random(n); w[m < 0.5] = 0; idx = (m >= 0.5) & (m < 0.75); w[idx] = small[rng.integers(0, 1 << 16, size=int(idx.sum()))]
cases["witness-like (50% 0, 25% <2^16)"] = w
cases["booleans"] = np.where((rng.random(n) < 0.5)[:, None], np.tile(mont([1]), (n, 1)), 0).astype(np.uint64)
for name, a in cases.items():
    s = torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(dev)
    torch.cuda.synchronize()
    ctx.msm_dev(s, n); ctx.msm_dev(s, n)
    t0 = time.time(); reps = 5
    for _ in range(reps): ctx.msm_dev(s, n)
    print("%-36s %.3f ms" % (name, (time.time() - t0) / reps * 1e3), flush=True)
