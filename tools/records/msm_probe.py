import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import plonkit_amd as pa
from oracle import oracle_lib as ol
ctx = pa.Context(0); dev = torch.device("cuda:0")
def rnd(n, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 60) - 1); return a
lns = [int(x) for x in sys.argv[1:]] or [12, 17, 20]
srs = ol.crs42(1 << max(lns)); ctx.srs_upload(srs)
for ln in lns:
    m = 1 << ln
    s = torch.from_numpy(rnd(m, 3).view(np.int64)).to(dev)
    ctx.msm_dev(s, m)
    t0 = time.time(); reps = 3
    for _ in range(reps): ctx.msm_dev(s, m)
    dt = (time.time() - t0) / reps
    print(f"msm 2^{ln}: {dt*1e3:.3f} ms  {m/dt/1e6:.2f} Mscalar-mul/s", flush=True)
