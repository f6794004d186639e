"""window-width probe: PLK_MSM_C=<c> python tools/records/msm_c_probe.py <log_n...> — time + result fingerprint"""
import sys, time, hashlib
import os; sys.path.insert(0, os.path.abspath(os.environ.get("PLK_AB_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))   # PLK_AB_ROOT=ab_old: tools/ab_build.sh
import numpy as np, torch
import plonkit_amd as pa
ctx = pa.Context(0); dev = torch.device("cuda:0")
lns = [int(x) for x in sys.argv[1:]] or [20]
ctx.srs_generate(1 << max(lns), start=0, tau=42)
for ln in lns:
    m = 1 << ln
    rng = np.random.default_rng(3)
    a = rng.integers(0, 1 << 64, size=(m, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 60) - 1)    # uniform 252-bit values
    s = torch.from_numpy(a.view(np.int64)).to(dev)
    out = ctx.msm_dev(s, m); ctx.msm_dev(s, m)            # both MSM slots warm (scratch allocated)
    torch.cuda.synchronize()
    t0 = time.time(); reps = 5
    for _ in range(reps): ctx.msm_dev(s, m)
    dt = (time.time() - t0) / reps
    print(f"msm 2^{ln}: {dt*1e3:.3f} ms  {m/dt/1e6:.2f} Mscalar-mul/s  {hashlib.sha1(np.asarray(out).tobytes()).hexdigest()[:12]}", flush=True)
