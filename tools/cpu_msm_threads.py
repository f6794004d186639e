"""thread scaling of the CPU baseline (oracle dense_multiexp restatement) in both work splits:
python tools/cpu_msm_threads.py [log_n=20]   — "chunks" = bellman 0.3.2's split, "windows" = one task per (window, chunk)"""
import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import oracle_lib as ol
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log_n
bases = ol.crs42(n, threads=os.cpu_count())
rng = np.random.default_rng(1)
s = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); s[:, 3] &= np.uint64((1 << 60) - 1)
print("cpus", os.cpu_count(), " terms 2^%d" % log_n)
ref = None
for split in ("windows", "chunks"):
    for t in (8, 16, 32, 64, 128, 256):
        if t > (os.cpu_count() or 1):
            continue
        ol.msm(bases[:1024], s[:1024], threads=t, split=split)
        t0 = time.time(); r = ol.msm(bases, s, threads=t, split=split); dt = time.time() - t0
        assert ref is None or np.array_equal(r, ref)
        ref = r
        print("%-8s threads %3d: %.3f s  %.3f Mscalar-mul/s" % (split, t, dt, n / dt / 1e6), flush=True)
