import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import oracle_lib as ol
n = 1 << 18
bases = ol.crs42(n)
rng = np.random.default_rng(1)
s = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); s[:, 3] &= np.uint64((1 << 60) - 1)
print("cpus", os.cpu_count())
for t in (8, 16, 32, 64, 128, 256):
    ol.msm(bases[:1024], s[:1024], threads=t)
    t0 = time.time(); ol.msm(bases, s, threads=t); dt = time.time() - t0
    print("threads %3d: %.3f s  %.3f Mscalar-mul/s" % (t, dt, n / dt / 1e6), flush=True)
