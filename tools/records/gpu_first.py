"""Quick timing probe of the first HIP path (NTT + MSM) — scratch tool, not the bench."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
import plonkit_amd as pa
from oracle import oracle_lib as ol

ctx = pa.Context(0)
dev = torch.device("cuda:0")
def rnd(n, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 60) - 1); return a
for log_n in (16, 20, 22, 24):
    n = 1 << log_n
    t = torch.from_numpy(rnd(n, 1).view(np.int64)).to(dev)
    st = torch.cuda.Stream(); torch.cuda.set_stream(st)
    for inv in (False, True):
        ctx.ntt_dev(t, log_n, inverse=inv, stream=st); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); reps = 10
        for _ in range(reps): ctx.ntt_dev(t, log_n, inverse=inv, stream=st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"ntt 2^{log_n} inv={inv}: {ms:.3f} ms  algo {64*n/ms/1e6:.1f} GB/s", flush=True)
n = 1 << 20
t0 = time.time(); srs = ol.crs42(n); print("crs42 gen", time.time() - t0, flush=True)
ctx.srs_upload(srs)
for ln in (12, 14, 16, 17, 18, 20):
    m = 1 << ln
    s = torch.from_numpy(rnd(m, 3).view(np.int64)).to(dev)
    st = torch.cuda.current_stream()
    ctx.msm_dev(s, m, stream=st)
    t0 = time.time(); reps = 5
    for _ in range(reps): ctx.msm_dev(s, m, stream=st)
    dt = (time.time() - t0) / reps
    print(f"msm 2^{ln}: {dt*1e3:.3f} ms  {m/dt/1e6:.2f} Mscalar-mul/s", flush=True)
