"""the PCIe-inclusive commitment: plk_msm_g1 on HOST scalars (pageable numpy / page-locked torch memory) against the device-pointer entry,
2^20 terms — the figure DESIGN.md §5 quotes beside `value` (which is measured with the scalars resident in HBM).
usage (GPU box): python tools/msm_host_ptr_probe.py [log_n=20] [reps=20]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import plonkit_amd as pa
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = 1 << log_n
ctx = pa.Context(0)
ctx.srs_generate(n, 0, 42)
rng = np.random.default_rng(7)
host = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); host[:, 3] >>= 1        # top limb < 2^61 < the modulus: valid residues
dev = torch.from_numpy(host.view(np.int64)).to("cuda:0")
pinned = torch.from_numpy(host.view(np.int64)).pin_memory()
ref = ctx.msm(host)
def timed(fn):
    fn(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); out = fn(); ts.append(time.perf_counter() - t0)
    ts.sort(); return ts[len(ts) // 2] * 1e3, out
t_dev, o1 = timed(lambda: ctx.msm_dev(dev, n))
t_host, o2 = timed(lambda: ctx.msm(host))
t_pin, o3 = timed(lambda: ctx.msm(pinned.numpy().view(np.uint64)))
ok = all(np.array_equal(ref, o) for o in (o1, o2, o3))
print("2^%d terms, one commitment at a time (median of %d): device pointer %.3f ms (%.0f M/s) | pageable host pointer %.3f ms (%.0f M/s) | "
      "page-locked host pointer %.3f ms (%.0f M/s) | same point: %s" % (log_n, reps, t_dev, n / t_dev / 1e3, t_host, n / t_host / 1e3, t_pin, n / t_pin / 1e3, ok))
