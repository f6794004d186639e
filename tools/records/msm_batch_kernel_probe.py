"""accumulate-kernel time per commitment for batches of 1..4 vectors of 2^log_n uniform scalars, one batch at a time
(HIP events around msm_accumulate): python tools/records/msm_batch_kernel_probe.py [log_n]"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.environ.get("PLK_AB_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))   # PLK_AB_ROOT=ab_old: tools/ab_build.sh
import numpy as np, torch
import plonkit_amd as pa
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = pa.Context(0); dev = torch.device("cuda:0")
n = 1 << log_n
ctx.srs_generate(n, 0, 42)
g = torch.Generator(device=dev); g.manual_seed(5)
vs = []
for m in range(4):
    t = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device=dev, generator=g); t[:, 3] &= (1 << 60) - 1
    vs.append(t)
zero = torch.zeros((n, 4), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
ctx.set_kernel_timing(True)
for name, batch in (("1", vs[:1]), ("2", vs[:2]), ("3", vs[:3]), ("4", vs[:4]), ("3 + one all-zero vector (the prover's wire batch)", vs[:3] + [zero])):
    for _ in range(3): ctx.msm_batch_dev(batch, n)
    ks, t0 = [], time.perf_counter()
    for _ in range(20):
        ctx.msm_batch_dev(batch, n); ks.append(ctx.msm_last_kernel_ms())
    dt = (time.perf_counter() - t0) / 20 * 1e3
    real = sum(1 for v in batch if v is not zero)
    print("batch %-50s accumulate %.3f ms = %.3f per non-empty commitment; whole batch %.3f ms = %.3f per commitment" % (name, np.mean(ks), np.mean(ks) / real, dt, dt / real), flush=True)
# the same batch of 2 with a pause before every call: does the accumulation slow down after the GPU has been (nearly) idle?
for pause_ms in (0.0, 0.5, 1.0, 2.0, 5.0):
    ks = []
    for _ in range(20):
        if pause_ms: time.sleep(pause_ms / 1e3)
        ctx.msm_batch_dev(vs[:2], n); ks.append(ctx.msm_last_kernel_ms())
    print("batch 2 after %.1f ms of idle GPU: accumulate %.3f ms (min %.3f max %.3f)" % (pause_ms, np.mean(ks), min(ks), max(ks)), flush=True)
# what keeps the clock up?  2 ms before every call filled with (a) nothing, (b) a memory-bound torch kernel, (c) a compute-bound one
big = torch.empty((1 << 28,), dtype=torch.float32, device=dev).normal_()
ma = torch.randn((4096, 4096), dtype=torch.float32, device=dev); mb = torch.randn((4096, 4096), dtype=torch.float32, device=dev)
small = torch.randn((64, 1024), dtype=torch.float32, device=dev)
torch.cuda.synchronize()
def fill(kind, ms):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        if kind == "mem": big.mul_(1.0000001)
        elif kind == "mm": torch.mm(ma, mb)
        elif kind == "tiny": small.mul_(1.0000001)          # a stream of tiny launches: the GPU is "active" but almost empty
        torch.cuda.synchronize()
for kind in ("sleep", "mem", "mm", "tiny"):
    ks = []
    for _ in range(20):
        if kind == "sleep": time.sleep(0.002)
        else: fill(kind, 2.0)
        ctx.msm_batch_dev(vs[:2], n); ks.append(ctx.msm_last_kernel_ms())
    print("batch 2 after 2 ms of %-5s: accumulate %.3f ms (min %.3f max %.3f)" % (kind, np.mean(ks), min(ks), max(ks)), flush=True)
