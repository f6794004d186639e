"""batched commitments vs single ones at 2^log_n: python tools/records/msm_batch_probe.py <log_n> <batch>"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.environ.get("PLK_AB_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))   # PLK_AB_ROOT=ab_old: tools/ab_build.sh
import numpy as np, torch
import plonkit_amd as pa
log_n = int(sys.argv[1]); batch = int(sys.argv[2])
ctx = pa.Context(0); dev = torch.device("cuda:0")
n = 1 << log_n
ctx.srs_generate(n, 0, 42)
g = torch.Generator(device=dev); g.manual_seed(5)
vs = []
for m in range(batch):
    t = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device=dev, generator=g); t[:, 3] &= (1 << 60) - 1
    vs.append(t)
torch.cuda.synchronize()
single = [np.asarray(ctx.msm_dev(v, n)) for v in vs]
print("singles ok", flush=True)
t0 = time.time(); out = ctx.msm_batch_dev(vs, n); dt = time.time() - t0
print("batch ok %.2f ms" % (dt * 1e3), all(np.array_equal(np.asarray(out[m]), single[m]) for m in range(batch)), flush=True)
