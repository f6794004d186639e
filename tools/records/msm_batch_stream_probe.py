"""commitment stream with B vectors per pass of the kernels, two passes in flight: python tools/records/msm_batch_stream_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.environ.get("PLK_AB_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))   # PLK_AB_ROOT=ab_old: tools/ab_build.sh
import numpy as np, torch
import plonkit_amd as pa
from plonkit_amd.sharded import ShardedMsm
ctx = pa.Context(0); dev = torch.device("cuda:0")
n = 1 << 20
ctx.srs_generate(n, 0, 42)
g = torch.Generator(device=dev); g.manual_seed(5)
vecs = []
for k in range(8):
    s = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device=dev, generator=g); s[:, 3] &= (1 << 60) - 1
    vecs.append(s)
torch.cuda.synchronize()
st = torch.cuda.Stream(device=dev)
msm = ShardedMsm(ctx, None, dev)
single = [np.asarray(ctx.msm_dev(v, n)) for v in vecs]
for B in (1, 2, 4, 8):
    K = 48 // B
    def gen(k): return (vecs[:B] for _ in range(k))
    for out in msm.commit_batches(gen(2), n, stream=st): pass
    assert all(np.array_equal(out[i], single[i]) for i in range(B))
    t0 = time.perf_counter()
    for out in msm.commit_batches(gen(K), n, stream=st): pass
    dt = time.perf_counter() - t0
    print("batch %d: %.3f ms per commitment  (%.0f M scalar-mul/s)" % (B, dt / (K * B) * 1e3, K * B * n / dt / 1e6), flush=True)
