"""back-to-back commitments: one at a time vs several in flight (ShardedMsm.commit_stream)"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.environ.get("PLK_AB_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # PLK_AB_ROOT=ab_old: tools/ab_build.sh
import torch
import plonkit_amd as pa
from plonkit_amd.sharded import ShardedMsm
ctx = pa.Context(0); dev = torch.device("cuda:0")
n = 1 << 20
ctx.srs_generate(n, 0, 42)
g = torch.Generator(device=dev); g.manual_seed(5)
s = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device=dev, generator=g); s[:, 3] &= (1 << 60) - 1
torch.cuda.synchronize()
st = torch.cuda.Stream(device=dev)
msm = ShardedMsm(ctx, None, dev)
for timing in (False, True):
    ctx.set_kernel_timing(timing)
    for _ in range(3): msm.commit(s, n, stream=st)
    K = 30
    t0 = time.perf_counter()
    kms = []
    for _ in range(K):
        msm.commit(s, n, stream=st)
        if timing: kms.append(ctx.msm_last_kernel_ms())
    t1 = time.perf_counter()
    for out in msm.commit_stream((s for _ in range(K)), n, stream=st):
        if timing: ctx.msm_last_kernel_ms()
    t2 = time.perf_counter()
    print("timing=%s  one at a time %.3f ms (accumulate kernel %.3f)   three in flight %.3f ms" % (timing, (t1 - t0) / K * 1e3, sum(kms) / len(kms) if kms else 0.0, (t2 - t1) / K * 1e3), flush=True)
