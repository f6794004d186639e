"""commitment time against the number of terms (one GPU, key of exactly that many points, one commitment in flight and three):
the building blocks of DESIGN.md section 6's multi-GPU predictions.  python tools/msm_size_probe.py [log_n ...]"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.environ.get("PLK_AB_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import plonkit_amd as pa
from plonkit_amd.sharded import ShardedMsm
dev = torch.device("cuda:0")
for log_n in [int(a) for a in sys.argv[1:]] or [14, 16, 17, 18, 19, 20, 21, 22, 24]:
    n = 1 << log_n
    ctx = pa.Context(0)
    ctx.srs_generate(n, 0, 42)
    g = torch.Generator(device=dev); g.manual_seed(log_n)
    s = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device=dev, generator=g); s[:, 3] &= (1 << 60) - 1
    torch.cuda.synchronize()
    for _ in range(3): ctx.msm_dev(s, n)
    reps = 20 if log_n <= 21 else 5
    t0 = time.perf_counter()
    for _ in range(reps): ctx.msm_dev(s, n)
    one = (time.perf_counter() - t0) / reps * 1e3
    msm = ShardedMsm(ctx, None, dev)
    st = torch.cuda.Stream(device=dev)
    for _ in msm.commit_stream((s for _ in range(3)), n, stream=st): pass
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in msm.commit_stream((s for _ in range(reps)), n, stream=st): pass
    torch.cuda.synchronize()
    three = (time.perf_counter() - t0) / reps * 1e3
    print("2^%-2d terms: one at a time %8.3f ms   three in flight %8.3f ms per commitment" % (log_n, one, three), flush=True)
    ctx.close()
