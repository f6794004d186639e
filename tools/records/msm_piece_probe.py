"""one long commitment as pieces over successive SRS ranges, up to three pieces in flight: python tools/records/msm_piece_probe.py [log_n] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.environ.get("PLK_AB_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))   # PLK_AB_ROOT=ab_old: tools/ab_build.sh
import numpy as np
import torch
import plonkit_amd as pa

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = 1 << log_n
ctx = pa.Context(0); dev = torch.device("cuda:0")
ctx.srs_generate(n, 0, 42)
g = torch.Generator(device=dev); g.manual_seed(7)
s = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device=dev, generator=g)
s[:, 3] &= (1 << 60) - 1
torch.cuda.synchronize()


def whole():
    return np.asarray(ctx.msm_dev(s, n))


def pieces(log_p, depth=3):
    p = 1 << log_p
    parts, inflight, off = [], 0, 0
    while off < n or inflight:
        while off < n and inflight < depth:
            ctx.msm_enqueue_dev(s[off:off + p], p, base_offset=off)
            off += p; inflight += 1
        parts.append(ctx.msm_finish()); inflight -= 1
    return np.asarray(pa.g1_sum_jacobian(np.stack(parts)))


def timed(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


t, ref = timed(whole)
print("2^%d in one pass: %.2f ms" % (log_n, t), flush=True)
for log_p in range(min(log_n - 1, 23), 19, -1):
    for depth in (3, 1):
        t, out = timed(lambda: pieces(log_p, depth))
        print("pieces of 2^%d, %d in flight: %.2f ms  same point: %s" % (log_p, depth, t, bool(np.array_equal(out, ref))), flush=True)
