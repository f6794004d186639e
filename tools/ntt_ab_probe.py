"""Timings of the transforms a proof uses: python tools/ntt_ab_probe.py [log_n ...]  (forward, inverse, coset inverse at 4n, LDE x4)"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.environ.get("PLK_AB_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # PLK_AB_ROOT=ab_old: tools/ab_build.sh
import torch
import plonkit_amd as pa
ctx = pa.Context(0); dev = torch.device("cuda:0")


def timed(fn, reps):
    fn(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for log_n in [int(a) for a in sys.argv[1:]] or [20, 22, 24]:
    n = 1 << log_n
    reps = 20 if log_n <= 22 else 6
    x = torch.randint(0, 1 << 60, (n, 4), dtype=torch.int64, device=dev)
    row = {"fwd": timed(lambda: ctx.ntt_dev(x.data_ptr(), log_n), reps),
           "inv": timed(lambda: ctx.ntt_dev(x.data_ptr(), log_n, inverse=True), reps)}
    if log_n <= 24:
        y = torch.empty((4 * n, 4), dtype=torch.int64, device=dev)
        row["lde4"] = timed(lambda: ctx.lde4_dev(x.data_ptr(), log_n, y.data_ptr()), reps)
        del y
    print("2^%d: " % log_n + "  ".join("%s %.4f ms" % kv for kv in row.items()), flush=True)
    del x
