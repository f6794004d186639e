"""prove with and without a resident Lagrange-form key (`prove -l`, src/plonk.rs:138-146) at the 2^log_n domain: python tools/records/prove_lagrange_probe.py [log_n] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.environ.get("PLK_AB_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))   # PLK_AB_ROOT=ab_old: tools/ab_build.sh
import torch
import plonkit_amd as pa
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
n = 1 << log_n
ctx = pa.Context(0); ctx.srs_generate(n, 0, 42)
circ = pa.Circuit.synthetic(n - 2)
setup = pa.SetupForProver(ctx, circ)
def run(tag):
    first = setup.prove(circ); setup.prove(circ)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); p = setup.prove(circ); ts.append((time.perf_counter() - t0) * 1e3)
        assert p == first
    ts.sort()
    print("%-28s median %.2f ms  min %.2f   rounds %s" % (tag, ts[len(ts) // 2], ts[0], {k: round(v, 2) for k, v in setup.timings_ms().items()}), flush=True)
    return first
a = run("monomial key only")
lag = torch.zeros((n, 8), dtype=torch.int64, device="cuda:0")
t0 = time.perf_counter(); ctx.g1_intt_srs_dev(log_n, lag.data_ptr()); ctx.synchronize()
print("dump-lagrange (G1 iNTT of the key): %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
ctx.srs_lagrange_set_dev(lag.data_ptr(), n)
b = run("with the Lagrange-form key")
print("same proof bytes:", a == b)
