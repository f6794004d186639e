"""one setup + N proves at the 2^log_n domain (for rocprofv3 --kernel-trace --stats): python tools/prove_probe.py [log_n] [proves] [lc_terms]
(lc_terms > 0: the dense synthetic circuit — long linear combinations folded through the d column, 11 of 11 commitments non-trivial)"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.environ.get("PLK_AB_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # PLK_AB_ROOT=ab_old: tools/ab_build.sh
import plonkit_amd as pa
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ctx = pa.Context(0)
ctx.srs_generate(1 << log_n, 0, 42)
lc_terms = int(sys.argv[3]) if len(sys.argv) > 3 else 0
circ = pa.Circuit.synthetic_ex((1 << log_n) - 2, lc_terms=lc_terms) if lc_terms else pa.Circuit.synthetic((1 << log_n) - 2)
setup = pa.SetupForProver(ctx, circ)
setup.prove(circ)
tot, rounds = [], {}
for _ in range(reps):
    t0 = time.perf_counter(); setup.prove(circ); dt = time.perf_counter() - t0
    tm = setup.timings_ms()
    tot.append(dt * 1e3)
    for k, v in tm.items(): rounds.setdefault(k, []).append(v)
    if reps <= 8: print("prove 2^%d: %.2f ms  %s" % (log_n, dt * 1e3, {k: round(v, 2) for k, v in tm.items()}), flush=True)
tot.sort()
print("prove 2^%d over %d: median %.2f ms  min %.2f  mean %.2f   rounds (mean) %s" % (log_n, reps, tot[len(tot) // 2], tot[0], sum(tot) / len(tot),
      {k: round(sum(v) / len(v), 2) for k, v in rounds.items()}), flush=True)
if os.environ.get("PROBE_VERIFY"):
    vk = setup.verification_key_bytes(pa.crs42_g2_bytes())
    t0 = time.perf_counter(); ok = pa.verify(vk, setup.prove(circ)); dt = time.perf_counter() - t0
    print("host verifier (real pairing) accepts the 2^%d proof: %s  (prove + verify %.1f ms)" % (log_n, ok, dt * 1e3), flush=True)
