import os, sys, time
sys.path.insert(0, "/root/repo")
import plonkit_amd as pa
log_n = 20
ctx = pa.Context(0)
ctx.srs_generate(1 << log_n, 0, 42)
ctx.srs_precompute() if hasattr(ctx, "srs_precompute") else None
circ = pa.Circuit.synthetic((1 << log_n) - 2)
setup = pa.SetupForProver(ctx, circ)
ctx.synchronize()
print("MARK cold prove begins", flush=True)
t0 = time.perf_counter(); setup.prove(circ); t1 = time.perf_counter()
print("cold prove %.1f ms %s" % ((t1 - t0) * 1e3, {k: round(v, 2) for k, v in setup.timings_ms().items()}), flush=True)
t0 = time.perf_counter(); setup.prove(circ); t1 = time.perf_counter()
print("second prove %.1f ms" % ((t1 - t0) * 1e3), flush=True)
