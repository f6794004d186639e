"""crs_42 generation time (plk_srs_generate) at a few sizes, checked against the oracle at 2^12 and by spot indices above: python tools/srs_generate_probe.py [log_n ...]
(PLK_SRS_DIRECT=1: the one-kernel path of rounds 1-5)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import plonkit_amd as pa
from oracle import oracle_lib as ol
from oracle.oracle_lib import R_MOD
ctx = pa.Context(0)
ctx.srs_generate(1 << 12, 0, 42)
assert np.array_equal(ctx.srs_download(0, 1 << 12), ol.crs42(1 << 12)), "crs_42 mismatch vs oracle at 2^12"
for log_n in [int(a) for a in sys.argv[1:]] or [16, 20, 22]:
    n = 1 << log_n
    ctx.srs_generate(n, 0, 42)
    t0 = time.perf_counter(); ctx.srs_generate(n, 0, 42); dt = time.perf_counter() - t0
    G = ol.g1_generator(); ok = True
    for i in [0, 1, 7, 8, 31, 32, 33, n // 2 + 5, n - 9, n - 1]:
        ok &= bool(np.array_equal(ctx.srs_download(i, 1)[0], ol.g1_mul(G, pow(42, i, R_MOD))))
    print("crs_42 2^%d: %.2f ms  ok=%s" % (log_n, dt * 1e3, ok), flush=True)
ctx.srs_generate(1000, 5, 42)                                   # ragged length, non-zero start
ok = all(np.array_equal(ctx.srs_download(i, 1)[0], ol.g1_mul(ol.g1_generator(), pow(42, 5 + i, R_MOD))) for i in (0, 1, 31, 32, 999))
print("crs_42 1000 points from power 5: ok=%s" % ok)
