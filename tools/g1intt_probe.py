import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import plonkit_amd as pa
from oracle import oracle_lib as ol
from oracle.oracle_lib import R_MOD
ctx = pa.Context(0)
for log_n in [int(x) for x in sys.argv[1:]] or [12, 16]:
    n = 1 << log_n
    ctx.srs_generate(n, 0, 42)
    out = torch.empty((n, 8), dtype=torch.int64, device="cuda:0")
    ctx.g1_intt_srs_dev(log_n, out); ctx.synchronize()
    t0 = time.time(); ctx.g1_intt_srs_dev(log_n, out); ctx.synchronize(); dt = time.time() - t0
    res = out.cpu().numpy().view(np.uint64)
    w = ol.omega(log_n); zh = (pow(42, n, R_MOD) - 1) % R_MOD; G = ol.g1_generator(); ok = True
    for i in (0, 1, n // 3, n - 1):
        wi = pow(w, i, R_MOD); li = wi * zh % R_MOD * pow(n * (42 - wi) % R_MOD, -1, R_MOD) % R_MOD
        ok &= bool(np.array_equal(res[i], ol.g1_mul(G, li)))
    print(f"g1_intt 2^{log_n}: {dt*1e3:.1f} ms  {128*n/dt/1e9:.3f} GB/s algorithmic  ok={ok}", flush=True)
