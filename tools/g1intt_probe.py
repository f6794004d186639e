import sys, time
import os; sys.path.insert(0, os.path.abspath(os.environ.get("PLK_AB_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # PLK_AB_ROOT=ab_old: tools/ab_build.sh
sys.path.append(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # (the oracle always comes from the working tree)
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
    import random
    rnd = random.Random(log_n)
    for i in [0, 1, n - 1] + [rnd.randrange(n) for _ in range(61)]:          # 64 indices
        wi = pow(w, i, R_MOD); li = wi * zh % R_MOD * pow(n * (42 - wi) % R_MOD, -1, R_MOD) % R_MOD
        ok &= bool(np.array_equal(res[i], ol.g1_mul(G, li)))
    # sum_i L_i(tau) = 1: the points must add up to G (the host sum takes Jacobian input: affine with Z = 1)
    one = np.array([0xd35d438dc58f0d9d, 0x0a78eb28f5c70b3d, 0x666ea36f7879462c, 0x0e0a77c19a07df2f], dtype=np.uint64)
    jac = np.concatenate([res, np.tile(one, (n, 1))], axis=1)
    jac[~res.any(axis=1), 8:] = 0
    ok &= bool(np.array_equal(pa.g1_sum_jacobian(jac), G))
    print(f"g1_intt 2^{log_n}: {dt*1e3:.1f} ms  {128*n/dt/1e9:.3f} GB/s algorithmic  ok={ok}", flush=True)
