"""the NTT at the sizes above the tested 2^26, up to the 2-adicity of Fr (2^28: plk_ntt's limit): a sparse polynomial against its closed form at
a few indices (plain and coset 7), and the forward-inverse round trip of a dense random vector compared ON the device.
usage (GPU box): python tools/ntt_max_size_probe.py [log_n ...]   (default 27 28)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import plonkit_amd as pa
from oracle import oracle_lib as ol
R = ol.R_MOD
dev = torch.device("cuda:0")
ctx = pa.Context(0)
for log_n in [int(x) for x in sys.argv[1:]] or [27, 28]:
    n = 1 << log_n
    w = ol.omega(log_n)
    pos = [0, 1, 12345, n // 2 + 3, n - 1]
    coef = [5, R - 2, 0x1234567890abcdef, 7, R - 1]
    t = torch.zeros((n, 4), dtype=torch.int64, device=dev)
    for p, c in zip(pos, coef):
        t[p] = torch.from_numpy(np.asarray(ol.fr_mont(c), dtype=np.uint64).view(np.int64).reshape(4).copy()).to(dev)
    ok = True
    for coset in (None, 7):
        x = t.clone()
        torch.cuda.synchronize()                                  # the library runs on its own stream: torch's kernels must have finished
        t0 = time.perf_counter()
        ctx.ntt_dev(x, log_n, coset=ol.fr_mont(coset) if coset else None); ctx.synchronize()
        dt = time.perf_counter() - t0
        for k in (0, 1, 2, 977, n // 3, n // 2, n - 2, n - 1):
            pt = (coset or 1) * pow(w, k, R) % R
            want = sum(c * pow(pt, p, R) for p, c in zip(pos, coef)) % R
            got = ol.fr_ints(x[k:k + 1].cpu().numpy().view(np.uint64))[0]
            if got != want: print('  index', k, 'coset', coset, 'differs'); ok = False
        ctx.ntt_dev(x, log_n, inverse=True, coset=ol.fr_mont(coset) if coset else None); ctx.synchronize()
        if not torch.equal(x, t): print('  sparse round trip differs, coset', coset); ok = False
        del x
    # dense: uniform 253-bit values reduced to valid residues by clearing the top three bits
    g = torch.Generator(device=dev); g.manual_seed(log_n)
    d = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device=dev, generator=g)
    d[:, 3] &= (1 << 60) - 1
    e = d.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); ctx.ntt_dev(e, log_n); ctx.synchronize(); fwd = time.perf_counter() - t0
    changed = not torch.equal(e, d)
    ctx.ntt_dev(e, log_n, inverse=True); ctx.synchronize()
    if not (changed and torch.equal(e, d)): print('  dense round trip differs', changed); ok = False
    print("ntt 2^%d: sparse closed form (plain + coset 7, 8 indices each) and dense round trip ok=%s; forward %.1f ms" % (log_n, ok, fwd * 1e3), flush=True)
    del t, d, e
    torch.cuda.empty_cache()
