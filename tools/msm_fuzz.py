"""differential fuzz of the MSM entry points against the tau = 42 trapdoor: random lengths, offsets, batch sizes and scalar
distributions, single / batched / two-in-flight.  python tools/msm_fuzz.py [cases] [seed]"""
import os, random, sys
sys.path.insert(0, os.path.abspath(os.environ.get("PLK_AB_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # PLK_AB_ROOT=ab_old: tools/ab_build.sh
import numpy as np, torch
import plonkit_amd as pa
from oracle import oracle_lib as ol
R = ol.R_MOD
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
LOG = 17
ctx = pa.Context(0); ctx.srs_generate(1 << LOG, 0, 42)
G = ol.g1_generator()
pow42 = [1]
for _ in range(1 << LOG): pow42.append(pow42[-1] * 42 % R)
def expect(ks, off):
    acc = 0
    for i, k in enumerate(ks): acc = (acc + k * pow42[off + i]) % R
    return ol.g1_mul(G, acc)
def scalars(n):
    kind = rng.choice(["uniform", "small", "sparse", "repeat", "near_r", "mixed"])
    if kind == "uniform": return [rng.randrange(R) for _ in range(n)]
    if kind == "small": return [rng.randrange(1 << rng.choice([1, 8, 16, 40])) for _ in range(n)]
    if kind == "sparse": return [rng.randrange(R) if rng.random() < 0.05 else 0 for _ in range(n)]
    if kind == "repeat":
        pool = [rng.randrange(R) for _ in range(rng.choice([1, 2, 5]))]
        return [rng.choice(pool) for _ in range(n)]
    if kind == "near_r": return [R - 1 - rng.randrange(1 << 30) for _ in range(n)]
    return [rng.choice([0, 1, R - 1, rng.randrange(R), rng.randrange(1 << 17)]) for _ in range(n)]
bad = 0
for c in range(cases):
    n = rng.choice([1, 7, 100, 4095, 4096, 4097, 5000, 1 << 13, 12345, 1 << 15, 50000, 1 << 16, 100000, (1 << 17) - 3])
    off = rng.randrange(0, (1 << LOG) - n + 1)
    batch = rng.choice([1, 1, 2, 3, 8])
    vecs = [scalars(n) for _ in range(batch)]
    want = [expect(v, off) for v in vecs]
    dev = [torch.from_numpy(ol.fr_vec(v).view(np.int64)).to("cuda:0") for v in vecs]
    torch.cuda.synchronize()
    mode = rng.choice(["single", "batch", "flight"])
    if mode == "single":
        got = [np.asarray(ctx.msm_dev(d, n, base_offset=off)) for d in dev]
    elif mode == "batch":
        got = [np.asarray(g) for g in ctx.msm_batch_dev(dev, n, base_offset=off)]
    else:
        got = []
        pending = 0
        for d in dev:
            if pending == 2: got.append(pa.g1_sum_jacobian(ctx.msm_finish())); pending -= 1
            ctx.msm_enqueue_dev(d, n, off); pending += 1
        while pending: got.append(pa.g1_sum_jacobian(ctx.msm_finish())); pending -= 1
    ok = all(np.array_equal(g, w) for g, w in zip(got, want))
    if not ok: bad += 1
    print("case %3d n=%6d off=%6d batch=%d mode=%-6s %s" % (c, n, off, batch, mode, "ok" if ok else "MISMATCH"), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
