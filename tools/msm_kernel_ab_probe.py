"""same-box A/B of builds of the library: msm_accumulate (HIP events, one 2^20-term commitment in flight) for every root given,
interleaved twice so that clock drift does not favour one: python tools/msm_kernel_ab_probe.py <root> [<root> ...]   ('.' = working tree)"""
import os, subprocess, sys
CODE = r'''
import os, sys
sys.path.insert(0, os.path.abspath(sys.argv[1]))
import numpy as np, torch
import plonkit_amd as pa
ctx = pa.Context(0); dev = torch.device("cuda:0"); n = 1 << 20
ctx.srs_generate(n, 0, 42)
g = torch.Generator(device=dev); g.manual_seed(5)
t = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device=dev, generator=g); t[:, 3] &= (1 << 60) - 1
torch.cuda.synchronize(); ctx.set_kernel_timing(True)
for _ in range(30): ctx.msm_dev(t, n)
ks = []
for _ in range(60):
    ctx.msm_dev(t, n); ks.append(ctx.msm_last_kernel_ms())
ks.sort()
print("%.4f %.4f %.4f" % (ks[len(ks) // 2], ks[0], sum(ks) / len(ks)))
'''
roots = sys.argv[1:] or ["."]
res = {r: [] for r in roots}
for rep in range(2):
    for r in (roots if rep == 0 else roots[::-1]):
        out = subprocess.run([sys.executable, "-c", CODE, r], capture_output=True, text=True)
        line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "nan nan nan"
        res[r].append(line)
for r in roots:
    flags = open(os.path.join(r, "FLAGS")).read().strip() if os.path.exists(os.path.join(r, "FLAGS")) else "(working tree)"
    print("%-14s accumulate median/min/mean ms: %s | %s    flags: %s" % (r, res[r][0], res[r][1], flags), flush=True)
