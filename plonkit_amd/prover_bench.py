"""`prove` leg of bench.py: full PLONK prove wall-clock at the 2^log_n domain on one MI355X
(BASELINE.json metric, configs[1]: synthetic R1CS with 2^20 - 2 gates + 1 public input, monomial
tau = 42 SRS of 2^20 points already resident on the GPU)."""
import time

from . import _lib


def cold(device, log_n, circ):
    """first proof of a process as a `plonkit prove` user meets it (the reference is always in this state: it passes
    `None` precomputations, src/plonk.rs:152-159): a fresh context with only the key resident — the fixed-base table of
    the MSM, the 12 constant coset extensions, every workspace allocation and the HIP module load are inside."""
    ctx = _lib.Context(device)
    ctx.srs_generate(1 << log_n, 0, 42)
    ctx.synchronize()
    t0 = time.perf_counter()
    setup = _lib.SetupForProver(ctx, circ)
    ctx.synchronize()
    t_setup = time.perf_counter() - t0
    t0 = time.perf_counter()
    proof = setup.prove(circ)
    t_first = time.perf_counter() - t0
    phases = setup.timings_ms()
    setup.close()
    ctx.close()
    return proof, {"first_prove_s": round(t_first, 4), "setup_prepare_s": round(t_setup, 3),
                   "rounds_ms": {k: round(v, 2) for k, v in phases.items()}}


def run(ctx, log_n, reps=10):
    t0 = time.perf_counter()
    circ = _lib.Circuit.synthetic((1 << log_n) - 2)
    t_synth = time.perf_counter() - t0
    cold_proof, cold_info = cold(ctx.device, log_n, circ)
    t0 = time.perf_counter()
    setup = _lib.SetupForProver(ctx, circ)
    ctx.synchronize()
    t_setup = time.perf_counter() - t0
    assert setup.domain_size == 1 << log_n
    proof = setup.prove(circ)                      # warm-up (allocations, tables)
    runs = []
    for _ in range(reps):
        t0 = time.perf_counter()
        p = setup.prove(circ)
        dt = time.perf_counter() - t0
        assert p == proof                          # deterministic prover: identical bytes every time
        runs.append((dt, setup.timings_ms()))
    runs.sort(key=lambda r: r[0])
    best, phases = runs[len(runs) // 2]            # the MEDIAN proof of `reps` back-to-back ones (round 2 reported the best of two)
    gpu_ms = sum(v for k, v in phases.items() if k.startswith("round"))
    assert cold_proof == proof
    return {"wall_s": round(best, 4), "wall_s_min": round(runs[0][0], 4), "wall_s_max": round(runs[-1][0], 4), "proves_timed": reps,
            "cold": cold_info, "domain": 1 << log_n, "proof_bytes": len(proof),
            "rounds_ms": {k: round(v, 2) for k, v in phases.items()},
            "gpu_rounds_s": round(gpu_ms / 1e3, 4),
            "setup_prepare_s": round(t_setup, 3), "circuit_generation_s": round(t_synth, 3),
            "what": "SetupForProver::prove (witness synthesis + satisfiability check on the host, rounds 1-5 on the GPU, "
                    "Proof::write); setup_prepare = transpile + 11 iNTT, timed separately as in the reference's CLI; "
                    "wall_s = median of `proves_timed` warm proofs (tables, cached constant extensions and allocations in place), cold = first proof of a fresh context"}


def kernel_table(ctx, device):
    """HIP-event timings of the other kernels on the path, with their algorithmic HBM bytes
    (NTT 64*M, LDE4 160*N) — the per-kernel roofline rows of DESIGN.md §4."""
    import torch
    out = {}
    st = torch.cuda.Stream(device=device)
    g = torch.Generator(device=device)
    g.manual_seed(1)
    with torch.cuda.stream(st):
        for log_n in (20, 22, 24, 26):                      # 2^24 / 2^26: the shapes of the recursive prover
            n = 1 << log_n
            t = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device=device, generator=g)
            t[:, 3] &= (1 << 60) - 1
            # steady-state kernel rate: warm-up launches first and enough timed ones that the region lasts >= 10 ms — after
            # ~1 ms of low activity this chip runs the next VALU-bound kernels 8-15 % slower until its clock has ramped
            # (profiles/r03_clock_sag.txt); five launches of a 0.12 ms transform measured mostly that ramp
            warm, reps = (20, 100) if log_n <= 20 else (10, 30) if log_n <= 22 else (3, 8) if log_n <= 24 else (2, 4)
            for _ in range(warm):
                ctx.ntt_dev(t, log_n, stream=st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(reps):
                ctx.ntt_dev(t, log_n, stream=st)
            e1.record(st)
            e1.synchronize()
            ms = e0.elapsed_time(e1) / reps
            out["ntt_2^%d" % log_n] = {"ms": round(ms, 4), "algorithmic_GBs": round(64 * n / ms / 1e6, 1),
                                       "hbm_frac": round(64 * n / ms / 1e6 / 8000.0, 4), "launches_timed": reps}
            del t
        n = 1 << 20
        c = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device=device, generator=g)
        c[:, 3] &= (1 << 60) - 1
        o = torch.empty((4 * n, 4), dtype=torch.int64, device=device)
        for _ in range(10):
            ctx.lde4_dev(c, 20, o, stream=st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(30):
            ctx.lde4_dev(c, 20, o, stream=st)
        e1.record(st)
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 30
        out["lde4_2^20"] = {"ms": round(ms, 4), "algorithmic_GBs": round(160 * n / ms / 1e6, 1), "hbm_frac": round(160 * n / ms / 1e6 / 8000.0, 4), "launches_timed": 30}
        # the prover's own extension (coset-major, four polynomials per launch: lde4cm_batch_dev) and the coset iNTT back
        cs = [torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device=device, generator=g) for _ in range(4)]
        for x in cs:
            x[:, 3] &= (1 << 60) - 1
        os_ = [torch.empty((4 * n, 4), dtype=torch.int64, device=device) for _ in range(4)]
        for _ in range(5):
            ctx.lde4_coset_major_dev(cs, 20, os_, stream=st)
        e0.record(st)
        for _ in range(15):
            ctx.lde4_coset_major_dev(cs, 20, os_, stream=st)
        e1.record(st)
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 15 / 4
        out["lde4_coset_major_2^20_per_polynomial"] = {"ms": round(ms, 4), "algorithmic_GBs": round(160 * n / ms / 1e6, 1), "hbm_frac": round(160 * n / ms / 1e6 / 8000.0, 4),
                                                        "what": "four polynomials per launch, 15 launches timed"}
        for _ in range(5):
            ctx.icoset4_coset_major_dev(os_[0], 20, stream=st)
        e0.record(st)
        for _ in range(20):
            ctx.icoset4_coset_major_dev(os_[0], 20, stream=st)
        e1.record(st)
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 20
        out["icoset4_coset_major_2^20"] = {"ms": round(ms, 4), "algorithmic_GBs": round(256 * n / ms / 1e6, 1), "hbm_frac": round(256 * n / ms / 1e6 / 8000.0, 4),
                                           "what": "4n values -> 4n coefficients (64 B x 4n algorithmic), 20 launches timed"}
        del cs, os_
        del c, o
    # one 2^24-term commitment (config 5; SRS of 2^24 points generated on the GPU, its 15-copy table is 15 GiB)
    import time
    keep = ctx.srs_size()
    n = 1 << 24
    ctx.srs_generate(n, 0, 42)
    s = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device=device, generator=g)
    s[:, 3] &= (1 << 60) - 1
    torch.cuda.synchronize()
    ctx.msm_dev(s, n); ctx.msm_dev(s, n)
    t0 = time.perf_counter()
    for _ in range(3):
        ctx.msm_dev(s, n)
    ms = (time.perf_counter() - t0) / 3 * 1e3
    out["msm_2^24"] = {"ms": round(ms, 3), "Mscalar_mul_s": round(n / ms / 1e3, 1), "algorithmic_GBs": round(96 * n / ms / 1e6, 1)}
    del s
    if keep:
        ctx.srs_generate(keep, 0, 42)
    return out
