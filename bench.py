#!/usr/bin/env python3
"""bench.py — headline benchmark of the PLONK prove hot path on MI355X (BASELINE.json metric:
"PLONK prove wall-clock (s) + G1 MSM throughput (Mscalar·mul/s) at 2^20 domain, 1/2/4/8 GPU").

  python bench.py --gpus N --steps K --warmup W          (N > 1: spawns its own N ranks under torch.distributed.run)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (the driver's form)

A step = one KZG commitment (G1 MSM) of 2^20 uniform scalars per GPU against that GPU's resident
shard of a tau = 42 monomial SRS (BASELINE.json configs[1]; with N GPUs the job is one commitment
of N*2^20 terms with the bases split across ranks, partial sums exchanged over RCCL — "weak").
`value` = total scalar·muls per second over all ranks, inputs resident in HBM.  Up to three commitments are in flight
(the library's FIFO).  The headline region is exactly what the contract says: W untimed warm-up steps, then K timed steps
between barrier + synchronize brackets, nothing else before it (`--settle-steps` defaults to 0).  Because a step is
~1.3 ms, W = 5 is only ~7 ms of load and the GPU reaches its sustained rate after ~70 ms of it; the line therefore also
carries `sustained` = the same K-step region timed once more after 50 further commitments (value_sustained), so both
protocols can be read from one run.

The one JSON line also carries
  roofline      dominant kernel msm_accumulate.  `kernel_ms` is its HIP-event duration with ONE commitment in flight
                (--pipeline-depth 1 region, timed right after the headline region); `kernel_ms_pipelined` is what the
                same kernel takes inside the headline region, where it shares the GPU with the bucket reduction of the
                previous commitment.  `valu` prices the kernel against the measured v_mad_u64_u32 issue rate.
  cpu_baseline  the oracle's restatement of bellman's dense_multiexp / best_fft / prove on the host cores (kind "port"),
                MSM and NTT at 2^20, whole prove at the 2^20 domain; with PLONKIT_REF_BIN set, also the real
                `plonkit prove` on the same files (kind "reference").
  prove         (N = 1) wall-clock of a full prove at the 2^20 domain: warm (tables, cached extensions, allocations
                in place) and cold (first proof of a fresh context).
  strong        (N > 1) ONE 2^24-term commitment with the SRS split 2^24/N per rank (BASELINE.json configs[2]) next
                to the same commitment on rank 0 alone: the figure north_star's ">= 6x at 1 -> 8 GPUs" reads.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

# the library keeps up to three commitments in flight on three streams; with HIP's default of 4 hardware queues per device
# those streams can land on one queue and serialise (measured: 2.04 ms per commitment instead of 1.66 ms)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# dmabuf IPC: on this driver RCCL (and any device-memory sharing) across processes fails with `hipIpcGetMemHandle: invalid argument`
# unless this is set BEFORE the first HIP call of every rank — whoever launched us (this script's own spawn_ranks, the driver's
# torch.distributed.run, a batch system): set it here, not only in the spawner
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
ALGO_BYTES_PER_TERM = 96         # 64 B base + 32 B scalar
# PMC traffic of msm_accumulate for ONE 2^20-term launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of
# `bench.py --msm-only --pipeline-depth 1`, profiles/r05_pmc_traffic.txt, 77 launches, spread < 0.1 %; rounds 3-4: 1,016 k + 50-52 k):
# 1,015,901 KB fetched (15.7 M gathers of one 64-byte point each from the 0.94 GiB fixed-base table = 1.007 GB, + the 63 MB of
# entries, read once: consistent with 64 B per gather, i.e. no over-fetch; these are not wide coalesced streams, so the
# guide's x2 correction for 16 B/lane streaming reads does not apply) + 101,189 KB written (lane partial sums: 50 MB of payload in 144-byte
# flushes, counted as partial 64-byte requests — see the profile file).  Only for --log-n 20.  It is NOT measured in the run that prints the
# line (PMC collection needs the profiler): `traffic_source` says so.
PMC_TRAFFIC_BYTES_2POW20 = (1015901 + 101189) * 1024
PMC_TRAFFIC_SOURCE = ("not measured in this run: FETCH_SIZE + WRITE_SIZE of msm_accumulate from separate rocprofv3 --pmc passes of the same "
                      "command on the same code, profiles/r05_pmc_traffic.txt")
MSM_WINDOWS = 15                 # 17-bit signed windows over the 254-bit scalars: mixed additions per term
# VALU yardsticks.  Hardware: v_mad_u64_u32 issues at 576.1 G wave-instructions/s chip-wide
# (profiles/r01_ubench_int.txt, k_mad64: 4.27 cycles per wave-instruction per SIMD at 2.4 GHz) = 36.87 T lane-mads/s;
# one XYZZ mixed addition on the 9 x 29-bit layer is 1467 of them (6 products of 162 + 2 squarings of 126 + one fused
# double product of 243; counted in the hot block of the code object) -> 25.1 G mixed additions/s if nothing but the
# multiply-adds were issued.  The loop as compiled (round 3: 1467 mads + 625 other VALU instructions per addition, products in
# lockstep pairs) reaches 16.3 G/s in isolation (tools/ubench_w, profiles/r03_ubench_mulw_forms.txt; 15.6 in round 2).
MAD_RATE_TLANE_S = 576.11e9 * 64 / 1e12
MADS_PER_MIXED_ADD = 1467
VALU_PEAK_GMADD = MAD_RATE_TLANE_S * 1e3 / MADS_PER_MIXED_ADD
LOOP_ISOLATED_GMADD = 16.3


FAILED_LEGS = []                 # (leg, repr(exception), is_correctness) of every secondary leg that raised: the line keeps going, the exit status does not lie
CURRENT_LEG = [None]             # what the leg watchdog names when it fires
NATIVE_EXCHANGE = [True]         # False: the library's communicator could not be opened, partial sums travel through torch.distributed


def leg(name, fn):
    """run a secondary leg of the line: an exception becomes {"error": ...} in its place AND an entry of the top-level `failed_legs`;
    an AssertionError / a failed byte-identity or verification check also makes the process exit non-zero after the line is printed"""
    CURRENT_LEG[0] = name
    try:
        if os.environ.get("PLK_BENCH_TEST_STALL_LEG") == name:     # TEST HOOK (tests/test_gpu_sharded_prove.py): this leg never returns
            time.sleep(3600)
        return fn()
    except Exception as exc:                                       # noqa: BLE001
        FAILED_LEGS.append((name, repr(exc), isinstance(exc, AssertionError) or "differs" in repr(exc) or "mismatch" in repr(exc)))
        return {"error": repr(exc)}


def start_leg_watchdog(rank, line_box):
    """N > 1 only.  The legs after the headline hold collectives of RCCL and of the library's own communicator; a rank parked in one
    that its peers never enter (a host-blocking ncclGroupEnd, a leg that desynchronised) would hang the whole launch and cost the
    headline line already measured.  After PLK_BENCH_LEG_TIMEOUT_S (default 420; the legs take about a minute on 8 GPUs) rank 0
    prints the line with what it has — the stuck leg named in `failed_legs` — and the process exits with status 3; the launcher
    then ends the other ranks (they fire 20 s later on their own if it does not)."""
    import threading
    limit = float(os.environ.get("PLK_BENCH_LEG_TIMEOUT_S", "420"))

    def fire():
        line = line_box[0]
        if rank == 0 and line is not None:
            line["failed_legs"] = [{"leg": a, "error": b, "correctness": c} for a, b, c in FAILED_LEGS] + [
                {"leg": CURRENT_LEG[0], "error": "did not return within %g s (PLK_BENCH_LEG_TIMEOUT_S): the line is printed by the watchdog, "
                                                 "the legs after it never ran" % limit, "correctness": False}]
            emit(line)
        os._exit(3)

    t = threading.Timer(limit + (0 if rank == 0 else 20), fire)
    t.daemon = True
    t.start()
    return t


def all_ok(dist, device, ok):
    """collective AND of a per-rank flag: every rank calls it, so a rank that failed locally does not leave the others in a barrier"""
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=red_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


def red_device(device):
    """where the timing reductions live: the GPU over RCCL, the host when torch.distributed runs over gloo (shared-device test)"""
    return torch.device("cpu") if os.environ.get("PLK_BENCH_SHARE_DEVICE") else device


def rand_scalars(n, seed, device):
    """uniform 252-bit residues (valid Montgomery Fr representatives), generated on the GPU"""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    t = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device=device, generator=g)
    t[:, 3] &= (1 << 60) - 1
    return t


# ------------------------------------------------------------------------------------------ CPU baselines
def cpu_quota_cores():
    """CPUs' worth of time the container may use per scheduling period (cgroup cpu.max / cfs quota), or None when unlimited.  The GPU
    boxes of this pool show 256 logical CPUs (2 x EPYC 9575F) but run the job under `cpu.max = 1600000 100000`: sixteen CPUs of quota —
    beyond ~32 runnable threads the CFS throttles the whole group every period, which is the "collapse" rounds 1-3 attributed to the
    port's memory behaviour (profiles/r04_cpu_host_quota.txt: nr_throttled 401 of 593 periods during a thread sweep)."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:                                              # noqa: BLE001
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except Exception:                                              # noqa: BLE001
        return None


CPU_BEST = {"split": "chunks", "threads": 16}          # filled by cpu_msm_baseline: the fastest (work split, thread count) of the port on this host


def cpu_msm_baseline(ctx, log_sample, seed):
    """bellman dense_multiexp restatement (oracle/, kind "port") on the host cores, bounded sample.  SURVEY.md §8(d) says
    "all host cores": the port is timed at 16 / 64 / 128 / every logical core in two work splits — "chunks" = bellman 0.3.2's
    own (each thread owns 2^c - 1 buckets per window and folds them whatever its share of the terms: 2 * 16383 full additions
    per thread per window at 2^20, which outweighs the useful work from ~32 threads up — that is the algorithm, not this
    file), and "windows" = one task per (window, chunk) with thread-local, first-touched buckets, as later bellman revisions
    split the work, which is what a 256-thread host needs.  The fastest of all of them is the baseline."""
    from oracle import oracle_lib as ol          # checker / baseline only — never on the product path
    m = 1 << log_sample
    bases = ctx.srs_download(0, m)
    rng = np.random.default_rng(seed)
    s = rng.integers(0, 1 << 62, size=(m, 4), dtype=np.uint64)
    s[:, 3] &= np.uint64((1 << 60) - 1)
    ncpu = os.cpu_count() or 1
    ol.msm(bases[:1024], s[:1024], threads=min(ncpu, 16))     # warm the library
    quota = cpu_quota_cores()
    counts = sorted({min(ncpu, 16), min(ncpu, 32), min(ncpu, 64), min(ncpu, 128), ncpu})
    best = None
    tried = {"chunks": {}, "windows": {}}
    for split in ("windows", "chunks"):
        for cores in counts:
            if split == "chunks" and cores > 64 and best and best[0] < 2.0:
                continue                                   # (bellman 0.3.2's split beyond 64 threads: seconds per call, measured in profiles/, never the best)
            t0 = time.perf_counter()
            ref = ol.msm(bases, s, threads=cores, split=split)
            dt = time.perf_counter() - t0
            tried[split][str(cores)] = round(m / dt / 1e6, 3)
            if best is None or dt < best[0]:
                best = (dt, cores, ref, split)
    dt, cores, ref, split = best
    CPU_BEST["split"], CPU_BEST["threads"] = split, cores
    return {"value": m / dt / 1e6, "unit": "Mscalar·mul/s", "cores": cores, "kind": "port", "host_cores": ncpu, "host_cpu_quota_cores": quota, "work_split": split,
            "by_threads_Mscalar_mul_s": tried,
            "sample": "one dense_multiexp (c=ceil(ln n)) of 2^%d uniform scalars, %.2f s; best of %d thread counts x 2 work splits" % (log_sample, dt, len(counts))}, ref, s


def cpu_msm_rows(ctx, device, log_n=20, big_log_n=24):
    """the four scalar distributions of SURVEY.md §8(d) at 2^20 and uniform scalars at 2^24, the
    oracle's dense_multiexp restatement (16 threads, its best setting) beside the HIP commitment on the same scalars and
    bases; every pair of results must be the same group element."""
    from oracle import oracle_lib as ol
    n = 1 << log_n
    keep = ctx.srs_size()
    ctx.srs_generate(n, 0, 42)
    bases = ctx.srs_download(0, n)
    rng = np.random.default_rng(77)
    uni = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    uni[:, 3] &= np.uint64((1 << 60) - 1)
    one, minus_one = ol.fr_vec([1])[0], ol.fr_vec([ol.R_MOD - 1])[0]
    small = ol.fr_vec(list(range(1 << 16)))
    wl = uni.copy()
    pick = rng.random(n)
    wl[pick < 0.5] = 0
    idx = np.nonzero((pick >= 0.5) & (pick < 0.75))[0]
    wl[idx] = small[rng.integers(0, 1 << 16, size=idx.shape[0])]
    cases = {"uniform": uni, "witness_like": wl, "all_ones": np.tile(one, (n, 1)), "all_r_minus_1": np.tile(minus_one, (n, 1))}
    cores = CPU_BEST["threads"]
    ol.MSM_SPLIT[0] = CPU_BEST["split"]
    out = {}
    for name, sc in cases.items():
        t0 = time.perf_counter()
        ref = ol.msm(bases, sc, threads=cores)
        cpu_ms = (time.perf_counter() - t0) * 1e3
        d = torch.from_numpy(np.ascontiguousarray(sc).view(np.int64)).to(device)
        torch.cuda.synchronize()
        got = ctx.msm_dev(d, n)
        t0 = time.perf_counter()
        for _ in range(3):
            ctx.msm_dev(d, n)
        gpu_ms = (time.perf_counter() - t0) / 3 * 1e3
        out["2^%d_%s" % (log_n, name)] = {"cpu_ms": round(cpu_ms, 1), "gpu_ms": round(gpu_ms, 3), "cores": cores, "same_point": bool(np.array_equal(got, ref))}
    del bases
    # 2^24 terms (MSM at 2^20 / 2^24 with the four distributions).  The CPU port is timed on the uniform case only (6 s;
    # the bounded-sample rule); the other three are checked against the tau = 42 trapdoor instead: sum s_i tau^i G = (sum s_i 42^i) G
    # is ONE fixed-base multiplication on the host, so every GPU result is still compared with an independent value.
    m = 1 << big_log_n
    ctx.srs_generate(m, 0, 42)
    bases = ctx.srs_download(0, m)
    big = rng.integers(0, 1 << 62, size=(m, 4), dtype=np.uint64)
    big[:, 3] &= np.uint64((1 << 60) - 1)
    t0 = time.perf_counter()
    ref = ol.msm(bases, big, threads=cores)
    cpu_ms = (time.perf_counter() - t0) * 1e3
    del bases
    wl = big.copy()
    pick = rng.random(m)
    wl[pick < 0.5] = 0
    idx = np.nonzero((pick >= 0.5) & (pick < 0.75))[0]
    wl[idx] = small[rng.integers(0, 1 << 16, size=idx.shape[0])]
    for name, sc in (("uniform", big), ("witness_like", wl), ("all_ones", None), ("all_r_minus_1", None)):
        if sc is None:
            d = torch.from_numpy(np.ascontiguousarray(one if name == "all_ones" else minus_one).view(np.int64)).to(device).repeat(m, 1).contiguous()
        else:
            d = torch.from_numpy(np.ascontiguousarray(sc).view(np.int64)).to(device)
        torch.cuda.synchronize()
        got = ctx.msm_dev(d, m)
        t0 = time.perf_counter()
        ctx.msm_dev(d, m)
        gpu_ms = (time.perf_counter() - t0) * 1e3
        row = {"gpu_ms": round(gpu_ms, 3), "Mscalar_mul_s": round(m / gpu_ms / 1e3, 1)}
        if name == "uniform":
            row.update({"cpu_ms": round(cpu_ms, 1), "cores": cores, "same_point": bool(np.array_equal(got, ref))})
        else:
            if name == "witness_like":
                k = ol.poly_eval(sc, 42)                    # sum s_i 42^i (Horner, C)
            else:                                            # sum_{i<m} 42^i = (42^m - 1) / 41, times 1 or (r - 1)
                geo = (pow(42, m, ol.R_MOD) - 1) * pow(41, ol.R_MOD - 2, ol.R_MOD) % ol.R_MOD
                k = geo if name == "all_ones" else (ol.R_MOD - 1) * geo % ol.R_MOD
            row.update({"same_point": bool(np.array_equal(got, ol.g1_mul(ol.g1_generator(), k))), "checked_against": "tau = 42 trapdoor (one host scalar multiplication)"})
        out["2^%d_%s" % (big_log_n, name)] = row
        del d
    del big, wl
    if keep:
        ctx.srs_generate(keep, 0, 42)
    return out


def cpu_g1_intt_row(ctx, device, log_n=16):
    """BASELINE.md §3 row B5 on a bounded sample: Crs::<Lagrange>::from_powers (the G1 iNTT of dump-lagrange) of the first
    2^16 crs_42 points — 2^15 * 16 scalar multiplications of 254 bits on every host core; 2^20 would be ~25x more (the GPU
    side of the full-size configuration is kernels["g1_intt_2^20"])"""
    from oracle import oracle_lib as ol
    n = 1 << log_n
    keep = ctx.srs_size()
    ctx.srs_generate(n, 0, 42)
    pts = ctx.srs_download(0, n)
    cores = CPU_BEST["threads"]
    t0 = time.perf_counter()
    ref = ol.g1_intt(pts, log_n, threads=cores)
    cpu_s = time.perf_counter() - t0
    out = torch.empty((n, 8), dtype=torch.int64, device=device)
    ctx.g1_intt_srs_dev(log_n, out)
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.g1_intt_srs_dev(log_n, out)
    ctx.synchronize()
    gpu_s = time.perf_counter() - t0
    same = bool(np.array_equal(out.cpu().numpy().view(np.uint64), ref))
    if keep:
        ctx.srs_generate(keep, 0, 42)
    return {"points": n, "cpu_s": round(cpu_s, 2), "gpu_s": round(gpu_s, 4), "cores": cores, "kind": "port", "identical": same}


def cpu_ntt_baseline(sizes=(20, 22)):
    """bellman best_fft restatement: serial radix-2 below log2(cpus), the classic split into
    2^log_cpus interleaved sub-FFTs above; timed at 16 threads and at every host core, the better one is reported"""
    from oracle import oracle_lib as ol
    ncpu = os.cpu_count() or 1
    out = {}
    for log_n in sizes:
        n = 1 << log_n
        rng = np.random.default_rng(log_n)
        a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
        a[:, 3] &= np.uint64((1 << 60) - 1)
        best = None
        for cores in sorted({min(ncpu, 16), ncpu}):
            t0 = time.perf_counter()
            ol.ntt(a, log_n, threads=cores)
            dt = time.perf_counter() - t0                    # (includes one 32*n-byte copy, as bellman's Polynomial::fft moves its input)
            if best is None or dt < best[0]:
                best = (dt, cores)
        out["ntt_2^%d" % log_n] = {"ms": round(best[0] * 1e3, 2), "cores": best[1], "kind": "port",
                                   "algorithmic_GBs": round(64 * n / best[0] / 1e9, 2)}
        if log_n == 22:                                                # the coset NTT on the 4N domain of a 2^20-gate proof
            t0 = time.perf_counter()
            ol.ntt(a, log_n, coset=7, threads=best[1])
            out["coset_ntt_2^22"] = {"ms": round((time.perf_counter() - t0) * 1e3, 2), "cores": best[1], "kind": "port"}
    return out


def cpu_prove_baseline(ctx, log_domain):
    """the oracle's restatement of the whole prove at the headline domain (kind "port"), beside the HIP prover on the SAME
    circuit, witness and SRS; the two proofs must be byte-identical.  Compiled code end to end since round 4: the files are
    parsed, the circuit synthesised with the witness and the gates checked by the C front end of oracle/c/oracle.c
    (single-threaded, as the reference's synthesis is), the rounds are OpenMP C kernels on the thread count and MSM work
    split that cpu_msm_baseline found fastest on this host; `cpu_c_share` = the part of `cpu_s` spent inside liboracle.so
    (the remainder is numpy buffer handling)."""
    import plonkit_amd as pa
    from oracle import oracle_lib as ol, plonk_oracle as po      # checker / baseline only
    circ = pa.Circuit.synthetic((1 << log_domain) - 2)
    raw_r1cs, raw_wtns = circ.export("r1cs"), circ.export("wtns")
    srs_keep = ctx.srs_size()
    ctx.srs_generate(1 << log_domain, 0, 42)
    crs = po.Crs(ctx.srs_download(0, 1 << log_domain), b"\x01" * 256)
    # the MSM sweep's best thread count can be 64 on a box whose cgroup quota is 16 CPUs (a lucky burst); a 16-26 s prove cannot burst:
    # measured 16.2 s at 32 threads against 23.8-26.3 s at 64 on such boxes.  With a quota, the prove runs on at most twice its CPUs.
    quota = cpu_quota_cores()
    prove_threads = CPU_BEST["threads"] if not quota else max(1, min(CPU_BEST["threads"], int(2 * quota)))
    ol.set_threads(prove_threads)
    ol.MSM_SPLIT[0] = CPU_BEST["split"]
    t0 = time.perf_counter()
    rf, wit = po.load_r1cs_flat(raw_r1cs), ol.wtns_parse(raw_wtns)      # row B2 counts the parsing
    parse_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    S = po.setup_flat(rf)
    setup_s = time.perf_counter() - t0
    ol.C_SECONDS[0] = 0.0
    t0 = time.perf_counter()
    ref = po.write_proof(po.prove(rf, wit, crs, S))
    cpu_s = time.perf_counter() - t0
    c_s = ol.C_SECONDS[0]
    ol.set_threads(None)
    ol.MSM_SPLIT[0] = "chunks"
    setup = pa.SetupForProver(ctx, circ)
    setup.prove(circ)
    t0 = time.perf_counter()
    got = setup.prove(circ)
    gpu_s = time.perf_counter() - t0
    setup.close(); circ.close()
    if srs_keep:
        ctx.srs_generate(srs_keep, 0, 42)
    return {"domain": 1 << log_domain, "cpu_s": round(cpu_s, 3), "cpu_c_kernels_s": round(c_s, 3), "cpu_c_share": round(c_s / cpu_s, 3),
            "cpu_setup_s": round(setup_s, 2), "cpu_parse_s": round(parse_s, 2), "cpu_whole_s": round(parse_s + setup_s + cpu_s, 1),
            "gpu_s": round(gpu_s, 5), "threads": prove_threads, "msm_work_split": CPU_BEST["split"], "host_cores": os.cpu_count(),
            "host_cpu_quota_cores": cpu_quota_cores(),
            "kind": "port", "proof_bytes_identical": bool(got == ref),
            "speedup_vs_cpu_total": round(cpu_s / gpu_s, 1), "speedup_vs_cpu_c_kernels": round(c_s / gpu_s, 1),
            "sample": "one prove (synthesis + gate check, rounds 1-5: 11 MSM + 25 NTT-equivalents) of a synthetic 2^%d-gate circuit; "
                      "a port of bellman's algorithms, not the reference binary (no Rust toolchain in this image)" % log_domain}


def reference_binary_baseline(log_domain):
    """if PLONKIT_REF_BIN names a real `plonkit` binary, time its `prove` on the same .r1cs / .wtns /
    key files this library is given, byte-compare the two proof.bin and let the reference verify ours.  The files are
    produced through the C ABI (plk_circuit_export, `plonkit setup` of this package for the tau = 42 key)."""
    ref = os.environ.get("PLONKIT_REF_BIN")
    if not ref:
        return None
    import plonkit_amd as pa
    ours = os.path.join(os.path.dirname(pa.lib_path()), "plonkit")
    d = tempfile.mkdtemp(prefix="plonkit_ref_")
    f = lambda name: os.path.join(d, name)
    circ = pa.Circuit.synthetic((1 << log_domain) - 2)
    open(f("circuit.r1cs"), "wb").write(circ.export("r1cs"))
    open(f("witness.wtns"), "wb").write(circ.export("wtns"))
    circ.close()
    out = {"binary": ref, "domain": 1 << log_domain, "kind": "reference", "cores": os.cpu_count()}
    try:
        subprocess.check_call([ours, "setup", "-p", str(log_domain), "-m", f("key.bin"), "--overwrite"], stderr=subprocess.DEVNULL)
        subprocess.check_call([ours, "export-verification-key", "-m", f("key.bin"), "-c", f("circuit.r1cs"), "-v", f("vk.bin"), "--overwrite"], stderr=subprocess.DEVNULL)
        common = ["-m", f("key.bin"), "-c", f("circuit.r1cs"), "-w", f("witness.wtns")]
        t0 = time.perf_counter()
        subprocess.check_call([ours, "prove"] + common + ["-p", f("proof_ours.bin"), "-j", f("pj_ours.json"), "-i", f("ij_ours.json"), "--overwrite"], stderr=subprocess.DEVNULL)
        out["ours_cli_prove_s"] = round(time.perf_counter() - t0, 3)
        t0 = time.perf_counter()
        subprocess.check_call([ref, "prove"] + common + ["-p", f("proof_ref.bin"), "-j", f("pj_ref.json"), "-i", f("ij_ref.json"), "--overwrite"],
                              stderr=subprocess.DEVNULL, timeout=3600)
        out["reference_cli_prove_s"] = round(time.perf_counter() - t0, 3)
        out["proof_bytes_identical"] = open(f("proof_ours.bin"), "rb").read() == open(f("proof_ref.bin"), "rb").read()
        out["reference_verifies_ours"] = subprocess.call([ref, "verify", "-p", f("proof_ours.bin"), "-v", f("vk.bin")], stderr=subprocess.DEVNULL) == 0
        out["speedup_whole_cli"] = round(out["reference_cli_prove_s"] / out["ours_cli_prove_s"], 1)
    except Exception as exc:                                       # noqa: BLE001 — a broken reference binary must not cost the bench line
        out["error"] = repr(exc)
    return out


def poseidon_shaped_circuit(perms=7, seed=84, rp=20):
    """the 2^12-domain circuit of the by-domain table: a circom-Poseidon-SHAPED hash chain (tests/gen/poseidon_like.py — the shape of the
    reference's CI circuit test/circuits/poseidon, whose artifacts are not in its tree; S-box inputs that are linear combinations of up to 24
    signals, folded through the d column: parity unpinned).  The generator only makes INPUTS and imports nothing of oracle/
    (its own xoshiro256**); what is timed is the product's prover."""
    import plonkit_amd as pa
    from tests.gen import poseidon_like as pl
    ni, nv, cons, wit = pl.build(perms, seed, rp=rp)
    js = pl.as_circom_json(ni, nv, cons)
    return pa.Circuit(json.dumps(js).encode(), True, json.dumps([str(x) for x in wit]).encode(), True), js, wit


def prove_by_domain(ctx, main_row, with_cpu):
    """prove wall clock at the 2^12 (Poseidon-shaped), 2^16, 2^20, 2^22 and 2^24 domains — median warm proof, hbm fraction by
    SURVEY.md 8(d)'s 8416 B per domain point, and the CPU port's seconds where it runs in < 20 s (2^12, 2^16 here; 2^20 is cpu_baseline.prove).
    Every proof is accepted by the host verifier (real pairing) in the same run; the 2^12 and 2^16 proofs equal the CPU port's byte for byte."""
    import plonkit_amd as pa
    from plonkit_amd import prover_bench
    keep = ctx.srs_size()
    out = {}
    for log_n, reps in ((12, 30), (16, 30), (20, 0), (22, 4), (24, 2)):
        key = "2^%d" % log_n
        if log_n == 20 and isinstance(main_row, dict) and "wall_s" in main_row:
            out[key] = {"domain": 1 << 20, "wall_s": main_row["wall_s"], "hbm_frac": main_row.get("hbm_frac"), "see": "prove"}
            continue
        ctx.srs_generate(1 << log_n, 0, 42)
        if log_n == 12:
            circ, js, wit = poseidon_shaped_circuit()
        else:
            circ = pa.Circuit.synthetic((1 << log_n) - 2)
        row, proof = prover_bench.prove_row(ctx, circ, max(reps, 2))
        row["circuit"] = "poseidon-shaped hash chain (tests/gen/poseidon_like.py), 11/11 commitments live" if log_n == 12 else "synthetic, 2^%d - 2 gates" % log_n
        if with_cpu and log_n <= 16:
            from oracle import oracle_lib as ol, plonk_oracle as po          # CPU port: baseline / checker only
            crs = po.Crs(ctx.srs_download(0, 1 << log_n), b"\x01" * 256)
            quota = cpu_quota_cores()
            ol.set_threads(CPU_BEST["threads"] if not quota else max(1, min(CPU_BEST["threads"], int(2 * quota))))
            ol.MSM_SPLIT[0] = CPU_BEST["split"]
            try:
                if log_n == 12:
                    r_o = po.load_r1cs_json(js)
                    S = po.setup(r_o)
                    t0 = time.perf_counter()
                    ref = po.write_proof(po.prove(r_o, wit, crs, S))
                else:
                    rf, w = po.load_r1cs_flat(circ.export("r1cs")), ol.wtns_parse(circ.export("wtns"))
                    S = po.setup_flat(rf)
                    t0 = time.perf_counter()
                    ref = po.write_proof(po.prove(rf, w, crs, S))
                row["cpu_port_s"] = round(time.perf_counter() - t0, 3)
                row["proof_bytes_identical"] = bool(ref == proof)
            finally:
                ol.set_threads(None)
                ol.MSM_SPLIT[0] = "chunks"
        circ.close()
        out[key] = row
    if keep:
        ctx.srs_generate(keep, 0, 42)
    return out


def cli_ci_shape():
    """the reference's CI run itself (.github/workflows/integration-test.yml:105-154): `setup --power 20`, then export-verification-key / prove /
    dump-lagrange / prove -l / verify of a 2^12-domain Poseidon(-shaped) circuit against that 2^20 key"""
    from plonkit_amd import prover_bench
    circ, _, _ = poseidon_shaped_circuit()
    files = (circ.export("r1cs"), circ.export("wtns"))
    circ.close()
    return prover_bench.cli_table(12, circuit_files=files, key_log_n=20)


NOTES = ("field definitions: DESIGN.md section 5 (bench line).  value = G1 MSM throughput, scalars resident in HBM; roofline = msm_accumulate, "
         "achieved = 96 B x 2^20 / kernel_ms (HIP events, one commitment in flight), bound by VALU issue (valu.*), traffic from profiles/ (PMC, not this run); "
         "cpu_baseline = oracle/ port of bellman's algorithms on this host's cores under its cgroup quota (kind port: never the reference binary); "
         "prove = SetupForProver::prove at the 2^20 domain, median warm proof; hbm_frac = SURVEY 8(d) bytes (8416 B per domain point) / wall / 8 TB/s; "
         "cli = whole-process seconds of this package's plonkit binary in the reference CI's command order; "
         "the full line with every secondary field is written to bench_line_full.json beside bench.py")


PROSE_KEYS = ("what", "note", "sample", "derivation", "algorithmic_bytes_what", "traffic_source", "checked_against", "survey_items", "this_prover_items")


def strip_prose(obj):
    """a copy of a (nested) row without its explanatory strings"""
    if isinstance(obj, dict):
        return {k: strip_prose(v) for k, v in obj.items() if k not in PROSE_KEYS and not (isinstance(v, str) and len(v) > 200)}
    if isinstance(obj, list):
        return [strip_prose(v) for v in obj]
    return obj


def compact_line(full):
    """the printed line must fit the driver's 8 KB tail (round 5's 14 KB line lost prove.wall_s and cpu_baseline.prove there): the numbers the
    review reads, no prose — the prose is NOTES / DESIGN.md section 5, the complete record is bench_line_full.json"""
    def pick(d, keys):
        return {k: d[k] for k in keys if isinstance(d, dict) and k in d} if isinstance(d, dict) else d
    L = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                              "dtype", "data") if k in full}
    cfg = full.get("config", {})
    L["config"] = {"workload": "Pippenger G1 MSM (KZG commitment), 2^%d uniform scalars per GPU, tau=42 monomial SRS sharded by rank (BASELINE configs[1])"
                               % int(np.log2(cfg.get("terms_per_gpu", 1 << 20))),
                   "terms_per_gpu": cfg.get("terms_per_gpu"), "parallelism": cfg.get("parallelism"), "exchange": cfg.get("exchange"),
                   "pipeline_depth": cfg.get("pipeline_depth"), "settle_steps": cfg.get("settle_steps")}
    if "comm_ranks" in cfg:
        L["config"]["comm_ranks"] = cfg["comm_ranks"]
    r = full.get("roofline", {})
    L["roofline"] = pick(r, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "kernel_ms_pipelined",
                             "ms_per_step_one_in_flight", "algorithmic_bytes", "valu_frac"))
    if isinstance(r.get("valu"), dict):
        L["roofline"]["valu"] = pick(r["valu"], ("achieved_gmadd_s", "peak_gmadd_s", "frac_of_isolated_loop"))
    src = r.get("traffic_source") or ""
    L["roofline"]["traffic_source"] = ("this run (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child runs of --msm-only)" if src.startswith("measured in this run")
                                       else "profiles/ (rocprofv3 --pmc of the same command; not this run)")
    if "traffic_over_algorithmic" in r:
        L["roofline"]["traffic_over_algorithmic"] = r["traffic_over_algorithmic"]
    if "value_sustained" in full:
        L["value_sustained"] = full["value_sustained"]
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        c = pick(cb, ("value", "unit", "cores", "kind", "host_cores", "host_cpu_quota_cores", "work_split", "matches_gpu"))
        c["value"] = round(c.get("value", 0.0), 3)
        c["sample"] = "one dense_multiexp of 2^20 uniform scalars (best of thread counts x work splits)"
        if isinstance(cb.get("prove"), dict):
            c["prove"] = pick(cb["prove"], ("domain", "cpu_s", "cpu_whole_s", "gpu_s", "threads", "proof_bytes_identical", "speedup_vs_cpu_total", "error"))
        if isinstance(cb.get("ntt"), dict):
            c["ntt_ms"] = {k: v.get("ms") for k, v in cb["ntt"].items() if isinstance(v, dict)} or cb["ntt"]
        if isinstance(cb.get("g1_intt"), dict):
            c["g1_intt_2^16"] = pick(cb["g1_intt"], ("cpu_s", "gpu_s", "identical", "error"))
        if isinstance(cb.get("msm_rows"), dict):
            c["msm_rows"] = {k: ([v.get("cpu_ms"), v.get("gpu_ms"), v.get("same_point")] if isinstance(v, dict) else v) for k, v in cb["msm_rows"].items()}
            c["msm_rows_columns"] = ["cpu_ms", "gpu_ms", "same_point"]
        if "reference_binary" in cb:
            c["reference_binary"] = cb["reference_binary"]
        L["cpu_baseline"] = c
    pv = full.get("prove")
    if isinstance(pv, dict) and full.get("n_gpus", 1) > 1:                  # the sharded prove of an N > 1 line: its own (short) rows
        L["prove"] = strip_prose(pv)
    elif isinstance(pv, dict):
        P = pick(pv, ("wall_s", "wall_s_min", "wall_s_max", "proves_timed", "gpu_rounds_s", "rounds_ms", "domain", "commitments_nonempty", "hbm_frac",
                      "hbm_frac_this_prover", "setup_prepare_s", "error", "world", "mode"))
        if isinstance(pv.get("cold"), dict):
            P["cold_first_prove_s"] = pv["cold"].get("first_prove_s")
        if isinstance(pv.get("dense"), dict):
            P["dense"] = pick(pv["dense"], ("wall_s", "commitments_nonempty", "verified", "parity", "error"))
        for key, short in (("throughput", "in_flight_2"), ("throughput_in_flight_3", "in_flight_3"), ("throughput_dense", "in_flight_2_dense")):
            if isinstance(pv.get(key), dict):
                P[short] = pick(pv[key], ("ms_per_proof", "proofs_per_s", "byte_identical_to_sequential", "error"))
        for key in ("by_domain", "scatter"):
            if key in pv:
                P[key] = pv[key]
        L["prove"] = P
    for key in ("cli", "cli_ci_shape"):
        if isinstance(full.get(key), dict):
            L[key] = {k: v for k, v in full[key].items() if k not in ("prove_phases_s", "files_MB", "where")}
    if isinstance(full.get("kernels"), dict):
        L["kernels_ms"] = {k: (v.get("ms") if isinstance(v, dict) else v) for k, v in full["kernels"].items()}
    for key in ("strong", "strong_value", "strong_unit", "strong_scaling_vs_1gpu", "prove_throughput"):
        if key in full:
            L[key] = strip_prose(full[key])
    L["failed_legs"] = full.get("failed_legs", [])
    L["notes"] = NOTES
    return L


WRITE_FULL_RECORD = [True]           # False in --msm-only runs (profiling / counter children): they must not overwrite the record of the run that started them


def emit(line):
    """prints the compact line (<= ~7 KB) and leaves the full record beside bench.py"""
    try:
        if WRITE_FULL_RECORD[0]:
            with open(os.path.join(ROOT, "bench_line_full.json"), "w") as fh:
                json.dump(line, fh, ensure_ascii=False, indent=1)
    except OSError:
        pass
    out = json.dumps(compact_line(line), ensure_ascii=False)
    if len(out.encode()) > 7600:                                   # never past the driver's tail: drop the least-read tables first
        c = compact_line(line)
        for key in ("notes", "cli_ci_shape", "kernels_ms"):
            c.pop(key, None)
            out = json.dumps(c, ensure_ascii=False)
            if len(out.encode()) <= 7600:
                break
    print(out, flush=True)


def measure_traffic_live(log_n):
    """roofline.traffic measured in THIS run (round-5 review: the figure was a constant pasted from profiles/): two child runs of this script's --msm-only
    region (one commitment in flight) under `rocprofv3 --pmc <counter> --kernel-trace` — FETCH_SIZE and WRITE_SIZE need separate passes (TCC counter slots:
    MI355X_MICROARCH.md) and no other tracing —, the median KB per msm_accumulate launch of each, summed.  No correction factor is applied: the kernel's reads
    are 64-byte gathers, not the wide coalesced streams whose FETCH_SIZE the guide says to double.  Returns (bytes, source) or raises."""
    import csv
    import glob
    import shutil
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        raise RuntimeError("rocprofv3 not found")
    total, parts = 0, {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="plk_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
                env.pop(k, None)
            cmd = [prof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
                   "--msm-only", "--pipeline-depth", "1", "--steps", "6", "--warmup", "2", "--log-n", str(log_n), "--sustained-settle-steps", "0"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                raise RuntimeError("rocprofv3 --pmc %s failed (rc %d): %s" % (counter, r.returncode, (r.stderr or r.stdout)[-200:]))
            vals = sorted(float(row["Counter_Value"]) for row in csv.DictReader(open(files[0]))
                          if "msm_accumulate" in row["Kernel_Name"] and row["Counter_Name"] == counter)
            if len(vals) < 4:
                raise RuntimeError("rocprofv3 --pmc %s: %d msm_accumulate launches in the trace" % (counter, len(vals)))
            parts[counter] = vals[len(vals) // 2]
            total += vals[len(vals) // 2] * 1024.0
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return int(total), ("measured in this run: median per msm_accumulate launch of two rocprofv3 --pmc child runs of `bench.py --msm-only --pipeline-depth 1` "
                        "(FETCH_SIZE %.0f KB + WRITE_SIZE %.0f KB; separate passes, no correction factor: 64-byte gathers)" % (parts["FETCH_SIZE"], parts["WRITE_SIZE"]))


# ------------------------------------------------------------------------------------------ multi-GPU legs
def sharded_prove_scatter(ctx, dist, device, log_n, rank, world, proofs=3):
    """the same sharded prove in OWNER-COMPUTES mode (PLK_SHARD_SCATTER, round 5): rank 0 alone runs the prover and sends every other
    rank its slice of each commitment's scalars (grouped ncclSend / ncclRecv, N/G x 32 B per vector and link); the others hold their
    slice of the key and sit in plk_comm_serve — their VALU is free of the replicated transforms.  Timed on rank 0."""
    import plonkit_amd as pa
    n = 1 << log_n
    local = n // world
    ctx.srs_generate(local, rank * local, 42)
    ctx.comm_set_shard(rank * local)
    # the point-to-point transport first, on every rank (header broadcast + one grouped ring step, 45 s deadline): a node where it does not
    # work costs this leg, not a hang
    err0 = None
    if not os.environ.get("PLK_BENCH_SHARE_DEVICE"):              # (the single-GPU test tier runs over the TCP transport: nothing to probe)
        keep = os.environ.get("PLK_COMM_TIMEOUT_MS")
        os.environ["PLK_COMM_TIMEOUT_MS"] = keep or "45000"
        try:
            ctx.comm_selftest()
        except Exception as exc:                                   # noqa: BLE001
            err0 = exc
        finally:
            if keep is None:
                del os.environ["PLK_COMM_TIMEOUT_MS"]
    if not all_ok(dist, device, err0 is None):
        raise err0 or RuntimeError("plk_comm_selftest failed on another rank")
    ctx.comm_set_mode("scatter")
    # a batch job: an owner that fails without reaching comm_stop_workers must not leave its workers waiting for ever
    os.environ.setdefault("PLK_COMM_IDLE_TIMEOUT_MS", "90000")
    res, err = None, None
    try:
        if rank == 0:
            circ = pa.Circuit.synthetic(n - 2)
            setup = pa.SetupForProver(ctx, circ)
            proof = setup.prove(circ)
            times = []
            for _ in range(proofs):
                t0 = time.perf_counter()
                p = setup.prove(circ)
                times.append(time.perf_counter() - t0)
                assert p == proof
            vk = setup.verification_key_bytes(pa.crs42_g2_bytes())
            verified = bool(pa.verify(vk, proof))
            setup.close(); circ.close()
            res = {"wall_s": round(sorted(times)[len(times) // 2], 4), "wall_s_min": round(min(times), 4), "domain": n, "n_gpus": world, "srs_points_per_gpu": local,
                   "verified": verified, "proof_sha256": __import__("hashlib").sha256(proof).hexdigest()[:16],
                   "what": "owner-computes mode: rank 0 proves (transforms, quotient, openings once), ranks 1..G-1 commit their slice of every "
                           "vector (ncclSend/ncclRecv of N/G scalars per vector, 96 B back) — same proof bytes as one GPU"}
        else:
            ctx.comm_serve()
    except Exception as exc:                                       # noqa: BLE001 — (a worker whose owner died returns from comm_serve with an error)
        err = exc
    if rank == 0:                                                   # whatever happened to the owner: its workers wait without a deadline
        try:
            ctx.comm_stop_workers()
        except Exception as exc:                                   # noqa: BLE001
            err = err or exc
    ctx.comm_set_mode("replicate")
    good = all_ok(dist, device, err is None)
    if err is not None:
        raise err
    if not good:
        raise RuntimeError("another rank failed in this leg")
    return res


def sharded_prove(ctx, dist, device, log_n, rank, world):
    """whole prove with every commitment sharded over the ranks (plonkit_amd.sharded.ShardedProver)"""
    import plonkit_amd as pa
    n = 1 << log_n
    local = n // world
    ctx.srs_generate(local, rank * local, 42)                     # this rank's slice of the 2^log_n key
    ctx.comm_set_shard(rank * local)                              # built-in combiner (RCCL all-gather + EC sum in the library)
    circ = pa.Circuit.synthetic(n - 2)
    setup = pa.SetupForProver(ctx, circ)
    proof = setup.prove(circ)                                      # warm-up: tables, allocations, cached extensions
    best = None
    for _ in range(2):
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        p = setup.prove(circ)
        dt = time.perf_counter() - t0
        assert p == proof
        t = torch.tensor([dt], dtype=torch.float64, device=red_device(device))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        best = float(t.item()) if best is None else min(best, float(t.item()))
    # every rank must hold the same bytes, and the host verifier (real pairing) must accept them against the sharded key
    import hashlib
    digests = [None] * world
    dist.all_gather_object(digests, hashlib.sha256(proof).hexdigest())
    vk = setup.verification_key_bytes(pa.crs42_g2_bytes())        # 11 sharded commitments: every rank takes part
    verified = bool(pa.verify(vk, proof)) if rank == 0 else None
    setup.close(); circ.close()
    return {"wall_s": round(best, 4), "domain": n, "n_gpus": world, "srs_points_per_gpu": local, "proof_bytes": len(proof),
            "same_proof_on_every_rank": len(set(digests)) == 1, "verified": verified,
            "what": "SetupForProver::prove with every commitment computed as the sum over ranks of MSM(slice of the scalars, "
                    "slice of the SRS): ncclAllGather of the Jacobian partial sums inside the library (plk_comm_init) + host EC sum; "
                    "NTTs and point-wise work replicated"}


def replica_prove_throughput(dist, device, log_n, rank, world, in_flight=2, proofs_each=6):
    """prove THROUGHPUT of the node: every GPU proves on its own — a full key per GPU, no exchange ("replicas only": proofs are
    independent jobs), `in_flight` proofs in flight per GPU (plonkit_amd.prover_bench.throughput).  north_star asks
    for "prove-time throughput ... at 1/2/4/8 GPUs"; sharding ONE proof's commitments (sharded_prove below) lowers its latency by at
    most ~2.3x at the 2^20 domain, N independent provers give N x the proofs per second.  Timed as the contract says: barrier +
    synchronize, every rank proves its share, barrier + synchronize, MAX over ranks; value = all proofs / that time."""
    import plonkit_amd as pa
    from plonkit_amd import prover_bench
    # Every rank executes every collective of this leg whatever happens to it locally (a rank that raised before a barrier would leave
    # the others waiting for the process-group timeout): local work runs under try, the verdicts are AND-ed across the ranks.
    ctx2 = setup = None
    circs = []
    err = None
    try:
        ctx2 = pa.Context(device.index if device.index is not None else 0)      # a context of its own: no communicator installed
        ctx2.srs_generate(1 << log_n, 0, 42)
        n_gates = (1 << log_n) - 2
        circs = [pa.Circuit.synthetic_ex(n_gates, witness_seed=(0 if k == 0 else 1000 * (rank + 1) + k)) for k in range(in_flight)]
        setup = pa.SetupForProver(ctx2, circs[0])
        prover_bench.throughput(ctx2, log_n, in_flight=in_flight, proofs_each=1, setup=setup, circs=circs)  # warm-up of every context path
    except Exception as exc:                                       # noqa: BLE001
        err = exc
    r = {"proofs_per_s": 0.0}
    ready = all_ok(dist, device, err is None)
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    if ready:
        try:
            r = prover_bench.throughput(ctx2, log_n, in_flight=in_flight, proofs_each=proofs_each, setup=setup, circs=circs)
        except Exception as exc:                                   # noqa: BLE001
            err = exc
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=red_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    per_gpu = torch.tensor([r["proofs_per_s"]], dtype=torch.float64, device=red_device(device))
    dist.all_reduce(per_gpu, op=dist.ReduceOp.SUM)
    good = all_ok(dist, device, err is None)
    if setup:
        setup.close()
    for c in circs:
        c.close()
    if ctx2:
        ctx2.close()
    if err is not None:
        raise err
    if not good:
        raise RuntimeError("another rank failed in this leg (its own line names the error)")
    return {"n_gpus": world, "in_flight_per_gpu": in_flight, "proofs_per_gpu": in_flight * proofs_each, "domain": 1 << log_n,
            "proofs_per_s": round(float(per_gpu.item()), 2), "ms_per_proof_node": round(1e3 / float(per_gpu.item()), 3),
            "region_wall_s": round(float(t.item()), 3), "scaling": "replicas only (independent proofs, one full key per GPU, no exchange)",
            "what": "sum over ranks of the proofs per second of each GPU's concurrent phase (%d proofs in flight per GPU, every proof byte-identical "
                    "to the one made alone); region_wall_s also contains the sequential reference proofs each rank makes first" % in_flight}


def strong_scaling_msm(ctx, dist, device, rank, world, log_total=24, reps=5):
    """BASELINE.json configs[2] / north_star ">= 6x MSM scaling 1 -> 8 GPUs": ONE commitment of 2^24 terms, the SRS
    split 2^24 / world per rank (total work fixed: strong scaling).  Rank 0 also times the whole 2^24-term commitment
    on its own GPU in the same run, so `scaling_vs_1gpu` is a same-run ratio, not a number carried over from a file."""
    from plonkit_amd.sharded import ShardedMsm
    total = 1 << log_total
    local = total // world
    scal = rand_scalars(local, 0x24 + rank, device)
    torch.cuda.synchronize()
    ctx.srs_generate(local, rank * local, 42)
    msm = ShardedMsm(ctx, dist, device if NATIVE_EXCHANGE[0] else red_device(device), native=NATIVE_EXCHANGE[0])   # the communicator main() created (or torch.distributed: see main)
    stream = torch.cuda.Stream(device=device)
    for _ in msm.commit_stream((scal for _ in range(2)), local, stream=stream):
        pass
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for out in msm.commit_stream((scal for _ in range(reps)), local, stream=stream):
        pass
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=red_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    sharded_ms = float(t.item()) / reps * 1e3
    # latency of a single sharded commitment (nothing else in flight)
    torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    msm.commit(scal, local, stream=stream)
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=red_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    single_ms = float(t.item()) * 1e3
    one_gpu_ms = None
    if rank == 0:                                                   # the same 2^24 terms on one GPU (16 GiB fixed-base table)
        full = rand_scalars(total, 0x24, device)
        ctx.srs_generate(total, 0, 42)
        solo = ShardedMsm(ctx, None, device)
        for _ in solo.commit_stream((full for _ in range(2)), total, stream=stream):
            pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in solo.commit_stream((full for _ in range(3)), total, stream=stream):
            pass
        torch.cuda.synchronize()
        one_gpu_ms = (time.perf_counter() - t0) / 3 * 1e3
        del full
    dist.barrier()
    res = {"terms_total": total, "terms_per_gpu": local, "n_gpus": world, "scaling": "strong",
           "ms_per_commitment": round(sharded_ms, 3), "ms_single_commitment_latency": round(single_ms, 3),
           "Mscalar_mul_s": round(total / sharded_ms / 1e3, 1)}
    if one_gpu_ms:
        res["one_gpu_ms_per_commitment"] = round(one_gpu_ms, 3)
        res["scaling_vs_1gpu"] = round(one_gpu_ms / sharded_ms, 2)
    return res


def time_commitments(ctx, msm, scalars, n, steps, stream, depth):
    """K commitments back to back.  depth 2 / 3: that many in flight (ShardedMsm.commit_stream); depth 1: one at a time."""
    kernel_ms = []
    out = None
    t0 = time.perf_counter()
    if depth >= 2:
        for out in msm.commit_stream((scalars for _ in range(steps)), n, stream=stream, depth=depth):
            kernel_ms.append(ctx.msm_last_kernel_ms())
    else:
        for _ in range(steps):
            out = msm.commit(scalars, n, stream=stream)
            kernel_ms.append(ctx.msm_last_kernel_ms())
    torch.cuda.synchronize()
    return time.perf_counter() - t0, kernel_ms, out


def spawn_ranks(n_gpus):
    """re-run this command line as N ranks: python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>; returns the launcher's exit status"""
    import socket
    port = os.environ.get("MASTER_PORT")
    if not port:
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
        sk.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    env.pop("MASTER_PORT", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-n", type=int, default=20, help="log2 of the per-GPU commitment size")
    ap.add_argument("--cpu-log-n", type=int, default=20, help="log2 of the CPU-baseline samples (MSM terms, prove domain)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--strong-log-n", type=int, default=24, help="N > 1: log2 of the ONE commitment whose SRS is split over the ranks (configs[2]: 24)")
    ap.add_argument("--pipeline-depth", type=int, default=3, choices=(1, 2, 3),
                    help="commitments in flight in the timed region (3 = the library's three-slot FIFO; 1 = one at a time: "
                         "the region roofline.kernel_ms is taken from)")
    ap.add_argument("--settle-steps", type=int, default=0,
                    help="extra untimed commitments BEFORE the W warm-up steps of the headline region (default 0: the headline is "
                         "W warm-up + K timed steps and nothing else; round 2 defaulted to 50)")
    ap.add_argument("--sustained-settle-steps", type=int, default=50,
                    help="untimed commitments between the headline region and the `sustained` region (the same K steps timed again "
                         "once the GPU has been under load for >= 70 ms: DESIGN.md §5); 0 = no sustained region")
    ap.add_argument("--no-live-traffic", action="store_true", help="skip the two rocprofv3 --pmc child runs that measure roofline.traffic (the constant from profiles/ is reported instead)")
    ap.add_argument("--msm-only", action="store_true", help="only the timed commitments (no cpu_baseline / prove / kernels legs): "
                                                            "the command the rocprofv3 summary under profiles/ is taken from")
    args = ap.parse_args()
    WRITE_FULL_RECORD[0] = not args.msm_only

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: spawn the N ranks ourselves, exactly as the driver's documented command does
        # (one process per GPU under torch.distributed.run, rendezvous on 127.0.0.1); their rank 0 prints the JSON line
        raise SystemExit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N ranks for --gpus N, or run it plainly and let it spawn them)"
                         % (args.gpus, world))
    # PLK_BENCH_SHARE_DEVICE=1 (test tier with ONE GPU): every rank uses device 0, torch.distributed runs over gloo and the
    # partial sums travel over the library's TCP transport, because RCCL refuses two ranks on one device.  It exercises the
    # N > 1 control flow (slicing, strong-scaling leg, sharded prove); its timings mean nothing.
    share = bool(os.environ.get("PLK_BENCH_SHARE_DEVICE"))
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    force_dist = bool(os.environ.get("PLK_FORCE_GATHER")) and "RANK" in os.environ     # one-rank test of the RCCL path
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # a rank that fails must not leave the others waiting for the default 10 minutes in a collective
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, timeout=datetime.timedelta(seconds=180))

    import plonkit_amd as pa
    from plonkit_amd.sharded import ShardedMsm

    ctx = pa.Context(local_rank)
    n = 1 << args.log_n
    ctx.srs_generate(n, start=rank * n, tau=42)          # this rank's shard of the N*2^20 monomial SRS
    scalars = rand_scalars(n, 0x706c6f6e6b6974 + rank, device)
    torch.cuda.synchronize()                             # the scalars are consumed on another stream
    stream = torch.cuda.Stream(device=device)
    multi = world > 1 or force_dist
    native_exchange = True
    if multi:
        # the exchange of the partial sums runs inside the library (comm.cpp: ncclAllGather on the communicator's own stream + host
        # EC sum); torch.distributed only carries the 128-byte RCCL id to the other ranks and the timing barriers
        # Every rank runs every collective of this block whatever happens to it locally; if ANY rank cannot open the library's
        # communicator (a librccl the loader cannot find, an RCCL that refuses the id), ALL ranks fall back to exchanging the partial
        # sums through torch.distributed — the headline needs an all-gather of 96 bytes per rank, not a particular carrier of it.
        err, box = None, [None]
        try:
            if os.environ.get("PLK_BENCH_TEST_NO_COMM"):           # TEST HOOK (tests/test_gpu_sharded_prove.py): the fallback must work
                raise RuntimeError("PLK_BENCH_TEST_NO_COMM")
            if rank == 0 and share:
                # rank 0 picks a free port for the library's TCP hub and tells the others (MASTER_PORT + k may be taken or exceed 65535)
                import socket
                sk = socket.socket()
                sk.bind(("127.0.0.1", 0))
                box = [sk.getsockname()[1]]
                sk.close()
            elif rank == 0:
                box = [pa.comm_unique_id()]
        except Exception as exc:                                   # noqa: BLE001
            err = exc
        dist.broadcast_object_list(box, src=0)
        try:
            if box[0] is None:
                raise err or RuntimeError("rank 0 could not create the communicator's id")
            if share:
                ctx.comm_init_tcp(rank, world, int(box[0]), rank * n)
            else:
                ctx.comm_init(rank, world, box[0], rank * n)
            # one small commitment through the exchange before anything is timed, under a short deadline: a communicator that
            # opens but whose first all-gather never completes is found here (45 s), not in the warm-up steps (180 s, fatal)
            keep = os.environ.get("PLK_COMM_TIMEOUT_MS")
            os.environ["PLK_COMM_TIMEOUT_MS"] = keep or "45000"
            try:
                ShardedMsm(ctx, None, device, native=True).commit(scalars, min(n, 4096))
            finally:
                if keep is None:
                    del os.environ["PLK_COMM_TIMEOUT_MS"]
        except Exception as exc:                                   # noqa: BLE001
            err = exc
        if not all_ok(dist, device, err is None):
            try:
                ctx.comm_destroy()
            except Exception:                                      # noqa: BLE001
                pass
            native_exchange = NATIVE_EXCHANGE[0] = False
            FAILED_LEGS.append(("comm_init", repr(err) if err else "another rank could not open the library's communicator", False))
    msm = ShardedMsm(ctx, dist if multi else None, device if native_exchange else red_device(device), native=multi and native_exchange)
    ctx.set_kernel_timing(True)

    # W warm-up steps, then EXACTLY K timed steps bracketed by barrier + synchronize; the exchange of commitment k
    # overlaps the kernels of k+1 (ShardedMsm.commit_stream)
    if args.settle_steps:                                 # sustained load first (see --settle-steps); reported in the line
        time_commitments(ctx, msm, scalars, n, args.settle_steps, stream, args.pipeline_depth)
    time_commitments(ctx, msm, scalars, n, args.warmup, stream, args.pipeline_depth)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    _, kernel_ms, out = time_commitments(ctx, msm, scalars, n, args.steps, stream, args.pipeline_depth)
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_device(device))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # the same K steps once more after sustained load (`sustained`: what round 2's default --settle-steps 50 measured)
    sustained = None
    if args.sustained_settle_steps:
        time_commitments(ctx, msm, scalars, n, args.sustained_settle_steps, stream, args.pipeline_depth)
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t_s = time.perf_counter()
        time_commitments(ctx, msm, scalars, n, args.steps, stream, args.pipeline_depth)
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        sustained = time.perf_counter() - t_s
        if dist:
            t = torch.tensor([sustained], dtype=torch.float64, device=red_device(device))
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            sustained = float(t.item())
    # the dominant kernel alone: one commitment in flight, so nothing shares the GPU with msm_accumulate
    solo_elapsed, solo_kernel_ms, _ = time_commitments(ctx, msm, scalars, n, max(5, args.steps // 2), stream, 1)
    if dist:
        dist.barrier()

    line = None
    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = world * n / (elapsed / args.steps) / 1e6
        k_pipe = float(np.mean(kernel_ms))
        k_solo = float(np.mean(solo_kernel_ms))
        k_ms = k_solo if args.pipeline_depth >= 2 else k_pipe
        achieved = ALGO_BYTES_PER_TERM * n / (k_ms * 1e-3) / 1e9
        gmadd = n * MSM_WINDOWS / (k_ms * 1e-3) / 1e9
        line = {
            "metric": "G1 MSM throughput at 2^%d terms per GPU (KZG commitment of the PLONK prover)" % args.log_n,
            "value": round(value, 3), "unit": "Mscalar·mul/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u256 (BN254 Fr/Fq Montgomery; 9x29-bit limbs in registers, v_mad_u64_u32)", "data": "synthetic",
            "config": {"workload": "Pippenger G1 MSM, 2^%d uniform scalars per GPU, tau=42 monomial SRS sharded by rank "
                                   "(BASELINE.json configs[1]: SRS 2^20, single MI355X at N=1)" % args.log_n,
                       "terms_per_gpu": n, "parallelism": "srs-shard x%d + all_gather of partial sums" % world,
                       "exchange": ("none (one GPU)" if not multi else "ncclAllGather inside the library (plk_comm_init)" if native_exchange
                                    else "torch.distributed all_gather (fallback: the library's communicator could not be opened, see failed_legs)"),
                       "pipeline_depth": args.pipeline_depth, "settle_steps": args.settle_steps,
                       # ranks RCCL itself counts in the library's communicator (ncclCommCount): N for a real N-GPU run, 0 when the partial sums
                       # travel another way (fallback / the shared-device test tier) — DESIGN.md section 6 says what a SCALE line must show
                       "comm_ranks": (ctx.comm_nccl_count() if multi else 1),
                       "result_x_be": pa.g1_to_bytes(out).hex()[:64]},
            # `bound`: what limits the kernel is VALU issue (v_mad_u64_u32), not HBM and not MFMA (integer modular arithmetic) — said so here;
            # achieved / peak / frac stay the HBM figures north_star and the bench contract ask for (hbm_frac repeats frac under its own name),
            # the VALU figures are in `valu`
            "roofline": {"bound": "valu", "kernel": "msm_accumulate", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "hbm_frac": round(achieved / HBM_PEAK_GBS, 5),
                         "valu_frac": round(gmadd / VALU_PEAK_GMADD, 3),
                         "traffic": PMC_TRAFFIC_BYTES_2POW20 if (args.log_n == 20) else None,
                         "traffic_source": PMC_TRAFFIC_SOURCE if (args.log_n == 20) else None,
                         "kernel_ms": round(k_ms, 4), "kernel_ms_pipelined": round(k_pipe, 4),
                         "ms_per_step_one_in_flight": round(solo_elapsed * 1e3 / len(solo_kernel_ms), 4),
                         "algorithmic_bytes": ALGO_BYTES_PER_TERM * n,
                         "valu": {"achieved_gmadd_s": round(gmadd, 2), "peak_gmadd_s": round(VALU_PEAK_GMADD, 2),
                                  "frac": round(gmadd / VALU_PEAK_GMADD, 3),
                                  "loop_isolated_gmadd_s": LOOP_ISOLATED_GMADD,
                                  "frac_of_isolated_loop": round(gmadd / LOOP_ISOLATED_GMADD, 3),
                                  "derivation": "peak = measured v_mad_u64_u32 issue rate (576.1 G wave-instr/s x 64 lanes, "
                                                "profiles/r01_ubench_int.txt) / 1467 mads per XYZZ mixed addition; "
                                                "loop_isolated = the same loop with its 625 non-mad instructions, alone on the chip (profiles/r03_ubench_mulw_forms.txt)"},
                         "note": "kernel_ms = HIP-event duration of msm_accumulate with one commitment in flight (the region timed right "
                                 "after the headline one; rocprofv3 of `bench.py --msm-only --pipeline-depth 1` agrees, profiles/); "
                                 "kernel_ms_pipelined = the same kernel inside the headline region, where up to three commitments are in "
                                 "flight and it shares the GPU with the bucket reduction of the previous one and the partition of the next "
                                 "(its duration there can exceed the step time: kernels of consecutive commitments overlap).  The kernel is bound by v_mad_u64_u32 issue, "
                                 "not HBM (SURVEY.md §8d); `traffic` is 11x the algorithmic bytes because Pippenger gathers one "
                                 "64-byte point per (term, window): 15 windows, each from its own shifted copy of the SRS "
                                 "(0.94 GiB fixed-base table in HBM)"},
        }
        if sustained is not None:
            # (top-level copy of sustained.value: round 2's `value` was measured after 50 settle commitments, i.e. it corresponds to THIS figure)
            line["value_sustained"] = round(world * n / (sustained / args.steps) / 1e6, 3)
            line["sustained"] = {"value": round(world * n / (sustained / args.steps) / 1e6, 3), "ms_per_step": round(sustained * 1e3 / args.steps, 4),
                                 "steps": args.steps, "untimed_steps_before": args.settle_steps + args.warmup + args.steps + args.sustained_settle_steps,
                                 "what": "the headline region repeated after %d further untimed commitments (GPU under load for >= 70 ms)" % args.sustained_settle_steps}
        if world > 1 and not args.no_cpu_baseline and not args.msm_only:
            # the N > 1 line is self-contained: the same dense_multiexp port on this box's host cores (rank 0's GPU shard supplies the bases)
            cb, ref, s_host = cpu_msm_baseline(ctx, min(args.cpu_log_n, args.log_n), 1234)
            cb["note"] = "MSM leg only at N > 1; the N = 1 line carries the NTT / prove / G1-iNTT rows"
            line["cpu_baseline"] = cb
        if world == 1 and not args.no_cpu_baseline and not args.msm_only:
            cb, ref, s_host = cpu_msm_baseline(ctx, min(args.cpu_log_n, args.log_n), 1234)
            got = ctx.msm(s_host)                         # same sample through the HIP path
            cb["matches_gpu"] = bool(np.array_equal(got, ref))
            for key, fn in (("msm_rows", lambda: cpu_msm_rows(ctx, device)), ("g1_intt", lambda: cpu_g1_intt_row(ctx, device)),
                            ("ntt", cpu_ntt_baseline), ("prove", lambda: cpu_prove_baseline(ctx, min(args.cpu_log_n, args.log_n)))):
                cb[key] = leg("cpu_baseline." + key, fn)               # a secondary row must not cost the line
            for key in ("msm_rows",):                                  # a result that is not the same group element is a failure, not a field
                if isinstance(cb[key], dict) and any(isinstance(r, dict) and r.get("same_point") is False for r in cb[key].values()):
                    FAILED_LEGS.append(("cpu_baseline." + key, "a GPU commitment differs from its reference value", True))
            if isinstance(cb.get("prove"), dict) and cb["prove"].get("proof_bytes_identical") is False:
                FAILED_LEGS.append(("cpu_baseline.prove", "GPU proof differs from the CPU port's", True))
            rb = reference_binary_baseline(min(args.cpu_log_n, args.log_n))
            if rb:
                cb["reference_binary"] = rb
            line["cpu_baseline"] = cb
        if world == 1 and not force_dist and not args.msm_only and not args.no_live_traffic:
            try:
                tb, src = measure_traffic_live(args.log_n)
                line["roofline"]["traffic"], line["roofline"]["traffic_source"] = tb, src
                line["roofline"]["traffic_over_algorithmic"] = round(tb / (ALGO_BYTES_PER_TERM * n), 2)
            except Exception as exc:                               # noqa: BLE001 — the constant stays, and says why
                FAILED_LEGS.append(("roofline.traffic", repr(exc), False))
        if world == 1 and not force_dist and not args.msm_only:
            from plonkit_amd import prover_bench
            line["prove"] = leg("prove", lambda: prover_bench.run(ctx, args.log_n))      # a failing leg must not cost the headline line
            # the same circuit shape with a live d column (11 of 11 commitments), prove THROUGHPUT (two / three proofs in flight on this GPU),
            # and the whole `plonkit prove` process (SURVEY.md 8(d)'s third timed region)
            for key, fn in (("dense", lambda: prover_bench.run_dense(ctx, args.log_n)),
                            ("throughput", lambda: prover_bench.throughput(ctx, args.log_n, in_flight=2, proofs_each=10)),
                            ("throughput_in_flight_3", lambda: prover_bench.throughput(ctx, args.log_n, in_flight=3, proofs_each=8)),
                            ("throughput_dense", lambda: prover_bench.throughput(ctx, args.log_n, in_flight=2, proofs_each=6, lc_terms=7)),
                            ("by_domain", lambda: prove_by_domain(ctx, line["prove"], not args.no_cpu_baseline))):
                line["prove"][key] = leg("prove." + key, fn)
            bd = line["prove"].get("by_domain")
            if isinstance(bd, dict) and any(isinstance(r, dict) and (r.get("proof_bytes_identical") is False or r.get("verified") is False) for r in bd.values()):
                FAILED_LEGS.append(("prove.by_domain", "a proof differs from the CPU port's or is rejected by the verifier", True))
            # the reference CI's command sequence as whole processes: at the 2^20 domain, and the CI's own shape (2^12 circuit, 2^20 key)
            line["cli"] = leg("cli", lambda: prover_bench.cli_table(args.log_n))
            line["cli_ci_shape"] = leg("cli_ci_shape", cli_ci_shape)
            for key in ("cli", "cli_ci_shape"):
                if isinstance(line[key], dict) and (line[key].get("verified") is False or line[key].get("prove_l_same_bytes") is False):
                    FAILED_LEGS.append((key, "verify rejected the proof, or prove -l gave other bytes than prove", True))
            if isinstance(line["prove"].get("dense"), dict) and line["prove"]["dense"].get("verified") is False:
                FAILED_LEGS.append(("prove.dense", "the host verifier rejects the dense proof", True))
            line["kernels"] = leg("kernels", lambda: prover_bench.kernel_table(ctx, device))
    watchdog = None
    if world > 1 or force_dist:
        watchdog = start_leg_watchdog(rank, [line])
        # (a) strong scaling of ONE 2^24-term commitment (configs[2]); (b) multi-GPU prove at the 2^log_n domain: the SRS
        # sliced across the ranks, commitments combined over RCCL, NTTs replicated.  Every rank takes part;
        # a failure here must not cost the headline line.
        # (rank 0 files every leg as soon as it returns: the watchdog prints what is there)
        strong = leg("strong", lambda: strong_scaling_msm(ctx, dist, device, rank, world, log_total=args.strong_log_n))
        if rank == 0:
            line["strong"] = strong
            # the figure north_star's ">= 6x MSM scaling 1 -> 8 GPUs" reads, where a reader will look for it: `value` above is
            # WEAK scaling (2^log_n terms per GPU), these two are STRONG scaling of one fixed 2^strong_log_n-term commitment
            line["strong_value"] = strong.get("Mscalar_mul_s")
            line["strong_unit"] = "Mscalar·mul/s (one 2^%d-term commitment, SRS split over %d GPUs)" % (args.strong_log_n, world)
            line["strong_scaling_vs_1gpu"] = strong.get("scaling_vs_1gpu")
        sharded = leg("prove(sharded)", lambda: sharded_prove(ctx, dist, device, args.log_n, rank, world))
        if rank == 0:
            line["prove"] = sharded
        scattered = leg("prove(scatter)", lambda: sharded_prove_scatter(ctx, dist, device, args.log_n, rank, world))
        if rank == 0 and isinstance(line["prove"], dict):
            line["prove"]["scatter"] = scattered
        replicas = leg("prove_throughput", lambda: replica_prove_throughput(dist, device, args.log_n, rank, world))
        if rank == 0:
            line["prove_throughput"] = replicas
    if watchdog:
        watchdog.cancel()
    if rank == 0:
        line["failed_legs"] = [{"leg": a, "error": b, "correctness": c} for a, b, c in FAILED_LEGS]
        emit(line)
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    if any(c for _, _, c in FAILED_LEGS):
        raise SystemExit("bench.py: a correctness check failed in: " + ", ".join(a for a, _, c in FAILED_LEGS if c))


if __name__ == "__main__":
    main()
