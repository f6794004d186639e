#!/usr/bin/env python3
"""bench.py — headline benchmark of the PLONK prove hot path on MI355X (BASELINE.json metric:
"PLONK prove wall-clock (s) + G1 MSM throughput (Mscalar·mul/s) at 2^20 domain, 1/2/4/8 GPU").

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one KZG commitment (G1 MSM) of 2^20 uniform scalars per GPU against that GPU's resident
shard of a tau = 42 monomial SRS (BASELINE.json configs[1]; with N GPUs the job is one commitment
of N*2^20 terms with the bases split across ranks, partial sums exchanged over RCCL — "weak").
`value` = total scalar·muls per second over all ranks, inputs resident in HBM.
The one JSON line also carries `roofline` (dominant kernel: msm_accumulate, HIP-event timed),
`cpu_baseline` (oracle restatement of bellman's dense_multiexp on the host cores, bounded sample)
and, at N=1, `prove` (wall-clock of a full prove at the 2^20 domain once the prover is built in).
"""
import argparse
import json
import os
import sys
import time

# the library keeps two commitments in flight on two streams; with HIP's default of 4 hardware queues per device
# those streams can land on one queue and serialise (measured: 2.04 ms per commitment instead of 1.66 ms)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
ALGO_BYTES_PER_TERM = 96         # 64 B base + 32 B scalar
# PMC traffic of msm_accumulate for ONE 2^20-term launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
# separate passes, profiles/r01_pmc_traffic_v3.txt, median of the single-commitment launches):
# 1,034,100 KB fetched (15.7 M gathers of one 64-byte point each from the 0.94 GiB fixed-base table
# = 1.007 GB, + 63 MB of sorted entries: the counter is consistent with 64 B per gather, i.e. no
# over-fetch) + 63,146 KB written (lane partial sums).  Only valid for --log-n 20, N=1.
PMC_TRAFFIC_BYTES_2POW20 = (1034100 + 63146) * 1024
MSM_WINDOWS = 15                # 17-bit signed windows over the 254-bit scalars: mixed additions per term
VALU_PEAK_GMADD = 15.6           # tools/ubench_w: isolated mixed-addition loop, G additions/s (profiles/r01_ubench_w.txt)


def rand_scalars(n, seed, device):
    """uniform 252-bit residues (valid Montgomery Fr representatives), generated on the GPU"""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    t = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device=device, generator=g)
    t[:, 3] &= (1 << 60) - 1
    return t


def cpu_baseline(ctx, log_sample, seed):
    """bellman dense_multiexp restatement (oracle/, kind "port") on the host cores, bounded sample"""
    from oracle import oracle_lib as ol          # checker / baseline only — never on the product path
    m = 1 << log_sample
    bases = ctx.srs_download(0, m)
    rng = np.random.default_rng(seed)
    s = rng.integers(0, 1 << 62, size=(m, 4), dtype=np.uint64)
    s[:, 3] &= np.uint64((1 << 60) - 1)
    # the restatement keeps bellman's per-thread bucket arrays, whose reduction grows with the thread count: on the
    # 256-core GPU host 16 threads is the fastest setting (profiles/r01_cpu_msm_threads.txt: 1.49 M/s at 16, 0.06 at 256)
    cores = min(os.cpu_count() or 1, 16)
    ol.msm(bases[:1024], s[:1024], threads=cores)            # warm the library
    t0 = time.perf_counter()
    ref = ol.msm(bases, s, threads=cores)
    dt = time.perf_counter() - t0
    return {"value": m / dt / 1e6, "unit": "Mscalar·mul/s", "cores": cores, "kind": "port",
            "sample": "one dense_multiexp (c=ceil(ln n), per-thread buckets) of 2^%d uniform scalars, %.2f s" % (log_sample, dt)}, ref, s


def cpu_prove_baseline(ctx, log_domain):
    """the oracle's restatement of the whole prove (numpy glue + OpenMP C kernels, kind "port") on a smaller
    domain than the headline one, beside the HIP prover on the SAME circuit, witness and SRS; the two proofs
    must be byte-identical.  Bounded sample: 2^16 gates is ~5-10 s of CPU work."""
    import plonkit_amd as pa
    from oracle import oracle_lib as ol, plonk_oracle as po      # checker / baseline only
    circ = pa.Circuit.synthetic((1 << log_domain) - 2)
    r1cs, wit = po.load_r1cs_bin(circ.export("r1cs")), po.parse_wtns(circ.export("wtns"))
    srs_keep = ctx.srs_size()
    ctx.srs_generate(1 << log_domain, 0, 42)
    crs = po.Crs(ctx.srs_download(0, 1 << log_domain), b"\x01" * 256)
    S = po.setup(r1cs)
    t0 = time.perf_counter()
    ref = po.write_proof(po.prove(r1cs, wit, crs, S))
    cpu_s = time.perf_counter() - t0
    setup = pa.SetupForProver(ctx, circ)
    setup.prove(circ)
    t0 = time.perf_counter()
    got = setup.prove(circ)
    gpu_s = time.perf_counter() - t0
    setup.close(); circ.close()
    ctx.srs_generate(srs_keep, 0, 42)
    return {"domain": 1 << log_domain, "cpu_s": round(cpu_s, 3), "gpu_s": round(gpu_s, 5), "threads": ol.ncpu(),
            "kind": "port", "proof_bytes_identical": bool(got == ref),
            "sample": "one prove (rounds 1-5, 11 MSM + 25 NTT-equivalents) of a synthetic 2^%d-gate circuit" % log_domain}


def sharded_prove(ctx, dist, device, log_n, rank, world):
    """whole prove with every commitment sharded over the ranks (plonkit_amd.sharded.ShardedProver)"""
    import plonkit_amd as pa
    from plonkit_amd.sharded import ShardedProver
    n = 1 << log_n
    local = n // world
    ctx.srs_generate(local, rank * local, 42)                     # this rank's slice of the 2^log_n key
    sp = ShardedProver(ctx, dist, device)
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
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        best = float(t.item()) if best is None else min(best, float(t.item()))
    sp.close()
    return {"wall_s": round(best, 4), "domain": n, "n_gpus": world, "srs_points_per_gpu": local, "proof_bytes": len(proof),
            "what": "SetupForProver::prove with every commitment computed as the sum over ranks of MSM(slice of the scalars, "
                    "slice of the SRS): all_gather of the Jacobian partial sums + host EC sum; NTTs and point-wise work replicated"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-n", type=int, default=20, help="log2 of the per-GPU commitment size")
    ap.add_argument("--cpu-log-n", type=int, default=20, help="log2 of the CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--msm-only", action="store_true", help="only the timed commitments (no cpu_baseline / prove / kernels legs): "
                                                            "the command the rocprofv3 summary under profiles/ is taken from")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    force_dist = bool(os.environ.get("PLK_FORCE_GATHER")) and "RANK" in os.environ     # one-rank test of the RCCL path
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # a rank that fails must not leave the others waiting for the default 10 minutes in a collective
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, timeout=datetime.timedelta(seconds=180))

    import plonkit_amd as pa
    from plonkit_amd.sharded import ShardedMsm

    ctx = pa.Context(local_rank)
    n = 1 << args.log_n
    ctx.srs_generate(n, start=rank * n, tau=42)          # this rank's shard of the N*2^20 monomial SRS
    scalars = rand_scalars(n, 0x706c6f6e6b6974 + rank, device)
    torch.cuda.synchronize()                             # the scalars are consumed on another stream
    stream = torch.cuda.Stream(device=device)
    msm = ShardedMsm(ctx, dist if (world > 1 or force_dist) else None, device)
    ctx.set_kernel_timing(True)

    # K commitments back to back; the exchange of commitment k overlaps the kernels of k+1 (ShardedMsm.commit_stream)
    for out in msm.commit_stream((scalars for _ in range(args.warmup)), n, stream=stream):
        pass
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    kernel_ms = []
    t0 = time.perf_counter()
    for out in msm.commit_stream((scalars for _ in range(args.steps)), n, stream=stream):
        kernel_ms.append(ctx.msm_last_kernel_ms())
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = world * n / (elapsed / args.steps) / 1e6
        k_ms = float(np.mean(kernel_ms))
        achieved = ALGO_BYTES_PER_TERM * n / (k_ms * 1e-3) / 1e9
        line = {
            "metric": "G1 MSM throughput at 2^%d terms per GPU (KZG commitment of the PLONK prover)" % args.log_n,
            "value": round(value, 3), "unit": "Mscalar·mul/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u256 (BN254 Fr/Fq Montgomery; 9x29-bit limbs in registers, v_mad_u64_u32)", "data": "synthetic",
            "config": {"workload": "Pippenger G1 MSM, 2^%d uniform scalars per GPU, tau=42 monomial SRS sharded by rank "
                                   "(BASELINE.json configs[1]: SRS 2^20, single MI355X at N=1)" % args.log_n,
                       "terms_per_gpu": n, "parallelism": "srs-shard x%d + all_gather of partial sums" % world,
                       "result_x_be": pa.g1_to_bytes(out).hex()[:64]},
            "roofline": {"bound": "hbm", "kernel": "msm_accumulate", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": PMC_TRAFFIC_BYTES_2POW20 if (args.log_n == 20) else None,
                         "kernel_ms": round(k_ms, 4), "algorithmic_bytes": ALGO_BYTES_PER_TERM * n,
                         "valu": {"achieved_gmadd_s": round(n * MSM_WINDOWS / (k_ms * 1e-3) / 1e9, 2),
                                  "peak_gmadd_s": VALU_PEAK_GMADD,
                                  "frac": round(n * MSM_WINDOWS / (k_ms * 1e-3) / 1e9 / VALU_PEAK_GMADD, 3)},
                         "note": "commitments are pipelined two deep, so this kernel runs beside the bucket reduction of the previous "
                                 "commitment and its HIP-event duration (kernel_ms) equals the step time; alone it takes about 1.3 ms "
                                 "(profiles/r01_bench_final_kernel_stats.csv).  "
                                 "The kernel is bound by v_mad_u64_u32 issue, not HBM (SURVEY.md §8d): `valu` compares its "
                                 "mixed-addition rate with the same loop measured in isolation (tools/ubench_w); `traffic` "
                                 "is 11x the algorithmic bytes because Pippenger gathers one 64-byte point per (term, window): "
                                 "15 windows, each from its own shifted copy of the SRS (0.94 GiB fixed-base table in HBM)"},
        }
        if world == 1 and not args.no_cpu_baseline and not args.msm_only:
            cb, ref, s_host = cpu_baseline(ctx, min(args.cpu_log_n, args.log_n), 1234)
            got = ctx.msm(s_host)                         # same sample through the HIP path
            cb["matches_gpu"] = bool(np.array_equal(got, ref))
            cb["prove"] = cpu_prove_baseline(ctx, min(16, args.log_n))
            line["cpu_baseline"] = cb
        if world == 1 and not force_dist and not args.msm_only:
            from plonkit_amd import prover_bench
            line["prove"] = prover_bench.run(ctx, args.log_n)
            line["kernels"] = prover_bench.kernel_table(ctx, device)
    if world > 1 or force_dist:
        # multi-GPU prove at the same 2^log_n domain: the SRS sliced across the ranks, commitments combined over RCCL,
        # NTTs replicated.  Every rank takes part; a failure here must not cost the headline line.
        try:
            sharded = sharded_prove(ctx, dist, device, args.log_n, rank, world)
        except Exception as exc:                                   # noqa: BLE001
            sharded = {"error": repr(exc)}
        if rank == 0:
            line["prove"] = sharded
    if rank == 0:
        print(json.dumps(line, ensure_ascii=False), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
