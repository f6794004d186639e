"""`prove` leg of bench.py: full PLONK prove wall-clock at the 2^log_n domain on one MI355X
(BASELINE.json metric, configs[1]: synthetic R1CS with 2^20 - 2 gates + 1 public input, monomial
tau = 42 SRS of 2^20 points already resident on the GPU)."""
import time

from . import _lib


HBM_PEAK_BS = 8.0e12            # MI355X_MICROARCH.md: 8 TB/s


def prove_bytes(n, commitments_nonempty=11):
    """ALGORITHMIC bytes of one proof at domain n (one compulsory read + one compulsory write of each operand, independent of
    pass count).  `survey` = SURVEY.md 8(d)'s itemisation of the REFERENCE's prove_by_steps (11 MSM + 6 NTT(N) + 18 LDE(4N) + one
    coset iNTT(4N) + point-wise passes = 8416 B per domain point); `this_prover` = the same itemisation for what THIS prover does per
    proof: the twelve constant extensions (7 selectors, 4 permutations, ...) are cached in HBM across proofs, so 6 extensions are
    left (four wires, z, public inputs); the quotient kernel reads 22 vectors of 4N and writes one; empty commitments cost nothing."""
    survey = {"11 MSM x 96N": 11 * 96 * n, "6 NTT(N) x 64N": 6 * 64 * n, "18 LDE x 160N": 18 * 160 * n, "coset iNTT(4N) 64 x 4N": 64 * 4 * n,
              "20 point-wise arrays of 4N x 32 B": 20 * 128 * n, "40 point-wise arrays of N x 32 B": 40 * 32 * n}
    this = {"%d MSM x 96N" % commitments_nonempty: commitments_nonempty * 96 * n, "6 NTT(N) x 64N": 6 * 64 * n, "6 LDE x 160N (12 constant ones cached)": 6 * 160 * n,
            "coset iNTT(4N) 64 x 4N": 64 * 4 * n, "quotient: 23 arrays of 4N x 32 B": 23 * 128 * n, "40 point-wise arrays of N x 32 B": 40 * 32 * n}
    return {"survey": sum(survey.values()), "survey_items": survey, "this_prover": sum(this.values()), "this_prover_items": this}


def with_roofline(row, n, seconds_per_proof, commitments_nonempty):
    """adds algorithmic_bytes / hbm_frac (north_star: prove-time throughput "as fraction of the HBM roofline") to a prove row"""
    b = prove_bytes(n, commitments_nonempty)
    row["algorithmic_bytes"] = b["survey"]
    row["algorithmic_GBs"] = round(b["survey"] / seconds_per_proof / 1e9, 1)
    row["hbm_frac"] = round(b["survey"] / seconds_per_proof / HBM_PEAK_BS, 4)
    row["algorithmic_bytes_this_prover"] = b["this_prover"]
    row["hbm_frac_this_prover"] = round(b["this_prover"] / seconds_per_proof / HBM_PEAK_BS, 4)
    row["algorithmic_bytes_what"] = ("algorithmic_bytes = SURVEY.md 8(d): 8416 B per domain point for the reference's 11 MSM + 6 NTT(N) + 18 LDE + coset iNTT + point-wise "
                                     "passes; _this_prover = the same itemisation for this prover (12 constant extensions cached, %d non-empty commitments, "
                                     "quotient kernel 23 arrays of 4N); hbm_frac = bytes / seconds per proof / 8 TB/s.  A proof is VALU-bound (the commitments), "
                                     "so this fraction is small by construction" % commitments_nonempty)
    return row


def cli_whole(log_n, runs=3):
    """SURVEY.md 8(d)'s third timed region: the whole `plonkit prove` process of this package (parse .r1cs/.wtns, HIP init, key read +
    upload, MSM table, setup, prove, files written) on the 2^log_n synthetic circuit, files in /dev/shm, median of `runs`; the phases are
    the binary's own PLK_CLI_TIMING=1 report of the median run.  The reference's counterpart is src/bin/main.rs:384-410."""
    import os
    import shutil
    import subprocess
    import tempfile
    cli = os.path.join(os.path.dirname(_lib.lib_path()), "plonkit")
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    d = tempfile.mkdtemp(prefix="plonkit_cli_", dir=base)
    f = lambda name: os.path.join(d, name)
    try:
        circ = _lib.Circuit.synthetic((1 << log_n) - 2)
        r1cs, wtns = circ.export("r1cs"), circ.export("wtns")
        open(f("circuit.r1cs"), "wb").write(r1cs); open(f("witness.wtns"), "wb").write(wtns)
        circ.close()
        env = dict(os.environ, PLK_CLI_TIMING="1")
        subprocess.check_call([cli, "setup", "-p", str(log_n), "-m", f("key.bin"), "--overwrite"], stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL)
        subprocess.check_call([cli, "export-verification-key", "-m", f("key.bin"), "-c", f("circuit.r1cs"), "-v", f("vk.bin"), "--overwrite"],
                              stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL)
        res = []
        for _ in range(runs):
            t0 = time.perf_counter()
            p = subprocess.run([cli, "prove", "-m", f("key.bin"), "-c", f("circuit.r1cs"), "-w", f("witness.wtns"), "-p", f("proof.bin"),
                                "-j", f("proof.json"), "-i", f("public.json"), "--overwrite"], env=env, capture_output=True, text=True, timeout=600)
            dt = time.perf_counter() - t0
            if p.returncode != 0:
                raise RuntimeError("plonkit prove exited %d: %s" % (p.returncode, p.stderr[-300:]))
            res.append((dt, p.stderr))
        res.sort(key=lambda r: r[0])
        whole, log = res[len(res) // 2]
        phases = {}
        import re
        for ln in log.splitlines():
            m = re.match(r"\[timing\]\s+(.*?)\s+\+([0-9.]+) s", ln)            # "[timing] <phase, may contain '+'>   +0.123 s (total)"
            if m:
                phases[m.group(1).strip()] = float(m.group(2))
        t0 = time.perf_counter()
        ok = subprocess.call([cli, "verify", "-p", f("proof.bin"), "-v", f("vk.bin")], stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL) == 0
        verify_s = time.perf_counter() - t0
        return {"whole_s": round(whole, 3), "whole_s_min": round(res[0][0], 3), "whole_s_max": round(res[-1][0], 3), "runs": runs, "phases_s": phases,
                "verify_whole_s": round(verify_s, 3), "verified": bool(ok), "domain": 1 << log_n,
                "files_MB": {"r1cs": round(len(r1cs) / 1e6, 1), "wtns": round(len(wtns) / 1e6, 1), "key": round(os.path.getsize(f("key.bin")) / 1e6, 1)},
                "what": "wall clock of the whole `plonkit prove` process of this package (C ABI only), files in %s, median of %d; phases = its own "
                        "PLK_CLI_TIMING report (main thread; the GPU thread's lines are indented), beside cpu_baseline.prove.cpu_whole_s" % (base or "the temp dir", runs)}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def _run_timed(cmd, env=None, timeout=900):
    import subprocess
    t0 = time.perf_counter()
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    dt = time.perf_counter() - t0
    if p.returncode != 0:
        raise RuntimeError("%s exited %d: %s" % (" ".join(cmd[:2]), p.returncode, p.stderr[-300:]))
    return dt, p.stderr


def cli_table(log_n, circuit_files=None, key_log_n=None):
    """Whole-PROCESS wall clock of every command of the reference's CI sequence (.github/workflows/integration-test.yml:105-154), in its order:
    setup, export-verification-key, prove, dump-lagrange, prove -l, verify — this package's `plonkit` binary (C ABI only), files in /dev/shm.
    `prove` is the median of three; the others run once.  circuit_files = (r1cs bytes, wtns bytes) of another circuit (the CI proves a 2^12-domain
    Poseidon circuit against the 2^20 key: key_log_n = 20); default: the synthetic 2^log_n circuit against a key of its own size."""
    import os
    import re
    import shutil
    import tempfile
    cli = os.path.join(os.path.dirname(_lib.lib_path()), "plonkit")
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    d = tempfile.mkdtemp(prefix="plonkit_cli_", dir=base)
    f = lambda name: os.path.join(d, name)
    try:
        if circuit_files is None:
            circ = _lib.Circuit.synthetic((1 << log_n) - 2)
            r1cs, wtns = circ.export("r1cs"), circ.export("wtns")
            circ.close()
        else:
            r1cs, wtns = circuit_files
        open(f("circuit.r1cs"), "wb").write(r1cs); open(f("witness.wtns"), "wb").write(wtns)
        env = dict(os.environ, PLK_CLI_TIMING="1")
        out = {"domain": 1 << log_n, "key_points": 1 << (key_log_n or log_n)}
        out["setup_s"] = round(_run_timed([cli, "setup", "-p", str(key_log_n or log_n), "-m", f("key.bin"), "--overwrite"])[0], 3)
        out["export_verification_key_s"] = round(_run_timed([cli, "export-verification-key", "-m", f("key.bin"), "-c", f("circuit.r1cs"), "-v", f("vk.bin"), "--overwrite"])[0], 3)
        prove = [cli, "prove", "-m", f("key.bin"), "-c", f("circuit.r1cs"), "-w", f("witness.wtns"), "-p", f("proof.bin"), "-j", f("proof.json"), "-i", f("public.json"), "--overwrite"]
        res = sorted((_run_timed(prove, env) for _ in range(3)), key=lambda r: r[0])
        out["prove_s"], out["prove_s_min"], out["prove_s_max"] = round(res[1][0], 3), round(res[0][0], 3), round(res[2][0], 3)
        phases = {}
        for ln in res[1][1].splitlines():
            m = re.match(r"\[timing\]\s+(.*?)\s+\+([0-9.]+) s", ln)            # "[timing] <phase, may contain '+'>   +0.123 s (total)"
            if m:
                phases[m.group(1).strip()] = float(m.group(2))
        out["prove_phases_s"] = phases
        proof = open(f("proof.bin"), "rb").read()
        out["dump_lagrange_s"] = round(_run_timed([cli, "dump-lagrange", "-m", f("key.bin"), "-l", f("key_lagrange.bin"), "-c", f("circuit.r1cs"), "--overwrite"])[0], 3)
        out["prove_l_s"] = round(_run_timed(prove[:4] + ["-l", f("key_lagrange.bin")] + prove[4:], env)[0], 3)
        out["prove_l_same_bytes"] = open(f("proof.bin"), "rb").read() == proof
        import subprocess
        t0 = time.perf_counter()
        ok = subprocess.call([cli, "verify", "-p", f("proof.bin"), "-v", f("vk.bin")], stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL) == 0
        out["verify_s"] = round(time.perf_counter() - t0, 3)
        out["verified"] = bool(ok)
        out["files_MB"] = {"r1cs": round(len(r1cs) / 1e6, 1), "wtns": round(len(wtns) / 1e6, 1), "key": round(os.path.getsize(f("key.bin")) / 1e6, 1)}
        out["where"] = base or "temp dir"
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def prove_row(ctx, circ, reps):
    """one setup + `reps` warm proofs of `circ` on ctx (its key must be resident and large enough): median wall clock, the hbm fraction of
    SURVEY.md 8(d)'s itemisation, setup time; returns (row, proof bytes)"""
    t0 = time.perf_counter()
    setup = _lib.SetupForProver(ctx, circ)
    ctx.synchronize()
    t_setup = time.perf_counter() - t0
    n = setup.domain_size
    proof = setup.prove(circ)
    setup.prove(circ)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        p = setup.prove(circ)
        ts.append(time.perf_counter() - t0)
        assert p == proof
    ts.sort()
    wall = ts[len(ts) // 2]
    vk = setup.verification_key_bytes(_lib.crs42_g2_bytes())
    ok = _lib.verify(vk, proof)
    setup.close()
    b = prove_bytes(n, nonempty_commitments(proof))
    return {"domain": n, "wall_s": round(wall, 5), "wall_s_min": round(ts[0], 5), "proofs_timed": reps, "setup_prepare_s": round(t_setup, 3),
            "hbm_frac": round(b["survey"] / wall / HBM_PEAK_BS, 5), "commitments_nonempty": nonempty_commitments(proof), "verified": bool(ok)}, proof


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


def nonempty_commitments(proof):
    """how many of the 11 commitments of a proof.bin (4 wires, z, 4 quotient chunks, 2 openings: SURVEY.md A.1) are not the
    point at infinity — the all-zero vector commits to infinity, written as 0x40 00.. (or all zeros)"""
    n_in = int.from_bytes(proof[8:16], "big")
    off = 16 + 32 * n_in
    pts = []
    off += 8
    pts += [proof[off + 64 * k: off + 64 * (k + 1)] for k in range(4)]; off += 256
    pts.append(proof[off: off + 64]); off += 64
    off += 8
    pts += [proof[off + 64 * k: off + 64 * (k + 1)] for k in range(4)]
    pts += [proof[-128:-64], proof[-64:]]
    return sum(1 for p in pts if any(p[1:]) or p[0] not in (0, 0x40))


def throughput(ctx, log_n, in_flight=2, proofs_each=10, lc_terms=0, setup=None, circs=None):
    """prove THROUGHPUT on one GPU: `in_flight` host threads, one context each on the same device (the extra ones borrow
    ctx's key and MSM table: plk_ctx_share_srs), ONE shared setup, every thread proving its own witness of the circuit
    `proofs_each` times back to back.  A proof's challenge chain is strict, so ~3.5 ms of a 2^20 proof is latency-bound or
    idle; a second proof in flight fills it.  Every proof must equal, byte for byte, the proof of the same witness made
    alone on `ctx` before.  Returns proofs/s and ms per proof next to the sequential figures of the same run."""
    import threading
    n_gates = (1 << log_n) - 2
    own_circs = circs is None
    if own_circs:
        circs = [_lib.Circuit.synthetic_ex(n_gates, witness_seed=(0 if k == 0 else 1000 + k), lc_terms=lc_terms) for k in range(in_flight)]
    own_setup = setup is None
    if own_setup:
        setup = _lib.SetupForProver(ctx, circs[0])
    ctxs = [ctx]
    try:
        for _ in range(in_flight - 1):
            c2 = _lib.Context(ctx.device)
            c2.share_srs_from(ctx)
            ctxs.append(c2)
        # reference proofs, made one at a time on the first context (also the warm-up of that context)
        want = [setup.prove(c) for c in circs]
        assert len(set(want)) == len(want), "different witnesses must give different proofs"
        for k in range(1, in_flight):                               # warm-up of the other contexts (workspaces, twiddle tables)
            assert setup.prove(circs[k], ctx=ctxs[k]) == want[k]
        # sequential: the same number of proofs one after the other on one context
        total = in_flight * proofs_each
        t0 = time.perf_counter()
        for i in range(total):
            p = setup.prove(circs[i % in_flight])
            assert p == want[i % in_flight]
        seq_s = time.perf_counter() - t0
        # concurrent
        lat = [[] for _ in range(in_flight)]
        bad = []
        gate = threading.Barrier(in_flight + 1)

        def worker(k):
            gate.wait()
            for _ in range(proofs_each):
                t = time.perf_counter()
                try:
                    p = setup.prove(circs[k], ctx=ctxs[k])
                except Exception as exc:                             # noqa: BLE001
                    bad.append(repr(exc)); return
                lat[k].append(time.perf_counter() - t)
                if p != want[k]:
                    bad.append("thread %d: proof differs from the sequential one" % k)
        th = [threading.Thread(target=worker, args=(k,)) for k in range(in_flight)]
        for t in th:
            t.start()
        gate.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        par_s = time.perf_counter() - t0
    finally:
        # the borrowers go first, whatever happened: while one exists the lender refuses to replace its key, and the caller's
        # next leg (bench.py: the kernel table regenerates the SRS) would fail with it
        for c2 in ctxs[1:]:
            c2.close()
        if own_setup:
            setup.close()
        if own_circs:
            for c in circs:
                c.close()
    if bad:
        raise RuntimeError("concurrent proving failed: " + "; ".join(bad[:3]))
    all_lat = sorted(x for l in lat for x in l)
    return with_roofline({"in_flight": in_flight, "proofs": total, "proofs_per_s": round(total / par_s, 2), "ms_per_proof": round(par_s / total * 1e3, 3),
            "latency_ms_median": round(all_lat[len(all_lat) // 2] * 1e3, 3), "latency_ms_max": round(all_lat[-1] * 1e3, 3),
            "sequential": {"proofs_per_s": round(total / seq_s, 2), "ms_per_proof": round(seq_s / total * 1e3, 3)},
            "speedup_vs_sequential": round(seq_s / par_s, 3), "byte_identical_to_sequential": True, "domain": 1 << log_n,
            "what": "%d host threads x %d proofs, one context per thread on one GPU (key and MSM table shared: plk_ctx_share_srs), one shared setup, "
                    "a different witness per thread; sequential = the same %d proofs one after the other on one context in the same run"
                    % (in_flight, proofs_each, total)}, 1 << log_n, par_s / total, nonempty_commitments(want[0]))


def run_dense(ctx, log_n, lc_terms=7, reps=6):
    """a proof that does ALL of the prover's work: circom-Poseidon-shaped constraints (A side = linear combination of
    `lc_terms` wires + constant) that the transpiler folds through the d column with q_d_next = -1 — d, q_d_next and the
    fourth quotient chunk are live, 11 of 11 commitments non-trivial.  PARITY UNPINNED for the chaining rule."""
    circ = _lib.Circuit.synthetic_ex((1 << log_n) - 2, lc_terms=lc_terms)
    setup = _lib.SetupForProver(ctx, circ)
    assert setup.domain_size == 1 << log_n
    proof = setup.prove(circ)
    runs = []
    for _ in range(reps):
        t0 = time.perf_counter()
        p = setup.prove(circ)
        dt = time.perf_counter() - t0
        assert p == proof
        runs.append((dt, setup.timings_ms()))
    runs.sort(key=lambda r: r[0])
    best, phases = runs[len(runs) // 2]
    vk = setup.verification_key_bytes(_lib.crs42_g2_bytes())
    ok = _lib.verify(vk, proof)
    bad = bytearray(proof); bad[-200] ^= 1                             # an evaluation
    rejected = not _lib.verify(vk, bytes(bad))
    setup.close(); circ.close()
    return with_roofline({"wall_s": round(best, 4), "wall_s_min": round(runs[0][0], 4), "proves_timed": reps, "domain": 1 << log_n, "lc_terms": lc_terms,
            "commitments_nonempty": nonempty_commitments(proof), "rounds_ms": {k: round(v, 2) for k, v in phases.items()},
            "verified": bool(ok), "tampered_rejected": bool(rejected), "parity": "unpinned",
            "what": "synthetic circuit whose constraints carry %d-term linear combinations (Poseidon-round shape): folded through the d column, "
                    "so the d wire, q_d_next and t_3 are live; verified by the host verifier (real pairing) in the same run" % lc_terms},
                         1 << log_n, best, nonempty_commitments(proof))


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
    return with_roofline({"wall_s": round(best, 4), "wall_s_min": round(runs[0][0], 4), "wall_s_max": round(runs[-1][0], 4), "proves_timed": reps,
            "commitments_nonempty": nonempty_commitments(proof),
            "cold": cold_info, "domain": 1 << log_n, "proof_bytes": len(proof),
            "rounds_ms": {k: round(v, 2) for k, v in phases.items()},
            "gpu_rounds_s": round(gpu_ms / 1e3, 4),
            "setup_prepare_s": round(t_setup, 3), "circuit_generation_s": round(t_synth, 3),
            "what": "SetupForProver::prove (witness synthesis + satisfiability check on the host, rounds 1-5 on the GPU, "
                    "Proof::write); setup_prepare = transpile + 11 iNTT, timed separately as in the reference's CLI; "
                    "wall_s = median of `proves_timed` warm proofs (tables, cached constant extensions and allocations in place), cold = first proof of a fresh context"},
                         1 << log_n, best, nonempty_commitments(proof))


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
    # BASELINE.json configs[3] at its stated size: dump-lagrange = Crs::<Lagrange>::from_powers (src/plonk.rs:179-185), the inverse
    # NTT over G1 of the first 2^20 points of the resident key (the 2^24-point key generated above)
    lg = 20
    m = 1 << lg
    o = torch.empty((m, 8), dtype=torch.int64, device=device)
    ctx.g1_intt_srs_dev(lg, o)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        ctx.g1_intt_srs_dev(lg, o)
    ctx.synchronize()
    ms = (time.perf_counter() - t0) / 2 * 1e3
    muls = (m // 2) * lg - (m - 1) + m                                # butterflies with a twiddle != 1, + the 1/N scaling of every point
    ec_ops = muls * (129 + 86) + (m // 2) * lg * 2                    # GLV: 129 doublings + 86 additions per multiplication; 2 additions per butterfly
    out["g1_intt_2^20"] = {"ms": round(ms, 2), "scalar_muls": muls, "G_ec_ops_s": round(ec_ops / ms / 1e6, 2), "algorithmic_GBs": round(128 * m / ms / 1e6, 3),
                           "hbm_frac": round(128 * m / ms / 1e6 / 8000.0, 6),
                           "what": "2^20 points in, 2^20 Lagrange-basis points out (64 + 64 B per point algorithmic); 20 radix-2 stages of "
                                   "254-bit scalar multiplications on group elements, GLV (129 doublings + 86 additions each): VALU-bound"}
    del o
    if keep:
        ctx.srs_generate(keep, 0, 42)
    return out
