"""Multi-GPU prove with the SRS sliced across ranks (plk_set_commit_shard + plonkit_amd.sharded.ShardedProver).
The GPU box of the test tier has ONE MI355X, so the two ranks share device 0 and exchange their partial sums over
gloo; on a real node the same code runs one rank per GPU over RCCL (bench.py --gpus N).  Every rank must end with the
proof and verification-key bytes a single GPU produces with the whole SRS."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank(rank, world, port, log_n, lagrange, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import plonkit_amd as pa
    from plonkit_amd.sharded import ShardedProver
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 1 << log_n
        local = n // world
        ctx = pa.Context(0)
        ctx.srs_generate(local, rank * local, 42)                  # this rank's slice of the tau = 42 key only
        keep = None
        if lagrange:                                                # slice of the Lagrange-form key L_i(42)*G
            full = pa.Context(0)
            full.srs_generate(n, 0, 42)
            keep = torch.zeros((n, 8), dtype=torch.int64, device="cuda:0")
            full.g1_intt_srs_dev(log_n, keep.data_ptr())
            full.synchronize()
            ctx.srs_lagrange_upload(keep.cpu().numpy().view(np.uint64)[rank * local:(rank + 1) * local])
            full.close()
        sp = ShardedProver(ctx, dist, None)
        circ = pa.Circuit.synthetic(n - 2)
        setup = pa.SetupForProver(ctx, circ)
        vk = setup.verification_key_bytes(pa.crs42_g2_bytes())
        proof = setup.prove(circ)
        sp.close()
        q.put((rank, vk, proof))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("log_n,lagrange", [(12, False), (14, True)])
def test_two_ranks_produce_the_single_gpu_proof(log_n, lagrange):
    import plonkit_amd as pa
    n = 1 << log_n
    ctx = pa.Context(0)
    ctx.srs_generate(n, 0, 42)
    circ = pa.Circuit.synthetic(n - 2)
    setup = pa.SetupForProver(ctx, circ)
    want_vk, want_proof = setup.verification_key_bytes(pa.crs42_g2_bytes()), setup.prove(circ)
    assert pa.verify(want_vk, want_proof)
    world, port = 2, _free_port()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_rank, args=(r, world, port, log_n, lagrange, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    for rank, vk, proof in res:
        assert vk == want_vk, rank
        assert proof == want_proof, rank


# ---------------------------------------------------------------- the exchange below the C ABI (comm.cpp, plk_comm_init)
def test_rccl_communicator_of_one_rank():
    """plk_comm_unique_id + plk_comm_init over the real RCCL (librccl.so.1 bound at run time): a communicator of one
    rank on device 0, every commitment of the verification key and the proof goes through ncclAllGather; same bytes as
    without it.  (RCCL refuses two ranks on one device, so world > 1 over RCCL needs a multi-GPU node: bench.py --gpus N.)"""
    import plonkit_amd as pa
    n = 1 << 12
    ctx = pa.Context(0)
    ctx.srs_generate(n, 0, 42)
    circ = pa.Circuit.synthetic(n - 2)
    setup = pa.SetupForProver(ctx, circ)
    want_vk, want_proof = setup.verification_key_bytes(pa.crs42_g2_bytes()), setup.prove(circ)
    ctx.comm_init(0, 1, pa.comm_unique_id(), 0)
    assert ctx.comm_info()[:2] == (0, 1)
    assert setup.verification_key_bytes(pa.crs42_g2_bytes()) == want_vk
    assert setup.prove(circ) == want_proof
    assert ctx.comm_info()[2] == 2 + 4                           # 11 commitments of the key in two batches; a proof has 4 batches (4 wires, z, 4 quotient parts, 2 openings)
    # the transport of owner-computes mode over the same communicator: ncclBroadcast + grouped ncclSend / ncclRecv (one rank: to and from itself)
    ctx.comm_selftest()
    ctx.comm_selftest()
    assert setup.prove(circ) == want_proof                       # ... and the exchange stream is still in step afterwards
    ctx.comm_destroy()
    with pytest.raises(Exception):
        ctx.comm_selftest()                                      # no communicator: an error, not a crash
    assert setup.prove(circ) == want_proof and ctx.comm_info() == (0, 1, 0)
    ctx.close()


def test_scatter_step_through_rccl_with_scalars_in_flight():
    """comm_send_work -> comm_recv_work through the REAL RCCL branch (one-rank communicator, rank 0 receives its own
    share inside the sender's group), 200 times, each time with the scalar vector still being rewritten on the producer stream behind a
    kernel that parks it for 0.2 ms when the step is called — the TCP tier synchronises the producer first and cannot see a missing wait.
    The commitment of what arrived equals the commitment of the vector itself every time; the proof bytes and the exchange counter are
    untouched afterwards; ncclCommCount says one rank.  Negative control in a subprocess: with the producer wait left out
    (PLK_COMM_TEST_SKIP_PRODUCER_WAIT=1) the same loop DOES see stale vectors, i.e. the test can fail."""
    import subprocess
    import sys
    import plonkit_amd as pa
    n = 1 << 12
    ctx = pa.Context(0)
    ctx.srs_generate(n, 0, 42)
    circ = pa.Circuit.synthetic(n - 2)
    setup = pa.SetupForProver(ctx, circ)
    want = setup.prove(circ)
    ctx.comm_init(0, 1, pa.comm_unique_id(), 0)
    assert ctx.comm_nccl_count() == 1
    assert ctx.comm_scatter_selftest(12, 200) == 0
    assert ctx.comm_scatter_selftest(11, 20) == 0                   # a shorter vector than the key
    assert setup.prove(circ) == want                                # the communicator is still in step (replicate mode, one rank)
    ctx.comm_selftest()
    ctx.comm_destroy()
    assert ctx.comm_nccl_count() == 0
    setup.close(); circ.close(); ctx.close()
    code = """
import plonkit_amd as pa
ctx = pa.Context(0)
ctx.srs_generate(1 << 12, 0, 42)
ctx.comm_init(0, 1, pa.comm_unique_id(), 0)
print("BAD", ctx.comm_scatter_selftest(12, 40))
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, PLK_COMM_TEST_SKIP_PRODUCER_WAIT="1"))
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    bad = int([ln for ln in r.stdout.splitlines() if ln.startswith("BAD")][0].split()[1])
    assert bad >= 30, "without the producer wait the received vectors must be stale: %d of 40 differed" % bad


def _scatter_unsat_rank(rank, world, port, log_n, q):
    """owner-computes mode: the owner's first proof has a witness that does not satisfy the circuit (PLK_ERR_UNSAT after the wire batch went
    out to the workers), the second one is good"""
    import time
    import plonkit_amd as pa
    n = 1 << log_n
    local = n // world
    ctx = pa.Context(0)
    ctx.srs_generate(local, rank * local, 42)
    ctx.comm_init_tcp(rank, world, port, rank * local)
    ctx.comm_set_mode("scatter")
    if rank == 0:
        res = None
        try:
            good = pa.Circuit.synthetic(n - 2)
            r1cs, wtns = good.export("r1cs"), bytearray(good.export("wtns"))
            wtns[len(wtns) // 2 & ~31] ^= 1                          # one witness value of the middle of the file is off by one bit
            bad = pa.Circuit(r1cs, False, bytes(wtns), False)
            setup = pa.SetupForProver(ctx, good)
            t0 = time.perf_counter()
            try:
                setup.prove(bad)
                err = None
            except pa.PlkError as e:
                err = e.code
            dt = time.perf_counter() - t0
            proof = setup.prove(good)                               # the communicator must still work, and at once
            res = (0, err, dt, proof)
        finally:
            ctx.comm_stop_workers()
        q.put(res)
    else:
        try:
            q.put((rank, None, 0.0, ctx.comm_serve()))
        except Exception as exc:                                    # noqa: BLE001
            q.put((rank, repr(exc), 0.0, -1))
    ctx.close()


def test_scatter_mode_survives_an_unsatisfied_witness():
    """round-5 advisor finding (medium): in owner-computes mode the wire batch is sent to the workers BEFORE the satisfiability verdict is read;
    a "must satisfy" return used to leave them in that batch's all-gather for the 180 s exchange deadline and then break the communicator.
    Now the exchange is run with empty sums on the error path: the owner gets PLK_ERR_UNSAT within a second, the NEXT proof on the same
    communicator equals the single-GPU one, and every worker has served both batches' worth."""
    import plonkit_amd as pa
    log_n = 12
    n = 1 << log_n
    ctx = pa.Context(0)
    ctx.srs_generate(n, 0, 42)
    circ = pa.Circuit.synthetic(n - 2)
    setup = pa.SetupForProver(ctx, circ)
    want = setup.prove(circ)
    setup.close(); circ.close(); ctx.close()
    world = 4
    port = _free_port()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_scatter_unsat_rank, args=(r, world, port, log_n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    _, err, dt, proof = res[0]
    assert err == 5 and dt < 5.0, (err, dt)                         # PLK_ERR_UNSAT, not a 180 s wait
    assert proof == want
    for rank, e, _, served in res[1:]:
        assert e is None and served == 1 + 4, (rank, e, served)     # the broken proof's wire batch + the good proof's four


def test_two_plonkit_processes_share_the_key(tmp_path):
    """one `plonkit` process per rank (PLONKIT_WORLD / PLONKIT_RANK / PLONKIT_COMM, cli_main.cpp), no Python and no torch
    in the exchange: each rank keeps half of the key resident, the built-in combiner joins the partial sums (TCP hub,
    because both ranks sit on this box's single GPU); rank 0 writes vk.bin / proof.bin, identical to the one-process files"""
    import subprocess
    import plonkit_amd as pa
    cli = os.path.join(os.path.dirname(pa.lib_path()), "plonkit")
    log_n = 14
    circ = pa.Circuit.synthetic((1 << log_n) - 2)
    f = lambda name: str(tmp_path / name)
    open(f("c.r1cs"), "wb").write(circ.export("r1cs"))
    open(f("w.wtns"), "wb").write(circ.export("wtns"))
    subprocess.check_call([cli, "setup", "-p", str(log_n), "-m", f("key.bin")], stderr=subprocess.DEVNULL)
    subprocess.check_call([cli, "export-verification-key", "-m", f("key.bin"), "-c", f("c.r1cs"), "-v", f("vk1.bin")], stderr=subprocess.DEVNULL)
    subprocess.check_call([cli, "prove", "-m", f("key.bin"), "-c", f("c.r1cs"), "-w", f("w.wtns"), "-p", f("p1.bin"), "-j", f("j1.json"), "-i", f("i1.json")],
                          stderr=subprocess.DEVNULL)
    for cmd, outs in ((["export-verification-key", "-m", f("key.bin"), "-c", f("c.r1cs"), "-v", f("vk2.bin")], ("vk1.bin", "vk2.bin")),
                      (["prove", "-m", f("key.bin"), "-c", f("c.r1cs"), "-w", f("w.wtns"), "-p", f("p2.bin"), "-j", f("j2.json"), "-i", f("i2.json")], ("p1.bin", "p2.bin"))):
        port = _free_port()
        procs = []
        for rank in range(2):
            env = dict(os.environ, PLONKIT_WORLD="2", PLONKIT_RANK=str(rank), PLONKIT_COMM="tcp:%d" % port, PLONKIT_DEVICE="0")
            procs.append(subprocess.Popen([cli] + cmd, env=env, stderr=subprocess.PIPE))
        for p in procs:
            _, err = p.communicate(timeout=300)
            assert p.returncode == 0, err.decode()[-2000:]
        assert open(f(outs[0]), "rb").read() == open(f(outs[1]), "rb").read()
    assert pa.verify(open(f("vk2.bin"), "rb").read(), open(f("p2.bin"), "rb").read())
    # the RCCL transport through the same binary: a communicator of one rank (id file written and read back).  A file left
    # at the path by an earlier run (here: garbage of the right size) must not survive: rank 0 unlinks it first
    open(f("rccl.id"), "wb").write(b"\x5a" * 128)
    env = dict(os.environ, PLONKIT_WORLD="1", PLONKIT_RANK="0", PLONKIT_COMM="rccl:" + f("rccl.id"))
    subprocess.check_call([cli, "prove", "-m", f("key.bin"), "-c", f("c.r1cs"), "-w", f("w.wtns"), "-p", f("p3.bin"), "-j", f("j3.json"), "-i", f("i3.json")],
                          env=env, stderr=subprocess.DEVNULL)
    assert open(f("p3.bin"), "rb").read() == open(f("p1.bin"), "rb").read() and not os.path.exists(f("rccl.id"))   # rank 0 removes the id file once the communicator exists


def _native_rank(rank, world, port, log_n, q):
    import plonkit_amd as pa
    n = 1 << log_n
    local = n // world
    ctx = pa.Context(0)
    ctx.srs_generate(local, rank * local, 42)
    ctx.comm_init_tcp(rank, world, port, rank * local)
    circ = pa.Circuit.synthetic(n - 2)
    setup = pa.SetupForProver(ctx, circ)
    vk, proof, info = setup.verification_key_bytes(pa.crs42_g2_bytes()), setup.prove(circ), ctx.comm_info()
    # the commitment-level form: every rank enqueues ITS slice of three global scalar vectors (a batch), the built-in
    # combiner joins the partial sums — one exchange for the batch
    import numpy as np
    import torch
    from plonkit_amd.sharded import ShardedMsm
    rng = np.random.default_rng(4242)                               # the same global vectors on every rank
    glob = rng.integers(0, 1 << 62, size=(3, n, 4), dtype=np.uint64)
    glob[:, :, 3] &= np.uint64((1 << 60) - 1)
    mine = [torch.from_numpy(np.ascontiguousarray(glob[k, rank * local:(rank + 1) * local]).view(np.int64)).to("cuda:0") for k in range(3)]
    torch.cuda.synchronize()
    outs = list(ShardedMsm(ctx, None, None, native=True).commit_batches([mine], local))
    q.put((rank, vk, proof, info, outs[0].tobytes()))
    ctx.close()


def test_two_ranks_with_the_builtin_combiner():
    """the Python mirror of the same path: Context.comm_init_tcp instead of ShardedProver's torch.distributed combiner"""
    import plonkit_amd as pa
    log_n, world = 13, 2
    n = 1 << log_n
    ctx = pa.Context(0)
    ctx.srs_generate(n, 0, 42)
    circ = pa.Circuit.synthetic(n - 2)
    setup = pa.SetupForProver(ctx, circ)
    want = (setup.verification_key_bytes(pa.crs42_g2_bytes()), setup.prove(circ))
    rng = np.random.default_rng(4242)
    glob = rng.integers(0, 1 << 62, size=(3, n, 4), dtype=np.uint64)
    glob[:, :, 3] &= np.uint64((1 << 60) - 1)
    want_commitments = np.stack([ctx.msm(glob[k]) for k in range(3)]).tobytes()
    port = _free_port()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_native_rank, args=(r, world, port, log_n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    for rank, vk, proof, info, commitments in res:
        assert (vk, proof) == want, rank
        assert info == (rank, world, 6)
        assert commitments == want_commitments, rank


def test_bench_n2_control_flow_on_one_gpu():
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank), with both ranks on this
    box's single GPU (PLK_BENCH_SHARE_DEVICE: gloo + the TCP transport instead of RCCL): the weak-scaling headline, the
    strong-scaling leg (one commitment, SRS split over the ranks, next to the same commitment on rank 0 alone) and the sharded
    prove all complete and the line has the fields the driver and DESIGN.md §5 name.  Timings are meaningless here."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PLK_BENCH_SHARE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
           "--log-n", "16", "--strong-log-n", "18"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0 and line["steps"] == 4
    assert line["roofline"]["kernel_ms"] > 0 and line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["value"] > 0
    assert line["value_sustained"] > 0 and line["config"]["settle_steps"] == 0 and line["config"]["comm_ranks"] == 0   # (TCP tier: RCCL sees no rank)
    st = line["strong"]
    assert "error" not in st and st["terms_total"] == 1 << 18 and st["terms_per_gpu"] == 1 << 17 and st["scaling_vs_1gpu"] > 0
    assert "error" not in line["prove"] and line["prove"]["n_gpus"] == 2 and line["prove"]["proof_bytes"] == 1144
    assert line["strong_value"] == st["Mscalar_mul_s"] and line["strong_scaling_vs_1gpu"] == st["scaling_vs_1gpu"]
    sc = line["prove"]["scatter"]                                     # the same prove in owner-computes mode (rank 0 proves, rank 1 serves)
    assert "error" not in sc and sc["n_gpus"] == 2 and sc["verified"] is True and sc["wall_s"] > 0
    assert line["failed_legs"] == []
    pt = line["prove_throughput"]                                     # every GPU proving on its own (replicas), two proofs in flight each
    assert "error" not in pt and pt["n_gpus"] == 2 and pt["in_flight_per_gpu"] == 2 and pt["proofs_per_s"] > 0


def test_bench_leg_watchdog_prints_the_line_and_exits():
    """a leg after the headline that never returns (here: every rank parked before the sharded prove) must not cost the line already
    measured: after PLK_BENCH_LEG_TIMEOUT_S rank 0 prints it, the stuck leg named in `failed_legs`, and the launch ends non-zero"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PLK_BENCH_SHARE_DEVICE="1", PLK_BENCH_LEG_TIMEOUT_S="25", PLK_BENCH_TEST_STALL_LEG="prove(sharded)")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--log-n", "14", "--strong-log-n", "16", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (r.stdout + r.stderr)[-3000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0
    assert "error" not in line["strong"] and line["strong_scaling_vs_1gpu"] > 0      # the leg before the stuck one is in the line
    assert [f["leg"] for f in line["failed_legs"]] == ["prove(sharded)"] and "did not return" in line["failed_legs"][0]["error"]


def test_bench_falls_back_to_torch_distributed_without_the_library_communicator():
    """if the library's communicator cannot be opened on some rank, every rank exchanges the partial sums through torch.distributed
    instead: the headline and the strong-scaling leg still come out, the legs that need the communicator say so in `failed_legs`"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PLK_BENCH_SHARE_DEVICE="1", PLK_BENCH_TEST_NO_COMM="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--log-n", "14", "--strong-log-n", "16", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and "fallback" in line["config"]["exchange"]
    assert "error" not in line["strong"] and line["strong_scaling_vs_1gpu"] > 0
    failed = [f["leg"] for f in line["failed_legs"]]
    assert failed[0] == "comm_init" and not any(f["correctness"] for f in line["failed_legs"])
    assert "error" not in line["prove_throughput"] and line["prove_throughput"]["proofs_per_s"] > 0      # replicas need no communicator


def _run_plain_bench(n_gpus, extra, timeout=900):
    """`python bench.py --gpus N ...` invoked PLAINLY (no launcher): bench.py must spawn its own N ranks"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PLK_BENCH_SHARE_DEVICE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n_gpus)] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines                                    # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_plain_bench_gpus_2_spawns_its_ranks():
    """the driver's N = 1 form with N = 2: `python3 bench.py --gpus 2 --steps K --warmup W` and nothing else"""
    line = _run_plain_bench(2, ["--steps", "3", "--warmup", "1", "--log-n", "14", "--strong-log-n", "16", "--no-cpu-baseline"])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["value"] > 0
    assert "error" not in line["strong"] and "error" not in line["prove"] and line["strong_scaling_vs_1gpu"] > 0


def test_eight_rank_rehearsal_on_one_gpu():
    """the 8-GPU job of BASELINE.json configs[2] / north_star, rehearsed with EIGHT ranks on this box's single GPU
    (PLK_BENCH_SHARE_DEVICE: gloo for the barriers, the library's TCP transport for the partial sums, because RCCL refuses
    two ranks on one device): all three legs complete at world 8 — weak-scaling headline, strong scaling of one commitment
    with the key split in eight, sharded prove — and every rank derived the same proof (sharded_prove asserts equality with
    its warm-up proof on each rank; the combiner gives every rank the same bytes).  Timings mean nothing here; what this
    guards is that the day an 8-GPU node runs `bench.py --gpus 8`, nothing in the control flow is new."""
    line = _run_plain_bench(8, ["--steps", "3", "--warmup", "1", "--log-n", "14", "--strong-log-n", "17"], timeout=1200)
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["value"] > 0
    assert line["config"]["terms_per_gpu"] == 1 << 14
    st = line["strong"]
    assert "error" not in st, st
    assert st["n_gpus"] == 8 and st["terms_total"] == 1 << 17 and st["terms_per_gpu"] == 1 << 14 and st["scaling_vs_1gpu"] > 0
    pr = line["prove"]
    assert "error" not in pr, pr
    assert pr["n_gpus"] == 8 and pr["srs_points_per_gpu"] == (1 << 14) // 8 and pr["proof_bytes"] == 1144
    assert pr["same_proof_on_every_rank"] is True and pr["verified"] is True
    assert "error" not in pr["scatter"] and pr["scatter"]["verified"] is True and pr["scatter"]["n_gpus"] == 8      # owner-computes mode at world 8
    assert line["failed_legs"] == []
    assert line["strong_value"] > 0 and line["cpu_baseline"]["kind"] == "port"


def test_rccl_watchdog_aborts_instead_of_hanging():
    """comm.cpp watches the RCCL exchange with a deadline (hipStreamQuery polling + ncclCommAbort) instead of a blind
    hipStreamSynchronize.  Deterministic: the test hook PLK_COMM_TEST_STALL_MS parks the exchange stream for 1.5 s before
    every all-gather (a peer that does not answer) and the deadline is 100 ms, so the abort path MUST run: the first
    proof comes back with PLK_ERR_HIP "RCCL exchange aborted ... before the deadline" after ONE stall, the second
    fails at once as "aborted earlier", and after plk_comm_destroy the context proves again.  (The reference panics and
    exits when a worker fails, src/bin/main.rs:335,371,399.)"""
    import subprocess
    import sys
    code = r"""
import os, sys, time
os.environ["PLK_COMM_TIMEOUT_MS"] = "100"
os.environ["PLK_COMM_TEST_STALL_MS"] = "1500"
import plonkit_amd as pa
n = 1 << 12
ctx = pa.Context(0)
ctx.srs_generate(n, 0, 42)
circ = pa.Circuit.synthetic(n - 2)
setup = pa.SetupForProver(ctx, circ)
want = setup.prove(circ)
ctx.comm_init(0, 1, pa.comm_unique_id(), 0)
errs = []
for _ in range(2):
    t0 = time.perf_counter()
    try:
        setup.prove(circ)
        errs.append(("no error", 0, time.perf_counter() - t0))
    except pa.PlkError as e:
        errs.append((str(e), e.code, time.perf_counter() - t0))
ctx.comm_destroy()
assert setup.prove(circ) == want
print("ERRS", repr(errs))
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    out = [ln for ln in r.stdout.splitlines() if ln.startswith("ERRS")][0]
    errs = eval(out[5:])                                                # noqa: S307 — our own repr of two tuples
    assert errs[0][1] == 4 and "RCCL exchange aborted" in errs[0][0] and "deadline" in errs[0][0], out
    # (the deadline fires after 100 ms; ncclCommAbort then waits for the test's own spin kernel, which RCCL cannot kill — a hung
    #  collective it can — so the call returns when the 1.5 s stall ends: well before the 6 s that four stalled exchanges would take)
    assert errs[0][2] < 3.0, "the first exchange must be abandoned, not waited out four times: %r" % (errs,)
    assert errs[1][1] == 4 and "aborted earlier" in errs[1][0] and errs[1][2] < 0.5, out


# ---------------------------------------------------------------- owner-computes ("scatter") mode, round 5
def _scatter_rank(rank, world, port, log_n, lagrange, q):
    """rank 0 proves; the others hold a slice of the key and serve its commitments (plk_comm_serve)"""
    import numpy as np
    import torch
    import plonkit_amd as pa
    n = 1 << log_n
    local = n // world
    ctx = pa.Context(0)
    ctx.srs_generate(local, rank * local, 42)
    if lagrange:
        full = pa.Context(0)
        full.srs_generate(n, 0, 42)
        keep = torch.zeros((n, 8), dtype=torch.int64, device="cuda:0")
        full.g1_intt_srs_dev(log_n, keep.data_ptr())
        full.synchronize()
        ctx.srs_lagrange_upload(keep.cpu().numpy().view(np.uint64)[rank * local:(rank + 1) * local])
        full.close()
    ctx.comm_init_tcp(rank, world, port, rank * local)
    ctx.comm_set_mode("scatter")
    if rank == 0:
        try:
            circ = pa.Circuit.synthetic(n - 2)
            setup = pa.SetupForProver(ctx, circ)
            vk = setup.verification_key_bytes(pa.crs42_g2_bytes())
            proofs = [setup.prove(circ) for _ in range(2)]
            try:                                                    # commitment-level sharded calls are replicate-mode only: refused, not hung
                ctx.msm_finish_sharded()
                raise AssertionError("plk_msm_g1_finish_sharded must be refused in owner-computes mode")
            except pa.PlkError as e:
                assert "owner-computes" in str(e)
        finally:
            ctx.comm_stop_workers()                                 # (the workers wait without a deadline)
        q.put((0, vk, proofs, ctx.comm_info()[2]))
        setup.close(); circ.close()
    else:
        try:
            served = ctx.comm_serve()
            q.put((rank, None, None, served))
        except Exception as exc:                                    # noqa: BLE001
            q.put((rank, None, repr(exc), -1))
    ctx.close()


@pytest.mark.parametrize("world,log_n,lagrange", [(2, 12, False), (4, 14, True), (8, 13, False)])
def test_scatter_mode_gives_the_single_gpu_proof(world, log_n, lagrange):
    """PLK_SHARD_SCATTER: only rank 0 runs the prover; every batch of commitments sends each other rank its slice of the scalar
    vectors and gets 96 bytes back.  Same verification key and proof bytes as one GPU with the whole key; the workers serve
    2 (key) + 2 x 4 (proofs) batches.  (Ranks share this box's single GPU: TCP transport of the test tier.)"""
    import plonkit_amd as pa
    n = 1 << log_n
    ctx = pa.Context(0)
    ctx.srs_generate(n, 0, 42)
    circ = pa.Circuit.synthetic(n - 2)
    setup = pa.SetupForProver(ctx, circ)
    want_vk, want_proof = setup.verification_key_bytes(pa.crs42_g2_bytes()), setup.prove(circ)
    setup.close(); circ.close(); ctx.close()
    port = _free_port()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_scatter_rank, args=(r, world, port, log_n, lagrange, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    rank0 = res[0]
    assert rank0[1] == want_vk and rank0[2] == [want_proof, want_proof]
    assert rank0[3] == 2 + 2 * 4                                    # exchanges on the owner: the key in two batches, four per proof
    for rank, _, err, served in res[1:]:
        assert err is None and served == 2 + 2 * 4, (rank, err, served)


def test_scatter_mode_through_the_plonkit_binary(tmp_path):
    """PLK_SHARD_MODE=scatter with three `plonkit` processes (cli_main.cpp): rank 0 proves, ranks 1-2 print how many batches they
    served; vk.bin / proof.bin equal the one-process files.  A worker that is asked to prove itself is refused."""
    import subprocess
    import plonkit_amd as pa
    cli = os.path.join(os.path.dirname(pa.lib_path()), "plonkit")
    log_n = 12
    circ = pa.Circuit.synthetic((1 << log_n) - 2)
    f = lambda name: str(tmp_path / name)
    open(f("c.r1cs"), "wb").write(circ.export("r1cs"))
    open(f("w.wtns"), "wb").write(circ.export("wtns"))
    circ.close()
    subprocess.check_call([cli, "setup", "-p", str(log_n), "-m", f("key.bin")], stderr=subprocess.DEVNULL)
    subprocess.check_call([cli, "export-verification-key", "-m", f("key.bin"), "-c", f("c.r1cs"), "-v", f("vk1.bin")], stderr=subprocess.DEVNULL)
    subprocess.check_call([cli, "prove", "-m", f("key.bin"), "-c", f("c.r1cs"), "-w", f("w.wtns"), "-p", f("p1.bin"), "-j", f("j1.json"), "-i", f("i1.json")],
                          stderr=subprocess.DEVNULL)
    world = 4
    for cmd, outs, batches in ((["export-verification-key", "-m", f("key.bin"), "-c", f("c.r1cs"), "-v", f("vk2.bin")], ("vk1.bin", "vk2.bin"), 2),
                               (["prove", "-m", f("key.bin"), "-c", f("c.r1cs"), "-w", f("w.wtns"), "-p", f("p2.bin"), "-j", f("j2.json"), "-i", f("i2.json")], ("p1.bin", "p2.bin"), 4)):
        port = _free_port()
        procs = []
        for rank in range(world):
            env = dict(os.environ, PLONKIT_WORLD=str(world), PLONKIT_RANK=str(rank), PLONKIT_COMM="tcp:%d" % port, PLONKIT_DEVICE="0", PLK_SHARD_MODE="scatter")
            procs.append(subprocess.Popen([cli] + cmd, env=env, stderr=subprocess.PIPE))
        for rank, p in enumerate(procs):
            _, err = p.communicate(timeout=300)
            assert p.returncode == 0, err.decode()[-2000:]
            if rank:
                assert ("served %d batches" % batches) in err.decode()
        assert open(f(outs[0]), "rb").read() == open(f(outs[1]), "rb").read()
    assert pa.verify(open(f("vk2.bin"), "rb").read(), open(f("p2.bin"), "rb").read())
