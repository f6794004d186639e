"""world_size-2 `gloo` test (CPU) of the multi-GPU commitment's exchange step: each rank holds the
Jacobian partial sum of its SRS shard (here computed by the oracle, standing in for the rank's GPU),
ranks all_gather the 96-byte partials and add them on the host with the product's plk_g1_sum_jacobian.
The result must equal the single-process MSM over the whole SRS.  (On the GPU box the same
combine_partials runs over RCCL; the driver launches it through bench.py --gpus N.)"""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle_lib as ol
from oracle.oracle_lib import R_MOD


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_per_rank, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from plonkit_amd.sharded import combine_partials
        srs = ol.crs42(world * n_per_rank, threads=2)
        rng = np.random.default_rng(99)                       # same scalars on every rank
        s = rng.integers(0, 1 << 62, size=(world * n_per_rank, 4), dtype=np.uint64)
        s[:, 3] &= np.uint64((1 << 60) - 1)
        lo, hi = rank * n_per_rank, (rank + 1) * n_per_rank
        partial = ol.msm_jacobian(srs[lo:hi], s[lo:hi], threads=2)    # this rank's shard
        total = combine_partials(partial, dist, None)
        full = ol.msm(srs, s, threads=2)
        q.put((rank, bool(np.array_equal(total, full))))
    finally:
        dist.destroy_process_group()


def test_sharded_commit_combination_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 300, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def test_combine_single_process_is_affine_conversion():
    from plonkit_amd.sharded import combine_partials
    srs = ol.crs42(64, threads=1)
    s = ol.fr_vec([(i * 7919 + 13) % R_MOD for i in range(64)])
    assert np.array_equal(combine_partials(ol.msm_jacobian(srs, s, threads=1)), ol.msm(srs, s, threads=1))


class _OracleShard:
    """stands in for the rank's plk_ctx (no GPU in this test): msm_enqueue_dev / msm_finish over this rank's
    SRS shard, computed by the oracle; records the call order so that the pipelining can be checked"""
    def __init__(self, bases):
        self.bases, self.pending, self.log = bases, [], []

    def msm_enqueue_dev(self, scalars, n, base_offset=0, stream=None):
        self.log.append("enqueue")
        self.pending.append(ol.msm_jacobian(self.bases[base_offset:base_offset + n], scalars[:n], threads=2))

    def msm_finish(self):
        self.log.append("finish")
        return self.pending.pop(0)


def _stream_worker(rank, world, port, n_per_rank, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from plonkit_amd.sharded import ShardedMsm
        srs = ol.crs42(world * n_per_rank, threads=2)
        lo, hi = rank * n_per_rank, (rank + 1) * n_per_rank
        vecs = []
        for k in range(3):
            rng = np.random.default_rng(7 + k)                # same scalars on every rank
            s = rng.integers(0, 1 << 62, size=(world * n_per_rank, 4), dtype=np.uint64)
            s[:, 3] &= np.uint64((1 << 60) - 1)
            vecs.append(s)
        shard = _OracleShard(srs[lo:hi])
        msm = ShardedMsm(shard, dist, None)
        got = list(msm.commit_stream((v[lo:hi] for v in vecs), n_per_rank, depth=2))
        ok = all(np.array_equal(g, ol.msm(srs, v, threads=2)) for g, v in zip(got, vecs))
        # two in flight: commitment k+1 is enqueued before commitment k is finished and exchanged
        ok = ok and shard.log == ["enqueue", "enqueue", "finish", "enqueue", "finish", "finish"]
        # the default keeps three in flight (the library's FIFO depth)
        shard.log.clear()
        got3 = list(msm.commit_stream((v[lo:hi] for v in vecs + vecs[:1]), n_per_rank))
        ok = ok and len(got3) == 4 and all(np.array_equal(a, b) for a, b in zip(got3, got + got[:1]))
        ok = ok and shard.log == ["enqueue"] * 3 + ["finish", "enqueue"] + ["finish"] * 3
        q.put((rank, bool(ok), len(got)))
    finally:
        dist.destroy_process_group()


def test_sharded_commit_stream_world2():
    """ShardedMsm.commit_stream (what bench.py --gpus N times): three commitments in a row, every one equal to the
    single-process MSM over the whole SRS"""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_stream_worker, args=(r, world, port, 200, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True, 3), (1, True, 3)]


# ------------------------------------------------------------------ the combiner below the C ABI (comm.cpp)
def _native_worker(rank, world, port, n_per_rank, q):
    """the library's own combiner (plk_comm_combine over the TCP transport; on a GPU node the same function runs over an
    RCCL all-gather): no torch.distributed, no Python in the exchange"""
    import ctypes
    import plonkit_amd as pa
    L = pa.lib()
    comm = ctypes.c_void_p()
    assert L.plk_comm_open_tcp(ctypes.c_int32(rank), ctypes.c_int32(world), ctypes.c_uint16(port), ctypes.byref(comm)) == 0, pa.last_error()
    try:
        srs = ol.crs42(world * n_per_rank, threads=2)
        rng = np.random.default_rng(7)
        count = 3                                                 # a batch of three commitments, the third one empty on rank 1
        sums = np.zeros((count, 12), dtype=np.uint64)
        want = []
        for k in range(count):
            s = rng.integers(0, 1 << 62, size=(world * n_per_rank, 4), dtype=np.uint64)
            s[:, 3] &= np.uint64((1 << 60) - 1)
            if k == 2:
                s[n_per_rank:] = 0                                # rank 1's share of this commitment is the point at infinity
            lo, hi = rank * n_per_rank, (rank + 1) * n_per_rank
            sums[k] = ol.msm_jacobian(srs[lo:hi], s[lo:hi], threads=2)
            want.append(ol.msm(srs, s, threads=2))
        assert L.plk_comm_combine(comm, sums.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(count)) == 0, pa.last_error()
        ok = all(np.array_equal(ol.jac_to_affine(sums[k])[0], want[k]) for k in range(count))
        q.put((rank, ok, sums.tobytes()))
    finally:
        L.plk_comm_close(comm)


def test_native_combiner_world2_world3_and_world8():
    """world 8 = the node size of BASELINE.json configs[2]: eight processes, one hub, every rank ends with the same sums"""
    for world in (2, 3, 8):
        port = _free_port()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_native_worker, args=(r, world, port, 200 if world < 8 else 64, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=240) for _ in range(world))
        for p in procs:
            p.join(timeout=60)
        assert [r[:2] for r in res] == [(r, True) for r in range(world)]
        assert len({r[2] for r in res}) == 1                      # every rank ends with the same bytes


# ------------------------------------------------------------------ owner-computes mode: the scatter step on host buffers (comm.cpp)
def _scatter_worker(rank, world, port, n, slice_, q):
    """rank 0 scatters three batches (2 vectors of n elements, 1 vector, 8 short vectors that leave the last ranks empty-handed) and
    stops; every other rank must receive exactly its share of every vector; a combine after every batch keeps the exchange in step"""
    import ctypes
    import plonkit_amd as pa
    L = pa.lib()
    comm = ctypes.c_void_p()
    assert L.plk_comm_open_tcp(ctypes.c_int32(rank), ctypes.c_int32(world), ctypes.c_uint16(port), ctypes.byref(comm)) == 0, pa.last_error()
    try:
        rng = np.random.default_rng(2025)                          # the same data on every rank: workers know what to expect
        batches = [(2, n), (1, n), (8, slice_ + slice_ // 2)]
        ok = True
        for count, length in batches:
            vecs = [rng.integers(0, 1 << 63, size=(length, 4), dtype=np.uint64) for _ in range(count)]
            cnt, ln = ctypes.c_uint32(0), ctypes.c_uint64(0)
            if rank == 0:
                arr = (ctypes.c_void_p * count)(*[v.ctypes.data for v in vecs])
                assert L.plk_comm_scatter_host(comm, arr, ctypes.c_uint32(count), ctypes.c_uint64(length), ctypes.c_uint64(slice_), None, ctypes.c_uint64(0),
                                               ctypes.byref(cnt), ctypes.byref(ln)) == 0, pa.last_error()
            else:
                mine = np.zeros((8 * slice_, 4), dtype=np.uint64)
                assert L.plk_comm_scatter_host(comm, None, ctypes.c_uint32(0), ctypes.c_uint64(0), ctypes.c_uint64(0), mine.ctypes.data_as(ctypes.c_void_p),
                                               ctypes.c_uint64(mine.nbytes), ctypes.byref(cnt), ctypes.byref(ln)) == 0, pa.last_error()
                lo = min(rank * slice_, length); hi = min((rank + 1) * slice_, length)
                ok &= cnt.value == count and ln.value == hi - lo
                for k in range(count):
                    ok &= bool(np.array_equal(mine[k * (hi - lo):(k + 1) * (hi - lo)], vecs[k][lo:hi]))
            sums = np.zeros((1, 12), dtype=np.uint64)               # (infinity from everyone: only the framing matters here)
            assert L.plk_comm_combine(comm, sums.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(1)) == 0, pa.last_error()
        cnt, ln = ctypes.c_uint32(9), ctypes.c_uint64(9)
        if rank == 0:
            assert L.plk_comm_scatter_host(comm, None, ctypes.c_uint32(0), ctypes.c_uint64(0), ctypes.c_uint64(0), None, ctypes.c_uint64(0), None, None) == 0
        else:
            assert L.plk_comm_scatter_host(comm, None, ctypes.c_uint32(0), ctypes.c_uint64(0), ctypes.c_uint64(0), None, ctypes.c_uint64(0),
                                           ctypes.byref(cnt), ctypes.byref(ln)) == 0, pa.last_error()
            ok &= cnt.value == 0 and ln.value == 0                  # the stop message
        q.put((rank, bool(ok)))
    finally:
        L.plk_comm_close(comm)


def test_scatter_step_world2_world3_and_world8():
    """PLK_SHARD_SCATTER's transport without a GPU: header, shares (incl. ragged and empty ones), sequence numbers, stop"""
    for world in (2, 3, 8):
        port = _free_port()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        n = 1 << 10
        slice_ = (n + world - 1) // world                          # world 3: the last share is shorter
        procs = [ctx.Process(target=_scatter_worker, args=(r, world, port, n, slice_, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=240) for _ in range(world))
        for p in procs:
            p.join(timeout=60)
        assert res == [(r, True) for r in range(world)]


def _idle_worker(rank, world, port, q):
    """the owner opens the communicator and then never sends a batch (it "failed" before plk_comm_stop_workers)"""
    import ctypes
    import time
    import plonkit_amd as pa
    if rank != 0:
        os.environ["PLK_COMM_IDLE_TIMEOUT_MS"] = "400"
    L = pa.lib()
    comm = ctypes.c_void_p()
    assert L.plk_comm_open_tcp(ctypes.c_int32(rank), ctypes.c_int32(world), ctypes.c_uint16(port), ctypes.byref(comm)) == 0, pa.last_error()
    try:
        if rank == 0:
            time.sleep(3.0)                                        # alive, connected, silent
            q.put((rank, 0, 0.0, ""))
        else:
            cnt, ln = ctypes.c_uint32(0), ctypes.c_uint64(0)
            t0 = time.perf_counter()
            rc = L.plk_comm_scatter_host(comm, None, ctypes.c_uint32(0), ctypes.c_uint64(0), ctypes.c_uint64(0), None, ctypes.c_uint64(0),
                                         ctypes.byref(cnt), ctypes.byref(ln))
            q.put((rank, rc, time.perf_counter() - t0, pa.last_error()))
    finally:
        L.plk_comm_close(comm)


def test_worker_idle_deadline():
    """PLK_COMM_IDLE_TIMEOUT_MS: a worker of owner-computes mode whose owner stays silent gives up with an error instead of waiting
    for ever (unset, the wait has no deadline: waiting for work is not a fault)"""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_idle_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    rank, rc, waited, why = res[1]
    assert rank == 1 and rc != 0 and 0.3 <= waited < 2.5 and "owner" in why, res
