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
