"""Multi-GPU KZG commitment: the SRS bases are split across ranks (one process per GPU), every rank
runs the Pippenger MSM of its shard on its own MI355X, and the 96-byte Jacobian partial sums are
exchanged with ONE all_gather over RCCL/xGMI, then added on the host (EC addition is not an RCCL
reduction op, so a literal all_reduce is impossible — SURVEY.md §8e).  NTTs stay single-GPU.

In the reference there is no counterpart (one process, bellman's Worker threads, src/plonk.rs:41);
the sharded commitment is mathematically the same sum commit_using_monomials computes.
"""
import os

# dmabuf IPC: RCCL between the per-GPU processes fails with `hipIpcGetMemHandle: invalid argument` on this driver unless this is set
# before the first HIP call of the process; this module is the multi-rank entry of the package, so it is the one to ask
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import collections

import numpy as np
import torch

from . import _lib


def combine_partials(partial, dist=None, device=None):
    """partial: uint64[12] Jacobian of this rank -> affine uint64[8] of the sum over all ranks.
    With dist == None (single process) it is just the Jacobian -> affine conversion."""
    partial = np.ascontiguousarray(partial, dtype=np.uint64).reshape(12)
    if dist is None or (dist.get_world_size() == 1 and not os.environ.get("PLK_FORCE_GATHER")):
        return _lib.g1_sum_jacobian(partial)          # (PLK_FORCE_GATHER: exercise the collective with one rank)
    world = dist.get_world_size()
    t = torch.from_numpy(partial.view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    out = torch.empty((world, 12), dtype=torch.int64, device=t.device)
    dist.all_gather_into_tensor(out, t) if hasattr(dist, "all_gather_into_tensor") and t.device.type != "cpu" else \
        dist.all_gather(list(out.unbind(0)), t)
    parts = out.cpu().numpy().view(np.uint64)
    return _lib.g1_sum_jacobian(parts)


class ShardedMsm:
    """commit(scalars) = sum over ranks of MSM(local scalars, local SRS shard)."""

    def __init__(self, ctx, dist=None, device=None, native=False):
        """native=True: the context carries the library's own communicator (Context.comm_init: RCCL all-gather + EC sum in
        C++, one stream synchronisation per commitment) — no torch.distributed call, no Python in the exchange"""
        self.ctx, self.dist, self.device, self.native = ctx, dist, device, native

    def _finish(self):
        if self.native:
            return self.ctx.msm_finish_sharded()
        return combine_partials(self.ctx.msm_finish(), self.dist, self.device)

    def commit(self, scalars_dev, n, base_offset=0, stream=None):
        self.ctx.msm_enqueue_dev(scalars_dev, n, base_offset, stream=stream)
        return self._finish()

    def commit_stream(self, batches, n, base_offset=0, stream=None, depth=3):
        """Generator over a sequence of scalar vectors.  The library keeps up to three commitments in flight (three
        scratch sets, three streams): k+1 and k+2 are enqueued before k is finished, so the latency-bound bucket
        reduction of k, its host Horner and its exchange (all_gather + host EC sum) overlap the accumulation of k+1, and
        the digit / partition kernels of k+2 are done by the time that accumulation ends — the overlap SURVEY.md §8(e)
        asks for.  depth = 2 or 1 keeps fewer in flight.  The scalars of a commitment must stay untouched until it has
        been yielded (the generator holds a reference to them until then)."""
        it = iter(batches)
        held = collections.deque()
        exhausted = False
        while True:
            while not exhausted and len(held) < depth:
                cur = next(it, None)
                if cur is None:
                    exhausted = True
                    break
                self.ctx.msm_enqueue_dev(cur, n, base_offset, stream=stream)
                held.append(cur)
            if not held:
                return
            out = self._finish()                               # waits for the oldest commitment, host Horner over the windows, exchange
            held.popleft()
            yield out

    def commit_batches(self, batches, n, base_offset=0, stream=None, depth=3):
        """the same pipeline with a BATCH of vectors per slot (the prover's shape: 4 wire / 4 quotient commitments share one
        pass of the kernels): `batches` yields lists of up to 8 scalar vectors; yields [count, 8] affine commitments per
        batch, up to `depth` batches in flight.  Needs native=True when ranks > 1 (one exchange per batch inside the library)."""
        it = iter(batches)
        held = collections.deque()
        exhausted = False
        while True:
            while not exhausted and len(held) < depth:
                cur = next(it, None)
                if cur is None:
                    exhausted = True
                    break
                self.ctx.msm_enqueue_batch_dev(cur, n, base_offset, stream=stream)
                held.append(cur)
            if not held:
                return
            cur = held[0]
            if self.native:
                out = self.ctx.msm_finish_batch_sharded(len(cur))
            else:
                out = np.stack([combine_partials(p, self.dist, self.device) for p in self.ctx.msm_finish_batch(len(cur))])
            held.popleft()
            yield out


# Montgomery form of 1 in Fq (R mod q): the Z coordinate of an affine point written back as Jacobian
_FQ_ONE = np.array([0xd35d438dc58f0d9d, 0x0a78eb28f5c70b3d, 0x666ea36f7879462c, 0x0e0a77c19a07df2f], dtype=np.uint64)


class ShardedProver:
    """Multi-GPU prove: one process per GPU, the SRS split into contiguous slices, NTTs and
    point-wise work replicated, every commitment of plk_prove / make_verification_key computed as the sum over ranks of
    MSM(slice of the scalars, slice of the SRS).  Installs the combiner of plk_set_commit_shard: one all_gather of
    96 bytes per commitment (a batch of up to 8 per call) and a host EC sum.  Every rank ends with the same proof bytes
    as a single GPU would produce.

        sp = ShardedProver(ctx, dist, device, n_total)      # ctx holds SRS points [first, first + local) only
        setup = SetupForProver(ctx, circuit); proof = setup.prove(circuit)
    """

    def __init__(self, ctx, dist, device=None, first_index=None, n_local=None):
        self.ctx, self.dist, self.device = ctx, dist, device
        rank = dist.get_rank() if dist is not None else 0
        n_local = ctx.srs_size() if n_local is None else n_local
        self.first = rank * n_local if first_index is None else first_index
        ctx.set_commit_shard(self.first, self._combine)

    def close(self):
        self.ctx.set_commit_shard(0, None)

    def _combine(self, sums):
        """sums: uint64[count, 12] (this rank's partial sums) -> overwritten with the sums over all ranks"""
        count = sums.shape[0]
        if self.dist is None or (self.dist.get_world_size() == 1 and not os.environ.get("PLK_FORCE_GATHER")):
            parts = sums.reshape(1, count, 12).copy()
        else:
            world = self.dist.get_world_size()
            t = torch.from_numpy(sums.view(np.int64).copy())
            if self.device is not None:
                t = t.to(self.device)
            out = torch.empty((world, count, 12), dtype=torch.int64, device=t.device)
            if t.device.type != "cpu" and hasattr(self.dist, "all_gather_into_tensor"):
                self.dist.all_gather_into_tensor(out, t)
            else:
                self.dist.all_gather(list(out.unbind(0)), t)
            parts = out.cpu().numpy().view(np.uint64)
        for k in range(count):
            a = _lib.g1_sum_jacobian(np.ascontiguousarray(parts[:, k, :]))
            if not a.any():
                sums[k, :] = 0                                         # infinity: Z = 0
            else:
                sums[k, 0:8] = a
                sums[k, 8:12] = _FQ_ONE
