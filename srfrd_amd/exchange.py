"""Data-parallel gradient exchange of the fused train step (no reference counterpart: the reference trains on one device,
trainer.py:117; SURVEY.md 8e defines the path).

One process per GPU, the global batch split by sequence, parameters replicated.  Per step every rank holds SUM gradients
of its sequences in one flat vector ``[item table | pad | dense]`` and four loss statistics (sum softplus(-pos), sum
softplus(neg), non-pad target count, 0).  Two forms of the exchange, both reproducing the single-process update
(mean over the GLOBAL count of non-pad targets, trainer.py:36-38):

``sharded`` (default)   reduce-scatter of the flat gradient -> every rank runs Adam on its contiguous 1/N slice only (optimizer
                        moments exist only for that slice: Adam's HBM traffic and state memory divide by N) -> all-gather of
                        the stepped parameters.  Same wire volume as an all-reduce (which is a reduce-scatter + all-gather
                        inside RCCL), 1/N of the optimizer traffic: the lever SURVEY 8e prices as necessary at C4 / C5,
                        where the dense Adam pass is 70-90 % of a step's bytes.
``allreduce``           one SUM all-reduce of ``[flat gradient | statistics]``, then the full Adam on every rank.

The statistics travel in their own 16-byte all-reduce, started right after the forward (they are forward outputs) so it
overlaps the backward kernel.  Collectives are RCCL (torch.distributed backend "nccl") over xGMI on GPUs.  The ``gloo``
backend - CPU tests here, or a several-ranks-on-one-GPU rehearsal - lacks reduce-scatter / all-gather for device
tensors, so the same shard arithmetic runs on all-reduce + per-shard broadcasts there.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def exchange_forced() -> bool:
    """SRFRD_FORCE_EXCHANGE=1: take the data-parallel code paths (collectives, shard arithmetic, split step) in a process
    group of ONE rank too - how the RCCL arms are exercised on a one-GPU box (tests/test_gpu_nccl_single.py)."""
    return os.environ.get("SRFRD_FORCE_EXCHANGE", "") == "1"


def shard_bounds(n: int, world: int, rank: int, align: int = 4):
    """[i0, i1) of this rank's contiguous slice of an n-element vector, i0 aligned for float4 access."""
    per = (n + world - 1) // world
    per = (per + align - 1) // align * align
    i0 = min(rank * per, n)
    return i0, min(i0 + per, n)


def shard_size(n: int, world: int, align: int = 4) -> int:
    """elements per rank when an n-element vector is padded to world equal, `align`-aligned shards."""
    per = (n + world - 1) // world
    return (per + align - 1) // align * align


class GradExchange:
    """Shard arithmetic + collectives for one flat vector of ``n`` floats (padded to ``n_pad = per * world``)."""

    def __init__(self, n: int, group=None):
        self.group = group
        self.on = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or exchange_forced())
        self.world = dist.get_world_size(group) if self.on else 1
        self.rank = dist.get_rank(group) if self.on else 0
        self.n = int(n)
        self.per = shard_size(self.n, self.world)
        self.n_pad = self.per * self.world
        self.i0, self.i1 = self.rank * self.per, (self.rank + 1) * self.per
        self.backend = dist.get_backend(group) if self.on else "none"
        self.native = self.backend == "nccl"          # RCCL: reduce_scatter_tensor / all_gather_into_tensor on device

    # ---- statistics (16 bytes): async, overlapped with the backward ---------------------------------------------
    def all_reduce_stats(self, stats: torch.Tensor):
        """SUM over ranks, in place; returns a work handle (``.wait()``) or None on a single rank."""
        if not self.on:
            return None
        return dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    # ---- all-reduce form ----------------------------------------------------------------------------------------
    def all_reduce(self, flat: torch.Tensor):
        if self.on:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        return flat

    # ---- sharded form -------------------------------------------------------------------------------------------
    def reduce_scatter(self, grad_padded: torch.Tensor, out_shard: torch.Tensor) -> torch.Tensor:
        """out_shard (per,) <- sum over ranks of grad_padded[i0:i1]; grad_padded has n_pad elements."""
        assert grad_padded.numel() == self.n_pad and out_shard.numel() == self.per
        if not self.on:
            out_shard.copy_(grad_padded)
        elif self.native:
            dist.reduce_scatter_tensor(out_shard, grad_padded, op=dist.ReduceOp.SUM, group=self.group)
        else:                                          # gloo: no reduce-scatter -> all-reduce, keep the own slice
            dist.all_reduce(grad_padded, op=dist.ReduceOp.SUM, group=self.group)
            out_shard.copy_(grad_padded[self.i0:self.i1])
        return out_shard

    def all_gather(self, flat_padded: torch.Tensor) -> torch.Tensor:
        """every rank's stepped slice flat_padded[i0:i1] -> all of flat_padded (n_pad elements), in place."""
        assert flat_padded.numel() == self.n_pad
        if not self.on:
            return flat_padded
        if self.native:
            dist.all_gather_into_tensor(flat_padded, flat_padded[self.i0:self.i1], group=self.group)
        else:                                          # gloo: one broadcast per shard
            ranks = dist.get_process_group_ranks(self.group) if self.group is not None else list(range(self.world))
            for r in range(self.world):
                dist.broadcast(flat_padded[r * self.per:(r + 1) * self.per], src=ranks[r], group=self.group)
        return flat_padded
