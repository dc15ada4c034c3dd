"""Row-sharded full-catalog ranking (BASELINE configs[4]: 1M items, row-sharded embedding table).

``predict`` over the whole catalog (reference SRFR_model.py:144-152 / :241-259 / :532-540 / :668-681 with every item as a
candidate) is the one part of the path whose cost grows with the catalog: 2 B I d flops, and a (B, I) logits matrix the
reference would materialise.  Here the item rows are split into contiguous shards; each shard is ranked by
``srfrd_logits_topk`` over its own ``[lo, hi)`` (logits never reach HBM) and the per-shard top-k lists are merged by
``srfrd_topk_merge`` into the order one unsharded ranking returns (value desc, item id asc - ties across shard boundaries
included).

* one process (``n_shards`` given, no process group): the shards are ranked one after the other on this GPU - bounds the
  ranking workspace at 1M+ items and is what the single-GPU tests exercise;
* data parallel (one process per GPU): every rank owns shard ``rank`` of the rows.  The last-position hidden states of ALL
  ranks' users are all-gathered (B x d_out floats per rank: tiny), every rank ranks all users against its rows only, the
  (users, k) lists are all-gathered and each rank merges the lists of its own users - SURVEY 8e's
  "all-gather h_last -> per-shard top-k -> merge".  Only rows ``[lo, hi)`` of the table are read on a rank, so the same
  code serves a table whose other rows are not resident.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.distributed as dist

from . import _lib
from ._lib import check, ptr


def row_shards(n_rows: int, n_shards: int):
    """contiguous [lo, hi) row ranges, sizes differing by at most one"""
    base, extra = divmod(n_rows, n_shards)
    out, lo = [], 0
    for s in range(n_shards):
        hi = lo + base + (1 if s < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def topk_merge(cand_idx: torch.Tensor, cand_val: torch.Tensor, k: int):
    """(B, n_cand) candidate lists -> (idx int64 (B,k), val (B,k)) in stable descending order (srfrd::topk_merge)."""
    if cand_idx.device.type != "cuda":
        raise RuntimeError("topk_merge runs on the ROCm GPU only")
    idx, val = torch.ops.srfrd.topk_merge(cand_idx, cand_val, k)
    return idx, val


class ShardedRanker:
    def __init__(self, model, n_shards: int | None = None, process_group=None):
        self.model, self.group = model, process_group
        from .exchange import exchange_forced
        self.dist_on = dist.is_available() and dist.is_initialized() and (dist.get_world_size(process_group) > 1 or exchange_forced())
        self.world = dist.get_world_size(process_group) if self.dist_on else 1
        self.rank = dist.get_rank(process_group) if self.dist_on else 0
        self.n_shards = self.world if self.dist_on else int(n_shards or 1)
        if self.dist_on and n_shards not in (None, self.world):
            raise ValueError("with a process group the number of shards is the world size")
        self.shards = row_shards(model.layout.n_items + 1, self.n_shards)

    def _rank_shard(self, h_last, ulab, lo, hi, k, exclude_pad):
        from . import ops
        idx, val = torch.ops.srfrd.logits_topk(h_last.unsqueeze(1), ulab, ops.register_model(self.model), lo, hi, k, bool(exclude_pad))
        return idx, val

    @torch.no_grad()
    def topk(self, user_ids, input_ids, fake_ids, k: int = 10, exclude_pad: bool = True, check_batch: bool = True):
        """-> (indices int64 (B,k), scores (B,k)) of this rank's users over the WHOLE catalog.  An index is -1 (score -inf) where
        fewer than k items are rankable.  ``check_batch`` (data parallel): verify that every rank passed the same batch size
        (one 8-byte all-gather + host read per call; pass False in a loop whose batches are known to be equal)."""
        m = self.model
        ids = m._prep(input_ids, fake_ids, None, None, None, None)
        h_last = m._launch_fwd_last(ids[0], ids[1])[:, 0, :]          # (B, d_out)
        ulab = m.user_labels(ids[1]) if m._kind == "SRFRN" else None
        B = h_last.shape[0]
        if not self.dist_on:
            lists = [self._rank_shard(h_last, ulab, lo, hi, k, exclude_pad) for lo, hi in self.shards if hi > lo]
            return topk_merge(torch.cat([i for i, _ in lists], 1), torch.cat([v for _, v in lists], 1), k)
        # ---- one shard per rank: gather every rank's users, rank them against the own rows, exchange the lists
        if check_batch:
            # every rank must bring the same number of users (the gathers below are sized world x B from the LOCAL B: a ragged
            # last evaluation batch would hang or mis-assign rows) - one tiny all-gather, then a clear error on every rank
            nb = torch.tensor([B], device=h_last.device, dtype=torch.int64)
            nb_all = torch.empty(self.world, device=h_last.device, dtype=torch.int64)
            _all_gather(nb_all, nb, self.group)
            sizes = nb_all.tolist()
            if any(x != B for x in sizes):
                raise ValueError(f"ShardedRanker.topk: ranks passed different batch sizes {sizes}; pad the last batch to a common size "
                                 "(rows of padding ids rank like any other user and can be dropped afterwards)")
        h_all = torch.empty(self.world * B, h_last.shape[1], device=h_last.device, dtype=torch.float32)
        _all_gather(h_all, h_last, self.group)
        lab_all = None
        if ulab is not None:
            lab_all = torch.empty(self.world * B, device=h_last.device, dtype=torch.int64)
            _all_gather(lab_all, ulab, self.group)
        lo, hi = self.shards[self.rank]
        idx, val = self._rank_shard(h_all, lab_all, lo, hi, k, exclude_pad)        # (world * B, k) against the own rows
        idx_all = torch.empty(self.world, self.world * B, k, device=idx.device, dtype=torch.int64)
        val_all = torch.empty(self.world, self.world * B, k, device=idx.device, dtype=torch.float32)
        _all_gather(idx_all, idx, self.group)
        _all_gather(val_all, val, self.group)
        mine = slice(self.rank * B, (self.rank + 1) * B)
        ci = idx_all[:, mine].permute(1, 0, 2).reshape(B, self.world * k)
        cv = val_all[:, mine].permute(1, 0, 2).reshape(B, self.world * k)
        return topk_merge(ci, cv, k)


def _all_gather(out: torch.Tensor, part: torch.Tensor, group):
    """out (world * n ...) <- concatenation of every rank's `part`; RCCL natively, through the host on gloo"""
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(out, part.contiguous(), group=group)
    else:
        world = dist.get_world_size(group)
        parts = [torch.empty_like(part, device="cpu") for _ in range(world)]
        dist.all_gather(parts, part.detach().cpu().contiguous(), group=group)
        out.copy_(torch.cat([p.reshape(-1) for p in parts]).view_as(out))
