"""Batched HR@10 / NDCG@10 evaluation: the metric of reference utils.py:544-602 (``evaluation``), computed for a
whole batch of users per launch instead of one user (and one host sync) at a time.

For each user: candidates = [held-out next item] + 100 sampled negatives, scores = predict(...), rank = position of
the held-out item = number of candidates scoring strictly higher (``argsort().argsort()[0]`` in the reference),
HR@10 += rank < 10, NDCG@10 += 1 / log2(rank + 2).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib, ops  # noqa: F401  (ops: registers torch.ops.srfrd.*)
from ._lib import check, ptr


def ranks_from_logits(logits: torch.Tensor, metric_acc: torch.Tensor | None = None) -> torch.Tensor:
    """rank of candidate 0 per row of (B, n_cand) logits; optionally accumulates [ndcg_sum, hit_sum, users] (fp64)."""
    if logits.device.type != "cuda":
        raise RuntimeError("ranks_from_logits runs on the ROCm GPU only")
    if metric_acc is None:
        return torch.ops.srfrd.eval_rank(logits)
    logits = logits.contiguous()
    B, n = logits.shape
    rank = torch.empty(B, device=logits.device, dtype=torch.int32)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(_lib.lib().srfrd_eval_rank(ptr(logits), B, n, ptr(rank), ptr(metric_acc), st), "srfrd_eval_rank")
    return rank


@torch.no_grad()
def evaluate_batches(model, batches):
    """batches: iterable of (user_ids, seq (B,L), rsq (B,L), candidates (B, 1 + n_neg)).  -> (NDCG@10, HR@10)."""
    was_training = model.training
    model.eval()
    acc = None
    for user_ids, seq, rsq, cand in batches:
        logits = model.predict(user_ids, seq, rsq, cand)
        if logits.dim() == 1:
            logits = logits.unsqueeze(0)
        if acc is None:
            acc = torch.zeros(3, device=logits.device, dtype=torch.float64)
        ranks_from_logits(logits, acc)
    model.train(was_training)
    if hasattr(model, "check_ids"):
        model.check_ids()
    a = acc.cpu()
    return float(a[0] / a[2]), float(a[1] / a[2])
