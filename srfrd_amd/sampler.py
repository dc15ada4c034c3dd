"""Synthetic batches in the reference sampler's layout (reference utils.py:21-57, ``sample_function_fr``).

One batch is the 7-tuple ``(user, seq, rsq, pos, prs, neg, nrs)``: ``user`` (B,), the rest int64 (B, L),
LEFT-padded with 0; ``pos[t] = seq[t+1]`` with the held-out next item in the last column; ``neg[t]`` is a random
item outside the user's own items wherever ``pos[t] != 0``; ``rsq / prs`` hold 1 (fake) / 2 (real) / 0 (pad) and
``nrs`` is 1 wherever set (reference utils.py:52 draws ``randint(1, 2)``).

The generator is counter-seeded (seed, batch index, rank) so every data-parallel rank can regenerate its own shard,
and it runs on the target device so the timed loop never waits for a host sampler.
"""
from __future__ import annotations

import torch


def synthetic_batch(n_items: int, max_len: int, batch: int, *, seed: int = 1, index: int = 0, rank: int = 0,
                    device="cpu", fake_prob: float = 0.3, n_users: int | None = None, min_len: int = 2,
                    packed: bool = False):
    """Uniform item ids over [1, n_items], lengths ~ U[min_len, max_len] (SURVEY 8d workload definition)."""
    g = torch.Generator(device="cpu")
    g.manual_seed((seed * 1_000_003 + index) * 1009 + rank)
    B, L = batch, max_len
    lens = torch.randint(min(min_len, L), L + 1, (B,), generator=g)
    items = torch.randint(1, n_items + 1, (B, L + 1), generator=g)          # L inputs + the final target
    revs = torch.where(torch.rand(B, L + 1, generator=g) < fake_prob, 1, 2)
    negs = torch.randint(1, n_items + 1, (B, L), generator=g)
    col = torch.arange(L).unsqueeze(0)
    valid = col >= (L - lens).unsqueeze(1)                                   # left padding
    seq = torch.where(valid, items[:, :L], 0)
    rsq = torch.where(valid, revs[:, :L], 0)
    pos = torch.where(valid, items[:, 1:], 0)
    prs = torch.where(valid, revs[:, 1:], 0)
    # negatives must avoid the user's own items (utils.py:14-19 random_neq): resample collisions a few rounds
    own = torch.cat([seq, pos[:, -1:]], dim=1)
    for _ in range(8):
        clash = (negs.unsqueeze(2) == own.unsqueeze(1)).any(dim=2)
        if not bool(clash.any()):
            break
        negs = torch.where(clash, torch.randint(1, n_items + 1, (B, L), generator=g), negs)
    neg = torch.where(valid, negs, 0)
    nrs = valid.to(torch.int64)
    user = torch.randint(1, (n_users or B) + 1, (B,), generator=g)
    if packed:
        return user.to(device), torch.stack([seq, rsq, pos, prs, neg, nrs]).to(device=device, dtype=torch.int64)
    return tuple(t.to(device=device, dtype=torch.int64) for t in (user, seq, rsq, pos, prs, neg, nrs))


def eval_candidates(n_items: int, seq: torch.Tensor, target: torch.Tensor, n_neg: int = 100, *, seed: int = 7):
    """(B, 1 + n_neg) candidates per user: the true next item followed by sampled negatives that are neither 0 nor
    in the user's history (reference utils.py:576-583)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    seq_c, tgt = seq.cpu(), target.cpu().view(-1, 1)
    B = seq_c.shape[0]
    negs = torch.randint(1, n_items + 1, (B, n_neg), generator=g)
    for _ in range(8):
        clash = (negs.unsqueeze(2) == seq_c.unsqueeze(1)).any(dim=2)
        if not bool(clash.any()):
            break
        negs = torch.where(clash, torch.randint(1, n_items + 1, (B, n_neg), generator=g), negs)
    return torch.cat([tgt, negs], dim=1).to(device=seq.device, dtype=torch.int64)
