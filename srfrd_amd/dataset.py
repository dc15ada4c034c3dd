"""Dataset side of the hot path: interaction-table ingest with the semantics of the reference's
``df_data_partition`` (utils.py:92-139), the evaluation-input construction of ``evaluation`` /
``evaluation_with_label`` (utils.py:544-602, 628-752) done for all users at once, and a device-side batch sampler
with the layout of ``sample_function_fr`` (utils.py:21-57).

Histories are stored as CSR over user ids 0..usernum (user 0 never occurs; ids are 1-based like the reference's),
so every consumer is a vectorised numpy / device operation instead of the reference's per-row Python loops.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr


@dataclass
class InteractionData:
    usernum: int
    itemnum: int
    train_ptr: np.ndarray       # int64 (usernum + 2): user u's training rows are [train_ptr[u], train_ptr[u + 1])
    train_items: np.ndarray     # int32
    train_reviews: np.ndarray   # int32, 1 = fake, 2 = real
    test_item: np.ndarray       # int32 (usernum + 1): held-out item, 0 if the user has none
    test_review: np.ndarray     # int32 (usernum + 1)

    def train_len(self):
        return np.diff(self.train_ptr)[: self.usernum + 1]

    def to_device(self, device):
        return (torch.from_numpy(self.train_ptr).to(device), torch.from_numpy(self.train_items).to(device),
                torch.from_numpy(self.train_reviews).to(device))


def partition(user_ids, item_ids, is_fake, is_valid: bool = False) -> InteractionData:
    """``df_data_partition`` on column arrays (rows in file order): per-user chronological lists, fake -> 1 / real -> 2,
    leave-last-out (``is_valid``: leave the second-to-last out and drop the last, utils.py:103-105, 130-135); users
    with fewer than two interactions keep everything in train and have no test item (utils.py:124-128)."""
    u = np.asarray(user_ids, dtype=np.int64)
    it = np.asarray(item_ids, dtype=np.int64)
    rv = np.where(np.asarray(is_fake, dtype=bool), 1, 2).astype(np.int32)
    usernum = int(u.max()) if u.size else 0
    itemnum = int(it.max()) if it.size else 0
    order = np.argsort(u, kind="stable")                 # group by user, file order preserved inside a group
    us, its, rvs = u[order], it[order].astype(np.int32), rv[order]
    counts = np.bincount(us, minlength=usernum + 2)[: usernum + 2]
    start = np.zeros(usernum + 3, dtype=np.int64)
    np.cumsum(counts, out=start[1:])
    final = 2 if is_valid else 1                         # rows dropped from the tail of users with >= 2 rows
    n = counts[: usernum + 1]
    has_test = n >= 2
    keep = np.where(has_test, n - final, n)              # is_valid with n == 2 keeps 0 rows (seq[:-2])
    keep = np.maximum(keep, 0)
    train_ptr = np.zeros(usernum + 2, dtype=np.int64)
    np.cumsum(keep, out=train_ptr[1:])
    # gather the kept prefix of every user
    pos_in_user = np.arange(us.size, dtype=np.int64) - start[us]
    sel = pos_in_user < keep[us]
    test_item = np.zeros(usernum + 1, dtype=np.int32)
    test_review = np.zeros(usernum + 1, dtype=np.int32)
    tu = np.nonzero(has_test)[0]
    tidx = start[tu] + n[tu] - final
    test_item[tu], test_review[tu] = its[tidx], rvs[tidx]
    return InteractionData(usernum, itemnum, train_ptr, its[sel].copy(), rvs[sel].copy(), test_item, test_review)


def load_csv(path: str, is_valid: bool = False) -> InteractionData:
    """CSV with columns ``user_id,item_id,fake_review`` ('fake' marks a fake review; reference trainer.py:145-147)."""
    import pandas as pd
    df = pd.read_csv(path)
    return partition(df["user_id"].to_numpy(), df["item_id"].to_numpy(), (df["fake_review"] == "fake").to_numpy(), is_valid)


def eval_inputs(data: InteractionData, maxlen: int, n_neg: int = 100, seed: int = 0, users=None, candidates=None):
    """For every user with a non-empty train list and a test item (utils.py:558-583): left-padded most recent ``maxlen``
    train items / review ids, and candidates [test item] + ``n_neg`` uniform items outside ``set(train) | {0}``.
    ``candidates``: an (U, 1 + n_neg) integer array to use instead of drawing negatives (row r belongs to the r-th
    evaluated user, column 0 must be that user's test item) - replays a recorded evaluation exactly.
    -> (user ids (U,), seq (U,L), rsq (U,L), cand (U, 1+n_neg)) int64 CPU tensors."""
    lens = data.train_len()
    if users is None:
        users = np.arange(1, data.usernum + 1)
    users = np.asarray(users, dtype=np.int64)
    users = users[(lens[users] >= 1) & (data.test_item[users] != 0)]
    U, L = users.size, maxlen
    seq = np.zeros((U, L), np.int64)
    rsq = np.zeros((U, L), np.int64)
    ends = data.train_ptr[users + 1]
    take = np.minimum(lens[users], L)
    col = np.arange(L)[None, :]
    src = ends[:, None] - L + col                            # last L rows of each user, right-aligned
    ok = col >= (L - take)[:, None]
    src = np.where(ok, src, 0)
    seq[ok] = data.train_items[src[ok]]
    rsq[ok] = data.train_reviews[src[ok]]
    if candidates is not None:
        cand = np.asarray(candidates, dtype=np.int64)
        if cand.ndim != 2 or cand.shape[0] != U or not (cand[:, 0] == data.test_item[users]).all():
            raise ValueError("candidates must be (evaluated users, 1 + n_neg) with the held-out item in column 0")
        return (torch.from_numpy(users), torch.from_numpy(seq), torch.from_numpy(rsq), torch.from_numpy(cand.copy()))
    rng = np.random.RandomState(seed)
    cand = np.zeros((U, 1 + n_neg), np.int64)
    cand[:, 0] = data.test_item[users]
    neg = rng.randint(1, data.itemnum + 1, size=(U, n_neg))
    for r in range(U):                                       # exact exclusion against the FULL train set (utils.py:576-583)
        rated = data.train_items[data.train_ptr[users[r]]:data.train_ptr[users[r] + 1]]
        bad = np.isin(neg[r], rated)
        while bad.any():
            neg[r, bad] = rng.randint(1, data.itemnum + 1, size=int(bad.sum()))
            bad = np.isin(neg[r], rated)
    cand[:, 1:] = neg
    return (torch.from_numpy(users), torch.from_numpy(seq), torch.from_numpy(rsq), torch.from_numpy(cand))


def window_labels(rsq: torch.Tensor):
    """utils.py:604-626 on the padded review window: binary (fake-majority -> 1 else 2), frequency (#fake),
    ratio floor(10 * fake / (fake + real)) (0 when the window is empty)."""
    n1 = (rsq == 1).sum(1)
    n2 = (rsq == 2).sum(1)
    tot = (n1 + n2).clamp(min=1)
    ratio = torch.floor(n1.double() / tot.double() * 10).long()
    return torch.where(n1 > n2, 1, 2), n1, torch.where(n1 + n2 == 0, torch.zeros_like(ratio), ratio)


@torch.no_grad()
def evaluation(model, data: InteractionData, maxlen: int, batch: int = 2048, n_neg: int = 100, seed: int = 0,
               max_users: int = 10000, with_labels: bool = False, candidates=None):
    """``evaluation`` (NDCG@10, HR@10) - and with ``with_labels`` the per-label breakdowns of ``evaluation_with_label``
    (utils.py:628-752) as {label: [HR, NDCG, count]} dicts - batched on the GPU.  ``candidates``: see eval_inputs."""
    from .evaluate import ranks_from_logits
    users = np.arange(1, data.usernum + 1)
    if data.usernum > max_users:                             # utils.py:551-552
        users = np.random.RandomState(seed).choice(users, max_users, replace=False)
    uid, seq, rsq, cand = eval_inputs(data, maxlen, n_neg, seed, users, candidates)
    dev = next(model.parameters()).device
    was = model.training
    model.eval()
    ranks = []
    for s in range(0, uid.numel(), batch):
        sl = slice(s, s + batch)
        logits = model.predict(uid[sl].to(dev), seq[sl].to(dev), rsq[sl].to(dev), cand[sl].to(dev))
        if logits.dim() == 1:
            logits = logits.unsqueeze(0)
        ranks.append(ranks_from_logits(logits).cpu().long())
    model.train(was)
    if hasattr(model, "check_ids"):
        model.check_ids()                                    # an out-of-table id anywhere above raises here
    rank = torch.cat(ranks)
    hit = rank < 10
    ndcg_u = torch.where(hit, 1.0 / torch.log2(rank.double() + 2.0), torch.zeros((), dtype=torch.float64))
    ndcg, hr = float(ndcg_u.mean()), float(hit.double().mean())
    if not with_labels:
        return ndcg, hr
    out = []
    for lab in window_labels(rsq):
        d = {}
        for v in torch.unique(lab).tolist():
            m = lab == v
            d[int(v)] = [float(hit[m].double().mean()), float(ndcg_u[m].mean()), int(m.sum())]
        out.append(dict(sorted(d.items())))
    per_user = {int(u): [int(r), float(h), float(n)] for u, r, h, n in zip(uid.tolist(), rank.tolist(), hit.tolist(), ndcg_u.tolist())}
    return ndcg, hr, per_user, out[0], out[1], out[2]


class DeviceSampler:
    """``WarpSampler_fr`` replacement: ``next_batch()`` returns the same 7-tuple, already on the device, generated by one
    kernel launch from the CSR histories (no worker processes, reproducible)."""

    def __init__(self, data: InteractionData, batch_size: int = 64, maxlen: int = 10, seed: int = 0, device="cuda",
                 model=None):
        """``model``: the module (or FusedTrainer) the batches will feed - its item table must cover ``data.itemnum``
        (checked once here; the reference would fail at the first nn.Embedding lookup of a larger id)."""
        self.data, self.B, self.L, self.seed, self.device = data, int(batch_size), int(maxlen), int(seed), torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DeviceSampler generates batches on the ROCm GPU; there is no CPU fallback")
        if model is not None:
            lay = getattr(model, "lay", None) or model.layout
            if data.itemnum > lay.n_items:
                raise IndexError(f"the dataset holds item ids up to {data.itemnum} but the model's item table covers "
                                 f"[0, {lay.n_items}]")
        if not bool((data.train_len()[1:] > 1).any()):
            raise ValueError("no user has more than one training interaction")
        self.ptr, self.items, self.reviews = data.to_device(self.device)
        self.index = 0

    def next_batch(self, packed: bool = False, out: torch.Tensor | None = None):
        """``out``: an int64 (6, B, L) device tensor to fill in place - e.g. ``trainer.ids_ring[slot]``, so the batch goes
        from the sampler kernel straight into the fused step's input slot (``trainer.step_slot(slot)``), no copy."""
        user = torch.empty(self.B, device=self.device, dtype=torch.int64)
        if out is None:
            out = torch.empty(6, self.B, self.L, device=self.device, dtype=torch.int64)
        elif out.shape != (6, self.B, self.L) or out.dtype != torch.int64 or not out.is_contiguous() or out.device.type != "cuda":
            raise ValueError("out must be a contiguous int64 (6, B, L) tensor on the sampler's device")
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(_lib.lib().srfrd_sample_batch(ptr(self.ptr), ptr(self.items), ptr(self.reviews), self.data.usernum,
                                            self.data.itemnum, self.B, self.L, self.seed & 0xFFFFFFFF,
                                            self.index & 0xFFFFFFFF, ptr(user), ptr(out), st), "srfrd_sample_batch")
        self.index += 1
        if packed:
            return user, out
        return (user, out[0], out[1], out[2], out[3], out[4], out[5])

    def close(self):
        pass


def load_reference_checkpoint(model, path: str, map_location="cpu"):
    """Load a ``state_dict`` file written by the reference (``torch.save(model.state_dict(), 'model/SRFR_<i>.pt')``,
    trainer.py:409-411; the legacy script's files, fake_label_main.py:163-167, share the SASRec key set) into one of the
    drop-in modules.  ``weights_only=True``: nothing in the file is executed."""
    sd = torch.load(path, map_location=map_location, weights_only=True)
    res = model.load_state_dict(sd, strict=True)
    return res
