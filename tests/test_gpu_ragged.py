"""-m gpu: the ragged seq_len-50 kernel pair (srfrd_encoder_fwd_ragged_kernel.inc / srfrd_encoder_bwd_ragged_kernel.inc): only
the rows a left-padded sequence really has are computed, the t0 leading pads act as ONE representative key with multiplicity
t0 (reference semantics: no key-padding mask, SRFR_model.py:112 - pads are keys with k = b_k, v = b_v), and sequences are
scheduled over the workgroups by length (srfrd_seq_order).  Every other L = 50 test of the suite runs this pair too (it is
the default); here: the cases that are specific to it, against the oracle and against the full-row kernels."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import srfrd_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
I, L = 400, 50


def _cfg(kind, dropout=0.0):
    if kind == "SASRec":
        return O.Cfg(kind, I, L, 50, dropout=dropout)
    if kind in ("SRFR", "SRFRN"):
        return O.Cfg(kind, I, L, 45, d_fake=5, dropout=dropout)
    return O.Cfg(kind, I, L, 50, n_labels=3, dropout=dropout)


def _batch_with_every_pad_count(B=64, seed=3, interior=True, pad_targets=True):
    """sequences with 0, 1, 2, ... leading pads (every tile boundary of the ragged kernels is crossed), all-pad rows, a few
    INTERIOR pads (id 0 behind the first item: an ordinary masked row, not part of the merged prefix) and - pad_targets - target
    ids on padded positions (an upstream gradient in front of the first item: the backward's head must cover it)."""
    g = torch.Generator().manual_seed(seed)
    seq = torch.randint(1, I + 1, (B, L), generator=g)
    pos = torch.randint(1, I + 1, (B, L), generator=g)
    neg = torch.randint(1, I + 1, (B, L), generator=g)
    for b in range(B):
        t0 = min(b, L) if b <= L else int(torch.randint(0, L + 1, (1,), generator=g))
        seq[b, :t0] = 0
        if not (pad_targets and b % 5 == 0):
            pos[b, :t0] = 0
            neg[b, :t0] = 0
        if interior and b % 7 == 3 and t0 + 3 < L:
            seq[b, t0 + 2] = 0
    rsq = torch.where(seq != 0, torch.randint(1, 3, (B, L), generator=g), torch.zeros_like(seq))
    prs = torch.where(pos != 0, torch.randint(1, 3, (B, L), generator=g), torch.zeros_like(pos))
    nrs = (neg != 0).long()
    return seq, rsq, pos, prs, neg, nrs


def _loss(pl, nl, pos):
    idx = torch.where(pos != 0)
    crit = torch.nn.BCEWithLogitsLoss()
    return crit(pl[idx], torch.ones_like(pl)[idx]) + crit(nl[idx], torch.zeros_like(nl)[idx])


@pytest.mark.parametrize("kind", ["SASRec", "SRFR", "SRFRN", "SRFU_B"])
def test_every_leading_pad_count_forward_and_gradients_match_the_oracle(kind):
    import srfrd_amd
    from tests.gpu_util import build_model, cuda, maxerr, random_sd
    cfg = _cfg(kind)
    sd = random_sd(cfg, 31)
    model = build_model(cfg, sd).train()
    batch = _batch_with_every_pad_count()
    loss_o, grads_o, h_o, pl_o, nl_o = O.grads_of(cfg, sd, batch)
    seq, rsq, pos, prs, neg, nrs = cuda(*batch)
    h, pl, nl = model(None, seq, rsq, pos, prs, neg, nrs)
    assert maxerr(h, h_o) < TOL and maxerr(pl, pl_o) < TOL and maxerr(nl, nl_o) < TOL
    loss = _loss(pl, nl, pos)
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_o)) < TOL
    for k, p in model.named_parameters():
        assert maxerr(p.grad, grads_o[k]) < TOL, k
    # eval forward and the ranking forward (last position only) on the same batch
    model.eval()
    with torch.no_grad():
        h2 = model(None, seq, rsq)[0]
    assert maxerr(h2, h_o) < TOL
    cand = torch.arange(1, 102).repeat(seq.shape[0], 1)
    assert maxerr(model.predict(None, seq, rsq, cand.cuda()), O.predict(cfg, sd, batch[0], batch[1], cand)) < TOL


@pytest.mark.parametrize("kind", ["SASRec", "SRFRN"])
def test_every_leading_pad_count_fused_step_with_dropout_matches_the_oracle(kind):
    """train mode, dropout 0.5: the merged pads' attention-dropout count c_i (kept ones among the coordinates (i, 0 .. t0 - 1))
    must be the oracle's element-wise masks summed; two fused Adam steps."""
    import srfrd_amd
    from tests.gpu_util import build_model, cuda, random_sd
    from tests.helpers import assert_post_adam, oracle_step_with_grads
    cfg = _cfg(kind, dropout=0.5)
    sd = random_sd(cfg, 32)
    model = build_model(cfg, {k: v.clone() for k, v in sd.items()}).train()
    B, base = 64, 11
    tr = srfrd_amd.FusedTrainer(model, batch_size=B, seq_len=L, seed=base, use_graph=False)
    opt = O.Adam(sd)
    hist = []
    for step in range(2):
        batch = _batch_with_every_pad_count(B, seed=40 + step, pad_targets=False)
        loss = tr.step(None, *cuda(*batch))
        loss_o, g_o = oracle_step_with_grads(cfg, sd, opt, batch, train=True, seed=O.step_seed(base, step + 1), b0=0)
        hist.append(g_o)
        assert abs(float(loss.cpu()) - float(loss_o)) < TOL, step
    assert_post_adam(model.state_dict(), sd, hist, cfg.D)


def test_upstream_hidden_gradient_reaches_the_padded_rows():
    """a loss on `hidden` itself puts an upstream gradient on every position, pads included (their hidden state is
    LayerNorm(0) = beta: the last LayerNorm's bias gradient sums over them): the backward's head covers every row then."""
    from tests.gpu_util import build_model, cuda, maxerr, random_sd
    cfg = _cfg("SRFR")
    sd = random_sd(cfg, 33)
    model = build_model(cfg, sd).train()
    batch = _batch_with_every_pad_count(32, seed=5)
    seq, rsq, pos, prs, neg, nrs = cuda(*batch)
    h, pl, nl = model(None, seq, rsq, pos, prs, neg, nrs)
    w = torch.randn(h.shape, generator=torch.Generator().manual_seed(1)).cuda()
    ((h * w).sum() / h.shape[0] + _loss(pl, nl, pos)).backward()
    ref = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ho, plo, nlo = O.forward(cfg, ref, *batch)
    lo = (ho * w.cpu()).sum() / ho.shape[0] + _loss(plo, nlo, batch[2])
    lo.backward()
    for k, p in model.named_parameters():
        assert maxerr(p.grad, ref[k].grad) < TOL, k


def test_ragged_pair_equals_the_full_row_kernels_and_every_schedule():
    """child processes (the switches are read per launch, the comparison wants clean state): the same fused dropout step
    under (a) the ragged pair with and without the length-ordered schedule, (b) the ragged kernels forced to compute every row
    (SRFRD_RAGGED_FULL_ROWS), (c) the full-row kernels (SRFRD_NO_RAGGED): loss and every parameter agree to rounding."""
    script = r'''
import json, os, sys, torch
sys.path.insert(0, os.getcwd())
import srfrd_amd
from tests.test_gpu_ragged import _batch_with_every_pad_count
torch.manual_seed(0)
m = srfrd_amd.SASRec(400, 50, 50, 0.5, 2, 1, "cuda")
for _, p in m.named_parameters():
    if p.dim() >= 2:
        torch.nn.init.xavier_normal_(p.data)
m = m.cuda().train()
B = 600
tr = srfrd_amd.FusedTrainer(m, B, 50, seed=3, use_graph=False, deterministic=True)
batch = _batch_with_every_pad_count(B, seed=9, pad_targets=False)
loss = float(tr.step(None, *[t.cuda() for t in batch]).cpu())
torch.save({"loss": loss, "flat": tr.flat[:m.n_flat].cpu(), "sched": tr.sched_mode}, sys.argv[1])
'''
    import tempfile
    outs = {}
    with tempfile.TemporaryDirectory() as td:
        for name, env in (("ordered", {"SRFRD_SCHED": "1"}), ("none", {"SRFRD_SCHED": "0"}),
                          ("full_rows", {"SRFRD_RAGGED_FULL_ROWS": "1"}), ("old_kernels", {"SRFRD_NO_RAGGED": "1"})):
            path = os.path.join(td, name + ".pt")
            r = subprocess.run([sys.executable, "-c", script, path], cwd=ROOT, env=dict(os.environ, **env), capture_output=True,
                               text=True, timeout=600)
            assert r.returncode == 0, (name, r.stderr[-3000:])
            outs[name] = torch.load(path, weights_only=True)
    ref = outs["old_kernels"]
    for name, o in outs.items():
        assert abs(o["loss"] - ref["loss"]) < 2e-6, (name, o["loss"], ref["loss"])
        d = (o["flat"] - ref["flat"]).abs()
        # one Adam step from identical weights: equal up to the sign of noise-level gradients (<= 2 lr), 1e-6 in the mean
        assert float(d.max()) <= 2.2e-3 and float(d.mean()) < 2e-6, (name, float(d.max()), float(d.mean()))
    # with or without the schedule the per-sequence arithmetic is the same: forward-side results are bit-equal
    assert outs["ordered"]["loss"] == outs["none"]["loss"]


def test_seq_order_writes_the_first_item_position_of_every_sequence():
    from srfrd_amd import _lib
    from srfrd_amd._lib import check, ptr
    lib = _lib.lib()
    for B, Lx in ((512, 50), (37, 50), (1000, 20), (3, 50), (5, 200)):
        g = torch.Generator().manual_seed(B)
        t0 = torch.randint(0, Lx + 1, (B,), generator=g)
        ids = torch.randint(1, 100, (B, Lx), generator=g)
        for b in range(B):
            ids[b, :int(t0[b])] = 0
        ids = ids.cuda()
        n = int(lib.srfrd_sched_ints(B))
        sched = torch.full((n,), -7, device="cuda", dtype=torch.int32)
        check(lib.srfrd_seq_order(ptr(ids), B, Lx, 256, ptr(sched), None), "srfrd_seq_order")
        torch.cuda.synchronize()
        s = sched.cpu()
        assert int(s[0]) == 256 and torch.equal(s[16:16 + B].long(), t0)
