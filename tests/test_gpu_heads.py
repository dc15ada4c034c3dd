"""-m gpu: num_heads > 1 (reference SRFR_model.py: every class hands ``num_heads`` to nn.MultiheadAttention; the shipped configs
use 1).  Several heads run the generic instantiation of the encoder kernels (heads take turns in the score buffer): checked
here against fixtures made from the REFERENCE's classes with 2 and 5 heads (tests/golden/<kind>_h<heads>.npz) and, with
dropout on (one coordinate-hash mask per head), against the CPU oracle.  Tolerance 1e-4 absolute (fp32)."""
import numpy as np
import pytest
import torch

from oracle import srfrd_oracle as O
from tests.helpers import HEAD_CASES, drop_kbias, golden_cfg, load_golden, sub

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.mark.parametrize("kind,heads", HEAD_CASES)
def test_heads_forward_and_predict_match_golden(kind, heads):
    from tests.gpu_util import build_model, cuda, maxerr
    g, sd, batch = load_golden(kind, heads)
    model = build_model(golden_cfg(kind, heads=heads), sd).eval()
    assert model.layout.n_heads == heads
    with torch.no_grad():
        h, pl, nl = model(None, *cuda(*batch))
    assert maxerr(h, torch.from_numpy(g["hidden"])) < TOL
    assert maxerr(pl, torch.from_numpy(g["pos_logits"])) < TOL
    assert maxerr(nl, torch.from_numpy(g["neg_logits"])) < TOL
    cands = torch.from_numpy(g["cands"]).cuda()
    out = model.predict(None, batch[0].cuda(), batch[1].cuda(), cands)          # (last-position forward)
    assert maxerr(out, torch.from_numpy(g["pred_logits"])) < TOL
    assert (np.argsort(-out.cpu().numpy(), axis=1, kind="stable")[:, :10]
            == np.argsort(-g["pred_logits"], axis=1, kind="stable")[:, :10]).all()
    # the last-position kernel path is bit-equal to the last row of the full forward
    ids = model._prep(batch[0].cuda(), batch[1].cuda(), None, None, None, None)
    last = model._launch_fwd_last(ids[0], ids[1])
    assert torch.equal(last[:, 0], h[:, -1])


@pytest.mark.parametrize("kind,heads", HEAD_CASES)
def test_heads_fused_trainer_matches_golden(kind, heads):
    """3 fused steps (dropout_rate = 0): the reference's loss curve, gradients through step-1 weights, step-3 weights."""
    import srfrd_amd
    from tests.gpu_util import build_model, cuda, maxerr
    g, sd, batch = load_golden(kind, heads)
    cfg = golden_cfg(kind, heads=heads)
    model = build_model(cfg, sd).train()
    # gradients of the reference's loss through the module + autograd
    seq, rsq, pos, prs, neg, nrs = cuda(*batch)
    h, pl, nl = model(None, seq, rsq, pos, prs, neg, nrs)
    idx = torch.where(pos != 0)
    crit = torch.nn.BCEWithLogitsLoss()
    loss = crit(pl[idx], torch.ones_like(pl)[idx]) + crit(nl[idx], torch.zeros_like(nl)[idx])
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss0"])) < TOL
    gg = sub(g, "g/")
    bad = {k: maxerr(p.grad, gg[k]) for k, p in model.named_parameters() if not maxerr(p.grad, gg[k]) < TOL}
    assert not bad, bad
    model.zero_grad(set_to_none=True)
    tr = srfrd_amd.FusedTrainer(model, 8, 20, lr=1e-3, betas=(0.9, 0.98))
    w1, w3 = sub(g, "w1/"), sub(g, "w3/")
    for step in range(3):
        loss = tr.step(None, seq, rsq, pos, prs, neg, nrs)
        assert abs(float(loss.cpu()) - float(g[f"loss{step}"])) < TOL, step
        if step == 0:
            msd = model.state_dict()
            for k in w1:
                assert maxerr(drop_kbias(k, msd[k].cpu(), cfg.D), drop_kbias(k, w1[k], cfg.D)) < TOL, k
    msd = model.state_dict()
    for k in w3:
        assert maxerr(drop_kbias(k, msd[k].cpu(), cfg.D), drop_kbias(k, w3[k], cfg.D)) < 2e-4, k


def _dropout_case(cfg, sd, batch, p=0.5, seed=0xBEEF, seq0=17, tol_g=2e-4):
    from tests.gpu_util import build_model, cuda, maxerr
    model = build_model(cfg, sd).train()
    ids = model._prep(*cuda(*batch))
    out = model._launch_fwd(*ids, p, seed, save=True, seq0=seq0)
    ho, plo, nlo = O.forward(cfg, sd, *batch, train=True, seed=seed, b0=seq0)
    assert maxerr(out["hidden"], ho) < TOL and maxerr(out["pos_logits"], plo) < TOL and maxerr(out["neg_logits"], nlo) < TOL
    dpl = torch.full_like(out["pos_logits"], 0.3)
    dnl = torch.full_like(out["neg_logits"], -0.2)
    gflat = model._launch_bwd(*ids, p, seed, out, None, dpl, dnl, seq0=seq0)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    _, p2, n2 = O.forward(cfg, leaves, *batch, train=True, seed=seed, b0=seq0)
    (0.3 * p2.sum() - 0.2 * n2.sum()).backward()
    for k, prm in model.named_parameters():
        off = next(o for q, o in model._slots if q is prm)
        got = gflat[off:off + prm.numel()].view(prm.shape)
        ref = leaves[k].grad if leaves[k].grad is not None else torch.zeros_like(leaves[k])
        if k.endswith("item_embed.weight") or k == "item_emb.weight" or k.endswith("fake_embed.weight"):
            ref = ref.clone()
            ref[0] = 0
        assert maxerr(got, ref) < tol_g, k


@pytest.mark.parametrize("kind,heads", HEAD_CASES)
def test_heads_dropout_matches_oracle_masks(kind, heads):
    """p = 0.5: each head draws its own attention mask (site_attn(block, head)); outputs and every gradient."""
    g, sd, batch = load_golden(kind, heads)
    _dropout_case(golden_cfg(kind, dropout=0.5, heads=heads), sd, batch)


@pytest.mark.parametrize("kind,L,D,heads", [("SASRec", 50, 48, 4), ("SRFU_B", 37, 64, 8), ("SASRec", 100, 50, 2), ("SRFRN", 130, 50, 5)])
def test_heads_other_shapes_dropout(kind, L, D, heads):
    """head widths that are / are not multiples of 4, a ragged length, and two lengths whose working set leaves LDS (the
    global-scratch build of the same kernels)."""
    from tests.gpu_util import random_sd
    from srfrd_amd.sampler import synthetic_batch
    if kind == "SRFRN":
        cfg = O.Cfg(kind, 300, L, D - 5, d_fake=5, dropout=0.5, num_heads=heads)
    elif kind == "SRFU_B":
        cfg = O.Cfg(kind, 300, L, D, n_labels=3, dropout=0.5, num_heads=heads)
    else:
        cfg = O.Cfg(kind, 300, L, D, dropout=0.5, num_heads=heads)
    sd = random_sd(cfg, seed=5)
    u, *batch = synthetic_batch(300, L, 6, seed=3)
    _dropout_case(cfg, sd, tuple(batch), tol_g=3e-4)
