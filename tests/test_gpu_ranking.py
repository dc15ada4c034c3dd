"""-m gpu: full-catalog top-k, batched evaluation and the C2-size (BASELINE.json configs[1]) property tests."""
import numpy as np
import pytest
import torch

from oracle import srfrd_oracle as O
from tests.helpers import golden_cfg, load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["SASRec", "SRFR", "SRFRN", "SRFU_F"])
def test_topk_matches_oracle_bit_exact(kind):
    """fused top-10 over the whole catalog == stable descending sort of the oracle's logits (indices bit-exact)."""
    from tests.gpu_util import build_model
    g, sd, batch = load_golden(kind)
    cfg = golden_cfg(kind)
    model = build_model(cfg, sd).eval()
    idx, val = model.topk(None, batch[0].cuda(), batch[1].cuda(), k=10)
    all_items = torch.arange(1, cfg.item_number + 1)
    ref = O.predict(cfg, sd, batch[0], batch[1], all_items)             # (B, I)
    order = np.argsort(-ref.numpy(), axis=1, kind="stable")[:, :10]
    assert (idx.cpu().numpy() == order + 1).all()
    assert float((val.cpu() - torch.gather(ref, 1, torch.from_numpy(order))).abs().max()) < 1e-4
    # including the padding item 0 and a sub-range of the catalog (row-sharded table use case)
    idx0, _ = model.topk(None, batch[0].cuda(), batch[1].cuda(), k=5, exclude_pad=False, item_range=(0, 64))
    ref0 = O.predict(cfg, sd, batch[0], batch[1], torch.arange(0, 64))
    assert (idx0.cpu().numpy() == np.argsort(-ref0.numpy(), axis=1, kind="stable")[:, :5]).all()


def test_topk_tie_break_prefers_lower_item_id():
    import srfrd_amd
    m = srfrd_amd.SASRec(700, 20, 50, 0.0, 1, 1, "cuda").cuda().eval()
    with torch.no_grad():
        m.item_emb.weight[1:] = m.item_emb.weight[1:2]                 # every item scores identically
    seq = torch.randint(1, 700, (3, 20)).cuda()
    idx, val = m.topk(None, seq, None, k=7)
    assert (idx.cpu() == torch.arange(1, 8)).all() and float((val - val[:, :1]).abs().max()) == 0.0


def test_topk_mass_ties_take_the_exhaustive_fallback():
    """more tied scores than a candidate list holds: the device-armed exhaustive path must still give the stable order."""
    import srfrd_amd
    m = srfrd_amd.SASRec(6000, 20, 50, 0.0, 1, 1, "cuda").cuda().eval()
    with torch.no_grad():
        m.item_emb.weight[1:] = m.item_emb.weight[1:2]
        m.item_emb.weight[4000] = m.item_emb.weight[1] * 1.5          # one item that may outrank the tie
    seq = torch.randint(1, 6000, (5, 20)).cuda()
    idx, val = m.topk(None, seq, None, k=8)
    with torch.no_grad():
        h = m(None, seq, None)[0][:, -1]
        ref = h @ m.item_emb.weight[1:].T
    order = np.argsort(-ref.cpu().numpy(), axis=1, kind="stable")[:, :8] + 1
    assert (idx.cpu().numpy() == order).all()


def test_batched_evaluation_matches_oracle_metric():
    """HR@10 / NDCG@10 over 101 candidates per user (reference utils.py:576-598) for a golden model."""
    import srfrd_amd
    from tests.gpu_util import build_model
    g, sd, batch = load_golden("SRFU_B")
    cfg = golden_cfg("SRFU_B")
    model = build_model(cfg, sd).eval()
    cands = torch.from_numpy(g["cands"])
    ndcg, hr = srfrd_amd.evaluate_batches(model, [(None, batch[0][:4].cuda(), batch[1][:4].cuda(), cands[:4].cuda()),
                                                  (None, batch[0][4:].cuda(), batch[1][4:].cuda(), cands[4:].cuda())])
    ranks = O.rank_of_first(torch.from_numpy(g["pred_logits"]))
    nd, h = O.hr_ndcg_at_10(ranks)
    assert abs(ndcg - nd) < 1e-12 and abs(hr - h) < 1e-12


# ---------------------------------------------------------------------------------------------------------------
# C2 size: 50 000 items, seq_len 50, batch 512 - too large for the CPU oracle in seconds, so size-independent
# properties of the path are checked instead
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def c2():
    import srfrd_amd
    torch.manual_seed(0)
    m = srfrd_amd.SASRec(50_000, 50, 50, 0.5, 2, 1, "cuda")
    for _, p in m.named_parameters():
        if p.dim() >= 2:
            torch.nn.init.xavier_normal_(p.data)
    m = m.cuda()
    batch = srfrd_amd.synthetic_batch(50_000, 50, 512, seed=1, device="cuda")
    return m, batch


def test_c2_forward_deterministic_causal_and_batch_invariant(c2):
    m, (u, seq, rsq, pos, prs, neg, nrs) = c2
    m.eval()
    with torch.no_grad():
        h1, p1, n1 = m(u, seq, rsq, pos, prs, neg, nrs)
        h2, p2, n2 = m(u, seq, rsq, pos, prs, neg, nrs)
        assert torch.equal(h1, h2) and torch.equal(p1, p2) and torch.equal(n1, n2)          # bitwise repeatable
        # a sequence's result does not depend on its neighbours or its slot in the batch
        perm = torch.randperm(512, device="cuda")
        h3, p3, _ = m(u, seq[perm][:100], rsq, pos[perm][:100], prs, neg[perm][:100], nrs)
        assert torch.equal(h3, h1[perm][:100]) and torch.equal(p3, p1[perm][:100])
        # causality: changing the last item leaves every earlier position untouched
        seq2 = seq.clone()
        seq2[:, -1] = (seq2[:, -1] % 49_999) + 1
        h4, _, _ = m(u, seq2, rsq, pos, prs, neg, nrs)
        assert torch.equal(h4[:, :-1], h1[:, :-1]) and not torch.equal(h4[:, -1], h1[:, -1])
        # padded positions are fed zeros: logits there equal <LN-of-constant, E[0]>, identical across rows with pads
        assert bool(torch.isfinite(h1).all())


def test_c2_gradients_are_linear_in_the_upstream_gradient(c2):
    m, (u, seq, rsq, pos, prs, neg, nrs) = c2
    m.eval()                                              # dropout off: exact linearity
    ids = m._prep(seq, None, pos, None, neg, None)
    out = m._launch_fwd(*ids, 0.0, 0, save=True)
    g1 = torch.randn_like(out["pos_logits"])
    g2 = torch.randn_like(out["neg_logits"])
    a = m._launch_bwd(*ids, 0.0, 0, out, None, g1, g2)
    b = m._launch_bwd(*ids, 0.0, 0, out, None, 2 * g1, 2 * g2)
    dense = slice(m.n_table_pad, m.n_table_pad + m.layout.n_dense)
    assert torch.equal(b[dense], 2 * a[dense])                                   # dense part: bitwise (fixed-order sums)
    assert float((b[:m.layout.n_table] - 2 * a[:m.layout.n_table]).abs().max()) < 1e-4   # table: float atomics
    rows = a[:m.layout.n_table].view(-1, 50)
    assert float(rows[0].abs().max()) == 0.0                                      # padding_idx row never written
    touched = torch.unique(torch.cat([seq.flatten(), pos.flatten(), neg.flatten()]))
    mask = torch.ones(rows.shape[0], dtype=torch.bool, device="cuda")
    mask[touched] = False
    assert float(rows[mask].abs().max()) == 0.0                                   # untouched rows get exactly zero


def test_c2_fused_step_equals_autograd_step_and_learns(c2):
    import copy
    import srfrd_amd
    m, (u, seq, rsq, pos, prs, neg, nrs) = c2
    m1, m2 = copy.deepcopy(m), copy.deepcopy(m)
    m1.dropout_rate = m2.dropout_rate = 0.0
    m1.train(); m2.train()
    tr = srfrd_amd.FusedTrainer(m1, 512, 50, use_graph=True)
    opt = torch.optim.Adam(m2.parameters(), lr=1e-3, betas=(0.9, 0.98))
    crit = torch.nn.BCEWithLogitsLoss()
    from tests.helpers import adam_tolerance
    losses = []
    for step in range(3):
        l1 = tr.step(u, seq, rsq, pos, prs, neg, nrs)
        h, pl, nl = m2(u, seq, rsq, pos, prs, neg, nrs)
        idx = torch.where(pos != 0)
        l2 = crit(pl[idx], torch.ones_like(pl)[idx]) + crit(nl[idx], torch.zeros_like(nl)[idx])
        opt.zero_grad()
        l2.backward()
        g0 = {k: p.grad.detach().clone() for k, p in m2.named_parameters()}
        opt.step()
        assert abs(float(l1) - float(l2.detach())) < 1e-5
        losses.append(float(l1))
        if step == 0:
            # Weights after ONE step from identical state, element-wise (tests/helpers.adam_tolerance): 1e-4 or tighter
            # wherever the gradient is real, relaxing to lr only for elements whose gradient is rounding noise (Adam normalises
            # the magnitude away, so their sign - which depends on summation order - is the step).  Later steps are compared
            # through the loss: two runs that scatter the table gradient with float atomics differ by an ulp in some
            # embeddings after step 0, units on their ReLU threshold then change derivative in one run only, and with 2.5 M
            # ReLU units per step some always do - small-gradient elements then take opposite Adam steps.
            sd1, sd2 = m1.state_dict(), m2.state_dict()
            for k in sd1:
                d = (sd1[k] - sd2[k]).abs().double().cpu()
                bad = d > adam_tolerance([g0[k].cpu()])
                assert not bool(bad.any()), (k, float(d[bad].max()), int(bad.sum()))
    assert losses[2] < losses[0]                                                  # it trains
    sd1, sd2 = m1.state_dict(), m2.state_dict()
    for k in sd1:                                   # two more steps: at most 2 lr apart per step (opposite Adam steps)
        assert float((sd1[k] - sd2[k]).abs().max()) <= 2 * 2 * 1e-3 * 1.1 + 1e-4, k


def test_c2_untrained_hit_rate_is_chance(c2):
    import srfrd_amd
    m, (u, seq, rsq, pos, prs, neg, nrs) = c2
    cand = srfrd_amd.eval_candidates(50_000, seq, pos[:, -1], 100, seed=3)
    ndcg, hr = srfrd_amd.evaluate_batches(m, [(u, seq, rsq, cand)])
    assert 0.04 < hr < 0.18                       # 10 / 101 for a random ranker (512 users)
    idx, val = m.topk(u, seq, rsq, k=10)
    assert idx.shape == (512, 10) and bool((val[:, :-1] >= val[:, 1:]).all()) and bool((idx >= 1).all())


@pytest.mark.parametrize("kind,L", [("SASRec", 50), ("SRFR", 50), ("SRFRN", 20), ("SRFU_F", 37), ("SASRec", 100), ("SRFRN", 200)])
def test_last_position_forward_equals_last_row_of_full_forward(kind, L):
    """srfrd_encoder_fwd_last (what predict() / topk() rank with): the last block works on the one 16-row tile that holds
    position L - 1 - bit for bit row L - 1 of the full eval-mode forward, for every kind, on the LDS-resident builds
    (L <= 112) and the global-scratch build (L = 200), with left-padded and full sequences in the batch."""
    import srfrd_amd
    from tests.gpu_util import build_model, cuda, random_sd
    I, B = 300, 9
    if kind == "SASRec":
        cfg = O.Cfg(kind, I, L, 50)
    elif kind in ("SRFR", "SRFRN"):
        cfg = O.Cfg(kind, I, L, 45, d_fake=5)
    else:
        cfg = O.Cfg(kind, I, L, 50, n_labels=L + 1)          # SRFU_F: the label is the number of fake reviews
    sd = random_sd(cfg, 4)
    model = build_model(cfg, sd).eval()
    batch = srfrd_amd.synthetic_batch(I, L, B, seed=11, device="cpu")
    seq, rsq = cuda(batch[1], batch[2])
    seq[0] = torch.randint(1, I + 1, (L,), device=seq.device)          # one sequence without padding
    with torch.no_grad():
        ids = model._prep(seq, rsq, None, None, None, None)
        full = model._launch_fwd(*ids, 0.0, 0, save=False)["hidden"]
        last = model._launch_fwd_last(ids[0], ids[1])
    assert last.shape == (B, 1, cfg.d_out)
    assert torch.equal(last[:, 0], full[:, -1])
    ho, _, _ = O.forward(cfg, sd, seq.cpu(), rsq.cpu(), None, None, None, None)
    assert float((last[:, 0].cpu() - ho[:, -1]).abs().max()) < 1e-4


def test_topk_on_the_fp32_matrix_cores_agrees(monkeypatch):
    """SRFRD_TOPK_FP32=1 keeps the fp32-matrix threshold passes (the path of item widths above 52): same indices, values
    within fp32 rounding of the bf16-split path's."""
    import srfrd_amd
    from tests.gpu_util import build_model, cuda, random_sd
    cfg = O.Cfg("SRFRN", 6000, 20, 45, d_fake=5)
    model = build_model(cfg, random_sd(cfg, 12)).eval()
    batch = srfrd_amd.synthetic_batch(6000, 20, 40, seed=2, device="cpu")
    seq, rsq = cuda(batch[1], batch[2])
    i0, v0 = model.topk(None, seq, rsq, k=10)
    monkeypatch.setenv("SRFRD_TOPK_FP32", "1")
    i1, v1 = model.topk(None, seq, rsq, k=10)
    assert torch.equal(i0, i1) and float((v0 - v1).abs().max()) < 1e-5
