"""-m gpu: the instantiations that produce the C4 / C5 bench numbers (BASELINE configs[3] / [4]), at the batch geometry
the bench runs them in: MORE sequences than persistent workgroups, so that every long-sequence kernel goes round its
sequence loop and the dense gradients go through the slab read-modify-write -

  * seq_len 100: the 16-wave forward `encoder_fwd_kernel<50, 112, 16, 100, ...>` (SASRec) / the 8-wave one (SRFRN) and
    the slot-placed backward `encoder_bwd_slots_kernel<50, 100, K, DI, true>` (the read-modify-write form is selected only
    when B > grid, srfrd_encoder_bwd_slots.hip);
  * seq_len 144 / 200: the row-owner training forward and the row-chunked backward, several sequences per workgroup
    (slab tiles initialised from the previous sequence's) - 144 is three chunks exactly, 200 has an 8-row tail.

tests/test_gpu_long.py runs the same lengths at B <= 7 (one sequence per workgroup: the store-only instantiations).
Reference path: SRFR_model.py:109-136, trainer.py:36-41.  Then C4 at FULL size (200k items, seq_len 100, 512 sequences per
GPU) through the size-independent properties tests/test_gpu_c3.py holds C3 to."""
import copy

import pytest
import torch

from oracle import srfrd_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4
B_MANY = 300          # > 256 persistent workgroups (one per CU at these lengths)


def _cfg(kind, L, I=400, dropout=0.0):
    if kind == "SASRec":
        return O.Cfg(kind, I, L, 50, dropout=dropout)
    return O.Cfg(kind, I, L, 45, d_fake=5, dropout=dropout)


def _loss(pl, nl, pos):
    idx = torch.where(pos != 0)
    crit = torch.nn.BCEWithLogitsLoss()
    return crit(pl[idx], torch.ones_like(pl)[idx]) + crit(nl[idx], torch.zeros_like(nl)[idx])


@pytest.mark.parametrize("kind,L", [("SASRec", 100), ("SRFRN", 100), ("SASRec", 144), ("SASRec", 200), ("SRFRN", 200)])
def test_more_sequences_than_workgroups_at_c4_c5_lengths(kind, L):
    """B = 300 > 256 workgroups: gradients through the autograd path and one fused step with dropout 0.5 (the train-mode
    instantiations the bench times) against the oracle - the seq_len 100 / 200 twin of
    test_gpu_train.py::test_more_sequences_than_workgroups_accumulate_in_the_slabs."""
    import ctypes as C
    import srfrd_amd
    from srfrd_amd import _lib
    from tests.gpu_util import build_model, cuda, maxerr, random_sd
    from tests.helpers import assert_post_adam, oracle_step_with_grads
    B, I = B_MANY, 400
    cfg = _cfg(kind, L)
    sd = random_sd(cfg, 21)
    model = build_model(cfg, sd).train()
    grid = _lib.lib().srfrd_bwd_grid(C.byref(model.layout), B, L)
    assert 0 < grid < B, (grid, B)            # really several sequences per backward workgroup
    batch = srfrd_amd.synthetic_batch(I, L, B, seed=13, device="cpu")[1:]
    loss_o, grads_o, h_o, pl_o, nl_o = O.grads_of(cfg, sd, batch)
    seq, rsq, pos, prs, neg, nrs = cuda(*batch)
    h, pl, nl = model(None, seq, rsq, pos, prs, neg, nrs)
    assert maxerr(h, h_o) < TOL and maxerr(pl, pl_o) < TOL and maxerr(nl, nl_o) < TOL
    loss = _loss(pl, nl, pos)
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_o)) < TOL
    for k, p in model.named_parameters():
        assert maxerr(p.grad, grads_o[k]) < TOL, k
    # fused step with dropout: forward in training mode (checkpoints, loss sums), backward with the masks regenerated / read back
    cfg_d = _cfg(kind, L, dropout=0.5)
    sd_d = random_sd(cfg_d, 22)
    model_d = build_model(cfg_d, {k: v.clone() for k, v in sd_d.items()}).train()
    tr = srfrd_amd.FusedTrainer(model_d, batch_size=B, seq_len=L, lr=1e-3, betas=(0.9, 0.98), seed=5, use_graph=False)
    full = srfrd_amd.synthetic_batch(I, L, B, seed=14, device="cpu")
    loss_f = tr.step(*cuda(*full))
    opt = O.Adam(sd_d)
    loss_fo, g_o = oracle_step_with_grads(cfg_d, sd_d, opt, full[1:], train=True, seed=O.step_seed(5, 1), b0=0)
    assert abs(float(loss_f.cpu()) - float(loss_fo)) < TOL
    # (2 M - 6 M ReLU units per step: a unit at its threshold may flip - tests/helpers.assert_post_adam, `outliers`)
    assert_post_adam(model_d.state_dict(), sd_d, [g_o], cfg_d.D, outliers=2e-3)


# ---- C4 at full size: 200 000 items, seq_len 100, 512 sequences per GPU (the per-GPU share of the 4096 global batch) ----
I4, L4, B4 = 200_000, 100, 512


@pytest.fixture(scope="module")
def c4():
    import srfrd_amd
    torch.manual_seed(0)
    m = srfrd_amd.SASRec(I4, L4, 50, 0.5, 2, 1, "cuda")
    for _, p in m.named_parameters():
        if p.dim() >= 2:
            torch.nn.init.xavier_normal_(p.data)
    m = m.cuda()
    return m, srfrd_amd.synthetic_batch(I4, L4, B4, seed=1, device="cuda")


def test_c4_forward_deterministic_causal_and_batch_invariant(c4):
    m, (u, seq, rsq, pos, prs, neg, nrs) = c4
    m.eval()
    with torch.no_grad():
        h1, p1, n1 = m(u, seq, rsq, pos, prs, neg, nrs)
        h2, p2, n2 = m(u, seq, rsq, pos, prs, neg, nrs)
        assert torch.equal(h1, h2) and torch.equal(p1, p2) and torch.equal(n1, n2)
        perm = torch.randperm(B4, device="cuda")[:100]
        h3, p3, n3 = m(u, seq[perm], rsq[perm], pos[perm], prs[perm], neg[perm], nrs[perm])
        assert torch.equal(h3, h1[perm]) and torch.equal(p3, p1[perm]) and torch.equal(n3, n1[perm])
        seq2 = seq.clone()
        seq2[:, -1] = (seq2[:, -1] % (I4 - 1)) + 1
        h4, _, _ = m(u, seq2, rsq, pos, prs, neg, nrs)
        assert torch.equal(h4[:, :-1], h1[:, :-1]) and not torch.equal(h4[:, -1], h1[:, -1])
        assert bool(torch.isfinite(h1).all())


def test_c4_gradients_are_linear_and_untouched_rows_stay_zero(c4):
    m, (u, seq, rsq, pos, prs, neg, nrs) = c4
    m.eval()
    ids = m._prep(seq, rsq, pos, prs, neg, nrs)
    out = m._launch_fwd(*ids, 0.0, 0, save=True)
    g1, g2 = torch.randn_like(out["pos_logits"]), torch.randn_like(out["neg_logits"])
    a = m._launch_bwd(*ids, 0.0, 0, out, None, g1, g2)
    b = m._launch_bwd(*ids, 0.0, 0, out, None, 2 * g1, 2 * g2)
    dense = slice(m.n_table_pad, m.n_table_pad + m.layout.n_dense)
    assert torch.equal(b[dense], 2 * a[dense])
    di = m.layout.d_item
    assert float((b[:m.layout.n_table] - 2 * a[:m.layout.n_table]).abs().max()) < 1e-4
    rows = a[:m.layout.n_table].view(-1, di)
    assert float(rows[0].abs().max()) == 0.0
    touched = torch.unique(torch.cat([seq.flatten(), pos.flatten(), neg.flatten()]))
    mask = torch.ones(rows.shape[0], dtype=torch.bool, device="cuda")
    mask[touched] = False
    assert float(rows[mask].abs().max()) == 0.0


def test_c4_fused_step_equals_autograd_step_and_learns(c4):
    """fused step (16-wave forward + read-modify-write slot backward + fused Adam, graph-replayed) == module forward +
    autograd + torch.optim.Adam after ONE step from identical weights, element-wise (tests/helpers.adam_tolerance); the
    following steps through the loss (ReLU-threshold flips)."""
    import srfrd_amd
    from tests.helpers import adam_tolerance
    m, (u, seq, rsq, pos, prs, neg, nrs) = c4
    m1, m2 = copy.deepcopy(m), copy.deepcopy(m)
    m1.dropout_rate = m2.dropout_rate = 0.0
    m1.train(); m2.train()
    tr = srfrd_amd.FusedTrainer(m1, B4, L4, use_graph=True)
    opt = torch.optim.Adam(m2.parameters(), lr=1e-3, betas=(0.9, 0.98))
    losses = []
    for step in range(3):
        l1 = tr.step(u, seq, rsq, pos, prs, neg, nrs)
        h, pl, nl = m2(u, seq, rsq, pos, prs, neg, nrs)
        l2 = _loss(pl, nl, pos)
        opt.zero_grad()
        l2.backward()
        g0 = {k: p.grad.detach().clone() for k, p in m2.named_parameters()}
        opt.step()
        assert abs(float(l1) - float(l2.detach())) < 1e-5
        losses.append(float(l1))
        if step == 0:
            sd1, sd2 = m1.state_dict(), m2.state_dict()
            for k in sd1:
                d = (sd1[k] - sd2[k]).abs().double().cpu()
                bad = d > adam_tolerance([g0[k].cpu()])
                assert not bool(bad.any()), (k, float(d[bad].max()), int(bad.sum()))
    assert losses[2] < losses[0]


def test_c4_dropout_step_is_replayable_and_masks_depend_on_the_seed(c4):
    """dropout 0.5 at full size: two trainers from the same weights and seed take the same step (dense parameters bit for
    bit: fixed slab tree; the item table up to the float-atomic order), a different seed draws other masks."""
    import srfrd_amd
    m, (u, seq, rsq, pos, prs, neg, nrs) = c4

    def run(seed):
        mm = copy.deepcopy(m).train()
        tr = srfrd_amd.FusedTrainer(mm, B4, L4, seed=seed, use_graph=False)
        loss = float(tr.step(u, seq, rsq, pos, prs, neg, nrs).cpu())
        return loss, {k: v.detach().clone() for k, v in mm.state_dict().items()}

    l1, w1 = run(7)
    l2, w2 = run(7)
    l3, _ = run(8)
    assert l1 == l2 and l1 != l3
    for k in w1:
        if k == "item_emb.weight":
            assert float((w1[k] - w2[k]).abs().max()) <= 2e-3 * 1.1      # an atomic-order ulp can flip an Adam step's sign
        else:
            assert torch.equal(w1[k], w2[k]), k
