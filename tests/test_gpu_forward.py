"""-m gpu: the HIP forward / predict / label kernels against the golden fixtures and the CPU oracle.
Tolerance: 1e-4 absolute on fp32 hidden states and logits (BASELINE.json north_star); integer results bit-exact."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import srfrd_oracle as O
from tests.helpers import GOLDEN, KINDS, golden_cfg, load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _dbg_run(model, ids, seq_idx=0, p=0.0, seed=0):
    from srfrd_amd import _lib
    lay = model.layout
    L = ids[0].shape[1]
    slot, n = C.c_int64(0), C.c_int32(0)
    _lib.lib().srfrd_debug_shape(C.byref(lay), L, C.byref(slot), C.byref(n))
    dbg = torch.zeros(n.value, slot.value, device="cuda")
    out = model._launch_fwd(*ids, p, seed, save=True, dbg=dbg, dbg_seq=seq_idx)
    torch.cuda.synchronize()
    return out, dbg.cpu(), slot.value


@pytest.mark.parametrize("kind", KINDS)
def test_forward_taps_localise(kind):
    """Every intermediate of sequence 0 (x0, and per block LN1, q, k, v, P, h1, LN2, y) against the oracle's taps."""
    from tests.gpu_util import build_model
    g, sd, batch = load_golden(kind)
    cfg = golden_cfg(kind)
    model = build_model(cfg, sd).eval()
    ids = model._prep(batch[0], batch[1], batch[2], batch[3], batch[4], batch[5])
    out, dbg, slot = _dbg_run(model, ids, 0)
    taps = {}
    O.forward(cfg, sd, *batch, taps=taps)
    L, D = batch[0].shape[1], cfg.D

    def got(s, rows, cols):
        return dbg[s, :rows * cols].view(rows, cols)

    errs = {"x0": float((got(0, L, D) - taps["x0"][0]).abs().max())}
    for i in range(cfg.num_blocks):
        b = 1 + 8 * i
        for off, name, cols in ((0, "qn", D), (1, "q", D), (2, "k", D), (3, "v", D), (4, "p", L), (5, "h1", D), (6, "h2", D), (7, "y", D)):
            ref = taps[f"{name}{i}"][0]
            if name == "p":
                ref = ref[0]
            errs[f"{name}{i}"] = float((got(b + off, L, cols) - ref).abs().max())
    bad = {k: v for k, v in errs.items() if not v < TOL}
    assert not bad, f"first mismatching taps: {bad}  (all: {errs})"


@pytest.mark.parametrize("kind", KINDS)
def test_forward_matches_golden(kind):
    from tests.gpu_util import build_model, cuda, maxerr
    g, sd, batch = load_golden(kind)
    model = build_model(golden_cfg(kind), sd).eval()
    with torch.no_grad():
        h, pl, nl = model(None, *cuda(*batch))
    assert tuple(h.shape) == g["hidden"].shape
    assert maxerr(h, torch.from_numpy(g["hidden"])) < TOL
    assert maxerr(pl, torch.from_numpy(g["pos_logits"])) < TOL
    assert maxerr(nl, torch.from_numpy(g["neg_logits"])) < TOL
    # no targets -> logits are None (reference SRFR_model.py:126-136)
    with torch.no_grad():
        h2, a, b = model(None, batch[0].cuda(), batch[1].cuda())
    assert a is None and b is None and maxerr(h2, h) == 0.0


@pytest.mark.parametrize("kind", KINDS)
def test_predict_matches_golden_and_topk_bit_exact(kind):
    from tests.gpu_util import build_model, maxerr
    g, sd, batch = load_golden(kind)
    model = build_model(golden_cfg(kind), sd).eval()
    cands = torch.from_numpy(g["cands"]).cuda()
    ref = torch.from_numpy(g["pred_logits"])
    # batched, per-user candidate lists
    out = model.predict(None, batch[0].cuda(), batch[1].cuda(), cands)
    assert maxerr(out, ref) < TOL
    assert (np.argsort(-out.cpu().numpy(), axis=1, kind="stable")[:, :10]
            == np.argsort(-g["pred_logits"], axis=1, kind="stable")[:, :10]).all()
    # the reference's own call shape: one user, shared (I_c,) candidates -> (I_c,)
    one = model.predict(None, batch[0][2:3].cuda(), batch[1][2:3].cuda(), cands[2])
    assert one.shape == (101,) and maxerr(one, ref[2]) < TOL


def test_user_labels_bit_exact():
    import srfrd_amd
    z = np.load(f"{GOLDEN}/labels_edge.npz")
    edge = torch.from_numpy(z["fake_ids"]).cuda()
    for kind, nl in (("SRFU_B", 3), ("SRFU_F", 11), ("SRFU_R", 11)):
        m = getattr(srfrd_amd, kind)(50, 10, 50, nl, 0.0, 1, 1, "cuda").cuda()
        got = m.get_Labels(edge).cpu().numpy()
        if kind == "SRFU_R":
            assert got[0] == 0 and (got[1:] == z[kind]).all()
        else:
            assert (got == z[kind]).all()
    m = srfrd_amd.SRFRN(50, 10, 45, 5, 0.0, 1, 1, "cuda").cuda()
    assert (m.user_labels(edge).cpu().numpy() == z["SRFRN_predict"]).all()
    with pytest.raises(TypeError):
        srfrd_amd.SRFU(50, 10, 50, 3, 0.0, 1, 1, "cuda").cuda().get_Labels(edge)


def test_eval_rank_and_metric():
    from srfrd_amd import ranks_from_logits
    torch.manual_seed(0)
    logits = torch.randn(300, 101)
    acc = torch.zeros(3, device="cuda", dtype=torch.float64)
    r = ranks_from_logits(logits.cuda(), acc).cpu()
    ro = O.rank_of_first(logits)
    assert (r.long() == ro).all()
    ndcg, hr = O.hr_ndcg_at_10(ro)
    a = acc.cpu()
    assert abs(float(a[0] / a[2]) - ndcg) < 1e-12 and abs(float(a[1] / a[2]) - hr) < 1e-12


def _decile_rows(L):
    """fake/real windows whose fake ratio sits exactly on a decile boundary (n1 / tot = k / 10), plus their neighbours:
    the cases where a division lowered to x * rcp(y) lands one ulp under the integer and the floor drops by one."""
    rows = []
    for tot in range(1, L + 1):
        for n1 in range(0, tot + 1):
            if (10 * n1) % tot == 0 or (10 * n1 + 10) % tot == 0:
                rows.append([0] * (L - tot) + [1] * n1 + [2] * (tot - n1))
    return torch.tensor(rows, dtype=torch.int64)


def test_user_labels_inside_the_encoder_match_get_labels():
    """The label the fused forward / backward kernels derive in place (relaxed floating-point flags in those translation
    units) selects the same ``user_label_embed`` row as ``get_Labels`` and as the reference-pinned oracle: edge matrix of
    labels_edge.npz plus every exact-decile fake ratio up to L = 50, for all three SRFU kinds."""
    import srfrd_amd
    from tests.gpu_util import build_model, maxerr, random_sd
    z = np.load(f"{GOLDEN}/labels_edge.npz")
    edge = torch.from_numpy(z["fake_ids"])
    edge = torch.cat([torch.zeros(edge.shape[0], 40, dtype=torch.int64), edge], dim=1)      # left-pad to L = 50
    rows = torch.cat([edge[1:], _decile_rows(50)])                                             # (row 0 is all-pad: 0/0 in the reference)
    g = torch.Generator().manual_seed(5)
    for kind, nl in (("SRFU_B", 3), ("SRFU_F", 51), ("SRFU_R", 11)):
        cfg = O.Cfg(kind, 80, 50, 50, n_labels=nl)
        sd = random_sd(cfg, seed=11)
        sd["embedding_layer.user_label_embed.weight"] = torch.randn(nl, 50, generator=g)     # rows far apart: a wrong label shows
        model = build_model(cfg, sd).eval()
        want = O.get_labels(kind, rows)
        assert torch.equal(model.get_Labels(rows.cuda()).cpu(), want.to(torch.int64))
        seq = torch.randint(1, 81, rows.shape, generator=g) * (rows != 0)
        h, _, _ = model(None, seq.cuda(), rows.cuda())
        ho, _, _ = O.forward(cfg, sd, seq, rows)
        assert maxerr(h, ho) < TOL, kind
        # and through the backward kernel: the label row is where the user-label gradient lands
        model.train()
        for p in model.parameters():
            p.grad = None
        pos = torch.roll(seq, -1, 1) * (seq != 0)
        neg = torch.randint(1, 81, rows.shape, generator=g) * (seq != 0)
        model.dropout_rate = 0.0
        _, pl, nl_ = model(None, seq.cuda(), rows.cuda(), pos.cuda(), None, neg.cuda(), None)
        (pl.sum() + nl_.sum()).backward()
        got_rows = (model.embedding_layer.user_label_embed.weight.grad.abs().sum(1) > 0).cpu()
        live = (seq != 0).any(1)
        expect_rows = torch.zeros(nl, dtype=torch.bool)
        expect_rows[want[live].long()] = True
        assert torch.equal(got_rows, expect_rows), kind


def test_out_of_range_ids_are_clamped_and_reported():
    """An id outside the embedding tables must neither fault nor corrupt neighbouring parameters; the reference's
    nn.Embedding raises IndexError - lazily here (check_ids()), or at the call with validate_ids = 'eager'."""
    import srfrd_amd
    from tests.gpu_util import build_model, random_sd
    cfg = O.Cfg("SRFRN", 60, 20, 45, d_fake=5)
    sd = random_sd(cfg, seed=2)
    model = build_model(cfg, sd)
    model.train()
    model.dropout_rate = 0.0
    _, seq, rsq, pos, prs, neg, nrs = srfrd_amd.synthetic_batch(60, 20, 16, seed=1, device="cuda")
    before = model.flat_parameters().clone()
    ok = model(None, seq, rsq, pos, prs, neg, nrs)
    model.check_ids()                                               # clean batch: nothing to report
    bad_seq, bad_neg, bad_rsq = seq.clone(), neg.clone(), rsq.clone()
    bad_seq[3, -1] = 61                                             # one past the table
    bad_neg[5, -2] = 2 ** 33 + 7                                    # would alias a valid row if truncated to 32 bits
    bad_neg[6, -1] = -4
    bad_rsq[2, -1] = 3
    _, pl, nl = model(None, bad_seq, bad_rsq, pos, prs, bad_neg, nrs)
    (pl.sum() + nl.sum()).backward()
    torch.cuda.synchronize()
    assert torch.equal(model.flat_parameters(), before)             # nothing was written outside the gradient buffers
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())
    with pytest.raises(IndexError, match="item id.*fake / review id"):
        model.check_ids()
    model.check_ids()                                               # the word was cleared
    model.validate_ids = "eager"
    with pytest.raises(IndexError, match="item id"):
        model(None, bad_seq, rsq, pos, prs, neg, nrs)
    with pytest.raises(IndexError):
        model.predict(None, seq[:1], rsq[:1], torch.tensor([1, 2, 999], device="cuda"))
    model(None, seq, rsq, pos, prs, neg, nrs)                       # a clean call still passes
    # fused trainer: step() validates its inputs; a sampler is checked against the model's table when it is wired
    tr = srfrd_amd.FusedTrainer(model, 16, 20, use_graph=False)
    tr.step(None, seq, rsq, pos, prs, neg, nrs)
    tr.check()
    tr.step(None, bad_seq, rsq, pos, prs, neg, nrs)
    with pytest.raises(IndexError, match="item id"):
        tr.check()
    from srfrd_amd import dataset as DS
    d = DS.partition([1, 1, 1, 2, 2, 2], [5, 70, 9, 3, 4, 5], [False] * 6)          # item 70 > the model's 60
    with pytest.raises(IndexError, match="item ids up to 70"):
        srfrd_amd.DeviceSampler(d, 4, 20, model=tr)
    srfrd_amd.DeviceSampler(d, 4, 20)                               # unwired: no model to check against


def test_custom_ops_pass_opcheck_and_carry_autograd():
    """torch.library.opcheck on the registered ops: schema, fake-tensor (meta) implementation and autograd registration
    of srfrd::encoder_fwd against its real CUDA implementation."""
    import srfrd_amd
    from srfrd_amd import ops
    from tests.gpu_util import build_model, random_sd
    cfg = O.Cfg("SRFRN", 90, 20, 45, d_fake=5)
    model = build_model(cfg, random_sd(cfg, seed=4)).train()
    model.dropout_rate = 0.0
    _, seq, rsq, pos, prs, neg, nrs = srfrd_amd.synthetic_batch(90, 20, 6, seed=2, device="cuda")
    ids = model._prep(seq, rsq, pos, prs, neg, nrs)
    key = ops.register_model(model)
    params = [q for q, _ in model._slots]
    args = (params, *ids, key, 0.0, 0, 0, True)
    torch.library.opcheck(torch.ops.srfrd.encoder_fwd.default, args, test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
    out = torch.ops.srfrd.encoder_fwd(*args)
    assert out[0].requires_grad and out[1].grad_fn is not None
    torch.library.opcheck(torch.ops.srfrd.user_labels.default, (rsq, 2), test_utils=("test_schema", "test_faketensor"))
    h = out[0].detach()
    torch.library.opcheck(torch.ops.srfrd.logits_topk.default, (h, model.user_labels(rsq), key, 0, 91, 5, True),
                          test_utils=("test_schema", "test_faketensor"))
    cand = torch.randint(1, 91, (6, 11), device="cuda")
    torch.library.opcheck(torch.ops.srfrd.predict_logits.default, (h, cand, model.user_labels(rsq), key), test_utils=("test_schema", "test_faketensor"))


@pytest.mark.parametrize("kind,L,B", [("SASRec", 1, 3), ("SRFRN", 5, 1), ("SRFU_R", 16, 3), ("SASRec", 17, 3), ("SRFR", 50, 2)])
def test_edge_shapes_and_ragged_sequences(kind, L, B):
    """Ragged and degenerate inputs: a sequence that is all padding, one with a single real item (left-padded, as the
    reference's sampler produces them), a full one; seq_len 1 / 5 / 16 / 17 (below, at and just past a 16-row tile), batch 1.
    Forward (hidden states, target logits), predict and one fused training step against the oracle."""
    import srfrd_amd
    from tests.gpu_util import build_model, cuda, maxerr, random_sd
    from tests.helpers import assert_post_adam, oracle_step_with_grads
    I = 60
    if kind == "SASRec":
        cfg = O.Cfg(kind, I, L, 50)
    elif kind in ("SRFR", "SRFRN"):
        cfg = O.Cfg(kind, I, L, 45, d_fake=5)
    else:
        cfg = O.Cfg(kind, I, L, 50, n_labels=11)
    sd = random_sd(cfg, 9)
    g = torch.Generator().manual_seed(L * 7 + B)
    seq = torch.randint(1, I + 1, (B, L), generator=g)
    pos = torch.randint(1, I + 1, (B, L), generator=g)
    neg = torch.randint(1, I + 1, (B, L), generator=g)
    rsq, prs, nrs = (torch.randint(1, 3, (B, L), generator=g) for _ in range(3))
    if B >= 2:                                       # sequence 0: nothing but padding
        seq[0] = 0; pos[0] = 0; neg[0] = 0; rsq[0] = 0; prs[0] = 0; nrs[0] = 0
    if B >= 3 and L > 1:                             # sequence 1: one real interaction, left-padded
        seq[1, :-1] = 0; pos[1, :-1] = 0; neg[1, :-1] = 0; rsq[1, :-1] = 0; prs[1, :-1] = 0; nrs[1, :-1] = 0
    batch = (seq, rsq, pos, prs, neg, nrs)
    model = build_model(cfg, {k: v.clone() for k, v in sd.items()}).eval()
    with torch.no_grad():
        h, pl, nl = model(None, *cuda(*batch))
        cand = torch.arange(1, I + 1)
        pr = model.predict(None, *cuda(seq, rsq), cand.cuda())
    ho, plo, nlo = O.forward(cfg, sd, *batch)
    assert maxerr(h, ho) < 1e-4 and maxerr(pl, plo) < 1e-4 and maxerr(nl, nlo) < 1e-4
    assert maxerr(pr.reshape(B, -1), O.predict(cfg, sd, seq, rsq, cand).reshape(B, -1)) < 1e-4
    model.train()
    tr = srfrd_amd.FusedTrainer(model, batch_size=B, seq_len=L, lr=1e-3, betas=(0.9, 0.98), seed=1, use_graph=False)
    opt = O.Adam(sd)
    loss = tr.step(None, *cuda(*batch))
    loss_o, g_o = oracle_step_with_grads(cfg, sd, opt, batch, train=False)
    assert abs(float(loss.cpu()) - float(loss_o)) < 1e-4
    assert_post_adam(model.state_dict(), sd, [g_o], cfg.D)
