"""-m gpu: the bf16 item-table shadow (BASELINE configs[1] / [4] say "bf16"; the reference's table is fp32, so this is the
build's variant): gathers read bf16(E) - half the bytes -, the fp32 master, its gradient and Adam are unchanged.  Parity
is like for like: the oracle with ``table_bf16`` rounds the table the same way (nearest even, straight-through gradient),
so the bar stays 1e-4 on outputs / loss / gradients and bit-exact on top-k indices."""
import numpy as np
import pytest
import torch

from oracle import srfrd_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _cfg(kind, L=50, I=400, dropout=0.0, bf16=True):
    if kind == "SASRec":
        return O.Cfg(kind, I, L, 50, dropout=dropout, table_bf16=bf16)
    if kind == "SRFRN":
        return O.Cfg(kind, I, L, 45, d_fake=5, dropout=dropout, table_bf16=bf16)
    return O.Cfg(kind, I, L, 50, n_labels=3, dropout=dropout, table_bf16=bf16)


@pytest.mark.parametrize("kind,L", [("SASRec", 50), ("SRFRN", 50), ("SRFU_B", 20)])
def test_bf16_table_forward_grads_and_topk_match_the_bf16_oracle(kind, L):
    import srfrd_amd
    from tests.gpu_util import build_model, cuda, maxerr, random_sd
    cfg = _cfg(kind, L)
    sd = random_sd(cfg, 7)
    model = build_model(cfg, sd).train().use_bf16_table()
    assert model.bf16_table
    batch = srfrd_amd.synthetic_batch(400, L, 9, seed=6, device="cpu")[1:]
    loss_o, grads_o, h_o, pl_o, nl_o = O.grads_of(cfg, sd, batch)
    seq, rsq, pos, prs, neg, nrs = cuda(*batch)
    h, pl, nl = model(None, seq, rsq, pos, prs, neg, nrs)
    assert maxerr(h, h_o) < TOL and maxerr(pl, pl_o) < TOL and maxerr(nl, nl_o) < TOL
    # the rounding is visible: the fp32 oracle differs by more than the tolerance
    _, _, h_f, pl_f, _ = O.grads_of(_cfg(kind, L, bf16=False), sd, batch)
    assert maxerr(pl, pl_f) > 10 * TOL
    idx = torch.where(pos != 0)
    crit = torch.nn.BCEWithLogitsLoss()
    loss = crit(pl[idx], torch.ones_like(pl)[idx]) + crit(nl[idx], torch.zeros_like(nl)[idx])
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_o)) < TOL
    for k, p in model.named_parameters():
        assert maxerr(p.grad, grads_o[k]) < TOL, k
    # ranking over the shadow: indices bit-exact against the bf16 oracle's logits
    model.eval()
    idx10, val10 = model.topk(None, seq, rsq, k=10)
    ref = O.predict(cfg, sd, batch[0], batch[1], torch.arange(1, 401))
    order = np.argsort(-ref.numpy(), axis=1, kind="stable")[:, :10]
    assert (idx10.cpu().numpy() == order + 1).all()
    cand = torch.randint(1, 401, (9, 31))
    assert maxerr(model.predict(None, seq, rsq, cand.cuda()), O.predict(cfg, sd, batch[0], batch[1], cand)) < TOL


@pytest.mark.parametrize("graph", [False, True])
def test_bf16_table_fused_training_keeps_the_shadow_current(graph):
    """FusedTrainer on the bf16 shadow, dropout on: per-step loss vs the bf16 oracle, element-wise Adam tolerance on the fp32
    master, and the shadow the optimizer wrote == bf16(master) bit for bit after every step."""
    import srfrd_amd
    from tests.gpu_util import build_model, cuda, random_sd
    from tests.helpers import assert_post_adam, oracle_step_with_grads
    cfg = _cfg("SASRec", dropout=0.5)
    sd = random_sd(cfg, 9)
    model = build_model(cfg, {k: v.clone() for k, v in sd.items()}).train().use_bf16_table()
    B, base = 12, 77
    tr = srfrd_amd.FusedTrainer(model, batch_size=B, seq_len=50, seed=base, use_graph=graph)
    opt = O.Adam(sd)
    hist = []
    for step in range(2):
        batch = srfrd_amd.synthetic_batch(400, 50, B, seed=50 + step, device="cpu")
        loss = tr.step(*cuda(*batch))
        loss_o, g_o = oracle_step_with_grads(cfg, sd, opt, batch[1:], train=True, seed=O.step_seed(base, step + 1), b0=0)
        hist.append(g_o)
        assert abs(float(loss.cpu()) - float(loss_o)) < TOL, step
        table = model.item_emb.weight.detach()
        want = table.to(torch.bfloat16).view(torch.int16).flatten()
        assert torch.equal(model._table16, want)
    assert_post_adam(model.state_dict(), sd, hist, cfg.D)
    # switching the shadow off returns to fp32 gathers
    model.use_bf16_table(False)
    assert not model.bf16_table


@pytest.mark.parametrize("hidden", [64, 54, 51])
def test_wide_bf16_rows_rank_like_fp64_on_a_large_catalog(hidden):
    """hidden 54..64 over the bf16 shadow on a catalog large enough for the 512-row chunks of the streamed ranking (ADVICE
    round 2): a 512-row chunk of such rows does not fit the staging registers (13 x 1024 copy slots), the launcher must
    take 256-row chunks - with 512 the rows past slot 13312 of every chunk were never copied and stale rows were ranked.
    51 = the widest odd width (2-byte copies, 256-row chunks).  Property check against torch.topk in fp64 on the GPU, as
    test_c5_size_one_million_items_property does (the oracle cannot rank 320k items x 256 users in seconds)."""
    import srfrd_amd
    torch.manual_seed(3)
    I, L, B, k = 320_000, 20, 256, 10
    m = srfrd_amd.SASRec(I, L, hidden, 0.0, 2, 1, "cuda")
    for _, p in m.named_parameters():
        if p.dim() >= 2:
            torch.nn.init.xavier_normal_(p.data)
    m = m.cuda().eval()
    with torch.no_grad():
        m.item_emb.weight.mul_(30.0)
    m.use_bf16_table()
    _, seq, rsq, *_ = srfrd_amd.synthetic_batch(I, L, B, seed=8, device="cuda")
    idx, val = m.topk(None, seq, None, k=k)
    with torch.no_grad():
        h = m(None, seq, None)[0][:, -1].double()
        scores = h @ m.item_emb.weight.to(torch.bfloat16).double().T
        scores[:, 0] = -float("inf")
        tv, ti = torch.topk(scores, k + 1, dim=1)
    assert float((val.double() - tv[:, :k]).abs().max()) < 1e-4
    safe = ((tv[:, :-1] - tv[:, 1:]).abs() > 1e-5).all(dim=1)
    assert int(safe.sum()) > B // 2
    assert torch.equal(idx[safe], ti[safe][:, :k])
