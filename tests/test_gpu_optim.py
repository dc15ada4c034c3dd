"""-m gpu: srfrd_amd.Adam - torch.optim.Adam's update (reference trainer.py:390) as one launch over the model's flat parameter
vector, for the module-level drop-in loop (trainer.py:29-41 unchanged but for the optimizer's constructor)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(kind="SASRec"):
    import srfrd_amd
    torch.manual_seed(0)
    m = srfrd_amd.SASRec(400, 50, 50, 0.0, 2, 1, "cuda") if kind == "SASRec" else srfrd_amd.SRFRN(400, 50, 45, 5, 0.0, 2, 1, "cuda")
    for _, p in m.named_parameters():
        if p.dim() >= 2:
            torch.nn.init.xavier_normal_(p.data)
    return m.cuda().train()


def _close(sd1, sd2, D=50):
    """same gradients up to float-atomic order in the item table => same Adam steps, except where a gradient is rounding
    noise and its SIGN decides the step (tests/helpers.drop_kbias; a handful of near-zero table elements)"""
    from tests.helpers import drop_kbias
    for k in sd1:
        d = (drop_kbias(k, sd1[k].cpu(), D) - drop_kbias(k, sd2[k].cpu(), D)).abs()
        assert float(d.mean()) < 5e-7 and float((d > 2e-5).float().mean()) < 2e-3 and float(d.max()) < 1e-2, (k, float(d.max()), float(d.mean()))


def _step(m, opt, batch):
    u, seq, rsq, pos, prs, neg, nrs = batch
    h, pl, nl = m(u, seq, rsq, pos, prs, neg, nrs)
    idx = torch.where(pos != 0)
    crit = torch.nn.BCEWithLogitsLoss()
    loss = crit(pl[idx], torch.ones_like(pl)[idx]) + crit(nl[idx], torch.zeros_like(nl)[idx])
    opt.zero_grad()
    loss.backward()
    opt.step()
    return float(loss.detach())


@pytest.mark.parametrize("kind", ["SASRec", "SRFRN"])
def test_flat_adam_equals_torch_adam_and_exchanges_state(kind):
    import srfrd_amd
    m1 = _model(kind)
    m2 = copy.deepcopy(m1)
    o1 = srfrd_amd.Adam(m1.parameters(), lr=1e-3, betas=(0.9, 0.98))
    o2 = torch.optim.Adam(m2.parameters(), lr=1e-3, betas=(0.9, 0.98))
    batches = [srfrd_amd.synthetic_batch(400, 50, 24, seed=3, index=i, device="cuda") for i in range(4)]
    for i in range(3):
        l1, l2 = _step(m1, o1, batches[i]), _step(m2, o2, batches[i])
        assert abs(l1 - l2) < 1e-5
    sd1, sd2 = m1.state_dict(), m2.state_dict()
    _close(sd1, sd2)
    # state in torch.optim.Adam's own format, both directions
    s1, s2 = o1.state_dict(), o2.state_dict()
    assert s1["param_groups"][0]["params"] == s2["param_groups"][0]["params"]
    for i in s2["state"]:
        assert float(s1["state"][i]["step"]) == float(s2["state"][i]["step"]) == 3.0
        assert float((s1["state"][i]["exp_avg"] - s2["state"][i]["exp_avg"]).abs().max()) < 1e-5
        assert float((s1["state"][i]["exp_avg_sq"] - s2["state"][i]["exp_avg_sq"]).abs().max()) < 1e-7
    o3 = srfrd_amd.Adam(m2.parameters(), lr=1e-3, betas=(0.9, 0.98))
    o3.load_state_dict(s2)                         # torch's state into the flat optimizer: the fourth step continues it
    o2b = torch.optim.Adam(m1.parameters(), lr=1e-3, betas=(0.9, 0.98))
    o2b.load_state_dict(s1)                        # ... and the flat optimizer's state into torch's
    l1, l2 = _step(m1, o2b, batches[3]), _step(m2, o3, batches[3])
    assert abs(l1 - l2) < 1e-5
    _close(m1.state_dict(), m2.state_dict())


def test_flat_adam_takes_gradients_that_are_not_views_of_one_vector():
    """two backward calls accumulate into .grad (torch then owns separate tensors): the optimizer gathers them"""
    import srfrd_amd
    m1 = _model()
    m2 = copy.deepcopy(m1)
    o1 = srfrd_amd.Adam(m1.parameters(), lr=1e-3, betas=(0.9, 0.98))
    o2 = torch.optim.Adam(m2.parameters(), lr=1e-3, betas=(0.9, 0.98))
    b = [srfrd_amd.synthetic_batch(400, 50, 16, seed=5, index=i, device="cuda") for i in range(2)]
    for m, o in ((m1, o1), (m2, o2)):
        o.zero_grad()
        for u, seq, rsq, pos, prs, neg, nrs in b:
            h, pl, nl = m(u, seq, rsq, pos, prs, neg, nrs)
            (pl.sum() * 1e-3 - nl.sum() * 1e-3).backward()
        o.step()
    _close(m1.state_dict(), m2.state_dict())


def test_replaced_parameters_and_submodules_are_picked_up_by_the_next_forward():
    """forward() resolves its parameter slots through cached module paths (srfrd_amd/modules.py:_current_slots): a parameter
    object or a whole submodule replaced between two calls must be what the next call computes with - checked against a fresh
    model loaded from the modified model's state_dict (bit-equal outputs), and training must go on from there."""
    import srfrd_amd
    m = _model().eval()
    batch = srfrd_amd.synthetic_batch(400, 50, 32, seed=5, device="cuda")
    u, seq, rsq, pos, prs, neg, nrs = batch
    with torch.no_grad():
        h0 = m(u, seq, rsq, pos, prs, neg, nrs)[0].clone()
        # (a) a new Parameter object in place of an old one
        m.last_layernorm.weight = torch.nn.Parameter(torch.full((50,), 1.5, device="cuda"))
        # (b) a whole submodule replaced
        ln = torch.nn.LayerNorm(50, eps=1e-8).cuda()
        ln.weight.data.uniform_(0.5, 1.5)
        m.attention_layernorms[1] = ln
        # (c) in-place edits of an existing parameter
        m.pos_emb.weight.data.mul_(0.5)
        h1 = m(u, seq, rsq, pos, prs, neg, nrs)[0].clone()
    assert float((h1 - h0).abs().max()) > 1e-3
    ref = _model().eval()
    ref.load_state_dict(m.state_dict())
    with torch.no_grad():
        h2 = ref(u, seq, rsq, pos, prs, neg, nrs)[0]
    assert torch.equal(h1, h2)
    # every parameter is a view of the (re-built) flat vector again, and the optimizer steps the new objects
    m.train()
    opt = srfrd_amd.Adam(m.parameters(), lr=1e-3, betas=(0.9, 0.98))
    before = m.last_layernorm.weight.detach().clone()
    _step(m, opt, batch)
    assert float((m.last_layernorm.weight.detach() - before).abs().max()) > 0
    base = m.flat_parameters().data_ptr()
    assert all(base <= p.data_ptr() < base + 4 * m.flat_parameters().numel() for p in m.parameters())
