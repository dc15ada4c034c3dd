"""The CPU oracle against fixtures generated from the reference's own classes (tests/golden/make_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import srfrd_oracle as O
from tests.helpers import GOLDEN, HEAD_CASES, KINDS, drop_kbias, golden_cfg, load_golden, sub

TOL = 2e-6
CASES = [(k, 1) for k in KINDS] + list(HEAD_CASES)        # (class, num_heads)


@pytest.mark.parametrize("kind,heads", CASES)
def test_forward_matches_reference(kind, heads):
    g, sd, batch = load_golden(kind, heads)
    cfg = golden_cfg(kind, heads=heads)
    h, pl, nl = O.forward(cfg, sd, *batch)
    assert h.shape == g["hidden"].shape
    np.testing.assert_allclose(h.numpy(), g["hidden"], atol=TOL, rtol=0)
    np.testing.assert_allclose(pl.numpy(), g["pos_logits"], atol=TOL, rtol=0)
    np.testing.assert_allclose(nl.numpy(), g["neg_logits"], atol=TOL, rtol=0)


@pytest.mark.parametrize("kind,heads", CASES)
def test_predict_matches_reference(kind, heads):
    g, sd, batch = load_golden(kind, heads)
    cfg = golden_cfg(kind, heads=heads)
    out = O.predict(cfg, sd, batch[0], batch[1], torch.from_numpy(g["cands"]))
    np.testing.assert_allclose(out.numpy(), g["pred_logits"], atol=TOL, rtol=0)
    # top-10 indices bit-exact against the reference's logits
    ref_top = np.argsort(-g["pred_logits"], axis=1, kind="stable")[:, :10]
    my_top = np.argsort(-out.numpy(), axis=1, kind="stable")[:, :10]
    assert (ref_top == my_top).all()


@pytest.mark.parametrize("kind,heads", CASES)
def test_train_step_matches_reference(kind, heads):
    g, sd, batch = load_golden(kind, heads)
    cfg = golden_cfg(kind, heads=heads)
    loss, grads, *_ = O.grads_of(cfg, sd, batch)
    assert abs(float(loss) - float(g["loss0"])) < TOL
    gg = sub(g, "g/")
    assert set(gg) == set(grads)
    for k in gg:
        np.testing.assert_allclose(grads[k].numpy(), gg[k].numpy(), atol=TOL, rtol=0, err_msg=k)
    opt = O.Adam(sd)
    w1, w3 = sub(g, "w1/"), sub(g, "w3/")
    for step in range(3):
        loss = O.train_step(cfg, sd, opt, batch, train=False)
        assert abs(float(loss) - float(g[f"loss{step}"])) < 5e-6
        if step == 0:
            for k in w1:
                np.testing.assert_allclose(drop_kbias(k, sd[k], cfg.D).numpy(), drop_kbias(k, w1[k], cfg.D).numpy(),
                                           atol=2e-5, rtol=0, err_msg=k)
    for k in w3:
        np.testing.assert_allclose(drop_kbias(k, sd[k], cfg.D).numpy(), drop_kbias(k, w3[k], cfg.D).numpy(),
                                   atol=1e-4, rtol=0, err_msg=k)


def test_get_labels_bit_exact():
    z = np.load(f"{GOLDEN}/labels_edge.npz")
    edge = torch.from_numpy(z["fake_ids"])
    assert (O.get_labels("SRFU_B", edge).numpy() == z["SRFU_B"]).all()
    assert (O.get_labels("SRFU_F", edge).numpy() == z["SRFU_F"]).all()
    assert (O.get_labels("SRFU_R", edge[1:]).numpy() == z["SRFU_R"]).all()
    assert int(O.get_labels("SRFU_R", edge[:1])[0]) == 0          # guarded all-pad row
    assert (O.srfrn_predict_label(edge).numpy() == z["SRFRN_predict"]).all()
    for kind in ("SRFU_B", "SRFU_F", "SRFU_R"):
        g, _, batch = load_golden(kind)
        assert (O.get_labels(kind, batch[1]).numpy() == g["labels"]).all()


def test_keep_mask_statistics_and_determinism():
    m1 = O.keep_mask(123, 4, 0, 16, 50, 50, 0.5)
    m2 = O.keep_mask(123, 4, 0, 16, 50, 50, 0.5)
    assert torch.equal(m1, m2)
    frac = float((m1 > 0).float().mean())
    assert abs(frac - 0.5) < 0.02
    assert set(m1.unique().tolist()) == {0.0, 2.0}
    # shifting the global sequence index shifts the mask
    m3 = O.keep_mask(123, 4, 8, 8, 50, 50, 0.5)
    assert torch.equal(m3, m1[8:])
    m4 = O.keep_mask(123, 4, 0, 4, 20, 20, 0.25)
    assert abs(float((m4 > 0).float().mean()) - 0.75) < 0.05


def test_rank_metric():
    logits = torch.tensor([[0.5, 0.1, 0.9, 0.4], [2.0, 0.1, 0.9, 0.4]])
    r = O.rank_of_first(logits)
    assert r.tolist() == [1, 0]
    assert r.tolist() == [int((-logits[i]).argsort().argsort()[0]) for i in range(2)]
    ndcg, hr = O.hr_ndcg_at_10(r)
    assert hr == 1.0 and abs(ndcg - (1 / np.log2(3) + 1) / 2) < 1e-12


def test_l2_emb_train_step_matches_reference():
    """reference trainer.py:39 with config.l2_emb = 0.05: loss += l2_emb * torch.norm(p) for every parameter tensor
    (tests/golden/SRFRN_l2.npz, make_golden.py --l2): loss curve, gradients, weights after 1 and 3 Adam steps."""
    g, sd, batch = load_golden("SRFRN", l2=True)
    cfg = golden_cfg("SRFRN")
    l2 = float(g["l2_emb"])
    assert l2 == 0.05
    loss, grads, *_ = O.grads_of(cfg, sd, batch, l2_emb=l2)
    assert abs(float(loss) - float(g["loss0"])) < 5e-6
    gg = sub(g, "g/")
    for k in gg:
        np.testing.assert_allclose(grads[k].numpy(), gg[k].numpy(), atol=TOL, rtol=0, err_msg=k)
    opt = O.Adam(sd)
    w1, w3 = sub(g, "w1/"), sub(g, "w3/")
    for step in range(3):
        loss = O.train_step(cfg, sd, opt, batch, train=False, l2_emb=l2)
        assert abs(float(loss) - float(g[f"loss{step}"])) < 1e-5
        if step == 0:
            for k in w1:
                np.testing.assert_allclose(sd[k].numpy(), w1[k].numpy(), atol=2e-5, rtol=0, err_msg=k)
    for k in w3:
        np.testing.assert_allclose(sd[k].numpy(), w3[k].numpy(), atol=1e-4, rtol=0, err_msg=k)
