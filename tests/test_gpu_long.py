"""-m gpu: sequences too long for the LDS-resident kernels (C4: seq_len 100, C5: seq_len 200) run the global-scratch
build of the same kernel sources; parity against the CPU oracle at 1e-4, plus C4-shaped training parity."""
import pytest
import torch

from oracle import srfrd_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _setup(kind, L, B=6, I=300, seed=0):
    import srfrd_amd
    from tests.gpu_util import build_model, random_sd
    if kind == "SASRec":
        cfg = O.Cfg(kind, I, L, 50)
    elif kind in ("SRFRN", "SRFR"):
        cfg = O.Cfg(kind, I, L, 45, d_fake=5)
    else:
        cfg = O.Cfg(kind, I, L, 50, n_labels=3)
    sd = random_sd(cfg, seed)
    model = build_model(cfg, sd)
    batch = srfrd_amd.synthetic_batch(I, L, B, seed=5, device="cpu")[1:]
    return cfg, sd, model, batch


@pytest.mark.parametrize("kind,L", [("SASRec", 100), ("SRFRN", 100), ("SRFU_B", 128), ("SASRec", 200)])
def test_long_forward_matches_oracle(kind, L):
    from srfrd_amd import _lib
    from tests.gpu_util import cuda, maxerr
    cfg, sd, model, batch = _setup(kind, L)
    if L > 112:
        assert _lib.lds_bytes(model.layout, L)[0] == 0           # really the scratch build
    model.eval()
    with torch.no_grad():
        h, pl, nl = model(None, *cuda(*batch))
    ho, plo, nlo = O.forward(cfg, sd, *batch)
    assert maxerr(h, ho) < TOL and maxerr(pl, plo) < TOL and maxerr(nl, nlo) < TOL


@pytest.mark.parametrize("kind,L", [("SASRec", 100), ("SRFRN", 100), ("SASRec", 200), ("SRFR", 128), ("SRFU_B", 144)])
def test_c4_c5_length_training_matches_oracle(kind, L):
    """seq_len 100 / 200 (BASELINE configs[3] / [4] geometry) and two lengths in between for the other kinds: gradients, loss
    and two fused Adam steps vs the oracle.  The fused step runs the slot-placed backward at 100 and the row-chunked one
    above it (three chunks exactly at 144, a 32-row tail at 128, an 8-row tail at 200)."""
    import srfrd_amd
    from srfrd_amd import _lib
    from tests.gpu_util import cuda, maxerr
    from tests.helpers import drop_kbias
    cfg, sd, model, batch = _setup(kind, L, B=5)
    assert _lib.lds_bytes(model.layout, L)[1] == 0 and _lib.scratch_floats(model.layout, 5, L)[1] > 0
    model.train()                                            # dropout_rate = 0 in cfg
    loss_o, grads_o, *_ = O.grads_of(cfg, sd, batch)
    seq, rsq, pos, prs, neg, nrs = cuda(*batch)
    h, pl, nl = model(None, seq, rsq, pos, prs, neg, nrs)
    idx = torch.where(pos != 0)
    crit = torch.nn.BCEWithLogitsLoss()
    loss = crit(pl[idx], torch.ones_like(pl)[idx]) + crit(nl[idx], torch.zeros_like(nl)[idx])
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_o)) < TOL
    for k, p in model.named_parameters():
        assert maxerr(p.grad, grads_o[k]) < TOL, k
    # fused trainer on the same shape
    model.zero_grad(set_to_none=True)
    tr = srfrd_amd.FusedTrainer(model, 5, L, use_graph=False)
    opt = O.Adam(sd)
    for step in range(2):
        l = tr.step(None, seq, rsq, pos, prs, neg, nrs)
        lo = O.train_step(cfg, sd, opt, batch, train=False)
        assert abs(float(l.cpu()) - float(lo)) < TOL
    msd = model.state_dict()
    for k in sd:
        assert maxerr(drop_kbias(k, msd[k].cpu(), cfg.D), drop_kbias(k, sd[k], cfg.D)) < 3e-4, k


def test_c4_geometry_fused_step_with_dropout_matches_oracle():
    """seq_len 100, hidden 50, SASRec, dropout 0.5, fused step: the compile-time-shaped instantiations of the forward
    (LDS-resident) and of the global-scratch backward vs the oracle's step with the same masks."""
    import srfrd_amd
    from tests.gpu_util import build_model, cuda, random_sd
    from tests.helpers import drop_kbias
    cfg = O.Cfg("SASRec", 300, 100, 50, dropout=0.5)
    sd = random_sd(cfg, 2)
    model = build_model(cfg, {k: v.clone() for k, v in sd.items()}).train()
    B, base = 7, 99
    tr = srfrd_amd.FusedTrainer(model, batch_size=B, seq_len=100, lr=1e-3, betas=(0.9, 0.98), seed=base, use_graph=False)
    from tests.helpers import assert_post_adam, oracle_step_with_grads
    opt = O.Adam(sd)
    hist = []
    for step in range(2):
        batch = srfrd_amd.synthetic_batch(300, 100, B, seed=40 + step, device="cpu")
        loss = tr.step(*cuda(*batch))
        loss_o, g_o = oracle_step_with_grads(cfg, sd, opt, batch[1:], train=True, seed=O.step_seed(base, step + 1), b0=0)
        hist.append(g_o)
        assert abs(float(loss.cpu()) - float(loss_o)) < TOL, step
    assert_post_adam(model.state_dict(), sd, hist, cfg.D)


@pytest.mark.parametrize("kind,L", [("SRFRN", 128), ("SASRec", 200), ("SASRec", 101), ("SRFU_B", 150), ("SRFRN", 207)])
def test_long_fused_step_with_dropout_matches_oracle(kind, L):
    """seq_len > 100 training: the forward (first generation up to 112, the row-owner kernel in its training mode above:
    dropout at the four sites, checkpoints written from the transposed score layout, target logits, loss sums) feeding the
    row-chunked backward - two fused Adam steps with dropout 0.5 against the oracle's steps with the same masks; lengths that
    are not multiples of 4 or 16 and the largest one the kernels take (207: 13 row tiles, five chunks) included."""
    import srfrd_amd
    from tests.gpu_util import build_model, cuda, random_sd
    from tests.helpers import assert_post_adam, oracle_step_with_grads
    I, B, base = 300, 5, 77
    if kind == "SASRec":
        cfg = O.Cfg(kind, I, L, 50, dropout=0.5)
    elif kind == "SRFRN":
        cfg = O.Cfg(kind, I, L, 45, d_fake=5, dropout=0.5)
    else:
        cfg = O.Cfg(kind, I, L, 50, n_labels=3, dropout=0.5)
    sd = random_sd(cfg, 3)
    model = build_model(cfg, {k: v.clone() for k, v in sd.items()}).train()
    tr = srfrd_amd.FusedTrainer(model, batch_size=B, seq_len=L, lr=1e-3, betas=(0.9, 0.98), seed=base, use_graph=False)
    opt = O.Adam(sd)
    hist = []
    for step in range(2):
        batch = srfrd_amd.synthetic_batch(I, L, B, seed=60 + step, device="cpu")
        loss = tr.step(*cuda(*batch))
        loss_o, g_o = oracle_step_with_grads(cfg, sd, opt, batch[1:], train=True, seed=O.step_seed(base, step + 1), b0=0)
        hist.append(g_o)
        assert abs(float(loss.cpu()) - float(loss_o)) < TOL, step
    assert_post_adam(model.state_dict(), sd, hist, cfg.D)


def test_row_owner_forward_at_the_default_geometry(monkeypatch):
    """The row-owner kernel forced onto seq_len 50 (one row tile per wave): eval forward with target logits and a fused
    training step with dropout, against the oracle - the same arithmetic contract as the first-generation kernel."""
    import srfrd_amd
    from tests.gpu_util import build_model, cuda, maxerr, random_sd
    from tests.helpers import assert_post_adam, oracle_step_with_grads
    monkeypatch.setenv("SRFRD_ROWS_ALWAYS", "1")
    I, L, B, base = 300, 50, 6, 5
    cfg = O.Cfg("SRFRN", I, L, 45, d_fake=5, dropout=0.5)
    sd = random_sd(cfg, 8)
    model = build_model(cfg, {k: v.clone() for k, v in sd.items()})
    batch = srfrd_amd.synthetic_batch(I, L, B, seed=9, device="cpu")
    model.eval()
    with torch.no_grad():
        h, pl, nl = model(None, *cuda(*batch[1:]))
    ho, plo, nlo = O.forward(cfg, sd, *batch[1:])
    assert maxerr(h, ho) < TOL and maxerr(pl, plo) < TOL and maxerr(nl, nlo) < TOL
    model.train()
    tr = srfrd_amd.FusedTrainer(model, batch_size=B, seq_len=L, lr=1e-3, betas=(0.9, 0.98), seed=base, use_graph=False)
    opt = O.Adam(sd)
    loss = tr.step(*cuda(*batch))
    loss_o, g_o = oracle_step_with_grads(cfg, sd, opt, batch[1:], train=True, seed=O.step_seed(base, 1), b0=0)
    assert abs(float(loss.cpu()) - float(loss_o)) < TOL
    assert_post_adam(model.state_dict(), sd, [g_o], cfg.D)


def test_fallback_builds_still_agree(monkeypatch):
    """The global-scratch build stays the fallback for what the long-sequence kernels do not cover (other hidden widths,
    debug taps): with the LDS-resident kernels switched off (SRFRD_NO_ROWS / SRFRD_NO_SLOTS) a seq_len-128 SRFRN forward and
    one fused training step must reproduce what the default path computes."""
    import srfrd_amd
    from tests.gpu_util import build_model, cuda, maxerr, random_sd
    I, L, B = 300, 128, 5
    cfg = O.Cfg("SRFRN", I, L, 45, d_fake=5)
    sd = random_sd(cfg, 6)
    batch = srfrd_amd.synthetic_batch(I, L, B, seed=21, device="cpu")

    def run():
        model = build_model(cfg, {k: v.clone() for k, v in sd.items()})
        model.eval()
        with torch.no_grad():
            h = model(None, *cuda(*batch[1:3]))[0]
        model.train()
        tr = srfrd_amd.FusedTrainer(model, batch_size=B, seq_len=L, lr=1e-3, betas=(0.9, 0.98), seed=3, use_graph=False)
        loss = tr.step(*cuda(*batch))
        return h, float(loss.cpu()), {k: v.detach().clone() for k, v in model.state_dict().items()}

    h0, l0, w0 = run()
    monkeypatch.setenv("SRFRD_NO_ROWS", "1")
    monkeypatch.setenv("SRFRD_NO_SLOTS", "1")
    h1, l1, w1 = run()
    assert maxerr(h0, h1.cpu()) < 1e-5 and abs(l0 - l1) < 1e-6
    from tests.helpers import drop_kbias
    for k in w0:
        # one Adam step from identical weights: identical up to the sign of noise-level gradients (lr), 1e-5 in the mean
        d = (drop_kbias(k, w0[k].cpu(), cfg.D) - drop_kbias(k, w1[k].cpu(), cfg.D)).abs()
        assert float(d.max()) <= 2.2e-3 and float(d.mean()) < 2e-5, k
