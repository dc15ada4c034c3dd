"""-m gpu: backward, Adam and the fused train step against the golden fixtures and the CPU oracle.
Tolerances: gradients / loss 1e-4 absolute (fp32; observed ~1e-6); post-Adam weights 1e-4 with the K-bias slice
excluded (tests/helpers.drop_kbias explains why that slice is rounding noise in any implementation)."""
import numpy as np
import pytest
import torch

from oracle import srfrd_oracle as O
from tests.helpers import KINDS, drop_kbias, golden_cfg, load_golden, sub

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _loss(pl, nl, pos):
    idx = torch.where(pos != 0)
    crit = torch.nn.BCEWithLogitsLoss()
    return crit(pl[idx], torch.ones_like(pl)[idx]) + crit(nl[idx], torch.zeros_like(nl)[idx])


@pytest.mark.parametrize("kind", KINDS)
def test_autograd_grads_match_golden(kind):
    """reference trainer.py:35-40 driven through the drop-in module + torch autograd."""
    from tests.gpu_util import build_model, cuda, maxerr
    g, sd, batch = load_golden(kind)
    cfg = golden_cfg(kind)
    model = build_model(cfg, sd).train()            # dropout_rate = 0 in the golden config
    seq, rsq, pos, prs, neg, nrs = cuda(*batch)
    h, pl, nl = model(user_ids=None, input_ids=seq, fake_ids=rsq, positive_ids=pos, positive_fake_ids=prs,
                      negative_ids=neg, negative_fake_ids=nrs)
    loss = _loss(pl, nl, pos)
    for p in model.parameters():                      # trainer.py:39 with l2_emb = 0.0
        loss = loss + 0.0 * torch.norm(p)
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss0"])) < TOL
    gg = sub(g, "g/")
    errs = {k: maxerr(p.grad, gg[k]) for k, p in model.named_parameters()}
    bad = {k: v for k, v in errs.items() if not v < TOL}
    assert not bad, f"gradient mismatch: {bad}"
    assert set(errs) == set(gg)


@pytest.mark.parametrize("kind", ["SASRec", "SRFRN"])
def test_autograd_upstream_hidden_grad(kind):
    """a loss on hidden_state itself (d_hidden path of the backward kernel) against oracle autograd."""
    from tests.gpu_util import build_model, cuda, maxerr
    g, sd, batch = load_golden(kind)
    cfg = golden_cfg(kind)
    model = build_model(cfg, sd).train()
    seq, rsq, pos, prs, neg, nrs = cuda(*batch)
    w = torch.randn(8, 20, cfg.d_out, generator=torch.Generator().manual_seed(1))
    h, pl, nl = model(None, seq, rsq, pos, prs, neg, nrs)
    ((h * w.cuda()).sum() + 0.5 * pl.sum() - 0.25 * nl.sum()).backward()
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ho, plo, nlo = O.forward(cfg, leaves, *batch)
    ((ho * w).sum() + 0.5 * plo.sum() - 0.25 * nlo.sum()).backward()
    for k, p in model.named_parameters():
        ref = leaves[k].grad if leaves[k].grad is not None else torch.zeros_like(leaves[k])
        if k.endswith("item_embed.weight") or k == "item_emb.weight" or k.endswith("fake_embed.weight"):
            ref = ref.clone()
            ref[0] = 0            # padding_idx rows
        assert maxerr(p.grad, ref) < 5e-4, k


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("graph", [False, True])
def test_fused_trainer_matches_golden(kind, graph):
    """3 fused steps (fwd + BCE + bwd + Adam) vs the reference's loss curve and post-step weights."""
    import srfrd_amd
    from tests.gpu_util import build_model, cuda, maxerr
    g, sd, batch = load_golden(kind)
    cfg = golden_cfg(kind)
    model = build_model(cfg, sd).train()
    tr = srfrd_amd.FusedTrainer(model, 8, 20, lr=1e-3, betas=(0.9, 0.98), use_graph=graph)
    seq, rsq, pos, prs, neg, nrs = cuda(*batch)
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


@pytest.mark.parametrize("kind", ["SASRec", "SRFR", "SRFRN", "SRFU_B"])
def test_dropout_train_mode_matches_oracle_masks(kind):
    """p = 0.5: forward outputs and every gradient with the coordinate-hash masks the oracle rebuilds."""
    from tests.gpu_util import build_model, cuda, maxerr
    g, sd, batch = load_golden(kind)
    cfg = golden_cfg(kind, dropout=0.5)
    model = build_model(cfg, sd).train()
    ids = model._prep(*cuda(*batch))
    seed, seq0 = 0xC0FFEE, 40
    out = model._launch_fwd(*ids, 0.5, seed, save=True, seq0=seq0)
    ho, plo, nlo = O.forward(cfg, sd, *batch, train=True, seed=seed, b0=seq0)
    assert maxerr(out["hidden"], ho) < TOL and maxerr(out["pos_logits"], plo) < TOL and maxerr(out["neg_logits"], nlo) < TOL
    # gradients of  sum(0.3 * pos_logits - 0.2 * neg_logits)
    dpl = torch.full_like(out["pos_logits"], 0.3)
    dnl = torch.full_like(out["neg_logits"], -0.2)
    gflat = model._launch_bwd(*ids, 0.5, seed, out, None, dpl, dnl, seq0=seq0)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    h2, p2, n2 = O.forward(cfg, leaves, *batch, train=True, seed=seed, b0=seq0)
    (0.3 * p2.sum() - 0.2 * n2.sum()).backward()
    named = dict(model.named_parameters())
    for k, p in named.items():
        off = next(o for q, o in model._slots if q is p)
        got = gflat[off:off + p.numel()].view(p.shape)
        ref = leaves[k].grad if leaves[k].grad is not None else torch.zeros_like(leaves[k])
        if k.endswith("item_embed.weight") or k == "item_emb.weight" or k.endswith("fake_embed.weight"):
            ref = ref.clone()
            ref[0] = 0
        assert maxerr(got, ref) < 2e-4, k


def test_dropout_statistics_and_backward_consistency():
    """keep rate ~ 1 - p, and two forwards with the same seed agree bit-for-bit while another seed differs."""
    from tests.gpu_util import build_model, cuda
    g, sd, batch = load_golden("SASRec")
    cfg = golden_cfg("SASRec", dropout=0.5)
    model = build_model(cfg, sd).train()
    ids = model._prep(*cuda(*batch))
    a = model._launch_fwd(*ids, 0.5, 11, save=False)["hidden"]
    b = model._launch_fwd(*ids, 0.5, 11, save=False)["hidden"]
    c = model._launch_fwd(*ids, 0.5, 12, save=False)["hidden"]
    assert torch.equal(a, b) and not torch.equal(a, c)


@pytest.mark.parametrize("kind,d_item,d_fake,L", [("SASRec", 64, 0, 20), ("SRFR", 43, 5, 33), ("SRFRN", 50, 10, 17),
                                                  ("SRFU_R", 32, 0, 48), ("SASRec", 16, 0, 5), ("SASRec", 48, 0, 64),
                                                  ("SRFRN", 54, 10, 60)])
def test_other_widths_and_lengths_match_oracle(kind, d_item, d_fake, L):
    """Generic (run-time geometry) kernels: widths that are / are not multiples of 16 (bias-gradient folding on and off),
    the reference constructors' default 50 + 10, odd lengths; forward, loss and every gradient vs the oracle."""
    import srfrd_amd
    from tests.gpu_util import build_model, cuda, maxerr, random_sd
    I, B = 200, 7
    if kind == "SASRec":
        cfg = O.Cfg(kind, I, L, d_item)
    elif kind in ("SRFR", "SRFRN"):
        cfg = O.Cfg(kind, I, L, d_item, d_fake=d_fake)
    else:
        cfg = O.Cfg(kind, I, L, d_item, n_labels=11)
    sd = random_sd(cfg, 3)
    model = build_model(cfg, sd).train()
    batch = srfrd_amd.synthetic_batch(I, L, B, seed=9, device="cpu", min_len=1)[1:]
    loss_o, grads_o, h_o, pl_o, nl_o = O.grads_of(cfg, sd, batch)
    seq, rsq, pos, prs, neg, nrs = cuda(*batch)
    h, pl, nl = model(None, seq, rsq, pos, prs, neg, nrs)
    assert maxerr(h, h_o) < TOL and maxerr(pl, pl_o) < TOL and maxerr(nl, nl_o) < TOL
    loss = _loss(pl, nl, pos)
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_o)) < TOL
    for k, p in model.named_parameters():
        assert maxerr(p.grad, grads_o[k]) < TOL, k


@pytest.mark.parametrize("kind", ["SASRec", "SRFRN"])
def test_generic_kernels_on_the_specialised_shape(kind, monkeypatch):
    """SRFRD_GENERIC=1 routes hidden-50 shapes through the run-time-geometry instantiation: same gradients as the
    oracle at the bench geometry (L = 50: an [L][D] block is larger than one copy chunk of the generic kernel)."""
    import srfrd_amd
    from tests.gpu_util import build_model, cuda, maxerr, random_sd
    monkeypatch.setenv("SRFRD_GENERIC", "1")
    I, B, L = 300, 9, 50
    cfg = O.Cfg(kind, I, L, 50) if kind == "SASRec" else O.Cfg(kind, I, L, 45, d_fake=5)
    sd = random_sd(cfg, 4)
    model = build_model(cfg, sd).train()
    batch = srfrd_amd.synthetic_batch(I, L, B, seed=2, device="cpu", min_len=1)[1:]
    loss_o, grads_o, h_o, pl_o, nl_o = O.grads_of(cfg, sd, batch)
    seq, rsq, pos, prs, neg, nrs = cuda(*batch)
    h, pl, nl = model(None, seq, rsq, pos, prs, neg, nrs)
    assert maxerr(h, h_o) < TOL
    loss = _loss(pl, nl, pos)
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_o)) < TOL
    for k, p in model.named_parameters():
        assert maxerr(p.grad, grads_o[k]) < TOL, k


def test_fused_tail_keeps_packed_weights_in_sync():
    """srfrd_adam_pack_step writes the stepped weights into both fragment forms: after two fused steps the trainer's
    packed buffer equals a fresh srfrd_pack_weights of the stepped parameters bit for bit (SRFR: last_conv included)."""
    import srfrd_amd
    from tests.gpu_util import build_model, cuda, random_sd
    cfg = O.Cfg("SRFR", 150, 24, 43, d_fake=5)
    model = build_model(cfg, random_sd(cfg, 1)).train()
    B = 6
    tr = srfrd_amd.FusedTrainer(model, batch_size=B, seq_len=24, lr=1e-3, use_graph=False)
    batch = cuda(*srfrd_amd.synthetic_batch(150, 24, B, seed=4, device="cpu"))
    for _ in range(2):
        tr.step(*batch)
    torch.cuda.synchronize()
    fused = tr.packed.clone()
    fresh = model.pack_weights().clone()          # re-packs from the (stepped) flat parameters
    assert torch.equal(fused, fresh)
    assert int(tr.state[0]) == 3 and int(tr.state[6]) == 0      # state advanced once per step, ticket counter back at 0


def _cfg50(kind, dropout=0.0):
    if kind == "SASRec":
        return O.Cfg(kind, 400, 50, 50, dropout=dropout)
    if kind in ("SRFR", "SRFRN"):
        return O.Cfg(kind, 400, 50, 45, d_fake=5, dropout=dropout)      # the reference trainer's default 45 + 5
    return O.Cfg(kind, 400, 50, 50, n_labels=7, dropout=dropout)


@pytest.mark.parametrize("kind", ["SASRec", "SRFR", "SRFRN", "SRFU_B", "SRFU_R"])
def test_bench_geometry_autograd_matches_oracle(kind):
    """seq_len 50, hidden 50: the instantiations specialised on length, width split and kind (inference-shaped variant:
    dropout off, upstream gradients through autograd) - outputs, loss and every gradient vs the oracle."""
    import srfrd_amd
    from tests.gpu_util import build_model, cuda, maxerr, random_sd
    cfg = _cfg50(kind)
    sd = random_sd(cfg, 6)
    model = build_model(cfg, sd).train()
    batch = srfrd_amd.synthetic_batch(400, 50, 10, seed=12, device="cpu")[1:]
    loss_o, grads_o, h_o, pl_o, nl_o = O.grads_of(cfg, sd, batch)
    seq, rsq, pos, prs, neg, nrs = cuda(*batch)
    h, pl, nl = model(None, seq, rsq, pos, prs, neg, nrs)
    assert maxerr(h, h_o) < TOL and maxerr(pl, pl_o) < TOL and maxerr(nl, nl_o) < TOL
    loss = _loss(pl, nl, pos)
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_o)) < TOL
    for k, p in model.named_parameters():
        assert maxerr(p.grad, grads_o[k]) < TOL, k


@pytest.mark.parametrize("first_gen", [False, True])
@pytest.mark.parametrize("kind", ["SASRec", "SRFR", "SRFRN", "SRFU_B"])
def test_bench_geometry_fused_step_with_dropout_matches_oracle(kind, first_gen, monkeypatch):
    """The train-mode instantiations (fused BCE, dropout 0.5, checkpoints, loss sums): two FusedTrainer steps at
    seq_len 50 / hidden 50 vs the oracle's step with the same coordinate-hash masks (step seeds from the same base).
    The backward is the slot-placed kernel (two workgroups per CU); first_gen: SRFRD_NO_SLOTS50 selects the first-generation
    backward on the same grid and slabs."""
    import srfrd_amd
    if first_gen:
        monkeypatch.setenv("SRFRD_NO_SLOTS50", "1")
    from tests.gpu_util import build_model, cuda, maxerr, random_sd
    from tests.helpers import drop_kbias
    cfg = _cfg50(kind, dropout=0.5)
    sd = random_sd(cfg, 8)
    model = build_model(cfg, {k: v.clone() for k, v in sd.items()}).train()
    B, base = 12, 1234
    tr = srfrd_amd.FusedTrainer(model, batch_size=B, seq_len=50, lr=1e-3, betas=(0.9, 0.98), seed=base, use_graph=False)
    from tests.helpers import assert_post_adam, oracle_step_with_grads
    opt = O.Adam(sd)
    hist = []
    for step in range(2):
        batch = srfrd_amd.synthetic_batch(400, 50, B, seed=20 + step, device="cpu")
        loss = tr.step(*cuda(*batch))
        loss_o, g_o = oracle_step_with_grads(cfg, sd, opt, batch[1:], train=True, seed=O.step_seed(base, step + 1), b0=0)
        hist.append(g_o)
        assert abs(float(loss.cpu()) - float(loss_o)) < TOL, step
    # element-wise: 1e-4 or tighter wherever the gradient is real, up to steps * lr only where it is rounding noise
    frac = assert_post_adam(model.state_dict(), sd, hist, cfg.D)
    assert frac > 0.3, frac       # (share of elements held to 1e-4 or tighter)


@pytest.mark.parametrize("first_gen", [False, True])
@pytest.mark.parametrize("kind", ["SASRec", "SRFRN"])
def test_more_sequences_than_workgroups_accumulate_in_the_slabs(kind, first_gen, monkeypatch):
    """B = 1100 > 2 x (2 x 256 CUs): persistent backward workgroups process up to three sequences each, so the dense
    gradients go through the slab read-modify-write (old tile values as the MFMA accumulators' initial value) twice - in
    the slot-placed kernel's read-modify-write instantiation, and (first_gen) in the first-generation kernel.
    Gradients (autograd path) and one fused dropout step vs the oracle."""
    import srfrd_amd
    if first_gen:
        monkeypatch.setenv("SRFRD_NO_SLOTS50", "1")
    from tests.gpu_util import build_model, cuda, maxerr, random_sd
    B = 1100
    cfg = _cfg50(kind)
    sd = random_sd(cfg, 11)
    model = build_model(cfg, sd).train()
    batch = srfrd_amd.synthetic_batch(400, 50, B, seed=3, device="cpu")[1:]
    loss_o, grads_o, h_o, pl_o, nl_o = O.grads_of(cfg, sd, batch)
    seq, rsq, pos, prs, neg, nrs = cuda(*batch)
    h, pl, nl = model(None, seq, rsq, pos, prs, neg, nrs)
    loss = _loss(pl, nl, pos)
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_o)) < TOL
    for k, p in model.named_parameters():
        assert maxerr(p.grad, grads_o[k]) < TOL, k
    # fused step with dropout (train-mode instantiation), same batch size
    cfg_d = _cfg50(kind, dropout=0.5)
    sd_d = random_sd(cfg_d, 12)
    model_d = build_model(cfg_d, {k: v.clone() for k, v in sd_d.items()}).train()
    tr = srfrd_amd.FusedTrainer(model_d, batch_size=B, seq_len=50, lr=1e-3, betas=(0.9, 0.98), seed=5, use_graph=False)
    full = srfrd_amd.synthetic_batch(400, 50, B, seed=4, device="cpu")
    loss_f = tr.step(*cuda(*full))
    from tests.helpers import assert_post_adam, oracle_step_with_grads
    opt = O.Adam(sd_d)
    loss_fo, g_o = oracle_step_with_grads(cfg_d, sd_d, opt, full[1:], train=True, seed=O.step_seed(5, 1), b0=0)
    assert abs(float(loss_f.cpu()) - float(loss_fo)) < TOL
    assert_post_adam(model_d.state_dict(), sd_d, [g_o], cfg_d.D)


def test_input_ring_slots_match_copy_in_steps():
    """step_slot(k) on a pre-filled ring == step_packed(batch_k): same losses and weights after four graph-replayed steps."""
    import srfrd_amd
    from tests.gpu_util import build_model, random_sd
    cfg = _cfg50("SASRec", dropout=0.5)
    sd = random_sd(cfg, 3)
    B = 9
    batches = [srfrd_amd.synthetic_batch(400, 50, B, seed=7, index=i, device="cuda", packed=True)[1] for i in range(3)]
    outs = []
    for ring in (1, 3):
        model = build_model(cfg, {k: v.clone() for k, v in sd.items()}).train()
        tr = srfrd_amd.FusedTrainer(model, batch_size=B, seq_len=50, seed=11, use_graph=True, slots=ring)
        if ring == 3:
            for i in range(3):
                tr.ids_ring[i].copy_(batches[i])
        losses = []
        for i in range(4):
            loss = tr.step_slot(i % 3) if ring == 3 else tr.step_packed(batches[i % 3])
            losses.append(float(loss.cpu()))
        outs.append((losses, {k: v.detach().clone() for k, v in model.state_dict().items()}))
    # (not bit-equal: the item-table scatter adds with float atomics, so two runs differ in the last bits)
    assert max(abs(a - b) for a, b in zip(outs[0][0], outs[1][0])) < 1e-5
    from tests.helpers import drop_kbias
    for k in outs[0][1]:                    # (K-bias: its gradient is rounding noise, which Adam normalises to +-lr)
        d = (drop_kbias(k, outs[0][1][k].cpu(), cfg.D) - drop_kbias(k, outs[1][1][k].cpu(), cfg.D)).abs()
        assert float(d.max()) < 1e-4 and float(d.mean()) < 1e-6, k


def test_unsupported_geometry_is_rejected():
    """Hidden width > 64 is outside the fused kernels: the module must refuse on the device (no silent fallback), both at the
    first forward and when a FusedTrainer is built on it."""
    import srfrd_amd
    ids = torch.ones(2, 5, dtype=torch.int64, device="cuda")
    for m in (srfrd_amd.SASRec(10, 5, 128, 0.0, 1, 1, "cuda").cuda(), srfrd_amd.SASRec(10, 5, 96, 0.0, 1, 2, "cuda").cuda()):
        with pytest.raises(NotImplementedError, match="hidden width <= 64"):
            m(None, ids, ids, ids, ids, ids, ids)
        with pytest.raises(NotImplementedError):
            m.flat_parameters()
        with pytest.raises(NotImplementedError):
            srfrd_amd.FusedTrainer(m, 2, 5)


def test_deterministic_table_scatter_is_bitwise_repeatable():
    """FusedTrainer(deterministic=True): item-table gradient by stable sort + ordered per-item sums instead of float atomics.
    Two runs (graph replay, dropout on, popular items shared by many sequences) end bit-identical; the atomic form agrees
    with it to rounding (same loss to 1e-6 per step) but is not repeatable bit for bit in general."""
    import srfrd_amd
    from tests.gpu_util import build_model, random_sd
    cfg = _cfg50("SASRec", dropout=0.5)
    sd = random_sd(cfg, 21)
    B = 64
    batches = []
    for i in range(4):
        b = srfrd_amd.synthetic_batch(400, 50, B, seed=31, index=i, device="cuda", packed=True)[1]
        b[0][:, -5:] = torch.tensor([7, 7, 9, 7, 11], device="cuda")        # hot items: long per-item runs in the sort
        b[2][:, -5:] = torch.tensor([7, 9, 7, 11, 7], device="cuda")
        batches.append(b)

    def run(det, graph=True):
        model = build_model(cfg, {k: v.clone() for k, v in sd.items()}).train()
        tr = srfrd_amd.FusedTrainer(model, batch_size=B, seq_len=50, seed=5, use_graph=graph, deterministic=det)
        losses = [float(tr.step_packed(b).cpu()) for b in batches]
        return losses, model.flat_parameters().detach().clone()

    la, fa = run(True)
    lb, fb = run(True)
    assert la == lb and torch.equal(fa, fb)                                   # bit for bit
    le, fe = run(True, graph=False)
    assert le == la and torch.equal(fe, fa)                                   # eager == graph replay, bit for bit
    lc, fc = run(False)
    assert max(abs(x - y) for x, y in zip(la[:2], lc[:2])) < 1e-6            # same arithmetic up to summation order


def test_trainer_state_dict_resumes_bitwise_and_interoperates_with_torch_adam():
    """FusedTrainer.state_dict() / load_state_dict(): (1) a run interrupted after two steps and resumed in a fresh model +
    trainer takes a third step bit-identical to the uninterrupted run's (dropout on: the mask seed sequence continues;
    deterministic scatter, so the comparison can be bitwise); (2) the dict IS a torch.optim.Adam state_dict: torch's Adam
    loads it, and a trainer loads the state of a torch Adam that stepped the module path, the next fused step then matching
    the next autograd step."""
    import copy
    import srfrd_amd
    from tests.gpu_util import build_model, cuda, random_sd
    from tests.helpers import adam_tolerance
    cfg = O.Cfg("SRFRN", 300, 20, 45, d_fake=5, dropout=0.5)
    sd0 = random_sd(cfg, 5)
    batches = [cuda(*srfrd_amd.synthetic_batch(300, 20, 8, seed=70 + i, device="cpu")) for i in range(3)]
    # ---- (1) interrupted and resumed
    mA = build_model(cfg, {k: v.clone() for k, v in sd0.items()}).train()
    tA = srfrd_amd.FusedTrainer(mA, 8, 20, seed=9, use_graph=True, deterministic=True)
    for i in range(2):
        tA.step(*batches[i])
    ckpt_model = {k: v.detach().clone() for k, v in mA.state_dict().items()}
    ckpt_opt = copy.deepcopy(tA.state_dict())
    tA.step(*batches[2])
    mB = build_model(cfg, {k: v.clone().cpu() for k, v in ckpt_model.items()}).train()
    tB = srfrd_amd.FusedTrainer(mB, 8, 20, seed=12345, use_graph=True, deterministic=True)      # (its own seed is overwritten)
    tB.load_state_dict(ckpt_opt)
    assert tB.steps_done == 2
    tB.step(*batches[2])
    for k, v in mA.state_dict().items():
        assert torch.equal(v, mB.state_dict()[k]), k
    # ---- (2a) torch's Adam takes the dict
    opt = torch.optim.Adam(mB.parameters(), lr=1e-3, betas=(0.9, 0.98))
    opt.load_state_dict(tB.state_dict())
    st = opt.state_dict()["state"]
    assert len(st) == len(list(mB.parameters())) and all(float(s["step"]) == 3.0 for s in st.values())
    # ---- (2b) a trainer takes torch Adam's state (dropout off for an element-wise comparison)
    cfg0 = O.Cfg("SRFRN", 300, 20, 45, d_fake=5)
    mC = build_model(cfg0, {k: v.clone() for k, v in sd0.items()}).train()
    optC = torch.optim.Adam(mC.parameters(), lr=1e-3, betas=(0.9, 0.98))
    crit = torch.nn.BCEWithLogitsLoss()

    def autograd_step(model, o, batch):
        h, pl, nl = model(*batch)
        idx = torch.where(batch[3] != 0)
        loss = crit(pl[idx], torch.ones_like(pl)[idx]) + crit(nl[idx], torch.zeros_like(nl)[idx])
        o.zero_grad()
        loss.backward()
        g = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
        o.step()
        return g

    for i in range(2):
        autograd_step(mC, optC, batches[i])
    mD = build_model(cfg0, {k: v.detach().clone().cpu() for k, v in mC.state_dict().items()}).train()
    tD = srfrd_amd.FusedTrainer(mD, 8, 20, use_graph=False)
    tD.load_state_dict(optC.state_dict())
    g = autograd_step(mC, optC, batches[2])
    tD.step(*batches[2])
    sdC, sdD = mC.state_dict(), mD.state_dict()
    for k in sdC:
        d = (sdC[k] - sdD[k]).abs().double().cpu()
        assert not bool((d > adam_tolerance([g[k].cpu()])).any()), k


@pytest.mark.parametrize("graph", [False, True])
def test_fused_trainer_l2_emb_matches_golden(graph):
    """reference trainer.py:39 with config.l2_emb = 0.05 (tests/golden/SRFRN_l2.npz): the fused step's loss curve and the
    weights after 1 and 3 steps - every parameter tensor, the item table's padding row included, carries the norm term."""
    import srfrd_amd
    from tests.gpu_util import build_model, cuda, maxerr
    g, sd, batch = load_golden("SRFRN", l2=True)
    cfg = golden_cfg("SRFRN")
    model = build_model(cfg, sd).train()
    tr = srfrd_amd.FusedTrainer(model, 8, 20, lr=1e-3, betas=(0.9, 0.98), l2_emb=float(g["l2_emb"]), use_graph=graph)
    seq, rsq, pos, prs, neg, nrs = cuda(*batch)
    w1, w3 = sub(g, "w1/"), sub(g, "w3/")
    for step in range(3):
        loss = tr.step(None, seq, rsq, pos, prs, neg, nrs)
        assert abs(float(loss.cpu()) - float(g[f"loss{step}"])) < TOL, step
        if step == 0:
            msd = model.state_dict()
            for k in w1:
                assert maxerr(msd[k].cpu(), w1[k]) < TOL, k
    msd = model.state_dict()
    for k in w3:
        assert maxerr(msd[k].cpu(), w3[k]) < 2e-4, k


def test_spin_up_leaves_the_training_trajectory_untouched():
    """FusedTrainer.spin_up() replays the captured step inside a snapshot (bench.py uses it to reach the device's steady
    state before a short timed region): parameters, moments, step counter and dropout seed must come back bit for bit, so
    the steps that follow are the ones that would have run without it."""
    import srfrd_amd
    from tests.gpu_util import build_model, random_sd
    cfg = _cfg50("SRFRN", dropout=0.5)
    sd = random_sd(cfg, 21)
    B = 16
    batches = [srfrd_amd.synthetic_batch(400, 50, B, seed=9, index=i, device="cuda", packed=True)[1] for i in range(3)]
    outs = []
    for spin in (False, True):
        model = build_model(cfg, {k: v.clone() for k, v in sd.items()}).train()
        tr = srfrd_amd.FusedTrainer(model, batch_size=B, seq_len=50, seed=77, deterministic=True)
        if spin:
            tr.ids_ring[0].copy_(batches[2])
            tr.spin_up(7)
        losses = [float(tr.step_packed(b).cpu()) for b in batches]
        outs.append((losses, {k: v.detach().clone() for k, v in model.state_dict().items()}, tr.state_dict()))
    assert outs[0][0] == outs[1][0]
    for k in outs[0][1]:
        assert torch.equal(outs[0][1][k], outs[1][1][k]), k
    s0, s1 = outs[0][2]["state"], outs[1][2]["state"]
    for i in s0:
        for name in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(s0[i][name], s1[i][name]), (i, name)


@pytest.mark.parametrize("kind", ["SASRec", "SRFRN"])
def test_module_forward_paths_agree(kind):
    """forward() under autograd runs the launchers through a light torch.autograd.Function; ``library_ops = True`` routes the
    same call through the registered srfrd::encoder_fwd custom op (srfrd_amd/ops.py): same outputs, same gradients."""
    import copy
    import srfrd_amd
    from tests.gpu_util import build_model, cuda, random_sd
    cfg = _cfg50(kind)
    sd = random_sd(cfg, 41)
    m1 = build_model(cfg, sd).train()
    m2 = copy.deepcopy(m1)
    m2.library_ops = True
    batch = cuda(*srfrd_amd.synthetic_batch(400, 50, 12, seed=8, device="cpu")[1:])
    outs = []
    for m in (m1, m2):
        h, pl, nl = m(None, *batch)
        _loss(pl, nl, batch[2]).backward()
        outs.append((h, pl, nl))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    # (the registered op materialises a zero hidden-state gradient, so its backward's head walks every row; the Function
    # path passes None and the head starts at the first item: the same sums in another order)
    for (k, p), (_, q) in zip(m1.named_parameters(), m2.named_parameters()):
        assert float((p.grad - q.grad).abs().max()) < 1e-6, k
