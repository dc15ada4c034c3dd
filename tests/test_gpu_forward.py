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
