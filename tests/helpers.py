"""Shared test helpers: golden-fixture loading and oracle configuration."""
import os

import numpy as np
import torch

from oracle import srfrd_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KINDS = O.KINDS
G_I, G_L, G_B = 1000, 20, 8


HEAD_CASES = (("SASRec", 2), ("SRFRN", 5), ("SRFU_B", 2))      # tests/golden/<kind>_h<heads>.npz (make_golden.py --heads)


def golden_cfg(kind, dropout=0.0, heads=1):
    if kind == "SASRec":
        return O.Cfg(kind, G_I, G_L, 50, dropout=dropout, num_heads=heads)
    if kind in ("SRFR", "SRFRN"):
        return O.Cfg(kind, G_I, G_L, 45, d_fake=5, dropout=dropout, num_heads=heads)
    nl = {"SRFU_B": 3, "SRFU_F": G_L + 1, "SRFU_R": 11}[kind]
    return O.Cfg(kind, G_I, G_L, 50, n_labels=nl, dropout=dropout, num_heads=heads)


def load_golden(kind, heads=1, l2=False):
    z = np.load(os.path.join(GOLDEN, f"{kind}_l2.npz" if l2 else (f"{kind}.npz" if heads == 1 else f"{kind}_h{heads}.npz")))
    g = {k: z[k] for k in z.files}
    sd = {k[2:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith("w/")}
    batch = tuple(torch.from_numpy(g[k]) for k in ("seq", "rsq", "pos", "prs", "neg", "nrs"))
    return g, sd, batch


def sub(g, prefix):
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix)}


def drop_kbias(name, a, D):
    """Post-Adam weights are compared with the K slice of ``in_proj_bias`` removed.

    The key bias adds the same q.b_k to every score of a softmax row, so its true gradient is
    identically 0 and the outputs do not depend on it; what autograd returns is rounding noise
    (~1e-9) whose SIGN Adam's g/sqrt(v) normalisation turns into a +-O(lr) update.  No two
    summation orders (the reference on CPU vs on a GPU, or the oracle) agree on it.
    """
    a = a.clone() if hasattr(a, "clone") else a.copy()
    if name.endswith("in_proj_bias"):
        a[D:2 * D] = 0
    return a


def oracle_step_with_grads(cfg, sd, opt, batch, **kw):
    """O.train_step that also hands back the gradients it stepped with (for adam_tolerance)."""
    loss, grads, *_ = O.grads_of(cfg, sd, batch, **kw)
    opt.step(sd, grads)
    return loss, grads


def adam_tolerance(grad_hist, lr=1e-3, noise=5e-6, base=1e-5):
    """Element-wise bound on |w_gpu - w_oracle| after len(grad_hist) Adam steps.

    Adam moves an element by lr * m_hat / (sqrt(v_hat) + eps): the MAGNITUDE of the gradient cancels, so an absolute
    gradient error d on an element whose gradient is g changes the step by about lr * d / |g| - nothing for a real
    gradient, a full +-lr when |g| is itself rounding noise (the sign is then noise).  With `noise` the absolute
    gradient agreement the gradient tests establish (observed ~1e-6, asserted 1e-4) the bound per element is
        base + steps * lr * min(1, 4 * noise / min_t |g_t|):
    1e-4 or tighter wherever every step's |gradient| exceeds ~5e-4 (at two steps), relaxing continuously to steps * lr
    only for elements whose gradient is at noise level.  A blanket `max < steps * lr` would also pass a real 1e-3
    error on an element with a real gradient; this does not.
    """
    steps = len(grad_hist)
    gabs = torch.stack([g.detach().abs().double() for g in grad_hist])
    gmin = gabs.min(0).values
    tol = base + steps * lr * torch.clamp(4.0 * noise / gmin.clamp_min(1e-300), max=1.0)
    # an element whose gradient is EXACTLY zero at every step (an item row no sequence touched) must not move at all
    return torch.where(gabs.max(0).values == 0, torch.full_like(tol, 1e-7), tol)


def assert_post_adam(msd, sd, grad_hists, D, lr=1e-3, noise=5e-6, outliers=0.0):
    """every parameter of `msd` (GPU) within adam_tolerance of `sd` (oracle); grad_hists = list (per step) of {name: grad}.
    Returns the fraction of elements that were held to 1e-4 or tighter.

    `outliers` (default 0: none): the fraction of a tensor's elements allowed outside the bound (each still within
    2 * steps * lr: opposite Adam steps).  For tests with millions of ReLU units per step only: fp32 training through ReLU is discontinuous - a
    unit whose pre-activation is within rounding of zero takes derivative 1 in one implementation and 0 in the other, and
    the gradient of THAT token (its item rows, a few per cent of them) differs while every forward output agrees to 1e-7.
    At 300 x 144 x 50 x 2 units about one such unit per step is expected (tests/diag_grad.py shows it: identical "error"
    from two independent backward kernels, gone with another batch seed).  A wrong kernel moves whole tensors, not a
    handful of elements of one item row."""
    tight = total = 0
    for k in sd:
        a, b = drop_kbias(k, msd[k].detach().cpu(), D), drop_kbias(k, sd[k], D)
        tol = adam_tolerance([drop_kbias(k, gh[k], D) if k.endswith("in_proj_bias") else gh[k] for gh in grad_hists], lr, noise)
        if k.endswith("in_proj_bias"):
            tol[D:2 * D] = 1.0          # (K-bias slice: excluded, see drop_kbias)
        d = (a.double() - b.double()).abs()
        bad = d > tol
        if outliers > 0.0 and bool(bad.any()):
            assert int(bad.sum()) <= max(1, int(outliers * bad.numel())) and float(d.max()) <= 2 * len(grad_hists) * lr * 1.1 + 1e-5, \
                (k, float(d[bad].max()), int(bad.sum()), bad.numel())
            bad = torch.zeros_like(bad)
        assert not bool(bad.any()), (k, float(d[bad].max()), float(tol[bad].min()), int(bad.sum()))
        tight += int((tol <= 1e-4).sum())
        total += tol.numel()
    return tight / max(total, 1)
