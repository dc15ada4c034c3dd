"""Shared test helpers: golden-fixture loading and oracle configuration."""
import os

import numpy as np
import torch

from oracle import srfrd_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KINDS = O.KINDS
G_I, G_L, G_B = 120, 20, 8


def golden_cfg(kind, dropout=0.0):
    if kind == "SASRec":
        return O.Cfg(kind, G_I, G_L, 50, dropout=dropout)
    if kind in ("SRFR", "SRFRN"):
        return O.Cfg(kind, G_I, G_L, 45, d_fake=5, dropout=dropout)
    nl = {"SRFU_B": 3, "SRFU_F": G_L + 1, "SRFU_R": 11}[kind]
    return O.Cfg(kind, G_I, G_L, 50, n_labels=nl, dropout=dropout)


def load_golden(kind):
    z = np.load(os.path.join(GOLDEN, f"{kind}.npz"))
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
