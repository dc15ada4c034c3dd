"""Helpers for the -m gpu parity tests: build a srfrd_amd model from a golden / oracle state_dict."""
import torch

import srfrd_amd
from oracle import srfrd_oracle as O


def build_model(cfg: O.Cfg, sd=None, device="cuda"):
    k = cfg.kind
    if k == "SASRec":
        m = srfrd_amd.SASRec(cfg.item_number, cfg.max_len, cfg.d_item, cfg.dropout, cfg.num_blocks, cfg.num_heads, device)
    elif k == "SRFR":
        m = srfrd_amd.SRFR(cfg.item_number, cfg.max_len, cfg.d_item, cfg.d_fake, cfg.dropout, cfg.num_blocks, cfg.num_heads, device)
    elif k == "SRFRN":
        m = srfrd_amd.SRFRN(cfg.item_number, cfg.max_len, cfg.d_item, cfg.d_fake, cfg.dropout, cfg.num_blocks, cfg.num_heads, device)
    else:
        m = getattr(srfrd_amd, k)(cfg.item_number, cfg.max_len, cfg.d_item, cfg.n_labels, cfg.dropout, cfg.num_blocks,
                                  cfg.num_heads, device)
    if sd is not None:
        missing = m.load_state_dict(sd, strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
    return m.to(device)


def random_sd(cfg: O.Cfg, seed=0):
    """xavier_normal_ on >= 2-D tensors (reference trainer.py:364-369) + small noise on 1-D ones."""
    torch.manual_seed(seed)
    m = build_model(cfg, None, device="cpu") if False else None
    ref = _cpu_model(cfg)
    g = torch.Generator().manual_seed(seed + 1)
    for _, p in ref.named_parameters():
        if p.dim() >= 2:
            torch.nn.init.xavier_normal_(p.data)
        else:
            p.data.add_(0.05 * torch.randn(p.shape, generator=g))
    return {k: v.detach().clone() for k, v in ref.state_dict().items()}


def _cpu_model(cfg):
    k = cfg.kind
    if k == "SASRec":
        return srfrd_amd.SASRec(cfg.item_number, cfg.max_len, cfg.d_item, cfg.dropout, cfg.num_blocks, cfg.num_heads, "cpu")
    if k == "SRFR":
        return srfrd_amd.SRFR(cfg.item_number, cfg.max_len, cfg.d_item, cfg.d_fake, cfg.dropout, cfg.num_blocks, cfg.num_heads, "cpu")
    if k == "SRFRN":
        return srfrd_amd.SRFRN(cfg.item_number, cfg.max_len, cfg.d_item, cfg.d_fake, cfg.dropout, cfg.num_blocks, cfg.num_heads, "cpu")
    return getattr(srfrd_amd, k)(cfg.item_number, cfg.max_len, cfg.d_item, cfg.n_labels, cfg.dropout, cfg.num_blocks,
                                 cfg.num_heads, "cpu")


def cuda(*ts):
    return tuple(None if t is None else t.cuda() for t in ts)


def maxerr(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())
