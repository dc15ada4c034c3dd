"""PyTorch-ROCm custom ops (namespace ``srfrd::``) over the C ABI of include/srfrd_hip.h  (SURVEY.md 8b).

Every launcher the drop-in modules use is registered with ``torch.library`` - visible to the dispatcher as
``torch.ops.srfrd.<name>``, with a fake (meta) implementation for shape propagation and, for the encoder, a registered
backward - instead of being an opaque ctypes call from Python:

    srfrd::encoder_fwd     embedding gather -> n_blocks x {LN, causal self-attention, FFN} -> last LN -> pos / neg logits
                           (reference SRFR_model.py:92-142 and twins); backward = srfrd::encoder_bwd + srfrd_reduce_dense
    srfrd::encoder_bwd     the fused backward (reference trainer.py:40)
    srfrd::predict_logits  candidate logits of predict() (SRFR_model.py:144-152 and twins)
    srfrd::logits_topk     full-catalog top-k over an item range, logits never in HBM
    srfrd::topk_merge      merge of per-shard top-k lists
    srfrd::user_labels     get_Labels (SRFR_model.py:546-570)
    srfrd::eval_rank       rank of candidate 0 (utils.py:589-597)

A model's geometry (the srfrd_layout descriptor, its flat parameter vector and packed weights) is not expressible as op
arguments one by one; the ops take ``model_key``, the registry key of a live model (``register_model``), and read
those from it.  The parameters themselves ARE op inputs (``params``), so autograd sees the dependence and
``encoder_fwd``'s registered backward returns one gradient per parameter.  There is no CPU implementation: the ops are
registered for device type "cuda" only and raise elsewhere.
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import List, Optional

import torch

from . import _lib
from ._lib import check, ptr

_MODELS: "weakref.WeakValueDictionary[int, torch.nn.Module]" = weakref.WeakValueDictionary()


def register_model(model) -> int:
    key = id(model)
    _MODELS[key] = model
    return key


def _model(key: int):
    m = _MODELS.get(key)
    if m is None:
        raise RuntimeError("srfrd op called with the key of a model that no longer exists")
    return m


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# ------------------------------------------------------------------------------------------------ encoder forward
@torch.library.custom_op("srfrd::encoder_fwd", mutates_args=(), device_types="cuda")
def encoder_fwd(params: List[torch.Tensor], input_ids: torch.Tensor, fake_ids: Optional[torch.Tensor],
                pos_ids: Optional[torch.Tensor], pos_fake: Optional[torch.Tensor], neg_ids: Optional[torch.Tensor],
                neg_fake: Optional[torch.Tensor], model_key: int, dropout_p: float, seed: int, seq0: int,
                save: bool) -> List[torch.Tensor]:
    """-> [hidden, pos_logits, neg_logits, save_x, save_h1, save_aux] (absent outputs are empty tensors)."""
    m = _model(model_key)
    out = m._launch_fwd(input_ids, fake_ids, pos_ids, pos_fake, neg_ids, neg_fake, dropout_p, seed, save, seq0=seq0)
    # (a fresh empty tensor per absent output: custom-op returns may not alias one another)
    return [out["hidden"]] + [out[k] if out[k] is not None else out["hidden"].new_empty(0)
                              for k in ("pos_logits", "neg_logits", "save_x", "save_h1", "save_aux")]


@encoder_fwd.register_fake
def _(params, input_ids, fake_ids, pos_ids, pos_fake, neg_ids, neg_fake, model_key, dropout_p, seed, seq0, save):
    m = _model(model_key)
    lay = m.layout
    B, L = input_ids.shape
    f = dict(device=input_ids.device, dtype=torch.float32)
    return [torch.empty(B, L, lay.d_out, **f), torch.empty(B, L, **f) if pos_ids is not None else torch.empty(0, **f),
            torch.empty(B, L, **f) if neg_ids is not None else torch.empty(0, **f),
            torch.empty(B, lay.n_blocks + 1, L, lay.D, **f) if save else torch.empty(0, **f),
            torch.empty(B, lay.n_blocks, L, lay.D, **f) if save else torch.empty(0, **f),
            torch.empty(_lib.lib().srfrd_aux_floats(C.byref(lay), B, L), **f) if save else torch.empty(0, **f)]


# ------------------------------------------------------------------------------------------------ encoder backward
@torch.library.custom_op("srfrd::encoder_bwd", mutates_args=(), device_types="cuda")
def encoder_bwd(input_ids: torch.Tensor, fake_ids: Optional[torch.Tensor], pos_ids: Optional[torch.Tensor],
                pos_fake: Optional[torch.Tensor], neg_ids: Optional[torch.Tensor], neg_fake: Optional[torch.Tensor],
                model_key: int, dropout_p: float, seed: int, seq0: int, hidden: torch.Tensor, pos_logits: torch.Tensor,
                neg_logits: torch.Tensor, save_x: torch.Tensor, save_h1: torch.Tensor, save_aux: torch.Tensor,
                d_hidden: Optional[torch.Tensor], d_pos: Optional[torch.Tensor], d_neg: Optional[torch.Tensor]) -> torch.Tensor:
    """-> the flat gradient vector [item table | pad | dense] of the model's flat parameter layout."""
    m = _model(model_key)
    out = {"hidden": hidden, "pos_logits": pos_logits if pos_ids is not None else None,
           "neg_logits": neg_logits if neg_ids is not None else None, "save_x": save_x, "save_h1": save_h1, "save_aux": save_aux}
    return m._launch_bwd(input_ids, fake_ids, pos_ids, pos_fake, neg_ids, neg_fake, dropout_p, seed, out, d_hidden, d_pos, d_neg,
                         seq0=seq0)


@encoder_bwd.register_fake
def _(input_ids, fake_ids, pos_ids, pos_fake, neg_ids, neg_fake, model_key, dropout_p, seed, seq0, hidden, pos_logits, neg_logits,
      save_x, save_h1, save_aux, d_hidden, d_pos, d_neg):
    return torch.empty(_model(model_key).n_flat, device=hidden.device, dtype=torch.float32)


def _fwd_setup(ctx, inputs, output):
    (params, input_ids, fake_ids, pos_ids, pos_fake, neg_ids, neg_fake, model_key, dropout_p, seed, seq0, save) = inputs
    if not save:
        raise RuntimeError("srfrd::encoder_fwd needs save=True to be differentiated (checkpoints for the backward)")
    # outputs and id tensors through save_for_backward (no output -> grad_fn -> ctx -> output cycle: the checkpoints - B x
    # (n_blocks + 1) x L x D floats - are released by refcount after the backward, and torch's version counters guard them);
    # only ints / flags stay on ctx
    ids = (input_ids, fake_ids, pos_ids, pos_fake, neg_ids, neg_fake)
    ctx.id_present = tuple(t is not None for t in ids)
    ctx.meta = (model_key, dropout_p, seed, seq0, len(params))
    ctx.save_for_backward(*output, *[t for t in ids if t is not None])


def _fwd_backward(ctx, grads):
    model_key, p, seed, seq0, n_params = ctx.meta
    m = _model(model_key)
    saved = ctx.saved_tensors
    hidden, pl, nl, sx, sh, sa = saved[:6]
    rest = iter(saved[6:])
    inp, fk, pos, pfk, neg, nfk = (next(rest) if present else None for present in ctx.id_present)
    d_hidden, d_pl, d_nl = grads[0], grads[1], grads[2]
    gflat = torch.ops.srfrd.encoder_bwd(inp, fk, pos, pfk, neg, nfk, model_key, p, seed, seq0, hidden, pl, nl, sx, sh, sa,
                                        None if d_hidden is None else d_hidden.contiguous(),
                                        None if (d_pl is None or pos is None) else d_pl.contiguous(),
                                        None if (d_nl is None or neg is None) else d_nl.contiguous())
    per_param = list(m._grad_views(gflat))
    assert len(per_param) == n_params
    return (per_param,) + (None,) * 11


encoder_fwd.register_autograd(_fwd_backward, setup_context=_fwd_setup)


# ------------------------------------------------------------------------------------------------ ranking side
@torch.library.custom_op("srfrd::user_labels", mutates_args=(), device_types="cuda")
def user_labels(fake_ids: torch.Tensor, kind: int) -> torch.Tensor:
    fk = fake_ids.contiguous()
    lab = torch.empty(fk.shape[0], device=fk.device, dtype=torch.int64)
    check(_lib.lib().srfrd_user_labels(kind, ptr(fk), fk.shape[0], fk.shape[1], ptr(lab), _stream()), "srfrd_user_labels")
    return lab


@user_labels.register_fake
def _(fake_ids, kind):
    return torch.empty(fake_ids.shape[0], device=fake_ids.device, dtype=torch.int64)


@torch.library.custom_op("srfrd::predict_logits", mutates_args=(), device_types="cuda")
def predict_logits(hidden: torch.Tensor, cand: torch.Tensor, user_label: Optional[torch.Tensor], model_key: int) -> torch.Tensor:
    m = _model(model_key)
    lay, tab = m._table_args()
    B, L = hidden.shape[0], hidden.shape[1]
    stride = 0 if cand.dim() == 1 else cand.shape[1]
    n_cand = cand.shape[-1]
    logits = torch.empty(B, n_cand, device=hidden.device, dtype=torch.float32)
    check(_lib.lib().srfrd_predict_logits(C.byref(lay), tab, C.c_void_p(m._flat.data_ptr() + 4 * m.n_table_pad), ptr(hidden),
                                          B, L, ptr(cand), n_cand, stride, ptr(user_label), ptr(logits), _stream()),
          "srfrd_predict_logits")
    return logits


@predict_logits.register_fake
def _(hidden, cand, user_label, model_key):
    return torch.empty(hidden.shape[0], cand.shape[-1], device=hidden.device, dtype=torch.float32)


@torch.library.custom_op("srfrd::logits_topk", mutates_args=(), device_types="cuda")
def logits_topk(hidden: torch.Tensor, user_label: Optional[torch.Tensor], model_key: int, item_lo: int, item_hi: int, k: int,
                exclude_pad: bool) -> List[torch.Tensor]:
    m = _model(model_key)
    lay, tab = m._table_args()
    B, L = hidden.shape[0], hidden.shape[1]
    dev = hidden.device
    ws = torch.empty(max(_lib.lib().srfrd_topk_workspace_bytes(B, k, item_hi - item_lo), 8), device=dev, dtype=torch.uint8)
    idx = torch.empty(B, k, device=dev, dtype=torch.int64)
    val = torch.empty(B, k, device=dev, dtype=torch.float32)
    check(_lib.lib().srfrd_logits_topk(C.byref(lay), tab, C.c_void_p(m._flat.data_ptr() + 4 * m.n_table_pad), ptr(hidden), B, L,
                                       item_lo, item_hi, 1 if exclude_pad else 0, ptr(user_label), k, ptr(idx), ptr(val), ptr(ws),
                                       _stream()), "srfrd_logits_topk")
    return [idx, val]


@logits_topk.register_fake
def _(hidden, user_label, model_key, item_lo, item_hi, k, exclude_pad):
    B = hidden.shape[0]
    return [torch.empty(B, k, device=hidden.device, dtype=torch.int64), torch.empty(B, k, device=hidden.device, dtype=torch.float32)]


@torch.library.custom_op("srfrd::topk_merge", mutates_args=(), device_types="cuda")
def topk_merge(cand_idx: torch.Tensor, cand_val: torch.Tensor, k: int) -> List[torch.Tensor]:
    cand_idx, cand_val = cand_idx.contiguous(), cand_val.contiguous()
    B, n = cand_idx.shape
    idx = torch.empty(B, k, device=cand_idx.device, dtype=torch.int64)
    val = torch.empty(B, k, device=cand_idx.device, dtype=torch.float32)
    check(_lib.lib().srfrd_topk_merge(ptr(cand_idx), ptr(cand_val), B, n, k, ptr(idx), ptr(val), _stream()), "srfrd_topk_merge")
    return [idx, val]


@topk_merge.register_fake
def _(cand_idx, cand_val, k):
    B = cand_idx.shape[0]
    return [torch.empty(B, k, device=cand_idx.device, dtype=torch.int64), torch.empty(B, k, device=cand_idx.device, dtype=torch.float32)]


@torch.library.custom_op("srfrd::eval_rank", mutates_args=(), device_types="cuda")
def eval_rank(logits: torch.Tensor) -> torch.Tensor:
    logits = logits.contiguous()
    B, n = logits.shape
    rank = torch.empty(B, device=logits.device, dtype=torch.int32)
    check(_lib.lib().srfrd_eval_rank(ptr(logits), B, n, ptr(rank), None, _stream()), "srfrd_eval_rank")
    return rank


@eval_rank.register_fake
def _(logits):
    return torch.empty(logits.shape[0], device=logits.device, dtype=torch.int32)


OPS = ("encoder_fwd", "encoder_bwd", "user_labels", "predict_logits", "logits_topk", "topk_merge", "eval_rank")
