"""Drop-in ``nn.Module`` surface of the reference model zoo, backed by the HIP kernels.

Same class names, positional constructor signatures, ``forward`` / ``predict`` signatures, attribute tree and
``state_dict`` keys as the reference (SRFR_model.py:53 SRFR, :154 SRFRN, :429 SRFU, :543/:553/:562 SRFU_B/F/R,
:572 SASRec).  The stock ``nn.Embedding`` / ``nn.LayerNorm`` / ``nn.MultiheadAttention`` /
``nn.Conv1d`` objects are kept ONLY as parameter containers (created in the reference's construction order, so a
given ``torch.manual_seed`` yields the same initial weights); none of their ``forward`` methods is ever called.
All arithmetic runs in ``libsrfrd_hip.so``; on a non-CUDA device or without the library, calls raise.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import check, ptr


FLAT_SLACK = 4096        # floats of zero padding kept behind a model's flat parameter vector


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ids(t, device, shape=None):
    if t is None:
        return None
    t = torch.as_tensor(t)
    if t.device != device or t.dtype != torch.int64 or not t.is_contiguous():
        t = t.to(device=device, dtype=torch.int64).contiguous()
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"id tensor has shape {tuple(t.shape)}, expected {tuple(shape)}")
    return t


class PointWiseFeedForward(nn.Module):
    """Parameter container for reference SRFR_model.py:36-51 (conv1 / conv2 with kernel_size=1)."""

    def __init__(self, in_channel, out_channel, pwff_dropout_rate):
        super().__init__()
        self.conv1 = nn.Conv1d(in_channel, out_channel, kernel_size=1)
        self.dropout1 = nn.Dropout(p=pwff_dropout_rate)
        self.relu = nn.ReLU()
        self.conv2 = nn.Conv1d(out_channel, out_channel, kernel_size=1)
        self.dropout2 = nn.Dropout(p=pwff_dropout_rate)


class SRFR_Embedding(nn.Module):
    """Parameter container for reference SRFR_model.py:6-15."""

    def __init__(self, item_number, item_embedding_size, fake_embedding_size, dropout_rate, maxlen, device):
        super().__init__()
        self.item_embed = nn.Embedding(item_number + 1, item_embedding_size, padding_idx=0)
        self.fake_embed = nn.Embedding(3, fake_embedding_size, padding_idx=0)   # 0 padding, 1 fake, 2 real
        self.pos_embed = nn.Embedding(maxlen, item_embedding_size)
        self.total_hiden_size = item_embedding_size + fake_embedding_size
        self.dropout = nn.Dropout(p=dropout_rate)
        self.device = device


class SRFU_Embedding(nn.Module):
    """Parameter container for reference SRFR_model.py:399-409."""

    def __init__(self, item_number, item_embedding_size, number_of_labels, dropout_rate, maxlen, device):
        super().__init__()
        self.item_embed = nn.Embedding(item_number + 1, item_embedding_size, padding_idx=0)
        self.user_label_embed = nn.Embedding(number_of_labels, item_embedding_size)
        self.pos_embed = nn.Embedding(maxlen, item_embedding_size)
        self.dropout = nn.Dropout(p=dropout_rate)
        self.device = device
        self.item_embedding_size = item_embedding_size
        self.number_of_labels = number_of_labels
        self.maxlen = maxlen

    def get_user_label_embed(self):
        return self.user_label_embed


class _EncoderFn(torch.autograd.Function):
    """forward() of the drop-in modules under autograd: the same two launches as the registered ``srfrd::encoder_fwd`` /
    ``srfrd::encoder_bwd`` custom ops (srfrd_amd/ops.py - dispatcher-visible, opcheck'ed, and what ``module.library_ops =
    True`` routes through), without the Python glue torch.library generates around a custom op's autograd (measured: ~300 us
    of host time per training step at C2, more than the kernels take).  Unused outputs hand back None instead of a zero
    tensor (``set_materialize_grads(False)``): the backward then knows no hidden-state gradient exists and its head runs
    over the sequence's own rows only."""

    @staticmethod
    def forward(ctx, model, p, seed, inp, fk, pos, pfk, neg, nfk, *params):
        out = model._launch_fwd(inp, fk, pos, pfk, neg, nfk, p, seed, True)
        ctx.model, ctx.meta = model, (p, seed)
        ctx.id_present = tuple(t is not None for t in (inp, fk, pos, pfk, neg, nfk))
        ctx.have = (pos is not None, neg is not None)
        ctx.set_materialize_grads(False)
        hidden = out["hidden"]
        pl = out["pos_logits"] if pos is not None else hidden.new_empty(0)
        nl = out["neg_logits"] if neg is not None else hidden.new_empty(0)
        ctx.save_for_backward(hidden, pl, nl, out["save_x"], out["save_h1"], out["save_aux"],
                              *[t for t in (inp, fk, pos, pfk, neg, nfk) if t is not None])
        return hidden, pl, nl

    @staticmethod
    def backward(ctx, d_hidden, d_pl, d_nl):
        m = ctx.model
        p, seed = ctx.meta
        saved = ctx.saved_tensors
        hidden, pl, nl, sx, sh, sa = saved[:6]
        rest = iter(saved[6:])
        inp, fk, pos, pfk, neg, nfk = (next(rest) if present else None for present in ctx.id_present)
        out = {"hidden": hidden, "pos_logits": pl if ctx.have[0] else None, "neg_logits": nl if ctx.have[1] else None,
               "save_x": sx, "save_h1": sh, "save_aux": sa}
        gflat = m._launch_bwd(inp, fk, pos, pfk, neg, nfk, p, seed, out,
                              None if d_hidden is None else d_hidden.contiguous(),
                              None if (d_pl is None or pos is None) else d_pl.contiguous(),
                              None if (d_nl is None or neg is None) else d_nl.contiguous())
        return (None,) * 9 + m._grad_views(gflat)


class _SRFRDBase(nn.Module):
    """Shared machinery: flat parameter storage, kernel launches, predict."""

    _kind = None

    # ---- construction helpers (same order as the reference constructors)
    def _build_blocks(self, hidden, num_blocks, num_heads, dropout_rate):
        for _ in range(num_blocks):
            self.attention_layernorms.append(nn.LayerNorm(hidden, eps=1e-8))
            self.attention_layers.append(nn.MultiheadAttention(hidden, num_heads, dropout_rate))
            self.forward_layernorms.append(nn.LayerNorm(hidden, eps=1e-8))
            self.forward_layers.append(PointWiseFeedForward(hidden, hidden, dropout_rate))

    def _init_lists(self):
        self.attention_layernorms = nn.ModuleList()
        self.attention_layers = nn.ModuleList()
        self.forward_layernorms = nn.ModuleList()
        self.forward_layers = nn.ModuleList()

    def _finish(self, item_number, max_len, d_item, d_fake, n_labels, num_blocks, num_heads, dropout_rate):
        self._cfg = dict(kind=self._kind, n_items=item_number, max_len=max_len, d_item=d_item, d_fake=d_fake,
                         n_labels=n_labels, n_blocks=num_blocks, n_heads=num_heads)
        self.dropout_rate = float(dropout_rate)
        self._lay = None
        self._flat = None
        self._slots = None
        self._packed = None
        self._scratch = None
        self._err = None
        self._table16 = None     # bf16 shadow of the item table (use_bf16_table)
        self._lay16 = None
        # "lazy": every call launches srfrd_check_ids on its id tensors and check_ids() (called by evaluation(), or by
        # the user at any synchronisation point) raises; "eager": raise at the call itself like nn.Embedding does (one
        # host sync per call); None: no validation launch (the kernels still clamp ids, so memory stays safe)
        self.validate_ids = "lazy"

    # ---- layout / flat storage
    @property
    def layout(self):
        if self._lay is None:
            self._lay = _lib.make_layout(**self._cfg)
        return self._lay

    def _item_param(self):
        raise NotImplementedError

    def _dense_params(self):
        """[(parameter, dense offset)] in the canonical order of include/srfrd_hip.h."""
        raise NotImplementedError

    def _block_params(self, lay):
        out = []
        for i in range(lay.n_blocks):
            o, mha, ff = lay.blk[i], self.attention_layers[i], self.forward_layers[i]
            out += [(self.attention_layernorms[i].weight, o.ln1_w), (self.attention_layernorms[i].bias, o.ln1_b),
                    (mha.in_proj_weight, o.in_w), (mha.in_proj_bias, o.in_b),
                    (mha.out_proj.weight, o.out_w), (mha.out_proj.bias, o.out_b),
                    (self.forward_layernorms[i].weight, o.ln2_w), (self.forward_layernorms[i].bias, o.ln2_b),
                    (ff.conv1.weight, o.c1_w), (ff.conv1.bias, o.c1_b), (ff.conv2.weight, o.c2_w), (ff.conv2.bias, o.c2_b)]
        return out

    @property
    def n_table_pad(self):
        return (self.layout.n_table + 3) // 4 * 4

    @property
    def n_flat(self):
        return (self.n_table_pad + self.layout.n_dense + 3) // 4 * 4

    def flat_parameters(self):
        """The single fp32 vector [item table | pad | dense parameters] every nn.Parameter is a view of."""
        self._ensure_flat()
        return self._flat

    def _current_slots(self, table):
        """[(parameter, flat offset)] of the parameters the module tree holds NOW.  forward() asks on every call (a parameter or
        a submodule may have been replaced since the last one); walking the tree through nn.Module.__getattr__ costs ~60 us of
        host time per call (62 attribute lookups), so the (module path, name) of every slot is resolved once and each call
        follows the paths through the _modules / _parameters dictionaries - the objects found are always the current ones."""
        paths = getattr(self, "_slot_paths", None)
        if paths is not None:
            try:
                out = []
                for path, pname, off in paths:
                    m = self
                    for n in path:
                        m = m._modules[n]
                    out.append((m._parameters[pname], off))
                if out[0][0] is table:
                    return out
            except (KeyError, AttributeError):
                pass
        slots = [(table, 0)] + [(p, self.n_table_pad + off) for p, off in self._dense_params()]
        where = {}
        for mname, mod in self.named_modules():
            for pname, q in mod._parameters.items():
                if q is not None:
                    where.setdefault(id(q), (tuple(mname.split(".")) if mname else (), pname))
        try:
            self._slot_paths = [where[id(q)] + (off,) for q, off in slots]
        except KeyError:
            self._slot_paths = None
        return slots

    def _ensure_flat(self):
        lay = self.layout
        table = self._item_param()
        dev = table.device
        if dev.type != "cuda":
            raise RuntimeError("srfrd_amd modules compute only on a ROCm GPU (device 'cuda'); move the model with "
                               ".to('cuda'). There is no CPU fallback.")
        if lay.D > _lib.MAX_D:
            raise NotImplementedError(f"the fused MI355X kernels cover hidden width <= 64 (got width {lay.D})")
        slots = self._current_slots(table)
        flat = self._flat
        if flat is not None and flat.device == dev:
            base = flat.data_ptr()
            if all(p.data_ptr() == base + 4 * off and p.dtype == torch.float32 for p, off in slots):
                self._slots = slots
                return
        # (a few KiB of zeroed slack behind the vector: the data-parallel all-gather pads it to world equal shards)
        self._flat_store = torch.zeros(self.n_flat + FLAT_SLACK, device=dev, dtype=torch.float32)
        flat = self._flat_store[:self.n_flat]
        covered = 0
        for p, off in slots:
            n = p.numel()
            flat[off:off + n].copy_(p.data.reshape(-1).to(device=dev, dtype=torch.float32))
            p.data = flat[off:off + n].view(p.shape)
            p.grad = None
            covered += n
        assert covered == lay.n_table + lay.n_dense, "parameter list does not match the dense layout"
        self._flat, self._slots = flat, slots

    # ---- bf16 item-table shadow (BASELINE configs[1] / [4] "bf16"; no reference counterpart: its table is fp32)
    def use_bf16_table(self, on: bool = True, auto_refresh: bool = True):
        """Gather item rows (embedding, pos / neg targets, predict / top-k candidates) from a bf16 shadow of the item
        table: half the gather bytes.  The parameter stays fp32 (state_dict, gradients, Adam unchanged); the shadow is
        rebuilt from it before every forward of the module path and rewritten by the fused optimizer in FusedTrainer.
        Forward values then carry bf16-rounded embeddings (parity is held against the oracle with ``table_bf16``).
        ``auto_refresh=False`` (frozen weights, e.g. serving / an evaluation loop): the module path stops re-deriving the
        shadow on every call (a pass over the whole table: 50 us at 1 M items); call ``refresh_bf16_table()`` yourself
        after changing the item embeddings."""
        self._ensure_flat()
        self._bf16_auto = bool(auto_refresh)
        if on:
            lay16 = _lib.Layout.from_buffer_copy(bytes(self.layout))
            lay16.table_bf16 = 1
            self._lay16 = lay16
            self._table16 = torch.empty(self.layout.n_table, device=self._flat.device, dtype=torch.int16)
            self.refresh_bf16_table()
        else:
            self._lay16 = self._table16 = None
        return self

    @property
    def bf16_table(self) -> bool:
        return self._table16 is not None

    def refresh_bf16_table(self):
        if self._table16 is not None:
            if self._table16.device != self._flat.device:
                self._table16 = torch.empty(self.layout.n_table, device=self._flat.device, dtype=torch.int16)
            check(_lib.lib().srfrd_table_to_bf16(ptr(self._flat), self.layout.n_table, ptr(self._table16), _stream()),
                  "srfrd_table_to_bf16")

    def _table_args(self):
        """(layout, item-table pointer) as the gathering launchers take them: fp32 parameter, or the bf16 shadow"""
        if self._table16 is not None:
            return self._lay16, ptr(self._table16)
        return self.layout, ptr(self._flat)

    def pack_weights(self):
        """Refresh the MFMA-fragment-ordered copy of the encoder weights (srfrd_pack_weights) from the parameters (and the
        bf16 item-table shadow, when in use)."""
        lay, flat = self.layout, self._flat
        if getattr(self, "_bf16_auto", True):
            self.refresh_bf16_table()
        if self._packed is None or self._packed.device != flat.device:
            self._packed = torch.empty(_lib.lib().srfrd_packed_floats(C.byref(lay)), device=flat.device, dtype=torch.float32)
        check(_lib.lib().srfrd_pack_weights(C.byref(lay), C.c_void_p(flat.data_ptr() + 4 * self.n_table_pad),
                                            ptr(self._packed), None, 0.0, 0.0, 0.0, _stream()), "srfrd_pack_weights")
        return self._packed

    def _scratch_for(self, B, L, backward):
        """global workspace for sequences whose working set does not fit LDS (None when it does)."""
        need = _lib.scratch_floats(self.layout, B, L)[1 if backward else 0]
        if need == 0:
            return None, 0
        if self._scratch is None or self._scratch.numel() < need or self._scratch.device != self._flat.device:
            self._scratch = torch.empty(need, device=self._flat.device, dtype=torch.float32)
        return self._scratch, need

    # ---- launches
    def _launch_fwd(self, inp, fk, pos, pfk, neg, nfk, dropout_p, seed, save, seq0=0, dbg=None, dbg_seq=0):
        lay, flat = self.layout, self._flat
        B, L = inp.shape
        dev = inp.device
        hidden = torch.empty(B, L, lay.d_out, device=dev, dtype=torch.float32)
        pl = torch.empty(B, L, device=dev, dtype=torch.float32) if pos is not None else None
        nl = torch.empty(B, L, device=dev, dtype=torch.float32) if neg is not None else None
        sx = torch.empty(B, lay.n_blocks + 1, L, lay.D, device=dev, dtype=torch.float32) if save else None
        sh = torch.empty(B, lay.n_blocks, L, lay.D, device=dev, dtype=torch.float32) if save else None
        sa = torch.empty(_lib.lib().srfrd_aux_floats(C.byref(lay), B, L), device=dev, dtype=torch.float32) if save else None
        packed = self.pack_weights()          # parameters may have been stepped since the last call
        scratch, n_scr = self._scratch_for(B, L, backward=False)
        lay_t, tab = self._table_args()
        check(_lib.lib().srfrd_encoder_fwd(
            C.byref(lay_t), tab, C.c_void_p(flat.data_ptr() + 4 * self.n_table_pad), ptr(packed),
            ptr(inp), ptr(fk), ptr(pos), ptr(pfk), ptr(neg), ptr(nfk), B, L, float(dropout_p), int(seed) & 0xFFFFFFFF,
            None, int(seq0), ptr(hidden), ptr(pl), ptr(nl), ptr(sx), ptr(sh), ptr(sa), None, ptr(scratch), n_scr, ptr(dbg),
            int(dbg_seq), _stream()), "srfrd_encoder_fwd")
        return {"hidden": hidden, "pos_logits": pl, "neg_logits": nl, "save_x": sx, "save_h1": sh, "save_aux": sa}

    def _launch_fwd_last(self, inp, fk):
        """Eval-mode encoder state of the last position only, (B, 1, d_out): what predict() / topk() rank with."""
        lay, flat = self.layout, self._flat
        B, L = inp.shape
        hidden = torch.empty(B, 1, lay.d_out, device=inp.device, dtype=torch.float32)
        packed = self.pack_weights()
        scratch, n_scr = self._scratch_for(B, L, backward=False)
        lay_t, tab = self._table_args()
        check(_lib.lib().srfrd_encoder_fwd_last(
            C.byref(lay_t), tab, C.c_void_p(flat.data_ptr() + 4 * self.n_table_pad), ptr(packed), ptr(inp), ptr(fk), B, L,
            ptr(hidden), ptr(scratch), n_scr, _stream()), "srfrd_encoder_fwd_last")
        return hidden

    def _launch_bwd(self, inp, fk, pos, pfk, neg, nfk, dropout_p, seed, out, d_hidden, d_pl, d_nl, seq0=0,
                    dbg=None, dbg_seq=0):
        lay, flat = self.layout, self._flat
        B, L = inp.shape
        dev = inp.device
        gflat = torch.zeros(self.n_flat, device=dev, dtype=torch.float32)
        n_slabs = _lib.lib().srfrd_bwd_grid(C.byref(lay), B, L)
        slabs = torch.empty(n_slabs, lay.n_dense, device=dev, dtype=torch.float32)
        scratch, n_scr = self._scratch_for(B, L, backward=True)
        lay_t, tab = self._table_args()
        check(_lib.lib().srfrd_encoder_bwd(
            C.byref(lay_t), tab, C.c_void_p(flat.data_ptr() + 4 * self.n_table_pad), ptr(self._packed),
            ptr(inp), ptr(fk), ptr(pos), ptr(pfk), ptr(neg), ptr(nfk), B, L, float(dropout_p), int(seed) & 0xFFFFFFFF,
            None, int(seq0), ptr(out["hidden"]), ptr(out["pos_logits"]), ptr(out["neg_logits"]), ptr(out["save_x"]),
            ptr(out["save_h1"]), ptr(out["save_aux"]), ptr(d_hidden), ptr(d_pl), ptr(d_nl), 0, ptr(gflat), None, ptr(slabs), ptr(scratch), n_scr,
            ptr(dbg), int(dbg_seq), _stream()), "srfrd_encoder_bwd")
        check(_lib.lib().srfrd_reduce_dense(ptr(slabs), n_slabs, lay.n_dense,
                                            C.c_void_p(gflat.data_ptr() + 4 * self.n_table_pad), None, B, None, None,
                                            _stream()),
              "srfrd_reduce_dense")
        return gflat

    def _validate(self, inp, fk, pos, pfk, neg, nfk):
        if not self.validate_ids:
            return
        self._err_word(inp.device)
        embeds_fake = self._kind in ("SRFR", "SRFRN")      # SRFU_* only count ids 1 / 2; SASRec ignores them
        check(_lib.lib().srfrd_check_ids(ptr(inp), ptr(pos), ptr(neg), ptr(fk) if embeds_fake else None,
                                         ptr(pfk), ptr(nfk), inp.numel(), self.layout.n_items, 2, ptr(self._err),
                                         _stream()), "srfrd_check_ids")
        if self.validate_ids == "eager":
            self.check_ids()

    def _err_word(self, dev):
        if self._err is None or self._err.device != dev:
            self._err = torch.zeros(1, device=dev, dtype=torch.int32)
        return self._err

    def check_ids(self):
        """Raise IndexError if any call since the last check saw an id outside the embedding tables (what the
        reference's nn.Embedding raises at the lookup).  Synchronises the device."""
        if self._err is None:
            return
        bits = int(self._err.item())
        if bits:
            self._err.zero_()
            what = " and ".join(n for b, n in ((1, f"an item id outside [0, {self.layout.n_items}]"),
                                               (2, "a fake / review id outside [0, 2]")) if bits & b)
            raise IndexError(f"index out of range in self: {what} (ids were clamped; results of that call are invalid)")

    def _prep(self, input_ids, fake_ids, positive_ids, positive_fake_ids, negative_ids, negative_fake_ids):
        self._ensure_flat()
        dev = self._flat.device
        inp = _ids(input_ids, dev)
        if inp.dim() != 2:
            raise ValueError("input_ids must be (batch, seq_len)")
        if inp.shape[1] > self.layout.max_len:
            raise IndexError(f"sequence length {inp.shape[1]} exceeds max_len {self.layout.max_len}")
        shp = inp.shape
        fk = _ids(fake_ids, dev, shp)
        pos, neg = _ids(positive_ids, dev, shp), _ids(negative_ids, dev, shp)
        pfk = _ids(positive_fake_ids, dev, shp) if self._kind == "SRFRN" and pos is not None else None
        nfk = _ids(negative_fake_ids, dev, shp) if self._kind == "SRFRN" and neg is not None else None
        if self._kind.startswith("SRFU") and fk is None:
            raise ValueError("SRFU models need fake_ids to derive the user label")
        self._validate(inp, fk, pos, pfk, neg, nfk)
        return inp, fk, pos, pfk, neg, nfk

    def forward(self, user_ids, input_ids, fake_ids, positive_ids=None, positive_fake_ids=None, negative_ids=None,
                negative_fake_ids=None):
        """-> (hidden_state (B,L,d_out), pos_logits (B,L) | None, neg_logits (B,L) | None); ``user_ids`` is unused,
        exactly as in the reference (SRFR_model.py:92, :192, :473, :651)."""
        ids = self._prep(input_ids, fake_ids, positive_ids, positive_fake_ids, negative_ids, negative_fake_ids)
        p = self.dropout_rate if self.training else 0.0
        seed = self._next_seed() if p > 0 else 0
        grad = torch.is_grad_enabled() and any(q.requires_grad for q, _ in self._slots)
        if getattr(self, "library_ops", False):
            # torch.ops.srfrd.encoder_fwd (srfrd_amd/ops.py): the parameters are op inputs, its registered backward runs
            # srfrd::encoder_bwd and hands one gradient per parameter to autograd
            hidden, pl, nl, *_ = torch.ops.srfrd.encoder_fwd([q for q, _ in self._slots], *ids, ops.register_model(self), p, seed, 0, grad)
        elif grad:
            hidden, pl, nl = _EncoderFn.apply(self, p, seed, *ids, *[q for q, _ in self._slots])
        else:
            out = self._launch_fwd(*ids, p, seed, False)
            hidden, pl, nl = out["hidden"], out["pos_logits"], out["neg_logits"]
        return hidden, (pl if ids[2] is not None else None), (nl if ids[4] is not None else None)

    def _next_seed(self):
        """dropout seed of one forward: a host-side counter hashed with torch's seed (torch.manual_seed reproduces a run; no
        device random number, no host sync - the reference's nn.Dropout draws on the device stream without either)"""
        base = torch.initial_seed() & 0xFFFFFFFF
        if getattr(self, "_seed_base", None) != base:
            self._seed_base, self._seed_ctr = base, 0
        self._seed_ctr += 1
        h = (base * 0x9E3779B1 + self._seed_ctr * 0x85EBCA6B) & 0xFFFFFFFF
        h ^= h >> 15
        return (h * 0x2C1B3C6D) & 0x7FFFFFFF

    def _grad_views(self, gflat):
        """one gradient view per parameter (order of _slots) of the flat gradient vector: a single split + reshapes"""
        if getattr(self, "_split_sizes", None) is None or self._split_n != gflat.numel():
            if any(b[1] < a[1] + a[0].numel() for a, b in zip(self._slots, self._slots[1:])):      # (never: the layout is ascending)
                return tuple(gflat[off:off + q.numel()].view(q.shape) for q, off in self._slots)
            sizes, keep, at = [], [], 0
            for q, off in self._slots:
                if off > at:
                    sizes.append(off - at); keep.append(False)
                sizes.append(q.numel()); keep.append(True)
                at = off + q.numel()
            if at < gflat.numel():
                sizes.append(gflat.numel() - at); keep.append(False)
            self._split_sizes, self._split_keep, self._split_n = sizes, keep, gflat.numel()
        parts = gflat.split_with_sizes(self._split_sizes)
        return tuple(t.view(q.shape) for (t, k), (q, _) in zip(((t, k) for t, k in zip(parts, self._split_keep) if k), self._slots))

    def user_labels(self, fake_ids):
        """get_Labels (SRFU_*) / the predict-time label (SRFRN) as an int64 (B,) tensor, computed on device."""
        self._ensure_flat()
        fk = _ids(fake_ids, self._flat.device)
        kind = _lib.KINDS[self._kind] if self._kind != "SRFR" else _lib.KINDS["SRFRN"]
        return torch.ops.srfrd.user_labels(fk, kind)

    def predict(self, user_ids, input_ids, fake_ids, label):
        """reference predict(): logits of the candidate items ``label`` against the last position's state.
        ``label`` is (I_c,) shared by the batch (reference utils.py:589) or (B, I_c) per user (batched eval);
        returns (I_c,) for a single sequence, else (B, I_c) - the reference's ``.squeeze()`` behaviour."""
        with torch.no_grad():
            ids = self._prep(input_ids, fake_ids, None, None, None, None)
            hidden = self._launch_fwd_last(ids[0], ids[1])
        logits = self.candidate_logits(hidden, ids[1], label)
        return logits.squeeze()

    def candidate_logits(self, hidden, fake_ids, cand):
        lay = self.layout
        dev = hidden.device
        B, L = hidden.shape[0], hidden.shape[1]
        cand = _ids(cand, dev)
        stride = 0 if cand.dim() == 1 else cand.shape[1]
        n_cand = cand.shape[-1]
        if cand.dim() == 2 and cand.shape[0] != B:
            raise ValueError("per-user candidates must be (B, I_c)")
        ulab = self.user_labels(fake_ids) if self._kind == "SRFRN" else None
        if self.validate_ids:
            check(_lib.lib().srfrd_check_ids(ptr(cand), None, None, None, None, None, cand.numel(), lay.n_items, 2,
                                             ptr(self._err_word(dev)), _stream()), "srfrd_check_ids")
            if self.validate_ids == "eager":
                self.check_ids()
        return torch.ops.srfrd.predict_logits(hidden.contiguous(), cand.contiguous(), ulab, ops.register_model(self))

    def topk(self, user_ids, input_ids, fake_ids, k=10, exclude_pad=True, item_range=None):
        """Full-catalog ranking: (indices int64 (B,k), scores (B,k)); the (B, I) logits never reach HBM."""
        with torch.no_grad():
            ids = self._prep(input_ids, fake_ids, None, None, None, None)
            hidden = self._launch_fwd_last(ids[0], ids[1])
        lay = self.layout
        lo, hi = item_range if item_range is not None else (0, lay.n_items + 1)
        ulab = self.user_labels(ids[1]) if self._kind == "SRFRN" else None
        idx, val = torch.ops.srfrd.logits_topk(hidden, ulab, ops.register_model(self), lo, hi, k, bool(exclude_pad))
        return idx, val


class SRFR(_SRFRDBase):
    """reference SRFR_model.py:53-152: [item || fake] input channel, last_conv D -> d_item, item-only targets."""
    _kind = "SRFR"

    def __init__(self, item_number, max_len=20, item_embedding_size=50, fake_embedding_size=10, dropout_rate=0.5,
                 num_blocks=2, num_heads=1, device="cpu"):
        super().__init__()
        self.device = device
        self.total_hidden_size = item_embedding_size + fake_embedding_size
        self.embedding_layer = SRFR_Embedding(item_number, item_embedding_size, fake_embedding_size, dropout_rate,
                                              max_len, device)
        self._init_lists()
        self.last_conv = nn.Conv1d(self.total_hidden_size, item_embedding_size, kernel_size=1)
        self.last_layernorm = nn.LayerNorm(item_embedding_size, eps=1e-8)
        self._build_blocks(self.total_hidden_size, num_blocks, num_heads, dropout_rate)
        self._finish(item_number, max_len, item_embedding_size, fake_embedding_size, 0, num_blocks, num_heads, dropout_rate)

    def _item_param(self):
        return self.embedding_layer.item_embed.weight

    def _dense_params(self):
        lay = self.layout
        return ([(self.embedding_layer.pos_embed.weight, lay.off_pos), (self.embedding_layer.fake_embed.weight, lay.off_side)]
                + self._block_params(lay)
                + [(self.last_conv.weight, lay.off_lc_w), (self.last_conv.bias, lay.off_lc_b),
                   (self.last_layernorm.weight, lay.off_ll_w), (self.last_layernorm.bias, lay.off_ll_b)])


class SRFRN(_SRFRDBase):
    """reference SRFR_model.py:154-259: [item || fake] channel in and out; targets carry their fake embedding."""
    _kind = "SRFRN"

    def __init__(self, item_number, max_len=20, item_embedding_size=50, fake_embedding_size=10, dropout_rate=0.5,
                 num_blocks=2, num_heads=1, device="cpu"):
        super().__init__()
        self.device = device
        self.total_hidden_size = item_embedding_size + fake_embedding_size
        self.embedding_layer = SRFR_Embedding(item_number, item_embedding_size, fake_embedding_size, dropout_rate,
                                              max_len, device)
        self._init_lists()
        self.last_layernorm = nn.LayerNorm(self.total_hidden_size, eps=1e-8)
        self._build_blocks(self.total_hidden_size, num_blocks, num_heads, dropout_rate)
        self._finish(item_number, max_len, item_embedding_size, fake_embedding_size, 0, num_blocks, num_heads, dropout_rate)

    def _item_param(self):
        return self.embedding_layer.item_embed.weight

    def _dense_params(self):
        lay = self.layout
        return ([(self.embedding_layer.pos_embed.weight, lay.off_pos), (self.embedding_layer.fake_embed.weight, lay.off_side)]
                + self._block_params(lay)
                + [(self.last_layernorm.weight, lay.off_ll_w), (self.last_layernorm.bias, lay.off_ll_b)])


class SRFU(_SRFRDBase):
    """reference SRFR_model.py:429-540: per-user label embedding added to every position."""
    _kind = None

    def __init__(self, item_number, max_len=20, item_embedding_size=50, number_of_labels=2, dropout_rate=0.5,
                 num_blocks=2, num_heads=1, device="cpu"):
        super().__init__()
        self.device = device
        self.maxlen = max_len
        self.embedding_layer = SRFU_Embedding(item_number, item_embedding_size, number_of_labels, dropout_rate, max_len,
                                              device)
        self._init_lists()
        self.last_layernorm = nn.LayerNorm(item_embedding_size, eps=1e-8)
        self._build_blocks(item_embedding_size, num_blocks, num_heads, dropout_rate)
        self._finish(item_number, max_len, item_embedding_size, 0, number_of_labels, num_blocks, num_heads, dropout_rate)

    def get_Labels(self, fake_ids):
        if self._kind is None:      # the reference base class raises too (SRFR_model.py:467-471)
            raise TypeError("not implemented result")
        return self.user_labels(fake_ids)

    def _ensure_flat(self):
        if self._kind is None:
            raise TypeError("not implemented result")
        super()._ensure_flat()

    def _item_param(self):
        return self.embedding_layer.item_embed.weight

    def _dense_params(self):
        lay = self.layout
        return ([(self.embedding_layer.pos_embed.weight, lay.off_pos),
                 (self.embedding_layer.user_label_embed.weight, lay.off_side)]
                + self._block_params(lay)
                + [(self.last_layernorm.weight, lay.off_ll_w), (self.last_layernorm.bias, lay.off_ll_b)])


class SRFU_B(SRFU):
    """binary user label, reference SRFR_model.py:543-551 (more fake -> 2, more real -> 1, tie -> 2)."""
    _kind = "SRFU_B"


class SRFU_F(SRFU):
    """frequency user label = number of fake reviews, reference SRFR_model.py:553-560."""
    _kind = "SRFU_F"


class SRFU_R(SRFU):
    """ratio user label = floor(10 * fake / (fake + real)), reference SRFR_model.py:562-570."""
    _kind = "SRFU_R"


class SASRec(_SRFRDBase):
    """reference SRFR_model.py:572-681."""
    _kind = "SASRec"

    def __init__(self, item_number, maxlen=20, hidden_units=50, dropout_rate=0.5, num_blocks=2, num_heads=1, device="cpu"):
        super().__init__()
        self.item_num = item_number
        self.dev = device
        self.item_emb = nn.Embedding(self.item_num + 1, hidden_units, padding_idx=0)
        self.pos_emb = nn.Embedding(maxlen, hidden_units)
        self.emb_dropout = nn.Dropout(p=dropout_rate)
        self._init_lists()
        self.last_layernorm = nn.LayerNorm(hidden_units, eps=1e-8)
        self._build_blocks(hidden_units, num_blocks, num_heads, dropout_rate)
        self._finish(item_number, maxlen, hidden_units, 0, 0, num_blocks, num_heads, dropout_rate)

    def _item_param(self):
        return self.item_emb.weight

    def _dense_params(self):
        lay = self.layout
        return ([(self.pos_emb.weight, lay.off_pos)] + self._block_params(lay)
                + [(self.last_layernorm.weight, lay.off_ll_w), (self.last_layernorm.bias, lay.off_ll_b)])

    def log2feats(self, log_seqs):
        return self.forward(None, log_seqs, None)[0]

    def forward(self, user_ids, input_ids, fake_ids, positive_ids=None, positive_fake_ids=None, negative_ids=None,
                negative_fake_ids=None):
        return super().forward(user_ids, input_ids, None, positive_ids, None, negative_ids, None)

    def predict(self, user_ids, log_seqs, fake_ids, item_indices):
        return super().predict(user_ids, log_seqs, None, item_indices)
