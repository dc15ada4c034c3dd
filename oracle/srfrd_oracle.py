"""CPU oracle for the SRFRD hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product path (``srfrd_amd``) never imports it and has no CPU
fallback: it raises if the HIP library is missing.

This is a plain-torch fp32 restatement (explicit matmuls, no ``nn.MultiheadAttention``,
no ``nn.Conv1d``) of the reference's forward / predict / train step / evaluation metric:

  reference SRFR_model.py:17-34    SRFR_Embedding.forward
  reference SRFR_model.py:411-424  SRFU_Embedding.forward
  reference SRFR_model.py:546-570  SRFU_{B,F,R}.get_Labels
  reference SRFR_model.py:92-142   SRFR.forward            (:192-239 SRFRN, :473-530 SRFU, :621-666 SASRec)
  reference SRFR_model.py:144-152  predict                 (:241-259 SRFRN, :532-540 SRFU, :668-681 SASRec)
  reference trainer.py:31-41       masked BCE x2, L2 term, backward, Adam(lr, betas=(0.9, 0.98))
  reference utils.py:576-598       HR@10 / NDCG@10 over 1 + 100 candidates
  torch nn/functional.py multi_head_attention_forward (explicit need_weights=True path)

Parity pinning: the reference ships no tests or golden vectors.  The oracle is pinned by
fixtures under ``tests/golden/`` that ``tests/golden/make_golden.py`` generated in the build
container by importing the reference's own ``SRFR_model.py`` classes (the reference source never
ships; only inputs/outputs do).  ``tests/test_oracle_golden.py`` checks this file against them.

Dropout: torch's RNG stream cannot be matched by a device kernel, so the product uses a
counter-based hash (``keep_mask`` below, integer-exact).  The oracle takes the same (seed, site,
sequence, row, col) coordinates and builds identical masks, so train-mode parity is bit-defined.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch

KINDS = ("SASRec", "SRFR", "SRFRN", "SRFU_B", "SRFU_F", "SRFU_R")
LN_EPS = 1e-8


@dataclass
class Cfg:
    kind: str
    item_number: int
    max_len: int
    d_item: int            # item embedding size (SASRec: hidden_units)
    d_fake: int = 0        # SRFR / SRFRN only
    n_labels: int = 0      # SRFU_* only
    num_blocks: int = 2
    num_heads: int = 1
    dropout: float = 0.0
    table_bf16: bool = False   # the build's bf16 item-table shadow (BASELINE configs[1] / [4]; no reference counterpart)

    @property
    def D(self) -> int:
        return self.d_item + (self.d_fake if self.kind in ("SRFR", "SRFRN") else 0)

    @property
    def d_out(self) -> int:
        return self.d_item if self.kind == "SRFR" else self.D


# ----------------------------------------------------------------------------------------------
# state_dict key helpers
# ----------------------------------------------------------------------------------------------
def key_item(cfg):
    return "item_emb.weight" if cfg.kind == "SASRec" else "embedding_layer.item_embed.weight"


def key_pos(cfg):
    return "pos_emb.weight" if cfg.kind == "SASRec" else "embedding_layer.pos_embed.weight"


def key_side(cfg):
    if cfg.kind in ("SRFR", "SRFRN"):
        return "embedding_layer.fake_embed.weight"
    if cfg.kind.startswith("SRFU"):
        return "embedding_layer.user_label_embed.weight"
    return None


def item_table(cfg, sd):
    """The item table as every GATHER sees it.  ``cfg.table_bf16`` (the build's bf16-shadow variant of BASELINE configs[1] and
    [4]; the reference has no counterpart - its table is fp32): each element rounded to bf16 (nearest even) for the
    forward value, gradient passed straight to the fp32 master, which is what the kernels do - they gather from the
    shadow, scatter the gradient into the fp32 gradient of the master, and Adam steps the master."""
    w = sd[key_item(cfg)]
    if not cfg.table_bf16:
        return w
    return w + (w.detach().to(torch.bfloat16).to(torch.float32) - w.detach())


# ----------------------------------------------------------------------------------------------
# counter-based dropout RNG (integer-exact; mirrored by srfrd_amd/csrc/srfrd_rng.h)
# ----------------------------------------------------------------------------------------------
def _fmix32(h: np.ndarray) -> np.ndarray:
    h = h.astype(np.uint32)
    with np.errstate(over="ignore"):
        h ^= h >> np.uint32(16)
        h = (h * np.uint32(0x85EBCA6B)).astype(np.uint32)
        h ^= h >> np.uint32(13)
        h = (h * np.uint32(0xC2B2AE35)).astype(np.uint32)
        h ^= h >> np.uint32(16)
    return h


def drop_threshold(p: float) -> int:
    """keep iff hash >= threshold;  P(drop) = threshold / 2**32."""
    return min(int(p * 4294967296.0), 0xFFFFFFFF)


def step_seed(base_seed: int, step: int) -> int:
    a = _fmix32(np.array([step & 0xFFFFFFFF], dtype=np.uint32))[0]
    return int(_fmix32(np.array([(base_seed & 0xFFFFFFFF) ^ int(a)], dtype=np.uint32))[0])


def keep_mask(seed: int, site: int, b0: int, B: int, R: int, C: int, p: float) -> torch.Tensor:
    """float mask (B,R,C): 1/(1-p) where kept, 0 where dropped.  b0 = global index of sequence 0."""
    if p <= 0.0:
        return torch.ones(B, R, C)
    with np.errstate(over="ignore"):
        h1 = _fmix32(np.array([(seed + site * 0x9E3779B9) & 0xFFFFFFFF], dtype=np.uint32))
        b = (np.arange(B, dtype=np.uint32) + np.uint32(b0 & 0xFFFFFFFF)).astype(np.uint32)
        h2 = _fmix32(h1 ^ b)                                          # (B,)
        rc = (np.arange(R, dtype=np.uint32)[:, None] * np.uint32(4096)
              + np.arange(C, dtype=np.uint32)[None, :]).astype(np.uint32)
        h3 = _fmix32(h2[:, None, None] ^ rc[None])
    keep = h3 >= np.uint32(drop_threshold(p))
    return torch.from_numpy(keep.astype(np.float32) * np.float32(1.0 / (1.0 - p)))


SITE_EMB = 0


def site_attn(i, head=0):
    return 1 + 3 * i + 1024 * head


def site_ffn1(i):
    return 2 + 3 * i


def site_ffn2(i):
    return 3 + 3 * i


# ----------------------------------------------------------------------------------------------
# get_Labels (integer semantics)
# ----------------------------------------------------------------------------------------------
def get_labels(kind: str, fake_ids: torch.Tensor) -> torch.Tensor:
    n1 = (fake_ids == 1).sum(dim=1)
    n2 = (fake_ids == 2).sum(dim=1)
    if kind == "SRFU_B":      # reference SRFR_model.py:548-549  round-half-even(sign*0.5+1.5): tie -> 2
        s = torch.sign(n1 - n2)
        return torch.where(s < 0, torch.ones_like(n1), torch.full_like(n1, 2))
    if kind == "SRFU_F":      # reference SRFR_model.py:558
        return n1
    if kind == "SRFU_R":      # reference SRFR_model.py:567-568 ; all-pad rows are 0/0 in the reference
        tot = n1 + n2         # (NaN -> INT_MIN -> OOB); the build guards that case to label 0.
        val = torch.floor(n1.to(torch.float32) / tot.clamp(min=1).to(torch.float32) * 10).to(torch.int64)
        return torch.where(tot == 0, torch.zeros_like(val), val)
    raise ValueError(kind)


def srfrn_predict_label(fake_ids: torch.Tensor) -> torch.Tensor:
    """reference SRFR_model.py:244  (sign*0.5+1.5).int(): truncation, tie -> 1, more fake -> 2."""
    n1 = (fake_ids == 1).sum(dim=1)
    n2 = (fake_ids == 2).sum(dim=1)
    return torch.where(n1 > n2, torch.full_like(n1, 2), torch.ones_like(n1))


# ----------------------------------------------------------------------------------------------
# forward
# ----------------------------------------------------------------------------------------------
def layer_norm(x, w, b):
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc * torch.rsqrt(var + LN_EPS) * w + b


def embed(cfg: Cfg, sd, input_ids, fake_ids, masks=None):
    """steps 1-4 of SURVEY 3.4 -> (B,L,D) masked input embeddings."""
    B, L = input_ids.shape
    item = item_table(cfg, sd)
    pos = sd[key_pos(cfg)]
    x = item[input_ids]
    if cfg.kind == "SASRec":
        x = x * (item.shape[1] ** 0.5)
        x = x + pos[:L].unsqueeze(0)
        if masks is not None:
            x = x * masks[SITE_EMB]
    elif cfg.kind in ("SRFR", "SRFRN"):
        x = x + pos[:L].unsqueeze(0)
        if fake_ids is None:
            fake_ids = torch.zeros_like(input_ids)
        x = torch.cat([x, sd[key_side(cfg)][fake_ids]], dim=2)
    else:
        lab = get_labels(cfg.kind, fake_ids)
        x = x + pos[:L].unsqueeze(0) + sd[key_side(cfg)][lab].unsqueeze(1)
    return x * (input_ids != 0).unsqueeze(-1).to(x.dtype)


def encoder_block(cfg: Cfg, sd, i, x, keep, m_attn=None, m_f1=None, m_f2=None, taps=None):
    """one block: LN -> MHA(q=LN(x), k=v=x, causal) -> +LN(x) -> LN -> PW-FFN(+res) -> pad mask."""
    B, L, D = x.shape
    H = cfg.num_heads
    dh = D // H
    pre = f"attention_layers.{i}."
    W, bias = sd[pre + "in_proj_weight"], sd[pre + "in_proj_bias"]
    qn = layer_norm(x, sd[f"attention_layernorms.{i}.weight"], sd[f"attention_layernorms.{i}.bias"])
    q = qn @ W[:D].T + bias[:D]
    k = x @ W[D:2 * D].T + bias[D:2 * D]
    v = x @ W[2 * D:].T + bias[2 * D:]
    q = q * math.sqrt(1.0 / dh)
    qh = q.view(B, L, H, dh).transpose(1, 2)
    kh = k.view(B, L, H, dh).transpose(1, 2)
    vh = v.view(B, L, H, dh).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2)
    causal = torch.tril(torch.ones(L, L, dtype=torch.bool))
    s = s.masked_fill(~causal, float("-inf"))
    p = torch.softmax(s, dim=-1)
    if taps is not None:
        taps[f"qn{i}"], taps[f"q{i}"], taps[f"k{i}"], taps[f"v{i}"], taps[f"p{i}"] = qn, q, k, v, p
    if m_attn is not None:
        p = p * (m_attn if m_attn.dim() == 4 else m_attn.unsqueeze(1))      # (B,H,L,L): one mask per head
    o = (p @ vh).transpose(1, 2).reshape(B, L, D)
    o = o @ sd[pre + "out_proj.weight"].T + sd[pre + "out_proj.bias"]
    h1 = qn + o
    h2 = layer_norm(h1, sd[f"forward_layernorms.{i}.weight"], sd[f"forward_layernorms.{i}.bias"])
    fp = f"forward_layers.{i}."
    a1 = h2 @ sd[fp + "conv1.weight"].squeeze(-1).T + sd[fp + "conv1.bias"]
    if m_f1 is not None:
        a1 = a1 * m_f1
    r = torch.relu(a1)
    a2 = r @ sd[fp + "conv2.weight"].squeeze(-1).T + sd[fp + "conv2.bias"]
    if m_f2 is not None:
        a2 = a2 * m_f2
    y = (a2 + h2) * keep
    if taps is not None:
        taps[f"o{i}"], taps[f"h1{i}"], taps[f"h2{i}"], taps[f"y{i}"] = o, h1, h2, y
    return y


def forward(cfg: Cfg, sd, input_ids, fake_ids, pos_ids=None, pos_fake=None, neg_ids=None, neg_fake=None,
            train=False, seed=0, b0=0, taps=None):
    """-> (hidden (B,L,d_out), pos_logits (B,L)|None, neg_logits (B,L)|None).  ``sd``: name -> tensor."""
    B, L = input_ids.shape
    D = cfg.D
    p = cfg.dropout if train else 0.0
    masks = None
    if p > 0.0:
        if seed is None:      # timing leg: torch's own Bernoulli stream, as the reference runs it
            def km(_seed, _site, _b0, b, r, c, pp):
                return torch.nn.functional.dropout(torch.ones(b, r, c), pp, True)
        else:
            km = keep_mask
        masks = {SITE_EMB: km(seed, SITE_EMB, b0, B, L, D, p)}
        for i in range(cfg.num_blocks):
            masks[site_attn(i)] = torch.stack([km(seed, site_attn(i, h), b0, B, L, L, p) for h in range(cfg.num_heads)], dim=1)
            masks[site_ffn1(i)] = km(seed, site_ffn1(i), b0, B, L, D, p)
            masks[site_ffn2(i)] = km(seed, site_ffn2(i), b0, B, L, D, p)
    x = embed(cfg, sd, input_ids, fake_ids, masks)
    keep = (input_ids != 0).unsqueeze(-1).to(x.dtype)
    if taps is not None:
        taps["x0"] = x
    for i in range(cfg.num_blocks):
        x = encoder_block(cfg, sd, i, x, keep,
                          None if masks is None else masks[site_attn(i)],
                          None if masks is None else masks[site_ffn1(i)],
                          None if masks is None else masks[site_ffn2(i)], taps)
    if cfg.kind == "SRFR":
        x = x @ sd["last_conv.weight"].squeeze(-1).T + sd["last_conv.bias"]
    h = layer_norm(x, sd["last_layernorm.weight"], sd["last_layernorm.bias"])
    item = item_table(cfg, sd)

    def tgt(ids, fids):
        e = item[ids]
        if cfg.kind == "SRFRN":
            e = torch.cat([e, sd[key_side(cfg)][fids]], dim=2)
        return (h * e).sum(dim=-1)

    pl = tgt(pos_ids, pos_fake) if pos_ids is not None else None
    nl = tgt(neg_ids, neg_fake) if neg_ids is not None else None
    return h, pl, nl


def predict(cfg: Cfg, sd, input_ids, fake_ids, cand: torch.Tensor) -> torch.Tensor:
    """cand (I_c,) shared or (B,I_c) per user -> logits (B,I_c)  (reference squeezes B==1 to (I_c,))."""
    h, _, _ = forward(cfg, sd, input_ids, fake_ids)
    hl = h[:, -1, :]
    e = item_table(cfg, sd)[cand]
    if cfg.kind == "SRFRN":
        lab = srfrn_predict_label(fake_ids)
        fe = sd[key_side(cfg)][lab]                                    # (B,d_f)
        if cand.dim() == 1:
            e = e.unsqueeze(0).expand(hl.shape[0], -1, -1)
        e = torch.cat([e, fe.unsqueeze(1).expand(-1, e.shape[1], -1)], dim=2)
        return torch.einsum("bic,bc->bi", e, hl)
    if cand.dim() == 1:
        return hl @ e.T
    return torch.einsum("bic,bc->bi", e, hl)


# ----------------------------------------------------------------------------------------------
# train step (trainer.py:31-41) and Adam
# ----------------------------------------------------------------------------------------------
def bce_sums(pos_logits, neg_logits, pos_ids):
    """-> (sum softplus(-pos), sum softplus(neg), count) over pos_ids != 0."""
    m = pos_ids != 0
    sp = torch.nn.functional.softplus(-pos_logits[m]).sum()
    sn = torch.nn.functional.softplus(neg_logits[m]).sum()
    return sp, sn, m.sum()


def loss_fn(pos_logits, neg_logits, pos_ids):
    sp, sn, n = bce_sums(pos_logits, neg_logits, pos_ids)
    return sp / n + sn / n


class Adam:
    """torch.optim.Adam(lr, betas, eps=1e-8) restated (single-tensor path, no amsgrad / weight decay)."""

    def __init__(self, sd, lr=1e-3, betas=(0.9, 0.98), eps=1e-8):
        self.lr, self.b1, self.b2, self.eps, self.t = lr, betas[0], betas[1], eps, 0
        self.m = {k: torch.zeros_like(v) for k, v in sd.items()}
        self.v = {k: torch.zeros_like(v) for k, v in sd.items()}

    def step(self, sd, grads):
        self.t += 1
        bc1 = 1.0 - self.b1 ** self.t
        bc2s = math.sqrt(1.0 - self.b2 ** self.t)
        with torch.no_grad():
            for k, p in sd.items():
                g = grads[k]
                self.m[k].mul_(self.b1).add_(g, alpha=1 - self.b1)
                self.v[k].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
                denom = (self.v[k].sqrt() / bc2s).add_(self.eps)
                p.addcdiv_(self.m[k], denom, value=-self.lr / bc1)


def grads_of(cfg: Cfg, sd, batch, train=False, seed=0, b0=0, l2_emb=0.0):
    """batch = (seq, rsq, pos, prs, neg, nrs) int64 (B,L).  -> (loss, {name: grad}, hidden, pl, nl).
    l2_emb: reference trainer.py:39 - ``loss += l2_emb * torch.norm(p)`` for EVERY parameter tensor (its gradient
    l2_emb * p / ||p|| reaches the padding rows too: it does not pass through the embedding lookup)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    seq, rsq, pos, prs, neg, nrs = batch
    h, pl, nl = forward(cfg, leaves, seq, rsq, pos, prs, neg, nrs, train=train, seed=seed, b0=b0)
    loss = loss_fn(pl, nl, pos)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    # nn.Embedding(padding_idx=0): row 0 of item_embed / fake_embed never receives gradient
    grads[key_item(cfg)][0].zero_()
    if cfg.kind in ("SRFR", "SRFRN"):
        grads[key_side(cfg)][0].zero_()
    loss = loss.detach()
    if l2_emb != 0.0:
        for k, v in sd.items():
            nrm = torch.norm(v.detach())
            loss = loss + l2_emb * nrm
            if float(nrm) > 0.0:
                grads[k] = grads[k] + l2_emb * v.detach() / nrm
    return loss, grads, h.detach(), pl.detach(), nl.detach()


def train_step(cfg: Cfg, sd, opt: Adam, batch, train=True, seed=0, b0=0, l2_emb=0.0):
    loss, grads, *_ = grads_of(cfg, sd, batch, train=train, seed=seed, b0=b0, l2_emb=l2_emb)
    opt.step(sd, grads)
    return loss


# ----------------------------------------------------------------------------------------------
# evaluation metric (utils.py:589-597)
# ----------------------------------------------------------------------------------------------
def rank_of_first(logits: torch.Tensor) -> torch.Tensor:
    """rank of candidate 0 among (U, C) logits: number of candidates scoring strictly higher
    (== ``(-logits).argsort().argsort()[0]`` when there are no exact ties)."""
    return (logits[:, 1:] > logits[:, :1]).sum(dim=1)


def hr_ndcg_at_10(ranks: torch.Tensor):
    hit = ranks < 10
    ndcg = torch.where(hit, 1.0 / torch.log2(ranks.to(torch.float64) + 2.0), torch.zeros((), dtype=torch.float64))
    n = ranks.numel()
    return float(ndcg.sum() / n), float(hit.sum().to(torch.float64) / n)


# ----------------------------------------------------------------------------------------------
# module-shaped view used by bench.py's cpu_baseline leg (times the restated trainer.py:27-41 step)
# ----------------------------------------------------------------------------------------------
class TorchStep:
    """autograd + torch.optim.Adam on the restated forward, dropout on with torch's own RNG
    (mask *statistics* as the reference; this leg is timing only)."""

    def __init__(self, cfg: Cfg, sd, lr=1e-3, betas=(0.9, 0.98)):
        self.cfg = cfg
        self.params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
        self.opt = torch.optim.Adam(list(self.params.values()), lr=lr, betas=betas)

    def step(self, batch):
        seq, rsq, pos, prs, neg, nrs = batch
        h, pl, nl = forward(self.cfg, self.params, seq, rsq, pos, prs, neg, nrs, train=True, seed=None)
        self.opt.zero_grad()
        loss = loss_fn(pl, nl, pos)
        for prm in self.params.values():          # trainer.py:39 with l2_emb = 0.0
            loss = loss + 0.0 * torch.norm(prm)
        loss.backward()
        self.opt.step()
        return loss


# ----------------------------------------------------------------------------------------------
# dataset side: loop restatements used to check srfrd_amd/dataset.py and the device sampler
# ----------------------------------------------------------------------------------------------
def partition_rows(rows, is_valid=False):
    """reference utils.py:92-139 (df_data_partition) on (user, item, 'fake'|other) rows in file order."""
    from collections import defaultdict
    usernum = itemnum = 0
    User, Rev = defaultdict(list), defaultdict(list)
    final_idx = -2 if is_valid else -1
    for u, i, f in rows:
        usernum, itemnum = max(u, usernum), max(i, itemnum)
        User[u].append(i)
        Rev[u].append(1 if f == "fake" else 2)
    train = {"item_ids": {}, "review_ids": {}}
    test = {"item_ids": {}, "review_ids": {}}
    for u in User:
        if len(User[u]) < 2:
            train["item_ids"][u], train["review_ids"][u] = User[u], Rev[u]
            test["item_ids"][u], test["review_ids"][u] = [], []
        else:
            train["item_ids"][u], train["review_ids"][u] = User[u][:final_idx], Rev[u][:final_idx]
            test["item_ids"][u], test["review_ids"][u] = [User[u][final_idx]], [Rev[u][final_idx]]
    return train, test, usernum, itemnum


def _samp_rnd(seed, batch, b, t, k):
    with np.errstate(over="ignore"):
        h = _fmix32(np.array([(seed & 0xFFFFFFFF) ^ ((batch * 0x9E3779B9) & 0xFFFFFFFF)], dtype=np.uint32))
        h = _fmix32((h + np.uint32(b)).astype(np.uint32))
        h = _fmix32(h ^ np.uint32((t * 4096 + k) & 0xFFFFFFFF))
    return int(h[0])


def sample_batch_ref(train_items, train_reviews, usernum, itemnum, B, L, seed, batch):
    """reference utils.py:21-57 (sample_function_fr) with the device sampler's counter RNG: per row a user with more
    than one interaction, sequences filled from the end, one negative outside the user's items per real position.
    train_items / train_reviews: {user: list}.  -> (user (B,), packed (6,B,L)) int64 numpy."""
    user = np.zeros(B, np.int64)
    out = np.zeros((6, B, L), np.int64)
    for b in range(B):
        u = 1
        for k in range(4096):
            u = 1 + ((_samp_rnd(seed, batch, b, 0xFFFFF, k) * usernum) >> 32)
            if len(train_items.get(u, [])) > 1:
                break
        user[b] = u
        items, revs = train_items[u], train_reviews[u]
        ts = set(items)
        nxt, nxtr = items[-1], revs[-1]
        idx = L - 1
        for i, r in zip(reversed(items[:-1]), reversed(revs[:-1])):
            out[0, b, idx], out[2, b, idx] = i, nxt
            out[1, b, idx], out[3, b, idx] = r, nxtr
            out[5, b, idx] = 1
            cand = 1
            for k in range(256):
                cand = 1 + ((_samp_rnd(seed, batch, b, idx, k) * itemnum) >> 32)
                if cand not in ts:
                    break
            out[4, b, idx] = cand
            nxt, nxtr = i, r
            idx -= 1
            if idx == -1:
                break
    return user, out


def window_labels_ref(rsq_row):
    """reference utils.py:604-626 on one padded review window -> (binary, frequency, ratio)."""
    n1 = int(np.count_nonzero(np.asarray(rsq_row) == 1))
    n2 = int(np.count_nonzero(np.asarray(rsq_row) == 2))
    binary = 1 if n1 > n2 else 2
    ratio = int(np.floor(n1 / (n1 + n2) * 10)) if (n1 + n2) else 0
    return binary, n1, ratio
