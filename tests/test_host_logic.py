"""CPU-only: host-side logic of srfrd_amd (module surface, sampler layout, sharding helpers, loud failure off-GPU)."""
import numpy as np
import pytest
import torch

import srfrd_amd
from tests.helpers import KINDS, golden_cfg, load_golden


def _build(kind, cfg):
    if kind == "SASRec":
        return srfrd_amd.SASRec(cfg.item_number, cfg.max_len, cfg.d_item, 0.5, 2, 1, "cpu")
    if kind in ("SRFR", "SRFRN"):
        return getattr(srfrd_amd, kind)(cfg.item_number, cfg.max_len, cfg.d_item, cfg.d_fake, 0.5, 2, 1, "cpu")
    return getattr(srfrd_amd, kind)(cfg.item_number, cfg.max_len, cfg.d_item, cfg.n_labels, 0.5, 2, 1, "cpu")


@pytest.mark.parametrize("kind", KINDS)
def test_state_dict_contract_matches_reference(kind):
    """key names and shapes equal the reference's state_dict (captured in the golden fixtures), and load_state_dict
    of reference weights works strictly."""
    g, sd, _ = load_golden(kind)
    m = _build(kind, golden_cfg(kind))
    mine = m.state_dict()
    assert list(mine.keys()) == list(sd.keys())          # same names, same registration order
    for k in sd:
        assert tuple(mine[k].shape) == tuple(sd[k].shape), k
    m.load_state_dict(sd, strict=True)
    n_params = sum(p.numel() for p in m.parameters())
    lay = m.layout
    assert lay.n_table + lay.n_dense == n_params


@pytest.mark.parametrize("kind", KINDS)
def test_trainer_style_init_and_cpu_call_fails_loudly(kind):
    m = _build(kind, golden_cfg(kind))
    for _, p in m.named_parameters():                     # reference trainer.py:364-369
        try:
            torch.nn.init.xavier_normal_(p.data)
        except Exception:
            pass
    ids = torch.ones(2, 20, dtype=torch.int64)
    with pytest.raises(RuntimeError, match="no CPU fallback|ROCm GPU"):
        m(None, ids, ids, ids, ids, ids, ids)
    assert hasattr(m, "predict") and hasattr(m, "forward")
    if kind != "SASRec":
        assert m.embedding_layer.item_embed.weight.shape[0] == 1001     # attribute path used at SRFR_model.py:130


def test_srfu_base_class_is_abstract_like_the_reference():
    m = srfrd_amd.SRFU(10, 5, 8, 2, 0.0, 1, 1, "cpu")
    with pytest.raises(TypeError):
        m.get_Labels(torch.zeros(1, 5, dtype=torch.int64))


def test_unsupported_geometry_is_described_by_the_layout():
    """(the rejection itself needs a device: tests/test_gpu_train.py::test_unsupported_geometry_is_rejected)"""
    m = srfrd_amd.SASRec(10, 5, 50, 0.0, 1, 2, "cpu")      # two heads
    assert m.layout.n_heads == 2
    big = srfrd_amd.SASRec(10, 5, 128, 0.0, 1, 1, "cpu")   # wider than the fused kernels cover
    assert big.layout.D == 128


def test_ratio_label_integer_form_equals_the_float32_expression():
    """csrc/srfrd_dev.h computes SRFU_R's label as (10 n1) / (n1 + n2) in integers; the reference (SRFR_model.py:567-568)
    evaluates floor(n1 / (n1 + n2) * 10) in float32.  Equal for every count pair a sequence of up to 2048 positions can
    produce - so the kernels need no floating-point division that compiler flags could loosen."""
    for tot in range(1, 2049):
        n1 = torch.arange(0, tot + 1)
        ref = torch.floor(n1 / torch.full_like(n1, tot) * 10).int()          # the reference's expression, verbatim dtype flow
        assert torch.equal(ref.long(), (10 * n1) // tot), tot


def test_synthetic_batch_layout():
    u, seq, rsq, pos, prs, neg, nrs = srfrd_amd.synthetic_batch(300, 20, 64, seed=3, index=1)
    for t in (seq, rsq, pos, prs, neg, nrs):
        assert t.shape == (64, 20) and t.dtype == torch.int64
    valid = seq != 0
    # left padding: once a row starts it never returns to 0
    assert bool(((valid.int().diff(dim=1)) >= 0).all())
    assert bool((valid.sum(1) >= 2).all())
    # pos is seq shifted by one position, with the held-out item in the last column
    assert bool((pos[:, :-1][valid[:, :-1]] == seq[:, 1:][valid[:, :-1]]).all())
    assert bool((pos[:, -1] != 0).all())
    assert bool(((pos != 0) == valid).all()) and bool(((neg != 0) == valid).all())
    assert set(rsq[valid].tolist()) <= {1, 2} and set(prs[valid].tolist()) <= {1, 2}
    assert bool((rsq[~valid] == 0).all()) and bool((nrs == valid.long()).all())
    # negatives avoid the user's own items
    own = torch.cat([seq, pos[:, -1:]], dim=1)
    assert not bool(((neg.unsqueeze(2) == own.unsqueeze(1)) & valid.unsqueeze(2)).any())
    # determinism and rank / index dependence
    again = srfrd_amd.synthetic_batch(300, 20, 64, seed=3, index=1)
    assert all(torch.equal(a, b) for a, b in zip(again, (u, seq, rsq, pos, prs, neg, nrs)))
    other = srfrd_amd.synthetic_batch(300, 20, 64, seed=3, index=1, rank=1)
    assert not torch.equal(other[1], seq)
    packed = srfrd_amd.synthetic_batch(300, 20, 64, seed=3, index=1, packed=True)[1]
    assert packed.shape == (6, 64, 20) and torch.equal(packed[0], seq) and torch.equal(packed[5], nrs)


def test_eval_candidates():
    _, seq, *_ , pos = srfrd_amd.synthetic_batch(300, 20, 32, seed=5)[:4]
    _, seq, rsq, pos, *_ = srfrd_amd.synthetic_batch(300, 20, 32, seed=5)
    cand = srfrd_amd.eval_candidates(300, seq, pos[:, -1], 100, seed=1)
    assert cand.shape == (32, 101) and torch.equal(cand[:, 0], pos[:, -1])
    assert bool((cand[:, 1:] >= 1).all()) and bool((cand[:, 1:] <= 300).all())
    assert not bool((cand[:, 1:].unsqueeze(2) == seq.unsqueeze(1)).any())


def test_shard_bounds_cover_and_align():
    for n in (1, 7, 4096, 2535651):
        for world in (1, 2, 4, 8):
            spans = [srfrd_amd.shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert all(i0 % 4 == 0 for i0, i1 in spans if i1 > i0)     # non-empty shards start float4-aligned


def test_reference_checkpoint_roundtrip(tmp_path):
    """a reference-format checkpoint (torch.save of the state_dict, trainer.py:409-411) loads strictly, safely."""
    g, sd, _ = load_golden("SRFU_R")
    path = tmp_path / "SRFR_3.pt"
    torch.save(sd, path)
    m = _build("SRFU_R", golden_cfg("SRFU_R"))
    srfrd_amd.load_reference_checkpoint(m, str(path))
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k])


def test_gradient_views_split_the_flat_vector_in_slot_order():
    """modules._grad_views: one split_with_sizes over [table | pad | dense ...] + a reshape per parameter (the module-level
    backward hands autograd views of ONE flat gradient vector, which srfrd_amd.Adam then steps in place)"""
    import torch
    from srfrd_amd.modules import _SRFRDBase

    class Fake:
        _grad_views = _SRFRDBase._grad_views

    f = Fake()
    a, b, c = torch.zeros(3, 5), torch.zeros(4), torch.zeros(2, 2)
    f._slots = [(a, 0), (b, 16), (c, 20)]              # 15 table floats, pad to 16, then two dense tensors; tail pad to 28
    g = torch.arange(28, dtype=torch.float32)
    va, vb, vc = f._grad_views(g)
    assert va.shape == a.shape and vb.shape == b.shape and vc.shape == c.shape
    assert torch.equal(va.reshape(-1), g[0:15]) and torch.equal(vb, g[16:20]) and torch.equal(vc.reshape(-1), g[20:24])
    assert va.data_ptr() == g.data_ptr() and vb.data_ptr() == g.data_ptr() + 64      # views, not copies


def test_forward_seed_is_a_host_side_function_of_the_torch_seed():
    import torch
    from srfrd_amd.modules import _SRFRDBase

    class Fake:
        _next_seed = _SRFRDBase._next_seed

    torch.manual_seed(123)
    f, g = Fake(), Fake()
    s1 = [f._next_seed() for _ in range(4)]
    s2 = [g._next_seed() for _ in range(4)]
    assert s1 == s2 and len(set(s1)) == 4 and all(0 <= s < 2 ** 31 for s in s1)
    torch.manual_seed(124)
    assert [f._next_seed() for _ in range(4)] != s1
