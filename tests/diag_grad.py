#!/usr/bin/env python3
"""Diagnostic: the fused step's GRADIENT (before Adam) against the oracle's, element by element, for one geometry.

    python tests/diag_grad.py --kind SASRec --L 144 --B 300 [--dropout 0.5] [--items 400]

Prints the largest absolute differences with parameter name / index and, for item-table rows, how often the row is touched."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="SASRec")
    ap.add_argument("--L", type=int, default=144)
    ap.add_argument("--B", type=int, default=300)
    ap.add_argument("--items", type=int, default=400)
    ap.add_argument("--dropout", type=float, default=0.5)
    ap.add_argument("--seed", type=int, default=14)
    ap.add_argument("--repeat", type=int, default=2)
    args = ap.parse_args()
    import torch
    import srfrd_amd
    from oracle import srfrd_oracle as O
    from tests.gpu_util import build_model, random_sd
    I, L, B = args.items, args.L, args.B
    cfg = O.Cfg(args.kind, I, L, 50, dropout=args.dropout) if args.kind == "SASRec" else O.Cfg(args.kind, I, L, 45, d_fake=5, dropout=args.dropout)
    sd = random_sd(cfg, 22)
    model = build_model(cfg, {k: v.clone() for k, v in sd.items()}).train()
    tr = srfrd_amd.FusedTrainer(model, batch_size=B, seq_len=L, seed=5, use_graph=False)
    full = srfrd_amd.synthetic_batch(I, L, B, seed=args.seed, device="cpu")
    loss_o, grads_o, *_ = O.grads_of(cfg, sd, full[1:], train=args.dropout > 0, seed=O.step_seed(5, 1), b0=0)
    tr.refresh()
    flat_o = torch.zeros(model.n_flat)
    names = {id(p): n for n, p in model.named_parameters()}
    spans = []
    for p, off in model._slots:
        n = names[id(p)]
        flat_o[off:off + p.numel()] = grads_o[n].reshape(-1)
        spans.append((off, off + p.numel(), n))
    for rep in range(args.repeat):
        for k, t in enumerate(full[1:]):
            tr.ids[k].copy_(t.cuda())
        tr.grad.zero_()
        tr._enqueue_compute()
        torch.cuda.synchronize()
        g = (tr.grad[:model.n_flat] / float(tr.stats[2].cpu())).cpu()
        d = (g - flat_o).abs()
        print(f"[rep {rep}] loss gpu {float(tr.loss.cpu()):.7f} oracle {float(loss_o):.7f}; max |dg| {float(d.max()):.3e}; "
              f"elements > 1e-5: {int((d > 1e-5).sum())}, > 1e-6: {int((d > 1e-6).sum())}")
        top = torch.topk(d, 12)
        di = model.layout.d_item
        for v, i in zip(top.values.tolist(), top.indices.tolist()):
            name = next(n for a, b_, n in spans if a <= i < b_)
            a0 = next(a for a, b_, n in spans if a <= i < b_)
            extra = ""
            if i < model.layout.n_table:
                row = i // di
                cnt = sum(int((t == row).sum()) for t in (full[1], full[3], full[5]))
                extra = f" item row {row} col {i % di} touched {cnt}x"
            print(f"    {name}[{i - a0}] gpu {float(g[i]):+.6e} oracle {float(flat_o[i]):+.6e} diff {v:.3e}{extra}")


if __name__ == "__main__":
    main()
