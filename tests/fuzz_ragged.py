#!/usr/bin/env python3
"""Randomised cross-check of the ragged seq_len-50 kernel pair against the full-row kernels (GPU box).

    python tests/fuzz_ragged.py [--cases 60] [--seed 0] [--oracle]

Per case: a random model kind, batch size (1 .. 700: below, at and above the 512-workgroup grid, so second sequences per
workgroup and the length-ordered selection with ties are hit), pad pattern (leading pads drawn from several distributions incl.
all-pad and pad-free sequences, interior pads, target ids on padded positions), dropout on / off.  The SAME fused training step
(one eager step from identical weights, deterministic scatter) runs under the default kernels and under SRFRD_NO_RAGGED=1 (the
switch is read per launch); compared: the loss (2e-6) and every stepped parameter with the bound the suite uses for one Adam
step from identical weights (|d| <= 2.2 lr, mean 2e-6); a batch without any target gives NaN under both, as in the reference.
The evaluation forward (hidden state, pos / neg logits) and the ranking forward's top-10 scores (2e-5) are compared the same way.  Exit code 0 = every case agrees.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

I, L = 400, 50


def make_batch(g, B, torch, L=L):
    seq = torch.randint(1, I + 1, (B, L), generator=g)
    pos = torch.randint(1, I + 1, (B, L), generator=g)
    neg = torch.randint(1, I + 1, (B, L), generator=g)
    mode = int(torch.randint(0, 4, (1,), generator=g))
    for b in range(B):
        if mode == 0:
            t0 = int(torch.randint(0, L + 1, (1,), generator=g))                 # uniform, all-pad included
        elif mode == 1:
            t0 = int(torch.randint(0, 6, (1,), generator=g))                      # long sequences: many ties at the top
        elif mode == 2:
            t0 = int(torch.randint(L - 6, L + 1, (1,), generator=g))              # short ones
        else:
            t0 = min(L, [0, 3, 4, 5, 19, 20, 21, 35, 36, 37, 49, 50][int(torch.randint(0, 12, (1,), generator=g))] * L // 50)   # tile boundaries (seq_len 50)
        seq[b, :t0] = 0
        if int(torch.randint(0, 5, (1,), generator=g)) != 0:
            pos[b, :t0] = 0
            neg[b, :t0] = 0
        if int(torch.randint(0, 6, (1,), generator=g)) == 0 and t0 + 3 < L:
            seq[b, t0 + 1 + int(torch.randint(0, L - t0 - 2, (1,), generator=g))] = 0
    rsq = torch.where(seq != 0, torch.randint(1, 3, (B, L), generator=g), torch.zeros_like(seq))
    prs = torch.where(pos != 0, torch.randint(1, 3, (B, L), generator=g), torch.zeros_like(pos))
    nrs = (neg != 0).long()
    return seq, rsq, pos, prs, neg, nrs


def against_oracle(a):
    bad = run_vs_oracle(a.cases, a.seed, L=a.seq_len, max_b=12 if a.seq_len <= 64 else 5)
    print(f"{a.cases - bad} of {a.cases} oracle cases agree")
    sys.exit(1 if bad else 0)


def run_vs_oracle(cases, seed, verbose=True, L=L, max_b=12):
    """--oracle: the default (ragged) kernels against the CPU oracle on random SMALL batches (1 .. 12 sequences): forward,
    every gradient through the autograd path, the ranking forward (predict) - at the suite's 1e-4.  Test infrastructure use of
    oracle/: this is a checker, nothing here is timed or shipped."""
    import torch
    from oracle import srfrd_oracle as O
    from tests.gpu_util import build_model, cuda, maxerr, random_sd
    g = torch.Generator().manual_seed(seed)
    bad = 0
    for case in range(cases):
        kind = ["SASRec", "SRFR", "SRFRN", "SRFU_B"][int(torch.randint(0, 4, (1,), generator=g))]
        B = int(torch.randint(1, max_b + 1, (1,), generator=g))
        cfg = O.Cfg(kind, I, L, 50) if kind == "SASRec" else (O.Cfg(kind, I, L, 45, d_fake=5) if kind in ("SRFR", "SRFRN")
                                                                else O.Cfg(kind, I, L, 50, n_labels=3))
        sd = random_sd(cfg, 100 + case)
        model = build_model(cfg, sd).train()
        batch = make_batch(g, B, torch, L)
        if bool((batch[2] == 0).all()):
            batch[2][0, L - 1] = 1 + case % I          # (the loss of a batch without targets is NaN: nothing to compare)
            batch[4][0, L - 1] = 1 + (case * 7) % I
        loss_o, grads_o, h_o, pl_o, nl_o = O.grads_of(cfg, sd, batch)
        seq, rsq, pos, prs, neg, nrs = cuda(*batch)
        h, pl, nl = model(None, seq, rsq, pos, prs, neg, nrs)
        idx = torch.where(pos != 0)
        crit = torch.nn.BCEWithLogitsLoss()
        loss = crit(pl[idx], torch.ones_like(pl)[idx]) + crit(nl[idx], torch.zeros_like(nl)[idx])
        loss.backward()
        cand = torch.arange(1, 102).repeat(B, 1)
        errs = dict(h=maxerr(h, h_o), pl=maxerr(pl, pl_o), nl=maxerr(nl, nl_o), loss=abs(float(loss.detach()) - float(loss_o)),
                    grad=max(maxerr(p.grad, grads_o[k]) for k, p in model.named_parameters()),
                    predict=maxerr(model.predict(None, seq, rsq, cand.cuda()), O.predict(cfg, sd, batch[0], batch[1], cand)))
        ok = max(errs.values()) < 1e-4
        bad += not ok
        if verbose or not ok:
            print(f"case {case:3d} {kind:7s} B={B:2d} " + " ".join(f"{k}={v:.1e}" for k, v in errs.items()) + ("" if ok else "   <-- MISMATCH"),
                  flush=True)
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--seq-len", type=int, default=L, help="--oracle only: other lengths run the long-sequence kernels (100: slots, 101 .. 208: row-chunked)")
    ap.add_argument("--oracle", action="store_true", help="small random batches against the CPU oracle instead of the full-row kernels")
    a = ap.parse_args()
    if a.oracle:
        return against_oracle(a)
    bad = run_vs_full(a.cases, a.seed)
    print(f"{a.cases - bad} of {a.cases} cases agree")
    sys.exit(1 if bad else 0)


def run_vs_full(cases, seed, verbose=True):
    import torch
    import srfrd_amd
    g = torch.Generator().manual_seed(seed)
    kinds = ["SASRec", "SRFR", "SRFRN", "SRFU_B"]
    bad = 0
    for case in range(cases):
        kind = kinds[int(torch.randint(0, 4, (1,), generator=g))]
        B = [1, 2, 7, 255, 256, 257, 511, 512, 513, 700][int(torch.randint(0, 10, (1,), generator=g))] if case % 2 == 0 \
            else int(torch.randint(1, 701, (1,), generator=g))
        p = [0.0, 0.5, 0.2][int(torch.randint(0, 3, (1,), generator=g))]
        batch = make_batch(g, B, torch)
        res = {}
        for name, env in (("ragged", None), ("full", "SRFRD_NO_RAGGED")):
            if env:
                os.environ[env] = "1"
            try:
                torch.manual_seed(1000 + case)
                if kind == "SASRec":
                    m = srfrd_amd.SASRec(I, L, 50, p, 2, 1, "cuda")
                elif kind == "SRFR":
                    m = srfrd_amd.SRFR(I, L, 45, 5, p, 2, 1, "cuda")
                elif kind == "SRFRN":
                    m = srfrd_amd.SRFRN(I, L, 45, 5, p, 2, 1, "cuda")
                else:
                    m = srfrd_amd.SRFU_B(I, L, 50, 3, p, 2, 1, "cuda")
                for _, q in m.named_parameters():
                    if q.dim() >= 2:
                        torch.nn.init.xavier_normal_(q.data)
                m = m.cuda()
                dev = [t.cuda() for t in batch]
                m.eval()
                with torch.no_grad():
                    h, pl, nl = m(None, dev[0], dev[1], dev[2], dev[3], dev[4], dev[5])
                    tk_i, tk_v = m.topk(None, dev[0], dev[1], k=10)          # the ranking forward (last position only)
                m.train()
                tr = srfrd_amd.FusedTrainer(m, B, L, seed=case, use_graph=False, deterministic=True)
                loss = float(tr.step(None, *dev).cpu())
                res[name] = dict(loss=loss, flat=tr.flat[:m.n_flat].cpu(), h=h.cpu(), pl=pl.cpu(), nl=nl.cpu(), tk_v=tk_v.cpu(), tk_i=tk_i.cpu())
            finally:
                if env:
                    del os.environ[env]
        r, f = res["ragged"], res["full"]
        d = (r["flat"] - f["flat"]).abs()
        errs = dict(loss=abs(r["loss"] - f["loss"]), wmax=float(d.max()), wmean=float(d.mean()),
                    h=float((r["h"] - f["h"]).abs().max()), pl=float((r["pl"] - f["pl"]).abs().max()),
                    nl=float((r["nl"] - f["nl"]).abs().max()), topk=float((r["tk_v"] - f["tk_v"]).abs().max()))
        ok = errs["loss"] < 2e-6 and errs["wmax"] <= 2.2e-3 and errs["wmean"] < 2e-6 and max(errs["h"], errs["pl"], errs["nl"], errs["topk"]) < 2e-5
        if r["loss"] != r["loss"] and f["loss"] != f["loss"]:
            # no position with a target in the whole batch: the mean over an empty selection is NaN in the reference too
            # (trainer.py:36-38), and so is everything the step touches - both kernels must say so
            ok = bool((batch[2] == 0).all()) and max(errs["h"], errs["pl"], errs["nl"], errs["topk"]) < 2e-5
            errs["loss"] = errs["wmax"] = errs["wmean"] = 0.0
        bad += not ok
        if verbose or not ok:
            print(f"case {case:3d} {kind:7s} B={B:3d} p={p:.1f} " + " ".join(f"{k}={v:.1e}" for k, v in errs.items()) + ("" if ok else "   <-- MISMATCH"),
                  flush=True)
    return bad


if __name__ == "__main__":
    main()
