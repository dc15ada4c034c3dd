#!/usr/bin/env python3
"""End-to-end metric fixture (build container only): TRAIN the reference's own classes, then run the reference's own
``evaluation`` / ``evaluation_with_label`` (reference utils.py:544-602, 628-752) and store inputs + outputs as data.

    python -B tests/golden/make_golden_e2e.py          # writes tests/golden/e2e_<kind>.npz

What runs from the reference: ``SRFR_model.{SASRec,SRFRN,SRFU_B}`` (imported), ``df_data_partition``, ``evaluation``,
``evaluation_with_label`` (utils.py executed as text with only the unfinished ``partional`` class skipped, as SURVEY 8c
records - utils.py itself raises IndentationError on import).  The train loop is reference trainer.py:27-41 restated
(the file needs wandb + a CUDA device and cannot run), with ``dropout_rate = 0`` so that the result is a function of the
stored inputs alone.  Batches come from the oracle's restatement of ``sample_function_fr`` with a counter RNG
(``oracle.srfrd_oracle.sample_batch_ref``) - the same batches the device sampler produces from (seed, index), so the
fixture stores seeds, not batches.

The candidate lists ``evaluation`` draws with ``np.random.randint`` are captured by recording the arguments of
``model.predict`` (no re-derivation of the RNG call sequence).  Only data is written; no reference source is stored.

Conditioning: 300 Adam steps are a chaotic map for some of the model kinds - a 1e-7 relative perturbation of the initial
weights moves SRFRN / SRFU_B weights by 2e-3 and changes dozens of ranks, while SASRec stays at 1e-6.  No two fp32
implementations (the reference on a CPU and on a GPU included) can agree on such a run to better than the run agrees
with itself.  The fixture therefore records (a) the reference's metric after EARLY = 40 steps as well, where every kind
is still well conditioned, and (b) a second reference run from weights perturbed by PERTURB, evaluated on the same
candidates: ``cond_metric`` / ``cond_ranks_differ`` say how far the reference is from itself at the final step.

Ties: ``evaluation`` excludes only the TRAIN items from the negatives (utils.py:574-583), so a negative can be the
held-out item itself (28 % of the users at 300 items).  It scores what candidate 0 scores - up to one ulp, the
reference's BLAS dot products depend on the row position - and ``predictions.argsort().argsort()[0]`` (utils.py:591),
torch's UNSTABLE sort, puts candidate 0 anywhere among them.  The fixture stores the reference's reported ranks / metric
(``eval_rank``, ``eval_metric``), the same expression with ``stable=True`` (``eval_rank_stable``), the number of such
duplicates per user (``eval_tied``) and the reference's logits (``eval_logits``), from which tests/test_e2e_metric.py
derives the tie-independent rank (other items scoring strictly higher) that product and oracle are held to.
"""
import os
import random
import sys

import numpy as np
import pandas as pd
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
import SRFR_model as ref  # noqa: E402

from oracle import srfrd_oracle as O  # noqa: E402

N_USERS, N_ITEMS, L, B = 240, 300, 20, 32
STEPS = int(os.environ.get("E2E_STEPS", "300"))
EARLY = 40                 # the metric is also recorded here, while fp32 trajectories are still one computation
PERTURB = 1e-7             # relative size of the perturbation of the conditioning run (about one float32 ulp)
SAMPLER_SEED = 20261004
EVAL_SEED = 7
KINDS = {"SASRec": 0, "SRFRN": 1, "SRFU_B": 2}


def load_reference_utils():
    lines = open("/root/reference/utils.py").read().split("\n")
    src = "\n".join(lines[:141] + lines[194:])
    ns = {}
    exec(compile(src, "reference_utils", "exec"), ns)
    return ns


def make_rows(seed=3):
    """Interaction table with learnable structure: a user walks the catalogue with a personal stride, 25 % random jumps;
    30 % of the reviews are fake, fake-heavy users exist (so the B / F / R label groups are all populated)."""
    g = np.random.RandomState(seed)
    rows, when = [], []
    for u in range(1, N_USERS + 1):
        if u in (17, 101):
            n = 1                                        # users without a test item (skipped by evaluation)
        else:
            n = int(g.randint(3, 34))                    # some longer than maxlen
        cur, stride = int(g.randint(1, N_ITEMS + 1)), int(g.choice([1, 2, 3, 5]))
        p_fake = 0.8 if u % 7 == 0 else 0.25
        t = np.sort(g.rand(n))                           # timestamps: users interleave in the file, each stays in order
        for j in range(n):
            rows.append((u, cur, "fake" if g.rand() < p_fake else "real"))
            when.append(t[j])
            cur = int(g.randint(1, N_ITEMS + 1)) if g.rand() < 0.25 else (cur - 1 + stride) % N_ITEMS + 1
    return [rows[i] for i in np.argsort(np.array(when), kind="stable")]


def build(kind):
    if kind == "SASRec":
        return ref.SASRec(N_ITEMS, L, 50, 0.0, 2, 1, "cpu")
    if kind == "SRFRN":
        return ref.SRFRN(N_ITEMS, L, 45, 5, 0.0, 2, 1, "cpu")
    return ref.SRFU_B(N_ITEMS, L, 50, 3, 0.0, 2, 1, "cpu")


def main():
    torch.set_num_threads(1)
    ns = load_reference_utils()
    rows = make_rows()
    df = pd.DataFrame(rows, columns=["user_id", "item_id", "fake_review"])
    dataset = ns["df_data_partition"](df, is_valid=False)
    train, test, usernum, itemnum = dataset
    assert usernum == N_USERS and itemnum <= N_ITEMS
    train_items = {int(k): [int(x) for x in v] for k, v in train["item_ids"].items()}
    train_reviews = {int(k): [int(x) for x in v] for k, v in train["review_ids"].items()}
    base = {"rows_user": np.array([r[0] for r in rows], np.int32), "rows_item": np.array([r[1] for r in rows], np.int32),
            "rows_fake": np.array([r[2] == "fake" for r in rows], np.bool_),
            "meta": np.array([N_USERS, itemnum, L, B, STEPS, SAMPLER_SEED, EVAL_SEED, EARLY], np.int64)}
    for kind, k_i in KINDS.items():
        torch.manual_seed(4321 + k_i)
        model = build(kind)
        for _, p in model.named_parameters():              # reference trainer.py:364-369
            try:
                torch.nn.init.xavier_normal_(p.data)
            except Exception:
                pass
        out = dict(base)
        for k, v in model.state_dict().items():
            out["w0/" + k] = v.detach().numpy().copy()
        w_init = {k: v.detach().clone() for k, v in model.state_dict().items()}

        def train(model, n_from, n_to, losses):
            """reference trainer.py:27-41, restated (dropout_rate = 0), steps n_from .. n_to - 1"""
            model.train()
            for step in range(n_from, n_to):
                user, packed = O.sample_batch_ref(train_items, train_reviews, usernum, itemnum, B, L, SAMPLER_SEED, step)
                u = torch.from_numpy(user)
                seq, rsq, pos, prs, neg, nrs = (torch.from_numpy(packed[i]) for i in range(6))
                _, pl, nl = model(user_ids=u, input_ids=seq, fake_ids=rsq, positive_ids=pos, positive_fake_ids=prs,
                                  negative_ids=neg, negative_fake_ids=nrs)
                model._opt.zero_grad()
                idx = torch.where(pos != 0)
                loss = crit(pl[idx], torch.ones_like(pl)[idx]) + crit(nl[idx], torch.zeros_like(nl)[idx])
                for p in model.parameters():
                    loss = loss + 0.0 * torch.norm(p)
                loss.backward()
                model._opt.step()
                losses.append(loss.item())

        def evaluate(model, with_label):
            """the reference's own evaluation() / evaluation_with_label(), candidates captured at model.predict"""
            model.eval()
            rec = {"cand": [], "seq": [], "rsq": [], "logits": [], "rank_stable": []}
            orig_predict = model.predict

            def recording_predict(u_, seq_, rsq_, cand_):
                res = orig_predict(u_, seq_, rsq_, cand_)
                rec["cand"].append(cand_.numpy().copy())
                rec["seq"].append(seq_.numpy().copy()[0])
                rec["rsq"].append(rsq_.numpy().copy()[0])
                rec["logits"].append(res.detach().numpy().copy())
                rec["rank_stable"].append((-res).argsort(stable=True).argsort(stable=True)[0].item())   # utils.py:589-591, stable
                return res

            model.predict = recording_predict
            np.random.seed(EVAL_SEED)
            random.seed(EVAL_SEED)
            with torch.no_grad():
                res = ns["evaluation_with_label" if with_label else "evaluation"](model, dataset, L, "cpu")
            del model.predict
            return res, {k: np.stack(v) if k != "rank_stable" else np.array(v, np.int32) for k, v in rec.items()}

        crit = torch.nn.BCEWithLogitsLoss()
        model._opt = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.98))
        losses = []
        train(model, 0, EARLY, losses)
        (ndcg_e, hr_e), rec_e = evaluate(model, False)
        out["early_metric"] = np.array([ndcg_e, hr_e], np.float64)
        out["early_logits"] = rec_e["logits"].astype(np.float32)
        train(model, EARLY, STEPS, losses)
        out["loss_curve"] = np.array(losses, np.float32)
        for k, v in model.state_dict().items():
            out["wT/" + k] = v.detach().numpy().copy()
        (ndcg, hr), rec = evaluate(model, False)
        n_eval = len(rec["cand"])
        assert (rec["cand"] == rec_e["cand"]).all()        # same seed => the same candidates at both checkpoints
        out["eval_cand"] = rec["cand"].astype(np.int32)
        out["eval_seq"] = rec["seq"].astype(np.int32)
        out["eval_rsq"] = rec["rsq"].astype(np.int32)
        out["eval_logits"] = rec["logits"].astype(np.float32)
        out["eval_metric"] = np.array([ndcg, hr], np.float64)
        out["eval_rank_stable"] = rec["rank_stable"]
        out["eval_tied"] = (out["eval_cand"][:, 1:] == out["eval_cand"][:, :1]).sum(1).astype(np.int32)
        # evaluation_with_label on the same seed: per-user ranks and the per-label groups
        (ndcg2, hr2, per_user, m_b, m_f, m_r), rec2 = evaluate(model, True)
        assert (rec2["cand"] == rec["cand"]).all() and ndcg2 == ndcg and hr2 == hr
        users = sorted(per_user)
        out["eval_users"] = np.array(users, np.int32)
        out["eval_rank"] = np.array([per_user[u][0] for u in users], np.int32)
        out["eval_user_labels"] = np.array([[per_user[u][3], per_user[u][4], per_user[u][5]] for u in users], np.int32)
        for name, m in (("B", m_b), ("F", m_f), ("R", m_r)):
            out[f"label_metric_{name}"] = np.array([[k] + list(v) for k, v in m.items()], np.float64)   # label, HR, NDCG, n

        # ---- conditioning run: the reference again, from weights perturbed by about one ulp
        def base_ranks(logits, cand):
            return ((logits[:, 1:] > logits[:, :1]) & (cand[:, 1:] != cand[:, :1])).sum(1)

        def metric(r):
            hit = r < 10
            return np.array([np.where(hit, 1 / np.log2(r + 2.0), 0.0).mean(), hit.mean()])

        gen = torch.Generator().manual_seed(77)
        model2 = build(kind)
        model2.load_state_dict({k: v * (1 + PERTURB * torch.randn(v.shape, generator=gen)) for k, v in w_init.items()})
        model2._opt = torch.optim.Adam(model2.parameters(), lr=1e-3, betas=(0.9, 0.98))
        train(model2, 0, STEPS, [])
        _, rec_p = evaluate(model2, False)
        r_a, r_b = base_ranks(rec["logits"], rec["cand"]), base_ranks(rec_p["logits"], rec_p["cand"])
        out["cond_metric"] = np.abs(metric(r_a) - metric(r_b))
        out["cond_ranks_differ"] = np.array([int((r_a != r_b).sum())], np.int32)
        out["cond_logit_diff"] = np.array([np.abs(rec["logits"] - rec_p["logits"]).max()], np.float32)
        path = os.path.join(HERE, f"e2e_{kind}.npz")
        np.savez_compressed(path, **out)
        print(f"{kind}: loss {losses[0]:.4f} -> {losses[-1]:.4f}; users {n_eval} ({int((out['eval_tied'] > 0).sum())} tied); "
              f"step {EARLY}: NDCG@10 {ndcg_e:.4f} HR@10 {hr_e:.4f}; step {STEPS}: NDCG@10 {ndcg:.4f} HR@10 {hr:.4f}; reference vs itself "
              f"under a {PERTURB:g} perturbation: |dNDCG| {out['cond_metric'][0]:.2e} |dHR| {out['cond_metric'][1]:.2e}, "
              f"{int(out['cond_ranks_differ'][0])} ranks, logits {float(out['cond_logit_diff'][0]):.1e}; {os.path.getsize(path) // 1024} KiB")


if __name__ == "__main__":
    main()
