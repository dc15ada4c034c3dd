#!/usr/bin/env python3
"""Golden fixture for the dataset side (build container only): runs the reference's own ``df_data_partition`` and label
helpers (reference utils.py:92-139, 604-626) on a small synthetic interaction table and stores inputs + outputs as JSON.

``utils.py`` cannot be imported as a module (IndentationError in the unfinished ``partional`` class, utils.py:142-194),
so - as SURVEY.md 8c records - its text is executed with only that class skipped.  Nothing but data is written.
"""
import json
import os

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference_utils():
    lines = open("/root/reference/utils.py").read().split("\n")
    src = "\n".join(lines[:141] + lines[194:])
    ns = {}
    exec(compile(src, "reference_utils", "exec"), ns)
    return ns


def main():
    ns = load_reference_utils()
    g = np.random.RandomState(11)
    rows = []
    n_users, n_items = 23, 40
    for u in range(1, n_users + 1):
        if u == 7:
            continue                                   # a user id with no rows at all
        n = 1 if u in (3, 12) else (2 if u == 5 else int(g.randint(3, 15)))
        for _ in range(n):
            rows.append((u, int(g.randint(1, n_items + 1)), "fake" if g.rand() < 0.3 else "real"))
    rows = [rows[i] for i in g.permutation(len(rows))]   # interleave users: file order matters, not grouping
    df = pd.DataFrame(rows, columns=["user_id", "item_id", "fake_review"])
    out = {"rows": rows}
    for is_valid in (False, True):
        train, test, usernum, itemnum = ns["df_data_partition"](df, is_valid=is_valid)
        out[f"valid{int(is_valid)}"] = {
            "usernum": int(usernum), "itemnum": int(itemnum),
            "train_items": {str(k): [int(x) for x in v] for k, v in train["item_ids"].items()},
            "train_reviews": {str(k): [int(x) for x in v] for k, v in train["review_ids"].items()},
            "test_items": {str(k): [int(x) for x in v] for k, v in test["item_ids"].items()},
            "test_reviews": {str(k): [int(x) for x in v] for k, v in test["review_ids"].items()},
        }
    windows = [[0, 0, 1, 2, 2], [1, 1, 1, 2, 2], [0, 0, 0, 1, 2], [2, 2, 2, 2, 2], [1, 1, 1, 1, 1], [0, 0, 0, 0, 2]]
    out["label_windows"] = windows
    out["binary"] = [int(ns["get_binary_label"](np.array(w))) for w in windows]
    out["frequency"] = [int(ns["get_frequency_label"](np.array(w))) for w in windows]
    out["ratio"] = [int(ns["get_ratio_label"](np.array(w))) for w in windows]
    json.dump(out, open(os.path.join(HERE, "dataset_partition.json"), "w"))
    print("rows", len(rows), "users", out["valid0"]["usernum"], "items", out["valid0"]["itemnum"])


if __name__ == "__main__":
    main()
