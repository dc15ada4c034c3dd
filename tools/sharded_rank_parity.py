#!/usr/bin/env python3
"""One-shard-per-rank full-catalog ranking == unsharded ranking (srfrd_amd/ranker.py), N ranks.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 \\
        tools/sharded_rank_parity.py [--backend nccl|gloo] [--kind SASRec|SRFRN]

Every rank holds different users; ShardedRanker all-gathers their last hidden states, ranks all users against the rank's
own rows, all-gathers the (users, k) lists and merges.  Each rank compares its result with model.topk() over the whole
catalog (bit-equal indices and values).  One JSON line on rank 0; exit code 0 = parity.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default=None)
    ap.add_argument("--kind", default="SRFRN", choices=["SASRec", "SRFRN"])
    args = ap.parse_args()
    world, rank, local = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    n_dev = torch.cuda.device_count()
    backend = args.backend or ("nccl" if n_dev >= world else "gloo")
    dev = torch.device("cuda", local % max(n_dev, 1))
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    import srfrd_amd
    I, L, B, k = 20011, 30, 19, 10
    torch.manual_seed(0)
    m = srfrd_amd.SASRec(I, L, 50, 0.0, 2, 1, dev) if args.kind == "SASRec" else srfrd_amd.SRFRN(I, L, 45, 5, 0.0, 2, 1, dev)
    for _, p in m.named_parameters():
        if p.dim() >= 2:
            torch.nn.init.xavier_normal_(p.data)
    with torch.no_grad():
        w = m._item_param()
        w[[10000, 10006, 19000]] = w[5].clone()              # exact ties across the shard boundary
    m = m.to(dev).eval()
    _, seq, rsq, *_ = srfrd_amd.synthetic_batch(I, L, B, seed=3, rank=rank, device=dev)
    r = srfrd_amd.ShardedRanker(m)
    assert r.world == world and r.n_shards == world
    idx, val = r.topk(None, seq, rsq, k=k)
    wi, wv = m.topk(None, seq, rsq, k=k)
    ok = bool(torch.equal(idx, wi) and torch.equal(val, wv))
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN) if backend != "nccl" else None
    if backend == "nccl":
        f = flag.to(dev)
        dist.all_reduce(f, op=dist.ReduceOp.MIN)
        flag = f.cpu()
    if rank == 0:
        print(json.dumps({"backend": backend, "world": world, "kind": args.kind, "items": I, "users_per_rank": B, "k": k,
                          "shards": r.shards, "ok": bool(int(flag.item()) == 1)}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
