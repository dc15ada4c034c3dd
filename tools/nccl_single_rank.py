#!/usr/bin/env python3
"""The RCCL arms of the data-parallel step, executed on ONE GPU: a process group of one rank on backend "nccl"
(= RCCL) with SRFRD_FORCE_EXCHANGE=1, so that FusedTrainer takes its `sharded` / `allreduce` code (reduce_scatter_tensor, the
in-place all_gather_into_tensor, the asynchronous 16-byte statistics all-reduce, HIP-graph capture with the collectives
inside - or graph replays beside an RCCL communicator in the split form) and ShardedRanker its nccl all-gathers.

With one rank every collective is an identity, so each form must reproduce the single-rank step: K steps with dropout on
and the DETERMINISTIC item-table scatter (float-atomic order is the one thing that differs between two runs of the same
code) - loss and every parameter are compared bit for bit (the sharded form steps with srfrd_adam_step + srfrd_pack_weights
where the single path runs the fused srfrd_adam_pack_step: same arithmetic per element).

    python tools/nccl_single_rank.py            (prints one JSON line; exit 0 = all forms equal the single path)

A child process of tests/test_gpu_nccl_single.py: the process group must not leak into the pytest process.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import srfrd_amd
    I, L, B, K = 400, 50, 24, 3

    def make_model(kind="SASRec"):
        torch.manual_seed(0)
        m = srfrd_amd.SASRec(I, L, 50, 0.5, 2, 1, dev) if kind == "SASRec" else srfrd_amd.SRFRN(I, L, 45, 5, 0.5, 2, 1, dev)
        for _, p in m.named_parameters():
            if p.dim() >= 2:
                torch.nn.init.xavier_normal_(p.data)
        return m.to(dev).train()

    batches = [srfrd_amd.synthetic_batch(I, L, B, seed=5, index=i, device=dev, packed=True)[1] for i in range(K)]

    def run(exchange, graph, forced, slots=1):
        os.environ["SRFRD_FORCE_EXCHANGE"] = "1" if forced else "0"
        m = make_model()
        tr = srfrd_amd.FusedTrainer(m, B, L, seed=17, use_graph=graph, exchange=exchange, deterministic=True, slots=slots)
        losses = []
        for i in range(K):
            losses.append(float(tr.step_packed(batches[i]).cpu()))
        torch.cuda.synchronize()
        return tr, losses, tr.flat[:m.n_flat].detach().clone()

    ref_tr, ref_loss, ref_flat = run("sharded", False, forced=False)
    assert ref_tr.mode == "single"
    report = {"backend": dist.get_backend(), "world": dist.get_world_size(), "forms": {}, "ok": True}
    for exchange in ("sharded", "allreduce"):
        for graph in (True, False):
            tr, losses, flat = run(exchange, graph, forced=True)
            same_loss = losses == ref_loss
            same_w = bool(torch.equal(flat, ref_flat))
            diff = float((flat - ref_flat).abs().max())
            key = f"{exchange}/{'graph' if graph else 'eager'}"
            report["forms"][key] = {"mode": tr.mode, "native": tr.ex.native, "graph_form": tr.graph_form,
                                    "capture_error": tr.graph_capture_error, "loss_bit_equal": same_loss,
                                    "weights_bit_equal": same_w, "max_weight_diff": diff}
            ok = tr.mode == exchange and tr.ex.native and same_loss and same_w
            report["ok"] = report["ok"] and ok
    # the split-graph form too (graphs replayed beside the communicator, collectives launched from the host)
    os.environ["SRFRD_DP_SPLIT_GRAPHS"] = "1"
    tr, losses, flat = run("sharded", True, forced=True)
    report["forms"]["sharded/split-graphs"] = {"graph_form": tr.graph_form, "loss_bit_equal": losses == ref_loss,
                                               "weights_bit_equal": bool(torch.equal(flat, ref_flat))}
    report["ok"] = report["ok"] and tr.graph_form == "split" and losses == ref_loss and bool(torch.equal(flat, ref_flat))
    os.environ.pop("SRFRD_DP_SPLIT_GRAPHS")
    # shadow_gather (SURVEY 8e for BASELINE configs[4]): bf16 item-table shadow all-gathered instead of the fp32 vector
    def run_bf16(forced, shadow):
        os.environ["SRFRD_FORCE_EXCHANGE"] = "1" if forced else "0"
        m = make_model()
        m.use_bf16_table()
        tr = srfrd_amd.FusedTrainer(m, B, L, seed=17, use_graph=True, exchange="sharded", deterministic=True, shadow_gather=shadow)
        losses = [float(tr.step_packed(batches[i]).cpu()) for i in range(K)]
        tr.sync_master()
        torch.cuda.synchronize()
        return tr, losses, tr.flat[:m.n_flat].detach().clone(), m._table16.clone()

    _, l0, f0, s0 = run_bf16(False, False)
    tr, l1, f1, s1 = run_bf16(True, True)
    ok = tr.shadow_gather and l0 == l1 and bool(torch.equal(f0, f1)) and bool(torch.equal(s0, s1))
    report["forms"]["sharded/shadow_gather"] = {"graph_form": tr.graph_form, "loss_bit_equal": l0 == l1,
                                                "weights_bit_equal": bool(torch.equal(f0, f1)), "shadow_bit_equal": bool(torch.equal(s0, s1))}
    report["ok"] = report["ok"] and ok
    # row-sharded ranking: the nccl arm of ranker._all_gather (one shard = the whole catalog at world 1)
    os.environ["SRFRD_FORCE_EXCHANGE"] = "1"
    for kind in ("SASRec", "SRFRN"):
        m = make_model(kind).eval()
        _, seq, rsq, *_ = srfrd_amd.synthetic_batch(I, L, B, seed=9, device=dev)
        r = srfrd_amd.ShardedRanker(m)
        assert r.dist_on and r.world == 1
        i1, v1 = r.topk(None, seq, rsq, k=10)
        i0, v0 = m.topk(None, seq, rsq, k=10)
        same = bool(torch.equal(i0, i1)) and bool(torch.equal(v0, v1))
        report["forms"][f"ranker/{kind}"] = {"equal_unsharded": same}
        report["ok"] = report["ok"] and same
    print(json.dumps(report), flush=True)
    dist.destroy_process_group()
    return 0 if report["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
