#!/usr/bin/env python3
"""Data-parallel parity of the REAL fused train step: N ranks on B / N sequences each == one rank on B sequences.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \\
        tools/dp_parity.py [--backend nccl|gloo] [--exchange sharded|allreduce] [--eager] [--kind SASRec] [--steps 4]

Backend: "nccl" (= RCCL) needs one GPU per rank; "gloo" lets all ranks share one GPU (the rehearsal form: same kernels,
same shard arithmetic, collectives through the host - srfrd_amd/exchange.py).

Rank 0 first trains alone (process group of itself => FusedTrainer's single-rank path) for `steps` steps on the whole
global batches, dropout ON, recording before every step the complete training state (parameters, Adam moments, step
counter / seed) and after it the loss, the gradient and the stepped parameters.  Then all ranks run the data-parallel
trainer: before step i every rank loads the recorded state i (its own slice of the moments in the sharded form), steps
once on its slice of global batch i (masks are keyed by the global sequence index, so they are the masks the single
process drew), and rank 0 compares
  * the loss (1e-5),
  * the stepped parameters, element-wise: 1e-4 or tighter wherever the gradient is real, relaxing to lr only where it is
    rounding noise (Adam normalises the magnitude of the gradient away, so the SIGN of a noise-level gradient - which
    depends on summation order: per-rank slabs and a cross-rank sum vs one slab set - is the step),
  * that all ranks hold bit-identical parameters after the all-gather / all-reduce.
Every step starts from the single run's state because a K-step free run cannot be compared: fp32 training through ReLU
is discontinuous - a one-ulp difference in one embedding element (float-atomic order in the item-table scatter) can
flip a unit that sits at its threshold and change that sequence's gradient by O(0.1), in the single-rank run against
ITSELF as much as against the data-parallel one (measured: tests/test_gpu_dp.py docstring).
Prints one JSON line on rank 0; exit code 0 = parity.  Started by the launcher before anything touches the GPU.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def adam_tolerance(grad_hist, lr=1e-3, noise=float(os.environ.get("DP_PARITY_NOISE", "2e-7")), base=1e-5):
    """element-wise bound on the weight difference after len(grad_hist) Adam steps (same rule as tests/helpers.py)"""
    import torch
    gabs = torch.stack([g.abs().double() for g in grad_hist])
    tol = base + len(grad_hist) * lr * torch.clamp(4.0 * noise / gabs.min(0).values.clamp_min(1e-300), max=1.0)
    return torch.where(gabs.max(0).values == 0, torch.full_like(tol, 1e-7), tol)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default=None)
    ap.add_argument("--exchange", default="sharded", choices=["sharded", "allreduce"])
    ap.add_argument("--eager", action="store_true")
    ap.add_argument("--kind", default="SASRec", choices=["SASRec", "SRFRN", "SRFU_B"])
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--batch", type=int, default=24, help="global batch (split evenly over the ranks)")
    ap.add_argument("--seq-len", type=int, default=50)
    ap.add_argument("--l2-emb", type=float, default=0.0, help="reference trainer.py:39 with a non-zero config.l2_emb")
    ap.add_argument("--spin-up", action="store_true", help="call FusedTrainer.spin_up() before every step (must change nothing)")
    ap.add_argument("--shadow-gather", action="store_true", help="bf16 item-table shadow; the all-gather carries the shadow, fp32 "
                    "master rows live on the owner rank only (FusedTrainer(shadow_gather=True)); the reference is the single-rank "
                    "bf16-table step")
    args = ap.parse_args()
    world, rank, local = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    n_dev = torch.cuda.device_count()            # (does not initialise the GPU on this image)
    backend = args.backend or ("nccl" if n_dev >= world else "gloo")
    dev = torch.device("cuda", local % max(n_dev, 1))
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    solo = dist.new_group([0])                   # (collective call: every rank makes it)
    import srfrd_amd
    I, L, Bg = 400, args.seq_len, args.batch
    assert Bg % world == 0
    Bl = Bg // world

    def make_model():
        torch.manual_seed(0)                     # identical replicas
        if args.kind == "SASRec":
            m = srfrd_amd.SASRec(I, L, 50, 0.5, 2, 1, dev)
        elif args.kind == "SRFRN":
            m = srfrd_amd.SRFRN(I, L, 45, 5, 0.5, 2, 1, dev)
        else:
            m = srfrd_amd.SRFU_B(I, L, 50, 3, 0.5, 2, 1, dev)
        for _, p in m.named_parameters():
            if p.dim() >= 2:
                torch.nn.init.xavier_normal_(p.data)
        m = m.to(dev).train()
        if args.shadow_gather:
            m.use_bf16_table()
        return m

    batches = [srfrd_amd.synthetic_batch(I, L, Bg, seed=5, index=i, device=dev, packed=True)[1] for i in range(args.steps)]
    model = make_model()
    n_flat = model.n_flat
    # ---- rank 0 alone: the reference trajectory and its per-step states (eager: the gradient is looked at before Adam uses it)
    K = args.steps
    pre_flat = torch.zeros(K, n_flat, device=dev)
    pre_m, pre_v = torch.zeros(K, n_flat, device=dev), torch.zeros(K, n_flat, device=dev)
    pre_state = torch.zeros(K, 32, device=dev, dtype=torch.int32)
    post_flat, grads, ref_loss = torch.zeros(K, n_flat, device=dev), [], []
    if rank == 0:
        ref = make_model()
        rt = srfrd_amd.FusedTrainer(ref, Bg, L, seed=17, use_graph=False, process_group=solo, l2_emb=args.l2_emb)
        assert rt.world == 1 and rt.mode == "single"
        rt.refresh()
        for i in range(K):
            pre_flat[i], pre_m[i], pre_v[i], pre_state[i] = rt.flat, rt.m, rt.v, rt.state
            rt.ids.copy_(batches[i])
            rt._enqueue_compute()
            grads.append((rt.grad[:n_flat] / float(rt.stats[2].cpu())).detach().clone())
            rt._enqueue_update()
            ref_loss.append(float(rt.loss.cpu()))
            post_flat[i] = rt.flat
        torch.cuda.synchronize()
    for t in (pre_flat, pre_m, pre_v, pre_state):
        if backend == "nccl":
            dist.broadcast(t, src=0)
        else:
            c = t.cpu()
            dist.broadcast(c, src=0)
            t.copy_(c)
    # ---- the data-parallel trainer, every step from the recorded state
    tr = srfrd_amd.FusedTrainer(model, Bl, L, seed=17, use_graph=not args.eager, exchange=args.exchange, l2_emb=args.l2_emb,
                                shadow_gather=args.shadow_gather)
    assert tr.shadow_gather == (args.shadow_gather and args.exchange == "sharded")
    assert tr.world == world and tr.mode == args.exchange
    dp_loss, dp_post = [], []
    for i in range(K):
        tr.flat.copy_(pre_flat[i])
        if tr.mode == "sharded":
            for dst, src in ((tr.m, pre_m[i]), (tr.v, pre_v[i])):
                dst.zero_()
                n_own = max(0, min(tr.ex.i1, n_flat) - tr.ex.i0)
                dst[:n_own].copy_(src[tr.ex.i0:tr.ex.i0 + n_own])
        else:
            tr.m.copy_(pre_m[i]); tr.v.copy_(pre_v[i])
        tr.state.copy_(pre_state[i])
        tr.refresh()
        model.refresh_bf16_table()                 # (the recorded fp32 state is complete on every rank: derive the shadow from it)
        if args.spin_up:
            tr.spin_up(3)
        dp_loss.append(float(tr.step_packed(batches[i][:, rank * Bl:(rank + 1) * Bl].contiguous()).cpu()))
        if tr.shadow_gather:
            # the gathers of the NEXT step would read this shadow: it must equal bf16(master) for every row, on every rank
            shadow_now = model._table16.clone()
            tr.sync_master()                       # (fp32 rows outside the own shard are stale until asked for)
            want = model.flat_parameters()[:model.layout.n_table].to(torch.bfloat16).view(torch.int16)
            assert torch.equal(shadow_now, want), "gathered bf16 shadow != bf16(all-gathered fp32 master)"
        dp_post.append(model.flat_parameters().detach().clone())
    tr.check()
    # replicas identical?
    last = dp_post[-1] if backend == "nccl" else dp_post[-1].cpu()
    allp = [torch.empty_like(last) for _ in range(world)]
    dist.all_gather(allp, last)
    allp = [x.cpu() for x in allp]
    ok, report = True, {}
    if rank == 0:
        replicas_equal = all(torch.equal(allp[0], g) for g in allp[1:])
        lay, D = model.layout, model.layout.D
        n_viol, worst, tight = 0, 0.0, []
        where = {}
        names = {id(p): n for n, p in model.named_parameters()}
        for i in range(K):
            d = (dp_post[i].double() - post_flat[i].double()).abs()
            tol = adam_tolerance([grads[i]]).to(d.device)
            for b in range(lay.n_blocks):     # K slice of in_proj_bias: true gradient identically zero (tests/helpers.drop_kbias)
                k0 = model.n_table_pad + lay.blk[b].in_b + D
                tol[k0:k0 + D] = 1.0
            n_viol += int((d > tol).sum())
            for p_, off in model._slots:          # (diagnostic: which tensors hold the violations)
                c = int((d[off:off + p_.numel()] > tol[off:off + p_.numel()]).sum())
                if c:
                    where[f"step{i}:{names[id(p_)]}"] = c
            worst = max(worst, float(d.max()))
            tight.append(float((tol <= 1e-4).double().mean()))
        loss_diff = max(abs(a - b) for a, b in zip(dp_loss, ref_loss))
        report = {"backend": backend, "world": world, "exchange": args.exchange, "graph": not args.eager, "kind": args.kind,
                  "steps": K, "global_batch": Bg, "seq_len": L, "dropout": 0.5, "dp_loss": dp_loss, "single_loss": ref_loss,
                  "max_loss_diff": loss_diff, "max_weight_diff": worst,
                  "weights_held_to_1e-4_or_tighter": min(tight), "weight_violations": n_viol,
                  "replicas_bit_identical": bool(replicas_equal), "violations_in": where, "shadow_gather": bool(tr.shadow_gather)}
        ok = loss_diff < 1e-5 and n_viol == 0 and replicas_equal
        report["ok"] = bool(ok)
        print(json.dumps(report), flush=True)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
    dist.broadcast(flag, src=0)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
