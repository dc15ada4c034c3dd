#!/usr/bin/env python3
"""bench.py -- sequences/sec of the fused SRFRD train step (reference trainer.py:27-41) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
        N > 1: one rank per GPU over RCCL.  Under torch.distributed.run (WORLD_SIZE set) this process IS a rank; started
        plainly, it launches `python -m torch.distributed.run --nproc-per-node N bench.py ...` itself as a child (before any
        GPU call), relays the child's output and exits with its status.

Workload (BASELINE.json configs[1], "C2"): SASRec (discriminator off), 50 000 items, seq_len 50, batch 512 per GPU,
hidden 50, 2 blocks, 1 head, dropout 0.5, Adam(1e-3, betas=(0.9, 0.98)), fp32 arithmetic, synthetic ids already in HBM.
One step = forward + masked BCE + backward + dense Adam (+ gradient all-reduce over RCCL when N > 1); weak scaling.
Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` and, at N = 1, `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

C2 = dict(kind="SASRec", n_items=50_000, seq_len=50, batch=512, hidden=50, blocks=2, heads=1, dropout=0.5)
# the other BASELINE.json configs: parity-test cases, selectable for exploration only (--workload); never the default
WORKLOADS = {
    "C2": C2,
    "C3": dict(kind="SRFRN", n_items=50_000, seq_len=50, batch=512, hidden=45, fake=5, blocks=2, heads=1, dropout=0.5),
    "C3u": dict(kind="SRFU_B", n_items=50_000, seq_len=50, batch=512, hidden=50, labels=3, blocks=2, heads=1, dropout=0.5),
    "C4": dict(kind="SASRec", n_items=200_000, seq_len=100, batch=512, hidden=50, blocks=2, heads=1, dropout=0.5),
    "C5": dict(kind="SASRec", n_items=1_000_000, seq_len=200, batch=512, hidden=50, blocks=2, heads=1, dropout=0.5),
    # the bf16 item-table shadow (configs[1] / [4] say "bf16"): a SECOND workload, never reported in place of the fp32 line
    "C2_bf16_table": dict(C2, bf16_table=True),
    "C5_bf16_table": dict(kind="SASRec", n_items=1_000_000, seq_len=200, batch=512, hidden=50, blocks=2, heads=1, dropout=0.5,
                          bf16_table=True),
}
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: peak FP32 (matrix) dense
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E spec


def algorithmic_flops_per_token(L, D, d_item, nb):
    """forward flops per token = nb (12 D^2 + 4 L D) + 2 D d_i + 4 d_i."""
    return nb * (12 * D * D + 4 * L * D) + 2 * D * d_item + 4 * d_item


def algorithmic_bytes_per_step(I, L, B, D, d_item, nb, p_dense):
    """SURVEY.md 8(d) `Algorithmic bytes per train step per GPU`, fp32 gathers (s_e = 4)."""
    T = B * L
    ids = 6 * T * 8 + B * 8
    gather_fwd = 3 * T * d_item * 4
    outputs = T * D * 4 + 2 * T * 4
    regather_bwd = 3 * T * d_item * 4
    grad_scatter = 2 * 3 * T * d_item * 4
    grad_zero = (I + 1) * d_item * 4
    adam_table = 7 * (I + 1) * d_item * 4
    adam_dense = 7 * p_dense * 4
    return ids + gather_fwd + outputs + regather_bwd + grad_scatter + grad_zero + adam_table + adam_dense


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_cores():
    """CPU threads this process may really use: affinity, capped by the cgroup CPU quota (a one-GPU box exposes
    every host core but grants a 16-CPU share; oversubscribing OpenMP there stalls for minutes)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
            if q != "max":
                quota = int(q) / int(per)
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = int(f.read()), int(g.read())
                if q > 0:
                    quota = q / per
        except Exception:
            pass
    if quota is not None:
        n = max(1, min(n, int(quota)))
    return max(1, min(n, int(os.environ.get("SRFRD_CPU_THREADS", "16"))))


def pmc_traffic(kernel_c_name):
    """(HBM bytes per launch of the dominant kernel, source file) from the committed rocprofv3 PMC passes (FETCH_SIZE /
    WRITE_SIZE, separate passes, read side doubled as MI355X_MICROARCH.md prescribes for gfx950) - NOT measured in this run:
    counters need their own profiler passes; (None, None) if no profile is committed."""
    ks = src = None
    for name in ("r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        try:
            ks = json.load(open(os.path.join(ROOT, "profiles", name)))["kernels"]
            src = "profiles/" + name
            break
        except Exception:
            continue
    if ks is None:
        return None, None
    stem = kernel_c_name.replace("srfrd_", "")                      # srfrd_encoder_bwd -> encoder_bwd[_slots]_kernel<...>
    for name, d in ks.items():
        if name.startswith(stem) and "_kernel" in name and "hbm_bytes_per_launch" in d and "<0, 0, 0>" not in name:
            return d["hbm_bytes_per_launch"]["total"], src
    return None, None


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 outside torch.distributed.run: start the N ranks as a child process.  Nothing in
    THIS process has touched the GPU (an exec / fork after HIP initialisation is not allowed on this pool); the child's
    stderr passes through, rank 0's JSON line is relayed on stdout, the exit status is the child's."""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()           # (counting devices does not initialise HIP)
    backend = os.environ.get("SRFRD_DIST_BACKEND", "nccl")
    if n_dev < args.gpus and backend == "nccl":
        raise SystemExit(f"--gpus {args.gpus}: this node exposes {n_dev} GPU(s); refusing to report a smaller job under that name "
                         "(SRFRD_DIST_BACKEND=gloo rehearses the multi-rank path on fewer GPUs, labelled as such)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"launching {args.gpus} ranks: {' '.join(cmd)}")
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, cwd=ROOT)
    line = None
    for ln in child.stdout:
        if ln.startswith("{") and '"metric"' in ln:
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = child.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        rc = 1
    raise SystemExit(rc)


def metric_parity():
    """End-to-end HR@10 / NDCG@10 against the REFERENCE (tests/golden/e2e_SASRec.npz: the reference's SASRec trained for 300
    restated trainer.py steps, then its own evaluation()): the same run through FusedTrainer + DeviceSampler + batched
    evaluation on this GPU.  -> {|dHR@10|, |dNDCG@10|, ...} or None if the fixture is absent.  Not part of the timed region."""
    import numpy as np
    import srfrd_amd
    path = os.path.join(ROOT, "tests", "golden", "e2e_SASRec.npz")
    if not os.path.exists(path):
        return None
    z = np.load(path)
    g = {k: z[k] for k in z.files}
    n_users, itemnum, L, B, steps, sampler_seed, _, early = (int(x) for x in g["meta"])
    w0 = {k[3:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith("w0/")}
    n_items = w0["item_emb.weight"].shape[0] - 1
    model = srfrd_amd.SASRec(n_items, L, 50, 0.0, 2, 1, "cuda")
    model.load_state_dict(w0)
    model = model.cuda().train()
    data = srfrd_amd.partition(g["rows_user"], g["rows_item"], g["rows_fake"])
    tr = srfrd_amd.FusedTrainer(model, B, L, use_graph=True)
    sampler = srfrd_amd.DeviceSampler(data, B, L, seed=sampler_seed, model=tr)
    for _ in range(steps):
        sampler.next_batch(out=tr.ids_ring[0])
        tr.step_slot(0)
    ndcg, hr = srfrd_amd.evaluation(model, data, L, candidates=g["eval_cand"])
    # the reference's metric with ties of the held-out item against its own duplicates among the negatives removed (the
    # reference's unstable argsort places them arbitrarily; tests/test_e2e_metric.py explains), from the reference's logits
    lg, cand = g["eval_logits"], g["eval_cand"]
    r = ((lg[:, 1:] > lg[:, :1]) & (cand[:, 1:] != cand[:, :1])).sum(1)
    hit = r < 10
    hr_ref, ndcg_ref = float(hit.mean()), float(np.where(hit, 1 / np.log2(r + 2.0), 0.0).mean())
    return {"model": "SASRec", "train_steps": steps, "eval_users": int(len(r)), "hr10": hr, "hr10_reference": hr_ref,
            "abs_diff_hr10": abs(hr - hr_ref), "ndcg10": ndcg, "ndcg10_reference": ndcg_ref, "abs_diff_ndcg10": abs(ndcg - ndcg_ref),
            "hr10_reference_as_reported": float(g["eval_metric"][1])}


def cpu_baseline(cfg, budget_s=12.0):
    """The oracle's restated trainer.py:27-41 step (torch CPU ops, dropout on) timed on this host's cores."""
    from oracle import srfrd_oracle as O
    import srfrd_amd
    n_thr = host_cores()
    torch.set_num_threads(n_thr)
    log(f"cpu_baseline: {n_thr} threads")
    ocfg = O.Cfg(cfg["kind"], cfg["n_items"], cfg["seq_len"], cfg["hidden"], num_blocks=cfg["blocks"],
                 num_heads=cfg["heads"], dropout=cfg["dropout"])
    torch.manual_seed(0)
    m = srfrd_amd.SASRec(cfg["n_items"], cfg["seq_len"], cfg["hidden"], cfg["dropout"], cfg["blocks"], cfg["heads"], "cpu")
    for _, p in m.named_parameters():
        if p.dim() >= 2:
            torch.nn.init.xavier_normal_(p.data)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    stepper = O.TorchStep(ocfg, sd)
    batch = srfrd_amd.synthetic_batch(cfg["n_items"], cfg["seq_len"], cfg["batch"], seed=1, index=0, device="cpu")[1:]
    for _ in range(2):
        stepper.step(batch)
    log("cpu_baseline: warm")
    n, t0 = 0, time.perf_counter()
    while True:
        stepper.step(batch)
        n += 1
        el = time.perf_counter() - t0
        if (n >= 10 and el > budget_s) or n >= 200 or el > 3 * budget_s:
            break
    log(f"cpu_baseline: {n} steps in {el:.1f} s")
    return {"value": cfg["batch"] * n / el, "unit": "sequences/s", "cores": n_thr, "kind": "port",
            "sample": f"{n} train steps of the same C2 batch shape (B={cfg['batch']}, L={cfg['seq_len']}, "
                      f"I={cfg['n_items']}), {el:.1f} s, torch CPU fp32, dropout 0.5"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP-graph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-metric-parity", action="store_true", help="skip the end-to-end HR@10 run (profiling passes: keeps the "
                    "per-kernel averages of the profile free of that run's small launches)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the C3 / C4 / C5 side measurements (config.other_workloads_ms)")
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS), help="exploration only; the contract line is C2")
    ap.add_argument("--deterministic", action="store_true", help="deterministic item-table scatter (sort + ordered sums) instead of float atomics")
    ap.add_argument("--autograd", action="store_true", help="exploration: time the module-level drop-in path (model(...) -> BCE -> "
                    "loss.backward() -> torch.optim.Adam) instead of FusedTrainer")
    ap.add_argument("--torch-adam", action="store_true", help="with --autograd: torch.optim.Adam instead of srfrd_amd.Adam")
    ap.add_argument("--profile-host", action="store_true", help="with --autograd: print a torch.profiler table of the loop (where the host time goes)")
    ap.add_argument("--predict", action="store_true", help="time forward + full-catalog top-10 instead of the train step")
    ap.add_argument("--spin-up", type=int, default=200, help="untimed replays of the captured step inside a state snapshot before the "
                    "W warm-up steps (steady-state clocks / caches; state restored bit for bit; reported in config.spin_up_replays)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)                      # never returns
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the line must report the job that ran")
    n_dev = torch.cuda.device_count()
    local = local % max(n_dev, 1)             # rehearsal of the multi-rank path on a one-GPU box (SRFRD_DIST_BACKEND=gloo)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    backend = "none"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SRFRD_DIST_BACKEND", "nccl")      # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import srfrd_amd
    cfg = WORKLOADS[args.workload]
    model = build_model(cfg, dev)
    B, L = cfg["batch"], cfg["seq_len"]
    if args.predict or args.autograd or args.workload != "C2":
        return explore(args, cfg, model, dev, rank)
    # eight synthetic batches resident in the trainer's input ring before the timed region starts (the contract's
    # "inputs already in HBM"): each step consumes one slot in place, as it would a slot a device sampler just filled
    ring = 1 if os.environ.get("SRFRD_BENCH_COPY") else 8
    tr = srfrd_amd.FusedTrainer(model, B, L, lr=1e-3, betas=(0.9, 0.98), seed=42, use_graph=not args.no_graph, slots=ring,
                                exchange=os.environ.get("SRFRD_DP_EXCHANGE", "sharded"), deterministic=args.deterministic)
    batches = [srfrd_amd.synthetic_batch(cfg["n_items"], L, B, seed=1, index=i, rank=rank, device=dev, packed=True)[1]
               for i in range(8)]
    if ring == 8:
        for i in range(8):
            tr.ids_ring[i].copy_(batches[i])
    step_i = (lambda i: tr.step_slot(i % 8)) if ring == 8 else (lambda i: tr.step_packed(batches[i % 8]))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log("model + batches ready")
    # device to steady state inside a snapshot (no training: state restored), then the W warm-up steps asked for.  The replays
    # are untimed work ahead of the timed region and are REPORTED (config.spin_up_replays); --spin-up 0 turns them off.
    tr.spin_up(args.spin_up)
    for i in range(args.warmup):
        step_i(i)
    barrier()
    log("warm-up done")
    t0 = time.perf_counter()
    for i in range(args.steps):
        step_i(i)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss = float(tr.loss.cpu())
    log(f"timed region: {elapsed:.4f} s for {args.steps} steps, loss {loss:.5f}")

    # ---- dominant-kernel timing: HIP events around each launch of the same K steps, eager, on the launch stream
    kt = {"srfrd_encoder_fwd": 0.0, "srfrd_encoder_bwd": 0.0}
    if rank == 0:
        kt = time_kernels(tr, batches, min(args.steps, 50))
        log(f"kernel ms: {kt}")

    if rank == 0:
        lay = model.layout
        T = B * L
        fwd_fl = algorithmic_flops_per_token(L, lay.D, lay.d_item, lay.n_blocks) * T
        dom = max(("srfrd_encoder_fwd", "srfrd_encoder_bwd"), key=lambda k: kt[k])
        dom_flops = fwd_fl * (2 if dom.endswith("bwd") else 1)     # backward = 2x forward (train ~ 3x fwd)
        dom_s = kt[dom] * 1e-3
        achieved = dom_flops / dom_s / 1e12 if dom_s > 0 else 0.0
        step_bytes = algorithmic_bytes_per_step(cfg["n_items"], L, B, lay.D, lay.d_item, lay.n_blocks, lay.n_dense)
        ms = elapsed / args.steps * 1e3
        traffic, traffic_src = pmc_traffic(dom)
        valid = float((batches[0][0] != 0).float().mean())       # non-pad share of the B x L token grid (lengths ~ U[2, L])
        out = {
            "metric": "sequences/sec", "value": world * B * args.steps / elapsed, "unit": "sequences/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C2: SASRec train step (fwd + masked BCE + bwd + dense Adam), 50k items, seq_len 50, "
                                   "batch 512 per GPU, hidden 50, 2 blocks, 1 head, dropout 0.5",
                       "global_batch": world * B, "seq_len": L, "n_items": cfg["n_items"],
                       "parallelism": f"dp{world}", "ranks": dist.get_world_size() if world > 1 else 1, "backend": "rccl" if backend == "nccl" else backend,
                       "exchange": tr.mode, "graph": not args.no_graph, "graph_form": tr.graph_form, "spin_up_replays": args.spin_up if not args.no_graph else 0,
                       "valid_token_fraction": valid, "sequence_schedule": {0: "none", 1: "length order (long + short sequence per CU)"}[tr.sched_mode],
                       "table_scatter": "sort + ordered sums" if args.deterministic else "float atomics",
                       "final_loss": loss},
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "avg_kernel_ms": kt[dom], "algorithmic_flops_per_launch": dom_flops,
                         "kernel_ms": kt,
                         "step_hbm": {"bound": "hbm", "algorithmic_bytes_per_step": step_bytes,
                                      "achieved": step_bytes / (ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                      "frac": step_bytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS}},
        }
        if world == 1 and not args.no_metric_parity:
            try:
                out["config"]["metric_parity"] = metric_parity()
            except Exception as e:          # never lose the throughput line to the side measurement
                out["config"]["metric_parity"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        if world == 1 and not args.no_secondary:
            del tr
            torch.cuda.empty_cache()
            out["config"]["other_workloads_ms"] = other_workloads(dev)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def build_model(cfg, dev):
    import srfrd_amd
    torch.manual_seed(0)                      # identical init on every rank (replicated parameters)
    if cfg["kind"] == "SASRec":
        model = srfrd_amd.SASRec(cfg["n_items"], cfg["seq_len"], cfg["hidden"], cfg["dropout"], cfg["blocks"], cfg["heads"], dev)
    elif cfg["kind"] == "SRFRN":
        model = srfrd_amd.SRFRN(cfg["n_items"], cfg["seq_len"], cfg["hidden"], cfg["fake"], cfg["dropout"], cfg["blocks"], cfg["heads"], dev)
    else:
        model = srfrd_amd.SRFU_B(cfg["n_items"], cfg["seq_len"], cfg["hidden"], cfg["labels"], cfg["dropout"], cfg["blocks"], cfg["heads"], dev)
    for _, p in model.named_parameters():     # reference trainer.py:364-369
        if p.dim() >= 2:
            torch.nn.init.xavier_normal_(p.data)
    return model.to(dev).train()


def other_workloads(dev):
    """The other BASELINE configs, measured in the same run on the same GPU (N = 1 only; never the contract value): ms per
    512-sequence train step, ms per 512-user exact top-10 over the catalog (forward included)."""
    import argparse
    out = {}
    plan = [("C3", False, 60), ("C4", False, 60), ("C5", False, 30), ("C2", True, 100), ("C5", True, 40), ("C5_bf16_table", True, 40)]
    for name, predict, steps in plan:
        key = f"{name}:{'top10' if predict else 'train'}"
        try:
            a = argparse.Namespace(predict=predict, autograd=False, no_graph=False, deterministic=False, steps=steps, warmup=15,
                                   workload=name)
            model = build_model(WORKLOADS[name], dev)
            out[key] = round(explore(a, WORKLOADS[name], model, dev, 0, quiet=True), 4)
            del model
            torch.cuda.empty_cache()
        except Exception as e:              # never lose the contract line to a side measurement
            out[key] = repr(e)
    return out


def explore(args, cfg, model, dev, rank, quiet=False):
    """Non-contract measurements of the other BASELINE configs (train step, or forward + full-catalog top-10)."""
    import srfrd_amd
    B, L = cfg["batch"], cfg["seq_len"]
    u, seq, rsq, pos, prs, neg, nrs = srfrd_amd.synthetic_batch(cfg["n_items"], L, B, seed=1, rank=rank, device=dev)
    if cfg.get("bf16_table"):
        # ranking with frozen weights (serving): the bf16 shadow is derived once, not on every call; training keeps it current itself
        model.use_bf16_table(auto_refresh=not args.predict)
    if args.predict:
        model.eval()
        fn = lambda: model.topk(u, seq, rsq, k=10)
    elif args.autograd:
        # the reference's own loop shape (trainer.py:29-41) on the drop-in modules: custom ops + torch autograd + torch Adam
        # srfrd_amd.Adam: torch.optim.Adam's update as one launch over the flat parameter vector (--torch-adam: torch's own)
        opt = (torch.optim.Adam if getattr(args, "torch_adam", False) else srfrd_amd.Adam)(model.parameters(), lr=1e-3, betas=(0.9, 0.98))
        crit = torch.nn.BCEWithLogitsLoss()

        def fn():
            h, pl, nl = model(u, seq, rsq, pos, prs, neg, nrs)
            opt.zero_grad()
            idx = torch.where(pos != 0)
            loss = crit(pl[idx], torch.ones_like(pl)[idx]) + crit(nl[idx], torch.zeros_like(nl)[idx])
            loss.backward()
            opt.step()
    else:
        # as on the contract line: the batch is resident in the trainer's input ring, the step consumes it in place
        tr = srfrd_amd.FusedTrainer(model, B, L, use_graph=not args.no_graph, slots=2, deterministic=args.deterministic)
        packed = srfrd_amd.synthetic_batch(cfg["n_items"], L, B, seed=1, rank=rank, device=dev, packed=True)[1]
        for k in range(2):
            tr.ids_ring[k].copy_(packed)
        cnt = {"i": 0}

        def fn():
            tr.step_slot(cnt["i"] & 1)
            cnt["i"] += 1
    for _ in range(args.warmup):
        fn()
    torch.cuda.synchronize()
    if getattr(args, "profile_host", False):
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=40), file=sys.stderr)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fn()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if quiet:
        return el / args.steps * 1e3
    print(json.dumps({"workload": args.workload, "mode": "predict_top10" if args.predict else ("train_step_autograd_path" if args.autograd else "train_step"),
                      "optimizer": ("torch.optim.Adam" if getattr(args, "torch_adam", False) else "srfrd_amd.Adam") if args.autograd else "fused",
                      "kind": cfg["kind"], "sequences_per_s": B * args.steps / el, "ms_per_step": el / args.steps * 1e3,
                      "batch": B, "seq_len": L, "n_items": cfg["n_items"], "item_table": "bf16 shadow" if cfg.get("bf16_table") else "fp32",
                      "contract_line": False}), flush=True)


def time_kernels(tr, batches, steps):
    """Average duration (ms) of each launch, bracketed by events on the stream the launches go to.  The launches are eager
    here (one host call each), so an interval also holds whatever time the host took to issue the next launch when the GPU
    had run dry: a few untimed iterations first, and an iteration whose interval exceeds 3x the median of its kernel is a
    host stall, not a launch, and is left out of that kernel's average."""
    import ctypes as C
    from srfrd_amd import _lib
    from srfrd_amd._lib import check, ptr
    # (the C entry points the trainer's step calls, in call order; reported under the kernels' plain names)
    names = (["srfrd_seq_order"] if tr.sched_mode else []) + ["srfrd_encoder_fwd_sched", "srfrd_encoder_bwd_sched", "srfrd_reduce_dense"]
    local_update = tr.mode != "sharded"        # (the sharded update needs the collectives around it: compute kernels only)
    if local_update:
        names.append("srfrd_adam_pack_step")
    if tr.mode == "allreduce":
        names.append("srfrd_loss_finalize")
    lib = _lib.lib()
    acc = {n: 0.0 for n in names}
    lead = 5
    steps += lead
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)] for _ in range(steps)]
    # re-use the trainer's own enqueue code, but with an event between launches: wrap the C entry points
    orig = {n: getattr(lib, n) for n in names}
    state = {"i": 0, "k": 0}

    def wrap(n, k):
        def f(*a):
            rc = orig[n](*a)
            ev[state["i"]][k + 1].record()
            return rc
        return f
    for k, n in enumerate(names):
        setattr(lib, n, wrap(n, k))
    try:
        torch.cuda.synchronize()
        for i in range(steps):
            state["i"] = i
            tr.ids.copy_(batches[i % 8], non_blocking=True)
            ev[i][0].record()
            tr._enqueue_compute()
            if local_update:
                tr._enqueue_update()      # rank-0-only pass: NO collective here (the other ranks are not in this loop)
        torch.cuda.synchronize()
    finally:
        for n in names:
            setattr(lib, n, orig[n])
    for k, n in enumerate(names):
        ts = sorted(ev[i][k].elapsed_time(ev[i][k + 1]) for i in range(lead, steps))
        ts = [t for t in ts if t <= 3.0 * ts[len(ts) // 2]]
        acc[n] = sum(ts) / len(ts)
    return {n.replace("_sched", ""): v for n, v in acc.items()}


if __name__ == "__main__":
    main()
