#!/usr/bin/env python3
"""Latency of the fused backward (+ slab reduction) against the number of sequences in flight (diagnostic; GPU box).

Companion of tools/fwd_latency_vs_batch.py: at the C2 geometry one sequence's backward is a ~64 us dependency chain; up to
256 sequences (one per CU) a launch takes that long whatever the batch, and BASELINE's batch of 512 is two rounds of it
(the backward keeps one workgroup per CU).  DESIGN.md section 9 records what running the two rounds concurrently would
buy and why the register budget eats it.
"""
import os, sys, torch
sys.path.insert(0, "/root/repo")
import srfrd_amd
dev = "cuda"
torch.manual_seed(0)
m = srfrd_amd.SASRec(50000, 50, 50, 0.5, 2, 1, dev)
for _, p in m.named_parameters():
    if p.dim() >= 2:
        torch.nn.init.xavier_normal_(p.data)
m = m.to(dev).train()
for B in (128, 256, 512, 1024):
    tr = srfrd_amd.FusedTrainer(m, B, 50, use_graph=False)
    tr.refresh()
    tr.ids.copy_(srfrd_amd.synthetic_batch(50000, 50, B, seed=1, device=dev, packed=True)[1])
    tr._enqueue_fwd()
    def f(): tr._enqueue_bwd()
    for _ in range(10): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): f()
    e1.record(); torch.cuda.synchronize()
    print(f"B = {B:5d}: bwd + reduce_dense {e0.elapsed_time(e1) / 100 * 1000:7.1f} us")
