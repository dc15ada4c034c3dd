#!/usr/bin/env python3
"""Latency of the fused forward against the number of sequences in flight (diagnostic; run on the GPU box).

Times srfrd_encoder_fwd at the C2 geometry (SASRec, 50 k items, seq_len 50) in its three modes and, in inference mode, for
batches of 64 .. 512 sequences.  The point it makes: up to 256 sequences (one workgroup per CU) the launch takes the same
~34 us - the time of ONE sequence's dependency chain (2 blocks x ~10 dependent GEMM / row-op stages) - and 512 sequences
(two workgroups per CU) take 1.45 x that.  At BASELINE's batch of 512 the chip holds two sequences per CU: throughput is
set by the length of that chain, not by matrix-pipe or HBM throughput.
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
m = m.to(dev)
_, seq, rsq, pos, prs, neg, nrs = srfrd_amd.synthetic_batch(50000, 50, 512, seed=1, device=dev)
ids = m._prep(seq, None, pos, None, neg, None)
def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000
m.pack_weights()
import ctypes as C
from srfrd_amd import _lib
from srfrd_amd._lib import ptr, check
lay, flat = m.layout, m._flat
B, L = 512, 50
hidden = torch.empty(B, L, 50, device=dev); pl = torch.empty(B, L, device=dev); nl = torch.empty(B, L, device=dev)
sx = torch.empty(3, B, L, 50, device=dev); sh = torch.empty(2, B, L, 50, device=dev)
sa = torch.empty(_lib.lib().srfrd_aux_floats(C.byref(lay), B, L), device=dev); lp = torch.empty(B, 3, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def launch(save, p):
    check(_lib.lib().srfrd_encoder_fwd(C.byref(lay), ptr(flat), C.c_void_p(flat.data_ptr() + 4 * m.n_table_pad), ptr(m._packed),
        ptr(ids[0]), None, ptr(ids[2]), None, ptr(ids[4]), None, B, L, p, 7, None, 0, ptr(hidden), ptr(pl), ptr(nl),
        ptr(sx) if save else None, ptr(sh) if save else None, ptr(sa) if save else None, ptr(lp) if save else None, None, 0, None, 0, st), "fwd")
for name, save, p in (("train (checkpoints, dropout 0.5)", True, 0.5), ("checkpoints, no dropout", True, 0.0), ("inference (no checkpoints)", False, 0.0)):
    print(f"{name:36s} {t(lambda: launch(save, p)):7.1f} us")
for Bx in (64, 128, 256, 384, 512):
    B = Bx
    print(f"inference, B = {Bx}: {t(lambda: launch(False, 0.0)):7.1f} us")
