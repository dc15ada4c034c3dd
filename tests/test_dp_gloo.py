"""CPU-only, world_size = 2 over gloo: the data-parallel reduction contract of srfrd_amd.trainer.

Each rank produces SUM gradients and (loss sums, target count) for its half of the batch; one all-reduce over the flat
[grads | stats] vector makes them global and Adam divides by the GLOBAL count - so two ranks must reproduce the
single-process step of reference trainer.py:36-41 (a mean over all non-pad targets, not a mean of per-rank means).
The per-rank gradients come from the CPU oracle here (the HIP kernels need a GPU); the reduction code is the product's.
"""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.helpers import golden_cfg, load_golden


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, kind, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from oracle import srfrd_oracle as O
    from srfrd_amd.trainer import flat_allreduce
    g, sd, batch = load_golden(kind)
    cfg = golden_cfg(kind)
    B = batch[0].shape[0]
    lo, hi = rank * B // world, (rank + 1) * B // world
    shard = tuple(t[lo:hi] for t in batch)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    h, pl, nl = O.forward(cfg, leaves, *shard)
    sp, sn, n = O.bce_sums(pl, nl, shard[2])
    (sp + sn).backward()                                  # SUM reduction, as the backward kernel produces
    names = list(sd.keys())
    flat = torch.cat([(leaves[k].grad if leaves[k].grad is not None else torch.zeros_like(leaves[k])).reshape(-1)
                      for k in names] + [torch.stack([sp.detach(), sn.detach(), n.float(), torch.zeros(())])])
    flat_allreduce(flat)                                  # product code under test
    stats = flat[-4:]
    off, grads = 0, {}
    for k in names:
        m = sd[k].numel()
        grads[k] = (flat[off:off + m] / stats[2]).view_as(sd[k])
        off += m
    grads[O.key_item(cfg)][0].zero_()
    opt = O.Adam(sd)
    opt.step(sd, grads)
    loss = float(stats[0] / stats[2] + stats[1] / stats[2])
    if rank == 0:
        np.savez(out_path, loss=loss, **{k: v.numpy() for k, v in sd.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_reduction_equals_single_process(tmp_path):
    kind = "SRFRN"
    out = str(tmp_path / "dp.npz")
    mp.spawn(_rank_main, args=(2, _free_port(), kind, out), nprocs=2, join=True)
    z = np.load(out)
    g, sd, batch = load_golden(kind)
    assert abs(float(z["loss"]) - float(g["loss0"])) < 1e-5
    from tests.helpers import drop_kbias, sub
    w1 = sub(g, "w1/")
    for k in w1:
        a = drop_kbias(k, torch.from_numpy(z[k]), 50)
        b = drop_kbias(k, w1[k], 50)
        assert float((a - b).abs().max()) < 2e-5, k


def test_single_process_allreduce_is_identity():
    from srfrd_amd.trainer import flat_allreduce
    x = torch.arange(5.0)
    assert torch.equal(flat_allreduce(x.clone()), x)
