"""CPU-only, world_size = 2 (and 3) over gloo: the data-parallel exchange of srfrd_amd/exchange.py - the code
FusedTrainer(world > 1) runs between its kernels - in both forms:

  sharded    16-byte all-reduce of the loss statistics -> reduce-scatter of the flat gradient -> Adam on the own 1/N slice
             (moments for that slice only) -> all-gather of the stepped parameters;
  allreduce  one all-reduce of [gradient | statistics] -> the full Adam on every rank.

Each rank produces SUM gradients and (loss sums, target count) for its part of the batch; Adam divides by the GLOBAL
count, so N ranks must reproduce the single-process step of reference trainer.py:36-41 (a mean over all non-pad targets,
not a mean of per-rank means) - checked against the reference's own post-step weights (golden `w1/`).
The per-rank gradients and the Adam arithmetic come from the CPU oracle here (the HIP kernels need a GPU: the same
comparison with the real kernels is tools/dp_parity.py, run by tests/test_gpu_dp.py); shard bounds, padding, collectives,
the gloo fall-backs and the biased shard views are the product's.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.helpers import golden_cfg, load_golden


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, kind, mode, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from oracle import srfrd_oracle as O
    from srfrd_amd.exchange import GradExchange
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
    sizes = [sd[k].numel() for k in names]
    n_flat = sum(sizes)
    gflat = torch.cat([(leaves[k].grad if leaves[k].grad is not None else torch.zeros_like(leaves[k])).reshape(-1) for k in names])
    off = names.index(O.key_item(cfg))
    gflat[sum(sizes[:off]):sum(sizes[:off]) + sd[O.key_item(cfg)].shape[1]] = 0          # padding_idx row
    stats = torch.stack([sp.detach(), sn.detach(), n.float(), torch.zeros(())])
    pflat = torch.cat([sd[k].reshape(-1) for k in names])
    ex = GradExchange(n_flat)                              # ---- product code under test from here
    assert ex.world == world and ex.rank == rank and ex.n_pad >= n_flat and ex.per % 4 == 0

    def adam(p, gr, cnt):                                  # oracle Adam, first step, on a flat slice
        m = 0.1 * (gr / cnt)
        v = 0.02 * (gr / cnt) ** 2
        bc1, bc2s = 1.0 - 0.9, (1.0 - 0.98) ** 0.5
        return p - (1e-3 / bc1) * m / (v.sqrt() / bc2s + 1e-8)

    if mode == "allreduce":
        vec = torch.cat([gflat, stats])
        ex.all_reduce(vec)
        stats = vec[-4:]
        pflat = adam(pflat, vec[:n_flat], stats[2])
    else:
        work = ex.all_reduce_stats(stats)
        gpad = torch.zeros(ex.n_pad)
        gpad[:n_flat] = gflat
        recv = torch.empty(ex.per)
        ex.reduce_scatter(gpad, recv)
        work.wait()
        ppad = torch.zeros(ex.n_pad)
        ppad[:n_flat] = pflat
        ppad[ex.i0:ex.i1] = adam(ppad[ex.i0:ex.i1], recv, stats[2])
        ppad[:ex.i0] = float("nan")                        # only the own slice may survive the all-gather
        ppad[ex.i1:] = float("nan")
        ex.all_gather(ppad)
        pflat = ppad[:n_flat]
        assert bool((ppad[n_flat:] == 0).all())            # the padding steps from 0 with gradient 0: stays 0
    loss = float(stats[0] / stats[2] + stats[1] / stats[2])
    o, res = 0, {}
    for k, m in zip(names, sizes):
        res[k] = pflat[o:o + m].view_as(sd[k]).numpy()
        o += m
    np.savez(out_path + f".{rank}.npz", loss=loss, **res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,world", [("sharded", 2), ("allreduce", 2), ("sharded", 3)])
def test_n_rank_exchange_equals_single_process(tmp_path, mode, world):
    kind = "SRFRN"
    out = str(tmp_path / "dp")
    mp.spawn(_rank_main, args=(world, _free_port(), kind, mode, out), nprocs=world, join=True)
    g, sd, batch = load_golden(kind)
    from tests.helpers import drop_kbias, sub
    w1 = sub(g, "w1/")
    zs = [np.load(out + f".{r}.npz") for r in range(world)]
    for z in zs:
        assert abs(float(z["loss"]) - float(g["loss0"])) < 1e-5
        for k in w1:
            a = drop_kbias(k, torch.from_numpy(z[k]), 50)
            b = drop_kbias(k, w1[k], 50)
            assert float((a - b).abs().max()) < 2e-5, k
    for k in w1:                                           # replicas stay bit-identical
        assert all((zs[0][k] == z[k]).all() for z in zs[1:]), k


def test_single_process_exchange_is_identity():
    from srfrd_amd.exchange import GradExchange
    from srfrd_amd.trainer import flat_allreduce
    x = torch.arange(5.0)
    assert torch.equal(flat_allreduce(x.clone()), x)
    ex = GradExchange(10)
    assert (ex.world, ex.rank, ex.per, ex.n_pad, ex.i0, ex.i1) == (1, 0, 12, 12, 0, 12)
    gsrc, out = torch.arange(12.0), torch.empty(12)
    assert torch.equal(ex.reduce_scatter(gsrc, out), gsrc) and ex.all_reduce_stats(torch.zeros(4)) is None
    assert torch.equal(ex.all_gather(gsrc.clone()), gsrc)
