"""Fused train step: the MI355X counterpart of the inner loop of reference trainer.py:27-41.

    forward -> masked BCE(pos, 1) + BCE(neg, 0) over pos != 0 -> backward -> Adam(lr, betas=(0.9, 0.98))

as five stream-ordered launches on persistent buffers (encoder_fwd, encoder_bwd, reduce_dense [+ loss], adam_step,
pack_weights [+ optimizer-state advance for the next step]), captured into one HIP graph when no collective sits in the middle.  Differences from the reference
loop, all behaviour-preserving: the loss is never synchronised to the host (``loss`` stays a device scalar), the
``l2_emb * ||p||`` terms (trainer.py:39: one Frobenius norm per parameter tensor) cost two small launches and one pass over
the gradient when l2_emb != 0 and nothing at the reference's 0.0, and dropout masks come from the coordinate hash of csrc/srfrd_rng.h instead of torch's Bernoulli stream.

Data parallel (no reference counterpart): one process per GPU, the global batch split by sequence,
parameters replicated, dropout masks keyed by the GLOBAL sequence index.  srfrd_amd/exchange.py holds the two forms of
the per-step exchange; both divide by the GLOBAL count of non-pad targets, so N ranks reproduce the single-process
mean-over-all-targets loss of trainer.py:36-38 (not an average of per-rank means):
  "sharded" (default)  fwd -> [16-byte all-reduce of the loss statistics, overlapped with] bwd -> reduce-scatter of the
                       flat gradient -> Adam on this rank's 1/N slice (moments exist for that slice only) -> all-gather
                       of the stepped parameters -> re-pack of the encoder weights;
  "allreduce"          fwd -> bwd -> one all-reduce of [gradient | statistics] -> the full fused Adam tail on every rank.
"""
from __future__ import annotations

import ctypes as C
import os

import torch
import torch.distributed as dist

from . import _lib
from ._lib import check, ptr
from .exchange import GradExchange, shard_bounds  # noqa: F401  (shard_bounds re-exported)


def flat_allreduce(flat: torch.Tensor, group=None):
    """SUM all-reduce of the flat gradient vector (RCCL on GPUs, gloo on CPU tests)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


class FusedTrainer:
    """Owns optimizer state and step buffers for one model on one GPU (one rank of a DP job)."""

    def __init__(self, model, batch_size: int, seq_len: int | None = None, lr: float = 1e-3, betas=(0.9, 0.98),
                 eps: float = 1e-8, l2_emb: float = 0.0, seed: int = 42, process_group=None, use_graph: bool = True,
                 slots: int = 1, exchange: str = "sharded", deterministic: bool = False, shadow_gather: bool = False):
        """``shadow_gather`` (sharded exchange over a bf16 item-table shadow, ``model.use_bf16_table()``; SURVEY 8e's alternative
        for BASELINE configs[4]): the fp32 master of an item row and its Adam moments live on the OWNER rank only; after the
        sharded Adam step the all-gather carries the bf16 SHADOW of the table (2 bytes per element: 90 MB instead of 180 MB at
        1 M items) and a small all-reduce the dense parameters.  Every gather of the step reads the shadow, so the arithmetic
        is that of the bf16-table mode; a rank's fp32 rows OUTSIDE its shard go stale - call ``sync_master()`` (one fp32
        all-gather) before reading ``model.state_dict()`` / saving a checkpoint.
        ``deterministic``: the item-table gradient is scattered by a stable sort + per-item ordered sums instead of float
        atomics - every step bitwise reproducible (the dense gradients already are: fixed slab tree), at the cost of one
        sort of 3 B L keys per step."""
        self.model = model
        model._ensure_flat()
        self.lay = model.layout
        self.flat = model._flat
        dev = self.flat.device
        self.B, self.L = int(batch_size), int(seq_len or self.lay.max_len)
        n_f, n_b = _lib.scratch_floats(self.lay, self.B, self.L)      # 0 when the working set fits LDS
        self.scratch = torch.empty(max(n_f, n_b), device=dev, dtype=torch.float32) if max(n_f, n_b) else None
        self.n_scratch = max(n_f, n_b)
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.group = process_group
        self.n_tab, self.n_flat = model.n_table_pad, model.n_flat
        self.ex = GradExchange(self.n_flat, process_group)
        self.world, self.rank = self.ex.world, self.ex.rank
        if exchange not in ("sharded", "allreduce"):
            raise ValueError("exchange must be 'sharded' or 'allreduce'")
        self.mode = exchange if self.ex.on else "single"      # (ex.on: world > 1, or a forced exchange in a group of one)
        self.shadow_gather = bool(shadow_gather) and self.mode == "sharded"
        if shadow_gather and self.mode == "sharded" and not model.bf16_table:
            raise ValueError("shadow_gather exchanges the bf16 item-table shadow: call model.use_bf16_table() first")
        lay, B, L = self.lay, self.B, self.L
        f32 = dict(device=dev, dtype=torch.float32)
        if self.mode == "sharded":
            # gradient padded to world equal shards; Adam moments and the reduce-scatter landing buffer for the own shard only
            if self.ex.n_pad > model._flat_store.numel():
                raise RuntimeError("flat parameter storage has too little slack for this world size")
            self.flat_pad = model._flat_store[:self.ex.n_pad]
            self.grad = torch.zeros(self.ex.n_pad, **f32)
            self.stats = torch.zeros(4, **f32)
            self.recv = torch.zeros(self.ex.per, **f32)
            self.m = torch.zeros(self.ex.per, **f32)
            self.v = torch.zeros(self.ex.per, **f32)
            if self.shadow_gather:
                # the shadow padded to the shard grid (the all-gather walks the flat index space in 2-byte elements); the
                # model's kernels keep reading its first n_table elements
                self.shadow_pad = torch.zeros(self.ex.n_pad, device=dev, dtype=torch.int16)
                self.shadow_pad[:self.lay.n_table].copy_(model._table16)
                model._table16 = self.shadow_pad[:self.lay.n_table]
                model._bf16_auto = False                     # (the trainer keeps the shadow current; the fp32 table is partly stale)
                self.n_dense_x = self.n_flat - self.n_tab    # dense parameters (+ tail padding): exchanged in fp32
                self.dense_x = torch.zeros(self.n_dense_x, **f32)
        else:
            self.grad = torch.zeros(self.n_flat + 4, **f32)      # [table | dense | stats(4)]: one vector, one all-reduce
            self.stats = self.grad[self.n_flat:]
            self.m = torch.zeros(self.n_flat, **f32)
            self.v = torch.zeros(self.n_flat, **f32)
        # l2_emb * sum_p ||p|| (trainer.py:39): one segment per parameter tensor of the flat vector (the item table first)
        self.l2 = float(l2_emb)
        if self.l2 != 0.0:
            segs = [(off, p.numel()) for p, off in model._slots]
            assert segs[0][0] == 0 and all(off >= self.n_tab for off, _ in segs[1:])
            self.seg_off = torch.tensor([o for o, _ in segs], device=dev, dtype=torch.int64)
            self.seg_len = torch.tensor([n for _, n in segs], device=dev, dtype=torch.int64)
            self.l2_partial = torch.zeros(240 + len(segs), **f32)
            self.l2buf = torch.zeros(4, **f32)
            self.l2_dense = torch.zeros(self.n_flat - self.n_tab, **f32)
        self.state = torch.zeros(32, device=dev, dtype=torch.int32)
        self.state[1] = int(seed) & 0x7FFFFFFF
        self.deterministic = bool(deterministic)
        self.contrib = torch.zeros(3, B, L, lay.d_item, **f32) if self.deterministic else None
        self.keys = torch.zeros(3, B, L, device=dev, dtype=torch.int64) if self.deterministic else None
        self.n_slabs = _lib.lib().srfrd_bwd_grid(C.byref(lay), B, L)
        self.slabs = torch.empty(self.n_slabs, lay.n_dense, **f32)
        # Input ring: `slots` resident (6, B, L) id buffers.  A producer (DeviceSampler, a loader thread's H2D copy)
        # fills slot k + 1 while step k runs and calls step_slot(k + 1): no staging copy on the step's stream.  Slot 0
        # is also the landing buffer of step() / step_packed(), which copy their arguments in.
        self.slots = max(1, int(slots))
        self.ids_ring = torch.zeros(self.slots, 6, B, L, device=dev, dtype=torch.int64)
        self.ids = self.ids_ring[0]
        self.hidden = torch.empty(B, L, lay.d_out, **f32)
        self.pl = torch.empty(B, L, **f32)
        self.nl = torch.empty(B, L, **f32)
        # (zeros, not empty: the ragged kernels write and read only the rows of a sequence's computed tiles; rows nobody
        # wrote must never hold a NaN pattern if a full-row kernel is switched in between two steps)
        self.save_x = torch.zeros(B, lay.n_blocks + 1, L, lay.D, **f32)
        self.save_h1 = torch.zeros(B, lay.n_blocks, L, lay.D, **f32)
        self.save_aux = torch.zeros(_lib.lib().srfrd_aux_floats(C.byref(lay), B, L), **f32)
        self.loss_part = torch.empty(B, 3, **f32)
        self.loss = torch.zeros(1, **f32)
        # Sequence -> workgroup schedule of the seq_len-50 ragged kernels (include/srfrd_hip.h, srfrd_seq_order): one small
        # launch per step ranks the batch by length, forward and backward pair a long with a short sequence on every CU.
        # SRFRD_SCHED = 0 (off: workgroup x takes sequence x) | 1 (length order; default)
        self.sched_mode = min(1, int(os.environ.get("SRFRD_SCHED", "1"))) if (L == 50 and lay.D == 50 and lay.n_heads == 1) else 0
        self.sched = torch.zeros(int(_lib.lib().srfrd_sched_ints(B)), device=dev, dtype=torch.int32) if self.sched_mode else None
        self.pair_stride = max(1, _lib.lib().srfrd_bwd_grid(C.byref(lay), 1 << 30, L) // 2)      # CUs: workgroups of the first round
        self.packed = model.pack_weights()
        check(_lib.lib().srfrd_step_begin(ptr(self.state), self.lr, self.betas[0], self.betas[1],
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream)), "srfrd_step_begin")
        self.use_graph = bool(use_graph)
        self._graph_a = self._graph_b = self._graph_f = self._graph_u = self._graph_one = None
        self.graph_form = None           # "one" | "split" once captured (data parallel: one graph with the collectives inside, or graphs around them)
        self.graph_capture_error = None
        self.steps_done = 0
        self.err = torch.zeros(1, device=dev, dtype=torch.int32)   # srfrd_check_ids word of step() / step_packed() inputs
        self._fresh = False      # packed weights known to match the parameters (see refresh())

    # ---- pieces -------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _dense_ptr(self, base):
        return C.c_void_p(base.data_ptr() + 4 * self.n_tab)

    def _ids_of(self, slot):
        ids = self.ids_ring[slot]
        fk = ids[1] if self.lay.kind != 0 else None
        pfk, nfk = (ids[3], ids[5]) if self.lay.kind == 2 else (None, None)
        p = self.model.dropout_rate if self.model.training else 0.0
        return ids, fk, pfk, nfk, p, C.c_void_p(self.state.data_ptr() + 8), self.rank * self.B

    def _enqueue_l2_apply(self, grad_ptr, param, i0, i1):
        """grad[i0:i1] += count * l2_emb * p / ||p|| (the Adam step that follows divides by the count)"""
        check(_lib.lib().srfrd_l2_apply(grad_ptr, ptr(param), i0, min(i1, self.n_flat), self.n_tab, ptr(self.l2buf),
                                        ptr(self.l2_dense), ptr(self.stats), self._stream()), "srfrd_l2_apply")

    def _enqueue_fwd(self, slot: int = 0):
        L_, lay, st = _lib.lib(), self.lay, self._stream()
        if self.l2 != 0.0:               # norms of the parameters this step's forward uses
            check(L_.srfrd_l2_norms(ptr(self.flat), ptr(self.seg_off), ptr(self.seg_len), self.seg_off.numel(), self.n_tab,
                                    self.l2, ptr(self.l2_partial), ptr(self.l2buf), ptr(self.l2_dense), st), "srfrd_l2_norms")
        ids, fk, pfk, nfk, p, seed_dev, seq0 = self._ids_of(slot)
        lay_t, tab = self.model._table_args()
        if self.sched_mode:
            check(L_.srfrd_seq_order(ptr(ids[0]), self.B, self.L, self.pair_stride, ptr(self.sched), st), "srfrd_seq_order")
        check(L_.srfrd_encoder_fwd_sched(C.byref(lay_t), tab, self._dense_ptr(self.flat), ptr(self.packed), ptr(ids[0]), ptr(fk), ptr(ids[2]),
                                         ptr(pfk), ptr(ids[4]), ptr(nfk), self.B, self.L, p, 0, seed_dev, seq0, ptr(self.hidden),
                                         ptr(self.pl), ptr(self.nl), ptr(self.save_x), ptr(self.save_h1), ptr(self.save_aux), ptr(self.loss_part),
                                         ptr(self.scratch), self.n_scratch, ptr(self.sched), self.sched_mode, st), "srfrd_encoder_fwd_sched")
        if self.mode == "sharded":       # the statistics are forward outputs: reduce them now, exchange them under the backward
            check(L_.srfrd_loss_stats(ptr(self.loss_part), self.B, ptr(self.stats), None, st), "srfrd_loss_stats")

    def _enqueue_bwd(self, slot: int = 0):
        L_, lay, st = _lib.lib(), self.lay, self._stream()
        ids, fk, pfk, nfk, p, seed_dev, seq0 = self._ids_of(slot)
        lay_t, tab = self.model._table_args()
        check(L_.srfrd_encoder_bwd_sched(C.byref(lay_t), tab, self._dense_ptr(self.flat), ptr(self.packed), ptr(ids[0]), ptr(fk), ptr(ids[2]),
                                         ptr(pfk), ptr(ids[4]), ptr(nfk), self.B, self.L, p, 0, seed_dev, seq0, ptr(self.hidden),
                                         ptr(self.pl), ptr(self.nl), ptr(self.save_x), ptr(self.save_h1), ptr(self.save_aux), None, None, None, 1,
                                         ptr(self.grad), ptr(self.contrib), ptr(self.slabs), ptr(self.scratch), self.n_scratch,
                                         ptr(self.sched), self.sched_mode, st), "srfrd_encoder_bwd_sched")
        if self.contrib is not None:
            # deterministic item-table scatter: stable sort of the 3 B L row keys (pos, neg, input ids - the row order of
            # `contrib`), then one wave per item adds its rows in that order
            torch.stack((ids[2], ids[4], ids[0]), out=self.keys)
            skeys, order = torch.sort(self.keys.view(-1), stable=True)
            check(L_.srfrd_table_reduce(ptr(skeys), ptr(order), ptr(self.contrib), skeys.numel(), lay.d_item, ptr(self.grad), st),
                  "srfrd_table_reduce")
        # single rank: the slab reduction also finalises the loss; all-reduce form: it leaves the local statistics behind the
        # gradient (one vector, one collective); sharded form: the statistics were reduced right after the forward
        fuse_stats = self.mode != "sharded"
        check(L_.srfrd_reduce_dense(ptr(self.slabs), self.n_slabs, lay.n_dense, self._dense_ptr(self.grad),
                                    ptr(self.loss_part) if fuse_stats else None, self.B, ptr(self.stats) if fuse_stats else None,
                                    ptr(self.loss) if self.mode == "single" else None, st), "srfrd_reduce_dense")

    def _enqueue_compute(self, slot: int = 0):
        self._enqueue_fwd(slot)
        self._enqueue_bwd(slot)

    def _enqueue_update(self):
        """single rank / all-reduce form: Adam over the whole flat vector + re-pack + optimizer-state advance, one launch"""
        L_, st = _lib.lib(), self._stream()
        if self.l2 != 0.0:
            self._enqueue_l2_apply(ptr(self.grad), self.flat, 0, self.n_flat)
        check(L_.srfrd_adam_pack_step(C.byref(self.lay), ptr(self.flat), ptr(self.grad), ptr(self.m), ptr(self.v),
                                      self.n_flat, self.n_tab, self.n_tab, self.lr, self.betas[0], self.betas[1], self.eps,
                                      ptr(self.state), ptr(self.stats), ptr(self.packed), ptr(self.model._table16), st),
              "srfrd_adam_pack_step")
        if self.mode != "single":
            check(L_.srfrd_loss_finalize(ptr(self.stats), ptr(self.loss), st), "srfrd_loss_finalize")
        if self.l2 != 0.0:
            self.loss.add_(self.l2buf[1:2])

    def _enqueue_shard_update(self):
        """sharded form, between the reduce-scatter and the all-gather: re-zero the local item-table gradient (the atomics
        of the next backward accumulate into it) and step this rank's slice; `recv`, `m`, `v` hold that slice only, so
        their pointers are biased by -i0 to be indexed with the global element index."""
        L_, st, ex = _lib.lib(), self._stream(), self.ex
        self.grad[:self.n_tab].zero_()
        bias = 4 * ex.i0
        if self.l2 != 0.0:
            self._enqueue_l2_apply(C.c_void_p(self.recv.data_ptr() - bias), self.flat_pad, ex.i0, ex.i1)
        sg = self.shadow_gather      # (the owner writes the bf16 shadow of the table elements it steps)
        check(L_.srfrd_adam_step(ptr(self.flat_pad), C.c_void_p(self.recv.data_ptr() - bias), C.c_void_p(self.m.data_ptr() - bias),
                                 C.c_void_p(self.v.data_ptr() - bias), ex.n_pad, ex.i0, ex.i1, 0, self.betas[0], self.betas[1],
                                 self.eps, ptr(self.state), ptr(self.stats), ptr(self.shadow_pad) if sg else None,
                                 self.lay.n_table if sg else 0, st), "srfrd_adam_step")
        if sg:
            # this rank's part of the dense parameters into the exchange buffer (zeros elsewhere: the SUM all-reduce is a gather)
            self.dense_x.zero_()
            lo, hi = max(ex.i0, self.n_tab), min(ex.i1, self.n_flat)
            if hi > lo:
                self.dense_x[lo - self.n_tab:hi - self.n_tab].copy_(self.flat_pad[lo:hi])

    def _gather_params(self):
        """sharded form: every rank's stepped slice to all ranks - the fp32 vector, or (shadow_gather) the bf16 shadow of the
        table + the dense parameters"""
        if not self.shadow_gather:
            self.ex.all_gather(self.flat_pad)
            return
        self.ex.all_gather(self.shadow_pad.view(torch.float16))      # (2-byte elements; RCCL has no int16: the bits travel as fp16)
        self.ex.all_reduce(self.dense_x)
        self.flat[self.n_tab:self.n_flat].copy_(self.dense_x)

    def sync_master(self):
        """shadow_gather: bring every rank's fp32 parameters up to date (one fp32 all-gather; a collective - every rank calls
        it).  No-op otherwise."""
        if self.shadow_gather:
            self.ex.all_gather(self.flat_pad)

    def _enqueue_shard_finish(self):
        """sharded form, after the all-gather: fragment-ordered copy of the stepped weights + optimizer-state advance + loss"""
        L_, st = _lib.lib(), self._stream()
        check(L_.srfrd_pack_weights(C.byref(self.lay), self._dense_ptr(self.flat), ptr(self.packed), ptr(self.state), self.lr,
                                    self.betas[0], self.betas[1], st), "srfrd_pack_weights")
        if self.model._table16 is not None and not self.shadow_gather:      # bf16 shadow of the all-gathered item table (every rank needs all rows)
            check(L_.srfrd_table_to_bf16(ptr(self.flat), self.lay.n_table, ptr(self.model._table16), st), "srfrd_table_to_bf16")
        check(L_.srfrd_loss_finalize(ptr(self.stats), ptr(self.loss), st), "srfrd_loss_finalize")
        if self.l2 != 0.0:
            self.loss.add_(self.l2buf[1:2])

    def _enqueue_dp_step(self, slot: int = 0):
        """the whole data-parallel step on the current stream, collectives included (eager, or under capture with RCCL)"""
        if self.mode == "allreduce":
            self._enqueue_compute(slot)
            self.ex.all_reduce(self.grad)
            self._enqueue_update()
        else:
            self._enqueue_fwd(slot)
            h = self.ex.all_reduce_stats(self.stats)               # 16 bytes, in flight under the backward
            self._enqueue_bwd(slot)
            self.ex.reduce_scatter(self.grad, self.recv)
            if h is not None:
                h.wait()
            self._enqueue_shard_update()
            self._gather_params()
            self._enqueue_shard_finish()

    def _capture(self):
        # warm-up on a side stream (sets the LDS attributes, loads code objects), then capture
        torch.cuda.synchronize()
        keep = [self.flat, self.m, self.v, self.state, self.grad, self.stats] + ([self.shadow_pad] if self.shadow_gather else [])
        snap = [t.clone() for t in keep]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            if self.mode != "single" and self.ex.native:
                # RCCL: one whole step EAGERLY, collectives included, before anything is captured - every collective type the
                # graph will hold has then run once on this communicator (lazy channel / buffer set-up cannot be captured)
                self._enqueue_dp_step()
            else:
                self._enqueue_compute()
                if self.mode == "sharded":
                    self._enqueue_shard_update()
                    self._enqueue_shard_finish()
                else:
                    self._enqueue_update()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()

        def restore():
            for dst, src in zip(keep, snap):
                dst.copy_(src)
            self.model.pack_weights()         # the warm-up step re-packed the stepped weights: restore that too
            if not self.shadow_gather:        # ... and re-derived the bf16 shadow of the stepped table (if one is in use;
                self.model.refresh_bf16_table()   # shadow_gather: the shadow itself is part of the snapshot)

        restore()
        # thread_local capture mode: a collective backend's watchdog thread may touch the HIP runtime while we capture

        def graph_of(fn):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                fn()
            return g

        # one graph per input slot (the kernels' id pointers are baked in)
        self.graph_form = "one" if self.mode == "single" else "split"
        if self.mode != "single" and self.ex.native and os.environ.get("SRFRD_DP_SPLIT_GRAPHS", "") != "1":
            # RCCL collectives are stream work: the WHOLE data-parallel step - compute, collectives, sharded Adam, re-pack -
            # goes into ONE graph per slot, so a step is one host launch and nothing on the host sits between the kernels and
            # the collectives.  (ProcessGroupNCCL forks its internal stream off the capturing one and joins it back: the
            # statistics all-reduce stays concurrent with the backward inside the graph.)  If this torch / RCCL pair refuses
            # the capture, the step falls back to the split form below; `graph_form` says which one runs.
            try:
                self._graph_one = [graph_of(lambda k=k: self._enqueue_dp_step(k)) for k in range(self.slots)]
                self.graph_form = "one"
            except Exception as e:                      # noqa: BLE001 - any capture failure selects the split form
                self._graph_one = None
                self.graph_capture_error = repr(e)
                torch.cuda.synchronize()
                restore()
        if self.graph_form == "one" and self.mode != "single":
            pass
        elif self.mode == "single":
            self._graph_a = [graph_of(lambda k=k: (self._enqueue_compute(k), self._enqueue_update())) for k in range(self.slots)]
        elif self.mode == "allreduce":
            self._graph_a = [graph_of(lambda k=k: self._enqueue_compute(k)) for k in range(self.slots)]
            self._graph_b = graph_of(self._enqueue_update)
        else:
            self._graph_f = [graph_of(lambda k=k: self._enqueue_fwd(k)) for k in range(self.slots)]
            self._graph_a = [graph_of(lambda k=k: self._enqueue_bwd(k)) for k in range(self.slots)]
            self._graph_u = graph_of(self._enqueue_shard_update)
            self._graph_b = graph_of(self._enqueue_shard_finish)
        # The first replay of a graph also uploads it to the device (tens of microseconds, once per graph - and there is one
        # graph per input slot): replay each once here, inside the snapshot, so that no step pays for it.  (Compute graphs
        # only: no collective is involved, every rank does the same.)
        if self._graph_one is None:
            for gs in (self._graph_f, self._graph_a, [self._graph_u], [self._graph_b]):
                for g in (gs or []):
                    if g is not None:
                        g.replay()
        else:
            for g in self._graph_one:       # (holds collectives: every rank replays the same graphs in the same order)
                g.replay()
        torch.cuda.synchronize()
        restore()

    # ---- public -------------------------------------------------------------------------------
    def spin_up(self, replays: int = 200):
        """Bring the device to its steady state (clocks, caches, TLBs of the step's buffers) without training: replays the
        captured single-rank step `replays` times on whatever slot 0 holds, inside a snapshot - parameters, optimizer
        moments, step counter and dropout seed are restored afterwards, so the next step is the one it would have been.
        (The first hundred steps after start-up run ~4 % slower than the rest; a short measurement that wants the
        steady-state rate calls this first.  No-op without graphs.  In data parallel every rank replays the same graphs the same
        number of times: the one-graph form with its collectives inside, the split form's local graphs without the collectives
        between them; everything is restored.)"""
        if not self.use_graph:
            return
        if not self._fresh:
            self.refresh()
        if self._graph_a is None and self._graph_one is None:
            self._capture()
        torch.cuda.synchronize()
        keep = [self.flat, self.m, self.v, self.state, self.grad, self.stats] + ([self.shadow_pad] if self.shadow_gather else [])
        snap = [t.clone() for t in keep]
        if self._graph_one is not None:      # (collectives inside: every rank replays the same number of times)
            seq = [self._graph_one[0]]
        else:
            seq = [g[0] if isinstance(g, list) else g for g in (self._graph_f, self._graph_a, self._graph_u, self._graph_b)
                   if g is not None]
        for _ in range(int(replays)):
            for g in seq:
                g.replay()
        torch.cuda.synchronize()
        for dst, src in zip(keep, snap):
            dst.copy_(src)
        self.model.pack_weights()
        if not self.shadow_gather:
            self.model.refresh_bf16_table()
        torch.cuda.synchronize()

    def refresh(self):
        """Re-derive everything the step keeps derived from the parameters (the MFMA-fragment-ordered weight copy).  Call
        after modifying parameters from outside the trainer - ``load_state_dict``, a re-initialisation, a manual edit;
        the trainer's own Adam tail keeps the copy current by itself.  Done automatically before the first step."""
        self.packed = self.model.pack_weights()
        self._fresh = True

    # ---- optimizer state in torch.optim.Adam's own format ------------------------------------------------------------
    def _moments_full(self):
        """(exp_avg, exp_avg_sq) over the whole flat vector; in the sharded form every rank holds a slice: all-gathered here
        (a collective - every rank must make the call)."""
        if self.mode != "sharded":
            return self.m, self.v
        out = []
        for part in (self.m, self.v):
            full = torch.zeros(self.ex.n_pad, device=part.device, dtype=torch.float32)
            full[self.ex.i0:self.ex.i1] = part
            out.append(self.ex.all_gather(full))
        return out[0], out[1]

    def state_dict(self):
        """The optimizer state as ``torch.optim.Adam(model.parameters(), lr, betas, eps).state_dict()`` would hold it after
        the same steps (per parameter ``step``, ``exp_avg``, ``exp_avg_sq``; one param group) - loadable by either side -
        plus ``"srfrd"``: the dropout base seed, so that a resumed run draws the masks the uninterrupted one would have.
        The reference saves the model only (trainer.py:409-411); this is what resuming TRAINING needs on top."""
        m, v = self._moments_full()
        params = list(self.model.parameters())
        off = {id(p): o for p, o in self.model._slots}
        state = {}
        for i, p in enumerate(params):
            o, n = off[id(p)], p.numel()
            state[i] = {"step": torch.tensor(float(self.steps_done)),
                        "exp_avg": m[o:o + n].view(p.shape).clone(), "exp_avg_sq": v[o:o + n].view(p.shape).clone()}
        group = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": 0, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "decoupled_weight_decay": False,
                 "params": list(range(len(params)))}
        return {"state": state, "param_groups": [group],
                "srfrd": {"seed": int(self.state[1].item()), "steps_done": int(self.steps_done)}}

    def load_state_dict(self, sd):
        """Inverse of state_dict(); also takes the state_dict of a torch.optim.Adam that stepped the same model's
        parameters (the module-level path), so a run can move between the two.  Hyper-parameters come from the dict."""
        params = list(self.model.parameters())
        off = {id(p): o for p, o in self.model._slots}
        group = sd["param_groups"][0]
        if any(group.get(k) for k in ("weight_decay", "amsgrad", "maximize")):
            raise ValueError("FusedTrainer implements plain Adam (no weight decay / amsgrad / maximize)")
        if list(group["params"]) != list(range(len(params))):
            raise ValueError("optimizer state does not cover this model's parameters in order")
        steps = {int(float(st["step"])) for st in sd["state"].values()}
        if len(steps) > 1:
            raise ValueError("parameters with different step counts")
        self.lr, self.betas, self.eps = float(group["lr"]), (float(group["betas"][0]), float(group["betas"][1])), float(group["eps"])
        dev = self.flat.device
        m = torch.zeros(self.ex.n_pad if self.mode == "sharded" else self.n_flat, device=dev, dtype=torch.float32)
        v = torch.zeros_like(m)
        for i, p in enumerate(params):
            st = sd["state"].get(i)
            if st is None:
                continue                                   # (a parameter that never received a gradient: zero moments)
            o, n = off[id(p)], p.numel()
            m[o:o + n] = st["exp_avg"].to(device=dev, dtype=torch.float32).reshape(-1)
            v[o:o + n] = st["exp_avg_sq"].to(device=dev, dtype=torch.float32).reshape(-1)
        if self.mode == "sharded":
            self.m.copy_(m[self.ex.i0:self.ex.i1]); self.v.copy_(v[self.ex.i0:self.ex.i1])
        else:
            self.m.copy_(m); self.v.copy_(v)
        self.steps_done = steps.pop() if steps else 0
        extra = sd.get("srfrd") or {}
        seed = int(extra.get("seed", int(self.state[1].item()))) & 0x7FFFFFFF
        self.state.zero_()                                 # (ticket words of the fused tail included)
        self.state[0] = self.steps_done
        self.state[1] = seed
        # step size, bias correction and dropout seed of the NEXT step (t = steps_done + 1), as after an uninterrupted run.
        # (The captured graphs bake lr / betas in as launch arguments: they are re-captured.)
        check(_lib.lib().srfrd_step_begin(ptr(self.state), self.lr, self.betas[0], self.betas[1], self._stream()), "srfrd_step_begin")
        self._graph_a = self._graph_b = self._graph_f = self._graph_u = self._graph_one = None
        self._fresh = False

    def _check_slot(self, slot: int = 0):
        ids, kind = self.ids_ring[slot], self.lay.kind
        n = self.B * self.L
        embeds_fake = kind in (1, 2)
        check(_lib.lib().srfrd_check_ids(ptr(ids[0]), ptr(ids[2]), ptr(ids[4]), ptr(ids[1]) if embeds_fake else None,
                                         ptr(ids[3]) if kind == 2 else None, ptr(ids[5]) if kind == 2 else None,
                                         n, self.lay.n_items, 2, ptr(self.err), self._stream()), "srfrd_check_ids")

    def check(self):
        """Raise IndexError if a batch given to step() / step_packed() held an id outside the embedding tables (the
        kernels clamp such ids: memory stays safe, the step's result does not count).  Synchronises the device."""
        bits = int(self.err.item())
        if bits:
            self.err.zero_()
            raise IndexError("index out of range in self: a training batch held " + " and ".join(
                n for b, n in ((1, f"an item id outside [0, {self.lay.n_items}]"), (2, "a fake / review id outside [0, 2]"))
                if bits & b))

    def step_packed(self, batch6: torch.Tensor) -> torch.Tensor:
        """One train step on a packed int64 (6, B, L) tensor [seq, rsq, pos, prs, neg, nrs]; returns the device loss."""
        self.ids.copy_(batch6, non_blocking=True)
        self._check_slot(0)
        return self._run()

    def step(self, user_ids, input_ids, fake_ids, positive_ids, positive_fake_ids, negative_ids, negative_fake_ids):
        """Same argument order as the reference model call at trainer.py:30 (``user_ids`` is unused there too)."""
        for k, t in enumerate((input_ids, fake_ids, positive_ids, positive_fake_ids, negative_ids, negative_fake_ids)):
            self.ids[k].copy_(t, non_blocking=True)
        self._check_slot(0)
        return self._run()

    def step_slot(self, slot: int) -> torch.Tensor:
        """One train step on input slot `slot` of `ids_ring` (already filled by the caller, stream-ordered before this
        call): the zero-copy form of step_packed().  The slot's ids are the producer's responsibility (DeviceSampler
        checks its dataset against the model once, at construction); call ``_check_slot(slot)`` + ``check()`` to
        validate a hand-filled slot."""
        if not 0 <= slot < self.slots:
            raise IndexError(f"slot {slot} outside the ring of {self.slots}")
        return self._run(slot)

    def _run(self, slot: int = 0):
        if not self._fresh:
            self.refresh()
        g = self.use_graph
        if g and self._graph_a is None and self._graph_one is None:
            self._capture()
        if not g:
            if self.mode == "single":
                self._enqueue_compute(slot)
                self._enqueue_update()
            else:
                self._enqueue_dp_step(slot)
        elif self._graph_one is not None:          # data parallel, collectives captured: ONE host launch per step
            self._graph_one[slot].replay()
        elif self.mode == "single":
            self._graph_a[slot].replay()
        elif self.mode == "allreduce":             # split form: graphs around the host-launched collectives
            self._graph_a[slot].replay()
            self.ex.all_reduce(self.grad)
            self._graph_b.replay()
        else:
            self._graph_f[slot].replay()
            h = self.ex.all_reduce_stats(self.stats)               # 16 bytes, in flight under the backward
            self._graph_a[slot].replay()
            self.ex.reduce_scatter(self.grad, self.recv)
            if h is not None:
                h.wait()
            self._graph_u.replay()
            self._gather_params()
            self._graph_b.replay()
        self.steps_done += 1
        return self.loss
