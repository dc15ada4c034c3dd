"""`srfrd_amd.Adam`: torch.optim.Adam's update (reference trainer.py:390: ``Adam(model.parameters(), lr, betas=(0.9, 0.98))``)
for the module-level drop-in path, stepped by ONE launch over the model's flat parameter vector.

The reference loop (trainer.py:29-41) stays as it is - ``model(...)`` -> BCE -> ``loss.backward()`` -> ``optimizer.step()`` - only
the optimizer's constructor changes.  Every parameter of a srfrd_amd module is a view of one fp32 vector
``[item table | pad | dense]`` and the backward op returns every gradient as a view of one flat gradient vector, so the step
is ``srfrd_adam_step`` over the whole vector (same arithmetic as the fused trainer's tail; torch's foreach Adam walks 31
tensors with a handful of launches each and costs more host time than the model's kernels take).  Falls back to gathering the
gradients into a flat buffer when they are not views of one vector (gradient accumulation over several backward calls,
hooks that replace them).  ``state_dict()`` / ``load_state_dict()`` use torch.optim.Adam's own format, so a run can move
between the two optimizers (and ``FusedTrainer``)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import check, ptr


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 amsgrad: bool = False):
        if weight_decay != 0.0 or amsgrad:
            raise ValueError("srfrd_amd.Adam implements plain Adam (the reference's optimizer): no weight decay / amsgrad")
        super().__init__(params, dict(lr=float(lr), betas=(float(betas[0]), float(betas[1])), eps=float(eps)))
        if len(self.param_groups) != 1:
            raise ValueError("srfrd_amd.Adam takes ONE parameter group: model.parameters() of one srfrd_amd module")
        self._flat = self._m = self._v = self._gbuf = self._dev_state = None
        self._spans = None
        self._p0 = self._pl = None
        self._steps = 0

    # ---- the flat vector behind the parameters -------------------------------------------------------------------------
    def _resolve(self):
        ps = self.param_groups[0]["params"]
        st = ps[0].untyped_storage()
        base = st.data_ptr()
        if any(p.untyped_storage().data_ptr() != base or p.dtype != torch.float32 or not p.is_contiguous() for p in ps):
            raise RuntimeError("srfrd_amd.Adam steps the flat parameter vector of a srfrd_amd module: run one forward first (the "
                               "parameters become views of it then), and pass model.parameters() of ONE model")
        spans = sorted((p.storage_offset(), p.numel()) for p in ps)
        n = (spans[-1][0] + spans[-1][1] + 3) // 4 * 4
        flat = torch.empty(0, device=ps[0].device, dtype=torch.float32).set_(st, 0, (n,))
        if self._flat is None or self._flat.data_ptr() != flat.data_ptr() or self._flat.numel() != n:
            dev = flat.device
            self._flat = flat
            m_old, v_old = self._m, self._v
            self._m = torch.zeros(n, device=dev, dtype=torch.float32)
            self._v = torch.zeros(n, device=dev, dtype=torch.float32)
            if m_old is not None and m_old.numel() == n:      # (the model re-flattened, e.g. moved: keep the moments)
                self._m.copy_(m_old); self._v.copy_(v_old)
            self._gbuf = None
            self._dev_state = torch.zeros(32, device=dev, dtype=torch.int32)
            self._dev_state[0] = self._steps
        self._spans = [(p, p.storage_offset(), p.numel()) for p in ps]

    def _flat_grad(self):
        """the gradients as one vector aligned with the flat parameters: in place when they already are views of one"""
        base = last = None
        for p, off, n in self._spans:
            g = p.grad
            if g is None or not g.is_contiguous():
                base = None
                break
            b = g.data_ptr() - 4 * off
            if base is None:
                base = b
            elif b != base:
                base = None
                break
            last = g
        if base is not None and last.dtype == torch.float32 and last.device == self._flat.device:
            # (one dtype / device test: tensors laid out at the parameters' own offsets of one buffer are views of one vector)
            st = last.untyped_storage()
            if st.data_ptr() <= base and base + 4 * self._flat.numel() <= st.data_ptr() + st.nbytes():
                return C.c_void_p(base), last          # (keep a reference alive until the launch is enqueued)
        if self._gbuf is None:
            self._gbuf = torch.zeros_like(self._flat)
        for p, off, n in self._spans:
            if p.grad is None:
                self._gbuf[off:off + n].zero_()
            else:
                self._gbuf[off:off + n].copy_(p.grad.reshape(-1))
        return ptr(self._gbuf), self._gbuf

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        ps = self.param_groups[0]["params"]
        if all(p.grad is None for p in ps):
            return loss
        if ps[0].device.type != "cuda":
            raise RuntimeError("srfrd_amd.Adam runs on the ROCm GPU only (no CPU fallback)")
        if self._spans is None or ps[0].data_ptr() != self._p0 or ps[-1].data_ptr() != self._pl:
            self._resolve()             # (first step, or the model re-flattened its parameters)
            self._p0, self._pl = ps[0].data_ptr(), ps[-1].data_ptr()
        g = self.param_groups[0]
        lr, (b1, b2), eps = g["lr"], g["betas"], g["eps"]
        gptr, keep = self._flat_grad()
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L_ = _lib.lib()
        # t += 1 and the bias corrections in double precision on the device, then the step over the whole vector
        check(L_.srfrd_step_begin(ptr(self._dev_state), lr, b1, b2, st), "srfrd_step_begin")
        n = self._flat.numel()
        check(L_.srfrd_adam_step(ptr(self._flat), gptr, ptr(self._m), ptr(self._v), n, 0, n, 0, b1, b2, eps,
                                 ptr(self._dev_state), None, None, 0, st), "srfrd_adam_step")
        self._steps += 1
        del keep
        return loss

    # ---- torch.optim.Adam's state format ---------------------------------------------------------------------------------
    def state_dict(self):
        g = self.param_groups[0]
        ps = g["params"]
        state = {}
        if self._spans is not None and self._steps > 0:
            for i, (p, off, n) in enumerate(self._spans):
                state[i] = {"step": torch.tensor(float(self._steps)), "exp_avg": self._m[off:off + n].view(p.shape).clone(),
                            "exp_avg_sq": self._v[off:off + n].view(p.shape).clone()}
        group = {"lr": g["lr"], "betas": g["betas"], "eps": g["eps"], "weight_decay": 0, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "decoupled_weight_decay": False,
                 "params": list(range(len(ps)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        group = sd["param_groups"][0]
        if any(group.get(k) for k in ("weight_decay", "amsgrad", "maximize")):
            raise ValueError("srfrd_amd.Adam implements plain Adam (no weight decay / amsgrad / maximize)")
        g = self.param_groups[0]
        g["lr"], g["betas"], g["eps"] = float(group["lr"]), (float(group["betas"][0]), float(group["betas"][1])), float(group["eps"])
        steps = {int(float(s["step"])) for s in sd["state"].values()}
        if len(steps) > 1:
            raise ValueError("parameters with different step counts")
        self._steps = steps.pop() if steps else 0
        self._resolve()
        self._m.zero_(); self._v.zero_()
        for i, (p, off, n) in enumerate(self._spans):
            s = sd["state"].get(i)
            if s is not None:
                self._m[off:off + n] = s["exp_avg"].to(device=p.device, dtype=torch.float32).reshape(-1)
                self._v[off:off + n] = s["exp_avg_sq"].to(device=p.device, dtype=torch.float32).reshape(-1)
        self._dev_state.zero_()
        self._dev_state[0] = self._steps
