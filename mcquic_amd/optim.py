"""The optimizer update of the training step (mcquic/train/trainer.py:283 `self._optimizer.step()`; the reference trains with
`Adam`, lr 1e-4, configs/a800_8.yaml:20-25) as ONE launch over the whole model: `mcq_adam_step_f32` (csrc/train_ops.hip).

torch.optim.Adam(fused=True) hands its tensor lists to the GPU through kernel arguments, 4 KB at a time: 19 launches of ~91 us for
the qp=2 model's 666 tensors, 1.7 ms per step at 0.8 TB/s (docs/experiments.md section 9.11).  Here the lists are device arrays
(pointer tables + a chunk table), built once and rebuilt only when an address changes, and both moments live in two flat buffers
this object owns; the update is a one-thread kernel (step count, bias corrections, on the device) plus one pass over 28 bytes per
element.  Arithmetic and state layout are torch.optim.Adam's / AdamW's: `state_dict()` / `load_state_dict()` exchange checkpoints with
them (per-parameter `step`, `exp_avg`, `exp_avg_sq`), a learning rate given as a device tensor is read by the kernel on every
call (a scheduler fills it; a captured step needs no re-capture), and nothing is read back by the host, so
`parallel.GraphedTrainStep` captures it like any capturable optimizer.  float32 parameters on a HIP device only: there is no CPU path."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from .ops import check, _guard, _stream

__all__ = ["Adam", "AdamW"]


class Adam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr, betas, eps, weight_decay, maximize=...) without `amsgrad` / `foreach` / `differentiable`;
    `decoupled=True` makes the decay AdamW's (`AdamW` below sets it and AdamW's default of 1e-2)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0, *, decoupled: bool = False,
                 maximize: bool = False):
        if not torch.is_tensor(lr) and not lr >= 0.0:
            raise ValueError(f"Invalid learning rate: {lr}")
        if not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"Invalid betas: {betas}")
        if not eps >= 0.0 or not weight_decay >= 0.0:
            raise ValueError("eps and weight_decay must be non-negative")
        # (capturable=True is what torch's load_state_dict looks at to keep `step` a float32 device tensor)
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, decoupled=bool(decoupled),
                                      maximize=bool(maximize), capturable=True))
        self._plans = {}                                      # group index -> _Plan

    # ---- state in flat buffers -------------------------------------------------------------------------------------------------
    class _Plan:
        __slots__ = ("key", "tables", "numel", "blk_tensor", "blk_first", "nblocks", "flat_m", "flat_v", "views", "step", "scalars", "ntensors",
                     "sizes", "ids", "adopted")

    def _flat_state(self, gi: int, params):
        """(flat_m, flat_v, [(m_view, v_view)], step) of group `gi` for `params`; state found in `self.state` that does not live in
        the flat buffers (loaded from a checkpoint, or set by hand) is copied in and replaced by views."""
        plan = self._plans.get(gi)
        ids = [id(p) for p in params]
        if plan is not None and plan.ids == ids and plan.adopted and plan.flat_m.device == params[0].device:
            return plan
        sizes = [p.numel() for p in params]
        if plan is None or plan.ids != ids or plan.flat_m.device != params[0].device:
            # (also when the set of parameters that carry a gradient changed: fresh buffers, the old state is copied over below)
            plan = self._plans[gi] = Adam._Plan()
            dev = params[0].device
            plan.sizes, plan.ids = sizes, ids
            offs, at = [], 0
            for n in sizes:
                offs.append(at)
                at += (n + 3) // 4 * 4                        # 16-byte aligned slices
            plan.flat_m = torch.zeros(at, dtype=torch.float32, device=dev)
            plan.flat_v = torch.zeros(at, dtype=torch.float32, device=dev)
            plan.views = [(plan.flat_m[o: o + n].view_as(p), plan.flat_v[o: o + n].view_as(p)) for o, n, p in zip(offs, sizes, params)]
            plan.step = torch.zeros((), dtype=torch.float32, device=dev)
            plan.scalars = torch.zeros(4, dtype=torch.float32, device=dev)
            plan.ntensors, plan.key = len(params), None
            # the chunk table depends on the sizes only; the pointer table is ONE device buffer for the plan's lifetime, refilled in
            # place when an address changes (a captured graph that reads it keeps a valid address and sees the current pointers)
            chunk = _lib.load().mcq_adam_chunk()
            blk_t, blk_f = [], []
            for i, n in enumerate(sizes):
                for first in range(0, n, chunk):
                    blk_t.append(i)
                    blk_f.append(first)
            plan.numel = torch.tensor(sizes, dtype=torch.int64).to(dev)
            plan.blk_tensor = torch.tensor(blk_t, dtype=torch.int32).to(dev)
            plan.blk_first = torch.tensor(blk_f, dtype=torch.int64).to(dev)
            plan.nblocks = len(blk_t)
            plan.tables = torch.zeros(4 * len(params), dtype=torch.int64, device=dev)
        for p, (mv, vv) in zip(params, plan.views):
            st = self.state[p]
            old_m, old_v, old_s = st.get("exp_avg"), st.get("exp_avg_sq"), st.get("step")
            if old_m is not None and old_m.data_ptr() != mv.data_ptr():
                mv.copy_(old_m.to(mv.device, torch.float32))
            if old_v is not None and old_v.data_ptr() != vv.data_ptr():
                vv.copy_(old_v.to(vv.device, torch.float32))
            if old_s is not None and old_s is not plan.step:  # (torch keeps one count per parameter; they move together)
                plan.step.copy_(torch.as_tensor(old_s, dtype=torch.float32).to(plan.step.device))
            st["exp_avg"], st["exp_avg_sq"], st["step"] = mv, vv, plan.step
        plan.adopted = True
        return plan

    def _tables(self, plan, params):
        key = tuple((p.data_ptr(), p.grad.data_ptr()) for p in params)
        if plan.key == key:
            return
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("mcquic_amd.optim.Adam: parameter / gradient addresses changed since the last step; call `prepare()` "
                               "before capturing (a host-to-device copy of the pointer table cannot be part of a graph)")
        ptrs = [p.data_ptr() for p in params] + [p.grad.data_ptr() for p in params] + [m.data_ptr() for m, _ in plan.views] + \
               [v.data_ptr() for _, v in plan.views]
        plan.tables.copy_(torch.tensor(ptrs, dtype=torch.int64))          # (stream-ordered behind the launches that read the old one)
        plan.key = key

    def _checked(self, group):
        params = [p for p in group["params"] if p.grad is not None]
        for p in params:
            g = p.grad
            if not (p.is_cuda and g.is_cuda and p.dtype == torch.float32 and g.dtype == torch.float32 and p.is_contiguous() and g.is_contiguous()
                    and not g.is_sparse):
                raise RuntimeError("mcquic_amd.optim.Adam: contiguous float32 parameters and gradients on a HIP device only "
                                   "(there is no CPU path)")
            if p.device != params[0].device:
                raise RuntimeError("mcquic_amd.optim.Adam: the parameters of one group must live on one device (one launch per group)")
        return params

    @torch.no_grad()
    def prepare(self) -> None:
        """Build the moment buffers and the device tables for the gradients the parameters hold NOW, without updating anything: what
        `parallel.GraphedTrainStep` calls before it captures `step()` (table uploads are host-to-device copies)."""
        for gi, group in enumerate(self.param_groups):
            params = self._checked(group)
            if params:
                self._tables(self._flat_state(gi, params), params)

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("mcquic_amd.optim.Adam: closures are not supported (the reference's trainer passes none)")
        lib = _lib.load()
        for gi, group in enumerate(self.param_groups):
            params = self._checked(group)
            if not params:
                continue
            plan = self._flat_state(gi, params)
            self._tables(plan, params)
            lr = group["lr"]
            lr_dev: Optional[torch.Tensor] = None
            if torch.is_tensor(lr):
                if lr.is_cuda:
                    if lr.dtype != torch.float32 or lr.numel() != 1:
                        raise TypeError("mcquic_amd.optim.Adam: a device learning rate must be one float32")
                    lr_dev = lr
                lr_host = 0.0 if lr_dev is not None else float(lr)
            else:
                lr_host = float(lr)
            b1, b2 = group["betas"]
            with _guard(params[0].device):
                check(lib.mcq_adam_step_f32(plan.tables.data_ptr(), plan.ntensors, plan.numel.data_ptr(), plan.blk_tensor.data_ptr(),
                                            plan.blk_first.data_ptr(), plan.nblocks, plan.step.data_ptr(),
                                            None if lr_dev is None else lr_dev.data_ptr(), float(lr_host), float(b1), float(b2),
                                            float(group["eps"]), float(group["weight_decay"]),
                                            1 if group["decoupled"] else 0, 1 if group["maximize"] else 0, plan.scalars.data_ptr(), _stream()),
                      "mcq_adam_step_f32")
        return None

    def state_dict(self):
        """torch.optim.Adam's layout.  Every parameter gets a `step` tensor of ITS OWN (here they all share one): torch's optimizers
        increment each entry they are handed, so a shared one would be advanced once per parameter after loading there."""
        sd = super().state_dict()
        sd["state"] = {k: {n: (v.clone() if n == "step" and torch.is_tensor(v) else v) for n, v in st.items()} for k, st in sd["state"].items()}
        # torch spells the decoupled decay `decoupled_weight_decay` (torch.optim.AdamW sets it): both names travel
        sd["param_groups"] = [dict(g, decoupled_weight_decay=bool(g.get("decoupled", False))) for g in sd["param_groups"]]
        return sd

    def load_state_dict(self, state_dict):
        """torch.optim.Adam / AdamW checkpoints load as they are (per-parameter `step`, `exp_avg`, `exp_avg_sq`); the tensors move
        into the flat buffers on the next `step()`."""
        super().load_state_dict(state_dict)
        for group in self.param_groups:
            # (torch replaces the groups by the saved ones: a torch.optim.AdamW checkpoint carries `decoupled_weight_decay: True` and
            #  no `decoupled` -- dropping it would turn the decay into Adam's L2 term silently)
            group["decoupled"] = bool(group.get("decoupled", group.get("decoupled_weight_decay", isinstance(self, AdamW))))
            group["decoupled_weight_decay"] = group["decoupled"]
            group.setdefault("maximize", False)
            group["capturable"] = True
            for k in ("amsgrad", "foreach", "fused", "differentiable"):
                if group.get(k):
                    raise NotImplementedError(f"mcquic_amd.optim.Adam: `{k}` checkpoints are not supported")
        # the loaded tensors are copied INTO the existing flat buffers on the next step (same parameters: same addresses, which a
        # captured update may hold); only a plan whose parameter set changed is rebuilt
        for plan in self._plans.values():
            plan.adopted = False


class AdamW(Adam):
    """torch.optim.AdamW: decoupled weight decay (param *= 1 - lr * weight_decay before the update), default 1e-2."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2, *, maximize: bool = False):
        super().__init__(params, lr, betas, eps, weight_decay, decoupled=True, maximize=maximize)
