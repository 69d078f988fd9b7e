"""FusedSGD -- ``torch.optim.SGD(params, lr, momentum, weight_decay)`` of reference ``train.py:239-246`` with the
update done by the multi-tensor HIP kernel (``sh_sgd_step``): a handful of launches per step instead of one per
parameter.  Subclasses ``torch.optim.Optimizer`` so ``zero_grad`` / ``state_dict`` / ``param_groups`` behave as usual;
the per-parameter state key is ``momentum_buffer`` like torch's.
"""
import torch

from . import ops


class FusedSGD(torch.optim.Optimizer):
    def __init__(self, params, lr, momentum=0.9, weight_decay=1e-4):
        if lr < 0 or momentum < 0 or weight_decay < 0:
            raise ValueError("lr, momentum and weight_decay must be non-negative")
        # the extra keys are torch.optim.SGD's own (at their defaults -- the only values the kernel implements), so that a
        # state_dict written here loads into the reference's ``optim.SGD`` and vice versa (train.py:419-435 checkpoints)
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=0, weight_decay=weight_decay, nesterov=False,
                                      maximize=False, foreach=None, differentiable=False, fused=None))

    def adopt_layouts(self):
        """After ``load_state_dict``: momentum buffers must live in their parameter's memory layout (OHWI conv weights)."""
        for group in self.param_groups:
            if group.get("dampening", 0) != 0 or group.get("nesterov", False) or group.get("maximize", False):
                raise ValueError("FusedSGD implements plain momentum SGD only (dampening=0, nesterov=False, maximize=False)")
            for p in group["params"]:
                st = self.state.get(p)
                if st and st.get("momentum_buffer") is not None and st["momentum_buffer"].stride() != p.stride():
                    st["momentum_buffer"] = torch.empty_like(p).copy_(st["momentum_buffer"])

    def zero_grad(self, set_to_none=True):
        """As ``torch.optim.Optimizer.zero_grad`` for the case the training step uses (set_to_none): a plain loop -- the base class's version
        costs 0.4 ms per call for these 161 parameters, at the step boundary where the GPU queue is empty."""
        if not set_to_none:
            return super().zero_grad(set_to_none=False)
        for group in self.param_groups:
            for p in group["params"]:
                p.grad = None

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            fresh, old = ([], [], []), ([], [], [])
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad
                if g.stride() != p.stride():            # the kernel walks flat memory: layouts must agree
                    g = torch.empty_like(p).copy_(g)
                st = self.state[p]
                dst = old
                if "momentum_buffer" not in st:
                    st["momentum_buffer"] = torch.empty_like(p)
                    dst = fresh
                dst[0].append(p); dst[1].append(g); dst[2].append(st["momentum_buffer"])
            for first, (ps, gs, vs) in ((True, fresh), (False, old)):
                if ps:
                    ops.sgd_step(ps, gs, vs, group["lr"], group["momentum"], group["weight_decay"], first, grad_scale)
        return loss
