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
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))

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
