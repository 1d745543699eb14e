"""AdamW on the HIP path: the optimizer step of ``train.py:181-182``.

``NativeAdamW`` is a ``torch.optim.Optimizer`` with torch.optim.AdamW's semantics (decoupled weight decay, bias
correction; verified against it in tests/test_kernels_gpu.py) whose update is one ``ocn_adamw_step`` launch per
tensor.  ``param_groups_like_reference`` reproduces the reference's grouping rule (optim.py:67-77,178-208:
1-D params and ``model.no_weight_decay()`` names get weight_decay 0)."""
import torch

from . import ops


def param_groups_like_reference(model, weight_decay=0.2):
    skip = set(model.no_weight_decay()) if hasattr(model, "no_weight_decay") else set()
    decay, no_decay = [], []
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        n = n[len("module."):] if n.startswith("module.") else n
        (no_decay if (p.ndim <= 1 or n in skip) else decay).append(p)  # optim.py:67-77 exclude_from_wd + no_weight_decay()
    return [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": weight_decay}]


class NativeAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=5e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                ops.adamw_step(p.view(-1) if p.ndim == 0 else p, g.view(-1) if g.ndim == 0 else g, st["exp_avg"].view(-1) if p.ndim == 0 else st["exp_avg"],
                               st["exp_avg_sq"].view(-1) if p.ndim == 0 else st["exp_avg_sq"], group["lr"], b1, b2, group["eps"],
                               group["weight_decay"], st["step"])
                torch.autograd.graph.increment_version(p)  # raw-pointer update: let the bf16 weight cache see it
        return loss
