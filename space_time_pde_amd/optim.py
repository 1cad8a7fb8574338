"""Fused gradient-value clipping + Adam.

``FusedClipAdam`` is a ``torch.optim.Optimizer`` whose ``step()`` does, per parameter tensor, in ONE HIP kernel
(``stpde_clip_adam``) what the reference does with ``clip_grad_value_`` followed by ``optim.Adam.step``
(experiments/rb2d/train.py:79-83): same state names (``step``, ``exp_avg``, ``exp_avg_sq``), so
``state_dict()`` / ``load_state_dict()`` interoperate with ``torch.optim.Adam`` checkpoints written by the reference.
"""
import ctypes as C
import math

import torch

from . import _lib


class FusedClipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip_grad=0.0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, clip_grad=clip_grad))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = None
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise RuntimeError("FusedClipAdam needs contiguous fp32 CUDA parameters (no CPU fallback)")
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                t = float(st["step"])
                d = _lib.AdamDesc()
                d.n = p.numel()
                d.clip, d.beta1, d.beta2, d.eps = float(group["clip_grad"] or 0.0), b1, b2, group["eps"]
                d.weight_decay = group["weight_decay"]
                d.step_size = group["lr"] / (1.0 - b1 ** t)
                d.bias2_sqrt = math.sqrt(1.0 - b2 ** t)
                if L is None:
                    L = _lib.lib()
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                _lib.check(L.stpde_clip_adam(C.byref(d), _lib.ptr(p), _lib.ptr(g), _lib.ptr(st["exp_avg"]),
                                             _lib.ptr(st["exp_avg_sq"]), _lib.stream_ptr()))
        return loss
