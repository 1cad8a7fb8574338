"""Fused gradient-value clipping + Adam.

``FusedClipAdam`` is a ``torch.optim.Optimizer`` whose ``step()`` does, for ALL parameter tensors of a group in ONE HIP
kernel launch (``stpde_clip_adam_multi``; ``stpde_clip_adam`` is the single-tensor entry point), what the reference does with ``clip_grad_value_`` followed by ``optim.Adam.step``
(experiments/rb2d/train.py:79-83): same state names (``step``, ``exp_avg``, ``exp_avg_sq``), so
``state_dict()`` / ``load_state_dict()`` interoperate with ``torch.optim.Adam`` checkpoints written by the reference.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib

_CHUNK = 1 << 16     # elements per block of the multi-tensor kernel
_TENSOR_DT = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<i8"), ("step_size", "<f4"),
                      ("bias2_sqrt", "<f4")])      # = stpde_adam_tensor
_CHUNK_DT = np.dtype([("tensor", "<i4"), ("pad", "<i4"), ("offset", "<i8")])   # = stpde_adam_chunk


class FusedClipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip_grad=0.0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, clip_grad=clip_grad))

    def __setstate__(self, state):
        # Optimizer.load_state_dict replaces param_groups with the SAVED groups: a checkpoint written by the
        # reference's torch.optim.Adam (train.py:390-397) has no ``clip_grad`` entry, and carries Adam options
        # (amsgrad, maximize, foreach, ...) this optimizer does not implement -- keep this instance's clip value and
        # refuse the options that would change the update rule.
        super().__setstate__(state)
        for group in self.param_groups:
            group.setdefault("clip_grad", self.defaults["clip_grad"])
            if group.get("amsgrad") or group.get("maximize"):
                raise ValueError("FusedClipAdam does not implement amsgrad / maximize")

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            rows, chunks, keep = [], [], []
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
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                keep.append(g)
                ptrs = (p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr())
                if any(q & 15 for q in ptrs):
                    raise RuntimeError("FusedClipAdam needs 16-byte aligned tensors")
                n = p.numel()
                for off in range(0, n, _CHUNK):
                    chunks.append((len(rows), 0, off))
                rows.append(ptrs + (n, group["lr"] / (1.0 - b1 ** t), math.sqrt(1.0 - b2 ** t)))
            if not rows:
                continue
            dev = keep[0].device
            # ONE launch for the whole group: tensor table + chunk table (a few KB) are uploaded per step
            tab = torch.from_numpy(np.array(rows, dtype=_TENSOR_DT).view(np.uint8)).to(dev)
            chk = torch.from_numpy(np.array(chunks, dtype=_CHUNK_DT).view(np.uint8)).to(dev)
            d = _lib.AdamDesc()
            d.n = 0
            d.clip, d.beta1, d.beta2, d.eps = float(group["clip_grad"] or 0.0), b1, b2, group["eps"]
            d.weight_decay, d.step_size, d.bias2_sqrt = group["weight_decay"], 0.0, 1.0
            with _lib.device_of(tab):
                _lib.check(L.stpde_clip_adam_multi(C.byref(d), _lib.ptr(tab), _lib.ptr(chk), len(chunks), _CHUNK,
                                                   _lib.stream_ptr()))
                tab.record_stream(torch.cuda.current_stream())
                chk.record_stream(torch.cuda.current_stream())
        return loss
