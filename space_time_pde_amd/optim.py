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
    """clip_grad_value_ + Adam over FLAT buffers (round 3).

    ``flat=True`` (default): on the first step the parameters of a group are moved into ONE contiguous fp32 buffer (every
    ``p.data`` becomes a view of it, values preserved), and ``exp_avg`` / ``exp_avg_sq`` are views of two more buffers
    allocated once.  A step is then: gradients gathered into the flat gradient buffer (``torch._foreach_copy_``: a
    handful of multi-tensor launches; skipped for gradients that already live there, see ``gather_grads``) and ONE
    ``stpde_clip_adam`` launch over the whole buffer -- no per-step table build, no host-to-device upload.
    ``gather_grads()`` returns that flat gradient buffer with every ``p.grad`` re-pointed into it, so a data-parallel
    step all-reduces exactly the buffer the optimizer reads (reference: DDP's bucket, train_ddp.py:401-406).
    The per-parameter state keeps torch.optim.Adam's names and shapes; ``state_dict()`` returns plain (cloned) tensors,
    so checkpoints interoperate with the reference's Adam both ways.
    A group falls back to the multi-tensor pointer-table kernel for a step in which some parameter has no gradient or the
    step counts differ (torch's Adam skips such parameters; a flat pass could not).

    Aliasing: building the flat buffers (first ``step()`` / ``gather_grads()`` after construction, ``load_state_dict`` or
    ``add_param_group``) RE-POINTS every ``p.data`` into the new buffer.  The values are preserved, the storage is not: an
    alias of the old storage taken before that (an EMA copy made with ``p.data`` views, a captured hipGraph, a
    ``DistributedDataParallel`` bucket view) keeps pointing at memory the optimizer no longer updates.  Take such aliases
    after the first step, or construct with ``flat=False`` (pointer-table kernel, storages untouched)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip_grad=0.0, flat=True):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1:
            raise ValueError("invalid Adam hyper-parameters")
        # ``flat`` travels in ``defaults`` (what Optimizer.__getstate__ serialises), so a deepcopy / pickle of an optimizer
        # built with flat=False comes back with flat=False
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, clip_grad=clip_grad,
                                      flat=bool(flat)))
        self._use_flat = bool(flat)
        self._flat = {}          # id(group) -> dict(P, G, M, V, offs, views...)

    def __setstate__(self, state):
        # Optimizer.load_state_dict replaces param_groups with the SAVED groups: a checkpoint written by the
        # reference's torch.optim.Adam (train.py:390-397) has no ``clip_grad`` entry, and carries Adam options
        # (amsgrad, maximize, foreach, ...) this optimizer does not implement -- keep this instance's clip value and
        # refuse the options that would change the update rule.
        super().__setstate__(state)
        for group in self.param_groups:
            group.setdefault("clip_grad", self.defaults["clip_grad"])
            group.setdefault("flat", self.defaults.get("flat", True))
            if group.get("amsgrad") or group.get("maximize"):
                raise ValueError("FusedClipAdam does not implement amsgrad / maximize")
        self._flat = {}          # loaded state tensors are fresh allocations: rebuild the flat views on the next step
        # Optimizer.__getstate__ serialises defaults / state / param_groups only: an instance that comes back from
        # copy.deepcopy or pickle takes its mode from ``defaults`` (it used to come back as flat=True)
        self._use_flat = bool(self.defaults.get("flat", getattr(self, "_use_flat", True)))

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._flat = {}

    def add_param_group(self, group):
        super().add_param_group(group)
        if hasattr(self, "_flat"):
            self._flat = {}

    def state_dict(self):
        sd = super().state_dict()
        # plain tensors in NEW dicts (the base class hands out the live per-parameter dicts): no views of the flat buffers,
        # no shared step tensor -- what torch.optim.Adam writes
        sd["state"] = {k: {n: (v.clone() if torch.is_tensor(v) and (v._base is not None or n == "step") else v)
                           for n, v in st.items()} for k, st in sd["state"].items()}
        return sd

    # ---------------------------------------------------------------------------------------------------------
    def _flat_group(self, gi, group):
        """Flat buffers of a group (built once): parameters re-pointed into P, moments into M / V."""
        ent = self._flat.get(gi)
        params = group["params"]
        if ent is not None and len(ent["params"]) == len(params) and all(a is b for a, b in zip(ent["params"], params)):
            return ent
        if not params or not all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and
                                 p.device == params[0].device for p in params):
            return None
        dev = params[0].device
        offs, tot = [], 0
        for p in params:
            offs.append(tot)
            tot += (p.numel() + 3) // 4 * 4            # every tensor starts 16-byte aligned
        P, G, M, V = (torch.zeros(tot, device=dev) for _ in range(4))
        pv, gv = [], []
        for p, o in zip(params, offs):
            n = p.numel()
            view = P[o:o + n].view_as(p)
            view.copy_(p.data)
            p.data = view
            pv.append(view)
            gv.append(G[o:o + n].view_as(p))
            st = self.state[p]
            if not st:
                st["step"] = torch.tensor(0.0)
            for name, buf in (("exp_avg", M), ("exp_avg_sq", V)):
                mv = buf[o:o + n].view_as(p)
                if name in st:
                    mv.copy_(st[name])
                st[name] = mv
        ent = dict(params=list(params), P=P, G=G, M=M, V=V, offs=offs, n=tot, gviews=gv, step=None)
        self._share_step(ent)
        self._flat[gi] = ent
        return ent

    def _share_step(self, ent):
        """All parameters at the same step count: ONE shared ``step`` tensor (one host increment per step instead of one
        per parameter); un-shared again (``_unshare_step``) before a pointer-table step could advance them unequally."""
        steps = {float(self.state[p]["step"]) for p in ent["params"]}
        if len(steps) == 1:
            ent["step"] = torch.tensor(steps.pop())
            for p in ent["params"]:
                self.state[p]["step"] = ent["step"]

    def _unshare_step(self, ent):
        if ent.get("step") is not None:
            for p in ent["params"]:
                self.state[p]["step"] = self.state[p]["step"].clone()
            ent["step"] = None

    def gather_grads(self, group_index=0):
        """Flat gradient buffer of a parameter group with every existing ``p.grad`` copied in and re-pointed to its view
        (so the following ``step()`` finds them in place).  Parameters without a gradient contribute zeros."""
        group = self.param_groups[group_index]
        ent = self._flat_group(group_index, group)
        if ent is None:
            raise RuntimeError("gather_grads needs contiguous fp32 CUDA parameters on one device")
        self._gather(ent)
        return ent["G"]

    @staticmethod
    def _gather(ent):
        src, dst = [], []
        for p, gvw in zip(ent["params"], ent["gviews"]):
            g = p.grad
            if g is None:
                gvw.zero_()
            elif g.data_ptr() != gvw.data_ptr():
                src.append(g)
                dst.append(gvw)
        if src:
            torch._foreach_copy_(dst, src)
            for p, gvw in zip(ent["params"], ent["gviews"]):
                if p.grad is not None:
                    p.grad = gvw

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            ent = self._flat_group(gi, group) if self._use_flat else None
            if ent is not None and all(p.grad is not None for p in ent["params"]):
                if ent["step"] is None:
                    self._share_step(ent)
                if ent["step"] is not None:
                    ent["step"] += 1
                    t = float(ent["step"])
                    self._gather(ent)
                    d = _lib.AdamDesc()
                    d.n = ent["n"]
                    d.clip, d.beta1, d.beta2, d.eps = float(group["clip_grad"] or 0.0), b1, b2, group["eps"]
                    d.weight_decay = group["weight_decay"]
                    d.step_size, d.bias2_sqrt = group["lr"] / (1.0 - b1 ** t), math.sqrt(1.0 - b2 ** t)
                    with _lib.device_of(ent["P"]):
                        _lib.check(L.stpde_clip_adam(C.byref(d), _lib.ptr(ent["P"]), _lib.ptr(ent["G"]),
                                                     _lib.ptr(ent["M"]), _lib.ptr(ent["V"]), _lib.stream_ptr()))
                    continue
            if ent is not None:
                self._unshare_step(ent)
            self._step_table(L, group)
        return loss

    def _step_table(self, L, group):
        """Multi-tensor pointer-table launch: parameters with a gradient only, per-tensor step counts."""
        b1, b2 = group["betas"]
        if True:
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
                return
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
