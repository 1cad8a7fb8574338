"""Deterministic, torch-version-independent parameter / input generation shared by ``make_golden.py`` (which applies
it to the REAL reference modules in the build container) and by the tests (which apply it to this package's modules).

Large networks (the C1 U-Net has 4 M parameters) cannot be stored in a fixture, and ``torch.manual_seed`` streams are
not guaranteed across torch builds; numpy's ``default_rng`` (PCG64) stream is.  Every tensor of a module's
``state_dict()`` (in its own order, which is identical between the reference and this package: same submodule names)
gets values from its own generator seeded with ``(seed, index)``.
"""
import math

import numpy as np
import torch


def fill_module_(module, seed):
    """Overwrite every parameter / buffer of ``module`` in state_dict order:
    weights (dim >= 2): U(-b, b), b = 1/sqrt(fan_in);  Linear / Conv biases (1-D, not BatchNorm): U(-b, b) of the
    preceding weight; BatchNorm weight: 1 + 0.1 U(-1, 1), bias: 0.1 U(-1, 1), running_mean: 0.1 U(-1, 1),
    running_var: 1 + 0.2 U(0, 1); integer buffers (num_batches_tracked) are left alone; scalar parameters (swish beta)
    are left alone.  Tensors that alias an earlier one (ImNet's fc0 / fc.0 duplicates) are written once."""
    sd = module.state_dict()
    seen = {}
    last_bound = 1.0
    with torch.no_grad():
        for i, (name, t) in enumerate(sd.items()):
            if not t.dtype.is_floating_point or t.dim() == 0:
                continue
            key = t.data_ptr()
            if key in seen:
                continue
            seen[key] = name
            rng = np.random.default_rng([seed, i])
            leaf = name.rsplit(".", 1)[-1]
            parent_is_bn = "bn" in name.rsplit(".", 2)[-2] if name.count(".") >= 1 else False
            if t.dim() >= 2:
                fan_in = int(np.prod(t.shape[1:]))
                last_bound = 1.0 / math.sqrt(fan_in)
                v = rng.uniform(-last_bound, last_bound, size=tuple(t.shape))
            elif parent_is_bn:
                u = rng.uniform(-1.0, 1.0, size=tuple(t.shape))
                if leaf == "weight":
                    v = 1.0 + 0.1 * u
                elif leaf == "running_var":
                    v = 1.0 + 0.1 * (u + 1.0)
                else:
                    v = 0.1 * u
            else:
                v = rng.uniform(-last_bound, last_bound, size=tuple(t.shape))
            t.copy_(torch.from_numpy(np.asarray(v, dtype=np.float32)))
    return module


def normal(seed, *shape):
    return torch.from_numpy(np.random.default_rng(seed).standard_normal(shape).astype(np.float32))


def uniform(seed, *shape, lo=0.0, hi=1.0):
    return torch.from_numpy(np.random.default_rng(seed).uniform(lo, hi, shape).astype(np.float32))
