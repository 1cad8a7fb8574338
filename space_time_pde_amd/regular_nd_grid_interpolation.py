"""Regular n-d grid multilinear interpolation (mirrors src/regular_nd_grid_interpolation.py of the reference).

Public API (same names / arguments / return values as the reference):
    clip_tensor(input_tensor, xmin, xmax)
    regular_nd_grid_interpolation_coefficients(grid, query_pts, xmin=0., xmax=1.)
    regular_nd_grid_interpolation(grid, query_pts, xmin=0., xmax=1.)

Dispatch: CUDA fp32 tensors with dim <= 4 whose query points do not need gradients go to the HIP kernels
(``stpde_interp_fwd`` / ``stpde_interp_bwd_grid``; gradient w.r.t. the grid is supported).  Anything that needs
derivatives w.r.t. the query coordinates through generic autograd (the reference's ``dif`` on an arbitrary model),
or lives on the CPU, uses the composed-torch-op formulation below, which is differentiable to any order.
The hot path (IM-NET decoder, dim=3) never comes through here: see local_implicit_grid.py / lig_jet.py.
"""
import ctypes as C
import itertools

import numpy as np
import torch

from . import _lib
from .lig_jet import cached_box_constants


def clip_tensor(input_tensor, xmin, xmax):
    """Clip tensor by per-column bounds (reference :9-11)."""
    return torch.max(torch.min(input_tensor, xmax), xmin)


def _bounds(grid, xmin, xmax):
    dim = grid.dim() - 2
    if isinstance(xmin, (int, float)) or isinstance(xmax, (int, float)):
        xmin = float(xmin) * torch.ones([dim], dtype=torch.float32, device=grid.device)
        xmax = float(xmax) * torch.ones([dim], dtype=torch.float32, device=grid.device)
    elif isinstance(xmin, (list, tuple, np.ndarray)) or isinstance(xmax, (list, tuple, np.ndarray)):
        xmin = torch.tensor(xmin).to(grid.device)
        xmax = torch.tensor(xmax).to(grid.device)
    return xmin, xmax


def _coefficients_autograd(grid, query_pts, xmin, xmax):
    """Composed-torch-op formulation (any device, any differentiation order)."""
    dim = grid.dim() - 2
    size = torch.tensor(grid.shape[1:-1]).float().to(grid.device)
    xmin, xmax = _bounds(grid, xmin, xmax)
    eps = 1e-6 * (xmax - xmin)
    q = clip_tensor(query_pts, xmin + eps, xmax - eps)
    cubesize = (xmax - xmin) / (size - 1)
    ind0 = torch.floor(q / cubesize).long()
    near = ind0.float() * cubesize
    far = (ind0.float() + 1) * cubesize
    nb = grid.shape[0]
    bsel = torch.arange(nb, device=grid.device).view(nb, 1).expand(nb, query_pts.shape[1])
    cv, wt, rel = [], [], []
    for bits in itertools.product((0, 1), repeat=dim):  # first dim most significant, as the reference's com_
        take = torch.tensor(bits, device=grid.device, dtype=torch.bool)
        pos = torch.where(take, far, near)
        opposite = torch.where(take, near, far)
        idx = ind0 + take.long()
        cv.append(grid[(bsel,) + tuple(idx[..., k] for k in range(dim))])
        wt.append(torch.prod(torch.abs(q - opposite) / cubesize, dim=-1))
        rel.append((q - pos) / cubesize)
    return torch.stack(cv, dim=2), torch.stack(wt, dim=2), torch.stack(rel, dim=2)


def _hip_eligible(grid, query_pts):
    return (grid.is_cuda and query_pts.is_cuda and grid.dtype == torch.float32 and query_pts.dtype == torch.float32
            and 1 <= grid.dim() - 2 <= 4 and not (query_pts.requires_grad and torch.is_grad_enabled()))


def _desc(grid, query_pts, xmin, xmax):
    """Kernel descriptor, or None when the bounds are outside the HIP envelope (xmin != 0: the reference's cell index
    ignores xmin, quirk a-Q1; the composed formulation below reproduces whatever the reference does there)."""
    dim = grid.dim() - 2
    try:
        lo_c, hi_c, cube = cached_box_constants(tuple(grid.shape[1:-1]), xmin, xmax)
    except ValueError:
        return None
    d = _lib.InterpDesc()
    d.B, d.N = query_pts.shape[0], query_pts.shape[1]
    d.P, d.dim, d.C = d.B * d.N, dim, grid.shape[-1]
    for k in range(dim):
        d.n[k], d.lo_c[k], d.hi_c[k], d.cube[k] = grid.shape[1 + k], lo_c[k], hi_c[k], cube[k]
    return d


class _InterpHip(torch.autograd.Function):
    @staticmethod
    @_lib.guarded
    def forward(ctx, grid, pts, desc, want_coeffs):
        L = _lib.lib()
        g = grid.contiguous()
        p = pts.detach().contiguous()
        dim, nc, P, Cc = desc.dim, 1 << desc.dim, desc.P, desc.C
        dev = g.device
        if want_coeffs:
            cv = torch.empty(desc.B, desc.N, nc, Cc, device=dev)
            wt = torch.empty(desc.B, desc.N, nc, device=dev)
            rl = torch.empty(desc.B, desc.N, nc, dim, device=dev)
            _lib.check(L.stpde_interp_fwd(C.byref(desc), _lib.ptr(g), _lib.ptr(p), None, _lib.ptr(cv), _lib.ptr(wt),
                                          _lib.ptr(rl), _lib.stream_ptr()))
            out = (cv, wt, rl)
            ctx.mark_non_differentiable(wt, rl)
        else:
            o = torch.empty(desc.B, desc.N, Cc, device=dev)
            _lib.check(L.stpde_interp_fwd(C.byref(desc), _lib.ptr(g), _lib.ptr(p), _lib.ptr(o), None, None, None,
                                          _lib.stream_ptr()))
            out = (o,)
        ctx.desc, ctx.pts, ctx.gshape, ctx.want = desc, p, g.shape, want_coeffs
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    @_lib.guarded
    def backward(ctx, *gouts):
        if not ctx.needs_input_grad[0]:
            return None, None, None, None
        L = _lib.lib()
        dgrid = torch.zeros(ctx.gshape, device=ctx.pts.device)
        gb = gouts[0].contiguous()
        if ctx.want:
            _lib.check(L.stpde_interp_bwd_grid(C.byref(ctx.desc), _lib.ptr(ctx.pts), None, _lib.ptr(gb),
                                               _lib.ptr(dgrid), _lib.stream_ptr()))
        else:
            _lib.check(L.stpde_interp_bwd_grid(C.byref(ctx.desc), _lib.ptr(ctx.pts), _lib.ptr(gb), None,
                                               _lib.ptr(dgrid), _lib.stream_ptr()))
        return dgrid, None, None, None


def regular_nd_grid_interpolation_coefficients(grid, query_pts, xmin=0., xmax=1.):
    """Batched regular n-d grid interpolation coefficients (reference :14-78).

    grid (batch, *size, in_features); query_pts (batch, num_points, dim) -> corner_values
    (batch, num_points, 2**dim, in_features), weights (batch, num_points, 2**dim), x_relative
    (batch, num_points, 2**dim, dim) in [-1, 1].
    """
    desc = _desc(grid, query_pts, xmin, xmax) if _hip_eligible(grid, query_pts) else None
    if desc is not None:
        return _InterpHip.apply(grid, query_pts, desc, True)
    return _coefficients_autograd(grid, query_pts, xmin, xmax)


def regular_nd_grid_interpolation(grid, query_pts, xmin=0., xmax=1.):
    """Batched multilinear interpolation of grid values at query points (reference :81-104)."""
    desc = _desc(grid, query_pts, xmin, xmax) if _hip_eligible(grid, query_pts) else None
    if desc is not None:
        return _InterpHip.apply(grid, query_pts, desc, False)[0]
    corner_values, weights, _ = _coefficients_autograd(grid, query_pts, xmin, xmax)
    return torch.sum(corner_values * weights.unsqueeze(-1), dim=-2)
