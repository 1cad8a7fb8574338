"""CPU oracle: a from-scratch restatement of the MeshfreeFlowNet hot path in plain PyTorch.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (``space_time_pde_amd``) may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it,
and there only as the checker / the timed CPU baseline ("port"), never as the thing shipped.

Parity pin: every function here is checked against golden vectors produced by importing the real
reference (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``) in ``tests/test_oracle_golden.py``,
and against the reference's own known-answer tests (``src/regular_nd_grid_interpolation_test.py:12-40``,
``src/pde_test.py:12-53``).

The algorithm (file:line refer to /root/reference):
  * clip + cell index + corner gather + weights + relative coords
        src/regular_nd_grid_interpolation.py:9-78
  * multilinear interpolation                     src/regular_nd_grid_interpolation.py:81-104
  * local implicit grid query                     src/local_implicit_grid.py:10-61
  * IM-NET MLP with input re-concatenation        src/implicit_net.py:8-54
  * activations                                   src/nonlinearities.py:5-22
  * ``dif`` = one reverse sweep w/ create_graph   src/pde.py:8-9
  * PDE layer (sympy string -> residual fn)       src/pde.py:12-151
  * Rayleigh-Benard residual strings              experiments/rb2d/physics.py:6-64
  * train-step glue (losses, backward)            experiments/rb2d/train.py:58-77
"""
import itertools
import math

import sympy
import torch
import torch.nn.functional as F
from sympy.parsing.sympy_parser import parse_expr


# ------------------------------------------------------------------------------------------------
# a1/a2/a3: regular n-d grid interpolation (src/regular_nd_grid_interpolation.py)
# ------------------------------------------------------------------------------------------------
def _bounds(lo, hi, dim, device):
    """xmin/xmax normalisation (reference :40-45): scalars broadcast, sequences become tensors."""
    if isinstance(lo, (int, float)) or isinstance(hi, (int, float)):
        lo = torch.full([dim], float(lo), dtype=torch.float32, device=device)
        hi = torch.full([dim], float(hi), dtype=torch.float32, device=device)
    elif not torch.is_tensor(lo) or not torch.is_tensor(hi):
        lo = torch.as_tensor(lo).to(device)
        hi = torch.as_tensor(hi).to(device)
    return lo, hi


def corner_table(dim):
    """Rows of {0,1}^dim in C order, first dim most significant (reference :55-56)."""
    return list(itertools.product((0, 1), repeat=dim))


def interp_coefficients(grid, pts, xmin=0.0, xmax=1.0):
    """Restates ``regular_nd_grid_interpolation_coefficients`` (reference :14-78).

    grid [b, n1..nd, c]; pts [b, p, d].  Returns corner_values [b,p,2^d,c], weights [b,p,2^d],
    x_relative [b,p,2^d,d].  Same fp32 operation order as the reference: clip with eps=1e-6*(hi-lo)
    (:48-49), cubesize=(hi-lo)/(size-1) (:51), ind0=floor(q/cubesize) (:52) -- note: no xmin offset,
    quirk a-Q1 --, pos=(ind0+bit)*cubesize (:69-70), w=prod|q-pos_opposite|/cubesize (:74-75),
    x_rel=(q-pos)/cubesize (:76).
    """
    dim = grid.dim() - 2
    dev = grid.device
    size = torch.tensor(grid.shape[1:-1], device=dev).float()
    lo, hi = _bounds(xmin, xmax, dim, dev)
    eps = 1e-6 * (hi - lo)
    q = torch.max(torch.min(pts, hi - eps), lo + eps)
    cube = (hi - lo) / (size - 1)
    i0 = torch.floor(q / cube).long()
    i0f = i0.float()
    lo_pos = i0f * cube
    hi_pos = (i0f + 1) * cube
    b = grid.shape[0]
    bidx = torch.arange(b, device=dev).view(b, 1).expand(b, pts.shape[1])
    vals, wts, rels = [], [], []
    for bits in corner_table(dim):
        idx = tuple(i0[..., k] + bits[k] for k in range(dim))
        vals.append(grid[(bidx,) + idx])                                     # [b,p,c]
        pos = torch.stack([hi_pos[..., k] if bits[k] else lo_pos[..., k] for k in range(dim)], -1)
        opp = torch.stack([lo_pos[..., k] if bits[k] else hi_pos[..., k] for k in range(dim)], -1)
        wts.append(torch.prod(torch.abs(q - opp) / cube, dim=-1))
        rels.append((q - pos) / cube)
    return torch.stack(vals, 2), torch.stack(wts, 2), torch.stack(rels, 2)


def interp(grid, pts, xmin=0.0, xmax=1.0):
    """Restates ``regular_nd_grid_interpolation`` (reference :81-104)."""
    v, w, _ = interp_coefficients(grid, pts, xmin, xmax)
    return torch.sum(v * w.unsqueeze(-1), dim=-2)


# ------------------------------------------------------------------------------------------------
# a5/a6: IM-NET + activations (src/implicit_net.py, src/nonlinearities.py)
# ------------------------------------------------------------------------------------------------
def activation_fn(name, beta=None):
    """Functional forms of NONLINEARITIES (reference nonlinearities.py:15-22, torch defaults)."""
    if name == "tanh":
        return torch.tanh
    if name == "relu":
        return F.relu
    if name == "softplus":
        return F.softplus
    if name == "elu":
        return F.elu
    if name == "leakyrelu":
        return lambda t: F.leaky_relu(t, 0.01)
    if name == "swish":
        return lambda t: t * torch.sigmoid(beta * t)           # nonlinearities.py:11-12
    raise KeyError(name)


def imnet_forward(params, x, act):
    """IM-NET forward (reference implicit_net.py:48-54).

    params: list of 6 (weight [out,in], bias [out]) pairs; x [rows, dim+in_features].
    fc0..fc3 are followed by activation and re-concatenation of the raw input; fc4 by activation;
    fc5 is linear.
    """
    h = x
    for k in range(4):
        w, b = params[k]
        h = torch.cat([act(F.linear(h, w, b)), x], dim=-1)
    w, b = params[4]
    h = act(F.linear(h, w, b))
    w, b = params[5]
    return F.linear(h, w, b)


def imnet_layer_dims(dim, in_features, out_features, nf):
    """(fan_in, fan_out) of fc0..fc5 (reference implicit_net.py:31-36)."""
    dz = dim + in_features
    return [(dz, 16 * nf), (16 * nf + dz, 8 * nf), (8 * nf + dz, 4 * nf), (4 * nf + dz, 2 * nf),
            (2 * nf + dz, nf), (nf, out_features)]


def imnet_init(dim=3, in_features=32, out_features=4, nf=32, seed=0, dtype=torch.float32):
    """Deterministic nn.Linear-style init (uniform +-1/sqrt(fan_in)); returns list of (w, b)."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for fi, fo in imnet_layer_dims(dim, in_features, out_features, nf):
        bound = 1.0 / math.sqrt(fi)
        w = (torch.rand(fo, fi, generator=g, dtype=torch.float64) * 2 - 1) * bound
        b = (torch.rand(fo, generator=g, dtype=torch.float64) * 2 - 1) * bound
        out.append((w.to(dtype), b.to(dtype)))
    return out


# ------------------------------------------------------------------------------------------------
# a4: local implicit grid query (src/local_implicit_grid.py:10-61)
# ------------------------------------------------------------------------------------------------
def query_lig(model_fn, latent_grid, pts, xmin, xmax):
    """y = sum_j w_j * model([x_rel_j ; latent_j])  (reference :47-59)."""
    v, w, rel = interp_coefficients(latent_grid, pts, xmin, xmax)
    feat = torch.cat([rel, v], dim=-1)
    shp = feat.shape
    out = model_fn(feat.reshape(-1, shp[-1])).reshape(shp[0], shp[1], shp[2], -1)
    return torch.sum(out * w.unsqueeze(-1), dim=-2)


# ------------------------------------------------------------------------------------------------
# a7/a8: dif + PDE layer (src/pde.py)
# ------------------------------------------------------------------------------------------------
def dif(y, x):
    """One reverse sweep d(sum y)/dx keeping the graph (reference pde.py:8-9)."""
    return torch.autograd.grad(y, x, grad_outputs=torch.ones_like(y), create_graph=True,
                               allow_unused=True)[0]


class PDEOracle:
    """Restates PDELayer (reference pde.py:12-151): sympy string -> lambdified residual with autograd ``dif``."""

    def __init__(self, in_vars, out_vars):
        iv, ov = sympy.symbols(in_vars), sympy.symbols(out_vars)
        self.in_vars = iv if isinstance(iv, tuple) else (iv,)
        self.out_vars = ov if isinstance(ov, tuple) else (ov,)
        self.n_in, self.n_out = len(self.in_vars), len(self.out_vars)
        self.fns = {}
        self.forward_method = None

    def add_equation(self, eqn_str, name, subs_dict=None):
        e = parse_expr(eqn_str)
        for k, v in (subs_dict or {}).items():                    # sequential subs, pde.py:70-72
            e = e.subs(k, v)
        if not e.free_symbols <= set(self.in_vars) | set(self.out_vars):
            raise ValueError("unknown symbols in equation")
        self.fns[name] = sympy.lambdify(list(self.in_vars) + list(self.out_vars), e, {"dif": dif})

    def __call__(self, x, return_residue=True):
        if not return_residue:
            return self.forward_method(x)
        cols = [x[..., i:i + 1] for i in range(x.shape[-1])]     # pde.py:131-135
        for c in cols:
            if not c.requires_grad:
                c.requires_grad = True
        y = self.forward_method(torch.cat(cols, dim=-1))
        outs = [y[..., i:i + 1] for i in range(y.shape[-1])]
        return y, {k: fn(*(cols + outs)) for k, fn in self.fns.items()}


def rb2_equations(mean=None, std=None, t_crop=2., z_crop=1., x_crop=2., prandtl=1., rayleigh=1e6,
                  use_continuity=False):
    """Rayleigh-Benard residual strings + normalisation substitutions (reference physics.py:19-54).

    Returns (in_vars, out_vars, [(name, eqn_str)], subs_dict|None).
    """
    P = (rayleigh * prandtl) ** (-1 / 2)
    R = (rayleigh / prandtl) ** (-1 / 2)
    nt, nz, nx = 1. / t_crop, 1. / z_crop, 1. / x_crop

    def transport(q, coef, extra):
        return (f"{nt}*dif({q},t)-{coef}*(({nx})**2*dif(dif({q},x),x)+({nz})**2*dif(dif({q},z),z))"
                f"{extra}+(u*{nx}*dif({q},x)+w*{nz}*dif({q},z))")

    eqs = [("transport_eqn_b", transport("b", P, "")),
           ("transport_eqn_u", transport("u", R, "+dif(p,x)")),
           ("transport_eqn_w", transport("w", R, "+dif(p,z)-b"))]
    if use_continuity:
        eqs.append(("continuity", f"{nx} * dif(u, x) + {nz} * dif(w, z)"))
    subs = None
    if mean is not None or std is not None:
        subs = {v: f"{v}*{std[i]}+{mean[i]}" for i, v in enumerate(["p", "b", "u", "w"])}
    return "t, x, z", "p, b, u, w", eqs, subs


def rb2_oracle(**kw):
    iv, ov, eqs, subs = rb2_equations(**kw)
    layer = PDEOracle(iv, ov)
    for name, s in eqs:
        layer.add_equation(s, name, subs)
    return layer


# ------------------------------------------------------------------------------------------------
# a11: train-step glue (experiments/rb2d/train.py:58-77) on a given latent grid
# ------------------------------------------------------------------------------------------------
def lig_pde_step(params, act_name, latent_grid, pts, targets, pde, alpha_reg=1.0, alpha_pde=0.0125,
                 xmin=0.0, xmax=1.0, beta=None, loss="l1", backward=True):
    """pred + residuals + losses (+ backward to params and latent_grid).

    latent_grid [b, n_t, n_z, n_x, c] (channels-last view as train.py:60); pts [b,p,3]; targets [b,p,o].
    Returns dict(pred, residues{name:[b,p,1]}, reg_loss, pde_loss, loss, grads(list aligned w/ params,
    flattened w0,b0,w1,b1..), dlatent).
    """
    lossf = {"l1": F.l1_loss, "l2": F.mse_loss, "huber": F.smooth_l1_loss}[loss]
    act = activation_fn(act_name, beta)
    leaves = []
    plist = []
    for w, b in params:
        w = w.detach().clone().requires_grad_(backward)
        b = b.detach().clone().requires_grad_(backward)
        plist.append((w, b))
        leaves += [w, b]
    lat = latent_grid.detach().clone().requires_grad_(backward)
    pde.forward_method = lambda p: query_lig(lambda f: imnet_forward(plist, f, act), lat, p, xmin, xmax)
    pred, res = pde(pts.detach().clone(), return_residue=True)
    reg = lossf(pred, targets)
    stack = torch.stack(list(res.values()), dim=0)
    pl = lossf(stack, torch.zeros_like(stack))
    total = alpha_reg * reg + alpha_pde * pl
    out = dict(pred=pred.detach(), residues={k: v.detach() for k, v in res.items()},
               reg_loss=reg.detach(), pde_loss=pl.detach(), loss=total.detach())
    if backward:
        total.backward()
        out["grads"] = [t.grad for t in leaves]
        out["dlatent"] = lat.grad
    return out
