"""Local implicit grid query (mirrors src/local_implicit_grid.py:10-61 of the reference).

``query_local_implicit_grid(model, latent_grid, query_pts, xmin, xmax)`` keeps the reference signature.  When it
is called from inside ``PDELayer.__call__`` (which announces the derivatives its equations need through
``jet_context``) with an ``ImNet`` decoder on CUDA tensors and 3-d query points, the whole
gather -> MLP -> corner-weighted sum, INCLUDING the coordinate derivatives, runs in the HIP jet kernels
(lig_jet.py).  A plain call (no PDE layer) on the same kind of inputs runs the value-only HIP path.
Other decoders / dimensions use the generic composed formulation.
"""
import threading

import torch

from . import lig_jet
from . import regular_nd_grid_interpolation as rgi
from .implicit_net import ImNet

_tls = threading.local()
stats = {"hip_jet_calls": 0, "hip_value_calls": 0, "generic_calls": 0,
         # slower-strategy counters of pde.py (each also warns once): sympy analysis failures and residuals evaluated with
         # the lambdified torch functions instead of the HIP residual program
         "combo_plan_errors": 0, "jet_compile_errors": 0, "residual_torch_fallbacks": 0}


class JetRequest:
    """Set by PDELayer around ``forward_method``: which derivatives of y w.r.t. the points it will need."""

    def __init__(self, x, first, pairs, combo=None):
        self.x, self.first, self.pairs = x, first, list(pairs)
        self.combo = combo     # {pair: alpha}: the equations use second derivatives only through this combination
        self.y = None          # the tensor handed back to the caller
        self.jets = None       # [S, n_out, P]
        self.pairs_out = None  # second-order pairs in stream order (after padding)


class jet_context:
    def __init__(self, req):
        self.req = req

    def __enter__(self):
        self.prev = getattr(_tls, "req", None)
        _tls.req = self.req
        return self.req

    def __exit__(self, *exc):
        _tls.req = self.prev
        return False


def _fast_eligible(model, latent_grid, query_pts):
    return (isinstance(model, ImNet) and model.dim == 3 and latent_grid.dim() == 5 and query_pts.dim() == 3
            and query_pts.shape[-1] == 3 and latent_grid.is_cuda and query_pts.is_cuda
            and latent_grid.dtype == torch.float32 and query_pts.dtype == torch.float32
            and model.nf % 16 == 0 and model.out_features <= 16 and model.in_features <= lig_jet.MAX_LATENT_CHANNELS
            and latent_grid.shape[-1] == model.in_features
            and lig_jet.activation_name(model.activ) is not None)


def _xmin_is_zero(xmin, xmax=None, shape3=None):
    """The HIP kernels reproduce the reference's cell index, which ignores xmin (quirk a-Q1), for xmin == 0 only; any
    other lower bound takes the generic formulation, which behaves like the reference on every device.  Tensor bounds
    are decided through the (cached, sync-free after the first call) box-constant evaluation."""
    if isinstance(xmin, (int, float)):
        return xmin == 0
    if torch.is_tensor(xmin):
        if torch.is_tensor(xmax) and shape3 is not None:
            try:
                lig_jet.cached_box_constants(shape3, xmin, xmax)
                return True
            except ValueError:
                return False
        return bool((xmin.detach() == 0).all())
    return all(float(v) == 0 for v in xmin)


def _query_data_parallel(dp, latent_grid, query_pts, xmin, xmax, req):
    """The reference's default multi-GPU mechanism (experiments/rb2d/train.py:352-355 wraps the decoder in nn.DataParallel):
    the rows of the decoder input are scattered over ``dp.device_ids``.  Here the QUERY POINTS are: every device gets a
    replica of the IM-NET (``torch.nn.parallel.replicate``: gradients flow back to the wrapped module, summed), a copy of the
    latent grid and its share of the points, and runs the HIP jet path on them; the jets are gathered on the first device.
    One process drives all devices, as nn.DataParallel does.  Costs that come with that mechanism, per CALL: the module is
    re-replicated, the full latent grid is copied to every device and the devices are driven one after another from this
    thread -- it exists so that a reference script with several ``device_ids`` keeps the HIP path instead of dropping to the
    composed formulation, not for speed.  ``train_step.sharded_step`` (one process per GPU, collectives over RCCL) is the
    supported multi-GPU route.  Hardware coverage: on the 1-GPU test box both "devices" are ``cuda:0``
    (tests/test_gpu_lig_jet.py); with two real devices the path has not been run by the builder."""
    devs = list(dp.device_ids)
    B, N = query_pts.shape[0], query_pts.shape[1]
    chunks = [c for c in torch.chunk(query_pts, len(devs), dim=1) if c.shape[1] > 0]
    devs = devs[:len(chunks)]
    replicas = torch.nn.parallel.replicate(dp.module, devs, detach=not torch.is_grad_enabled())
    want_jets = req is not None and req.x is query_pts
    outs, pairs = [], None
    for rep, dev, pts_c in zip(replicas, devs, chunks):
        d = torch.device("cuda", dev)
        with torch.cuda.device(d):
            lat_d, pts_d = latent_grid.to(d), pts_c.to(d).contiguous()
            if want_jets:
                jets, pairs = lig_jet.lig_jets(rep, lat_d, pts_d, xmin, xmax, req.first, req.pairs, combo=req.combo)
            else:
                jets, _ = lig_jet.lig_jets(rep, lat_d, pts_d, xmin, xmax, False, ())
        # [S, n_out, B * n_d] (batch-major) -> [S, n_out, B, n_d] on the output device
        outs.append(jets.reshape(jets.shape[0], jets.shape[1], B, pts_c.shape[1]).to(query_pts.device))
    jets = torch.cat(outs, dim=3).reshape(outs[0].shape[0], outs[0].shape[1], B * N)
    stats["data_parallel_calls"] = stats.get("data_parallel_calls", 0) + 1
    y = jets[0].t().reshape(B, N, jets.shape[1])
    if want_jets:
        stats["hip_jet_calls"] += 1
        req.y, req.jets, req.pairs_out = y, jets, pairs
    else:
        stats["hip_value_calls"] += 1
    return y


def query_local_implicit_grid(model, latent_grid, query_pts, xmin, xmax):
    """Query a local implicit grid: y = sum_j w_j * model([x_rel_j ; latent_j]) (reference :47-59).

    model: nn.Module taking [rows, d+c]; latent_grid [b, n1..nd, c]; query_pts [b, num_pts, d];
    xmin/xmax: float, sequence or tensor bounds of the grid.  Returns [b, num_pts, o].
    """
    req = getattr(_tls, "req", None)
    # the reference wraps its networks in nn.DataParallel (experiments/rb2d/train.py:352-355); on one device that
    # wrapper is the identity, so the HIP path reads the wrapped module's parameters directly
    if isinstance(model, torch.nn.DataParallel) and len(model.device_ids or []) <= 1:
        model = model.module
    if isinstance(model, torch.nn.DataParallel) and _fast_eligible(model.module, latent_grid, query_pts) \
            and _xmin_is_zero(xmin, xmax, tuple(latent_grid.shape[1:4])) \
            and ((req is not None and req.x is query_pts) or not (query_pts.requires_grad and torch.is_grad_enabled())):
        return _query_data_parallel(model, latent_grid, query_pts, xmin, xmax, req)
    if _fast_eligible(model, latent_grid, query_pts) and _xmin_is_zero(xmin, xmax, tuple(latent_grid.shape[1:4])):
        wants_point_grad = query_pts.requires_grad and torch.is_grad_enabled()
        if req is not None and req.x is query_pts:
            jets, pairs = lig_jet.lig_jets(model, latent_grid, query_pts, xmin, xmax, req.first, req.pairs,
                                           combo=req.combo)
            stats["hip_jet_calls"] += 1
            y = jets[0].t().reshape(query_pts.shape[0], query_pts.shape[1], jets.shape[1])
            req.y, req.jets, req.pairs_out = y, jets, pairs
            return y
        if not wants_point_grad:
            jets, _ = lig_jet.lig_jets(model, latent_grid, query_pts, xmin, xmax, False, ())
            stats["hip_value_calls"] += 1
            return jets[0].t().reshape(query_pts.shape[0], query_pts.shape[1], jets.shape[1])
    stats["generic_calls"] += 1
    corner_values, weights, x_relative = rgi._coefficients_autograd(latent_grid, query_pts, xmin, xmax) \
        if (query_pts.requires_grad and torch.is_grad_enabled()) else \
        rgi.regular_nd_grid_interpolation_coefficients(latent_grid, query_pts, xmin, xmax)
    feats = torch.cat([x_relative, corner_values], dim=-1)
    shp = feats.shape
    out = model(feats.reshape(-1, shp[-1])).reshape(shp[0], shp[1], shp[2], -1)
    return torch.sum(out * weights.unsqueeze(-1), dim=-2)
