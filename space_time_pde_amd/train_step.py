"""One MeshfreeFlowNet training step, optionally sharded over the query points (one process per GPU).

Reproduces experiments/rb2d/train.py:58-77 of the reference (UNet -> permute -> local-implicit-grid query through the
PDE layer -> regression + PDE-residual losses -> backward).  Multi-GPU semantics follow the reference's data
parallelism (experiments/rb2d/train_ddp.py:401-406: gradients averaged over ranks) but partition what actually shards
for free on this path -- the query points: every rank holds the same crop and the same weights, evaluates its slice
of the points, and
  * d(loss)/d(latent grid) partial sums are all-reduced (RCCL over xGMI on GPU, gloo on CPU) before the UNet backward,
    which is therefore replicated and needs no parameter all-reduce;
  * the IM-NET gradients (0.84 MB) are all-reduced after backward.
Losses are normalised by the GLOBAL element counts, so the sharded sum equals the single-process mean.
"""
import os

import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import _lib, lig_jet
from .local_implicit_grid import query_local_implicit_grid


# [(what, bytes)] of every collective the most recent distributed step issued, in issue order (bench.py prints them next to
# the per-rank compute time so that a multi-GPU line can be diagnosed: VERDICT r5 next #8)
last_collectives = []


class _SumGradAcrossRanks(torch.autograd.Function):
    """Identity in forward; all-reduce(sum) of the gradient in backward."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        hooks = lig_jet.sync_hooks
        done = hooks.get("dlatent_done") if hooks else None
        if done is None:
            g = g.contiguous()
            last_collectives.append(("d(loss)/d(latent grid), autograd node", g.numel() * g.element_size()))
            dist.all_reduce(g)
            return g
        # ``done`` is the tensor the HIP backward has already summed over ranks (behind the IM-NET weight gradients).  With
        # ONE consumer of the latent grid in the graph -- what sharded_step builds -- the incoming gradient IS that tensor
        # and passes through.  A second consumer (another query, a value query next to a jet query, a partial generic
        # fallback) makes autograd hand over reduced + local contributions: only the local remainder still has to be summed.
        if g.data_ptr() == done.data_ptr() and g.shape == done.shape and g.stride() == done.stride():
            return g
        rest = (g - done.view_as(g)).contiguous()
        dist.all_reduce(rest)
        return rest + done.view_as(g)


_LOSS_SUMS = {
    "l1": lambda a, b: (a - b).abs().sum(),
    "l2": lambda a, b: ((a - b) ** 2).sum(),
    "huber": lambda a, b: F.smooth_l1_loss(a, b, reduction="sum"),
}
_LOSS_KIND = {"l1": 0, "l2": 1, "huber": 2}


class _LossSumHip(torch.autograd.Function):
    """sum_i f(a_i - b_i) (b None: against zero) in one HIP pass; backward = one elementwise pass (stpde_loss_sum /
    stpde_loss_grad).  Replaces sub + abs + sum (+ zeros_like, stack) and their backward kernels of train.py:69-76."""

    @staticmethod
    @_lib.guarded
    def forward(ctx, a, b, kind):
        L = _lib.lib()
        a = a.contiguous()
        b = b.contiguous() if b is not None else None
        if _lib.deterministic:      # block sums into one long accumulator (integer atomics), then fp32: bit-reproducible
            acc = torch.zeros(2 * _lib.DET_K, device=a.device, dtype=torch.float32)
            _lib.check(L.stpde_loss_sum(kind | _lib.LOSS_DET, a.numel(), _lib.ptr(a), _lib.ptr(b), _lib.ptr(acc), _lib.stream_ptr()))
            out = torch.empty((), device=a.device, dtype=torch.float32)
            _lib.check(L.stpde_det_finalize(_lib.ptr(acc), 1, _lib.ptr(out), _lib.stream_ptr()))
        else:
            out = torch.zeros((), device=a.device, dtype=torch.float32)
            _lib.check(L.stpde_loss_sum(kind, a.numel(), _lib.ptr(a), _lib.ptr(b), _lib.ptr(out), _lib.stream_ptr()))
        ctx.save_for_backward(a, b)
        ctx.kind = kind
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    @_lib.guarded
    def backward(ctx, g):
        L = _lib.lib()
        a, b = ctx.saved_tensors
        ga = torch.empty_like(a)
        _lib.check(L.stpde_loss_grad(ctx.kind, a.numel(), _lib.ptr(a), _lib.ptr(b), _lib.ptr(g.contiguous().float()),
                                     _lib.ptr(ga), _lib.stream_ptr()))
        return ga, None, None


def loss_sum(a, b, loss_type):
    """Sum-reduced loss of a against b (None = zero): HIP kernels on CUDA fp32 tensors, torch ops otherwise."""
    if a.is_cuda and a.dtype == torch.float32 and (b is None or (b.dtype == torch.float32 and b.shape == a.shape)) \
            and not (b is not None and b.requires_grad):
        return _LossSumHip.apply(a, b, _LOSS_KIND[loss_type])
    return _LOSS_SUMS[loss_type](a, torch.zeros_like(a) if b is None else b)


def sharded_step(unet, imnet, pde_layer, input_grid, point_coord, point_value, n_points_global, alpha_reg=1.0,
                 alpha_pde=1.0, loss_type="l1", xmin=0.0, xmax=1.0, distributed=None, sync_unet_grads=False):
    """Forward + backward of one step on this rank's slice of the query points.

    input_grid [b, c, T, Z, X] (identical on every rank); point_coord / point_value [b, n_local, 3|o] (this rank's
    slice); n_points_global = total points per batch element over all ranks.  Gradients are left in ``.grad`` of
    the UNet / IM-NET parameters, already summed over ranks.  Returns (loss, reg_loss, pde_loss) global values.
    sync_unet_grads: the replicated UNet backward uses fp32 atomics, so its gradients agree across ranks only to
    rounding (~1e-7 relative); True averages them with one more all-reduce so that long runs keep the replicas
    bit-identical (the reference's DDP all-reduces every gradient, train_ddp.py:401-406).
    """
    if distributed is None:
        distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if loss_type not in _LOSS_SUMS:
        raise KeyError(loss_type)
    # this step calls loss.backward() itself, so the U-Net's weight gradients may run beside its input-gradient chain
    # (unet3d._DeferredGrads, opt-in); the caller's setting of the flag is restored when the step is over
    prev_deferred = getattr(unet, "deferred_weight_grads", None)
    if prev_deferred is not None:
        unet.deferred_weight_grads = os.environ.get("STPDE_UNET_DEFERRED", "1") != "0"
    try:
        return _sharded_step(unet, imnet, pde_layer, input_grid, point_coord, point_value, n_points_global, alpha_reg,
                             alpha_pde, loss_type, xmin, xmax, distributed, sync_unet_grads)
    finally:
        if prev_deferred is not None:
            unet.deferred_weight_grads = prev_deferred


def _sharded_step(unet, imnet, pde_layer, input_grid, point_coord, point_value, n_points_global, alpha_reg, alpha_pde,
                  loss_type, xmin, xmax, distributed, sync_unet_grads):
    latent_grid = unet(input_grid).permute(0, 2, 3, 4, 1)            # train.py:58-60
    if distributed:
        latent_grid = _SumGradAcrossRanks.apply(latent_grid)
    pde_layer.update_forward_method(lambda pts: query_local_implicit_grid(imnet, latent_grid, pts, xmin, xmax))
    # collectives issued from inside the HIP backward: d latent asynchronously behind the IM-NET weight gradients, the
    # IM-NET gradients in place in their flat buffer (no cat / copy-back); both are waited for before backward returns
    overlap_sync = distributed and os.environ.get("STPDE_OVERLAP_SYNC", "1") != "0"
    # (the forward's memory plan must know NOW that the backward will take the dgrad-first two-phase order)
    lig_jet.expect_two_phase = bool(overlap_sync)
    try:
        pred, residues = pde_layer(point_coord, return_residue=True)    # train.py:66-67
    finally:
        lig_jet.expect_two_phase = False
    b = point_coord.shape[0]
    reg = loss_sum(pred, point_value, loss_type) / (b * n_points_global * pred.shape[-1])
    res = list(residues.values())
    # the residuals of the HIP evaluator are rows of ONE [n_eq, P] tensor: reduce it in place of a stacked copy
    base = res[0]._base if all(r._base is not None and r._base is res[0]._base for r in res) else None
    stack = base if (base is not None and base.numel() == sum(r.numel() for r in res)) else torch.stack(res, dim=0)
    pde = loss_sum(stack, None, loss_type) / (len(res) * b * n_points_global)
    loss = alpha_reg * reg + alpha_pde * pde
    hooks = {}
    del last_collectives[:]

    def _all_reduce_async(name):
        def f(t):
            last_collectives.append((name, t.numel() * t.element_size()))
            return dist.all_reduce(t, async_op=True)
        return f

    if overlap_sync:
        hooks.update(dlatent=_all_reduce_async("d(loss)/d(latent grid), from inside the HIP backward"),
                     dw=_all_reduce_async("IM-NET gradients (flat dW buffer, in place)"))
    hooks = hooks or None
    lig_jet.sync_hooks = hooks
    try:
        loss.backward()
    finally:
        lig_jet.sync_hooks = None
    if distributed and not (hooks and hooks.get("dw_done")):
        grads = [p.grad for p in imnet.parameters() if p.grad is not None]
        flat = torch.cat([g.reshape(-1) for g in grads])
        last_collectives.append(("IM-NET gradients (concatenated .grad)", flat.numel() * 4))
        dist.all_reduce(flat)
        o = 0
        for g in grads:
            g.copy_(flat[o:o + g.numel()].view_as(g))
            o += g.numel()
    if distributed:
        if sync_unet_grads:
            ug = [p.grad for p in unet.parameters() if p.grad is not None]
            uflat = torch.cat([g.reshape(-1) for g in ug])
            last_collectives.append(("U-Net gradients (sync_unet_grads)", uflat.numel() * 4))
            dist.all_reduce(uflat)
            uflat /= dist.get_world_size()
            o = 0
            for g in ug:
                g.copy_(uflat[o:o + g.numel()].view_as(g))
                o += g.numel()
        stats = torch.stack([loss.detach(), reg.detach(), pde.detach()])
        last_collectives.append(("loss statistics (loss, reg, pde)", stats.numel() * 4))
        dist.all_reduce(stats)
        return stats[0], stats[1], stats[2]
    return loss.detach(), reg.detach(), pde.detach()


def data_parallel_step(unet, imnet, pde_layer, input_grid, point_coord, point_value, alpha_reg=1.0, alpha_pde=1.0,
                       loss_type="l1", xmin=0.0, xmax=1.0):
    """The reference's own split (experiments/rb2d/train_ddp.py:361-368, 401-406): every rank holds DIFFERENT crops
    (and their query points), runs the whole step locally, and all parameter gradients are averaged over ranks with
    one flat all-reduce.  BatchNorm statistics stay per rank, as under the reference's DistributedDataParallel."""
    loss, reg, pde = sharded_step(unet, imnet, pde_layer, input_grid, point_coord, point_value, point_coord.shape[1],
                                  alpha_reg, alpha_pde, loss_type, xmin, xmax, distributed=False)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        world = dist.get_world_size()
        grads = [p.grad for m in (unet, imnet) for p in m.parameters() if p.grad is not None]
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat)
        flat /= world
        o = 0
        for g in grads:
            g.copy_(flat[o:o + g.numel()].view_as(g))
            o += g.numel()
        stats = torch.stack([loss, reg, pde])
        dist.all_reduce(stats)
        stats /= world
        return stats[0], stats[1], stats[2]
    return loss, reg, pde


class GraphedStep:
    """One training step (``sharded_step`` on fixed shapes) captured ONCE in a HIP graph and replayed.

    Why: with the reference's own training regime (``experiments/rb2d/run_experiment.sh:16``: 10 crops x 512 points on the
    (4,16,16) latent grid) a step is ~500 kernel dispatches of a few microseconds each -- the host cannot enqueue them as fast
    as the device retires them, and the step is bound by launch latency, not by any kernel (VERDICT r5 missing #2).  A HIP
    graph replays the whole dispatch sequence (U-Net forward, gather, IM-NET layers, residuals, losses, the whole backward
    including the deferred weight-gradient side stream, which joins the capture through its events) with one host call.

    Everything the step launches goes to torch's current stream, which is the capturing stream during capture: the library's
    kernels are launched through ``hipLaunchKernelGGL`` on that stream and are captured like torch's own.  The library never
    allocates device memory or synchronises (include/stpde_hip.h), so nothing in it is illegal under capture; host-side
    decisions of the step (memory plan, kernel variants) depend on shapes only and are frozen at capture.

    Inputs are copied into static buffers before each replay; ``.grad`` of every parameter is a static tensor of the graph's
    memory pool, rewritten by each replay (the optimizer step runs outside the graph, on those tensors).  Not for a
    distributed step (collectives are not captured here): ``distributed`` is forced off.

        step = GraphedStep(unet, imnet, layer, crop, pts, tgt, n_points_global)
        loss, reg, pde = step(crop2, pts2, tgt2)        # device scalars, valid until the next replay
    """

    def __init__(self, unet, imnet, pde_layer, input_grid, point_coord, point_value, n_points_global, alpha_reg=1.0,
                 alpha_pde=1.0, loss_type="l1", xmin=0.0, xmax=1.0, warmup=2):
        if not input_grid.is_cuda:
            raise RuntimeError("GraphedStep needs CUDA/HIP tensors")
        self.params = [p for m in (unet, imnet) for p in m.parameters()]
        self.static = [t.detach().clone() for t in (input_grid, point_coord, point_value)]
        args = (unet, imnet, pde_layer) + tuple(self.static) + (n_points_global, alpha_reg, alpha_pde, loss_type, xmin, xmax)

        def run():
            for p in self.params:
                p.grad = None
            out = sharded_step(*args, distributed=False)
            # the layer's forward method closes over this step's latent grid, i.e. over its whole autograd graph: dropped, or
            # the graph -- and the parameters' AccumulateGrad nodes with it -- outlives the step
            pde_layer.forward_method = None
            return out

        # Steps that ran BEFORE this constructor on the default stream leave exactly that behind (pde_layer.forward_method ->
        # latent grid -> U-Net graph -> AccumulateGrad nodes bound to the default stream); the autograd engine would then make the
        # capture stream wait on the default stream, which ends the capture in a crash inside hipStreamEndCapture (seen: bench.py
        # after eager steps).  Drop that graph so that the warm-up below re-creates the nodes on its side stream.
        import gc
        pde_layer.forward_method = None
        gc.collect()

        # lazy initialisation (kernel modules, function attributes, cached index tables, sympy lambdas) must not happen inside
        # the capture: a few eager steps on a side stream first (torch's recipe for whole-network capture)
        # (the AccumulateGrad nodes of the parameters were created on whatever stream ran their first eager step; the warm-up /
        # capture streams differ from it by design -- torch's warning about that is not an error here)
        if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream(device=input_grid.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                run()
        cur.wait_stream(side)
        torch.cuda.synchronize(input_grid.device)
        for p in self.params:
            p.grad = None
        self.graph = torch.cuda.CUDAGraph()
        # "relaxed": the step's host code queries free memory for its memory plan (hipMemGetInfo), which the default capture
        # mode refuses although it touches no stream
        with torch.cuda.graph(self.graph, capture_error_mode="relaxed"):
            self.out = run()
        self.replays = 0

    def __call__(self, input_grid=None, point_coord=None, point_value=None):
        for dst, src in zip(self.static, (input_grid, point_coord, point_value)):
            if src is not None and src.data_ptr() != dst.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        self.replays += 1
        return self.out
