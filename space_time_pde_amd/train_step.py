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
import torch
import torch.distributed as dist
import torch.nn.functional as F

from .local_implicit_grid import query_local_implicit_grid


class _SumGradAcrossRanks(torch.autograd.Function):
    """Identity in forward; all-reduce(sum) of the gradient in backward."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        dist.all_reduce(g)
        return g


_LOSS_SUMS = {
    "l1": lambda a, b: (a - b).abs().sum(),
    "l2": lambda a, b: ((a - b) ** 2).sum(),
    "huber": lambda a, b: F.smooth_l1_loss(a, b, reduction="sum"),
}


def sharded_step(unet, imnet, pde_layer, input_grid, point_coord, point_value, n_points_global, alpha_reg=1.0,
                 alpha_pde=1.0, loss_type="l1", xmin=0.0, xmax=1.0, distributed=None):
    """Forward + backward of one step on this rank's slice of the query points.

    input_grid [b, c, T, Z, X] (identical on every rank); point_coord / point_value [b, n_local, 3|o] (this rank's
    slice); n_points_global = total points per batch element over all ranks.  Gradients are left in ``.grad`` of
    the UNet / IM-NET parameters, already summed over ranks.  Returns (loss, reg_loss, pde_loss) global values.
    """
    if distributed is None:
        distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    lsum = _LOSS_SUMS[loss_type]
    latent_grid = unet(input_grid).permute(0, 2, 3, 4, 1)            # train.py:58-60
    if distributed:
        latent_grid = _SumGradAcrossRanks.apply(latent_grid)
    pde_layer.update_forward_method(lambda pts: query_local_implicit_grid(imnet, latent_grid, pts, xmin, xmax))
    pred, residues = pde_layer(point_coord, return_residue=True)    # train.py:66-67
    b = point_coord.shape[0]
    reg = lsum(pred, point_value) / (b * n_points_global * pred.shape[-1])
    stack = torch.stack(list(residues.values()), dim=0)
    pde = lsum(stack, torch.zeros_like(stack)) / (stack.shape[0] * b * n_points_global)
    loss = alpha_reg * reg + alpha_pde * pde
    loss.backward()
    if distributed:
        grads = [p.grad for p in imnet.parameters() if p.grad is not None]
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat)
        o = 0
        for g in grads:
            g.copy_(flat[o:o + g.numel()].view_as(g))
            o += g.numel()
        stats = torch.stack([loss.detach(), reg.detach(), pde.detach()])
        dist.all_reduce(stats)
        return stats[0], stats[1], stats[2]
    return loss.detach(), reg.detach(), pde.detach()
