"""Dense-lattice inference: the query loop of experiments/rb2d/evaluation.py:26-74.

``evaluate_feat_grid`` keeps the reference signature and result layout.  Each pseudo-batch is one forward pass of the
HIP jet kernels (value + the residual derivatives, no autograd graph, no stash kept), so the pseudo-batch can be
millions of points instead of the reference's 10^4 (its higher-order autograd graph is O(points) memory).
"""
from collections import defaultdict

import numpy as np
import torch


def evaluate_feat_grid(pde_layer, latent_grid, t_seq, z_seq, x_seq, mins=None, maxs=None, pseudo_batch_size=1 << 20,
                       phys_channels=("p", "b", "u", "w")):
    """Evaluate the (already updated) pde_layer forward method on the lattice t_seq x z_seq x x_seq.

    Returns {channel or equation name: ndarray [len(t_seq), len(z_seq), len(x_seq)]} for batch element 0, exactly like
    the reference.  ``mins`` / ``maxs`` are accepted for signature compatibility (the reference does not use them).
    """
    device = latent_grid.device
    nb = latent_grid.shape[0]
    coord = torch.stack(torch.meshgrid(t_seq, z_seq, x_seq, indexing="ij"), dim=-1).reshape(-1, 3).to(device)
    n_query = coord.shape[0]
    res = defaultdict(list)
    for sid in range(0, n_query, pseudo_batch_size):
        batch = coord[sid:sid + pseudo_batch_size][None].expand(nb, -1, 3).contiguous()
        # HIP jet path: derivatives come from forward-mode streams, no autograd graph (and no stash) is needed.
        # Generic strategy (other forward methods / CPU): the reference's dif() sweeps need grad mode.
        jet = getattr(pde_layer, "_jet_request", lambda x: None)(batch) is not None
        with (torch.no_grad() if jet else torch.enable_grad()):
            pred, residues = pde_layer(batch, return_residue=True)
            pred = pred.detach().cpu().numpy()
        for cid, name in enumerate(phys_channels):
            res[name].append(pred[..., cid])
        for name, val in residues.items():
            res[name].append(val.detach().cpu().numpy()[..., 0])
    shape = [nb, len(t_seq), len(z_seq), len(x_seq)]
    return {k: np.concatenate(v, axis=1).reshape(shape)[0] for k, v in res.items()}
