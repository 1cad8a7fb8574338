"""Round-4 kernel variants of the bf16 mode against the kernels they replaced (same operands, same products):

  STPDE_WGRAD_SWAP=0      every wave of k_wgrad_coop runs its phases in the same order (round 3 schedule)
  STPDE_WGRAD_OCT_BF=0    fc2 / fc3 weight gradients through k_wgrad_quad (fp32 blocks in LDS) instead of k_wgrad_oct_bf
  STPDE_ACT16=1           the first hidden layer's weight gradient reads the forward's operand blocks (act16) instead of
                          re-evaluating the activation jets

The switches are read once per process, so every variant runs in its own interpreter on the same seeded problem (combined
second-order stream of the RB2 equations AND two separate second-order streams); the gradients must agree to fp32 summation
noise -- far below the mode's own 5e-4 / 3e-2 distance to the oracles (tests/test_gpu_lig_jet.py), which could hide a wrong
column or a dropped tile.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys, numpy as np, torch
from space_time_pde_amd import implicit_net, lig_jet, local_implicit_grid as lig, physics, _lib
out, combo = sys.argv[1], sys.argv[2] == "1"
dev = torch.device("cuda:0")
torch.manual_seed(3)
net = implicit_net.ImNet(nf=32, activation=torch.nn.Softplus).to(dev)
lat = (0.5 * torch.randn(1, 6, 10, 12, 32)).to(dev).requires_grad_(True)
pts = (0.02 + 0.96 * torch.rand(1, 6000, 3)).to(dev)
res = {}
with _lib.dispatch_trace() as tr:
    if combo:      # RB2 equations: one combined second-order stream, S = 5
        kw = dict(mean=(0.01, 0.0, 0.02, -0.01), std=(0.05, 0.3, 0.15, 0.12), t_crop=2., z_crop=1., x_crop=1., use_continuity=True)
        layer = physics.get_rb2_pde_layer(**kw)
        lig_jet.set_mlp_precision("bf16")
        layer.update_forward_method(lambda p: lig.query_local_implicit_grid(net, lat, p, 0., 1.))
        pred, r = layer(pts, return_residue=True)
        loss = pred.abs().mean() + 0.05 * torch.stack(list(r.values()), 0).abs().mean()
    else:          # two separate second-order streams, S = 6
        jets, _ = lig_jet.lig_jets(net, lat, pts, 0., 1., True, ((1, 1), (2, 2)), chunk_points=2048, precision="bf16")
        g = torch.Generator().manual_seed(5)
        loss = (jets * torch.randn(jets.shape, generator=g).to(dev)).sum()
    loss.backward()
torch.cuda.synchronize()
res["loss"] = loss.detach().cpu().numpy()
res["dlat"] = lat.grad.cpu().numpy()
for k in range(6):
    res["dw%d" % k] = net.fc[k].weight.grad.cpu().numpy()
    res["db%d" % k] = net.fc[k].bias.grad.cpu().numpy()
res["kernels"] = np.array("\n".join(tr.kernels))
np.savez(out, **res)
'''


def _run(tmp_path, tag, combo, **env):
    out = str(tmp_path / ("%s_%d.npz" % (tag, combo)))
    e = dict(os.environ, PYTHONPATH=ROOT, **env)
    p = subprocess.run([sys.executable, "-c", SCRIPT, out, "1" if combo else "0"], env=e, cwd=ROOT, capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return np.load(out)


@pytest.mark.parametrize("combo", [True, False])
def test_round4_bf16_kernels_equal_the_ones_they_replaced(hiplib, tmp_path, combo):
    # (round 5: the fused backward of the first hidden layer is the default; these variants belong to the two kernels it
    # replaced, which STPDE_FC1_FUSED=0 still selects -- test_fused_fc1_backward_equals_the_two_kernel_path ties the two together)
    base = _run(tmp_path, "base", combo, STPDE_FC1_FUSED="0")
    kern = str(base["kernels"])
    assert "k_wgrad_oct_bf" in kern and "k_fc1_fwd_spec" in kern, kern
    assert ("k_fc1_dgrad_spec" in kern) == combo, kern         # the wave-specialised input gradient is compiled for S2 <= 1
    variants = {"noswap": dict(STPDE_WGRAD_SWAP="0"), "quad": dict(STPDE_WGRAD_OCT_BF="0"), "act16": dict(STPDE_ACT16="1")}
    for tag, env in variants.items():
        other = _run(tmp_path, tag, combo, STPDE_FC1_FUSED="0", **env)
        ok = str(other["kernels"])
        if tag == "quad":
            assert "k_wgrad_oct_bf" not in ok and "k_wgrad_quad" in ok, ok
        if tag == "act16":
            assert "PKA | 8" in ok or ", 12>" in ok or "(PKA | 8)" in ok, ok     # the operand-block variant was dispatched
        assert abs(float(other["loss"]) - float(base["loss"])) <= 1e-6 * abs(float(base["loss"])), tag
        for k in base.files:
            if k in ("kernels", "loss"):
                continue
            a, b = base[k].astype(np.float64), other[k].astype(np.float64)
            err = np.abs(a - b).max() / max(np.abs(a).max(), 1e-30)
            assert err < 2e-5, (tag, k, err)


def test_fc2_forward_on_lds_dma_equals_the_cooperative_kernel(hiplib, tmp_path):
    """Round 5: k_fc2_fwd_bf (csrc/jet_spec_bf16.h -- persistent, stash tiles by LDS-DMA, launch-resident weights) against
    k_layer_coop<..., BF> (STPDE_FC2_FWD_SPEC=0) on the combined-stream set it is compiled for: same operand rounding, same
    accumulation order over the k-tile pairs, same epilogue -- loss and d latent must be bit-identical (6000 points x 8
    corners = 3000 row tiles on 256 persistent workgroups: 11-12 tiles each through the double buffer)."""
    base = _run(tmp_path, "fc2spec", True)
    other = _run(tmp_path, "fc2coop", True, STPDE_FC2_FWD_SPEC="0")
    kb, ko = str(base["kernels"]), str(other["kernels"])
    assert "k_fc2_fwd_bf" in kb and "k_fc2_fwd_bf" not in ko, kb
    # the forward pass and the deterministic d latent: bit for bit; the weight gradients behind atomics: to summation rounding
    assert np.array_equal(base["loss"], other["loss"]) and np.array_equal(base["dlat"], other["dlat"])
    for k in base.files:
        if k in ("kernels", "loss", "dlat"):
            continue
        a, b = base[k].astype(np.float64), other[k].astype(np.float64)
        assert np.abs(a - b).max() <= 2e-5 * max(np.abs(a).max(), 1e-30), k


def test_fused_fc1_backward_equals_the_two_kernel_path(hiplib, monkeypatch):
    """Round 5: k_fc1_bwd_fused (csrc/jet_fc1_bwd.hip -- input gradient + weight gradient of the first hidden layer in one
    kernel, one activation-jet evaluation per z0 element) against the two kernels it replaces (STPDE_FC1_FUSED=0:
    k_fc1_dgrad_spec + k_wgrad_coop): the layer-0 adjoint is computed in the same order (d latent and the layer-0 weight
    gradient equal to rounding), fc1's weight gradient sums the same bf16 products in another order (fp32 summation rounding).
    Both stream sets of the mode (combined second-order stream, and leaky-relu's S = 4), several row-tile counts incl. ones
    that leave workgroups without work and an odd pair count."""
    import torch
    from space_time_pde_amd import _lib, implicit_net, lig_jet
    dev = torch.device("cuda:0")
    monkeypatch.setattr(lig_jet, "mlp_precision", "bf16")
    for act, npts in ((torch.nn.Softplus, 4096), (torch.nn.LeakyReLU, 1024), (torch.nn.Softplus, 6), (torch.nn.Tanh, 70)):
        torch.manual_seed(3)
        net = implicit_net.ImNet(dim=3, in_features=32, out_features=4, nf=32, activation=act).to(dev)
        lat0 = 0.5 * torch.randn(1, 6, 9, 7, 32, device=dev)
        pts = 0.02 + 0.96 * torch.rand(1, npts, 3, device=dev)
        combo = {(0, 0): 0.7, (1, 1): 1.3} if act is not torch.nn.LeakyReLU else None
        cot = None
        res = {}
        for fused in ("1", "0"):
            monkeypatch.setenv("STPDE_FC1_FUSED", fused)
            lat = lat0.clone().requires_grad_(True)
            for p in net.parameters():
                p.grad = None
            with _lib.dispatch_trace() as tr:
                jets, _ = lig_jet.lig_jets(net, lat, pts, 0., 1., True, () if combo else ((0, 0), (1, 1)), combo=combo)
                if cot is None:
                    cot = torch.randn_like(jets)
                (jets * cot).sum().backward()
                torch.cuda.synchronize()
            assert tr.has("k_fc1_bwd_fused") == (fused == "1"), "\n".join(tr.kernels)
            assert tr.has("k_fc1_dgrad_spec") == (fused == "0"), "\n".join(tr.kernels)
            res[fused] = (lat.grad.clone(), [p.grad.clone() for p in net.parameters()])
        (la, ga), (lb, gb) = res["1"], res["0"]
        # (same expression sequence for the layer-0 adjoint in both kernels; the compiler may still contract its FMAs differently
        # inside the fused evaluation, so d latent / fc0's gradient are compared to rounding as well, not bit for bit)
        assert (la - lb).abs().max().item() <= 2e-4 * lb.abs().max().item(), (act, npts)
        for k, (a, b) in enumerate(zip(ga, gb)):
            # (fc0's bias gradient is a plain sum of the bf16 adjoint blocks: the two kernels' jets are compiled separately, a
            # handful of bf16 roundings of the layer-0 adjoint flip (2^-9 each) and show at up to ~1.2e-3 of the largest entry; the
            # mode's own distance to the exact gradients is 3e-2)
            assert (a - b).abs().max().item() <= 3e-3 * b.abs().max().item() + 1e-12, (act, npts, k)
