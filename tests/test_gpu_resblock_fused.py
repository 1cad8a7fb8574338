"""Fused residual block (unet3d._ResBlockHip over csrc/conv3d_fused.hip, round 4) -- reference src/unet3d.py:12-56.

Every case runs ONE ResBlock3D in training mode through the C-ABI library twice -- the fused path and the
one-kernel-per-layer path (STPDE_FUSED_RESBLOCK=0) -- and against torch's own Conv3d / BatchNorm3d in fp64:
output, input gradient, all 14 parameter gradients, running statistics.  Shapes cover what the U-Net of the bench
configurations sends through it: the padded 4-channel input block, tap-split deep levels (8 ... 512 voxels, where the
statistics / mask epilogues fall back to passes of their own), mid levels, and the 4-voxel-tiles-per-wave kernels of the
full-resolution level (524,288 voxels).
"""
import copy
import os
import subprocess
import sys

import pytest
import torch

from space_time_pde_amd import unet3d

pytestmark = pytest.mark.gpu


def _ref_block(blk, x, cot):
    """torch fp64 reference of the same block (NCDHW)."""
    ref = copy.deepcopy(blk).double()
    xr = x.detach().double().permute(0, 4, 1, 2, 3).contiguous().requires_grad_(True)
    h = torch.relu(ref.bn1(ref.conv1(xr)))
    h = torch.relu(ref.bn2(ref.conv2(h)))
    h = ref.bn3(ref.conv3(h)) + ref.shortcut(xr)
    if ref.final_relu:
        h = torch.relu(h)
    (h.permute(0, 2, 3, 4, 1) * cot.double()).sum().backward()
    return ref, h.permute(0, 2, 3, 4, 1).detach(), xr.grad.permute(0, 2, 3, 4, 1)


def _run(blk, x, cot, fused):
    blk = copy.deepcopy(blk)
    os.environ["STPDE_FUSED_RESBLOCK"] = "1" if fused else "0"
    try:
        before = unet3d.stats["fused_resblocks"]
        xx = x.detach().clone().requires_grad_(True)
        y = blk.forward_cl(xx)
        (y * cot).sum().backward()
        assert (unet3d.stats["fused_resblocks"] - before == 1) == fused
    finally:
        os.environ.pop("STPDE_FUSED_RESBLOCK", None)
    return blk, y.detach(), xx.grad


def _rel(a, b):
    b = b.double()
    return (a.double() - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def _rl2(a, b):
    """Frobenius-norm distance: a ReLU mask that flips between fp32 and fp64 (pre-activation within rounding of zero -- a
    few of the 16 M elements of the full-resolution level always are) moves single gradient entries by their full size."""
    b = b.double()
    return (a.double() - b).norm().item() / max(b.norm().item(), 1e-30)


CASES = [  # (B, T, Z, X), in, neck, out, final_relu
    ((1, 2, 2, 2), 256, 256, 256, True),        # deepest level: 8 voxels, tap-split 3x3x3
    ((2, 4, 4, 4), 128, 64, 128, True),         # 128 voxels
    ((1, 8, 8, 8), 256, 128, 128, True),        # 512 voxels, up-path shape (concatenated input)
    ((1, 4, 16, 16), 4, 16, 16, True),          # padded input channels (conv_in)
    ((1, 8, 32, 32), 64, 32, 32, False),        # 8192 voxels, no final relu (conv_out)
    ((1, 16, 64, 64), 16, 16, 32, True),        # 65,536 voxels: one voxel tile per wave, no tap split
    ((1, 32, 128, 128), 16, 16, 32, True),      # full-resolution level of configs[1]: four voxel tiles per wave
    ((1, 32, 128, 128), 32, 32, 32, False),
    ((1, 64, 256, 256), 16, 16, 32, True),      # full-resolution level of configs[3]: 4,194,304 voxels
]


@pytest.mark.parametrize("shape,ci,cn,co,final_relu", CASES)
def test_fused_block_vs_layerwise_and_fp64(hiplib, shape, ci, cn, co, final_relu):
    dev = torch.device("cuda:0")
    torch.manual_seed(sum(shape) + ci + cn + co)
    blk = unet3d.ResBlock3D(ci, cn, co, final_relu=final_relu).to(dev).train()
    with torch.no_grad():
        for bn in (blk.bn1, blk.bn2, blk.bn3):            # non-trivial affine parameters and running statistics
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.3, 0.3)
            bn.running_mean.uniform_(-0.2, 0.2)
            bn.running_var.uniform_(0.5, 2.0)
    x = torch.randn(*shape, ci, device=dev) + 0.5           # a mean of the order of the deviation
    cot = torch.randn(*shape, co, device=dev)
    ref, yr, dxr = _ref_block(blk, x, cot)
    bf, yf, dxf = _run(blk, x, cot, True)
    bl, yl, dxl = _run(blk, x, cot, False)
    nvox = shape[0] * shape[1] * shape[2] * shape[3]
    # BatchNorm over a handful of voxels amplifies fp32 rounding: looser bound on the 8-voxel level
    tol = 2e-3 if nvox <= 8 else (2e-4 if nvox <= 512 else 5e-5)
    assert _rel(yf, yr) < tol and _rel(yl, yr) < tol
    # mask flips: ~1e-6 of the elements sit within fp32 rounding of a ReLU kink, each moves one gradient entry by its full
    # size (Frobenius distance ~1e-3 at 16 M elements, for the layer-wise kernels just as much)
    ef, el = _rl2(dxf, dxr), _rl2(dxl, dxr)
    assert ef < max(20 * tol, 3e-3) and ef < 2 * el + 20 * tol, (ef, el)
    pr, pf, pl = dict(ref.named_parameters()), dict(bf.named_parameters()), dict(bl.named_parameters())
    worst = 0.0
    for k, g in pr.items():
        scale = max(g.grad.abs().max().item(), 1e-3 * max(p.grad.abs().max().item() for p in pr.values()))
        ef = (pf[k].grad.double() - g.grad).abs().max().item() / scale
        el = (pl[k].grad.double() - g.grad).abs().max().item() / scale
        # conv biases in front of a training-mode BatchNorm have an exactly zero gradient: rounding noise only
        if k in ("conv1.bias", "conv2.bias", "conv3.bias"):
            continue
        worst = max(worst, ef)
        # (524,288-voxel cases: a handful of mask flips against fp64 -- see above -- move a sum over the voxels by a few of its
        # terms, i.e. by ~sqrt(flips / voxels) = a few 1e-3 of its size, in either path)
        assert ef < (1e-2 if nvox > 100000 else 50 * tol), (k, ef, el)
    for name in ("bn1", "bn2", "bn3"):
        for stat in ("running_mean", "running_var"):
            r = getattr(getattr(ref, name), stat)
            assert _rel(getattr(getattr(bf, name), stat), r) < tol, (name, stat)
        assert int(getattr(bf, name).num_batches_tracked) == 1


def test_fused_block_statistics_survive_a_large_mean(hiplib):
    """Channel means 300 deviations away from zero: the per-wave shifted sums + fp64 conversion keep the variance."""
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    blk = unet3d.ResBlock3D(16, 16, 32).to(dev).train()
    with torch.no_grad():
        blk.conv1.bias.fill_(300.0)           # conv1's output: mean 300, deviation ~0.5
    x = torch.randn(1, 16, 64, 64, 16, device=dev)
    cot = torch.randn(1, 16, 64, 64, 32, device=dev)
    ref, yr, dxr = _ref_block(blk, x, cot)
    bf, yf, dxf = _run(blk, x, cot, True)
    assert _rel(yf, yr) < 2e-3                # (x - mean) itself loses 300 / 0.5 * 2^-24 relative
    assert _rel(bf.bn1.running_var, ref.bn1.running_var) < 1e-4


def test_unet_with_fused_blocks_equals_layerwise_unet(hiplib):
    """Whole UNet3d, training mode, deferred weight gradients on (the side-stream route of sharded_step): fused blocks vs
    the layer-wise path on a grid whose deepest level still has 64 voxels."""
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    net = unet3d.UNet3d(in_features=4, out_features=32, igres=(16, 32, 32), nf=16, mf=64).to(dev).train()
    x = torch.randn(1, 4, 16, 32, 32, device=dev)
    cot = torch.randn(1, 32, 16, 32, 32, device=dev)

    def run(fused, deferred):
        os.environ["STPDE_FUSED_RESBLOCK"] = "1" if fused else "0"
        try:
            n = copy.deepcopy(net)
            n.deferred_weight_grads = deferred
            xx = x.clone().requires_grad_(True)
            y = n(xx)
            (y * cot).sum().backward()
            torch.cuda.synchronize()
            return y.detach(), xx.grad, {k: p.grad for k, p in n.named_parameters()}
        finally:
            os.environ.pop("STPDE_FUSED_RESBLOCK", None)

    yl, dxl, gl = run(False, False)
    for deferred in (False, True):
        before = unet3d.stats["fused_resblocks"]
        yf, dxf, gf = run(True, deferred)
        assert unet3d.stats["fused_resblocks"] - before == len([m for m in net.modules() if isinstance(m, unet3d.ResBlock3D)])
        assert _rel(yf, yl) < 2e-3
        assert _rl2(dxf, dxl) < 2e-2
        gmax = max(g.abs().max().item() for g in gl.values())
        for k in gl:
            if k.endswith("bias") and ("conv1" in k or "conv2" in k or "conv3" in k):
                continue                          # exactly-zero gradients: noise
            scale = max(gl[k].abs().max().item(), 1e-3 * gmax)
            assert (gf[k] - gl[k]).abs().max().item() / scale < 5e-2, k


def test_lds_weight_gradient_grid_override(hiplib):
    """stpde_tune "conv_wgrad_lds_gx" only changes how the persistent workgroups share the voxel blocks: same weight gradient
    as the default grid and as torch."""
    code = r'''
import ctypes as C, sys, torch
from space_time_pde_amd import _lib, unet3d
_lib.tune("conv_wgrad_lds_gx", int(sys.argv[1]))
torch.manual_seed(3)
dev = torch.device("cuda:0")
x = torch.randn(1, 16, 64, 64, 32, device=dev)
gy = torch.randn(1, 16, 64, 64, 32, device=dev)
d = unet3d._desc(x, 32, 32, 3)
dw = torch.zeros(27, 32, 32, device=dev)
with _lib.dispatch_trace() as tr:
    _lib.check(_lib.lib().stpde_conv3d_wgrad(C.byref(d), _lib.ptr(x), _lib.ptr(gy), _lib.ptr(dw), _lib.stream_ptr()))
assert any("k_conv3d_wgrad_lds" in k for k in tr.kernels), tr.kernels
ref = torch.nn.grad.conv3d_weight(x.double().permute(0, 4, 1, 2, 3), (32, 32, 3, 3, 3), gy.double().permute(0, 4, 1, 2, 3), padding=1)
got = dw.permute(1, 2, 0).reshape(32, 32, 3, 3, 3).double()
err = (got - ref).abs().max().item() / ref.abs().max().item()
print("ERR", err)
assert err < 2e-5
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for gx in ("7", "256", "1000"):
        env = dict(os.environ, PYTHONPATH=root)
        out = subprocess.run([sys.executable, "-c", code, gx], env=env, cwd=root, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout + out.stderr


def test_fused_block_frozen_weights_still_deliver_bias_gradients(hiplib):
    """a block whose convolution WEIGHTS are frozen but whose biases train must still get its bias gradients
    from the fused node (it gated all weight AND bias jobs on the weights alone); same values as the layer-wise path."""
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    blk = unet3d.ResBlock3D(32, 16, 32, final_relu=True).to(dev).train()
    for c in (blk.conv1, blk.conv2, blk.conv3, blk.shortcut):
        c.weight.requires_grad_(False)
    x = torch.randn(1, 4, 8, 8, 32, device=dev)
    cot = torch.randn(1, 4, 8, 8, 32, device=dev)
    fus, yf, gxf = _run(blk, x, cot, True)
    lay, yl, gxl = _run(blk, x, cot, False)
    assert _rel(yf, yl) < 1e-5 and _rel(gxf, gxl) < 1e-4
    for name in ("conv1", "conv2", "conv3", "shortcut"):
        cf, cl = getattr(fus, name), getattr(lay, name)
        assert cf.weight.grad is None and cl.weight.grad is None
        assert cf.bias.grad is not None, name
        # (a bias in front of a training-mode BatchNorm has an exactly-zero gradient: compare absolutely)
        assert (cf.bias.grad - cl.bias.grad).abs().max().item() <= 1e-4 * max(1.0, cl.bias.grad.abs().max().item()), name
    assert fus.shortcut.bias.grad.abs().max().item() > 0
