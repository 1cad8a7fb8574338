"""UNet3d / ResBlock3D: state_dict compatibility and numerics vs vectors from the real reference (G7).

CPU (not gpu): module structure + the composed torch-op formulation.  GPU: every convolution goes through the HIP
implicit-GEMM kernels (stpde_conv3d_fwd / stpde_conv3d_wgrad) via the C-ABI library.
"""
import os

import numpy as np
import pytest
import torch

from space_time_pde_amd import unet3d


def _load_net(golden_dir, device):
    d = np.load(os.path.join(golden_dir, "g7_unet.npz"))
    net = unet3d.UNet3d(in_features=4, out_features=32, igres=(4, 8, 8), nf=16, mf=32)
    state = {k[len("state/"):]: torch.from_numpy(d[k]) for k in d.files if k.startswith("state/")}
    missing, unexpected = net.load_state_dict(state, strict=True), None
    return d, net.to(device)


def _check(d, net, device, tol):
    x = torch.from_numpy(d["x"]).to(device).requires_grad_(True)
    cot = torch.from_numpy(d["cot"]).to(device)
    net.train()
    y = net(x)
    assert y.shape == (2, 32, 4, 8, 8)
    assert y.permute(0, 2, 3, 4, 1).is_contiguous()          # train.py:60's permute is a free view
    (y * cot).sum().backward()

    def rel(a, b):
        b = torch.from_numpy(b).double()
        return (a.detach().double().cpu() - b).abs().max().item() / b.abs().max().item()

    assert rel(y, d["y_train"]) < tol
    assert rel(x.grad, d["dx"]) < 10 * tol
    params = dict(net.named_parameters())
    for k in d.files:
        if k.startswith("grad/"):
            assert rel(params[k[5:]].grad, d[k]) < 10 * tol, k
    sd = net.state_dict()
    for k in d.files:
        if k.startswith("after/"):
            got = sd[k[6:]]
            if "num_batches" in k:
                assert int(got) == int(d[k])
            else:
                assert rel(got, d[k]) < tol, k
    net.eval()
    with torch.no_grad():
        assert rel(net(x), d["y_eval"]) < tol


def test_state_dict_keys_match_reference(golden_dir):
    d = np.load(os.path.join(golden_dir, "g7_unet.npz"))
    ref_keys = sorted(k[len("state/"):] for k in d.files if k.startswith("state/"))
    net = unet3d.UNet3d(in_features=4, out_features=32, igres=(4, 8, 8), nf=16, mf=32)
    assert sorted(net.state_dict().keys()) == ref_keys


def test_layer_plan_of_the_bench_config():
    """igres (32,128,128), nf=16, mf=256 -> 64 convs, 9,417,696 parameters; default -> 1,458,400."""
    net = unet3d.UNet3d(in_features=4, out_features=32, igres=(32, 128, 128), nf=16, mf=256)
    assert sum(p.numel() for p in net.parameters()) == 9417696
    assert sum(isinstance(m, torch.nn.Conv3d) for m in net.modules()) == 64
    net = unet3d.UNet3d(in_features=4, out_features=32, igres=(4, 16, 16), nf=16, mf=256)
    assert sum(p.numel() for p in net.parameters()) == 1458400
    with pytest.raises(ValueError):
        unet3d.UNet3d(igres=(4, 12, 16))
    with pytest.raises(ValueError):
        unet3d.UNet3d(igres=(8, 8, 8), ogres=(4, 8, 8))


def test_unet_matches_reference_cpu(golden_dir):
    d, net = _load_net(golden_dir, "cpu")
    _check(d, net, "cpu", 1e-4)


@pytest.mark.gpu
def test_unet_matches_reference_hip(golden_dir, hiplib):
    d, net = _load_net(golden_dir, "cuda:0")
    _check(d, net, "cuda:0", 2e-4)


@pytest.mark.gpu
def test_conv3d_hip_vs_torch_fp64(hiplib):
    """Direct conv parity incl. odd sizes, batch 2, 4 -> 16 input-channel padding and both kernel sizes."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(9)
    for ci, co, k, shape in ((4, 16, 1, (2, 3, 5, 7)), (16, 32, 3, (2, 3, 5, 7)), (48, 16, 3, (1, 1, 2, 20)),
                             (32, 32, 3, (1, 4, 16, 16))):
        conv = torch.nn.Conv3d(ci, co, k, padding=(k - 1) // 2)
        x = torch.randn(*shape, ci, generator=g)
        cot = torch.randn(*shape, co, generator=g)
        xd = x.to(dev).requires_grad_(True)
        cd = conv.to(dev)
        y = unet3d._conv_cl(xd, cd)
        (y * cot.to(dev)).sum().backward()
        c64 = torch.nn.Conv3d(ci, co, k, padding=(k - 1) // 2).double()
        c64.load_state_dict({kk: v.double().cpu() for kk, v in cd.state_dict().items()})
        x64 = x.double().requires_grad_(True)
        y64 = c64(x64.permute(0, 4, 1, 2, 3)).permute(0, 2, 3, 4, 1)
        (y64 * cot.double()).sum().backward()

        def rel(a, b):
            return (a.double().cpu() - b).abs().max().item() / b.abs().max().item()

        assert rel(y.detach(), y64.detach()) < 1e-5
        assert rel(xd.grad, x64.grad) < 1e-5
        assert rel(cd.weight.grad, c64.weight.grad) < 1e-4
        assert rel(cd.bias.grad, c64.bias.grad) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("factors", [(2, 2, 2), (1, 2, 2), (2, 1, 1)])
def test_pool_and_upsample_hip_vs_torch(hiplib, factors):
    """stpde_resample3d against F.max_pool3d / repeat_interleave, values and gradients, including exact ties."""
    import torch.nn.functional as F
    from space_time_pde_amd import unet3d
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 4, 6, 8, 16, generator=g)
    x = torch.relu(x)                                   # many exact ties (zeros), as after a ResBlock
    x[0, :2, :2, :2, :] = 1.5                           # and a window of equal positive values
    xa = x.to(dev).requires_grad_(True)
    xb = x.to(dev).requires_grad_(True)
    y = unet3d._pool_cl(xa, factors)
    yr = F.max_pool3d(xb.permute(0, 4, 1, 2, 3), factors).permute(0, 2, 3, 4, 1)
    assert torch.equal(y, yr.contiguous())
    cot = torch.randn(y.shape, generator=g).to(dev)
    (y * cot).sum().backward()
    (yr * cot).sum().backward()
    assert torch.equal(xa.grad, xb.grad)
    ua = x.to(dev).requires_grad_(True)
    ub = x.to(dev).requires_grad_(True)
    u = unet3d._upsample_cl(ua, factors)
    ur = ub
    for dim, f in zip((1, 2, 3), factors):
        if f != 1:
            ur = ur.repeat_interleave(f, dim=dim)
    assert torch.equal(u, ur)
    cot = torch.randn(u.shape, generator=g).to(dev)
    (u * cot).sum().backward()
    (ur * cot).sum().backward()
    assert torch.allclose(ua.grad, ub.grad, rtol=1e-6, atol=1e-6)
