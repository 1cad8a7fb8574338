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


# ---------------------------------------------------------------------------------------------------------------
# The U-Net kernels AT BENCHMARK SIZE: the full-resolution levels of BASELINE configs[1]
# (32 x 128 x 128 = 524,288 voxels -> the 4-voxel-tiles-per-wave conv variants, ntiles >= 16384) and configs[3].
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("ci,co,k", [(4, 16, 1), (16, 16, 3), (16, 32, 1), (32, 32, 3), (32, 64, 1)])
def test_conv3d_at_bench_volume_vs_torch_fp64(hiplib, ci, co, k):
    """Channel shapes of the full-resolution C2 / C4 levels on the C2 volume: forward, input gradient, weight and bias
    gradient vs torch fp64 on the host; the dispatch trace asserts the big-volume kernel variants ran."""
    from space_time_pde_amd import _lib
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(100 + ci + co + k)
    shape = (1, 32, 128, 128)
    conv = torch.nn.Conv3d(ci, co, k, padding=(k - 1) // 2)
    x = torch.randn(*shape, ci, generator=g)
    cot = torch.randn(*shape, co, generator=g)
    xd = x.to(dev).requires_grad_(True)
    cd = conv.to(dev)
    with _lib.dispatch_trace() as tr:
        y = unet3d._conv_cl(xd, cd)
        (y * cot.to(dev)).sum().backward()
        torch.cuda.synchronize()
    mt = co // 16
    assert tr.has("k_conv3d_fwd<%d, 4>" % min(mt, 4)), "\n".join(tr.kernels)     # forward: 4 voxel tiles per wave
    # weight gradient: the LDS-tile kernels of these levels (3x3x3: halo tiles; 1x1x1: 256-voxel tiles, round 5), the per-wave
    # kernel for the 4-channel input layer (padded to one 16-channel tile by _conv_cl, and not worth a variant)
    want = "k_conv3d_wgrad_lds" if k == 3 else "k_conv1_wgrad_lds<%d, %d, false>" % (max(ci, 16) // 16, co // 16)
    assert tr.has(want), "\n".join(tr.kernels)
    c64 = torch.nn.Conv3d(ci, co, k, padding=(k - 1) // 2).double()
    c64.load_state_dict({kk: v.double().cpu() for kk, v in cd.state_dict().items()})
    x64 = x.double().requires_grad_(True)
    y64 = c64(x64.permute(0, 4, 1, 2, 3)).permute(0, 2, 3, 4, 1)
    (y64 * cot.double()).sum().backward()

    def rel(a, b):
        return (a.double().cpu() - b).abs().max().item() / b.abs().max().item()

    assert rel(y.detach(), y64.detach()) < 1e-5
    assert rel(xd.grad, x64.grad) < 1e-5
    # 524,288-term fp32 sums (block partial sums + atomics): error grows with sqrt(n) relative to the terms' size
    assert rel(cd.weight.grad, c64.weight.grad) < 2e-4
    assert rel(cd.bias.grad, c64.bias.grad) < 2e-4


@pytest.mark.gpu
@pytest.mark.parametrize("ci,co", [(16, 32), (32, 16), (32, 32), (16, 16), (64, 64)])
def test_conv3d_wgrad_lds_tiles_two_samples_vs_torch_fp64(hiplib, ci, co):
    """The LDS-tile weight-gradient kernel (csrc/conv3d.hip, k_conv3d_wgrad_lds) on a batch of TWO samples (the halo of a
    block must stop at the sample boundary) and on every channel-tile combination it is compiled for."""
    from space_time_pde_amd import _lib
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7 + ci + 2 * co)
    shape = (2, 16, 32, 128)
    conv = torch.nn.Conv3d(ci, co, 3, padding=1)
    x = torch.randn(*shape, ci, generator=g)
    cot = torch.randn(*shape, co, generator=g)
    xd = x.to(dev).requires_grad_(True)
    cd = conv.to(dev)
    with _lib.dispatch_trace() as tr:
        y = unet3d._conv_cl(xd, cd)
        (y * cot.to(dev)).sum().backward()
        torch.cuda.synchronize()
    # (64 channels, round 5: 16-voxel rows, one output tile per workgroup)
    want = "k_conv3d_wgrad_lds<4, 1, 16, 4>" if ci == 64 else "k_conv3d_wgrad_lds<%d, %d>" % (ci // 16, co // 16)
    assert tr.has(want), "\n".join(tr.kernels)
    c64 = torch.nn.Conv3d(ci, co, 3, padding=1).double()
    c64.load_state_dict({kk: v.double().cpu() for kk, v in cd.state_dict().items()})
    x64 = x.double().requires_grad_(True)
    y64 = c64(x64.permute(0, 4, 1, 2, 3)).permute(0, 2, 3, 4, 1)
    (y64 * cot.double()).sum().backward()
    err = (cd.weight.grad.double().cpu() - c64.weight.grad).abs().max().item() / c64.weight.grad.abs().max().item()
    assert err < 1e-4, err


@pytest.mark.gpu
@pytest.mark.parametrize("c,relu,res", [(16, True, False), (32, True, True), (32, False, True)])
def test_batchnorm_at_bench_volume_vs_torch_fp64(hiplib, c, relu, res):
    """Fused BatchNorm(+add)(+ReLU) over 524,288 voxels (training statistics, running-stat update, backward)."""
    import torch.nn.functional as F
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7 + c)
    n = 32 * 128 * 128
    x = torch.randn(1, 32, 128, 128, c, generator=g) * 1.7 + 0.3
    r = torch.randn(1, 32, 128, 128, c, generator=g) if res else None
    cot = torch.randn(1, 32, 128, 128, c, generator=g)
    bn = torch.nn.BatchNorm3d(c)
    with torch.no_grad():
        bn.weight.copy_(1 + 0.1 * torch.randn(c, generator=g))
        bn.bias.copy_(0.1 * torch.randn(c, generator=g))
    bnd = torch.nn.BatchNorm3d(c).to(dev)
    bnd.load_state_dict(bn.state_dict())
    bnd.train()
    x64 = x.double().requires_grad_(True)
    r64 = r.double().requires_grad_(True) if res else None
    w64, b64 = bn.weight.detach().double().requires_grad_(True), bn.bias.detach().double().requires_grad_(True)
    rm, rv = torch.zeros(c, dtype=torch.float64), torch.ones(c, dtype=torch.float64)
    h = F.batch_norm(x64.reshape(n, c), rm, rv, w64, b64, True, 0.1, bn.eps).reshape(x.shape)
    if res:
        h = h + r64
    y64 = F.relu(h) if relu else h
    # elements whose pre-ReLU value is within fp32 rounding of the kink may legitimately take the other branch
    # (~1 of 16.7 M; a single one moves a weight-gradient entry by O(1) of ~10^3): their cotangent is zeroed
    if relu:
        cot = cot * (h.detach().abs() > 1e-5).float()
    (y64 * cot.double()).sum().backward()
    xd = x.to(dev).requires_grad_(True)
    rd = r.to(dev).requires_grad_(True) if res else None
    y = unet3d._bn_act(xd, bnd, relu, residual=rd)
    (y * cot.to(dev)).sum().backward()

    def rel(a, b):
        return (a.double().cpu() - b).abs().max().item() / b.abs().max().item()

    relm = rel
    assert rel(y.detach(), y64.detach()) < 1e-5
    assert rel(bnd.running_mean, rm) < 5e-5 and rel(bnd.running_var, rv) < 5e-5   # fp32 mean of 524,288 values
    assert relm(xd.grad, x64.grad) < 2e-5
    if res:
        assert relm(rd.grad, r64.grad) < 1e-6
    assert rel(bnd.weight.grad, w64.grad) < 2e-4 and rel(bnd.bias.grad, b64.grad) < 2e-4


@pytest.mark.gpu
def test_resample_at_bench_volume(hiplib):
    import torch.nn.functional as F
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(4)
    x = torch.relu(torch.randn(1, 32, 128, 128, 32, generator=g)).to(dev)
    for factors in ((1, 2, 2), (2, 2, 2)):
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        y = unet3d._pool_cl(xa, factors)
        yr = F.max_pool3d(xb.permute(0, 4, 1, 2, 3), factors).permute(0, 2, 3, 4, 1).contiguous()
        if not torch.equal(y, yr):          # say WHICH side is off (host reference) and whether the input survived
            ref = F.max_pool3d(x.cpu().permute(0, 4, 1, 2, 3), factors).permute(0, 2, 3, 4, 1).contiguous()
            bad_y, bad_r = (y.detach().cpu() != ref), (yr.detach().cpu() != ref)
            raise AssertionError("pool %s: hip kernel wrong in %d elements (first %s), torch in %d, inputs intact: %s / %s" % (
                factors, int(bad_y.sum()), bad_y.nonzero()[:3].tolist(), int(bad_r.sum()),
                torch.equal(xa.detach(), x), torch.equal(xb.detach(), x)))
        cot = torch.randn(y.shape, generator=g).to(dev)
        (y * cot).sum().backward()
        (yr * cot).sum().backward()
        assert torch.equal(xa.grad, xb.grad)
    small = x[:, :16, :64, :64].contiguous()
    u = unet3d._upsample_cl(small, (2, 2, 2))
    assert torch.equal(u, small.repeat_interleave(2, 1).repeat_interleave(2, 2).repeat_interleave(2, 3))


@pytest.mark.gpu
@pytest.mark.parametrize("igres", [(32, 128, 128), (64, 256, 256)])
def test_unet_at_config_size_hip_vs_torch_cpu(hiplib, igres):
    """The whole encoder at BASELINE configs[1] / configs[3] size, HIP vs the same module on the host (torch ops).

    In TRAINING mode this network is numerically chaotic in fp32 at these sizes -- on the host, the same fp32 module run
    with 8 threads and with 1 thread differs by 54 % of the output maximum, fp32 vs fp64 by 28 % (7-8 pooling levels, the
    deepest BatchNorms see 8 voxels; C1 with 5 levels: 1e-3) -- so no implementation can be pinned on its training-mode
    output there.  Training-mode parity at size is therefore asserted per kernel (the three tests above: conv, fused
    BatchNorm, resample on the 524,288-voxel level) and end to end on C1 (G7 / G8); here the whole network runs in
    EVALUATION mode (running statistics, perturbed so that they are not the identity): every convolution variant, pooling,
    up-sampling, concatenation and BatchNorm-apply at full size, forward and (C2) input gradient."""
    import copy
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    net = unet3d.UNet3d(in_features=4, out_features=32, igres=igres, nf=16, mf=256)
    g = torch.Generator().manual_seed(12)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm3d):
                m.running_mean.copy_(0.1 * torch.randn(m.num_features, generator=g))
                m.running_var.copy_(0.6 + 0.2 * torch.rand(m.num_features, generator=g))
    x = torch.randn(1, 4, *igres, generator=g)
    nd = copy.deepcopy(net).to(dev).eval()
    net.eval()
    small = igres[0] == 32
    xd = x.to(dev).requires_grad_(small)
    xc = x.clone().requires_grad_(small)
    if small:
        y, yc = nd(xd), net(xc)
    else:
        with torch.no_grad():
            y, yc = nd(xd), net(xc)
    assert y.shape == (1, 32) + tuple(igres) and torch.isfinite(y).all()

    def rel(a, b):
        return (a.double().cpu() - b.double()).abs().max().item() / b.double().abs().max().item()

    assert rel(y.detach(), yc.detach()) < 1e-4
    if small:
        cot = torch.randn(y.shape, generator=g)
        (y * cot.to(dev)).sum().backward()
        (yc * cot).sum().backward()
        # a handful of the ~10^8 ReLU inputs sit within rounding of the kink and take the other branch on one side:
        # isolated gradient outliers, judged in the Frobenius norm with a loose bound on the maximum
        def nrm(a, b):
            return (a.double().cpu() - b.double()).norm().item() / b.double().norm().item()

        assert nrm(xd.grad, xc.grad) < 2e-4 and rel(xd.grad, xc.grad) < 5e-2
        for name in ("conv_in.conv2.weight", "down_modules.0.conv2.weight", "conv_out.conv3.weight", "conv_mid.conv2.weight"):
            a, b = dict(nd.named_parameters())[name].grad, dict(net.named_parameters())[name].grad
            assert nrm(a, b) < 3e-3, name       # kink flips (see above) accumulate in the weight gradients


@pytest.mark.gpu
def test_deferred_weight_gradients_equal_inline_ones(hiplib):
    """UNet3d.deferred_weight_grads (weight / bias gradients of the convolutions on a side stream, .grad assigned by one
    callback when loss.backward() is over) gives the same gradients as the in-line path, accumulates into existing .grad
    like autograd does, and leaves the input gradient alone."""
    from space_time_pde_amd import _lib
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    # evaluation-mode BatchNorm with perturbed running statistics: a well-conditioned map (in training mode the deepest
    # level normalises over a handful of voxels and two fp32 runs of the SAME code differ by tens of per cent)
    net = unet3d.UNet3d(in_features=4, out_features=32, igres=(8, 16, 16), nf=16, mf=32).to(dev).eval()
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm3d):
                m.running_mean.copy_(0.1 * torch.randn(m.num_features, generator=g))
                m.running_var.copy_(1 + 0.2 * torch.rand(m.num_features, generator=g))
    x = torch.randn(1, 4, 8, 16, 16, device=dev, requires_grad=True)
    cot = torch.randn(1, 32, 8, 16, 16, device=dev)

    def run(deferred, twice=False):
        net.deferred_weight_grads = deferred
        for p in net.parameters():
            p.grad = None
        x.grad = None
        for _ in range(2 if twice else 1):
            (net(x) * cot).sum().backward()
        torch.cuda.synchronize()
        return [p.grad.clone() for p in net.parameters()], x.grad.clone()

    g0, dx0 = run(False)
    with _lib.dispatch_trace() as tr:
        g1, dx1 = run(True)
    assert tr.has("k_conv3d_wgrad"), "\n".join(tr.kernels)
    assert all(g is not None for g in g1)
    nrm = lambda a, b: ((a - b).norm() / (b.norm() + 1e-12)).item()     # noqa: E731
    assert nrm(dx1, dx0) < 1e-5
    for (name, _), a, b in zip(net.named_parameters(), g1, g0):
        assert nrm(a, b) < 2e-5 or b.norm().item() < 1e-6, name        # biases in front of a BatchNorm: ~0 gradients
    g2, _ = run(True, twice=True)                                      # second backward accumulates into .grad
    for (name, _), a, b in zip(net.named_parameters(), g2, g0):
        assert nrm(a, 2 * b) < 5e-5 or b.norm().item() < 1e-6, name
    net.deferred_weight_grads = False


@pytest.mark.gpu
def test_unet_training_mode_at_bench_size_within_fp32_noise_of_fp64(hiplib):
    """VERDICT r2 #3c: the BENCHMARKED encoder -- UNet3d(igres=(32,128,128), nf=16, mf=256) in TRAINING mode (batch
    statistics, reference src/unet3d.py:39-56, 208-240) -- HIP fp32 vs the same module in fp64 on the host, with the
    G8-style bound: the HIP distance to fp64 must be within 3x the distance of torch's OWN fp32 run to fp64 (output and
    five gradient norms).  This network is ill-conditioned in fp32 at this depth (deepest BatchNorms see 8 voxels,
    DESIGN 2a), so an absolute tolerance would pin nothing; the bound says the HIP kernels add no error beyond what fp32
    arithmetic itself does to this map."""
    import copy
    dev = torch.device("cuda:0")
    igres = (32, 128, 128)
    torch.manual_seed(21)
    net32 = unet3d.UNet3d(in_features=4, out_features=32, igres=igres, nf=16, mf=256).train()
    g = torch.Generator().manual_seed(22)
    x = torch.randn(1, 4, *igres, generator=g)
    cot = torch.randn(1, 32, *igres, generator=g) / 1024.0
    net64 = copy.deepcopy(net32).double().train()
    nhip = copy.deepcopy(net32).to(dev).train()
    names = ("conv_in.conv2.weight", "down_modules.0.conv2.weight", "conv_mid.conv2.weight", "up_modules.0.conv2.weight",
             "conv_out.conv3.weight")

    def run(net, xx, cc):
        y = net(xx)
        (y * cc).sum().backward()
        prm = dict(net.named_parameters())
        return y.detach().double().cpu(), [prm[n].grad.detach().double().cpu().norm().item() for n in names]

    y64, g64 = run(net64, x.double(), cot.double())
    y32, g32 = run(net32, x, cot)
    yh, gh = run(nhip, x.to(dev), cot.to(dev))
    assert torch.isfinite(yh).all()

    def dist(a, b):
        return (a - b).norm().item() / b.norm().item()

    d32, dh = dist(y32, y64), dist(yh, y64)
    assert dh < 3 * d32 + 1e-5, ("latent grid", dh, d32)
    for n, a, b, c in zip(names, gh, g32, g64):
        e32, eh = abs(b - c) / c, abs(a - c) / c
        assert eh < 3 * e32 + 1e-3, (n, eh, e32)
    # BatchNorm running statistics after the step (momentum rule, unbiased variance) against the fp64 module
    sd64, sdh = net64.state_dict(), nhip.state_dict()
    sd32 = net32.state_dict()
    for k in ("conv_in.bn1.running_var", "down_modules.0.bn2.running_mean", "conv_out.bn3.running_var"):
        e32 = dist(sd32[k].double(), sd64[k])
        eh = dist(sdh[k].double().cpu(), sd64[k])
        assert eh < 3 * e32 + 1e-5, (k, eh, e32)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False])
def test_deterministic_mode_unet_backward_is_bit_reproducible(hiplib, monkeypatch, fused):
    """VERDICT r5 missing #4 / next #6: ``_lib.deterministic`` on BASELINE configs[1]'s training-mode U-Net (32,128,128): two
    forward + backward passes on the same input and output gradient give BIT-IDENTICAL latent grids, input gradients, parameter
    gradients (every convolution weight / bias and BatchNorm weight / bias) and running statistics -- what the reference's CPU
    path gives (experiments/rb2d/train.py:77) -- on the fused residual-block path and on the layer-wise one.  Without the mode
    the same two passes differ (fp32 / fp64 atomics), which the test also shows, and the deterministic result agrees with the
    default one to the default mode's own run-to-run distance (Frobenius norms: the deep levels amplify rounding)."""
    from space_time_pde_amd import _lib
    dev = torch.device("cuda:0")
    monkeypatch.setenv("STPDE_FUSED_RESBLOCK", "1" if fused else "0")
    torch.manual_seed(21)
    net0 = unet3d.UNet3d(in_features=4, out_features=32, igres=(32, 128, 128), nf=16, mf=256).to(dev).train()
    state = {k: v.clone() for k, v in net0.state_dict().items()}
    g = torch.Generator().manual_seed(22)
    x0 = torch.randn(1, 4, 32, 128, 128, generator=g).to(dev)
    cot = torch.randn(1, 32, 32, 128, 128, generator=g).to(dev)

    def run(det):
        monkeypatch.setattr(_lib, "deterministic", det)
        net0.load_state_dict(state)
        for p in net0.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(True)
        y = net0(x)
        y.backward(cot)
        torch.cuda.synchronize()
        return (y.detach().clone(), x.grad.clone(), [p.grad.clone() for p in net0.parameters()],
                [b.clone() for b in net0.buffers()])

    a, b = run(True), run(True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for k, (u, v) in enumerate(zip(a[2], b[2])):
        assert torch.equal(u, v), (k, (u - v).abs().max().item())
    for u, v in zip(a[3], b[3]):
        assert torch.equal(u, v)
    c, d = run(False), run(False)
    ndiff = sum(int(not torch.equal(u, v)) for u, v in zip(c[2], d[2]))
    print("default mode: %d of %d parameter gradients differ between two runs" % (ndiff, len(c[2])))

    def dist(u, v):
        return (u.double() - v.double()).norm().item() / max(v.double().norm().item(), 1e-30)

    # deterministic vs default: same mathematics, another summation order
    assert dist(a[0], c[0]) <= max(3 * dist(d[0], c[0]), 1e-3)
    assert dist(a[1], c[1]) <= max(3 * dist(d[1], c[1]), 5e-2)


@pytest.mark.gpu
def test_long_accumulator_finalize_matches_fp64(hiplib):
    """csrc/common.h det_add_f32 / det_value through the public pair (stpde_conv3d_wgrad with det = 1, stpde_det_finalize): a
    1x1x1 weight gradient over 300,000 voxels with values spanning 12 orders of magnitude equals the fp64 sum to fp32 rounding,
    is bit-identical across runs and across launch geometries (stpde_tune "conv1_wgrad_lds_gx": other partial sums, same result
    up to the fp32 rounding of the partials, which the integer windows add exactly)."""
    import ctypes as C
    from space_time_pde_amd import _lib
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    nv = 300_000
    x = torch.randn(1, 1, 1, nv, 16, device=dev) * torch.logspace(-6, 6, nv, device=dev)[None, None, None, :, None]
    gy = torch.randn(1, 1, 1, nv, 16, device=dev)
    d = unet3d._desc(x, 16, 16, 1)
    d.det = 1
    outs = []
    for rep in range(3):
        acc = torch.zeros(16 * 16 * 2 * _lib.DET_K, device=dev)
        _lib.check(hiplib.stpde_conv3d_wgrad(C.byref(d), _lib.ptr(x), _lib.ptr(gy), _lib.ptr(acc), _lib.stream_ptr()))
        out = torch.empty(16 * 16, device=dev)
        _lib.check(hiplib.stpde_det_finalize(_lib.ptr(acc), 256, _lib.ptr(out), _lib.stream_ptr()))
        torch.cuda.synchronize()
        outs.append(out.clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    ref = torch.einsum("vo,vi->oi", gy.double().reshape(nv, 16), x.double().reshape(nv, 16)).reshape(-1)
    got = outs[0].double()          # [1 tap][co][ci]
    scale = torch.einsum("vo,vi->oi", gy.double().abs().reshape(nv, 16), x.double().abs().reshape(nv, 16)).reshape(-1)
    assert ((got - ref).abs() <= 3e-6 * scale).all(), ((got - ref).abs() / scale).max().item()
