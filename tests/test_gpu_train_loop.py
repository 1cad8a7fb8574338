"""End-to-end: data pipeline -> UNet3d -> LIG/IM-NET jets -> RB2 residuals -> losses -> backward -> fused clip+Adam,
all through the product modules on the GPU (the loop body of experiments/rb2d/train.py:58-83)."""
import pytest
import torch


@pytest.mark.gpu
def test_short_training_run_reduces_the_loss(hiplib):
    from space_time_pde_amd import implicit_net, local_implicit_grid as lig, nonlinearities, physics, unet3d
    from space_time_pde_amd.dataloader_spacetime import RB2DeviceLoader
    from space_time_pde_amd.optim import FusedClipAdam
    from space_time_pde_amd.train_step import sharded_step
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    # smooth synthetic "simulation" [4, T, Z, X]
    t, z, x = torch.meshgrid(torch.linspace(0, 1, 24), torch.linspace(0, 1, 48), torch.linspace(0, 1, 40), indexing="ij")
    data = torch.stack([torch.sin(3 * x + t), torch.cos(2 * z) * t, torch.sin(2 * x) * torch.cos(3 * z), x * z - t], 0)
    ld = RB2DeviceLoader(data, nx=32, nz=32, nt=8, n_samp_pts_per_crop=1024, downsamp_xz=2, downsamp_t=2,
                         normalize_output=True, device=dev)
    unet = unet3d.UNet3d(in_features=4, out_features=32, igres=(4, 16, 16), nf=16, mf=64).to(dev).train()
    net = implicit_net.ImNet(dim=3, in_features=32, out_features=4, nf=16,
                             activation=nonlinearities.NONLINEARITIES["softplus"]).to(dev)
    layer = physics.get_rb2_pde_layer(mean=tuple(ld.channel_mean.tolist()), std=tuple(ld.channel_std.tolist()),
                                      t_crop=2., z_crop=1., x_crop=1., use_continuity=True)
    params = list(unet.parameters()) + list(net.parameters())
    opt = FusedClipAdam(params, lr=2e-3, clip_grad=1.0)
    g = torch.Generator(device=dev).manual_seed(1)
    calls0 = lig.stats["hip_jet_calls"]
    losses = []
    for step in range(30):
        lres, pts, vals = ld.get([3, 57], generator=g)
        opt.zero_grad(set_to_none=True)
        loss, reg, pde = sharded_step(unet, net, layer, lres, pts, vals, pts.shape[1], 1.0, 0.0125, "l1")
        opt.step()
        losses.append(float(loss))
        assert torch.isfinite(loss)
    assert lig.stats["hip_jet_calls"] == calls0 + 30          # the HIP jet path carried every step
    assert sum(losses[-5:]) / 5 < 0.8 * sum(losses[:5]) / 5, losses
