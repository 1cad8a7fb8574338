"""Timings of the SURVEY 8(f) "next" rows on one MI355X (prints one JSON line; not part of bench.py's metric).

  N1  FusedClipAdam.step() over the UNet3d + ImNet parameters of the bench config (one multi-tensor launch)
  N2  evaluate_feat_grid on a dense lattice (values + RB2 residuals, as experiments/rb2d/evaluation.py)
  N3  RB2DeviceLoader.get(): crops + low-res down-sampling + target interpolation for a batch, on device
"""
import json
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from space_time_pde_amd import implicit_net, local_implicit_grid as lig, nonlinearities, physics, unet3d  # noqa: E402
from space_time_pde_amd.dataloader_spacetime import RB2DeviceLoader  # noqa: E402
from space_time_pde_amd.inference import evaluate_feat_grid  # noqa: E402
from space_time_pde_amd.optim import FusedClipAdam  # noqa: E402


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = implicit_net.ImNet(dim=3, in_features=32, out_features=4, nf=32,
                             activation=nonlinearities.NONLINEARITIES["softplus"]).to(dev)
    unet = unet3d.UNet3d(in_features=4, out_features=32, igres=(32, 128, 128), nf=16, mf=256).to(dev)
    params = list(unet.parameters()) + list(net.parameters())
    for p in params:
        p.grad = torch.randn_like(p)
    n_par = sum(p.numel() for p in params)
    out = {}
    opt = FusedClipAdam(params, lr=1e-3, clip_grad=1.0)
    fresh = [torch.randn_like(p) for p in params]

    def step_gathered():            # what a training loop does: autograd left new gradient tensors in .grad
        for p, g in zip(params, fresh):
            p.grad = g
        opt.step()

    t_g = timed(step_gathered, 20)
    opt.gather_grads()
    t = timed(opt.step, 20)          # gradients already in the flat buffer (e.g. after the all-reduce on it)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(20):
        opt.step()
    ev1.record()
    torch.cuda.synchronize()
    t_dev = ev0.elapsed_time(ev1) / 20 * 1e-3
    out["N1_clip_adam"] = dict(params=n_par, tensors=len(params), ms_per_step=round(1e3 * t, 3),
                               ms_per_step_device=round(1e3 * t_dev, 3),
                               ms_per_step_with_gradient_gather=round(1e3 * t_g, 3),
                               hbm_GBps=round(28.0 * n_par / t / 1e9, 1), hbm_GBps_device=round(28.0 * n_par / t_dev / 1e9, 1),
                               note="28 B per element (16 read, 12 written); flat parameter / moment / gradient buffers, "
                                    "one launch; ms_per_step = host wall per step of a back-to-back loop, _device = HIP events")
    ref = torch.optim.Adam(params, lr=1e-3)

    def ref_step():
        torch.nn.utils.clip_grad_value_(params, 1.0)
        ref.step()
    out["N1_clip_adam"]["torch_clip_plus_adam_ms"] = round(1e3 * timed(ref_step, 20), 3)

    # N2: lattice inference, 32 x 512 x 512 = 8.4 M points through values + residuals
    with torch.no_grad():
        unet.eval()
        latent = unet(torch.randn(1, 4, 32, 128, 128, device=dev)).permute(0, 2, 3, 4, 1).contiguous()
    layer = physics.get_rb2_pde_layer(mean=(0.01, 0., 0.02, -0.01), std=(0.05, 0.3, 0.15, 0.12), t_crop=2., z_crop=1.,
                                      x_crop=1., use_continuity=True)
    layer.update_forward_method(lambda q: lig.query_local_implicit_grid(net, latent, q, 0., 1.))
    eps = 1e-6
    seqs = [torch.linspace(eps, 1 - eps, n) for n in (32, 512, 512)]
    t0 = time.perf_counter()
    res = evaluate_feat_grid(layer, latent, *seqs)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    npts = 32 * 512 * 512
    out["N2_evaluate_feat_grid"] = dict(points=npts, outputs=sorted(res), seconds=round(t, 3),
                                        points_per_s=round(npts / t), note="includes the device-to-host copies of "
                                        "4 channels + 4 residual fields, as the reference's evaluation does")

    # N3: data pipeline on device: synthetic RB2 run [4, 200, 512, 128], crops 16 x 128 x 128, 1024 points each
    data = torch.randn(4, 200, 512, 128)
    ld = RB2DeviceLoader(data, nx=128, nz=128, nt=16, n_samp_pts_per_crop=1024, downsamp_xz=4, downsamp_t=4,
                         normalize_output=True, device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    idx = torch.randint(0, len(ld), (64,)).tolist()
    t = timed(lambda: ld.get(idx, generator=g), 10)
    out["N3_device_loader"] = dict(batch=64, crop="16x128x128 -> 4x32x32 low-res + 1024 target points", ms_per_batch=round(1e3 * t, 3),
                                   samples_per_s=round(64 / t))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
