"""Generate golden vectors by importing the REAL reference (runs only in the build container).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

The reference lives at /root/reference (read-only) and never travels to the GPU box; the fixtures written
here are plain data (inputs, explicit weights, expected outputs).  ``numpy.int`` is shimmed harness-side
because the reference's unet3d.py:191 uses the removed alias (SURVEY.md a-Q6).
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "src"))
sys.path.insert(0, os.path.join(REF, "experiments", "rb2d"))
np.int = int  # noqa: harness-side shim only

import implicit_net  # noqa: E402
import local_implicit_grid as lig  # noqa: E402
import nonlinearities  # noqa: E402
import pde as rpde  # noqa: E402
import physics  # noqa: E402
import regular_nd_grid_interpolation as rgi  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)
import det_init  # noqa: E402  (numpy-seeded parameter / input generation shared with the tests)
ACTS = ["softplus", "leakyrelu", "tanh", "relu", "elu", "swish"]
MEAN = (0.01, 0.0, 0.02, -0.01)
STD = (0.05, 0.3, 0.15, 0.12)


def t2n(t):
    return t.detach().cpu().numpy().copy()


def make_net(act, dim=3, cin=32, cout=4, nf=16, seed=0):
    torch.manual_seed(seed)
    return implicit_net.ImNet(dim=dim, in_features=cin, out_features=cout, nf=nf,
                              activation=nonlinearities.NONLINEARITIES[act])


def net_params(net):
    d = {}
    for k in range(6):
        d[f"w{k}"] = t2n(net.fc[k].weight)
        d[f"b{k}"] = t2n(net.fc[k].bias)
    return d


def g3_corners():
    grid = torch.arange(8).float().view(1, 2, 2, 2, 1)
    q = torch.tensor([[[0.25, 0.5, 0.75]]])
    v, w, r = rgi.regular_nd_grid_interpolation_coefficients(grid, q, 0., 1.)
    np.savez(os.path.join(OUT, "g3_corners.npz"), grid=t2n(grid), q=t2n(q), v=t2n(v), w=t2n(w), r=t2n(r))


def g4_imnet():
    d = {}
    g = torch.Generator().manual_seed(11)
    x = torch.randn(256, 35, generator=g)
    d["x"] = t2n(x)
    for act in ACTS:
        net = make_net(act, nf=16, seed=5)
        if act == "swish":
            with torch.no_grad():
                net.activ.beta.fill_(1.3)
        if act == ACTS[0]:
            d.update(net_params(net))
        d[f"y_{act}"] = t2n(net(x))
    np.savez(os.path.join(OUT, "g4_imnet.npz"), **d)


def g5_composite():
    """LIG + RB2 residuals + losses + gradients on a small latent grid, per activation."""
    g = torch.Generator().manual_seed(21)
    lat0 = 0.5 * torch.randn(1, 4, 8, 8, 32, generator=g)
    pts = 0.02 + 0.96 * torch.rand(1, 256, 3, generator=g)
    tgt = torch.randn(1, 256, 4, generator=g)
    d = dict(latent=t2n(lat0), pts=t2n(pts), targets=t2n(tgt), mean=np.array(MEAN), std=np.array(STD),
             t_crop=2.0, z_crop=1.0, x_crop=1.0, alpha_reg=1.0, alpha_pde=0.0125)
    for act in ACTS:
        net = make_net(act, nf=16, seed=7)
        if act == "swish":
            with torch.no_grad():
                net.activ.beta.fill_(1.3)
        if act == ACTS[0]:
            d.update(net_params(net))
        lat = lat0.clone().requires_grad_(True)
        layer = physics.get_rb2_pde_layer(mean=MEAN, std=STD, t_crop=2., z_crop=1., x_crop=1., use_continuity=True)
        layer.update_forward_method(lambda p: lig.query_local_implicit_grid(net, lat, p, 0., 1.))
        pred, res = layer(pts.clone(), return_residue=True)
        reg = torch.nn.functional.l1_loss(pred, tgt)
        st = torch.stack(list(res.values()), 0)
        pl = torch.nn.functional.l1_loss(st, torch.zeros_like(st))
        loss = 1.0 * reg + 0.0125 * pl
        loss.backward()
        d[f"{act}_pred"] = t2n(pred)
        for k, v in res.items():
            d[f"{act}_res_{k}"] = t2n(v)
        d[f"{act}_reg_loss"] = t2n(reg)
        d[f"{act}_pde_loss"] = t2n(pl)
        d[f"{act}_dlatent"] = t2n(lat.grad)
        full = act in ("softplus", "leakyrelu")
        for k in range(6):
            if full or k >= 3:
                d[f"{act}_dw{k}"] = t2n(net.fc[k].weight.grad)
                d[f"{act}_db{k}"] = t2n(net.fc[k].bias.grad)
            else:
                d[f"{act}_dw{k}_norm"] = t2n(net.fc[k].weight.grad.norm())
                d[f"{act}_db{k}_norm"] = t2n(net.fc[k].bias.grad.norm())
        if act == "swish":
            d["swish_dbeta"] = t2n(net.activ.beta.grad)
    np.savez_compressed(os.path.join(OUT, "g5_composite.npz"), **d)


def g6_cell_index():
    """int64 ind0 for uniform + adversarial (near cell face) points on three grid sizes."""
    d = {}
    for seed, (tag, size) in enumerate((("c1", (16, 32, 32)), ("c2", (32, 128, 128)), ("c4", (64, 256, 256)))):
        g = torch.Generator().manual_seed(61 + seed)
        n = 4096
        pts = torch.rand(1, n, 3, generator=g)
        # adversarial: k*cubesize nudged by a few ulps either way, plus the box faces
        sz = torch.tensor(size).float()
        cube = 1.0 / (sz - 1)
        k = torch.stack([torch.randint(0, s, (n,), generator=g) for s in size], -1).float()
        face = k * cube
        ulps = torch.randint(-3, 4, (n, 3), generator=g).float()
        face = face + ulps * torch.finfo(torch.float32).eps * face.abs().clamp(min=1e-3)
        pts = torch.cat([pts, face[None], torch.tensor([[[0., 0., 0.], [1., 1., 1.], [-0.5, 0.5, 1.5]]])], 1)
        grid = torch.zeros(1, *size, 1)
        # the reference does not return ind0; recover it exactly from x_relative of corner 0 is lossy, so
        # recompute with the reference's own expression sequence (regular_nd_grid_interpolation.py:48-52)
        xmin = torch.zeros(3)
        xmax = torch.ones(3)
        eps = 1e-6 * (xmax - xmin)
        q = rgi.clip_tensor(pts, xmin + eps, xmax - eps)
        cubesize = (xmax - xmin) / (sz - 1)
        ind0 = torch.floor(q / cubesize).long()
        # cross-check against the reference function through a grid whose node value is its own index
        idg = torch.stack(torch.meshgrid([torch.arange(s) for s in size], indexing="ij"), -1).float()[None]
        v, _, _ = rgi.regular_nd_grid_interpolation_coefficients(idg, pts, 0., 1.)
        assert torch.equal(v[:, :, 0, :].long(), ind0)
        d[f"{tag}_pts"] = t2n(pts)
        d[f"{tag}_ind0"] = t2n(ind0).astype(np.int16)
        d[f"{tag}_size"] = np.array(size)
    np.savez_compressed(os.path.join(OUT, "g6_cell_index.npz"), **d)


def g9_generic():
    """User-string PDELayer (C5-style advection-diffusion, 5 channels) incl. products and a mixed derivative."""
    g = torch.Generator().manual_seed(31)
    lat = 0.5 * torch.randn(2, 6, 5, 7, 32, generator=g)
    pts = 0.02 + 0.96 * torch.rand(2, 128, 3, generator=g)
    net = make_net("softplus", cout=5, nf=16, seed=9)
    eqs = {
        "adv_diff": "dif(c,t)+u*dif(c,x)+v*dif(c,y)-0.01*(dif(dif(c,x),x)+dif(dif(c,y),y))",
        "prod_rule": "dif(u*c,x)+dif(v*c,y)",
        "mixed": "dif(dif(c,x),y)-w*p",
        "explicit_x": "x*dif(p,x)+t*dif(dif(p,t),t)",
    }
    layer = rpde.PDELayer("x, y, t", "c, u, v, w, p")
    for k, s in eqs.items():
        layer.add_equation(s, k)
    layer.update_forward_method(lambda p: lig.query_local_implicit_grid(net, lat, p, 0., 1.))
    pred, res = layer(pts.clone())
    d = dict(latent=t2n(lat), pts=t2n(pts), pred=t2n(pred), names=np.array(list(eqs.keys())),
             eqs=np.array(list(eqs.values())))
    d.update(net_params(net))
    for k, v in res.items():
        d[f"res_{k}"] = t2n(v)
    np.savez_compressed(os.path.join(OUT, "g9_generic.npz"), **d)


def g10_lig4d():
    g = torch.Generator().manual_seed(41)
    lat = torch.rand(2, 4, 5, 3, 6, 8, generator=g)
    pts = torch.rand(2, 64, 4, generator=g)
    net = make_net("leakyrelu", dim=4, cin=8, cout=3, nf=4, seed=13)
    y = lig.query_local_implicit_grid(net, lat, pts, 0., 1.)
    d = dict(latent=t2n(lat), pts=t2n(pts), y=t2n(y))
    d.update(net_params(net))
    # a non-unit box as well (xmin must be 0: quirk a-Q1)
    xmax = (2.0, 1.0, 4.0, 0.5)
    pts2 = pts * torch.tensor(xmax)
    y2 = lig.query_local_implicit_grid(net, lat, pts2, (0., 0., 0., 0.), xmax)
    d.update(pts2=t2n(pts2), y2=t2n(y2), xmax=np.array(xmax))
    np.savez_compressed(os.path.join(OUT, "g10_lig4d.npz"), **d)


def g1_interp():
    """Interpolation on random (non-analytic) grids for d = 1..4, with non-unit boxes."""
    g = torch.Generator().manual_seed(51)
    d = {}
    for dim, shape in ((1, (9,)), (2, (5, 7)), (3, (4, 6, 5)), (4, (3, 4, 5, 3))):
        grid = torch.randn(2, *shape, 3, generator=g)
        xmax = tuple(float(k + 1) for k in range(dim))
        pts = torch.rand(2, 40, dim, generator=g) * torch.tensor(xmax)
        out = rgi.regular_nd_grid_interpolation(grid, pts, tuple(0. for _ in range(dim)), xmax)
        v, w, r = rgi.regular_nd_grid_interpolation_coefficients(grid, pts, tuple(0. for _ in range(dim)), xmax)
        d.update({f"d{dim}_grid": t2n(grid), f"d{dim}_pts": t2n(pts), f"d{dim}_out": t2n(out),
                  f"d{dim}_xmax": np.array(xmax), f"d{dim}_v": t2n(v), f"d{dim}_w": t2n(w), f"d{dim}_r": t2n(r)})
    np.savez_compressed(os.path.join(OUT, "g1_interp.npz"), **d)


def g7_unet():
    """UNet3d(igres=(4,8,8), nf=16, mf=32): train-mode output + running stats + gradients, eval-mode output."""
    import unet3d  # the reference's (needs the numpy.int shim above)
    torch.manual_seed(3)
    net = unet3d.UNet3d(in_features=4, out_features=32, igres=(4, 8, 8), nf=16, mf=32)
    g = torch.Generator().manual_seed(71)
    x = torch.randn(2, 4, 4, 8, 8, generator=g).requires_grad_(True)
    cot = torch.randn(2, 32, 4, 8, 8, generator=g)
    d = {"state/" + k: t2n(v) for k, v in net.state_dict().items()}
    net.train()
    y = net(x)
    (y * cot).sum().backward()
    d.update(x=t2n(x), cot=t2n(cot), y_train=t2n(y), dx=t2n(x.grad))
    for k, v in net.state_dict().items():
        if "running" in k or "num_batches" in k:
            d["after/" + k] = t2n(v)
    for name in ("conv_in.conv2.weight", "conv_in.bn1.weight", "conv_mid.conv2.weight", "conv_out.shortcut.weight",
                 "conv_out.shortcut.bias", "down_modules.1.conv2.weight", "up_modules.0.conv1.weight"):
        d["grad/" + name] = t2n(dict(net.named_parameters())[name].grad)
    net.eval()
    with torch.no_grad():
        d["y_eval"] = t2n(net(x))
    np.savez_compressed(os.path.join(OUT, "g7_unet.npz"), **d)


def _rb2_step(unet, net, crop, pts, tgt, alpha_pde=0.0125, latent=None):
    """experiments/rb2d/train.py:58-77 around the imported reference modules (tensor bounds as train.py:48-49)."""
    xmin, xmax = torch.zeros(3, dtype=pts.dtype), torch.ones(3, dtype=pts.dtype)
    lat = unet(crop).permute(0, 2, 3, 4, 1) if latent is None else latent
    layer = physics.get_rb2_pde_layer(mean=MEAN, std=STD, t_crop=2., z_crop=1., x_crop=1., use_continuity=True)
    layer.update_forward_method(lambda p: lig.query_local_implicit_grid(net, lat, p, xmin, xmax))
    pred, res = layer(pts.clone(), return_residue=True)
    reg = torch.nn.functional.l1_loss(pred, tgt)
    st = torch.stack(list(res.values()), 0)
    pl = torch.nn.functional.l1_loss(st, torch.zeros_like(st))
    loss = 1.0 * reg + alpha_pde * pl
    loss.backward()
    return lat, pred, res, reg, pl, loss


def g5b_nf32():
    """G5 at the BENCHMARKED network width (nf = 32; softplus -> combined stream S = (3,1), leaky-relu -> S = (3,0)):
    inputs and weights come from det_init (regenerated by the test), only the reference's outputs are stored."""
    lat0 = 0.5 * det_init.normal(521, 1, 4, 8, 8, 32)
    pts = det_init.uniform(522, 1, 256, 3, lo=0.02, hi=0.98)
    tgt = det_init.normal(523, 1, 256, 4)
    d = {}
    for act in ("softplus", "leakyrelu"):
        net = det_init.fill_module_(make_net(act, nf=32, seed=0), 524)
        lat = lat0.clone().requires_grad_(True)
        _, pred, res, reg, pl, _ = _rb2_step(None, net, None, pts, tgt, latent=lat)
        d[f"{act}_pred"] = t2n(pred)
        for k, v in res.items():
            d[f"{act}_res_{k}"] = t2n(v)
        d[f"{act}_reg_loss"], d[f"{act}_pde_loss"] = t2n(reg), t2n(pl)
        d[f"{act}_dlatent"] = t2n(lat.grad)
        for k in range(6):
            gw, gb = net.fc[k].weight.grad, net.fc[k].bias.grad
            d[f"{act}_db{k}"] = t2n(gb)
            d[f"{act}_dw{k}_norm"] = t2n(gw.norm())
            d[f"{act}_dw{k}"] = t2n(gw[::3] if k < 2 else gw)      # fc0 / fc1: every third output row (fixture size)
    np.savez_compressed(os.path.join(OUT, "g5b_nf32.npz"), **d)


def g8_c1_step():
    """BASELINE configs[0] end to end: the reference's UNet3d(igres=(16,32,32), nf=16, mf=256) in
    training mode -> latent grid -> LIG + ImNet(nf=32) + RB2 residuals on 4096 points -> L1 losses -> backward to every
    UNet and ImNet parameter.  Weights / inputs from det_init; stored: losses, a slice of predictions / residuals / the
    latent grid, the gradient norm of EVERY parameter and a few small gradients in full.
    The same step is also run with the reference cast to float64 (``*_f64`` keys): BatchNorm over the 8 voxels of the
    deepest level amplifies fp32 rounding (the reference's fp32 latent grid differs from its own fp64 one by ~9e-4 of
    the maximum), so parity tolerances for this config are stated relative to the reference's own fp32-vs-fp64
    distance."""
    import unet3d
    crop = det_init.normal(800, 1, 4, 16, 32, 32)
    pts = det_init.uniform(801, 1, 4096, 3)
    tgt = det_init.normal(802, 1, 4096, 4)
    d = {}
    full = ("conv_in.conv2.weight", "conv_in.bn1.weight", "conv_in.shortcut.bias", "conv_out.conv3.weight",
            "conv_out.bn3.bias", "down_modules.0.conv2.weight", "up_modules.3.conv1.bias")
    for act in ("softplus", "leakyrelu"):
        for dt, sfx in ((torch.float32, ""), (torch.float64, "_f64")):
            unet = det_init.fill_module_(unet3d.UNet3d(in_features=4, out_features=32, igres=(16, 32, 32), nf=16, mf=256), 810)
            net = det_init.fill_module_(make_net(act, nf=32, seed=0), 820)
            unet.to(dt).train()
            net.to(dt).train()
            lat, pred, res, reg, pl, loss = _rb2_step(unet, net, crop.to(dt), pts.to(dt), tgt.to(dt))
            d[f"{act}_reg_loss{sfx}"], d[f"{act}_pde_loss{sfx}"], d[f"{act}_loss{sfx}"] = t2n(reg), t2n(pl), t2n(loss)
            d[f"{act}_pred{sfx}"] = t2n(pred[:, :512])
            for k, v in res.items():
                d[f"{act}_res_{k}{sfx}"] = t2n(v[:, :512])
            if act == "softplus":
                d["latent_slice" + sfx] = t2n(lat[0, ::4, ::8, ::8, :])
            names, norms = [], []
            for prefix, mod in (("unet.", unet), ("imnet.", net)):
                seen = set()
                for name, p in mod.named_parameters():
                    if id(p) in seen:
                        continue
                    seen.add(id(p))
                    names.append(prefix + name)
                    norms.append(float(p.grad.norm()))
            d[f"{act}_grad_names"], d[f"{act}_grad_norms{sfx}"] = np.array(names), np.array(norms)
            if act == "softplus":
                up = dict(unet.named_parameters())
                for name in full:
                    d["grad%s/unet.%s" % (sfx, name)] = t2n(up[name].grad)
                d["grad%s/imnet.fc5.weight" % sfx] = t2n(net.fc5.weight.grad)
                d["grad%s/imnet.fc4.bias" % sfx] = t2n(net.fc4.bias.grad)
                d["grad%s/imnet.fc2.bias" % sfx] = t2n(net.fc2.bias.grad)
                sd = unet.state_dict()
                for name in ("conv_in.bn1.running_mean", "conv_in.bn1.running_var", "conv_mid.bn2.running_var",
                             "conv_out.bn3.running_mean"):
                    d["after%s/%s" % (sfx, name)] = t2n(sd[name])
    np.savez_compressed(os.path.join(OUT, "g8_c1_step.npz"), **d)


def n3_dataloader():
    """N3: the reference's RB2DataLoader.__getitem__ (dataloader_spacetime.py:118-171) on a small synthetic .npz --
    low-res crop, sampled points and interpolated targets for every lres_filter / lres_interp / normalisation setting.
    The synthetic dataset itself is regenerated by the test from the stored seed."""
    import tempfile
    import dataloader_spacetime as dl
    T, X, Z = 12, 40, 24
    rng = np.random.default_rng(900)
    arrs = {k: rng.standard_normal((T, X, Z)).astype(np.float32) for k in ("p", "b", "u", "w")}   # stored [t, x, z]
    d = dict(T=T, X=X, Z=Z, data_seed=900, idx=np.array([0, 37, 1000]))
    with tempfile.TemporaryDirectory() as tmp:
        np.savez(os.path.join(tmp, "synth.npz"), **arrs)
        cfgs = [("none", "linear", False), ("none", "linear", True), ("gaussian", "linear", False),
                ("uniform", "linear", True), ("maximum", "linear", False), ("median", "linear", False),
                ("none", "nearest", False)]
        for filt, interp, norm in cfgs:
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                ds = dl.RB2DataLoader(data_dir=tmp, data_filename="synth.npz", nx=16, nz=16, nt=8,
                                      n_samp_pts_per_crop=64, downsamp_xz=4, downsamp_t=2, normalize_output=norm,
                                      lres_filter=filt, lres_interp=interp)
            tag = f"{filt}_{interp}_{int(norm)}"
            d["len"] = len(ds)
            d["mean"], d["std"] = ds.channel_mean, ds.channel_std
            for idx in (0, 37, 1000):
                np.random.seed(7 + idx)
                lres, pc, pv = ds[idx]
                d[f"{tag}/{idx}/lres"], d[f"{tag}/{idx}/pc"], d[f"{tag}/{idx}/pv"] = lres, pc, pv
    np.savez_compressed(os.path.join(OUT, "n3_dataloader.npz"), **d)


def n4_checkpoint():
    """N4: a checkpoint WRITTEN BY THE REFERENCE (src/train_utils.py:13-30 with the dict of train.py:390-397) after one
    Adam step, plus what the reference computes when it resumes from it (train.py:339-350) and takes a second step:
    forward output, losses and a few updated parameters.  The .pth.tar file is data (a pickled dict of tensors)."""
    import logging
    import unet3d
    import train_utils
    crop = det_init.normal(950, 1, 4, 4, 8, 8)
    pts = det_init.uniform(951, 1, 128, 3, lo=0.02, hi=0.98)
    tgt = det_init.normal(952, 1, 128, 4)
    lr, clip = 1e-2, 0.1

    def build():
        unet = det_init.fill_module_(unet3d.UNet3d(in_features=4, out_features=32, igres=(4, 8, 8), nf=16, mf=16), 960)
        net = det_init.fill_module_(make_net("softplus", nf=16, seed=0), 970)
        opt = torch.optim.Adam(list(unet.parameters()) + list(net.parameters()), lr=lr)
        return unet, net, opt

    def step(unet, net, opt):
        unet.train()
        opt.zero_grad()
        lat, pred, res, reg, pl, loss = _rb2_step(unet, net, crop, pts, tgt)
        torch.nn.utils.clip_grad_value_(unet.parameters(), clip)
        torch.nn.utils.clip_grad_value_(net.parameters(), clip)
        opt.step()
        return pred, reg, pl

    unet, net, opt = build()
    step(unet, net, opt)
    state = {"epoch": 1, "unet_state_dict": unet.state_dict(), "imnet_state_dict": net.state_dict(),
             "optim_state_dict": opt.state_dict(), "tracked_stats": 0.25, "global_step": np.ones(1, dtype=np.uint32)}
    train_utils.save_checkpoint(state, False, 1, os.path.join(OUT, "n4_ckpt"), "_pdenet", logging.getLogger("golden"))
    # resume exactly as train.py:339-350 does, then one more step
    unet2, net2, opt2 = build()
    ck = torch.load(os.path.join(OUT, "n4_ckpt_pdenet_001.pth.tar"), weights_only=False)
    unet2.load_state_dict(ck["unet_state_dict"])
    net2.load_state_dict(ck["imnet_state_dict"])
    opt2.load_state_dict(ck["optim_state_dict"])
    pred, reg, pl = step(unet2, net2, opt2)
    d = dict(lr=lr, clip=clip, pred=t2n(pred), reg_loss=t2n(reg), pde_loss=t2n(pl))
    for name in ("conv_in.conv1.weight", "conv_in.bn1.weight", "conv_out.conv3.weight", "conv_mid.conv2.bias"):
        d["after/unet." + name] = t2n(dict(unet2.named_parameters())[name])
    for name in ("fc0.bias", "fc3.weight", "fc5.weight"):
        d["after/imnet." + name] = t2n(dict(net2.named_parameters())[name])
    d["after/unet.conv_in.bn1.running_mean"] = t2n(unet2.state_dict()["conv_in.bn1.running_mean"])
    np.savez_compressed(os.path.join(OUT, "n4_resume.npz"), **d)


if __name__ == "__main__":
    only = sys.argv[1:]
    if only:
        for name in only:
            globals()[name]()
        sys.exit(0)
    g5b_nf32()
    g8_c1_step()
    n3_dataloader()
    n4_checkpoint()
    g1_interp()
    g3_corners()
    g4_imnet()
    g5_composite()
    g6_cell_index()
    g9_generic()
    g10_lig4d()
    g7_unet()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
