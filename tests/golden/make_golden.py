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


if __name__ == "__main__":
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
