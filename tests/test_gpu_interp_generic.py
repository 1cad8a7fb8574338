"""GPU parity of the remaining section-8 rows: plain n-d interpolation kernels (a2/a3, dim 1..4), bit-exact cell
selection (G6), user-string equations through the HIP jet path (G9, config-5 style), 4-D local implicit grid (G10)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize("dim", [1, 2, 3, 4])
def test_interp_kernels_match_reference(hiplib, golden_dir, dim):
    from space_time_pde_amd import regular_nd_grid_interpolation as rgi
    d = _load(golden_dir, "g1_interp.npz")
    grid = torch.from_numpy(d["d%d_grid" % dim]).to(DEV).requires_grad_(True)
    pts = torch.from_numpy(d["d%d_pts" % dim]).to(DEV)
    xmax = tuple(float(v) for v in d["d%d_xmax" % dim])
    xmin = tuple(0. for _ in range(dim))
    assert rgi._hip_eligible(grid, pts)
    v, w, r = rgi.regular_nd_grid_interpolation_coefficients(grid, pts, xmin, xmax)
    assert np.array_equal(v.detach().cpu().numpy(), d["d%d_v" % dim])      # gather: bit exact
    assert np.array_equal(w.cpu().numpy(), d["d%d_w" % dim])               # same fp32 op sequence: bit exact
    assert np.array_equal(r.cpu().numpy(), d["d%d_r" % dim])
    out = rgi.regular_nd_grid_interpolation(grid, pts, xmin, xmax)
    np.testing.assert_allclose(out.detach().cpu().numpy(), d["d%d_out" % dim], rtol=1e-6, atol=1e-6)
    # gradient w.r.t. the grid (scatter-add kernel) vs the composed torch formulation
    cot = torch.randn_like(out)
    (out * cot).sum().backward()
    g2 = grid.detach().clone().requires_grad_(True)
    v2, w2, _ = rgi._coefficients_autograd(g2, pts, xmin, xmax)
    ((v2 * w2.unsqueeze(-1)).sum(-2) * cot).sum().backward()
    assert (grid.grad - g2.grad).abs().max().item() < 1e-5 * g2.grad.abs().max().item()


def test_reference_kat_on_gpu(hiplib):
    """src/regular_nd_grid_interpolation_test.py:12-40 on the HIP kernels."""
    from space_time_pde_amd import regular_nd_grid_interpolation as rgi
    g = torch.Generator().manual_seed(0)
    for dim in (1, 2, 3):
        axes = torch.meshgrid(*[torch.arange(11)] * dim, indexing="ij")
        grid = torch.stack(axes, -1).float().unsqueeze(0).to(DEV)
        q = torch.rand(1, 100, dim, generator=g)
        out = rgi.regular_nd_grid_interpolation(grid, q.to(DEV), 0., 1.)
        np.testing.assert_allclose(out.cpu().numpy(), (10. * q).numpy(), atol=1e-4)


@pytest.mark.parametrize("tag", ["c1", "c2", "c4"])
def test_cell_index_bit_exact_in_gather_kernel(hiplib, golden_dir, tag):
    """G6: the gather kernel's floor(clip(q)/cubesize) equals the reference's int64 ind0, incl. points a few ulps
    either side of cell faces and outside the box (checked through the cell index the kernel writes)."""
    from space_time_pde_amd import _lib
    from space_time_pde_amd.lig_jet import box_constants
    d = _load(golden_dir, "g6_cell_index.npz")
    size = tuple(int(v) for v in d[tag + "_size"])
    pts = torch.from_numpy(d[tag + "_pts"]).reshape(-1, 3)
    if pts.shape[0] % 2:
        pts = torch.cat([pts, pts[-1:]])
    P = pts.shape[0]
    lo, hi, cube = box_constants(size, 0., 1.)
    gd = _lib.GatherDesc()
    gd.P, gd.N, gd.B, gd.n0, gd.n1, gd.n2, gd.C, gd.p_base = P, P, 1, size[0], size[1], size[2], 1, 0
    for k in range(3):
        gd.lo_c[k], gd.hi_c[k], gd.cube[k] = lo[k], hi[k], cube[k]
    latent = torch.zeros(1, *size, 1, device=DEV)
    X = torch.empty(P // 2 * 3 * 256, device=DEV)
    coef = torch.empty(P * 16, device=DEV)
    cell = torch.empty(P, device=DEV, dtype=torch.int32)
    _lib.check(hiplib.stpde_lig_gather(C.byref(gd), _lib.ptr(pts.to(DEV)), _lib.ptr(latent), _lib.ptr(X), None,
                                       _lib.ptr(coef), _lib.ptr(cell), None, _lib.stream_ptr()))
    ind0 = torch.from_numpy(d[tag + "_ind0"].astype(np.int64)).reshape(-1, 3)
    want = (ind0[:, 0] * size[1] + ind0[:, 1]) * size[2] + ind0[:, 2]
    got = cell.cpu().long()[:want.shape[0]]
    assert torch.equal(got, want)


@pytest.mark.parametrize("s2", [4, 6])
@pytest.mark.parametrize("prec", ["fp32", "fp32x3"])
def test_generic_equations_config5_style(hiplib, golden_dir, prec, s2, monkeypatch):
    """G9: 5-channel user-string PDELayer (products, mixed 2nd derivative, explicit coordinates) on the HIP jet path."""
    # the same test, same tolerances, with the wide layers' products as exact-split bf16 MFMAs ("fp32x3")
    from space_time_pde_amd import lig_jet as _lj
    monkeypatch.setattr(_lj, "mlp_precision", prec)
    monkeypatch.setenv("STPDE_S34", "1" if s2 == 4 else "0")      # the (3,4) stream set (default) / padded to (3,6)
    from space_time_pde_amd import _lib, implicit_net, local_implicit_grid as lig, pde
    d = _load(golden_dir, "g9_generic.npz")
    net = implicit_net.ImNet(dim=3, in_features=32, out_features=5, nf=16, activation=torch.nn.Softplus).to(DEV)
    with torch.no_grad():
        for k in range(6):
            net.fc[k].weight.copy_(torch.from_numpy(d["w%d" % k]))
            net.fc[k].bias.copy_(torch.from_numpy(d["b%d" % k]))
    lat = torch.from_numpy(d["latent"]).to(DEV)
    pts = torch.from_numpy(d["pts"]).to(DEV)
    layer = pde.PDELayer("x, y, t", "c, u, v, w, p")
    for name, eq in zip(d["names"], d["eqs"]):
        layer.add_equation(str(eq), str(name))
    layer.update_forward_method(lambda q: lig.query_local_implicit_grid(net, lat, q, 0., 1.))
    n0 = lig.stats["hip_jet_calls"]
    with _lib.dispatch_trace() as tr:
        pred, res = layer(pts)
    assert lig.stats["hip_jet_calls"] == n0 + 1                    # 4 second-order pairs -> the (3,4) stream set
    assert tr.has("S1 = 3, S2 = %d" % s2), "\n".join(tr.kernels)
    np.testing.assert_allclose(pred.detach().cpu().numpy(), d["pred"], rtol=2e-5, atol=2e-6)
    for name in d["names"]:
        ref = d["res_" + str(name)]
        tol = 2e-4 if str(name) == "explicit_x" else 3e-5
        assert np.abs(res[str(name)].detach().cpu().numpy() - ref).max() < tol * np.abs(ref).max(), name


def test_lig_4d_generic_decoder_path(hiplib, golden_dir):
    """G10: 4-D grid (16 corners) goes through the HIP coefficient kernel + the decoder module."""
    from space_time_pde_amd import implicit_net, local_implicit_grid as lig
    d = _load(golden_dir, "g10_lig4d.npz")
    net = implicit_net.ImNet(dim=4, in_features=8, out_features=3, nf=4, activation=torch.nn.LeakyReLU).to(DEV)
    with torch.no_grad():
        for k in range(6):
            net.fc[k].weight.copy_(torch.from_numpy(d["w%d" % k]))
            net.fc[k].bias.copy_(torch.from_numpy(d["b%d" % k]))
        y = lig.query_local_implicit_grid(net, torch.from_numpy(d["latent"]).to(DEV), torch.from_numpy(d["pts"]).to(DEV),
                                          0., 1.)
        xmax = tuple(float(v) for v in d["xmax"])
        y2 = lig.query_local_implicit_grid(net, torch.from_numpy(d["latent"]).to(DEV),
                                           torch.from_numpy(d["pts2"]).to(DEV), (0., 0., 0., 0.), xmax)
    np.testing.assert_allclose(y.cpu().numpy(), d["y"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(y2.cpu().numpy(), d["y2"], rtol=1e-5, atol=1e-6)


def test_large_size_properties(hiplib):
    """Size-independent properties at BASELINE sizes: linearity of the jets in the output layer and chunk invariance
    (2^17 points on the [1,32,128,128,32] grid evaluated in one chunk and in 8 chunks give identical values)."""
    from space_time_pde_amd import implicit_net, lig_jet
    g = torch.Generator().manual_seed(0)
    lat = (0.5 * torch.randn(1, 32, 128, 128, 32, generator=g)).to(DEV)
    pts = torch.rand(1, 1 << 17, 3, generator=g).to(DEV)
    torch.manual_seed(0)
    net = implicit_net.ImNet(nf=32, activation=torch.nn.Softplus).to(DEV)
    pairs = ((1, 1), (2, 2))
    with torch.no_grad():
        a, _ = lig_jet.lig_jets(net, lat, pts, 0., 1., True, pairs, chunk_points=1 << 17)
        b, _ = lig_jet.lig_jets(net, lat, pts, 0., 1., True, pairs, chunk_points=1 << 14)
        assert torch.equal(a, b)
        # fc5 is linear: scaling its weights and bias scales every stream
        net.fc5.weight.mul_(2.0)
        net.fc5.bias.mul_(2.0)
        c, _ = lig_jet.lig_jets(net, lat, pts, 0., 1., True, pairs, chunk_points=1 << 17)
        assert (c - 2 * a).abs().max().item() <= 1e-5 * a.abs().max().item()
    assert torch.isfinite(a).all()


@pytest.mark.parametrize("prec,grid", [("fp32", (32, 128, 128)), ("fp32x3", (32, 128, 128)), ("bf16", (32, 128, 128)),
                                       ("bf16", (64, 256, 256))])
def test_full_size_step_subset_vs_oracle_and_additivity(hiplib, prec, grid, monkeypatch):
    """(parametrised over the three MFMA operand modes, VERDICT r2 #3b: fp32x3 with the fp32 tolerances, bf16 with the
    mode's 3e-2 Frobenius bound; VERDICT r3 #1b: the bf16 mode also on BASELINE configs[3]'s OWN latent grid
    [1,64,256,256,32] = 512 MiB.)  BASELINE configs[1] / [3] size (2^20 points, RB2 + continuity, softplus):
    (a) points are independent, so pred / residuals of a random subset must equal the CPU oracle run on just that subset;
    (b) gradients are additive over points: grads(all points) == grads(first half) + grads(second half)."""
    from oracle import cpu_ref
    from space_time_pde_amd import implicit_net, lig_jet, local_implicit_grid as lig, physics
    monkeypatch.setattr(lig_jet, "mlp_precision", prec)
    g = torch.Generator().manual_seed(0)
    N = 1 << 20
    lat0 = (0.5 * torch.randn(1, *grid, 32, generator=g))
    pts = torch.rand(1, N, 3, generator=g)
    torch.manual_seed(0)
    net = implicit_net.ImNet(nf=32, activation=torch.nn.Softplus).to(DEV)
    kw = dict(mean=(0.01, 0.0, 0.02, -0.01), std=(0.05, 0.3, 0.15, 0.12), t_crop=2., z_crop=1., x_crop=1.,
              use_continuity=True)
    layer = physics.get_rb2_pde_layer(**kw)
    latd, ptsd = lat0.to(DEV), pts.to(DEV)

    def run(sl):
        lat = latd.clone().requires_grad_(True)
        for p in net.parameters():
            p.grad = None
        layer.update_forward_method(lambda q: lig.query_local_implicit_grid(net, lat, q, 0., 1.))
        pred, res = layer(ptsd[:, sl].contiguous())
        loss = pred.abs().sum() / N + 0.0125 * torch.stack(list(res.values()), 0).abs().sum() / N
        loss.backward()
        return pred.detach(), {k: v.detach() for k, v in res.items()}, lat.grad, [p.grad.clone() for p in net.parameters()]

    pred, res, gl, gp = run(slice(0, N))
    # (a) subset vs oracle
    sel = torch.randperm(N, generator=g)[:1024]
    params = [(net.fc[k].weight.detach().cpu(), net.fc[k].bias.detach().cpu()) for k in range(6)]
    ref = cpu_ref.lig_pde_step(params, "softplus", lat0, pts[:, sel], torch.zeros(1, 1024, 4), cpu_ref.rb2_oracle(**kw),
                               backward=False)
    if prec == "bf16":
        def nrm(a, b):
            return (a.double() - b.double()).norm().item() / b.double().norm().item()
        assert nrm(pred[:, sel].cpu(), ref["pred"]) < 3e-2
        for k, v in ref["residues"].items():
            assert nrm(res[k][:, sel].cpu(), v) < 3e-2, k
    else:
        assert (pred[:, sel].cpu() - ref["pred"]).abs().max().item() < 2e-5 * ref["pred"].abs().max().item()
    # second derivatives carry 1/cubesize^2 = 127^2: the fp32 reference path itself is only good to ~2e-4 of the
    # residual scale per point here (SURVEY a-Q8: fp32 vs fp64 of the reference, max-rel 1.7e-4), so: tight in the
    # bulk, bounded in the tail
    for k, v in ref["residues"].items():
        if prec == "bf16":
            break
        err = (res[k][:, sel].cpu() - v).abs() / v.abs().max()
        assert err.median().item() < 1e-5 and err.max().item() < 1e-3, (k, err.max().item())
    # (b) additivity of the gradients over the two halves
    _, _, gl1, gp1 = run(slice(0, N // 2))
    _, _, gl2, gp2 = run(slice(N // 2, N))
    assert (gl - (gl1 + gl2)).abs().max().item() < 1e-4 * gl.abs().max().item()
    for a, b, c in zip(gp, gp1, gp2):
        assert (a - (b + c)).abs().max().item() < 2e-4 * a.abs().max().item()


def test_config5_full_size_properties(hiplib):
    """VERDICT r3 #1c -- BASELINE configs[4] at ITS size: the 5-output user-string equation set (bench.py C5_EQS = the strings
    of fixture G9: products, a mixed second derivative, explicit coordinates -> stream set (3,4), S = 8; round 4: padded to (3,6)) on 2^20 points over
    the [1,32,128,128,32] grid, ImNet nf = 32, through whichever of stash / recomputation the memory plan picks:
    (a) a random subset of the points equals the oracle's reverse-sweep autograd on just that subset;
    (b) chunk invariance: one 2^20-point launch chunk and 2^18-point chunks give bit-identical jets;
    (c) gradients are additive over the points: grads(all) == grads(first half) + grads(second half)."""
    import bench
    from oracle import cpu_ref as O
    from space_time_pde_amd import _lib, implicit_net, lig_jet, local_implicit_grid as lig, pde
    g = torch.Generator().manual_seed(5)
    N = 1 << 20
    lat0 = 0.5 * torch.randn(1, 32, 128, 128, 32, generator=g)
    pts = torch.rand(1, N, 3, generator=g)
    torch.manual_seed(5)
    net = implicit_net.ImNet(dim=3, in_features=32, out_features=5, nf=32, activation=torch.nn.Softplus).to(DEV)
    layer = bench.c5_layer(pde)
    latd, ptsd = lat0.to(DEV), pts.to(DEV)
    cot = torch.randn(1, N, 5, generator=g).to(DEV)

    def run(sl, trace=False):
        lat = latd.clone().requires_grad_(True)
        for p in net.parameters():
            p.grad = None
        layer.update_forward_method(lambda q: lig.query_local_implicit_grid(net, lat, q, 0., 1.))
        n0 = lig.stats["hip_jet_calls"]
        pred, res = layer(ptsd[:, sl].contiguous())
        assert lig.stats["hip_jet_calls"] == n0 + 1
        loss = (pred * cot[:, sl]).sum() / N + 0.0125 * torch.stack(list(res.values()), 0).abs().sum() / N
        loss.backward()
        torch.cuda.synchronize()
        return pred.detach(), {k: v.detach() for k, v in res.items()}, lat.grad, [p.grad.clone() for p in net.parameters()]

    r0 = lig_jet.stats["recompute_steps"]
    with _lib.dispatch_trace() as tr:
        pred, res, gl, gp = run(slice(0, N))
    # round 5: exactly the four second derivatives the strings name (xx, yy, xy, tt) -> the (3,4) stream set, S = 8, with the
    # fused fc3 -> fc5 kernels; nothing of the padded (3,6) set
    assert tr.has("S1 = 3, S2 = 4") and tr.has("k_residual_bwd") and tr.has("k_tail_fwd") and tr.has("k_tail_bwd"), \
        "\n".join(sorted(set(tr.kernels)))
    assert not tr.has("S1 = 3, S2 = 6"), "\n".join(sorted(set(tr.kernels)))
    print("config5 full size: recompute path taken =", lig_jet.stats["recompute_steps"] > r0)
    assert all(torch.isfinite(v).all() for v in [pred, gl] + gp + list(res.values()))
    # (a) subset vs the oracle (reference formulation: one reverse sweep per dif, fp32 like the reference)
    sel = torch.randperm(N, generator=g)[:512]
    params = [(net.fc[k].weight.detach().cpu(), net.fc[k].bias.detach().cpu()) for k in range(6)]
    orc = O.PDEOracle(*bench.C5_VARS)
    for name, eq in bench.C5_EQS.items():
        orc.add_equation(eq, name)
    act = O.activation_fn("softplus")
    orc.forward_method = lambda q: O.query_lig(lambda x: O.imnet_forward(params, x, act), lat0, q, 0., 1.)
    y_ref, r_ref = orc(pts[:, sel].clone())
    assert (pred[:, sel].cpu() - y_ref.detach()).abs().max().item() < 2e-5 * y_ref.abs().max().item()
    for k, v in r_ref.items():
        # second derivatives carry 1/cubesize^2 = 127^2 (the fp32 reference itself: max-rel ~2e-4 per point, SURVEY a-Q8)
        err = (res[k][:, sel].cpu() - v.detach()).abs() / v.detach().abs().max()
        assert err.median().item() < 1e-5 and err.max().item() < 1e-3, (k, err.max().item())
    # (b) chunk invariance of the forward (all eight streams)
    pairs = ((0, 0), (0, 1), (1, 1), (2, 2))
    with torch.no_grad():
        a, _ = lig_jet.lig_jets(net, latd, ptsd, 0., 1., True, pairs, chunk_points=1 << 20)
        b, _ = lig_jet.lig_jets(net, latd, ptsd, 0., 1., True, pairs, chunk_points=1 << 18)
    assert a.shape[0] == 8 and torch.equal(a, b)
    del a, b
    # (c) additivity over the two halves of the points
    _, _, gl1, gp1 = run(slice(0, N // 2))
    _, _, gl2, gp2 = run(slice(N // 2, N))
    assert (gl - (gl1 + gl2)).abs().max().item() < 1e-4 * gl.abs().max().item()
    for a, b, c in zip(gp, gp1, gp2):
        assert (a - (b + c)).abs().max().item() < 2e-4 * a.abs().max().item()
