"""GPU parity: HIP jet path (through the C-ABI library) vs the CPU oracle on identical seeded inputs.

Tolerances (fp32 path vs fp64 oracle): values/jets 2e-5 of the tensor's max magnitude; parameter / latent
gradients 2e-4 relative to the gradient's max magnitude (they are sums of ~1e3..1e5 fp32 terms accumulated in a
different order, with fp32 atomics); PDE-residual loss 1e-5 relative (the north-star bound).
"""
import numpy as np
import pytest
import torch

from oracle import cpu_ref as O
from oracle import jet_ref as J

pytestmark = pytest.mark.gpu

ACTS = ["softplus", "leakyrelu", "tanh", "relu", "elu", "swish"]


def _net(act, nf=16, cout=4, seed=0, beta=1.3):
    from space_time_pde_amd import implicit_net, nonlinearities
    torch.manual_seed(seed)
    net = implicit_net.ImNet(dim=3, in_features=32, out_features=cout, nf=nf,
                             activation=nonlinearities.NONLINEARITIES[act])
    if act == "swish":
        with torch.no_grad():
            net.activ.beta.fill_(beta)
        net.activ.beta.requires_grad_(False)
    return net


def _params64(net):
    return [(net.fc[k].weight.detach().double().cpu(), net.fc[k].bias.detach().double().cpu()) for k in range(6)]


def _relerr(a, b):
    return (a.double().cpu() - b.double()).abs().max().item() / max(b.abs().max().item(), 1e-30)


def _normerr(a, b):
    return (a.double().cpu() - b.double()).norm().item() / max(b.double().norm().item(), 1e-30)


@pytest.mark.parametrize("act", ACTS)
@pytest.mark.parametrize("pairs", [(), ((1, 1), (2, 2)), ((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2))])
@pytest.mark.parametrize("prec", ["fp32", "fp32x3"])
def test_jets_match_oracle(hiplib, act, pairs, prec, monkeypatch):
    # the same test, same tolerances, with the wide layers' products as exact-split bf16 MFMAs ("fp32x3")
    from space_time_pde_amd import lig_jet as _lj
    monkeypatch.setattr(_lj, "mlp_precision", prec)
    from space_time_pde_amd import lig_jet
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    lat = 0.5 * torch.randn(2, 4, 6, 5, 32, generator=g)
    pts = torch.rand(2, 77, 3, generator=g)              # odd per-batch count, 154 points total
    net = _net(act).to(dev)
    jets, pp = lig_jet.lig_jets(net, lat.to(dev), pts.to(dev), 0., 1., True, pairs, chunk_points=64)
    beta = torch.tensor(1.3, dtype=torch.float64)
    ref = J.lig_jets(_params64(net), act, lat.double(), pts.double(), 0., 1., second=tuple(pp), beta=beta)
    ref = ref.permute(0, 3, 1, 2).reshape(ref.shape[0], 4, -1)
    assert jets.shape == ref.shape
    for s in range(ref.shape[0]):
        assert _relerr(jets[s], ref[s]) < 2e-5, "stream %d" % s


def test_value_only_and_nonunit_box(hiplib):
    from space_time_pde_amd import local_implicit_grid as lig
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(4)
    lat = torch.randn(1, 5, 4, 7, 32, generator=g)
    xmax = (2.0, 1.0, 4.0)
    pts = torch.rand(1, 301, 3, generator=g) * torch.tensor(xmax)
    net = _net("leakyrelu").to(dev)
    n0 = lig.stats["hip_value_calls"]
    with torch.no_grad():
        y = lig.query_local_implicit_grid(net, lat.to(dev), pts.to(dev), (0., 0., 0.), xmax)
    assert lig.stats["hip_value_calls"] == n0 + 1
    p32 = [(w.float(), b.float()) for w, b in _params64(net)]
    ref = O.query_lig(lambda f: O.imnet_forward(p32, f, O.activation_fn("leakyrelu")), lat, pts, (0., 0., 0.), xmax)
    assert _relerr(y, ref) < 2e-5


@pytest.mark.parametrize("act", ["softplus", "leakyrelu", "tanh", "elu", "swish"])
@pytest.mark.parametrize("prec", ["fp32", "fp32x3"])
def test_backward_matches_oracle_autograd(hiplib, act, prec, monkeypatch):
    # the same test, same tolerances, with the wide layers' products as exact-split bf16 MFMAs ("fp32x3")
    from space_time_pde_amd import lig_jet as _lj
    monkeypatch.setattr(_lj, "mlp_precision", prec)
    from space_time_pde_amd import lig_jet
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    lat = 0.5 * torch.randn(2, 4, 5, 6, 32, generator=g)
    pts = 0.02 + 0.96 * torch.rand(2, 200, 3, generator=g)
    pairs = ((1, 1), (2, 2))
    net = _net(act).to(dev)
    beta64 = torch.tensor(1.3, dtype=torch.float64, requires_grad=True)
    if act == "swish":
        net.activ.beta.requires_grad_(True)     # the learnable beta of the reference's Swish (nonlinearities.py:9)
    latd = lat.to(dev).requires_grad_(True)
    jets, pp = lig_jet.lig_jets(net, latd, pts.to(dev), 0., 1., True, pairs, chunk_points=128)
    cot = torch.randn(jets.shape, generator=g)
    (jets * cot.to(dev)).sum().backward()
    p64 = [(w.requires_grad_(True), b.requires_grad_(True)) for w, b in _params64(net)]
    lat64 = lat.double().requires_grad_(True)
    ref = J.lig_jets(p64, act, lat64, pts.double(), 0., 1., second=tuple(pp), beta=beta64)
    ref = ref.permute(0, 3, 1, 2).reshape(ref.shape[0], 4, -1)
    (ref * cot.double()).sum().backward()
    if act == "swish":
        assert abs(net.activ.beta.grad.item() - beta64.grad.item()) < 2e-4 * abs(beta64.grad.item())
    assert _relerr(latd.grad, lat64.grad) < 2e-4
    for k in range(6):
        assert _relerr(net.fc[k].weight.grad, p64[k][0].grad) < 2e-4, "dW%d" % k
        assert _relerr(net.fc[k].bias.grad, p64[k][1].grad) < 2e-4, "db%d" % k


@pytest.mark.parametrize("act", ACTS)
@pytest.mark.parametrize("prec", ["fp32", "fp32x3"])
def test_golden_composite_g5(hiplib, golden_dir, act, prec, monkeypatch):
    """LIG + RB2 residuals + L1 losses + backward vs vectors produced by the real reference (G5)."""
    # the same test, same tolerances, with the wide layers' products as exact-split bf16 MFMAs ("fp32x3")
    from space_time_pde_amd import lig_jet as _lj
    monkeypatch.setattr(_lj, "mlp_precision", prec)
    import os
    from space_time_pde_amd import local_implicit_grid as lig, physics
    d = np.load(os.path.join(golden_dir, "g5_composite.npz"))
    dev = torch.device("cuda:0")
    net = _net(act).to(dev)
    if act == "swish":
        net.activ.beta.requires_grad_(True)
    with torch.no_grad():
        for k in range(6):
            net.fc[k].weight.copy_(torch.from_numpy(d["w%d" % k]))
            net.fc[k].bias.copy_(torch.from_numpy(d["b%d" % k]))
    lat = torch.from_numpy(d["latent"]).to(dev).requires_grad_(True)
    pts = torch.from_numpy(d["pts"]).to(dev)
    tgt = torch.from_numpy(d["targets"]).to(dev)
    layer = physics.get_rb2_pde_layer(mean=tuple(d["mean"]), std=tuple(d["std"]), t_crop=2., z_crop=1., x_crop=1.,
                                      use_continuity=True)
    layer.update_forward_method(lambda p: lig.query_local_implicit_grid(net, lat, p, 0., 1.))
    n0 = lig.stats["hip_jet_calls"]
    pred, res = layer(pts, return_residue=True)
    assert lig.stats["hip_jet_calls"] == n0 + 1
    reg = torch.nn.functional.l1_loss(pred, tgt)
    st = torch.stack(list(res.values()), 0)
    pl = torch.nn.functional.l1_loss(st, torch.zeros_like(st))
    (1.0 * reg + 0.0125 * pl).backward()
    assert _relerr(pred, torch.from_numpy(d[act + "_pred"])) < 2e-5
    # piecewise-linear activations: a kink flip moves single residual values; check loss + bulk of the points
    for k, v in res.items():
        ref = torch.from_numpy(d["%s_res_%s" % (act, k)]).double()
        err = (v.detach().double().cpu() - ref).abs() / ref.abs().max()
        assert err.median().item() < 1e-5 and (err < 1e-3).double().mean().item() > 0.98, k
    assert abs(pl.item() - float(d[act + "_pde_loss"])) / float(d[act + "_pde_loss"]) < 1e-5
    assert abs(reg.item() - float(d[act + "_reg_loss"])) / float(d[act + "_reg_loss"]) < 1e-5
    # piecewise-linear activations: single kink / sign flips (fp32 rounding of a pre-activation near 0) move
    # individual gradient entries; the reference differs from its own fp64 run by 2e-3 max-rel there (SURVEY a-Q8),
    # so those are judged in the Frobenius norm with a loose max bound; smooth activations entry-wise.
    pl_act = act in ("leakyrelu", "relu")
    err = _normerr if pl_act else _relerr
    tol = 5e-3 if pl_act else 5e-4

    def close(a, b):
        assert err(a, b) < tol
        assert _relerr(a, b) < (5e-2 if pl_act else tol)

    close(lat.grad, torch.from_numpy(d[act + "_dlatent"]))
    if act == "swish":   # gradient of the learnable beta, from the reference's own backward
        close(net.activ.beta.grad.reshape(1), torch.from_numpy(d["swish_dbeta"]).reshape(1))
    for k in range(3, 6):
        close(net.fc[k].weight.grad, torch.from_numpy(d["%s_dw%d" % (act, k)]))
        close(net.fc[k].bias.grad, torch.from_numpy(d["%s_db%d" % (act, k)]))
    if act in ("softplus", "leakyrelu"):
        for k in range(3):
            close(net.fc[k].weight.grad, torch.from_numpy(d["%s_dw%d" % (act, k)]))


def test_smoke_entry(hiplib):
    import __graft_entry__ as ge
    ge.smoke()


@pytest.mark.parametrize("act", ["softplus", "tanh", "leakyrelu", "swish"])
def test_combined_second_order_stream(hiplib, act):
    """The single combined stream  L y = sum_k alpha_k d2y/dq_a dq_b  (S = 5) equals the same combination of the
    per-pair oracle jets, forward and backward (incl. a clipped / out-of-box point where kappa is 0 or 0.5)."""
    from space_time_pde_amd import lig_jet
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(8)
    lat = 0.5 * torch.randn(2, 4, 5, 6, 32, generator=g)
    pts = torch.rand(2, 101, 3, generator=g)
    pts[0, 0] = torch.tensor([1.5, 0.3, -0.2])          # outside the box: clip derivative 0 in two dims
    combo = {(0, 0): 0.3, (1, 1): 1.0, (1, 2): -0.7, (2, 2): 0.25}
    pairs = tuple(sorted(combo))
    net = _net(act).to(dev)
    latd = lat.to(dev).requires_grad_(True)
    jets, pp = lig_jet.lig_jets(net, latd, pts.to(dev), 0., 1., True, (), chunk_points=64, combo=combo)
    assert pp == ["combo"] and jets.shape[0] == 5
    cot = torch.randn(jets.shape, generator=g)
    (jets * cot.to(dev)).sum().backward()
    beta = torch.tensor(1.3, dtype=torch.float64)
    p64 = [(w.requires_grad_(True), b.requires_grad_(True)) for w, b in _params64(net)]
    lat64 = lat.double().requires_grad_(True)
    full = J.lig_jets(p64, act, lat64, pts.double(), 0., 1., second=pairs, beta=beta)
    full = full.permute(0, 3, 1, 2).reshape(full.shape[0], 4, -1)
    L = sum(combo[p] * full[4 + k] for k, p in enumerate(pairs))
    ref = torch.cat([full[:4], L[None]], 0)
    for s in range(5):
        assert _relerr(jets[s], ref[s]) < 2e-5, "stream %d" % s
    (ref * cot.double()).sum().backward()
    err = _normerr if act == "leakyrelu" else _relerr
    assert err(latd.grad, lat64.grad) < 5e-4
    for k in range(6):
        assert err(net.fc[k].weight.grad, p64[k][0].grad) < 5e-4, "dW%d" % k
        assert err(net.fc[k].bias.grad, p64[k][1].grad) < 5e-4, "db%d" % k


def test_edge_cases_empty_single_and_boundary_points(hiplib):
    """Empty query set, a single point (odd count -> padded tile), points exactly on the box faces / grid nodes."""
    from space_time_pde_amd import local_implicit_grid as lig, physics
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(2)
    lat = torch.randn(1, 3, 4, 5, 32, generator=g)
    net = _net("softplus").to(dev)
    with torch.no_grad():
        y = lig.query_local_implicit_grid(net, lat.to(dev), torch.zeros(1, 0, 3, device=dev), 0., 1.)
    assert y.shape == (1, 0, 4)
    pts = torch.tensor([[[0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [0.5, 1.0 / 3.0, 0.25], [0.3, 0.9, 0.1], [2.0, -1.0, 0.5]]])
    p32 = [(w.float(), b.float()) for w, b in _params64(net)]
    for n in (1, 5):
        with torch.no_grad():
            y = lig.query_local_implicit_grid(net, lat.to(dev), pts[:, :n].to(dev), 0., 1.)
        ref = O.query_lig(lambda f: O.imnet_forward(p32, f, O.activation_fn("softplus")), lat, pts[:, :n], 0., 1.)
        assert _relerr(y, ref) < 2e-5
    # residuals at those points (clip ties / abs'(0) quirks a-Q2, a-Q3) vs the reference-semantics oracle
    kw = dict(t_crop=2., z_crop=1., x_crop=1., use_continuity=True)
    layer = physics.get_rb2_pde_layer(**kw)
    latd = lat.to(dev)
    layer.update_forward_method(lambda q: lig.query_local_implicit_grid(net, latd, q, 0., 1.))
    pred, res = layer(pts.to(dev))
    out = O.lig_pde_step(p32, "softplus", lat, pts, torch.zeros(1, 5, 4), O.rb2_oracle(**kw), backward=False)
    for k, v in out["residues"].items():
        assert (res[k].cpu() - v).abs().max().item() < 1e-4 * max(v.abs().max().item(), 1e-3), k


def _bf16_layers(nf):
    """Layers whose hidden-to-hidden GEMM runs on bf16 MFMA operands (include/stpde_hip.h, stpde_layer_desc.mfma_bf16):
    those served by the workgroup-cooperative kernels, KT % 4 == 0, KT >= 8, MT % 8 == 0."""
    widths = [16 * nf, 8 * nf, 4 * nf, 2 * nf, nf]
    return tuple(l for l in range(1, 5) if (widths[l - 1] // 16) % 4 == 0 and widths[l - 1] // 16 >= 8
                 and (widths[l] // 16) % 8 == 0)


@pytest.mark.parametrize("act", ["softplus", "leakyrelu", "tanh"])
def test_bf16_mfma_mode_config4(hiplib, act):
    """BASELINE config 4: bf16 MFMA operands / fp32 accumulation in every hidden-to-hidden product (one bf16 term in the
    wide layers, two in the narrow ones), packed layer buffers.  Tolerances (explicit, looser than the fp32 path): vs an
    oracle that rounds the same operands / stored streams to bf16: 5e-4 (Frobenius); vs exact fp64: 3e-2."""
    assert _bf16_layers(32) == (1, 2)
    from space_time_pde_amd import lig_jet
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    lat = 0.5 * torch.randn(2, 4, 5, 6, 32, generator=g)
    pts = 0.02 + 0.96 * torch.rand(2, 150, 3, generator=g)
    pairs = ((1, 1), (2, 2))
    net = _net(act, nf=32).to(dev)
    latd = lat.to(dev).requires_grad_(True)
    jets, pp = lig_jet.lig_jets(net, latd, pts.to(dev), 0., 1., True, pairs, chunk_points=128, precision="bf16")
    with torch.no_grad():
        jets32, _ = lig_jet.lig_jets(net, lat.to(dev), pts.to(dev), 0., 1., True, pairs, chunk_points=128)
    # what the emulation covers: bf16 operands in the two wide layers, and their stored output rows (packed layer buffers:
    # derivative streams kept as bf16), which the next layer's forward reads.  The three narrow layers run on TWO-term bf16
    # operands (2^-16 relative: exact at this test's resolution); the packed stores of their rows are read by the backward
    # pass only.  (One-term bf16 operands in the narrow layers would put the second-derivative streams 2e-2 ... 5e-2 from
    # exact -- oracle, tools/micro/dbg_bf16_emul.py -- which is why they are not used.)
    emu = J.lig_jets(_params64(net), act, lat.double(), pts.double(), 0., 1., second=tuple(pp),
                     bf16_layers=_bf16_layers(32), bf16_pre_tangents=(1, 2))
    emu = emu.permute(0, 3, 1, 2).reshape(emu.shape[0], 4, -1)
    p64 = [(w.requires_grad_(True), b.requires_grad_(True)) for w, b in _params64(net)]
    lat64 = lat.double().requires_grad_(True)
    ref = J.lig_jets(p64, act, lat64, pts.double(), 0., 1., second=tuple(pp))
    ref = ref.permute(0, 3, 1, 2).reshape(ref.shape[0], 4, -1)
    pl_act = act == "leakyrelu"
    for s in range(ref.shape[0]):
        assert _normerr(jets[s], emu[s]) < (5e-3 if pl_act else 5e-4), "stream %d vs bf16-emulating oracle" % s
        assert _normerr(jets[s], ref[s].detach()) < 3e-2, "stream %d vs exact" % s
        assert _normerr(jets[s], jets32[s].double().cpu()) > 1e-5, "bf16 mode did not engage (stream %d)" % s
    cot = torch.randn(jets.shape, generator=g)
    (jets * cot.to(dev)).sum().backward()
    (ref * cot.double()).sum().backward()
    # piecewise-linear activations: operand rounding flips kinks (sigma' jumps), so the early layers' gradients of
    # this random-cotangent functional move more than for smooth activations
    gtol = 1e-1 if pl_act else 3e-2
    assert _normerr(latd.grad, lat64.grad) < gtol
    for k in range(6):
        assert _normerr(net.fc[k].weight.grad, p64[k][0].grad) < gtol, "dW%d" % k
        assert _normerr(net.fc[k].bias.grad, p64[k][1].grad) < gtol, "db%d" % k


@pytest.mark.parametrize("nf,cout", [(48, 5), (64, 3)])
def test_other_network_widths(hiplib, nf, cout):
    """Widths that take other kernel-dispatch branches than the nf = 16 / 32 cases (odd tile counts, 3 passes, ...)."""
    from space_time_pde_amd import lig_jet
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(21)
    lat = 0.5 * torch.randn(1, 4, 5, 6, 32, generator=g)
    pts = 0.02 + 0.96 * torch.rand(1, 90, 3, generator=g)
    pairs = ((1, 1), (2, 2))
    net = _net("softplus", nf=nf, cout=cout).to(dev)
    latd = lat.to(dev).requires_grad_(True)
    jets, pp = lig_jet.lig_jets(net, latd, pts.to(dev), 0., 1., True, pairs)
    cot = torch.randn(jets.shape, generator=g)
    (jets * cot.to(dev)).sum().backward()
    p64 = [(w.requires_grad_(True), b.requires_grad_(True)) for w, b in _params64(net)]
    lat64 = lat.double().requires_grad_(True)
    ref = J.lig_jets(p64, "softplus", lat64, pts.double(), 0., 1., second=tuple(pp))
    ref = ref.permute(0, 3, 1, 2).reshape(ref.shape[0], cout, -1)
    (ref * cot.double()).sum().backward()
    for s in range(ref.shape[0]):
        assert _relerr(jets[s], ref[s].detach()) < 2e-5, "stream %d" % s
    assert _relerr(latd.grad, lat64.grad) < 2e-4
    for k in range(6):
        assert _relerr(net.fc[k].weight.grad, p64[k][0].grad) < 2e-4, "dW%d" % k
        assert _relerr(net.fc[k].bias.grad, p64[k][1].grad) < 2e-4, "db%d" % k


@pytest.mark.parametrize("nf", [16, 32])
def test_value_only_query_backward(hiplib, nf):
    """Plain query (no PDE layer, e.g. a regression-only loss): the S = 1 kernels, forward and backward."""
    from space_time_pde_amd import local_implicit_grid as lig
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(8)
    lat = 0.5 * torch.randn(2, 4, 5, 6, 32, generator=g)
    pts = 0.02 + 0.96 * torch.rand(2, 333, 3, generator=g)
    net = _net("softplus", nf=nf).to(dev)
    latd = lat.to(dev).requires_grad_(True)
    n0 = lig.stats["hip_value_calls"]
    y = lig.query_local_implicit_grid(net, latd, pts.to(dev), 0., 1.)
    assert lig.stats["hip_value_calls"] == n0 + 1
    cot = torch.randn(y.shape, generator=g)
    (y * cot.to(dev)).sum().backward()
    p64 = [(w.requires_grad_(True), b.requires_grad_(True)) for w, b in _params64(net)]
    lat64 = lat.double().requires_grad_(True)
    ref = O.query_lig(lambda f: O.imnet_forward(p64, f, O.activation_fn("softplus")), lat64, pts.double(), 0., 1.)
    (ref * cot.double()).sum().backward()
    assert _relerr(y, ref.detach()) < 2e-5
    assert _relerr(latd.grad, lat64.grad) < 2e-4
    for k in range(6):
        assert _relerr(net.fc[k].weight.grad, p64[k][0].grad) < 2e-4, "dW%d" % k
        assert _relerr(net.fc[k].bias.grad, p64[k][1].grad) < 2e-4, "db%d" % k


@pytest.mark.parametrize("nf", [16, 32])
def test_backward_all_six_second_order_streams(hiplib, nf):
    """S = 10 (value, gradient, full Hessian): the widest stream set, forward and backward."""
    from space_time_pde_amd import lig_jet
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(13)
    lat = 0.5 * torch.randn(1, 4, 5, 6, 32, generator=g)
    pts = 0.02 + 0.96 * torch.rand(1, 150, 3, generator=g)
    pairs = ((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2))
    net = _net("tanh", nf=nf).to(dev)
    latd = lat.to(dev).requires_grad_(True)
    jets, pp = lig_jet.lig_jets(net, latd, pts.to(dev), 0., 1., True, pairs)
    assert jets.shape[0] == 10
    cot = torch.randn(jets.shape, generator=g)
    (jets * cot.to(dev)).sum().backward()
    p64 = [(w.requires_grad_(True), b.requires_grad_(True)) for w, b in _params64(net)]
    lat64 = lat.double().requires_grad_(True)
    ref = J.lig_jets(p64, "tanh", lat64, pts.double(), 0., 1., second=tuple(pp))
    ref = ref.permute(0, 3, 1, 2).reshape(ref.shape[0], 4, -1)
    (ref * cot.double()).sum().backward()
    for s in range(10):
        assert _relerr(jets[s], ref[s].detach()) < 2e-5, "stream %d" % s
    assert _relerr(latd.grad, lat64.grad) < 2e-4
    for k in range(6):
        assert _relerr(net.fc[k].weight.grad, p64[k][0].grad) < 2e-4, "dW%d" % k
        assert _relerr(net.fc[k].bias.grad, p64[k][1].grad) < 2e-4, "db%d" % k


def test_retain_graph_second_backward_rebuilds_the_stash(hiplib):
    """``loss.backward(retain_graph=True)`` followed by another backward (the reference's autograd graph allows it): the
    dgrad kernels consumed the stash in place, so the forward kernels are run again -- the second gradient equals the
    first bit for bit (same kernels, same inputs; d latent is deterministic), parameters accumulate 2x."""
    from space_time_pde_amd import lig_jet
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(9)
    lat = 0.5 * torch.randn(1, 4, 5, 6, 32, generator=g)
    pts = 0.02 + 0.96 * torch.rand(1, 130, 3, generator=g)
    net = _net("softplus").to(dev)
    latd = lat.to(dev).requires_grad_(True)
    jets, _ = lig_jet.lig_jets(net, latd, pts.to(dev), 0., 1., True, ((1, 1), (2, 2)), chunk_points=64)
    cot = torch.randn(jets.shape, generator=g).to(dev)
    loss = (jets * cot).sum()
    loss.backward(retain_graph=True)
    g1 = latd.grad.clone()
    w1 = net.fc[1].weight.grad.clone()
    latd.grad = None
    loss.backward()
    assert torch.equal(latd.grad, g1)
    assert torch.allclose(net.fc[1].weight.grad, 2 * w1, rtol=1e-5, atol=2e-6 * float(w1.abs().max()))   # fp32-atomic summation order


@pytest.mark.parametrize("cin,nf", [(8, 16), (16, 32), (24, 16)])
def test_fewer_latent_channels(hiplib, cin, nf):
    """Latent widths below the reference's 32: the augmented input then leaves the sparse third tile (and for 8 channels
    the second tile) empty and the latent-gradient GEMM runs with one 16-channel tile (k_xbar<1>) or a partly filled
    second one; forward jets and all gradients vs the fp64 oracle."""
    from space_time_pde_amd import _lib, implicit_net, lig_jet, nonlinearities
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(31 + cin)
    torch.manual_seed(5)
    net = implicit_net.ImNet(dim=3, in_features=cin, out_features=4, nf=nf,
                             activation=nonlinearities.NONLINEARITIES["softplus"]).to(dev)
    lat = 0.5 * torch.randn(2, 4, 5, 6, cin, generator=g)
    pts = 0.02 + 0.96 * torch.rand(2, 75, 3, generator=g)
    pairs = ((1, 1), (2, 2))
    latd = lat.to(dev).requires_grad_(True)
    with _lib.dispatch_trace() as tr:
        jets, pp = lig_jet.lig_jets(net, latd, pts.to(dev), 0., 1., True, pairs, chunk_points=64)
        cot = torch.randn(jets.shape, generator=g)
        (jets * cot.to(dev)).sum().backward()
        torch.cuda.synchronize()
    assert tr.has("k_xbar<1>" if cin <= 16 else "k_xbar<2>"), "\n".join(tr.kernels)
    p64 = [(w.requires_grad_(True), b.requires_grad_(True)) for w, b in _params64(net)]
    lat64 = lat.double().requires_grad_(True)
    ref = J.lig_jets(p64, "softplus", lat64, pts.double(), 0., 1., second=tuple(pp))
    ref = ref.permute(0, 3, 1, 2).reshape(ref.shape[0], 4, -1)
    (ref * cot.double()).sum().backward()
    for s in range(ref.shape[0]):
        assert _relerr(jets[s], ref[s].detach()) < 2e-5, "stream %d" % s
    assert _relerr(latd.grad, lat64.grad) < 2e-4
    for k in range(6):
        assert _relerr(net.fc[k].weight.grad, p64[k][0].grad) < 2e-4, "dW%d" % k
        assert _relerr(net.fc[k].bias.grad, p64[k][1].grad) < 2e-4, "db%d" % k


def test_bf16_mode_packed_buffers_on_the_side_paths(hiplib, monkeypatch):
    """bf16 mode (packed stash / adjoint buffers) through the paths next to the plain step: chunk-wise recomputation of the
    stash in the backward, a second backward after retain_graph, a forward without gradients, the atomic d-latent scatter,
    and the value-only query (which keeps the fp32 fc3 -> fc5 kernels: agreement to the two-term products' 2^-16)."""
    from space_time_pde_amd import lig_jet
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    lat = 0.5 * torch.randn(2, 4, 5, 6, 32, generator=g)
    pts = torch.rand(2, 777, 3, generator=g)
    net = _net("softplus", nf=32, seed=1).to(dev)
    combo = {(1, 1): 1.0, (2, 2): 0.25}

    def run():
        for p in net.parameters():
            p.grad = None
        latd = lat.to(dev).requires_grad_(True)
        jets, _ = lig_jet.lig_jets(net, latd, pts.to(dev), 0., 1., True, (), chunk_points=512, combo=combo, precision="bf16")
        cot = torch.randn(jets.shape, generator=torch.Generator().manual_seed(5)).to(dev)
        return jets, latd, cot

    jets, latd, cot = run()
    (jets * cot).sum().backward(retain_graph=True)
    j0, g0, w0 = jets.detach().clone(), latd.grad.clone(), [p.grad.clone() for p in net.parameters()]
    latd.grad = None
    (jets * cot).sum().backward()                      # second backward: the stash is rebuilt by the forward kernels
    assert torch.equal(latd.grad, g0)
    n0 = lig_jet.stats["recompute_steps"]
    monkeypatch.setattr(lig_jet, "force_recompute", True)
    jets, latd, cot = run()
    (jets * cot).sum().backward()
    monkeypatch.setattr(lig_jet, "force_recompute", False)
    assert lig_jet.stats["recompute_steps"] == n0 + 1
    assert torch.equal(jets.detach(), j0) and torch.equal(latd.grad, g0)
    for a, b in zip(w0, [p.grad for p in net.parameters()]):
        assert (a - b).abs().max().item() <= 5e-6 * a.abs().max().item()       # fp32-atomic summation order
    with torch.no_grad():
        j2, _ = lig_jet.lig_jets(net, lat.to(dev), pts.to(dev), 0., 1., True, (), chunk_points=512, combo=combo, precision="bf16")
        j3, _ = lig_jet.lig_jets(net, lat.to(dev), pts.to(dev), 0., 1., False, (), chunk_points=512, precision="bf16")
    assert torch.equal(j2, j0)
    assert _relerr(j3[0], j0[0].double().cpu()) < 5e-5
    monkeypatch.setattr(lig_jet, "deterministic_dlatent", False)
    jets, latd, cot = run()
    (jets * cot).sum().backward()
    assert _relerr(latd.grad, g0.double().cpu()) < 5e-6
