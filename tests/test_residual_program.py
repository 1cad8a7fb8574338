"""The residual evaluator (stpde_residual_fwd/_bwd): host-side SSA compiler on CPU, device kernels on GPU."""
import numpy as np
import pytest
import sympy
import torch

from space_time_pde_amd import pde as P, physics


def _interp(prog, jets, x):
    """numpy interpreter of the stpde_res_ins program (the semantics documented in include/stpde_hip.h)."""
    ops = {v: k for k, v in P._RES_OPS.items()}
    v, out = [], {}
    for op, a, b, c in prog.tolist():
        name = ops[op]
        if name == "JET":
            r = jets[a, b]
        elif name == "X":
            r = x[:, a]
        elif name == "CONST":
            r = np.full(jets.shape[2], c)
        elif name in ("ADD", "SUB", "MUL", "DIV"):
            r = {"ADD": np.add, "SUB": np.subtract, "MUL": np.multiply, "DIV": np.divide}[name](v[a], v[b])
        elif name == "NEG":
            r = -v[a]
        elif name == "POWI":
            r = v[a] ** b
        elif name == "OUT":
            r = v[a]
            out[b] = r
        else:
            r = {"SIN": np.sin, "COS": np.cos, "EXP": np.exp, "LOG": np.log, "SQRT": np.sqrt, "TANH": np.tanh,
                 "ABS": np.abs}[name](v[a])
        v.append(r)
    return [out[k] for k in sorted(out)]


def _layer_c5():
    layer = P.PDELayer("x, y, t", "c, u, v, w, p")
    layer.add_equation("dif(c,t)+u*dif(c,x)+v*dif(c,y)-0.01*(dif(dif(c,x),x)+dif(dif(c,y),y))", "adv_diff")
    layer.add_equation("dif(u*c,x)+dif(v*c,y)", "prod_rule")
    layer.add_equation("dif(dif(c,x),y)-w*p", "mixed")
    layer.add_equation("x*dif(p,x)+t*dif(dif(p,t),t)+sin(u)/(1+p**2)-sqrt(1+c**2)", "explicit_x")
    return layer


def _streams(layer):
    req_pairs = sorted({mi for prog in layer.eqns_jet.values() for _, mi in prog.atoms if len(mi) == 2})
    stream_of = {(): 0, (0,): 1, (1,): 2, (2,): 3}
    for k, pr in enumerate(req_pairs):
        stream_of[pr] = 4 + k
    return stream_of


@pytest.mark.parametrize("which", ["rb2", "c5"])
def test_ssa_program_matches_lambdify(which):
    if which == "rb2":
        layer = physics.get_rb2_pde_layer(mean=(0.01, 0., 0.02, -0.01), std=(0.05, 0.3, 0.15, 0.12), t_crop=2.,
                                          z_crop=1., x_crop=1., use_continuity=True)
    else:
        layer = _layer_c5()
    progs = layer.eqns_jet
    assert all(p is not None for p in progs.values())
    stream_of = _streams(layer)
    slots = {sym: (stream_of[k[1]], k[0]) for p in progs.values() for k, sym in p.syms.items()}
    comp = P._compile_residual_program([p.expr for p in progs.values()], list(layer.in_vars), slots)
    assert comp is not None
    prog, uses_x = comp
    assert uses_x == (which == "c5") and len(prog) <= P._RES_MAX_INS
    rng = np.random.default_rng(0)
    n = 50
    jets = rng.standard_normal((len(stream_of), layer.n_out, n))
    x = rng.random((n, 3))
    got = _interp(prog, jets, x)
    for k, p in enumerate(progs.values()):
        cols = [torch.from_numpy(x[:, i]) for i in range(3)]
        atoms = [torch.from_numpy(jets[stream_of[a[1]], a[0]]) for a in p.atoms]
        want = p.fn(*(cols + atoms))
        want = want.numpy() if torch.is_tensor(want) else np.full(n, float(want))
        np.testing.assert_allclose(got[k], want, rtol=2e-6, atol=2e-6)   # program constants are fp32


def test_unsupported_expression_falls_back():
    layer = P.PDELayer("x, y, t", "u")
    layer.add_equation("dif(u,x)**1.5 + u", "frac_pow")
    prog = layer.eqns_jet["frac_pow"]
    slots = {sym: (1, 0) if k[1] else (0, 0) for k, sym in prog.syms.items()}
    assert P._compile_residual_program([prog.expr], list(layer.in_vars), slots) is None


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["rb2", "c5"])
def test_residual_kernels_match_torch_path(hiplib, which):
    dev = torch.device("cuda:0")
    if which == "rb2":
        layer = physics.get_rb2_pde_layer(mean=(0.01, 0., 0.02, -0.01), std=(0.05, 0.3, 0.15, 0.12), t_crop=2.,
                                          z_crop=1., x_crop=1., use_continuity=True)
    else:
        layer = _layer_c5()
    progs = layer.eqns_jet
    stream_of = _streams(layer)
    g = torch.Generator().manual_seed(3)
    n = 1001
    jets0 = torch.randn(len(stream_of), layer.n_out, n, generator=g)
    jets0[:, :, :7] = 0.0                      # exact zeros: the integer-power rule must stay finite there
    jets = jets0.to(dev).requires_grad_(True)
    x = torch.rand(1, n, 3, generator=g).to(dev)
    shape = (1, n, 1)
    res = layer._residues_hip(x, jets, progs, stream_of, shape)
    assert res is not None and list(res) == list(progs)
    cot = {k: torch.randn(shape, generator=g).to(dev) for k in res}
    sum((res[k] * cot[k]).sum() for k in res).backward()
    got_bar = jets.grad.clone()
    jets.grad = None
    cols = [x[..., i:i + 1] for i in range(3)]
    ref = {name: p.fn(*(cols + [jets[stream_of[a[1]], a[0]].reshape(shape) for a in p.atoms])) for name, p in progs.items()}
    sum((ref[k] * cot[k]).sum() for k in ref).backward()
    for k in res:
        assert torch.allclose(res[k], ref[k], rtol=2e-5, atol=2e-5), k
    assert torch.isfinite(got_bar).all()
    assert torch.allclose(got_bar, jets.grad, rtol=2e-5, atol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["l1", "l2", "huber"])
@pytest.mark.parametrize("with_target", [True, False])
def test_loss_sum_kernels_match_torch(hiplib, kind, with_target):
    from space_time_pde_amd import train_step as T
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    a = (2.0 * torch.randn(3, 777, 4, generator=g)).to(dev).requires_grad_(True)
    b = torch.randn(3, 777, 4, generator=g).to(dev) if with_target else None
    w = torch.tensor(0.37, device=dev)
    got = T.loss_sum(a, b, kind)
    (w * got).backward()
    ga = a.grad.clone()
    a.grad = None
    want = T._LOSS_SUMS[kind](a, b if with_target else torch.zeros_like(a))
    (w * want).backward()
    assert abs(got.item() - want.item()) <= 2e-5 * abs(want.item())
    assert torch.allclose(ga, a.grad, rtol=1e-6, atol=1e-7)
