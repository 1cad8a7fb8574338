"""CPU oracle (part 2): closed-form forward-mode jets of the LIG + IM-NET composite.

TEST INFRASTRUCTURE ONLY (same rules as ``oracle/cpu_ref.py``).  This is the mathematical
specification the HIP kernels implement: instead of 25 reverse sweeps (reference pde.py:8-9 called from
the lambdified equations), propagate value + first-order + selected second-order derivative "streams"
through the network once.  It is dtype-generic (run it in float64 for a tight reference) and built from
differentiable torch ops, so ``autograd`` on its outputs gives the parameter / latent-grid gradients the
HIP backward must match.

It is itself pinned against ``oracle/cpu_ref.py`` (autograd ``dif`` sweeps) and the golden vectors in
``tests/test_oracle_golden.py``.

Notation (SURVEY.md section 8a "Derivative structure"): for corner j with bits b_d,
    r_{j,d} = (q_d - (i0_d+b_d) cs_d)/cs_d,  omega_{j,d} = |q_d - (i0_d+1-b_d) cs_d|/cs_d,  w_j = prod_d omega_{j,d}
    y = sum_j w_j f(x_j),   x_j = [r_j ; latent_j]
MLP streams are derivatives w.r.t. r (seed e_d); the corner reduction applies kappa_d = s_d/cs_d where s_d
is the derivative of the clip (1 inside, 0.5 on a tie, 0 outside; quirk a-Q2).
"""
import torch

from .cpu_ref import _bounds, corner_table


def act_derivs(name, a, beta=None):
    """sigma(a) and its first three derivatives, matching torch's formulas/conventions at kinks."""
    z = torch.zeros_like(a)
    o = torch.ones_like(a)
    if name == "tanh":
        t = torch.tanh(a)
        u = 1 - t * t
        return t, u, -2 * t * u, -2 * u * (1 - 3 * t * t)
    if name == "relu":
        m = (a > 0).to(a.dtype)
        return a * m, m, z, z
    if name == "leakyrelu":
        m = torch.where(a > 0, o, 0.01 * o)
        return a * m, m, z, z
    if name == "softplus":
        big = a > 20
        s = torch.sigmoid(a)
        f = torch.where(big, a, torch.log1p(torch.exp(torch.clamp(a, max=20.0))))
        d1 = torch.where(big, o, s)
        d2 = torch.where(big, z, s * (1 - s))
        d3 = torch.where(big, z, s * (1 - s) * (1 - 2 * s))
        return f, d1, d2, d3
    if name == "elu":
        e = torch.exp(torch.clamp(a, max=0.0))
        pos = a > 0
        return torch.where(pos, a, e - 1), torch.where(pos, o, e), torch.where(pos, z, e), torch.where(pos, z, e)
    if name == "swish":
        s = torch.sigmoid(beta * a)
        q = s * (1 - s)
        ba = beta * a
        d0 = a * s
        d1 = s + ba * q
        d2 = beta * q * (2 + ba * (1 - 2 * s))
        d3 = beta * beta * q * (3 * (1 - 2 * s) + ba * ((1 - 2 * s) ** 2 - 2 * q))
        return d0, d1, d2, d3
    raise KeyError(name)


def round_bf16(t):
    """Round-to-nearest-even to bfloat16, returned in the input dtype (emulates v_cvt_pk_bf16_f32 on fp32 data)."""
    return t.float().to(torch.bfloat16).to(t.dtype)


def mlp_jets(params, act_name, x, dim, second, beta=None, bf16_layers=(), bf16_pre_tangents=()):
    """Jets of IM-NET w.r.t. its first ``dim`` inputs.

    x [rows, dim+c].  Returns list of streams, each [rows, out]: [value, d/dr_0..d/dr_{dim-1}, d2/dr_a dr_b for
    (a,b) in second].
    bf16_layers: layer indices whose hidden-to-hidden product is taken on bf16-rounded operands (weights and input
    streams), accumulated in the working precision -- the emulation of the library's config-4 "bf16 MFMA" mode.
    bf16_pre_tangents: layer indices whose OUTPUT pre-activations are kept in the library's packed form (round 3: value
    stream fp32, derivative streams rounded to bf16 when they are stored) -- every later use of those derivative streams sees
    the rounded values.
    """
    rows = x.shape[0]
    nlayers = len(params)
    h = None
    hd = None
    hdd = None
    for l, (w, b) in enumerate(params):
        last = l == nlayers - 1
        if l == 0:
            a = x @ w.t() + b
            ad = [w[:, d].unsqueeze(0).expand(rows, -1) for d in range(dim)]
            add = [torch.zeros_like(a) for _ in second]
        else:
            kh = h.shape[1]
            wh = w[:, :kh]
            if l in bf16_layers:
                wh, h, hd, hdd = round_bf16(wh), round_bf16(h), [round_bf16(t) for t in hd], \
                    [round_bf16(t) for t in hdd]
            a = h @ wh.t() + b
            ad = [t @ wh.t() for t in hd]
            add = [t @ wh.t() for t in hdd]
            if w.shape[1] > kh:                       # skip block: re-concatenated raw input
                ws = w[:, kh:]
                a = a + x @ ws.t()
                ad = [ad[d] + ws[:, d].unsqueeze(0) for d in range(dim)]
        if last:
            return [a] + ad + add
        if l in bf16_pre_tangents:
            ad, add = [round_bf16(t) for t in ad], [round_bf16(t) for t in add]
        s0, s1, s2, _ = act_derivs(act_name, a, beta)
        h = s0
        hd = [s1 * t for t in ad]
        hdd = [s2 * ad[p] * ad[q] + s1 * add[k] for k, (p, q) in enumerate(second)]


def lig_jets(params, act_name, latent_grid, pts, xmin=0.0, xmax=1.0, second=((1, 1), (2, 2)), beta=None,
             bf16_layers=(), bf16_pre_tangents=()):
    """Jets of y = query_local_implicit_grid(...) w.r.t. the query point coordinates.

    latent_grid [b, n1..nd, c], pts [b,p,d].  Returns tensor [S, b, p, out] with S = 1 + d + len(second):
    stream 0 = y, 1..d = dy/dq_k, then d2y/dq_a dq_b for (a,b) in ``second``.
    """
    dim = latent_grid.dim() - 2
    dev, dt = latent_grid.device, latent_grid.dtype
    size = torch.tensor(latent_grid.shape[1:-1], device=dev).to(dt)
    lo, hi = _bounds(xmin, xmax, dim, dev)
    lo, hi = lo.to(dt), hi.to(dt)
    eps = 1e-6 * (hi - lo)
    up, dn = hi - eps, lo + eps
    m = torch.min(pts, up)
    q = torch.max(m, dn)
    one, half, zero = torch.ones_like(pts), 0.5 * torch.ones_like(pts), torch.zeros_like(pts)
    s = torch.where(pts < up, one, torch.where(pts == up, half, zero)) * \
        torch.where(m > dn, one, torch.where(m == dn, half, zero))
    cube = (hi - lo) / (size - 1)
    kap = s / cube
    i0 = torch.floor(q / cube).long()
    i0f = i0.to(dt)
    lo_pos, hi_pos = i0f * cube, (i0f + 1) * cube
    b, p = pts.shape[0], pts.shape[1]
    bidx = torch.arange(b, device=dev).view(b, 1).expand(b, p)
    nstream = 1 + dim + len(second)
    out = None
    for bits in corner_table(dim):
        idx = tuple(i0[..., k] + bits[k] for k in range(dim))
        lat = latent_grid[(bidx,) + idx]
        pos = torch.stack([hi_pos[..., k] if bits[k] else lo_pos[..., k] for k in range(dim)], -1)
        opp = torch.stack([lo_pos[..., k] if bits[k] else hi_pos[..., k] for k in range(dim)], -1)
        t = q - opp
        om = torch.abs(t) / cube                                  # [b,p,d]
        dom = torch.sign(t) / cube * s                            # d omega / d q (abs'(0)=0, quirk a-Q3)
        rel = (q - pos) / cube
        x = torch.cat([rel, lat], dim=-1).reshape(b * p, -1)
        f = [t_.reshape(b, p, -1) for t_ in mlp_jets(params, act_name, x, dim, list(second), beta, bf16_layers,
                                                          bf16_pre_tangents)]
        w = torch.prod(om, dim=-1, keepdim=True)

        def dw(d):
            r = dom[..., d:d + 1]
            for e in range(dim):
                if e != d:
                    r = r * om[..., e:e + 1]
            return r

        def ddw(d, e):
            if d == e:
                return torch.zeros_like(w)
            r = dom[..., d:d + 1] * dom[..., e:e + 1]
            for g in range(dim):
                if g != d and g != e:
                    r = r * om[..., g:g + 1]
            return r

        terms = [w * f[0]]
        for d in range(dim):
            terms.append(dw(d) * f[0] + w * kap[..., d:d + 1] * f[1 + d])
        for k, (d, e) in enumerate(second):
            kd, ke = kap[..., d:d + 1], kap[..., e:e + 1]
            terms.append(ddw(d, e) * f[0] + dw(d) * ke * f[1 + e] + dw(e) * kd * f[1 + d]
                         + w * kd * ke * f[1 + dim + k])
        cur = torch.stack(terms, 0)
        out = cur if out is None else out + cur
    assert out.shape[0] == nstream
    return out
