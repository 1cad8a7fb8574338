"""stpde_conv3d_fused / stpde_conv3d_wgrad_onload (csrc/conv3d_fused.hip, round 4) feature by feature through the C ABI
against torch fp64: every fold-in of the BatchNorm work of a ResBlock3D (reference src/unet3d.py:39-56) on its own --
statistics epilogue (double-format sums), mask + BatchNorm-backward sums epilogue, BatchNorm + ReLU on load (incl. the
statistics it writes and the running-statistics update), two outputs, two inputs, on-load weight gradient -- for the
one-voxel-tile-per-wave and the four-voxel-tiles-per-wave instantiations and both kernel sizes.
"""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from space_time_pde_amd import _lib, unet3d

pytestmark = pytest.mark.gpu
R = _lib.BN_REP
SHAPES = [(1, 4, 8, 24), (2, 8, 16, 32), (1, 16, 128, 128)]       # 768 (ragged tail tile), 8192, 262,144 voxels (VT = 4)


def _args(shape, ci, co, k):
    a = _lib.Conv3dFusedArgs()
    a.d.B, a.d.T, a.d.Z, a.d.X = shape
    a.d.Ci, a.d.Co, a.d.ksize = ci, co, k
    return a


def _packs(w, dev):
    co, ci, k = w.shape[0], w.shape[1], w.shape[2]
    fidx, bidx, _, _ = unet3d._pack_indices(co, ci, k, dev)
    wflat = torch.cat([w.reshape(-1), w.new_zeros(1)])
    return wflat[fidx].contiguous(), wflat[bidx].contiguous()


def _conv64(x, w, b):
    y = F.conv3d(x.double().permute(0, 4, 1, 2, 3), w.double(), None if b is None else b.double(), padding=(w.shape[2] - 1) // 2)
    return y.permute(0, 2, 3, 4, 1)


def _rel(a, b):
    b = b.double()
    return (a.double() - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def _sums(t):
    """per-channel sum x, sum x^2 of the double-format buffer [REP][2][C]"""
    c = t.numel() // (2 * R)
    v = t.view(R, 2, c).sum(0)
    return v[0], v[1]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("k,ci,co", [(1, 16, 32), (3, 16, 16), (3, 32, 32)])
def test_statistics_epilogue(hiplib, shape, k, ci, co):
    dev = torch.device("cuda:0")
    torch.manual_seed(k + ci + co + shape[3])
    x = torch.randn(*shape, ci, device=dev) + 0.7
    w = 0.2 * torch.randn(co, ci, k, k, k, device=dev)
    b = torch.randn(co, device=dev)
    fp, _ = _packs(w, dev)
    y = torch.empty(*shape, co, device=dev)
    sums = torch.zeros(R * 2 * co, device=dev, dtype=torch.float64)
    a = _args(shape, ci, co, k)
    a.x, a.w_pack, a.bias, a.y, a.out_sums = _lib.ptr(x), _lib.ptr(fp), _lib.ptr(b), _lib.ptr(y), _lib.ptr(sums)
    _lib.check(hiplib.stpde_conv3d_fused(C.byref(a), None, _lib.stream_ptr()))
    ref = _conv64(x, w, b)
    assert _rel(y, ref) < 2e-6
    s1, s2 = _sums(sums)
    r = ref.reshape(-1, co)
    assert _rel(s1, r.sum(0)) < 2e-6 and _rel(s2, (r * r).sum(0)) < 2e-6


@pytest.mark.parametrize("shape", SHAPES)
def test_two_outputs_with_statistics_of_the_first(hiplib, shape):
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    ci, c1, c2 = 32, 16, 32
    x = torch.randn(*shape, ci, device=dev)
    w1, w2 = 0.2 * torch.randn(c1, ci, 1, 1, 1, device=dev), 0.2 * torch.randn(c2, ci, 1, 1, 1, device=dev)
    b1, b2 = torch.randn(c1, device=dev), torch.randn(c2, device=dev)
    y1, y2 = torch.empty(*shape, c1, device=dev), torch.empty(*shape, c2, device=dev)
    sums = torch.zeros(R * 2 * c1, device=dev, dtype=torch.float64)
    p1, p2 = _packs(w1, dev)[0], _packs(w2, dev)[0]            # (kept alive: the struct holds raw pointers)
    a = _args(shape, ci, c1, 1)
    a.x, a.w_pack, a.bias, a.y = _lib.ptr(x), _lib.ptr(p1), _lib.ptr(b1), _lib.ptr(y1)
    a.y2, a.wo2_pack, a.bias2, a.Co2, a.out_sums = _lib.ptr(y2), _lib.ptr(p2), _lib.ptr(b2), c2, _lib.ptr(sums)
    _lib.check(hiplib.stpde_conv3d_fused(C.byref(a), None, _lib.stream_ptr()))
    r1, r2 = _conv64(x, w1, b1), _conv64(x, w2, b2)
    assert _rel(y1, r1) < 2e-6 and _rel(y2, r2) < 2e-6
    s1, s2 = _sums(sums)
    assert _rel(s1, r1.reshape(-1, c1).sum(0)) < 2e-6 and _rel(s2, (r1 * r1).reshape(-1, c1).sum(0)) < 2e-6


@pytest.mark.parametrize("shape", SHAPES)
def test_two_inputs(hiplib, shape):
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    c1, c2, co = 16, 32, 32
    x1, x2 = torch.randn(*shape, c1, device=dev), torch.randn(*shape, c2, device=dev)
    w1, w2 = 0.2 * torch.randn(co, c1, 1, 1, 1, device=dev), 0.2 * torch.randn(co, c2, 1, 1, 1, device=dev)
    y = torch.empty(*shape, co, device=dev)
    p1, p2 = _packs(w1, dev)[0], _packs(w2, dev)[0]
    a = _args(shape, c1, co, 1)
    a.x, a.w_pack, a.y = _lib.ptr(x1), _lib.ptr(p1), _lib.ptr(y)
    a.x2, a.w2_pack, a.Ci2 = _lib.ptr(x2), _lib.ptr(p2), c2
    _lib.check(hiplib.stpde_conv3d_fused(C.byref(a), None, _lib.stream_ptr()))
    assert _rel(y, _conv64(x1, w1, None) + _conv64(x2, w2, None)) < 2e-6


@pytest.mark.parametrize("shape", SHAPES)
def test_batchnorm_relu_on_load_and_its_weight_gradient(hiplib, shape):
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    ci, co, eps, mom = 32, 16, 1e-5, 0.1
    n = shape[0] * shape[1] * shape[2] * shape[3]
    x = 1.5 * torch.randn(*shape, ci, device=dev) + 0.4
    gam, bet = torch.rand(ci, device=dev) + 0.5, 0.3 * torch.randn(ci, device=dev)
    rm, rv = torch.randn(ci, device=dev), torch.rand(ci, device=dev) + 0.5
    rm0, rv0 = rm.clone(), rv.clone()
    w, b = 0.2 * torch.randn(co, ci, 1, 1, 1, device=dev), torch.randn(co, device=dev)
    xs = x.double().reshape(-1, ci)
    in_sums = torch.zeros(R, 2, ci, device=dev, dtype=torch.float64)
    in_sums[3, 0], in_sums[3, 1] = xs.sum(0), (xs * xs).sum(0)              # any replica: the consumers add them up
    stat, y = torch.empty(2 * ci, device=dev), torch.empty(*shape, co, device=dev)
    a = _args(shape, ci, co, 1)
    fp = _packs(w, dev)[0]
    a.x, a.w_pack, a.bias, a.y = _lib.ptr(x), _lib.ptr(fp), _lib.ptr(b), _lib.ptr(y)
    a.in_sums, a.in_gamma, a.in_beta, a.in_stat = _lib.ptr(in_sums), _lib.ptr(gam), _lib.ptr(bet), _lib.ptr(stat)
    a.in_running_mean, a.in_running_var, a.in_eps, a.in_momentum = _lib.ptr(rm), _lib.ptr(rv), eps, mom
    _lib.check(hiplib.stpde_conv3d_fused(C.byref(a), None, _lib.stream_ptr()))
    mean, var = xs.mean(0), xs.var(0, unbiased=False)
    h = torch.relu((x.double() - mean) / torch.sqrt(var + eps) * gam.double() + bet.double())
    assert _rel(y, _conv64(h, w, b)) < 5e-6
    assert _rel(stat[:ci], mean) < 1e-6 and _rel(stat[ci:], 1 / torch.sqrt(var + eps)) < 1e-6
    assert _rel(rm, (1 - mom) * rm0.double() + mom * mean) < 1e-6
    assert _rel(rv, (1 - mom) * rv0.double() + mom * var * n / (n - 1)) < 1e-6
    # weight / bias gradient of that convolution: the same transform on its operand
    gy = torch.randn(*shape, co, device=dev)
    dw, db = torch.zeros(1, co, ci, device=dev), torch.zeros(co, device=dev)
    d = unet3d._desc(x, ci, co, 1)
    _lib.check(hiplib.stpde_conv3d_wgrad_onload(C.byref(d), _lib.ptr(x), _lib.ptr(gy), _lib.ptr(dw), _lib.ptr(db), _lib.ptr(stat),
                                                _lib.ptr(gam), _lib.ptr(bet), _lib.stream_ptr()))
    ref = gy.double().reshape(-1, co).t() @ h.reshape(-1, ci)
    assert _rel(dw[0], ref) < 2e-5 and _rel(db, gy.double().reshape(-1, co).sum(0)) < 2e-5


@pytest.mark.parametrize("onload", [False, True])
@pytest.mark.parametrize("ci,co", [(16, 16), (16, 32), (16, 64), (32, 16), (32, 32), (32, 64), (64, 16), (64, 32), (64, 64)])
def test_conv1_weight_gradient_from_lds_tiles_ragged_volume(hiplib, ci, co, onload):
    """k_conv1_wgrad_lds (csrc/conv3d.hip, round 5): the 1x1x1 weight / bias gradient of volumes >= 65,536 voxels from
    256-voxel LDS tiles, every channel-tile combination it is compiled for, plain and with BatchNorm + ReLU on load, on a
    volume that is NOT a whole number of tiles (the last tile is zero-filled -- also AFTER the on-load transform, whose
    image of 0 is not 0)."""
    dev = torch.device("cuda:0")
    torch.manual_seed(11 + ci + 3 * co)
    shape = (1, 5, 111, 119)                            # 66,045 voxels = 257 tiles + 253 voxels
    x = 1.5 * torch.randn(*shape, ci, device=dev) + 0.4
    gy = torch.randn(*shape, co, device=dev)
    gam, bet = torch.rand(ci, device=dev) + 0.5, 0.3 * torch.randn(ci, device=dev) + 0.2
    xs = x.double().reshape(-1, ci)
    mean, rstd = xs.mean(0), 1 / torch.sqrt(xs.var(0, unbiased=False) + 1e-5)
    stat = torch.cat([mean, rstd]).float().contiguous()
    dw, db = torch.zeros(1, co, ci, device=dev), torch.zeros(co, device=dev)
    d = unet3d._desc(x, ci, co, 1)
    with _lib.dispatch_trace() as tr:
        if onload:
            _lib.check(hiplib.stpde_conv3d_wgrad_onload(C.byref(d), _lib.ptr(x), _lib.ptr(gy), _lib.ptr(dw), _lib.ptr(db),
                                                        _lib.ptr(stat), _lib.ptr(gam), _lib.ptr(bet), _lib.stream_ptr()))
        else:
            _lib.check(hiplib.stpde_conv3d_wgrad_bias(C.byref(d), _lib.ptr(x), _lib.ptr(gy), _lib.ptr(dw), _lib.ptr(db),
                                                      _lib.stream_ptr()))
        torch.cuda.synchronize()
    assert tr.has("k_conv1_wgrad_lds<%d, %d, %s>" % (ci // 16, co // 16, "true" if onload else "false")), "\n".join(tr.kernels)
    h = x.double()
    if onload:
        h = torch.relu((h - stat[:ci].double()) * (stat[ci:].double() * gam.double()) + bet.double())
    ref = gy.double().reshape(-1, co).t() @ h.reshape(-1, ci)
    assert _rel(dw[0], ref) < 2e-5 and _rel(db, gy.double().reshape(-1, co).sum(0)) < 2e-5


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("k,c", [(1, 32), (3, 16), (3, 32)])
def test_mask_and_batchnorm_backward_sums_epilogue(hiplib, shape, k, c):
    """input-gradient convolution: y = conv(x) is the gradient of relu(bn(m)); stored dz = y * [bn(m) > 0] and the sums the
    BatchNorm backward needs.  Elements whose fp64 pre-activation is within 1e-5 of the kink are excluded (mask flips)."""
    dev = torch.device("cuda:0")
    torch.manual_seed(4 + k + c)
    ci = 32
    x = torch.randn(*shape, ci, device=dev)
    w = 0.2 * torch.randn(c, ci, k, k, k, device=dev)
    m = torch.randn(*shape, c, device=dev) + 0.3
    gam, bet = torch.rand(c, device=dev) + 0.5, 0.3 * torch.randn(c, device=dev)
    ms = m.double().reshape(-1, c)
    mean, rstd = ms.mean(0), 1 / torch.sqrt(ms.var(0, unbiased=False) + 1e-5)
    stat = torch.cat([mean, rstd]).float().contiguous()
    y = torch.empty(*shape, c, device=dev)
    bsum = torch.zeros(R * 2 * c, device=dev)
    a = _args(shape, ci, c, k)
    fp = _packs(w, dev)[0]
    a.x, a.w_pack, a.y = _lib.ptr(x), _lib.ptr(fp), _lib.ptr(y)
    a.m, a.m_stat, a.m_gamma, a.m_beta, a.m_bsum = _lib.ptr(m), _lib.ptr(stat), _lib.ptr(gam), _lib.ptr(bet), _lib.ptr(bsum)
    done = C.c_int(1)
    _lib.check(hiplib.stpde_conv3d_fused(C.byref(a), C.byref(done), _lib.stream_ptr()))
    conv = _conv64(x, w, None)
    xhat = (m.double() - stat[:c].double()) * stat[c:].double()
    pre = xhat * gam.double() + bet.double()
    if not done.value:                       # tap-split volume: the unmasked gradient, the caller runs stpde_bn_bwd
        assert k == 3 and _rel(y, conv) < 2e-6
        return
    safe = pre.abs() > 1e-5
    dz = conv * (pre > 0)
    assert ((y.double() - dz).abs() * safe).max().item() < 2e-6 * conv.abs().max().item()
    tot = bsum.view(R, 2, c).sum(0).double()
    slack = ((~safe) * conv.abs()).reshape(-1, c).sum(0) * (1 + xhat.abs().max()) + 1e-30
    e1 = (tot[0] - dz.reshape(-1, c).sum(0)).abs()
    e2 = (tot[1] - (dz * xhat).reshape(-1, c).sum(0)).abs()
    scale = dz.abs().reshape(-1, c).sum(0)
    assert (e1 <= 2e-6 * scale + slack).all() and (e2 <= 1e-5 * scale * (1 + xhat.abs().max()) + slack).all()


@pytest.mark.parametrize("c", [16, 32, 64])
@pytest.mark.parametrize("epi", ["plain", "stats", "mask"])
def test_conv3_from_lds_halo_tiles_equals_the_per_wave_kernel(hiplib, monkeypatch, c, epi):
    """k_conv3_lds (csrc/conv3d_fused.hip, round 5): the square 3x3x3 convolutions of the wide levels from LDS halo tiles,
    on a batch of TWO samples (a block's halo must stop at the sample boundary and at every face of the volume) with more
    blocks than workgroups (2 x 8 x 32 x 64 voxels = 128 .. 512 blocks on 24 persistent workgroups -- stpde_tune "conv3_lds_gx" --
    5 - 22 blocks each through the register-pipelined staging; "conv3_lds_minblk" = 1 puts the kernel on this small volume).  Same MFMA order per output element as k_conv_fused: the outputs must be bit-identical;
    the epilogue sums (fp64 per wave over its blocks, one set of atomics per workgroup) against torch fp64."""
    dev = torch.device("cuda:0")
    torch.manual_seed(40 + c)
    shape = (2, 8, 32, 64)
    x = torch.randn(*shape, c, device=dev) + 0.3
    w = 0.2 * torch.randn(c, c, 3, 3, 3, device=dev)
    b = torch.randn(c, device=dev)
    m = torch.randn(*shape, c, device=dev) + 0.3
    gam, bet = torch.rand(c, device=dev) + 0.5, 0.3 * torch.randn(c, device=dev)
    ms = m.double().reshape(-1, c)
    stat = torch.cat([ms.mean(0), 1 / torch.sqrt(ms.var(0, unbiased=False) + 1e-5)]).float().contiguous()
    fp = _packs(w, dev)[0]
    out = {}
    for lds in ("1", "0"):
        tuned = _lib.tuned(conv3_lds_off=int(lds == "0"), conv3_lds_minblk=1, conv3_lds_gx=24)
        tuned.__enter__()
        y = torch.full((*shape, c), float("nan"), device=dev)
        sums = torch.zeros(R * 2 * c, device=dev, dtype=torch.float64)
        bsum = torch.zeros(R * 2 * c, device=dev)
        a = _args(shape, c, c, 3)
        a.x, a.w_pack, a.y = _lib.ptr(x), _lib.ptr(fp), _lib.ptr(y)
        if epi != "mask":
            a.bias = _lib.ptr(b)
        if epi == "stats":
            a.out_sums = _lib.ptr(sums)
        if epi == "mask":
            a.m, a.m_stat, a.m_gamma, a.m_beta, a.m_bsum = _lib.ptr(m), _lib.ptr(stat), _lib.ptr(gam), _lib.ptr(bet), _lib.ptr(bsum)
        done = C.c_int(1)
        with _lib.dispatch_trace() as tr:
            _lib.check(hiplib.stpde_conv3d_fused(C.byref(a), C.byref(done), _lib.stream_ptr()))
            torch.cuda.synchronize()
        tuned.__exit__(None, None, None)
        assert done.value == 1
        assert tr.has("k_conv3_lds<%d, %d>" % (c // 16, ("plain", "stats", "mask").index(epi))) == (lds == "1"), "\n".join(tr.kernels)
        assert tr.has("k_conv_fused<") == (lds == "0"), "\n".join(tr.kernels)
        out[lds] = (y, sums.clone(), bsum.clone())
    assert torch.equal(out["1"][0], out["0"][0])
    conv = _conv64(x, w, None if epi == "mask" else b)
    if epi == "plain":
        assert _rel(out["1"][0], conv) < 2e-6
    if epi == "stats":
        s1, s2 = _sums(out["1"][1])
        r = conv.reshape(-1, c)
        assert _rel(s1, r.sum(0)) < 2e-6 and _rel(s2, (r * r).sum(0)) < 2e-6
    if epi == "mask":
        t1, t0 = out["1"][2].view(R, 2, c).sum(0).double(), out["0"][2].view(R, 2, c).sum(0).double()
        dz = out["1"][0].double().reshape(-1, c)            # (the stored, masked gradient: identical in both kernels)
        xhat = ((m.double() - stat[:c].double()) * stat[c:].double()).reshape(-1, c)
        scale = dz.abs().sum(0)
        assert ((t1[0] - dz.sum(0)).abs() <= 2e-6 * scale + 1e-4).all()
        assert ((t1[1] - (dz * xhat).sum(0)).abs() <= 1e-5 * scale * (1 + xhat.abs().max()) + 1e-4).all()
        assert ((t1 - t0).abs() <= 1e-5 * scale * (1 + xhat.abs().max()) + 1e-4).all()
