"""GPU parity on the round-2 reference fixtures and on the EXACT kernel instantiations the benchmark times.

  * G5b: reference outputs / gradients at nf = 32 (softplus -> combined stream S = (3,1); leaky-relu -> S = (3,0)) with
    several launch chunks; the dispatch trace of the C library proves that k_wgrad_coop<3,1,MODE 1,KC 8>, the weight-ring
    (WRING) forward / dgrad kernels of the widest layer and the pass-split dgrad were the kernels under test;
  * fp64-oracle backward of the same instantiations (VERDICT r1 "weak" #1);
  * G8 = BASELINE configs[0] end to end through ``sharded_step`` on the HIP path;
  * configs[4]: backward of the user-string 5-channel equation set ((3,6) streams through k_residual_bwd) vs the oracle;
  * N3 / N4 fixtures on the device (data loader; reference-written checkpoint + FusedClipAdam resume).
"""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import test_reference_fixtures as F  # noqa: E402  (shared fixture plumbing)

from oracle import cpu_ref as O  # noqa: E402
from oracle import jet_ref as J  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _relerr(a, b):
    return F.rel(a.detach().cpu(), b)


def _normerr(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).double()
    return (a - b).norm().item() / max(b.norm().item(), 1e-30)


def _assert_benchmarked_kernels(tr, act):
    """The instantiations BENCH times for nf = 32 (profiles/r*_kernel_trace_stats.txt)."""
    s2 = 1 if act == "softplus" else 0
    cfg = "S1 = 3, S2 = %d" % s2
    dump = "\n".join(tr.kernels)
    # first hidden layer forward: layer 0 regenerated (PRO = 2), 4 output tiles per wave, weight ring
    assert tr.has("k_layer_coop", "1, true>)", cfg, "MCg = 4", "PRO = 2", "EPI = 0", "NW = 4"), dump
    # its dgrad into layer 0 (EPI = 2) and the dgrad of layer 2 (EPI = 1), weight ring, pass-split
    assert tr.has("k_layer_coop", "1, true>)", cfg, "MCg = 4", "PRO = 0", "EPI = 2"), dump
    assert tr.has("k_layer_coop", "1, true>)", cfg, "MCg = 4", "PRO = 0", "EPI = 1"), dump
    # layer-2 forward (activation-jet prologue)
    assert tr.has("k_layer_coop", cfg, "PRO = 1", "EPI = 0"), dump
    # weight gradients: first hidden layer (ring kernel, MODE = 1, KC = 8: ONE launch -- the raw-input k-tiles are folded into
    # the hidden k-groups, no HASX = true launch for this stream set; the value-stream launch of layer 0, S1 = S2 = 0, is the
    # only raw-input-only one left), layer 2 (eight waves on one row tile) and fc3 (four waves on one row tile), fc4 / fc5
    assert tr.has("k_wgrad_coop", "false>)", cfg, "MODE = 1", "KC = 8"), dump
    assert not tr.has("k_wgrad_coop", "true>)", cfg), dump
    assert tr.has("k_wgrad_quad", "false, 8>)", cfg), dump
    assert tr.has("k_wgrad_quad", "false>)", cfg), dump
    assert tr.has("k_wgrad_wave", cfg), dump
    assert tr.has("k_gather") and tr.has("k_xbar") and tr.has("k_reduce_bwd"), dump


@pytest.mark.parametrize("act", ["softplus", "leakyrelu"])
def test_g5b_reference_vectors_at_benchmarked_width(hiplib, golden_dir, act, monkeypatch):
    from space_time_pde_amd import _lib, lig_jet, local_implicit_grid as lig, physics
    d = np.load(os.path.join(golden_dir, "g5b_nf32.npz"))
    lat0, pts, tgt = F.g5b_inputs()
    net = F.make_imnet(act, 32, 524).to(DEV)
    lat = lat0.to(DEV).requires_grad_(True)
    monkeypatch.setattr(lig_jet, "DEFAULT_CHUNK", 96)          # 256 points -> 3 launch chunks (96 + 96 + 64)
    layer = physics.get_rb2_pde_layer(**F.RB2)
    layer.update_forward_method(lambda p: lig.query_local_implicit_grid(net, lat, p, torch.zeros(3, device=DEV),
                                                                        torch.ones(3, device=DEV)))
    n0 = lig.stats["hip_jet_calls"]
    with _lib.dispatch_trace() as tr:
        pred, res = layer(pts.to(DEV), return_residue=True)
        reg = torch.nn.functional.l1_loss(pred, tgt.to(DEV))
        st = torch.stack(list(res.values()), 0)
        pl = torch.nn.functional.l1_loss(st, torch.zeros_like(st))
        (1.0 * reg + 0.0125 * pl).backward()
        torch.cuda.synchronize()
    assert lig.stats["hip_jet_calls"] == n0 + 1
    _assert_benchmarked_kernels(tr, act)
    assert _relerr(pred, d[act + "_pred"]) < 2e-5
    for k, v in res.items():
        ref = torch.from_numpy(d["%s_res_%s" % (act, k)]).double()
        err = (v.detach().double().cpu() - ref).abs() / ref.abs().max()
        assert err.median().item() < 1e-5 and (err < 1e-3).double().mean().item() > 0.98, k
    assert abs(pl.item() - float(d[act + "_pde_loss"])) < 1e-5 * float(d[act + "_pde_loss"])
    assert abs(reg.item() - float(d[act + "_reg_loss"])) < 1e-5 * float(d[act + "_reg_loss"])
    pl_act = act == "leakyrelu"       # kink flips: Frobenius norm + loose max bound (as test_golden_composite_g5)
    err = _normerr if pl_act else _relerr
    tol = 5e-3 if pl_act else 5e-4

    def close(a, b, what):
        assert err(a, b) < tol, what
        assert _relerr(a, b) < (5e-2 if pl_act else tol), what

    close(lat.grad, d[act + "_dlatent"], "dlatent")
    for k in range(6):
        gw = net.fc[k].weight.grad
        close(gw[::3] if k < 2 else gw, d["%s_dw%d" % (act, k)], "dW%d" % k)
        close(net.fc[k].bias.grad, d["%s_db%d" % (act, k)], "db%d" % k)
        nref = float(d["%s_dw%d_norm" % (act, k)])
        assert abs(gw.norm().item() - nref) < tol * nref, "norm dW%d" % k


def _assert_mode_kernels(tr, prec):
    """The bf16-pipe instantiations the configs[3] / fp32x3 bench lines time for nf = 32 softplus: combined stream
    S = (3,1) (VERDICT r2 weak #1a: the earlier bf16 oracle test ran S = (3,2) only)."""
    cfg = "S1 = 3, S2 = 1"
    dump = "\n".join(tr.kernels)
    if prec == "fp32x3":
        for pro, epi in ((2, 0), (0, 2), (1, 0), (0, 1)):
            assert tr.has("k_layer_coop", "1, false, 3>)", cfg, "PRO = %d" % pro, "EPI = %d" % epi), dump
        # round 5: the split weight gradient of the first hidden layer with the raw-input k-tiles folded in (no separate HASX
        # launch any more); the second hidden layer's on the exact-fp32 eight-wave kernel (faster than its split ring kernel +
        # raw-input launch; "fp32x3" is a contract on accuracy, not on the pipe)
        assert tr.has("k_wgrad_coop", "false, true, 3>)", cfg, "MODE = 1", "KC = 8"), dump
        assert not tr.has("k_wgrad_coop", "true, true, 3>)", cfg, "MODE = 1"), dump
        assert tr.has("k_wgrad_quad", "0, false, 8>)", cfg, "MODE = 0"), dump
    else:
        _assert_bf16_kernels(tr, cfg, dump)
    assert tr.has("k_tail_fwd") and tr.has("k_tail_bwd") and tr.has("k_wgrad_wave", cfg), dump


def _assert_bf16_kernels(tr, cfg, dump):
    # first hidden layer forward: the wave-specialised persistent kernel (csrc/jet_spec_bf16.h)
    assert tr.has("k_fc1_fwd_spec", cfg), dump
    # second hidden layer: forward on the persistent LDS-DMA kernel (round 5; the cooperative kernel for the stream sets it is not
    # compiled for), input gradient on the cooperative kernel
    assert tr.has("k_fc2_fwd_bf", cfg) or tr.has("k_layer_coop", "true", cfg, "PRO = 1", "EPI = 0"), dump
    assert tr.has("k_layer_coop", "true", cfg, "PRO = 0", "EPI = 1"), dump
    if tr.has("k_fc1_bwd_fused"):
        # round 5 (default): input gradient + weight gradient of the first hidden layer in ONE kernel (csrc/jet_fc1_bwd.hip);
        # neither of the two kernels it replaces is launched
        assert not tr.has("k_wgrad_coop", "KC, false, true", cfg, "MODE = 1"), dump
    else:
        # STPDE_FC1_FUSED=0 (or a stream set the fused kernel does not serve): the cooperative input-gradient kernel + the ring
        # kernel with the raw-input k-tiles folded into the hidden-group launch (no HASX = true launch)
        assert tr.has("k_layer_coop", "true", cfg, "PRO = 0", "EPI = 2"), dump
        assert tr.has("k_wgrad_coop", "KC, false, true", cfg, "MODE = 1", "KC = 8"), dump
        assert not tr.has("k_wgrad_coop", "KC, true, true", cfg), dump
    assert tr.has("k_wgrad_oct_bf", "ACT>)", cfg), dump       # layer 2: eight waves on one row tile, bf16 blocks in LDS (round 4)
    assert tr.has("k_wgrad_oct_bf", "ACT, 4>)", cfg), dump    # fc3: four waves


@pytest.mark.parametrize("prec", ["fp32x3", "bf16"])
def test_g5b_reference_vectors_in_bf16_pipe_modes(hiplib, golden_dir, prec, monkeypatch):
    """G5b (outputs of the imported reference at the benchmarked width nf = 32, softplus -> combined stream S = (3,1), three
    launch chunks) through the two bf16-pipe modes.  "fp32x3" is judged with the exact-fp32 tolerances of
    test_g5b_reference_vectors_at_benchmarked_width including the 1e-5 loss bound; "bf16" (BASELINE configs[3]) with the
    mode's own tolerances (3e-2 Frobenius vs exact).  The dispatch trace pins the instantiations."""
    from space_time_pde_amd import _lib, lig_jet, local_implicit_grid as lig, physics
    act = "softplus"
    d = np.load(os.path.join(golden_dir, "g5b_nf32.npz"))
    lat0, pts, tgt = F.g5b_inputs()
    net = F.make_imnet(act, 32, 524).to(DEV)
    lat = lat0.to(DEV).requires_grad_(True)
    monkeypatch.setattr(lig_jet, "DEFAULT_CHUNK", 96)          # 256 points -> 3 launch chunks (96 + 96 + 64)
    monkeypatch.setattr(lig_jet, "mlp_precision", prec)
    layer = physics.get_rb2_pde_layer(**F.RB2)
    layer.update_forward_method(lambda p: lig.query_local_implicit_grid(net, lat, p, torch.zeros(3, device=DEV),
                                                                        torch.ones(3, device=DEV)))
    n0 = lig.stats["hip_jet_calls"]
    with _lib.dispatch_trace() as tr:
        pred, res = layer(pts.to(DEV), return_residue=True)
        reg = torch.nn.functional.l1_loss(pred, tgt.to(DEV))
        st = torch.stack(list(res.values()), 0)
        pl = torch.nn.functional.l1_loss(st, torch.zeros_like(st))
        (1.0 * reg + 0.0125 * pl).backward()
        torch.cuda.synchronize()
    assert lig.stats["hip_jet_calls"] == n0 + 1
    _assert_mode_kernels(tr, prec)
    exact = prec == "fp32x3"
    assert (_relerr(pred, d[act + "_pred"]) < 2e-5) if exact else (_normerr(pred, d[act + "_pred"]) < 3e-2)
    for k, v in res.items():
        ref = torch.from_numpy(d["%s_res_%s" % (act, k)]).double()
        if exact:
            err = (v.detach().double().cpu() - ref).abs() / ref.abs().max()
            assert err.median().item() < 1e-5 and (err < 1e-3).double().mean().item() > 0.98, k
        else:
            assert _normerr(v, ref) < 3e-2, k
    ltol = 1e-5 if exact else 3e-2
    assert abs(pl.item() - float(d[act + "_pde_loss"])) < ltol * float(d[act + "_pde_loss"])
    assert abs(reg.item() - float(d[act + "_reg_loss"])) < ltol * float(d[act + "_reg_loss"])
    tol = 5e-4 if exact else 3e-2
    err = _relerr if exact else _normerr
    assert err(lat.grad, d[act + "_dlatent"]) < tol, "dlatent"
    for k in range(6):
        gw = net.fc[k].weight.grad
        assert err(gw[::3] if k < 2 else gw, d["%s_dw%d" % (act, k)]) < tol, "dW%d" % k
        assert err(net.fc[k].bias.grad, d["%s_db%d" % (act, k)]) < tol, "db%d" % k
        nref = float(d["%s_dw%d_norm" % (act, k)])
        assert abs(gw.norm().item() - nref) < tol * nref, "norm dW%d" % k


def test_recompute_mode_gives_bit_identical_gradients(hiplib, monkeypatch):
    """Memory plan of LigJetFunction: when the stash does not fit, the forward keeps none and the backward
    re-runs the forward kernels chunk by chunk.  Forced here; jets and every gradient must equal the stash-keeping run
    bit for bit (same kernels, same inputs, deterministic d latent) -- weight gradients to fp32-atomic rounding."""
    from space_time_pde_amd import implicit_net, lig_jet, nonlinearities
    g = torch.Generator().manual_seed(41)
    lat = 0.5 * torch.randn(1, 4, 5, 6, 32, generator=g)
    pts = 0.02 + 0.96 * torch.rand(1, 700, 3, generator=g)
    torch.manual_seed(6)
    net = implicit_net.ImNet(dim=3, in_features=32, out_features=4, nf=32,
                             activation=nonlinearities.NONLINEARITIES["softplus"]).to(DEV)
    combo = {(1, 1): 1.0, (2, 2): 0.25}
    out = []
    cot = None
    for force in (False, True):
        monkeypatch.setattr(lig_jet, "force_recompute", force)
        for p in net.parameters():
            p.grad = None
        latd = lat.to(DEV).requires_grad_(True)
        n0 = lig_jet.stats["recompute_steps"]
        jets, _ = lig_jet.lig_jets(net, latd, pts.to(DEV), 0., 1., True, (), chunk_points=256, combo=combo)
        assert lig_jet.stats["recompute_steps"] == n0 + int(force)
        if cot is None:
            cot = torch.randn(jets.shape, generator=g).to(DEV)
        (jets * cot).sum().backward()
        out.append((jets.detach().clone(), latd.grad.clone(), [p.grad.clone() for p in net.parameters()]))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    for a, b in zip(out[0][2], out[1][2]):
        assert (a - b).abs().max().item() <= 5e-6 * a.abs().max().item()   # fp32-atomic summation order


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("act", ["softplus", "leakyrelu"])
def test_one_call_per_direction_equals_the_per_kernel_path(hiplib, act, prec, monkeypatch):
    """stpde_lig_imnet_jet_fwd / _bwd (one C call per direction, cell sort on the device) against the per-layer entry points
    sequenced from Python: the same kernels in the same order -> identical jets and d latent bit for bit, weight gradients
    to fp32-atomic rounding.  Several chunks, an odd point count, many points per cell.  "bf16": the mode with packed layer
    buffers (stash / adjoint formats, separate adjoint buffers) -- both hosts of it must lay the buffers out the same way."""
    from space_time_pde_amd import _lib, implicit_net, lig_jet, nonlinearities
    g = torch.Generator().manual_seed(51)
    lat = 0.5 * torch.randn(2, 3, 4, 5, 32, generator=g)
    pts = torch.rand(2, 1501, 3, generator=g)
    torch.manual_seed(7)
    net = implicit_net.ImNet(dim=3, in_features=32, out_features=4, nf=32,
                             activation=nonlinearities.NONLINEARITIES[act]).to(DEV)
    combo = {(1, 1): 1.0, (2, 2): 0.25}
    out, cot = [], None
    for pipe in (False, True):
        monkeypatch.setattr(lig_jet, "use_pipeline", pipe)
        for p in net.parameters():
            p.grad = None
        latd = lat.to(DEV).requires_grad_(True)
        with _lib.dispatch_trace() as tr:
            jets, _ = lig_jet.lig_jets(net, latd, pts.to(DEV), 0., 1., True, (), chunk_points=1024, combo=combo,
                                       precision=prec)
            if cot is None:
                cot = torch.randn(jets.shape, generator=g).to(DEV)
            (jets * cot).sum().backward()
            torch.cuda.synchronize()
        assert tr.has("k_dlat_reduce") and tr.has("k_gather") and tr.has("k_tail_bwd"), "\n".join(tr.kernels)
        assert tr.has("k_tail_bwd_bf") == (prec == "bf16"), "\n".join(tr.kernels)
        out.append((jets.detach().clone(), latd.grad.clone(), [p.grad.clone() for p in net.parameters()]))
    assert torch.equal(out[0][0], out[1][0])
    assert torch.equal(out[0][1], out[1][1])
    for a, b in zip(out[0][2], out[1][2]):
        assert (a - b).abs().max().item() <= 5e-6 * a.abs().max().item()   # fp32-atomic summation order


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("act", ["softplus", "leakyrelu", "swish"])
def test_dgrad_first_two_phase_backward_equals_one_call(hiplib, act, prec):
    """The order the point-sharded step uses (train_step.sharded_step -> lig_jet.sync_hooks): phase A = the whole
    input-gradient chain into fresh adjoint buffers + d latent, then the collective hook, then phase B = the weight gradients
    (stpde_lig_imnet_jet_bwd with STPDE_F_PHASE_A / _B, stpde_jet_layer_bwd_to).  With no-op hooks on one rank the result must
    equal the one-call order: d latent bit for bit, weight gradients to fp32-atomic rounding; the hooks must see d latent
    complete (first hook) and the finished flat IM-NET gradient."""
    from space_time_pde_amd import _lib, implicit_net, lig_jet, nonlinearities
    g = torch.Generator().manual_seed(61)
    lat = 0.5 * torch.randn(1, 4, 5, 6, 32, generator=g)
    pts = 0.02 + 0.96 * torch.rand(1, 900, 3, generator=g)
    torch.manual_seed(8)
    net = implicit_net.ImNet(dim=3, in_features=32, out_features=4, nf=32,
                             activation=nonlinearities.NONLINEARITIES[act]).to(DEV)
    combo = {(1, 1): 1.0, (2, 2): 0.25}
    out, cot, seen = [], None, []
    for two_phase in (False, True):
        for p in net.parameters():
            p.grad = None
        latd = lat.to(DEV).requires_grad_(True)
        jets, _ = lig_jet.lig_jets(net, latd, pts.to(DEV), 0., 1., True, (), chunk_points=512, combo=combo, precision=prec)
        if cot is None:
            cot = torch.randn(jets.shape, generator=g).to(DEV)
        if two_phase:
            lig_jet.sync_hooks = dict(dlatent=lambda t: seen.append(("dlatent", t.clone())) or None,
                                      dw=lambda t: seen.append(("dw", t.numel())) or None)
        try:
            with _lib.dispatch_trace() as tr:
                (jets * cot).sum().backward()
                torch.cuda.synchronize()
        finally:
            hooks, lig_jet.sync_hooks = lig_jet.sync_hooks, None
        if two_phase:
            assert hooks["used"] and hooks["dlatent_done"] is not None and hooks["dw_done"]
            assert tr.has("k_layer_coop", "EPI = 1") and tr.has("k_tail_bwd"), "\n".join(tr.kernels)
        out.append((latd.grad.clone(), [p.grad.clone() for p in net.parameters()]))
    assert seen[0][0] == "dlatent" and torch.equal(seen[0][1], out[1][0])          # complete when the hook fires
    assert torch.equal(out[0][0], out[1][0])
    for a, b in zip(out[0][1], out[1][1]):
        assert (a - b).abs().max().item() <= 5e-6 * a.abs().max().item()   # fp32-atomic summation order


def test_cell_sort_on_device_matches_torch(hiplib):
    """stpde_lig_cell_sort: stable order of the points by cell + exclusive cell offsets == torch.sort(stable) + bincount."""
    import ctypes as C
    from space_time_pde_amd import _lib
    g = torch.Generator().manual_seed(5)
    for P, n_nodes in ((4096, 300), (100000, 2 * 32 * 128 * 128), (2, 9)):
        cell = torch.randint(0, n_nodes - 1, (P,), generator=g, dtype=torch.int32).to(DEV)
        perm = torch.empty(P, device=DEV, dtype=torch.int32)
        start = torch.empty(n_nodes + 1, device=DEV, dtype=torch.int32)
        nb = int(hiplib.stpde_lig_sort_tmp_bytes(P, n_nodes))
        tmp = torch.empty(nb, device=DEV, dtype=torch.uint8)
        _lib.check(hiplib.stpde_lig_cell_sort(P, n_nodes, _lib.ptr(cell), _lib.ptr(perm), _lib.ptr(start), _lib.ptr(tmp), nb,
                                              _lib.stream_ptr()))
        want = torch.sort(cell, stable=True)[1].int()
        assert torch.equal(perm, want)
        cnt = torch.bincount(cell.long(), minlength=n_nodes)
        want_start = torch.cat([torch.zeros(1, device=DEV, dtype=torch.long), torch.cumsum(cnt, 0)]).int()
        assert torch.equal(start, want_start)


def test_data_parallel_wrapped_decoder_uses_the_hip_path(hiplib):
    """The reference wraps the IM-NET in nn.DataParallel (experiments/rb2d/train.py:352-355).  With several device ids the
    query points are split over the devices, each runs the HIP jet path on a replica, gradients flow back to the wrapped
    module (VERDICT r2 missing #5).  On the 1-GPU test box the two "devices" are both cuda:0 -- the same code path; values,
    residuals and every gradient must equal the unwrapped module's."""
    from space_time_pde_amd import implicit_net, local_implicit_grid as lig, nonlinearities, physics
    g = torch.Generator().manual_seed(71)
    lat0 = 0.5 * torch.randn(2, 4, 5, 6, 32, generator=g)
    pts = (0.02 + 0.96 * torch.rand(2, 301, 3, generator=g)).to(DEV)
    tgt = torch.randn(2, 301, 4, generator=g).to(DEV)
    torch.manual_seed(9)
    net = implicit_net.ImNet(dim=3, in_features=32, out_features=4, nf=16,
                             activation=nonlinearities.NONLINEARITIES["softplus"]).to(DEV)
    ids = [0, 1] if torch.cuda.device_count() >= 2 else [0, 0]
    res = []
    for wrap in (False, True):
        for p in net.parameters():
            p.grad = None
        model = torch.nn.DataParallel(net, device_ids=ids) if wrap else net
        lat = lat0.to(DEV).requires_grad_(True)
        layer = physics.get_rb2_pde_layer(**F.RB2)
        layer.update_forward_method(lambda q: lig.query_local_implicit_grid(model, lat, q, 0., 1.))
        n0 = lig.stats.get("data_parallel_calls", 0)
        pred, rs = layer(pts, return_residue=True)
        assert lig.stats.get("data_parallel_calls", 0) == n0 + int(wrap)
        loss = torch.nn.functional.l1_loss(pred, tgt) + 0.0125 * torch.stack(list(rs.values()), 0).abs().mean()
        loss.backward()
        res.append((pred.detach(), {k: v.detach() for k, v in rs.items()}, lat.grad.clone(),
                    [p.grad.clone() for p in net.parameters()]))
        with torch.no_grad():        # value-only query through the wrapper
            y = lig.query_local_implicit_grid(model, lat, pts, 0., 1.)
        assert (y - pred.detach()).abs().max().item() < 1e-6
    (p0, r0, gl0, gp0), (p1, r1, gl1, gp1) = res
    assert torch.equal(p0, p1)
    for k in r0:
        assert torch.equal(r0[k], r1[k]), k
    assert (gl0 - gl1).abs().max().item() <= 1e-5 * gl0.abs().max().item()
    for a, b in zip(gp0, gp1):
        assert (a - b).abs().max().item() <= 2e-5 * a.abs().max().item() + 1e-9


def test_wider_latent_takes_the_generic_path(hiplib):
    """an ImNet with 33..44 latent channels is outside the HIP envelope (3 + c + 1 <= 36) and must run the
    composed formulation instead of raising inside ImNetPlan."""
    from space_time_pde_amd import implicit_net, local_implicit_grid as lig
    net = implicit_net.ImNet(dim=3, in_features=40, out_features=4, nf=16, activation=torch.nn.Softplus).to(DEV)
    lat = torch.randn(1, 3, 4, 5, 40, device=DEV)
    pts = torch.rand(1, 64, 3, device=DEV)
    n0, h0 = lig.stats["generic_calls"], lig.stats["hip_value_calls"]
    y = lig.query_local_implicit_grid(net, lat, pts, 0., 1.)
    assert lig.stats["generic_calls"] == n0 + 1 and lig.stats["hip_value_calls"] == h0
    ref = O.query_lig(lambda f: net.cpu()(f), lat.cpu(), pts.cpu(), 0., 1.)
    assert _relerr(y, ref.detach()) < 1e-5


@pytest.mark.parametrize("act", ["softplus", "leakyrelu", "swish"])
def test_benchmarked_instantiations_backward_vs_fp64_oracle(hiplib, act):
    """nf = 32, combined second-order stream (leaky-relu: the MLP carries S = (3,0)), 3 launch chunks: forward jets and
    every gradient against the fp64 oracle, tolerances of test_backward_matches_oracle_autograd."""
    from space_time_pde_amd import _lib, implicit_net, lig_jet, nonlinearities
    g = torch.Generator().manual_seed(15)
    lat = 0.5 * torch.randn(2, 4, 5, 6, 32, generator=g)
    pts = 0.02 + 0.96 * torch.rand(2, 150, 3, generator=g)
    combo = {(1, 1): 1.0, (2, 2): 0.25}       # the RB2 anisotropic Laplacian pattern
    pairs = tuple(sorted(combo))
    torch.manual_seed(3)
    net = implicit_net.ImNet(dim=3, in_features=32, out_features=4, nf=32,
                             activation=nonlinearities.NONLINEARITIES[act]).to(DEV)
    beta64 = None
    if act == "swish":
        with torch.no_grad():
            net.activ.beta.fill_(1.3)
        beta64 = torch.tensor(1.3, dtype=torch.float64, requires_grad=True)
    latd = lat.to(DEV).requires_grad_(True)
    with _lib.dispatch_trace() as tr:
        jets, pp = lig_jet.lig_jets(net, latd, pts.to(DEV), 0., 1., True, (), chunk_points=128, combo=combo)
        cot = torch.randn(jets.shape, generator=g)
        (jets * cot.to(DEV)).sum().backward()
        torch.cuda.synchronize()
    if act != "swish":
        _assert_benchmarked_kernels(tr, act)
    p64 = [(net.fc[k].weight.detach().double().cpu().requires_grad_(True),
            net.fc[k].bias.detach().double().cpu().requires_grad_(True)) for k in range(6)]
    lat64 = lat.double().requires_grad_(True)
    full = J.lig_jets(p64, act, lat64, pts.double(), 0., 1., second=pairs, beta=beta64)
    full = full.permute(0, 3, 1, 2).reshape(full.shape[0], 4, -1)
    L = sum(combo[p] * full[4 + k] for k, p in enumerate(pairs))
    ref = torch.cat([full[:4], L[None]], 0)
    for s in range(5):
        assert _relerr(jets[s], ref[s].detach()) < 2e-5, "stream %d" % s
    (ref * cot.double()).sum().backward()
    err = _normerr if act == "leakyrelu" else _relerr
    assert err(latd.grad, lat64.grad) < 2e-4
    for k in range(6):
        assert err(net.fc[k].weight.grad, p64[k][0].grad) < 2e-4, "dW%d" % k
        assert err(net.fc[k].bias.grad, p64[k][1].grad) < 2e-4, "db%d" % k
    if act == "swish":
        assert abs(net.activ.beta.grad.item() - beta64.grad.item()) < 2e-4 * abs(beta64.grad.item())


@pytest.mark.parametrize("act", ["softplus", "leakyrelu"])
@pytest.mark.parametrize("prec", ["fp32", "fp32x3"])
def test_config0_c1_step_on_hip_matches_reference(hiplib, golden_dir, act, prec, monkeypatch):
    """BASELINE configs[0] (UNet3d(16,32,32) + 4096 points) through sharded_step on the HIP path vs the reference (G8)."""
    # the same test, same tolerances, with the wide layers' products as exact-split bf16 MFMAs ("fp32x3")
    from space_time_pde_amd import lig_jet as _lj
    monkeypatch.setattr(_lj, "mlp_precision", prec)
    from space_time_pde_amd import local_implicit_grid as lig, physics
    from space_time_pde_amd.train_step import sharded_step
    d = np.load(os.path.join(golden_dir, "g8_c1_step.npz"))
    unet, net = F.c1_models(act, DEV)
    crop, pts, tgt = F.c1_inputs(DEV)
    layer = physics.get_rb2_pde_layer(**F.RB2)
    n0 = lig.stats["hip_jet_calls"]
    loss, reg, pde = sharded_step(unet, net, layer, crop, pts, tgt, 4096, 1.0, 0.0125, "l1",
                                  xmin=torch.zeros(3, device=DEV), xmax=torch.ones(3, device=DEV), distributed=False)
    assert lig.stats["hip_jet_calls"] == n0 + 1
    pred, res, latent = F.c1_predictions(unet, net, layer, crop, pts)
    F.check_c1_step(d, act, unet, net, loss, reg, pde, pred, res, latent, slack=3.0 if act == "softplus" else 8.0)


@pytest.mark.parametrize("s2", [4, 6])
@pytest.mark.parametrize("prec", ["fp32", "fp32x3"])
def test_config4_user_equations_backward_vs_oracle(hiplib, golden_dir, prec, s2, monkeypatch):
    """BASELINE configs[4]: the 5-channel user-string equation set of G9 (products, a mixed second derivative, explicit
    coordinates -> the (3,6) stream set and k_residual_bwd): gradients of a random functional of prediction + residuals
    w.r.t. the latent grid and every IM-NET parameter vs the oracle's reverse-sweep autograd in fp64."""
    # the same test, same tolerances, with the wide layers' products as exact-split bf16 MFMAs ("fp32x3")
    # s2: the four second derivatives the strings name on the (3,4) stream set (round 5, the default) or padded to (3,6)
    from space_time_pde_amd import lig_jet as _lj
    monkeypatch.setattr(_lj, "mlp_precision", prec)
    monkeypatch.setenv("STPDE_S34", "1" if s2 == 4 else "0")
    from space_time_pde_amd import _lib, implicit_net, local_implicit_grid as lig, pde
    d = np.load(os.path.join(golden_dir, "g9_generic.npz"))
    net = implicit_net.ImNet(dim=3, in_features=32, out_features=5, nf=16, activation=torch.nn.Softplus).to(DEV)
    with torch.no_grad():
        for k in range(6):
            net.fc[k].weight.copy_(torch.from_numpy(d["w%d" % k]))
            net.fc[k].bias.copy_(torch.from_numpy(d["b%d" % k]))
    lat = torch.from_numpy(d["latent"]).to(DEV).requires_grad_(True)
    pts = torch.from_numpy(d["pts"])
    layer = pde.PDELayer("x, y, t", "c, u, v, w, p")
    for name, eq in zip(d["names"], d["eqs"]):
        layer.add_equation(str(eq), str(name))
    layer.update_forward_method(lambda q: lig.query_local_implicit_grid(net, lat, q, 0., 1.))
    g = torch.Generator().manual_seed(77)
    cot_y = torch.randn(2, 128, 5, generator=g)
    cot_r = {str(n): torch.randn(2, 128, 1, generator=g) for n in d["names"]}
    with _lib.dispatch_trace() as tr:
        pred, res = layer(pts.to(DEV))
        f = (pred * cot_y.to(DEV)).sum() + sum((res[k] * cot_r[k].to(DEV)).sum() for k in cot_r)
        f.backward()
        torch.cuda.synchronize()
    assert tr.has("k_residual_bwd") and tr.has("S1 = 3, S2 = %d" % s2), "\n".join(tr.kernels)
    assert not tr.has("S1 = 3, S2 = %d" % (10 - s2)), "\n".join(tr.kernels)
    # oracle: the reference's formulation (autograd dif sweeps) in float64
    p64 = [(torch.from_numpy(d["w%d" % k]).double().requires_grad_(True),
            torch.from_numpy(d["b%d" % k]).double().requires_grad_(True)) for k in range(6)]
    lat64 = torch.from_numpy(d["latent"]).double().requires_grad_(True)
    orc = O.PDEOracle("x, y, t", "c, u, v, w, p")
    for name, eq in zip(d["names"], d["eqs"]):
        orc.add_equation(str(eq), str(name))
    act = O.activation_fn("softplus")
    orc.forward_method = lambda q: O.query_lig(lambda x: O.imnet_forward(p64, x, act), lat64, q, 0., 1.)
    y64, r64 = orc(pts.double().clone())
    f64 = (y64 * cot_y.double()).sum() + sum((r64[k] * cot_r[k].double()).sum() for k in cot_r)
    f64.backward()
    assert abs(f.item() - f64.item()) < 2e-4 * abs(f64.item())
    assert _relerr(lat.grad, lat64.grad) < 3e-4
    for k in range(6):
        assert _relerr(net.fc[k].weight.grad, p64[k][0].grad) < 3e-4, "dW%d" % k
        assert _relerr(net.fc[k].bias.grad, p64[k][1].grad) < 3e-4, "db%d" % k


def test_dataloader_matches_reference_on_device(hiplib, golden_dir, tmp_path):
    from space_time_pde_amd import _lib
    with _lib.dispatch_trace() as tr:
        F.run_dataloader_fixture(golden_dir, tmp_path, "cuda:0")
        torch.cuda.synchronize()
    assert tr.has("k_interp"), "\n".join(tr.kernels)


def test_reference_written_checkpoint_resumes_on_hip_with_fused_adam(hiplib, golden_dir):
    """N4 + N1: load the reference's checkpoint (incl. its torch.optim.Adam state) into the HIP modules and
    FusedClipAdam, take the step the reference took after resuming, compare the updated parameters."""
    from space_time_pde_amd.optim import FusedClipAdam

    def make(params, lr, clip):
        return FusedClipAdam(params, lr=lr, clip_grad=clip)

    unet, net, opt, d = F.run_resume_fixture(golden_dir, DEV, make, 1e-4)
    assert opt.param_groups[0]["clip_grad"] == float(d["clip"])
    opt.step()
    F.check_after_step(unet, net, d, 3e-4)


def test_dlatent_is_bit_reproducible_and_matches_atomic_scatter(hiplib, monkeypatch):
    """d loss / d latent: per-node gather in a fixed order (k_dlat_reduce) -> two runs are bit-identical (the backward
    of the reference's index_put_(accumulate=True), regular_nd_grid_interpolation.py:65-66, is deterministic on its CPU
    path too); the fp32-atomic scatter kept for A/B timing agrees to rounding.  Many points per cell (coarse grid, 6000
    points, 3 chunks) so that the summation order matters."""
    from space_time_pde_amd import _lib, implicit_net, lig_jet, nonlinearities
    g = torch.Generator().manual_seed(31)
    lat = 0.5 * torch.randn(2, 3, 4, 5, 32, generator=g)
    pts = torch.rand(2, 3001, 3, generator=g)                     # odd count per batch element: padded tile
    torch.manual_seed(5)
    net = implicit_net.ImNet(dim=3, in_features=32, out_features=4, nf=16,
                             activation=nonlinearities.NONLINEARITIES["softplus"]).to(DEV)
    cot = None
    grads = []
    for mode in (True, True, False):
        monkeypatch.setattr(lig_jet, "deterministic_dlatent", mode)
        latd = lat.to(DEV).requires_grad_(True)
        with _lib.dispatch_trace() as tr:
            jets, _ = lig_jet.lig_jets(net, latd, pts.to(DEV), 0., 1., True, ((1, 1), (2, 2)), chunk_points=2048)
            if cot is None:
                cot = torch.randn(jets.shape, generator=g).to(DEV)
            (jets * cot).sum().backward()
            torch.cuda.synchronize()
        assert tr.has("k_dlat_reduce") == mode
        grads.append(latd.grad.clone())
    assert torch.equal(grads[0], grads[1])
    assert (grads[0] - grads[2]).abs().max().item() < 1e-5 * grads[0].abs().max().item()
    assert grads[0].abs().max().item() > 0


@pytest.mark.parametrize("act", ["softplus", "leakyrelu", "tanh"])
def test_fp32x3_split_mode_keeps_fp32_tolerances(hiplib, act):
    """"fp32x3": forward / input-gradient GEMMs of the wide layers on the bf16 MFMA pipe with every fp32 operand split
    exactly into three bf16 terms and six partial products accumulated in fp32.  Judged with the SAME tolerances as the
    exact-fp32 MFMA path (test_benchmarked_instantiations_backward_vs_fp64_oracle): it is an fp32-accurate product, not
    a reduced-precision mode (the plain "bf16" mode needs 3e-2 here)."""
    from space_time_pde_amd import _lib, implicit_net, lig_jet, nonlinearities
    g = torch.Generator().manual_seed(15)
    lat = 0.5 * torch.randn(2, 4, 5, 6, 32, generator=g)
    pts = 0.02 + 0.96 * torch.rand(2, 150, 3, generator=g)
    combo = {(1, 1): 1.0, (2, 2): 0.25}
    pairs = tuple(sorted(combo))
    torch.manual_seed(3)
    net = implicit_net.ImNet(dim=3, in_features=32, out_features=4, nf=32,
                             activation=nonlinearities.NONLINEARITIES[act]).to(DEV)
    res = {}
    for prec in ("fp32", "fp32x3"):
        for p in net.parameters():
            p.grad = None
        latd = lat.to(DEV).requires_grad_(True)
        with _lib.dispatch_trace() as tr:
            jets, pp = lig_jet.lig_jets(net, latd, pts.to(DEV), 0., 1., True, (), chunk_points=128, combo=combo,
                                        precision=prec)
            if "cot" not in res:
                res["cot"] = torch.randn(jets.shape, generator=g)
            (jets * res["cot"].to(DEV)).sum().backward()
            torch.cuda.synchronize()
        if prec == "fp32x3":      # the split kernels (last template argument 3) carried layers 1 and 2, forward and dgrad
            assert tr.has("k_layer_coop", "1, false, 3>)", "PRO = 2", "EPI = 0"), "\n".join(tr.kernels)
            assert tr.has("k_layer_coop", "1, false, 3>)", "PRO = 0", "EPI = 2"), "\n".join(tr.kernels)
            assert tr.has("k_layer_coop", "1, false, 3>)", "PRO = 1", "EPI = 0"), "\n".join(tr.kernels)
            assert tr.has("k_layer_coop", "1, false, 3>)", "PRO = 0", "EPI = 1"), "\n".join(tr.kernels)
        res[prec] = (jets.detach().clone(), latd.grad.clone(), [p.grad.clone() for p in net.parameters()])
    cot = res["cot"]
    p64 = [(net.fc[k].weight.detach().double().cpu().requires_grad_(True),
            net.fc[k].bias.detach().double().cpu().requires_grad_(True)) for k in range(6)]
    lat64 = lat.double().requires_grad_(True)
    full = J.lig_jets(p64, act, lat64, pts.double(), 0., 1., second=pairs)
    full = full.permute(0, 3, 1, 2).reshape(full.shape[0], 4, -1)
    L = sum(combo[p] * full[4 + k] for k, p in enumerate(pairs))
    ref = torch.cat([full[:4], L[None]], 0)
    (ref * cot.double()).sum().backward()
    jets, dlat, grads = res["fp32x3"]
    for s in range(5):
        assert _relerr(jets[s], ref[s].detach()) < 2e-5, "stream %d" % s
    err = _normerr if act == "leakyrelu" else _relerr
    assert err(dlat, lat64.grad) < 2e-4
    for k in range(6):
        assert err(net.fc[k].weight.grad, p64[k][0].grad) < 2e-4, "dW%d" % k
        assert err(net.fc[k].bias.grad, p64[k][1].grad) < 2e-4, "db%d" % k
    # and it is as close to the fp64 oracle as the exact-fp32 MFMA path is (within a factor 3 of its error)
    j32 = res["fp32"][0]
    e32 = max(_relerr(j32[s], ref[s].detach()) for s in range(5))
    e3 = max(_relerr(jets[s], ref[s].detach()) for s in range(5))
    assert e3 < 3 * e32 + 1e-6, (e3, e32)
    # forward-only value query in the same mode: value-tile kernels with split operands (S1 = 0, S2 = 3, last argument 3)
    from space_time_pde_amd import local_implicit_grid as lig
    prev = lig_jet.set_mlp_precision("fp32x3")
    try:
        with torch.no_grad(), _lib.dispatch_trace() as tr:
            y = lig.query_local_implicit_grid(net, lat.to(DEV), pts.to(DEV), 0., 1.)
            torch.cuda.synchronize()
    finally:
        lig_jet.set_mlp_precision(prev)
    assert tr.has("k_layer_coop", "1, false, 3>)", "S1 = 0, S2 = 3"), "\n".join(tr.kernels)
    yref = ref[0].detach().reshape(4, 2, 150).permute(1, 2, 0)
    assert _relerr(y, yref) < 2e-5
