"""Host-side logic that needs no GPU: PDELayer API/error behaviour, the jet compiler, physics strings, the C-ABI
library's symbols, dispatch rules, and the flat-name import style of the reference."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from space_time_pde_amd import implicit_net, local_implicit_grid as lig, nonlinearities, pde, physics
from space_time_pde_amd import regular_nd_grid_interpolation as rgi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol(hiplib):
    """Every function declared in include/stpde_hip.h is exported by libstpde_hip.so and bound in _lib.py."""
    from space_time_pde_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "stpde_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(stpde_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(hiplib, name), name
    assert sorted(n for n in _lib.exported_symbols() if not n.startswith("stpde_conv") and not n.startswith("stpde_unet")
                  ) == [d for d in declared if not d.startswith("stpde_conv") and not d.startswith("stpde_unet")] \
        or set(declared) <= set(_lib.exported_symbols())
    assert hiplib.stpde_version() >= 100
    buf = ctypes.create_string_buffer(64)
    assert hiplib.stpde_last_error(buf, 64) == 0


def test_abi_rejects_bad_arguments_without_gpu(hiplib):
    from space_time_pde_amd import _lib
    d = _lib.LayerDesc()
    d.ntiles, d.KT, d.MT = 0, 1, 1
    with pytest.raises(ValueError):
        _lib.check(hiplib.stpde_jet_layer_fwd(ctypes.byref(d), *([None] * 12)))
    g = _lib.GatherDesc()
    g.P = 3   # odd
    with pytest.raises(ValueError):
        _lib.check(hiplib.stpde_lig_gather(ctypes.byref(g), None, None, None, None, None, None, None, None))


def test_fused_residual_block_entry_points_check_arguments(hiplib):
    """Round 4: stpde_conv3d_fused / stpde_conv3d_wgrad_onload (the convolutions of a ResBlock3D with the BatchNorm work folded
    in) refuse inconsistent feature combinations before any launch."""
    from space_time_pde_amd import _lib
    a = _lib.Conv3dFusedArgs()
    with pytest.raises(ValueError):                                   # empty descriptor
        _lib.check(hiplib.stpde_conv3d_fused(ctypes.byref(a), None, None))
    a.d.B, a.d.T, a.d.Z, a.d.X, a.d.Ci, a.d.Co, a.d.ksize = 1, 2, 2, 2, 16, 16, 3
    a.x = a.w_pack = a.y = ctypes.c_void_p(256)
    a.y2, a.wo2_pack, a.Co2 = ctypes.c_void_p(256), ctypes.c_void_p(256), 16
    with pytest.raises(ValueError):                                   # a second output needs a 1x1x1 kernel
        _lib.check(hiplib.stpde_conv3d_fused(ctypes.byref(a), None, None))
    a.d.ksize = 1
    a.x2, a.w2_pack, a.Ci2 = ctypes.c_void_p(256), ctypes.c_void_p(256), 16
    with pytest.raises(ValueError):                                   # second input AND second output
        _lib.check(hiplib.stpde_conv3d_fused(ctypes.byref(a), None, None))
    d = _lib.Conv3dDesc()
    d.B, d.T, d.Z, d.X, d.Ci, d.Co, d.ksize = 1, 2, 2, 2, 16, 16, 3
    with pytest.raises(ValueError):                                   # on-load weight gradient: 1x1x1 only, statistics required
        _lib.check(hiplib.stpde_conv3d_wgrad_onload(ctypes.byref(d), None, None, None, None, None, None, None, None))
    b = _lib.BnDesc()
    assert [f[0] for f in _lib.BnDesc._fields_][-3:] == ["stats_mode", "reduce_done", "det"]


def test_one_call_per_direction_entry_points_exist_and_check_arguments(hiplib):
    """lig_imnet_jet_fwd / lig_imnet_jet_bwd are single entry points of the C ABI."""
    from space_time_pde_amd import _lib
    for name in ("stpde_lig_imnet_jet_fwd", "stpde_lig_imnet_jet_bwd", "stpde_lig_cell_sort", "stpde_lig_sort_tmp_bytes"):
        assert name in _lib.exported_symbols() and hasattr(hiplib, name)
    plan, ws, gd, cfg = _lib.ImNetPlanDesc(), _lib.LigWorkspace(), _lib.GatherDesc(), _lib.JetCfg()
    gd.P = 3    # odd
    with pytest.raises(ValueError):
        _lib.check(hiplib.stpde_lig_imnet_jet_fwd(ctypes.byref(plan), ctypes.byref(cfg), ctypes.byref(cfg), ctypes.byref(gd),
                                                  None, None, ctypes.byref(ws), None, 0, 0, None))
    with pytest.raises(ValueError):
        _lib.check(hiplib.stpde_lig_imnet_jet_bwd(ctypes.byref(plan), ctypes.byref(cfg), ctypes.byref(cfg), ctypes.byref(cfg),
                                                  ctypes.byref(gd), ctypes.byref(ws), None, 0, None, None, None, 0, None))
    with pytest.raises(ValueError):
        _lib.check(hiplib.stpde_lig_cell_sort(0, 10, None, None, None, None, 0, None))


def test_pde_layer_kat_and_api():
    """src/pde_test.py:12-53 through the product PDELayer (generic strategy on CPU tensors)."""
    layer = pde.PDELayer(in_vars="x, y, t", out_vars="u, v")
    for n in ("u", "v"):
        layer.add_equation("dif(%s, t) - (dif(dif(%s, x), x) + dif(dif(%s, y), y))" % (n, n, n), "diffusion_" + n)
    assert layer.eqn_num == 2 and layer.eqn_names == ["diffusion_u", "diffusion_v"]
    assert layer.n_in == 3 and layer.n_out == 2 and len(layer.all_vars) == 5
    with pytest.raises(RuntimeError):
        layer.eval(torch.zeros(1, 3))

    def fwd(i):
        u = i[..., 0:1] ** 2 + 3 * i[..., 1:2] ** 2 * i[..., 2:3] + i[..., 0:1] * i[..., 2:3]
        return torch.cat([u, u], -1)

    layer.update_forward_method(fwd)
    val, res = layer(torch.tensor([[1., 2., 3.]]))
    np.testing.assert_allclose(val.detach().numpy(), [[40., 40.]], atol=1e-4)
    for k in layer.eqn_names:
        np.testing.assert_allclose(res[k].detach().numpy(), [[-7.]])
    assert layer(torch.tensor([[1., 2., 3.]]), return_residue=False).shape == (1, 2)
    with pytest.raises(ValueError):
        layer.eval(torch.zeros(1, 4))                       # wrong trailing dim
    with pytest.raises(ValueError):
        layer.add_equation("dif(q, x)", "bad")             # unknown symbol
    layer.add_equation("u - v")                             # default name (reference quirk a-Q5 fixed)
    assert layer.eqn_names[-1] == "eqn_2"


def test_jet_compiler_atoms_and_fallback():
    layer = pde.PDELayer("t, x, z", "p, b, u, w")
    layer.add_equation("u*dif(b,x)", "adv")
    assert layer.eqns_jet["adv"].atoms == [(1, (1,)), (2, ())]
    layer.add_equation("dif(dif(dif(b,x),x),x)", "third")      # order 3: not expressible -> generic strategy
    assert layer.eqns_jet["third"] is None
    assert layer._jet_request(torch.zeros(1, 4, 3)) is None     # CPU tensor / unsupported equation


def test_rb2_layer_matches_reference_structure():
    layer = physics.get_rb2_pde_layer(mean=(0.01, 0, 0.02, -0.01), std=(0.05, 0.3, 0.15, 0.12), t_crop=2., z_crop=1.,
                                      x_crop=1., use_continuity=True)
    assert layer.eqn_names == ["transport_eqn_b", "transport_eqn_u", "transport_eqn_w", "continuity"]
    need = {mi for prog in layer.eqns_jet.values() for _, mi in prog.atoms}
    assert need == {(), (0,), (1,), (2,), (1, 1), (2, 2)}      # 6 jets per channel
    with pytest.raises(ValueError):
        physics.get_rb2_pde_layer(mean=(0, 0, 0, 0), std=None)
    with pytest.raises(ValueError):
        physics.get_rb2_pde_layer(mean=(0, 0, 0), std=(1, 1, 1))


def test_imnet_state_dict_keys_and_shapes():
    """src/implicit_net_test.py:15-26 (shape) + the duplicated state_dict keys of the reference (quirk a-Q7)."""
    net = implicit_net.ImNet(dim=4, in_features=32, out_features=3, nf=16)
    assert net(torch.rand(64, 36)).shape == (64, 3)
    keys = set(net.state_dict().keys())
    for k in range(6):
        assert {"fc%d.weight" % k, "fc%d.bias" % k, "fc.%d.weight" % k, "fc.%d.bias" % k} <= keys
    sw = implicit_net.ImNet(activation=nonlinearities.NONLINEARITIES["swish"])
    assert "activ.beta" in sw.state_dict()
    assert sum(p.numel() for p in implicit_net.ImNet().parameters()) == 209924


@pytest.mark.parametrize("dim", [3, 4])
def test_lig_generic_path_shapes_cpu(dim):
    """src/local_implicit_grid_test.py:16-30: 3-d and 4-d coordinates, batch 8, 512 points, 16^d... (smaller grid)."""
    grid = torch.rand(2, *([6] * dim), 32)
    pts = torch.rand(2, 64, dim)
    net = implicit_net.ImNet(dim=dim, in_features=32, out_features=3, nf=16)
    assert lig.query_local_implicit_grid(net, grid, pts, 0., 1.).shape == (2, 64, 3)


def test_lig_plus_pde_integration_cpu():
    """src/local_implicit_grid_integration_test.py:14-103 (shape check) on the generic strategy."""
    grid = torch.rand(2, 6, 6, 6, 32)
    pts = torch.rand(2, 32, 3)
    net = implicit_net.ImNet(dim=3, in_features=32, out_features=4, nf=16)
    layer = pde.PDELayer("t, x, z", "p, b, u, w")
    layer.add_equation("u*dif(b,x)", "transport_eqn_b")
    layer.update_forward_method(lambda q: lig.query_local_implicit_grid(net, grid, q, 0., 1.))
    val, res = layer(pts)
    assert val.shape == (2, 32, 4) and res["transport_eqn_b"].shape == (2, 32, 1)


def test_interpolation_generic_path_matches_golden(golden_dir):
    d = np.load(os.path.join(golden_dir, "g1_interp.npz"))
    for dim in (1, 2, 3, 4):
        grid, pts = torch.from_numpy(d["d%d_grid" % dim]), torch.from_numpy(d["d%d_pts" % dim])
        xmax = tuple(float(v) for v in d["d%d_xmax" % dim])
        out = rgi.regular_nd_grid_interpolation(grid, pts, tuple(0. for _ in range(dim)), xmax)
        np.testing.assert_allclose(out.numpy(), d["d%d_out" % dim], rtol=1e-6, atol=1e-6)


def test_box_constants_follow_reference_fp32_sequence():
    from space_time_pde_amd.lig_jet import box_constants
    lo, hi, cube = box_constants((32, 128, 128), 0., 1.)
    size = torch.tensor([32., 128., 128.])
    eps = 1e-6 * (torch.ones(3) - torch.zeros(3))
    assert hi == (torch.ones(3) - eps).tolist() and lo == eps.tolist()
    assert cube == (torch.ones(3) / (size - 1)).tolist()
    with pytest.raises(ValueError):
        box_constants((4, 4, 4), 0.5, 1.)          # xmin != 0 is refused (quirk a-Q1)


def test_flat_import_style_of_the_reference():
    code = ("import sys; sys.path.append(%r); import pde, local_implicit_grid, implicit_net, physics, nonlinearities;"
            "import regular_nd_grid_interpolation as rgi; import space_time_pde_amd.pde as p2;"
            "assert pde is p2 and hasattr(rgi, 'clip_tensor') and hasattr(physics, 'get_rb2_pde_layer'); print('ok')"
            % os.path.join(ROOT, "space_time_pde_amd", "flat"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp")
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr


def test_hot_path_never_falls_back_on_cuda_without_library(monkeypatch):
    """Eligibility is decided by tensor structure, never by library availability: a missing .so must raise."""
    from space_time_pde_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libstpde_hip.so")
    with pytest.raises(RuntimeError, match="not built"):
        _lib.lib()


def test_tensor_bounds_cache_survives_repeated_lookups():
    """train.py:48-49 hands the SAME xmin / xmax tensors to every step: repeated lookups must hit the cache (r1 bug: a
    WeakKeyDictionary compared the tensor keys elementwise and raised on the second call)."""
    from space_time_pde_amd import lig_jet
    a, b = torch.zeros(3), torch.ones(3)
    first = lig_jet.cached_box_constants((4, 8, 8), a, b)
    for _ in range(3):
        assert lig_jet.cached_box_constants((4, 8, 8), a, b) is first
    b.mul_(2.0)                                   # in-place change of the bounds: version counter invalidates the entry
    assert lig_jet.cached_box_constants((4, 8, 8), a, b)[2][0] == pytest.approx(2.0 / 3.0)
    with pytest.raises(ValueError):
        lig_jet.cached_box_constants((4, 8, 8), torch.full((3,), 0.5), b)


def test_fused_clip_adam_survives_deepcopy_and_pickle():
    """Optimizer.__getstate__ serialises defaults / state / param_groups only, so a copied FusedClipAdam came
    back without ``_use_flat`` and step() raised AttributeError.  (Host logic only: no launch on this machine.)"""
    import copy
    import pickle
    import torch
    from space_time_pde_amd.optim import FusedClipAdam
    p = torch.nn.Parameter(torch.zeros(8))
    for flat in (True, False):
        opt = FusedClipAdam([p], lr=1e-3, clip_grad=0.5, flat=flat)
        for clone in (copy.deepcopy(opt), pickle.loads(pickle.dumps(opt))):
            assert clone._use_flat == flat and clone._flat == {}
            assert clone.param_groups[0]["clip_grad"] == 0.5
            clone.step()          # no gradients anywhere: nothing to launch, but every attribute step() reads must exist


def _plan_meta(packed=0):
    """A jet-call descriptor as lig_jets() builds it (reference width, RB2 stream set), without any device."""
    from space_time_pde_amd import lig_jet
    meta = lig_jet._Meta()
    meta.plan = lig_jet.ImNetPlan.get(3, 32, 4, 32)
    meta.bf16, meta.nsplit, meta.packs16, meta.packed_mask = bool(packed), 1, None, packed
    meta.cfg_out, meta.S_out, _ = lig_jet.make_cfg("softplus", 0.0, True, [], {(1, 1): 1.0, (2, 2): 1.0})
    meta.cfg, meta.S = meta.cfg_out, meta.S_out
    meta.chunk, meta.tail, meta.budget = 1 << 20, 0, None
    return meta


def test_memory_plan_counts_the_two_phase_scratch_when_the_sharded_step_will_use_it(monkeypatch):
    """LigJetFunction.forward makes its memory plan BEFORE train_step installs ``sync_hooks`` (they are set
    around loss.backward() only), so the dgrad-first scratch of the last chunk was never budgeted.  The step now raises
    ``lig_jet.expect_two_phase`` around its forward; the plan must include the term under either flag, and the step must
    raise the flag exactly when its backward will install the hooks."""
    from space_time_pde_amd import lig_jet, train_step
    meta = _plan_meta()
    P = 1 << 20
    two = lig_jet._two_phase_bytes(meta)
    assert two > 50_000                                  # ~57 KB per point in exact fp32 (two more adjoint buffers)
    base = lig_jet._stash_bytes(meta, P)
    monkeypatch.setattr(lig_jet, "expect_two_phase", True)
    assert lig_jet._stash_bytes(meta, P) == base + P * two
    monkeypatch.setattr(lig_jet, "expect_two_phase", False)
    monkeypatch.setattr(lig_jet, "sync_hooks", {"dw": None})
    assert lig_jet._stash_bytes(meta, P) == base + P * two
    monkeypatch.setattr(lig_jet, "sync_hooks", None)
    # the recompute chunk shrinks by the same term (budget of 64 GiB, no device query)
    monkeypatch.setattr(lig_jet, "_free_bytes", lambda device: 1 << 60)
    shrunk = 0
    for mib in range(1024, 2049, 64):     # (chunks are powers of two: the term shows at some budgets, never grows the chunk)
        meta.budget = mib << 20
        monkeypatch.setattr(lig_jet, "expect_two_phase", False)
        c0 = lig_jet._recompute_chunk(meta, None)
        monkeypatch.setattr(lig_jet, "expect_two_phase", True)
        c1 = lig_jet._recompute_chunk(meta, None)
        assert c1 <= c0
        shrunk += c1 < c0
    assert shrunk >= 1
    monkeypatch.setattr(lig_jet, "expect_two_phase", False)

    # the step: flag up during the forward iff the backward will run with hooks (distributed + STPDE_OVERLAP_SYNC != 0)
    seen = {}

    class _Layer:
        def update_forward_method(self, f):
            pass

        def __call__(self, pts, return_residue=True):
            seen["flag"] = lig_jet.expect_two_phase
            y = lin(pts.sum(-1, keepdim=True))
            return y.expand(-1, -1, 4), {"e": y}

    class _Unet(torch.nn.Module):
        def forward(self, x):
            return x

    lin = torch.nn.Linear(1, 1)
    pts = torch.rand(1, 8, 3)
    monkeypatch.setattr(train_step, "_SumGradAcrossRanks", type("_Id", (), {"apply": staticmethod(lambda t: t)}))
    monkeypatch.setattr(train_step.dist, "all_reduce", lambda t, async_op=False: None)
    for distributed, env, want in ((False, "1", False), (True, "1", True), (True, "0", False)):
        monkeypatch.setenv("STPDE_OVERLAP_SYNC", env)
        train_step._sharded_step(_Unet(), lin, _Layer(), torch.zeros(1, 4, 2, 2, 2), pts, torch.zeros(1, 8, 4), 8, 1.0, 1.0,
                                 "l1", 0.0, 1.0, distributed, False)
        assert seen["flag"] is want, (distributed, env)
        assert lig_jet.expect_two_phase is False and lig_jet.sync_hooks is None


def test_no_unprotected_dpp_sequences_in_the_built_library():
    """hand-written DPP adds inside asm statements are invisible to the compiler's hazard recognizer.  The
    disassembly of every gfx950 code object of the build must hold no DPP instruction that reads a VGPR a VALU instruction
    wrote fewer than two wait states earlier (tools/check_dpp_hazard.py; 76 such sites in the round-5 library)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_dpp_hazard
    objs = [o for o in os.listdir(check_dpp_hazard.BUILD) if o.endswith(".hip.o")] if os.path.isdir(check_dpp_hazard.BUILD) else []
    if not objs:
        pytest.skip("no object files next to the library (a tree that received only the built .so)")
    # the scanner itself: a synthetic hazard is found, the protected forms are not
    bad = check_dpp_hazard.scan_disassembly(
        "0000 <k>:\n\tv_mov_b32_e32 v1, v2\n\tv_add_f32_dpp v1, v1, v1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
    assert len(bad) == 1
    ok = check_dpp_hazard.scan_disassembly(
        "0000 <k>:\n\tv_mov_b32_e32 v1, v2\n\ts_nop 1\n\tv_add_f32_dpp v1, v1, v1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
    assert not ok
    bad, ndpp, nobj = check_dpp_hazard.scan_build()
    assert nobj >= 20 and ndpp > 1000
    assert not bad, bad[:5]


def test_hazard_scanner_flags_a_store_whose_data_is_overwritten_at_once():
    """Round 6, found on hardware: buffer_store_dwordx4 with a scalar offset + a VALU write of its first data register in the
    next instruction stored the new value now and then (k_fc1_bwd_fused with two waves per SIMD).  The build scan flags it."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_dpp_hazard as chk
    bad = chk.scan_store_data("0000 <k>:\n\tbuffer_store_dwordx4 v[10:13], v238, s[8:11], s92 offen\n\tv_mov_b32_e32 v10, v204\n")
    assert len(bad) == 1 and bad[0][3] == 0
    ok = chk.scan_store_data("0000 <k>:\n\tbuffer_store_dwordx4 v[10:13], v238, s[8:11], s92 offen\n\ts_nop 2\n\tv_mov_b32_e32 v10, v204\n")
    assert not ok
    ok = chk.scan_store_data("0000 <k>:\n\tbuffer_store_dwordx4 v[10:13], v238, s[8:11], s92 offen\n\tv_mov_b32_e32 v20, v204\n"
                             "\tv_mov_b32_e32 v21, v204\n\tv_mov_b32_e32 v10, v204\n")
    assert not ok


def test_early_loads_are_not_waited_for_in_front_of_the_matrix_phase():
    """Round 6, found in the listings: three kernels requested the next row tile's operands ahead of their matrix phase and the
    compiler's (conservative, correct) counter waits made them wait for those loads at once -- k_wgrad_coop `s_waitcnt vmcnt(1)`
    behind its ten early loads (a prologue-loaded register used at the loop head, the early loads in run-time branches),
    k_fc1_bwd_fused / k_fc2_fwd_bf `vmcnt(0)` in front of the first LDS read after an LDS-DMA request the compiler could not prove
    disjoint.  The scan of tools/micro/isa_near_waits.py over those kernels' loops must stay clean: a compiler that
    decides differently would bring the stall back without failing any numerical test."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.path.insert(0, os.path.join(ROOT, "tools", "micro"))
    import check_dpp_hazard
    import isa_near_waits as nw
    # the scanner itself: ten loads, then "all but one", then a matrix phase, inside a loop
    body = [(4 * i, "global_load_dwordx4 v[%d:%d], v[0:1], off" % (8 + 4 * i, 11 + 4 * i)) for i in range(10)]
    body += [(40, "s_waitcnt vmcnt(1)")] + [(44 + 4 * i, "v_mfma_f32_16x16x4_f32 v[0:3], v4, v5, v[0:3]") for i in range(12)]
    body += [(92, "s_cbranch_scc1 65512")]          # back to address 0
    assert len(nw.scan(body)) == 1
    body[10] = (40, "s_waitcnt vmcnt(10)")
    assert not nw.scan(body)
    want = {"jet_wgrad_s31.hip.o": ["_Z12k_wgrad_coopILi3ELi1ELi1ELi2ELi8ELb0ELb0ELi1ELi0E", "_Z12k_wgrad_coopILi3ELi1ELi1ELi2ELi8ELb0ELb1ELi1E"],
            "jet_fc1_bwd.hip.o": ["_Z15k_fc1_bwd_fusedILi1ELi2ELb0E", "_Z15k_fc1_bwd_fusedILi1ELi2ELb1E"],
            "jet_layer_s31.hip.o": ["_Z12k_fc2_fwd_bfILi3ELi1ELi2E", "_Z14k_fc1_fwd_specILi3ELi1ELi2ELi4E"]}
    if not all(os.path.exists(os.path.join(check_dpp_hazard.BUILD, o)) for o in want):
        pytest.skip("no object files next to the library (a tree that received only the built .so)")
    import tempfile
    with tempfile.TemporaryDirectory() as scratch:
        for obj, needles in want.items():
            text = check_dpp_hazard.disassemble(os.path.join(check_dpp_hazard.BUILD, obj), scratch)
            seen = set()
            for name, kbody in nw.kernels(text):
                for nd in needles:
                    if name.startswith(nd):
                        seen.add(nd)
                        # "all but at most two" right behind five or more requests, a matrix phase behind it
                        hits = [h for h in nw.scan(kbody) if h[1] <= 2 and h[2] >= 5 and h[3] >= 20]
                        assert not hits, (name, [(hex(a), n, l, m) for a, n, l, m in hits])
                        if "k_fc1_fwd_spec" in nd:
                            # (its producers' prefetch of the next tile -- five requests -- sat in front of the layer-0 MFMAs of
                            # the step with `vmcnt(4)` behind it)
                            hits = [h for h in nw.scan(kbody) if h[2] >= 5 and h[1] < 5 and h[3] >= 8]
                            assert not hits, (name, [(hex(a), n, l, m) for a, n, l, m in hits])
                        if "k_fc2_fwd_bf" in nd:
                            # (its guarded read sat behind the previous tile's output stores and in front of the activation
                            # jets, not of MFMAs: no full drain of the counter within 30 instructions behind a store)
                            ins = [i for _, i in kbody]
                            drains = [k for k, i in enumerate(ins) if re.match(r"s_waitcnt.*vmcnt\(0\)", i)
                                      and any(j.startswith("global_store") for j in ins[max(0, k - 30):k])]
                            assert not drains, (name, drains)
            assert seen == set(needles), (obj, sorted(set(needles) - seen))


def test_tune_overrides_and_no_environment_in_the_library(hiplib):
    """Round 6: launch-geometry overrides for tests go through stpde_tune (process-wide, restored by the context manager); the
    library itself reads no environment variable -- no getenv call in csrc/ -- and the Python package is
    down to the ten documented switches."""
    from space_time_pde_amd import _lib
    assert _lib.tune("conv3_lds_gx", 7) == 0
    with _lib.tuned(conv3_lds_gx=24, conv3_lds_minblk=1):
        assert hiplib.stpde_tune(b"conv3_lds_gx", 24) == 24 and hiplib.stpde_tune(b"conv3_lds_minblk", 1) == 1
    assert _lib.tune("conv3_lds_gx", 0) == 7 and _lib.tune("conv3_lds_minblk", 0) == 0
    with pytest.raises(KeyError):
        _lib.tune("no_such_key", 1)
    csrc = os.path.join(ROOT, "space_time_pde_amd", "csrc")          # (the sort primitives of rocPRIM import getenv themselves)
    calls = [(fn, n + 1) for fn in os.listdir(csrc) if fn.endswith((".hip", ".h", ".cpp"))
             for n, ln in enumerate(open(os.path.join(csrc, fn))) if re.search(r"\bgetenv\s*\(", ln)]
    assert not calls, calls
    pkg = os.path.join(ROOT, "space_time_pde_amd")
    found = set()
    for fn in os.listdir(pkg) + ["../bench.py"]:
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            for m in re.finditer(r"environ[^\n]*?[\"'](STPDE_[A-Z0-9_]+)[\"']", src):
                found.add(m.group(1))
    assert found == {"STPDE_LIB", "STPDE_MLP_PRECISION", "STPDE_MEM_BUDGET_GB", "STPDE_DETERMINISTIC", "STPDE_S34", "STPDE_FC1_FUSED",
                     "STPDE_FUSED_RESBLOCK", "STPDE_UNET_DEFERRED", "STPDE_OVERLAP_SYNC", "STPDE_BENCH_ONE_DEVICE",
                     "STPDE_BENCH_BACKEND"}, sorted(found)
