"""Host logic: the index maps that turn nn.Linear parameters into MFMA operand packs (and back) are validated
against a numpy emulation of v_mfma_f32_16x16x4_f32 + the kernels' loop structure (no GPU needed)."""
import numpy as np
import torch

from space_time_pde_amd.lig_jet import ImNetPlan, XT
from tests import mfma_emu as E


def _plan_and_params(nf=16, cout=4, seed=0):
    plan = ImNetPlan.get(3, 32, cout, nf)
    g = torch.Generator().manual_seed(seed)
    params = []
    for lay in plan.layers:
        params += [torch.randn(lay["M"], lay["Kin"], generator=g, dtype=torch.float64).float(),
                   torch.randn(lay["M"], generator=g, dtype=torch.float64).float()]
    return plan, params


def _x_aug(rows, cin, rng):
    """Augmented input rows in SLOT order (what k_gather writes): tiles 0 / 1 in feature order, tile 2 sparse."""
    slot = ImNetPlan.get(3, cin, 4, 16).slot
    x = np.zeros((16, 16 * XT))
    x[:, slot[:3 + cin]] = rows
    x[:, slot[3 + cin]] = 1.0
    return x


def test_forward_packs_reproduce_linear_layers():
    plan, params = _plan_and_params()
    packs = plan.pack(params).double().numpy()
    rng = np.random.default_rng(0)
    for l in range(1, 6):
        lay = plan.layers[l]
        W, b = params[2 * l].double().numpy(), params[2 * l + 1].double().numpy()
        KT, MT = lay["KT"], lay["MT"]
        h = rng.standard_normal((16, lay["Kh"]))
        xr = rng.standard_normal((16, 35))
        wh = plan.pack_view(packs, l, "Wh").reshape(KT, MT, 64, 4)
        ws = plan.pack_view(packs, l, "Ws").reshape(XT, MT, 64, 4)
        assert np.all(ws[XT - 1, :, :, 1:] == 0) and np.any(ws[XT - 1, :, :, 0] != 0)   # sparse third tile: register 0 only
        out = E.gemm_frag(wh, E.to_frag(h), KT, MT) + E.gemm_frag(ws, E.to_frag(_x_aug(xr, 32, rng)), XT, MT)
        got = E.from_frag(out)[:, :lay["M"]]
        inp = np.concatenate([h, xr], 1) if lay["skip"] else h
        want = inp @ W.T + b
        np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-10)
        # tangent constants: tanc[d] in D layout == W_s[:, d]
        tc = plan.pack_view(packs, l, "tanc").reshape(3, MT, 64, 4)
        for d in range(3):
            col = E.from_frag(tc[d])[0, :lay["M"]]
            want_col = W[:, lay["Kh"] + d] if lay["skip"] else np.zeros(lay["M"])
            np.testing.assert_allclose(col, want_col)


def test_layer0_pack_and_transposed_packs():
    plan, params = _plan_and_params(seed=1)
    packs = plan.pack(params).double().numpy()
    rng = np.random.default_rng(1)
    lay = plan.layers[0]
    W, b = params[0].double().numpy(), params[1].double().numpy()
    xr = rng.standard_normal((16, 35))
    ws = plan.pack_view(packs, 0, "Ws").reshape(XT, lay["MT"], 64, 4)
    got = E.from_frag(E.gemm_frag(ws, E.to_frag(_x_aug(xr, 32, rng)), XT, lay["MT"]))
    np.testing.assert_allclose(got, xr @ W.T + b, rtol=1e-10, atol=1e-10)
    # dgrad pack: hbar^T = W_h^T abar^T
    for l in range(1, 6):
        lay = plan.layers[l]
        W = params[2 * l].double().numpy()
        KT, MT = lay["KT"], lay["MT"]
        ab = np.zeros((16, 16 * MT))
        ab[:, :lay["M"]] = rng.standard_normal((16, lay["M"]))
        wt = plan.pack_view(packs, l, "WhT").reshape(MT, KT, 64, 4)
        got = E.from_frag(E.gemm_frag(wt, E.to_frag(ab), MT, KT))
        np.testing.assert_allclose(got, ab[:, :lay["M"]] @ W[:, :lay["Kh"]], rtol=1e-10, atol=1e-10)
    # xbar pack: xbar^T = W_s^T abar^T (layers 0..4)
    for l in range(5):
        lay = plan.layers[l]
        W = params[2 * l].double().numpy()
        MT = lay["MT"]
        ab = rng.standard_normal((16, lay["M"]))
        wt = plan.pack_view(packs, l, "WsL").reshape(MT, plan.xl, 64, 4)      # latent channels only
        got = E.from_frag(E.gemm_frag(wt, E.to_frag(ab), MT, plan.xl))[:, :32]
        np.testing.assert_allclose(got, ab @ W[:, lay["Kh"] + 3:lay["Kh"] + 35], rtol=1e-10, atol=1e-10)


def test_wgrad_transpose_and_unpack():
    """dW = P^T Q via the k_wgrad operand transposition, then the unpack map back to (weight, bias) gradients."""
    plan, params = _plan_and_params(seed=2)
    rng = np.random.default_rng(2)
    dw_flat = np.zeros(plan.n_dw)
    want = []
    for l in range(6):
        lay = plan.layers[l]
        KT, MT = lay["KT"], lay["MT"]
        P = np.zeros((16, 16 * MT))
        P[:, :lay["M"]] = rng.standard_normal((16, lay["M"]))
        h = rng.standard_normal((16, lay["Kh"]))
        xr = rng.standard_normal((16, 35))
        Q = np.concatenate([h, _x_aug(xr, 32, rng)], 1)          # [16, 16*(KT+XT)]
        pf, qf = E.to_frag(P), E.to_frag(Q)
        off, mp, ka = plan.dw_off[l]
        dW = np.zeros((mp, ka))
        for mt in range(MT):
            pa = E.transpose_block(pf[mt])
            for kq in range(KT + XT):
                qb = E.transpose_block(qf[kq])
                acc = np.zeros((64, 4))
                for s in range(4):
                    acc = E.mfma4(pa[:, s], qb[:, s], acc)
                for r in range(4):
                    dW[16 * mt + 4 * E.G + r, 16 * kq + E.J] += acc[:, r]
        np.testing.assert_allclose(dW, P.T @ Q, rtol=1e-10, atol=1e-10)
        dw_flat[off:off + mp * ka] = dW.reshape(-1)
        inp = np.concatenate([h, xr], 1) if lay["skip"] else h
        want += [P[:, :lay["M"]].T @ inp, P[:, :lay["M"]].sum(0)]
    grads = plan.unpack_grads(torch.from_numpy(dw_flat), params)
    for g, w in zip(grads, want):
        np.testing.assert_allclose(g.numpy(), w, rtol=1e-10, atol=1e-10)


def test_plan_rejects_unsupported_architectures():
    import pytest
    with pytest.raises(ValueError):
        ImNetPlan(3, 32, 4, 4)        # nf not a multiple of 16
    with pytest.raises(ValueError):
        ImNetPlan(4, 32, 4, 16)       # dim != 3
    with pytest.raises(ValueError):
        ImNetPlan(3, 64, 4, 16)       # latent too wide for the augmented input
    with pytest.raises(ValueError):
        ImNetPlan(3, 33, 4, 16)       # the sparse third tile holds features 32..35 only
