"""adversarial test of the "fp32x3" product -- every fp32 operand split into three bf16 terms, six partial
products on the bf16 MFMA, fp32 accumulation (csrc/jet_layer_impl.h: k_layer_coop<..., SPL = 3>) -- against the exact-fp32
MFMA kernel of the same layer and an fp64 reference, one dot product at a time.

The layer kernel is driven directly through the C ABI (stpde_jet_layer_fwd, value stream only) with ReLU on strictly positive
inputs, i.e. the activation is the identity and out = W h exactly what one hidden-to-hidden product of IM-NET computes
(K = 256 -> M = 128: the second hidden layer of the reference width, src/implicit_net.py:31-36).
"""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
KT, MT, XT = 16, 8, 3
K, M = 16 * KT, 16 * MT


def _to_blocks_rows(H):
    """H [rows, K] -> column-major fragment image [tile][1][KT][64][4]: lane 16g+j = features 4g..4g+3 of row j."""
    T = H.shape[0] // 16
    x = H.reshape(T, 16, KT, 4, 4)                  # [tile, j, kt, g, r]
    return x.permute(0, 2, 3, 1, 4).reshape(T, 1, KT, 64, 4).contiguous()


def _from_blocks_rows(Y, T):
    """[tile][1][MT][64][4] -> [rows, M]"""
    y = Y.reshape(T, MT, 4, 16, 4)                  # [tile, mt, g, j, r]
    return y.permute(0, 3, 1, 2, 4).reshape(T * 16, M)


def _pack_w(W):
    """W [M, K] -> A-operand pack [KT][MT][64][4] = W[16mt + j, 16kt + 4g + r] (lig_jet.ImNetPlan, "Wh")."""
    x = W.reshape(MT, 16, KT, 4, 4)                 # [mt, j, kt, g, r]
    return x.permute(2, 0, 3, 1, 4).reshape(KT, MT, 64, 4).contiguous()


def _pack_w_split3(pack):
    """bf16 A-operand pack of the split mode: block (q, mt) = fp32 blocks (2q, mt), (2q+1, mt) lane by lane, three exact bf16
    terms stacked [3][KT/2][MT][64][8] (lig_jet.ImNetPlan.pack_bf16(nsplit=3))."""
    w = pack.view(KT // 2, 2, MT, 64, 4).permute(0, 2, 3, 1, 4)
    terms, r = [], w
    for t in range(3):
        h = r.to(torch.bfloat16)
        terms.append(h)
        r = r - h.float()
    return torch.stack(terms, 0).contiguous()


def _run_layer(hiplib, H, W, split):
    from space_time_pde_amd import _lib
    T = H.shape[0] // 16
    prev = _to_blocks_rows(H).to(DEV)
    pack = _pack_w(W).to(DEV)
    w16 = _pack_w_split3(pack) if split else None
    X = torch.zeros(T * XT * 256, device=DEV)
    Ws = torch.zeros(XT * MT * 256, device=DEV)
    out = torch.empty(T * MT * 256, device=DEV)
    d = _lib.LayerDesc()
    d.ntiles, d.KT, d.MT, d.first_hidden = T, KT, MT, 0
    d.cfg.S1, d.cfg.S2, d.cfg.act, d.cfg.act_param = 0, 0, _lib.ACT_CODES["relu"], 0.0
    d.mfma_bf16, d.packed = (3 if split else 0), 0
    with _lib.dispatch_trace() as tr:
        _lib.check(hiplib.stpde_jet_layer_fwd(C.byref(d), _lib.ptr(prev), _lib.ptr(X), _lib.ptr(pack), _lib.ptr(Ws), None,
                                              None, None, _lib.ptr(out), None, _lib.ptr(w16), None, _lib.stream_ptr()))
        torch.cuda.synchronize()
    assert tr.has("k_layer_coop"), tr.kernels
    if split:      # the three-term kernel really ran (template argument SPL = 3)
        assert any("k_layer_coop" in k and ("true, 1, false, 3" in k or "SPL = 3" in k) for k in tr.kernels), tr.kernels
    return _from_blocks_rows(out.cpu(), T)


def _classes(rng, rows):
    """Adversarial operand classes; every H is strictly positive (ReLU = identity), W carries the signs."""
    out = {}
    # (a) operands spanning 2^-60 .. 2^60, random signs in W
    H = np.exp2(rng.uniform(-60, 60, (rows, K)))
    W = np.exp2(rng.uniform(-60, 60, (M, K))) * rng.choice([-1.0, 1.0], (M, K))
    out["range_2pm60"] = (H, W)
    # (b) no cancellation: all terms positive, O(1) magnitudes with full 24-bit mantissas
    out["positive"] = (rng.uniform(0.5, 2.0, (rows, K)), rng.uniform(0.5, 2.0, (M, K)))
    # (c) cancelling rows: w_{2i+1} = -w_{2i} (1 + 2^-12 eps), h_{2i+1} = h_{2i} (1 + 2^-12 eps): the sum is ~2^-11 of sum|terms|
    Hc = rng.uniform(0.5, 2.0, (rows, K))
    Wc = rng.uniform(0.5, 2.0, (M, K))
    Hc[:, 1::2] = Hc[:, 0::2] * (1 + 2.0 ** -12 * rng.standard_normal((rows, K // 2)))
    Wc[:, 1::2] = -Wc[:, 0::2] * (1 + 2.0 ** -12 * rng.standard_normal((M, K // 2)))
    out["cancelling"] = (Hc, Wc)
    # (d) operands whose third (lo) bf16 term is subnormal: h ~ 2^-112 .. 2^-104 -> lo ~ 2^-129 .. 2^-120
    out["subnormal_lo"] = (np.exp2(rng.uniform(-112, -104, (rows, K))), rng.uniform(0.5, 2.0, (M, K)) * rng.choice([-1.0, 1.0], (M, K)))
    # (e) one huge term next to many small ones (absorption)
    He = rng.uniform(0.5, 2.0, (rows, K))
    He[:, 7] *= 2.0 ** 20
    out["absorption"] = (He, rng.uniform(0.5, 2.0, (M, K)) * rng.choice([-1.0, 1.0], (M, K)))
    return out


def test_fp32x3_product_error_vs_exact_fp32_mfma_and_fp64(hiplib):
    rng = np.random.default_rng(12)
    rows = 64 * 16
    report = {}
    for name, (H, W) in _classes(rng, rows).items():
        H32, W32 = torch.from_numpy(H).float(), torch.from_numpy(W).float()
        ref = H32.double() @ W32.double().t()                         # fp64 value of the fp32 operands
        mag = H32.double().abs() @ W32.double().abs().t()              # sum of |terms|
        y32 = _run_layer(hiplib, H32, W32, split=False).double()
        yx3 = _run_layer(hiplib, H32, W32, split=True).double()
        assert torch.isfinite(y32).all() and torch.isfinite(yx3).all(), name
        ulp_mag = torch.tensor(np.spacing(mag.float().numpy()).astype(np.float64))     # one fp32 ulp at sum|terms|
        ulp_ref = torch.tensor(np.spacing(ref.float().abs().numpy()).astype(np.float64))
        e32, ex3 = (y32 - ref).abs(), (yx3 - ref).abs()
        report[name] = dict(fp32_mfma_max_ulp_of_mag=float((e32 / ulp_mag).max()), fp32x3_max_ulp_of_mag=float((ex3 / ulp_mag).max()),
                            fp32_mfma_rms_ulp_of_mag=float((e32 / ulp_mag).pow(2).mean().sqrt()),
                            fp32x3_rms_ulp_of_mag=float((ex3 / ulp_mag).pow(2).mean().sqrt()),
                            fp32_mfma_max_ulp_of_value=float((e32 / ulp_ref).max()), fp32x3_max_ulp_of_value=float((ex3 / ulp_ref).max()))
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "fp32x3_product_errors.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1))
    # What the numbers say (MI355X, round 4; profiles/r4_fp32x3_product_errors.json): a K = 256 fp32 accumulation chain is NOT
    # within 2 ulp of the fp64 value on either pipe -- the exact-fp32 MFMA itself carries up to 13 ulp (rms 3 ulp) on
    # all-positive terms -- so the bound asserted here is relative to that kernel, measured in the same run on the same
    # operands: the split product stays within 2x of the exact-fp32 MFMA's maximum error and within 1.5x of its rms error on
    # EVERY class (measured: below it on four of five classes; 1.7x / 1.36x on the 2^+-60 dynamic-range class, where a few
    # dominant terms arrive in six partial products), and within 16 ulp of sum|terms| overall.
    for name, r in report.items():
        assert r["fp32x3_max_ulp_of_mag"] <= max(2.0, 2.0 * r["fp32_mfma_max_ulp_of_mag"]), (name, r)
        assert r["fp32x3_rms_ulp_of_mag"] <= max(0.5, 1.5 * r["fp32_mfma_rms_ulp_of_mag"]), (name, r)
        assert r["fp32x3_max_ulp_of_mag"] <= 16.0, (name, r)
    # without cancellation sum|terms| == |value|: the split product is as close to the fp64 value as the exact-fp32 MFMA
    assert report["positive"]["fp32x3_max_ulp_of_value"] <= report["positive"]["fp32_mfma_max_ulp_of_value"] + 2.0, report["positive"]
    # operands whose third bf16 term is subnormal lose nothing against the exact-fp32 MFMA
    assert report["subnormal_lo"]["fp32x3_max_ulp_of_mag"] <= report["subnormal_lo"]["fp32_mfma_max_ulp_of_mag"] + 2.0
