"""stpde_jet_fc1_bwd (fused) against stpde_jet_wgrad + stpde_jet_layer_bwd on the same random buffers: abar0, tangent row sums,
fc1's block of dW.  GPU box:  python tools/micro/check_fc1_fused.py [ntiles]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    from space_time_pde_amd import _lib
    from space_time_pde_amd.lig_jet import ImNetPlan, make_cfg
    dev = torch.device("cuda:0")
    L = _lib.lib()
    plan = ImNetPlan.get(3, 32, 4, 32)
    nt = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    cfg, S, _ = make_cfg("softplus", 0.0, True, [], {(1, 1): 1.0, (2, 2): 0.25})
    torch.manual_seed(0)
    # (a REAL parameter pack: the tangent-constant blocks must hold the same values in every row, which random pack contents
    # do not -- the fused kernel keeps them as 16 floats per block)
    from space_time_pde_amd import implicit_net
    net = implicit_net.ImNet(dim=3, in_features=32, out_features=4, nf=32, activation=torch.nn.Softplus).to(dev)
    prm = []
    for k in range(6):
        prm += [net.fc[k].weight, net.fc[k].bias]
    packs = plan.pack(prm)
    p16 = plan.pack_bf16(packs, 1)
    lay = plan.layers[1]
    MT0 = plan.layers[0]["MT"]
    abar1 = (0.1 * torch.randn(nt * S * lay["MT"] * 256, device=dev)).to(torch.bfloat16)
    X = torch.randn(nt * 3 * 256, device=dev)
    XR = torch.randn(nt * 3 * 256, device=dev)
    z0 = torch.randn(nt * MT0 * 256, device=dev)
    cw = torch.rand(nt * 2 * 8, device=dev)
    if os.environ.get("CHECK_CW_CONST") == "1":
        cw = torch.full_like(cw, 0.5)
    pv = plan.pack_view
    p = _lib.ptr
    st = _lib.stream_ptr()
    d = _lib.LayerDesc()
    d.ntiles, d.KT, d.MT, d.first_hidden, d.cfg, d.mfma_bf16, d.packed = nt, lay["KT"], lay["MT"], 1, cfg, 1, 6
    dwg = _lib.LayerDesc()
    dwg.ntiles, dwg.KT, dwg.MT, dwg.first_hidden, dwg.cfg, dwg.mfma_bf16, dwg.packed = nt, lay["KT"], lay["MT"], 1, cfg, 1, 4
    out = {}
    for mode in ("old", "fused"):
        abar0 = torch.full((nt * MT0 * 128,), float("nan"), device=dev)
        tan0 = torch.full((nt * MT0 * 48,), float("nan"), device=dev)
        dw = torch.zeros(256 * 16 * 35, device=dev)
        if mode == "old":
            _lib.check(L.stpde_jet_wgrad(C.byref(dwg), S, p(abar1), p(z0), p(XR), p(pv(packs, 0, "tanc")), p(dw), p(cw), st))
            _lib.check(L.stpde_jet_layer_bwd(C.byref(d), p(abar1), p(pv(packs, 1, "WhT")), None, p(X), p(pv(packs, 0, "Ws")),
                                             p(pv(packs, 0, "tanc")), p(abar0), p(cw), None, p(p16[(1, "WhT")]), p(tan0), p(z0), st))
        else:
            assert L.stpde_jet_fc1_bwd_supported(C.byref(d))
            _lib.check(L.stpde_jet_fc1_bwd(C.byref(d), p(abar1), p(p16[(1, "WhT")]), p(z0), p(pv(packs, 0, "tanc")), p(cw), p(XR),
                                           p(abar0), p(tan0), p(dw), None, st))
        torch.cuda.synchronize()
        out[mode] = (abar0.view(torch.int32).clone(), tan0.clone(), dw.clone())
    # stress: the fused call again and again on garbage-filled outputs -- every run must reproduce the first one (abar0, tan0
    # bit for bit; dW to atomic-order rounding)
    reps = int(os.environ.get("CHECK_REPS", "0"))
    for rep in range(reps):
        abar0 = torch.full((nt * MT0 * 128,), float("nan"), device=dev)
        tan0 = torch.full((nt * MT0 * 48,), 1e6 + rep, device=dev)
        dw = torch.zeros(256 * 16 * 35, device=dev)
        _lib.check(L.stpde_jet_fc1_bwd(C.byref(d), p(abar1), p(p16[(1, "WhT")]), p(z0), p(pv(packs, 0, "tanc")), p(cw), p(XR),
                                       p(abar0), p(tan0), p(dw), None, st))
        torch.cuda.synchronize()
        na = int((abar0.view(torch.int32) != out["fused"][0]).sum())
        tb = (tan0 != out["fused"][1]).nonzero().flatten()
        dwe = (dw - out["fused"][2]).abs().max().item() / out["fused"][2].abs().max().item()
        if na or tb.numel() or dwe > 1e-5:
            print("rep %d: abar0 words differ %d, tan0 entries differ %d %s, dW rel %.2e" % (
                rep, na, tb.numel(), [(i // (MT0 * 48), (i % (MT0 * 48)) // 48, i % 48, tan0[i].item()) for i in tb[:6].tolist()], dwe))
    if reps:
        print("stress: %d repetitions done" % reps)
    a0, t0, w0 = out["old"]
    a1, t1, w1 = out["fused"]
    # per feature tile: relative distance of the bf16 adjoint blocks (which tiles / which row tiles are off?)
    f0 = a0.view(torch.bfloat16).float().view(nt, MT0, 256)
    f1 = a1.view(torch.bfloat16).float().view(nt, MT0, 256)
    per_kt = ((f0 - f1).norm(dim=(0, 2)) / f0.norm(dim=(0, 2))).tolist()
    print("abar0 rel distance per feature tile:", " ".join("%.1e" % v for v in per_kt))
    per_tile = ((f0 - f1).norm(dim=(1, 2)) / f0.norm(dim=(1, 2)))
    print("abar0 rel distance per row tile: first 8", " ".join("%.1e" % v for v in per_tile[:8].tolist()), " max %.1e at tile %d, min %.1e"
          % (per_tile.max().item(), int(per_tile.argmax()), per_tile.min().item()))
    # per row (lane & 15) of the blocks
    # independent torch reference of row tile 0: hbar0 = W1h^T abar1 from the MFMA operand images, then the softplus adjoint
    def ref_tile0():
        W = p16[(1, "WhT")].float().view(8, MT0, 64, 8)                       # [kp][kt][lane][e]: A[i = lane & 15][k = 8 (lane >> 4) + e]
        Bk = abar1.float().view(nt, S, 16, 64, 4)[0]                          # [st][m][lane][r]
        lane = torch.arange(64, device=dev)
        g, j = lane >> 4, lane & 15
        hbar = torch.zeros(MT0, S, 16, 16, device=dev)                         # [kt][st][feature i][row n]
        for kp in range(8):
            A = torch.zeros(MT0, 16, 32, device=dev)
            A[:, j[:, None], (8 * g)[:, None] + torch.arange(8, device=dev)[None, :]] = W[kp]
            Bm = torch.zeros(S, 32, 16, device=dev)
            for half in range(2):
                Bm[:, (8 * g)[:, None] + 4 * half + torch.arange(4, device=dev)[None, :], j[:, None]] = Bk[:, 2 * kp + half]
            hbar += torch.einsum("tik,skn->tsin", A, Bm)
        z = z0.view(nt, MT0, 64, 4)[0]                                         # [kt][lane][r]: feature 4g + r, row j
        zz = torch.zeros(MT0, 16, 16, device=dev)
        zz[:, (4 * g)[:, None] + torch.arange(4, device=dev)[None, :], j[:, None]] = z
        tc = pv(packs, 0, "tanc").view(3, MT0, 64, 4)[:, :, ::16, :].reshape(3, MT0, 16)     # [d][kt][feature]
        e = torch.exp(-zz.abs())
        inv = 1 / (1 + e)
        s = torch.where(zz >= 0, inv, e * inv)
        q2 = e * inv * inv
        s1, s2, s3 = s, q2, q2 * (1 - 2 * s)
        cwt = cw.view(nt, 2, 8)[0]                                             # [point][8]
        cq = cwt[(torch.arange(16, device=dev) >> 3)]                          # [row][8]
        a = [tc[dd][:, :, None].expand(MT0, 16, 16) for dd in range(3)]
        hb = [hbar[:, st] for st in range(S)]
        c = [cq[None, None, :, i] for i in range(6)]
        qq = a[0] * (c[0] * a[0] + c[1] * a[1] + c[2] * a[2]) + a[1] * (c[3] * a[1] + c[4] * a[2]) + c[5] * a[2] * a[2]
        ab0 = s1 * hb[0] + s2 * (a[0] * hb[1] + a[1] * hb[2] + a[2] * hb[3]) + (s3 * qq) * hb[4]
        return ab0                                                              # [kt][feature][row]
    if cfg.S2 == 1:
        ab0 = ref_tile0()
        for name, f in (("old", f0), ("fused", f1)):
            got = torch.zeros(MT0, 16, 16, device=dev)
            lane = torch.arange(64, device=dev)
            got[:, (4 * (lane >> 4))[:, None] + torch.arange(4, device=dev)[None, :], (lane & 15)[:, None]] = f[0].view(MT0, 64, 4)
            print("row tile 0, abar0 vs torch reference: %-5s rel %.3e" % (name, ((got - ab0).norm() / ab0.norm()).item()))
    bad = (a0 != a1).nonzero().flatten()
    print("abar0: %d of %d words differ" % (bad.numel(), a0.numel()), bad[:8].tolist())
    tb = ((t0 - t1).abs() > 1e-6 * t0.abs().max()) | torch.isnan(t1)
    idx = tb.nonzero().flatten()
    print("tan0: %d of %d differ (nan in fused: %d)" % (idx.numel(), t0.numel(), int(torch.isnan(t1).sum())))
    for i in idx[:12].tolist():
        tile, r = divmod(i, MT0 * 48)
        kt, r = divmod(r, 48)
        dd, f = divmod(r, 16)
        print("   tile %d kt %d d %d f %d: old %.6g fused %.6g" % (tile, kt, dd, f, t0[i].item(), t1[i].item()))
    err = (w0 - w1).abs().max().item() / w0.abs().max().item()
    print("dW1: max rel diff %.3e" % err)
    w0v, w1v = w0.view(256, 560), w1.view(256, 560)
    print("   hidden columns %.3e   raw-input columns %.3e" % ((w0v[:, :512] - w1v[:, :512]).abs().max().item() / w0.abs().max().item(),
                                                              (w0v[:, 512:] - w1v[:, 512:]).abs().max().item() / w0.abs().max().item()))


if __name__ == "__main__":
    main()
