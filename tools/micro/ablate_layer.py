"""Timing-only ablations of the layer-1 kernels (forward / dgrad, S = (3,1) softplus, nf = 32): builds private copies of
the layer library with -DSTPDE_ABLATE=n (see jet_layer_impl.h) and times stpde_jet_layer_fwd / _bwd on random buffers.

    python tools/micro/ablate_layer.py build      # here (no GPU): compiles tools/micro/_abl/libabl_<n>.so
    python tools/micro/ablate_layer.py run        # on the GPU box
    python tools/micro/ablate_layer.py run bf16   # the same kernels with bf16 MFMA operands (BASELINE configs[3] mode)
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "space_time_pde_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "micro", "_abl")
VARIANTS = [0, 1, 2, 3, 4, 5, 6]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics"]


def build():
    os.makedirs(OUT, exist_ok=True)
    procs = []
    for n in VARIANTS:
        so = os.path.join(OUT, "libabl_%d.so" % n)
        srcs = [os.path.join(CSRC, f) for f in ("jet_layer.hip", "jet_layer_s31.hip", "api.cpp")]
        # the other stream configurations are stubbed out: only (3,1) is timed
        stub = os.path.join(OUT, "stub.cpp")
        open(stub, "w").write('#include <hip/hip_runtime.h>\nstruct LayerArgs;\n' + "".join(
            "int stpde_layer_launch_%s(const LayerArgs&, int, hipStream_t) { return 2; }\n" % k
            for k in ("0_0", "0_3", "3_0", "3_2", "3_4", "3_6")))
        cmd = ["hipcc"] + FLAGS + ["-DSTPDE_ABLATE=%d" % n, "-shared", "-o", so] + srcs + [stub]
        procs.append((n, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for n, p in procs:
        out, _ = p.communicate()
        print("variant", n, "rc", p.returncode, out.decode()[-300:] if p.returncode else "")


def build_stamp(extra=(), tag="stamp"):
    os.makedirs(OUT, exist_ok=True)
    so = os.path.join(OUT, "libabl_%s.so" % tag)
    srcs = [os.path.join(CSRC, f) for f in ("jet_layer.hip", "jet_layer_s31.hip", "api.cpp")]
    stub = os.path.join(OUT, "stub.cpp")
    open(stub, "w").write('#include <hip/hip_runtime.h>\nstruct LayerArgs;\n' + "".join(
        "int stpde_layer_launch_%s(const LayerArgs&, int, hipStream_t) { return 2; }\n" % k
        for k in ("0_0", "0_3", "3_0", "3_2", "3_4", "3_6")))
    r = subprocess.run(["hipcc"] + FLAGS + ["-DSTPDE_STAMP=1"] + list(extra) + ["-shared", "-o", so] + srcs + [stub], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT)
    print("stamp build rc", r.returncode, r.stdout.decode()[-300:] if r.returncode else "")


def stamp(bf16=False, x3=False):
    """Phase timeline of k_layer_coop (first hidden layer, forward and dgrad): mean cycles between the s_memtime stamps of
    256 mid-launch workgroups (all waves)."""
    import numpy as np
    import torch
    from space_time_pde_amd import _lib
    from space_time_pde_amd.lig_jet import ImNetPlan, make_cfg
    dev = torch.device("cuda:0")
    plan = ImNetPlan.get(3, 32, 4, 32)
    nt = 1 << 17
    cfg, S, _ = make_cfg("softplus", 0.0, True, [], {(1, 1): 1.0, (2, 2): 0.25})
    torch.manual_seed(0)
    packs = 0.05 * torch.randn(plan.n_pack, device=dev)
    X = torch.randn(nt * 3 * 256, device=dev)
    cw = torch.rand(nt * 2 * 8, device=dev)
    lay = plan.layers[1]
    out1 = torch.empty(nt * S * lay["MT"] * 256, device=dev)
    abar0 = torch.empty(nt * 4 * plan.layers[0]["MT"] * 256, device=dev)
    tan0 = torch.empty(nt * plan.layers[0]["MT"] * 48, device=dev)
    pv = plan.pack_view
    p16 = plan.pack_bf16(packs, 3 if x3 else 1) if (bf16 or x3) else {}
    L = C.CDLL(os.path.join(OUT, "libabl_stamp.so"))
    L.stpde_jet_layer_fwd.argtypes = [C.POINTER(_lib.LayerDesc)] + [C.c_void_p] * 12
    L.stpde_jet_layer_bwd.argtypes = [C.POINTER(_lib.LayerDesc)] + [C.c_void_p] * 13
    d = _lib.LayerDesc()
    d.ntiles, d.KT, d.MT, d.first_hidden, d.cfg, d.mfma_bf16 = nt, lay["KT"], lay["MT"], 1, cfg, (3 if x3 else int(bf16))
    st = _lib.stream_ptr()
    p = _lib.ptr

    def fwd():
        return L.stpde_jet_layer_fwd(C.byref(d), None, p(X), p(pv(packs, 1, "Wh")), p(pv(packs, 1, "Ws")),
                                     p(pv(packs, 1, "tanc")), p(pv(packs, 0, "Ws")), p(pv(packs, 0, "tanc")), p(out1),
                                     p(cw), p(p16.get((1, "Wh"))), p(abar0), st)

    def bwd():
        return L.stpde_jet_layer_bwd(C.byref(d), p(out1), p(pv(packs, 1, "WhT")), None, p(X), p(pv(packs, 0, "Ws")),
                                     p(pv(packs, 0, "tanc")), p(abar0), p(cw), None, p(p16.get((1, "WhT"))), p(tan0), p(abar0), st)

    names = ["start->ring free", "first produce", "barrier", "g0", "g1", "g2", "g3", "g4", "g5", "g6", "g7", "loop end",
             "epilogue"]
    for name, fn in (("fwd", fwd), ("dgrad", bwd)):
        for _ in range(2):
            assert fn() == 0
        torch.cuda.synchronize()
        host = (C.c_ulonglong * (256 * 8 * 16))()
        assert L.stpde_stamp_read(host) == 0
        a = np.frombuffer(host, dtype=np.uint64).reshape(256, 8, 16).astype(np.int64)
        print("== %s (%s): mean cycles per phase over the waves that recorded it; 100 MHz s_memtime ticks x 24 = shader cycles at 2.4 GHz"
              % (name, "fp32x3" if x3 else "bf16" if bf16 else "fp32"))
        idx = [0, 1, 2, 3] + list(range(4, 12)) + [12, 13]
        prev = 0
        for k, i in enumerate(idx[1:]):
            cur, pre = a[:, :, i], a[:, :, idx[k]]
            ok = (cur > 0) & (pre > 0) & (cur >= pre)
            if ok.any():
                print("  %-18s n=%5d  mean %9.1f ticks  (min %d, max %d)" % (names[k], ok.sum(), (cur - pre)[ok].mean(),
                                                                            (cur - pre)[ok].min(), (cur - pre)[ok].max()))
        tot = a[:, :, 13] - a[:, :, 0]
        ok = (a[:, :, 13] > 0) & (a[:, :, 0] > 0)
        print("  %-18s n=%5d  mean %9.1f ticks" % ("TOTAL wave life", ok.sum(), tot[ok].mean()))


def stamp_spec(tag="stamp"):
    """Step timeline of the wave-specialised bf16 forward (k_fc1_fwd_spec): iteration 64 of every workgroup; per role the mean
    cycles from the start of the iteration to the arrival at each step barrier and to its release."""
    import numpy as np
    import torch
    from space_time_pde_amd import _lib
    from space_time_pde_amd.lig_jet import ImNetPlan, make_cfg
    dev = torch.device("cuda:0")
    plan = ImNetPlan.get(3, 32, 4, 32)
    nt = 1 << 17
    cfg, S, _ = make_cfg("softplus", 0.0, True, [], {(1, 1): 1.0, (2, 2): 0.25})
    torch.manual_seed(0)
    packs = 0.05 * torch.randn(plan.n_pack, device=dev)
    X = torch.randn(nt * 3 * 256, device=dev)
    cw = torch.rand(nt * 2 * 8, device=dev)
    lay = plan.layers[1]
    out1 = torch.empty(nt * S * lay["MT"] * 256, device=dev)
    z0 = torch.empty(nt * plan.layers[0]["MT"] * 256, device=dev)
    pv = plan.pack_view
    p16 = plan.pack_bf16(packs, 1)
    L = C.CDLL(os.path.join(OUT, "libabl_%s.so" % tag))
    L.stpde_jet_layer_fwd.argtypes = [C.POINTER(_lib.LayerDesc)] + [C.c_void_p] * 12
    d = _lib.LayerDesc()
    d.ntiles, d.KT, d.MT, d.first_hidden, d.cfg, d.mfma_bf16 = nt, lay["KT"], lay["MT"], 1, cfg, 1
    p = _lib.ptr
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(3):
        if it == 1:
            e0.record()
        assert L.stpde_jet_layer_fwd(C.byref(d), None, p(X), p(pv(packs, 1, "Wh")), p(pv(packs, 1, "Ws")),
                                     p(pv(packs, 1, "tanc")), p(pv(packs, 0, "Ws")), p(pv(packs, 0, "tanc")), p(out1),
                                     p(cw), p(p16.get((1, "Wh"))), p(z0), _lib.stream_ptr()) == 0
    e1.record()
    torch.cuda.synchronize()
    print("== %s: %.3f ms per launch (2^18 points)" % (tag, e0.elapsed_time(e1) / 2))
    host = (C.c_ulonglong * (256 * 8 * 16))()
    assert L.stpde_stamp_read(host) == 0
    a = np.frombuffer(host, dtype=np.uint64).reshape(256, 8, 16).astype(np.int64)
    for role, sl in (("producers (waves 0-3)", slice(0, 4)), ("consumers (waves 4-7)", slice(4, 8))):
        r = a[:, sl, :]
        t0 = r[:, :, 0:1]
        print("== %s: mean cycles since the start of iteration 64" % role)
        for i, nm in ((1, "arrive barrier 0"), (2, "leave  barrier 0"), (3, "arrive barrier 1"), (4, "leave  barrier 1"),
                      (5, "arrive barrier 2"), (6, "leave  barrier 2"), (7, "arrive barrier 3"), (8, "leave  barrier 3"),
                      (9, "epilogue start (consumers)")):
            v = (r[:, :, i:i + 1] - t0)
            ok = (r[:, :, i:i + 1] > 0) & (t0 > 0)
            if ok.any():
                print("  %-28s %9.1f" % (nm, v[ok].mean()))


def run(bf16=False, tags=None, x3=False):
    import torch
    from space_time_pde_amd import _lib
    from space_time_pde_amd.lig_jet import ImNetPlan, make_cfg
    dev = torch.device("cuda:0")
    plan = ImNetPlan.get(3, 32, 4, 32)
    nt = 1 << 17                                  # 2^18 points
    cfg, S, _ = make_cfg("softplus", 0.0, True, [], {(1, 1): 1.0, (2, 2): 0.25})
    torch.manual_seed(0)
    packs = 0.05 * torch.randn(plan.n_pack, device=dev)
    X = torch.randn(nt * 3 * 256, device=dev)
    cw = torch.rand(nt * 2 * 8, device=dev)
    lay = plan.layers[1]
    out1 = torch.empty(nt * S * lay["MT"] * 256, device=dev)
    abar0 = torch.empty(nt * 4 * plan.layers[0]["MT"] * 256, device=dev)
    tan0 = torch.empty(nt * plan.layers[0]["MT"] * 48, device=dev)
    pv = plan.pack_view
    p16 = plan.pack_bf16(packs, 3 if x3 else 1) if (bf16 or x3) else {}
    res = {}
    for n in (tags or VARIANTS):                  # tags: private builds of build_flags (plain timings of fc1 fwd / dgrad)
        so = os.path.join(OUT, "libabl_%s.so" % n)
        if not os.path.exists(so):
            continue
        L = C.CDLL(so)
        L.stpde_jet_layer_fwd.argtypes = [C.POINTER(_lib.LayerDesc)] + [C.c_void_p] * 12
        L.stpde_jet_layer_bwd.argtypes = [C.POINTER(_lib.LayerDesc)] + [C.c_void_p] * 13
        d = _lib.LayerDesc()
        d.ntiles, d.KT, d.MT, d.first_hidden, d.cfg, d.mfma_bf16 = nt, lay["KT"], lay["MT"], 1, cfg, (3 if x3 else int(bf16))
        st = _lib.stream_ptr()
        p = _lib.ptr

        def fwd():
            return L.stpde_jet_layer_fwd(C.byref(d), None, p(X), p(pv(packs, 1, "Wh")), p(pv(packs, 1, "Ws")),
                                         p(pv(packs, 1, "tanc")), p(pv(packs, 0, "Ws")), p(pv(packs, 0, "tanc")), p(out1),
                                         p(cw), p(p16.get((1, "Wh"))), p(abar0), st)

        def bwd():
            return L.stpde_jet_layer_bwd(C.byref(d), p(out1), p(pv(packs, 1, "WhT")), None, p(X), p(pv(packs, 0, "Ws")),
                                         p(pv(packs, 0, "tanc")), p(abar0), p(cw), None, p(p16.get((1, "WhT"))), p(tan0), p(abar0), st)

        for name, fn in (("fwd", fwd), ("dgrad", bwd)):
            assert fn() == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res[(n, name)] = e0.elapsed_time(e1) / 3
    names = {0: "baseline", 1: "no activation jet in produce", 2: "no barrier in main loop", 3: "weights L1-resident",
             4: "no epilogue", 5: "no produce stage in loop"}
    for n in (tags or VARIANTS):
        if (n, "fwd") in res:
            print("fc1 %-28s fwd %7.3f ms   dgrad %7.3f ms" % (names.get(n, n), res[(n, "fwd")], res[(n, "dgrad")]))


def run2(stamp_too=True, tags=None, bf16=False, x3=False):
    """The SECOND hidden layer (fc2: 256 -> 128 features, exact fp32): ablation timings of its forward and input-gradient
    kernels, then the phase stamps of both (round 5: where do the 30 % idle matrix-pipe cycles of these two kernels go?)."""
    import numpy as np
    import torch
    from space_time_pde_amd import _lib
    from space_time_pde_amd.lig_jet import ImNetPlan, make_cfg
    dev = torch.device("cuda:0")
    plan = ImNetPlan.get(3, 32, 4, 32)
    nt = 1 << 17                                  # 2^18 points
    cfg, S, _ = make_cfg("softplus", 0.0, True, [], {(1, 1): 1.0, (2, 2): 0.25})
    torch.manual_seed(0)
    packs = 0.05 * torch.randn(plan.n_pack, device=dev)
    X = torch.randn(nt * 3 * 256, device=dev)
    cw = torch.rand(nt * 2 * 8, device=dev)
    lay = plan.layers[2]
    pre1 = torch.randn(nt * S * lay["KT"] * 256, device=dev)
    out2 = torch.empty(nt * S * lay["MT"] * 256, device=dev)
    abar2 = torch.randn(nt * S * lay["MT"] * 256, device=dev)
    pv = plan.pack_view
    p = _lib.ptr
    p16 = plan.pack_bf16(packs, 3 if x3 else 1) if (bf16 or x3) else {}       # (unpacked layer buffers: the timeline, not the bytes, is the point)

    def calls(L):
        L.stpde_jet_layer_fwd.argtypes = [C.POINTER(_lib.LayerDesc)] + [C.c_void_p] * 12
        L.stpde_jet_layer_bwd.argtypes = [C.POINTER(_lib.LayerDesc)] + [C.c_void_p] * 13
        d = _lib.LayerDesc()
        d.ntiles, d.KT, d.MT, d.first_hidden, d.cfg, d.mfma_bf16 = nt, lay["KT"], lay["MT"], 0, cfg, (3 if x3 else int(bf16))
        st = _lib.stream_ptr()
        work = pre1.clone()

        def fwd():
            return L.stpde_jet_layer_fwd(C.byref(d), p(pre1), p(X), p(pv(packs, 2, "Wh")), p(pv(packs, 2, "Ws")),
                                         p(pv(packs, 2, "tanc")), None, None, p(out2), p(cw), p(p16.get((2, "Wh"))), None, st)

        def bwd():
            return L.stpde_jet_layer_bwd(C.byref(d), p(abar2), p(pv(packs, 2, "WhT")), p(work), None, None, None, None, p(cw),
                                         None, p(p16.get((2, "WhT"))), None, None, st)
        return (("fwd", fwd), ("dgrad", bwd))

    res = {}
    if tags:                                      # private builds of build_stamp(extra flags, tag): plain timings
        for tag in tags:
            for name, fn in calls(C.CDLL(os.path.join(OUT, "libabl_%s.so" % tag))):
                assert fn() == 0, (tag, name)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                res[(tag, name)] = e0.elapsed_time(e1) / 5
        for tag in tags:
            print("%-24s fwd %7.3f ms   dgrad %7.3f ms" % (tag, res[(tag, "fwd")], res[(tag, "dgrad")]))
        return
    for n in VARIANTS:
        so = os.path.join(OUT, "libabl_%d.so" % n)
        if not os.path.exists(so):
            continue
        for name, fn in calls(C.CDLL(so)):
            assert fn() == 0, (n, name)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res[(n, name)] = e0.elapsed_time(e1) / 3
    names = {0: "baseline", 1: "no activation jet in produce", 2: "no barrier in main loop", 3: "weights L1-resident",
             4: "no epilogue", 5: "no produce stage in loop"}
    print("== fc2 (KT %d, MT %d), 2^18 points, exact fp32" % (lay["KT"], lay["MT"]))
    for n in VARIANTS:
        if (n, "fwd") in res:
            print("%-32s fwd %7.3f ms   dgrad %7.3f ms" % (names[n], res[(n, "fwd")], res[(n, "dgrad")]))
    so = os.path.join(OUT, "libabl_stamp.so")
    if not (stamp_too and os.path.exists(so)):
        return
    L = C.CDLL(so)
    labels = ["start->ring free", "first produce", "barrier", "g0", "g1", "g2", "g3", "g4", "g5", "g6", "g7", "loop end",
              "epilogue"]
    for name, fn in calls(L):
        for _ in range(2):
            assert fn() == 0
        torch.cuda.synchronize()
        host = (C.c_ulonglong * (256 * 8 * 16))()
        assert L.stpde_stamp_read(host) == 0
        a = np.frombuffer(host, dtype=np.uint64).reshape(256, 8, 16).astype(np.int64)
        print("== fc2 %s: mean ticks per phase over the waves that recorded it (100 MHz s_memtime ticks x 24 = shader cycles)" % name)
        idx = [0, 1, 2, 3] + list(range(4, 12)) + [12, 13]
        for k, i in enumerate(idx[1:]):
            cur, pre = a[:, :, i], a[:, :, idx[k]]
            ok = (cur > 0) & (pre > 0) & (cur >= pre)
            if ok.any():
                print("  %-18s n=%5d  mean %9.1f ticks  (min %d, max %d)" % (labels[k], ok.sum(), (cur - pre)[ok].mean(),
                                                                            (cur - pre)[ok].min(), (cur - pre)[ok].max()))
        tot = a[:, :, 13] - a[:, :, 0]
        ok = (a[:, :, 13] > 0) & (a[:, :, 0] > 0)
        print("  %-18s n=%5d  mean %9.1f ticks" % ("TOTAL wave life", ok.sum(), tot[ok].mean()))


if __name__ == "__main__":
    if sys.argv[1:] == ["build"]:
        build()
        build_stamp()
    elif sys.argv[1] == "stamp_spec":
        stamp_spec(sys.argv[2] if len(sys.argv) > 2 else "stamp")
    elif sys.argv[1] == "run2":
        run2(tags=[t for t in sys.argv[2:] if t not in ("bf16", "x3")] or None, bf16="bf16" in sys.argv[2:], x3="x3" in sys.argv[2:])
    elif sys.argv[1] == "build_flags":            # build_flags TAG -DFOO=1 ...  (a plain private build, no stamps)
        build_stamp(extra=["-DSTPDE_STAMP=0"] + sys.argv[3:], tag=sys.argv[2])
    elif sys.argv[1] == "stamp":
        stamp(bf16="bf16" in sys.argv[2:], x3="x3" in sys.argv[2:])
    elif sys.argv[1] == "run1":                   # run1 TAG ... [bf16]: the first hidden layer on private builds
        run(bf16="bf16" in sys.argv[2:], tags=[t for t in sys.argv[2:] if t not in ("bf16", "x3")], x3="x3" in sys.argv[2:])
    else:
        run(bf16="bf16" in sys.argv[2:], x3="x3" in sys.argv[2:])
