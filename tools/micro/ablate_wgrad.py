"""Timing-only ablations of the first-hidden-layer weight-gradient kernel in the bf16-pipe modes (S = (3,1) softplus,
nf = 32): private builds of jet_wgrad.hip + jet_wgrad_s31.hip with -DSTPDE_ABLATE_W=n (see jet_wgrad_impl.h).

    python tools/micro/ablate_wgrad.py build      # here (no GPU)
    python tools/micro/ablate_wgrad.py run        # on the GPU box
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "space_time_pde_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "micro", "_abl")
VARIANTS = [int(v) for v in os.environ.get("ABLW_VARIANTS", "0,1,2,3,4,5").split(",")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics"]


def build():
    os.makedirs(OUT, exist_ok=True)
    procs = []
    stub = os.path.join(OUT, "stubw.cpp")
    open(stub, "w").write('#include <hip/hip_runtime.h>\nstruct WgradArgs;\n' + "".join(
        "int stpde_wgrad_launch_%s(const WgradArgs&, int, hipStream_t) { return 2; }\n" % k
        for k in ("0_0", "3_0", "3_2", "3_4", "3_6")))
    for n in VARIANTS:
        so = os.path.join(OUT, "libablw_%d.so" % n)
        srcs = [os.path.join(CSRC, f) for f in ("jet_wgrad.hip", "jet_wgrad_s31.hip", "api.cpp")]
        cmd = ["hipcc"] + FLAGS + ["-DSTPDE_ABLATE_W=%d" % n, "-shared", "-o", so] + srcs + [stub]
        procs.append((n, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for n, p in procs:
        out, _ = p.communicate()
        print("variant", n, "rc", p.returncode, out.decode()[-300:] if p.returncode else "")


def build_flags(tag, extra):
    """private build libablw_<tag>.so with extra compiler flags (A/B of a compile-time switch)"""
    os.makedirs(OUT, exist_ok=True)
    stub = os.path.join(OUT, "stubw.cpp")
    open(stub, "w").write('#include <hip/hip_runtime.h>\nstruct WgradArgs;\n' + "".join(
        "int stpde_wgrad_launch_%s(const WgradArgs&, int, hipStream_t) { return 2; }\n" % k
        for k in ("0_0", "3_0", "3_2", "3_4", "3_6")))
    srcs = [os.path.join(CSRC, f) for f in ("jet_wgrad.hip", "jet_wgrad_s31.hip", "api.cpp")]
    r = subprocess.run(["hipcc"] + FLAGS + list(extra) + ["-shared", "-o", os.path.join(OUT, "libablw_%s.so" % tag)] + srcs + [stub],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    print("build", tag, "rc", r.returncode, r.stdout.decode()[-400:] if r.returncode else "")


def run(tags=None):
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
    XR = torch.randn(nt * 3 * 256, device=dev)
    z0 = torch.randn(nt * plan.layers[0]["MT"] * 256, device=dev)
    cw = torch.rand(nt * 2 * 8, device=dev)
    lay = plan.layers[1]
    abar1 = torch.randn(nt * S * lay["MT"] * 256, device=dev)
    off, mp, ka = plan.dw_off[1]
    dw = torch.zeros(mp * ka, device=dev)
    pv = plan.pack_view
    names = {0: "baseline", 1: "one partial product instead of six", 2: "abar transposed / split for the first tile only",
             3: "no produce stage in the loop", 4: "consumer operands not read from LDS", 5: "no barrier in the loop"}
    for mode, tag in ((3, "fp32x3"), (1, "bf16"), (0, "fp32")):
        for n in (tags or VARIANTS):
            so = os.path.join(OUT, "libablw_%s.so" % n)
            if not os.path.exists(so) or (not tags and mode != 3 and n not in (0, 2, 3, 5)):
                continue
            L = C.CDLL(so)
            L.stpde_jet_wgrad.argtypes = [C.POINTER(_lib.LayerDesc), C.c_int] + [C.c_void_p] * 7
            d = _lib.LayerDesc()
            d.ntiles, d.KT, d.MT, d.first_hidden, d.cfg, d.mfma_bf16 = nt, lay["KT"], lay["MT"], 1, cfg, mode
            st = _lib.stream_ptr()
            p = _lib.ptr

            def fn():
                return L.stpde_jet_wgrad(C.byref(d), S, p(abar1), p(z0), p(XR), p(pv(packs, 0, "tanc")), p(dw), p(cw), st)

            assert fn() == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                fn()
            e1.record()
            torch.cuda.synchronize()
            print("%-7s %-50s %7.3f ms" % (tag, names.get(n, n), e0.elapsed_time(e1) / 3))


if __name__ == "__main__":
    if sys.argv[1:] == ["build"]:
        build()
    elif sys.argv[1] == "build_flags":            # build_flags TAG -DFOO=1 ...
        build_flags(sys.argv[2], sys.argv[3:])
    else:
        run(tags=sys.argv[2:] or None)            # run [TAG ...]
