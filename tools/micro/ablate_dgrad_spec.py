"""Timing-only ablations of k_fc1_dgrad_spec (bf16 mode, packed buffers, S = (3,1) softplus, nf = 32): private builds of the
layer library with -DSTPDE_DSPEC_ABL=n (csrc/jet_spec_bf16.h; results are wrong by construction) timed on random buffers.

    python tools/micro/ablate_dgrad_spec.py build     # here (no GPU): tools/micro/_abl/libdspec_<n>.so
    python tools/micro/ablate_dgrad_spec.py run       # on the GPU box
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "space_time_pde_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "micro", "_abl")
VARIANTS = {0: "full kernel", 1: "weight fragments not re-fetched", 2: "no activation-jet adjoint", 3: "no LDS-DMA pieces",
            4: "no global stores", 5: "no MFMAs", 6: "no DPP row sums", 7: "cooperative kernel (STPDE_BF_SPEC_DGRAD=0)",
            8: "one output tile per wave and step (STPDE_DSPEC_TPS=1)"}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics"]


def build():
    os.makedirs(OUT, exist_ok=True)
    stub = os.path.join(OUT, "stub.cpp")
    open(stub, "w").write('#include <hip/hip_runtime.h>\nstruct LayerArgs;\n' + "".join(
        "int stpde_layer_launch_%s(const LayerArgs&, int, hipStream_t) { return 2; }\n" % k
        for k in ("0_0", "0_3", "3_0", "3_2", "3_6")))
    procs = []
    for n in VARIANTS:
        if n >= 7:
            continue
        so = os.path.join(OUT, "libdspec_%d.so" % n)
        srcs = [os.path.join(CSRC, f) for f in ("jet_layer.hip", "jet_layer_s31.hip", "api.cpp")]
        cmd = ["hipcc"] + FLAGS + ["-DSTPDE_DSPEC_ABL=%d" % n, "-shared", "-o", so] + srcs + [stub]
        procs.append((n, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for n, p in procs:
        out, _ = p.communicate()
        print("variant", n, "rc", p.returncode, out.decode()[-300:] if p.returncode else "")
    import shutil      # variant 7 = the full build under another name (the environment switch is read once per loaded library)
    shutil.copy(os.path.join(OUT, "libdspec_0.so"), os.path.join(OUT, "libdspec_7.so"))
    shutil.copy(os.path.join(OUT, "libdspec_0.so"), os.path.join(OUT, "libdspec_8.so"))


def run():
    import torch
    from space_time_pde_amd import _lib
    from space_time_pde_amd.lig_jet import ImNetPlan, make_cfg
    dev = torch.device("cuda:0")
    plan = ImNetPlan.get(3, 32, 4, 32)
    nt = 1 << 18
    cfg, S, _ = make_cfg("softplus", 0.0, True, [], {(1, 1): 1.0, (2, 2): 0.25})
    torch.manual_seed(0)
    packs = 0.05 * torch.randn(plan.n_pack, device=dev)
    p16 = plan.pack_bf16(packs, 1)
    lay = plan.layers[1]
    MT0 = plan.layers[0]["MT"]
    abar1 = (0.1 * torch.randn(nt * S * lay["MT"] * 128, device=dev)).to(torch.bfloat16)      # packed ADJOINT: every stream bf16
    X = torch.randn(nt * 3 * 256, device=dev)
    z0 = torch.randn(nt * MT0 * 256, device=dev)
    abar0 = torch.empty(nt * MT0 * 128, device=dev)                                              # bf16 blocks (value stream)
    tan0 = torch.empty(nt * MT0 * 48, device=dev)
    cw = torch.rand(nt * 2 * 8, device=dev)
    pv = plan.pack_view
    d = _lib.LayerDesc()
    d.ntiles, d.KT, d.MT, d.first_hidden, d.cfg, d.mfma_bf16, d.packed = nt, lay["KT"], lay["MT"], 1, cfg, 1, 6
    p = _lib.ptr
    res = {}
    for n, what in VARIANTS.items():
        if n == 7:
            os.environ["STPDE_BF_SPEC_DGRAD"] = "0"
        if n == 8:
            os.environ["STPDE_DSPEC_TPS"] = "1"
        L = C.CDLL(os.path.join(OUT, "libdspec_%d.so" % n))
        L.stpde_jet_layer_bwd.argtypes = [C.POINTER(_lib.LayerDesc)] + [C.c_void_p] * 13
        st = _lib.stream_ptr()

        def call():
            rc = L.stpde_jet_layer_bwd(C.byref(d), p(abar1), p(pv(packs, 1, "WhT")), None, p(X), p(pv(packs, 0, "Ws")),
                                       p(pv(packs, 0, "tanc")), p(abar0), p(cw), None, p(p16[(1, "WhT")]), p(tan0), p(z0), st)
            assert rc == 0, rc
        call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            call()
        e1.record()
        torch.cuda.synchronize()
        res[n] = e0.elapsed_time(e1) / 5
        print("%d  %-45s %7.3f ms per 2^18-row-tile launch (x2 = per 2^20-point step)" % (n, what, res[n]), flush=True)
        os.environ.pop("STPDE_BF_SPEC_DGRAD", None)
        os.environ.pop("STPDE_DSPEC_TPS", None)
    return res


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        run()
