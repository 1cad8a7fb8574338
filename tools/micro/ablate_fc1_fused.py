"""Timing of k_fc1_bwd_fused (csrc/jet_fc1_bwd.hip; bf16 mode, packed buffers, S = (3,1) softplus, nf = 32) alone, and of its
timing-only ablations: private builds with -DSTPDE_FC1F_ABL=n (results wrong by construction) on random buffers; the raw-input
launch of the ring kernel is stubbed out, so the figures are the fused kernel's own.

    python tools/micro/ablate_fc1_fused.py build [extra -D flags ...]    # here (no GPU): tools/micro/_abl/libfc1f_<n>.so
    python tools/micro/ablate_fc1_fused.py run                             # on the GPU box
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "space_time_pde_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "micro", "_abl")
VARIANTS = {0: "full kernel", 1: "no activation jets", 2: "no weight-gradient MFMAs", 3: "no input-gradient MFMAs",
            7: "full kernel with phase stamps"}
if os.environ.get("FC1F_VARIANTS"):      # subset, e.g. FC1F_VARIANTS=0,7
    VARIANTS = {int(v): VARIANTS[int(v)] for v in os.environ["FC1F_VARIANTS"].split(",")}
TAG = os.environ.get("FC1F_TAG", "")      # private builds with other compiler flags live side by side (FC1F_TAG=name)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics"]


def build(extra):
    os.makedirs(OUT, exist_ok=True)
    stub = os.path.join(OUT, "stub_fc1f.cpp")
    open(stub, "w").write('#include <hip/hip_runtime.h>\nstruct WgradArgs;\n' + "".join(
        "int stpde_wgrad_launch_%s(const WgradArgs&, int, hipStream_t) { return 0; }\n" % k for k in ("3_0", "3_1")))
    procs = []
    for n in VARIANTS:
        so = os.path.join(OUT, "libfc1f%s_%d.so" % (TAG, n))
        srcs = [os.path.join(CSRC, f) for f in ("jet_fc1_bwd.hip", "api.cpp")]
        defs = ["-DSTPDE_FC1F_STAMP=1"] if n == 7 else ["-DSTPDE_FC1F_ABL=%d" % n]
        cmd = ["hipcc"] + FLAGS + defs + extra + ["-shared", "-o", so] + srcs + [stub]
        procs.append((n, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for n, p in procs:
        out, _ = p.communicate()
        print("variant", n, "rc", p.returncode, out.decode()[-300:] if p.returncode else "")


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
    abar1 = (0.1 * torch.randn(nt * S * lay["MT"] * 128, device=dev)).to(torch.bfloat16)
    XR = torch.randn(nt * 3 * 256, device=dev)
    z0 = torch.randn(nt * MT0 * 256, device=dev)
    abar0 = torch.empty(nt * MT0 * 128, device=dev)
    tan0 = torch.empty(nt * MT0 * 48, device=dev)
    cw = torch.rand(nt * 2 * 8, device=dev)
    dw = torch.zeros(256 * 16 * 35, device=dev)
    d = _lib.LayerDesc()
    d.ntiles, d.KT, d.MT, d.first_hidden, d.cfg, d.mfma_bf16, d.packed = nt, lay["KT"], lay["MT"], 1, cfg, 1, 6
    p = _lib.ptr
    for n, what in VARIANTS.items():
        path = os.path.join(OUT, "libfc1f%s_%d.so" % (TAG, n))
        if not os.path.exists(path):
            continue
        L = C.CDLL(path)
        L.stpde_jet_fc1_bwd.argtypes = [C.POINTER(_lib.LayerDesc)] + [C.c_void_p] * 11
        st = _lib.stream_ptr()

        def call():
            rc = L.stpde_jet_fc1_bwd(C.byref(d), p(abar1), p(p16[(1, "WhT")]), p(z0), p(plan.pack_view(packs, 0, "tanc")), p(cw),
                                     p(XR), p(abar0), p(tan0), p(dw), None, st)
            assert rc == 0, rc

        call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(TAG, "variant %d  %-28s %7.3f ms per 2^18 row tiles  (x2 = %.2f ms per 2^20 points)" % (n, what, ms, 2 * ms))
        if n == 7:
            buf = (C.c_ulonglong * (8 * 4 * 8))()
            L.stpde_fc1f_stamp_read(buf)
            names = ["dgrad(0,1)", "epi 0,1", "dgrad(2,3)", "epi 2,3", "wgrad", "vmcnt(0)", "barrier"]
            for blk in range(0, 8, 3):
                for wv in range(4):
                    t = [buf[(blk * 4 + wv) * 8 + i] for i in range(8)]
                    print("   stamps block %d wave %d: " % (blk, wv) + "  ".join("%s %d" % (nm, t[i + 1] - t[i]) for i, nm in enumerate(names))
                          + "   | total %d cycles" % (t[7] - t[0]))


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        run()
