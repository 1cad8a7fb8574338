"""Per-layer timing of the UNet3d convolutions (HIP kernels) and of the whole UNet forward / backward.

Usage (GPU box):  python tools/unet_profile.py [T Z X]
"""
import collections
import ctypes as C
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from space_time_pde_amd import _lib, unet3d  # noqa: E402


CONV_CALLS = ("stpde_conv3d_fwd", "stpde_conv3d_wgrad", "stpde_conv3d_wgrad_bias", "stpde_conv3d_wgrad_onload",
              "stpde_conv3d_fused")
BN_CALLS = ("stpde_bn_fwd", "stpde_bn_bwd")


class _Proxy:
    """Brackets every convolution entry point of the library with a pair of events (round 5: the fused residual-block calls
    stpde_conv3d_fused / stpde_conv3d_wgrad_bias / stpde_conv3d_wgrad_onload as well -- round 4's tables came out empty
    because only the layer-wise entry points were wrapped).  Weight gradients that the U-Net defers to its side stream are
    bracketed on THAT stream (the events are recorded on the current stream of the call)."""

    def __init__(self, real, rec):
        self._real, self._rec = real, rec

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if name in BN_CALLS:
            return self._bn(name, fn)
        if name not in CONV_CALLS:
            return fn

        def wrapped(desc, *args):
            d = C.cast(desc, C.POINTER(_lib.Conv3dFusedArgs if name == "stpde_conv3d_fused" else _lib.Conv3dDesc)).contents
            ci2 = co2 = 0
            if name == "stpde_conv3d_fused":
                ci2, co2 = d.Ci2 if d.x2 else 0, d.Co2 if d.y2 else 0
                d = d.d
            key = (name[6:], d.B * d.T * d.Z * d.X, d.Ci + ci2, d.Co + co2, d.ksize)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(desc, *args)
            e1.record()
            self._rec.append((key, e0, e1))
            return rc
        return wrapped


def _bn_method(self, name, fn):
    """BatchNorm entry points: key carries the number of whole-tensor HBM passes the call makes (csrc/bn.hip), so that the
    table can print a rate.  forward: statistics pass (unless the convolution's epilogue delivered them: stats_mode 2) reads x;
    elementwise pass reads x (+ residual), writes y.  backward: reduction pass (unless reduce_done) reads dy, x (+ y under a
    ReLU); elementwise pass reads dy, x (+ y), writes dx (+ d residual)."""
    def wrapped(desc, *args):
        d = C.cast(desc, C.POINTER(_lib.BnDesc)).contents
        if name == "stpde_bn_fwd":
            x, res = args[0], args[1]
            passes = (1 if d.training and d.stats_mode != 2 else 0) + 2 + (1 if res else 0)
        else:
            x, y, dy, gamma, stat, bsum, dx, dres = args[:8]
            rd = 2 + (1 if d.relu else 0)
            passes = (0 if d.reduce_done else rd) + rd + (1 if dx else 0) + (1 if dres else 0)
        key = (name[6:], d.N, d.C, passes, 0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(desc, *args)
        e1.record()
        self._bnrec.append((key, e0, e1))
        return rc
    return wrapped


_Proxy._bn = _bn_method
_Proxy._bnrec = []


def main():
    igres = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (32, 128, 128)
    dev = torch.device("cuda:0")
    real = _lib.lib()
    rec = []
    proxy = _Proxy(real, rec)
    _lib.lib = lambda: proxy
    unet3d._lib.lib = _lib.lib
    torch.manual_seed(0)
    net = unet3d.UNet3d(in_features=4, out_features=32, igres=igres, nf=16, mf=256).to(dev).train()
    x = torch.randn(1, 4, *igres, device=dev)
    g = torch.randn(1, *igres, 32, device=dev).permute(0, 4, 1, 2, 3)   # channels-last, as the LIG backward delivers it
    tf, tb = [], []
    for it in range(4):
        if it == 1:
            rec.clear()
            del _Proxy._bnrec[:]
        for p in net.parameters():
            p.grad = None
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        y = net(x)
        e[1].record()
        y.backward(g)
        e[2].record()
        torch.cuda.synchronize()
        if it:
            tf.append(e[0].elapsed_time(e[1]))
            tb.append(e[1].elapsed_time(e[2]))
    print("UNet3d igres=%s: forward %.2f ms, backward %.2f ms (mean of 3)" % (igres, sum(tf) / 3, sum(tb) / 3))
    agg = collections.OrderedDict()
    for key, e0, e1 in rec:
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += e0.elapsed_time(e1)
    tot = 0.0
    print("%-14s %9s %4s %4s %2s %6s %9s %9s %8s" % ("kernel", "nvox", "Ci", "Co", "k", "calls", "ms/step", "us/call", "TFLOP/s"))
    for (kind, nvox, ci, co, k), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        # (a two-output call -- conv1 + shortcut -- reads Ci once for Co + Co2 outputs; a two-input call sums Ci + Ci2 inputs)
        flop = 2.0 * nvox * ci * co * k ** 3
        print("%-14s %9d %4d %4d %2d %6d %9.3f %9.1f %8.1f" % (kind, nvox, ci, co, k, n // 3, ms / 3, 1e3 * ms / n,
                                                              flop / (ms / n * 1e-3) / 1e12))
        tot += ms / 3
    print("conv kernels total: %.2f ms/step (event-bracketed, includes launch gaps)" % tot)
    agg = collections.OrderedDict()
    for key, e0, e1 in _Proxy._bnrec:
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += e0.elapsed_time(e1)
    tot = 0.0
    print("%-8s %9s %4s %6s %6s %9s %9s %7s" % ("bn call", "nvox", "C", "passes", "calls", "ms/step", "us/call", "TB/s"))
    for (kind, nvox, c, passes, _), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-8s %9d %4d %6d %6d %9.3f %9.1f %7.2f" % (kind, nvox, c, passes, n // 3, ms / 3, 1e3 * ms / n,
                                                         4.0 * nvox * c * passes / (ms / n * 1e-3) / 1e12))
        tot += ms / 3
    print("batch-norm calls total: %.2f ms/step" % tot)


if __name__ == "__main__":
    main()
