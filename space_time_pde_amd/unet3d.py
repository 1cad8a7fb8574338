"""3-D U-Net latent-grid encoder (mirrors src/unet3d.py:12-240 of the reference: same constructor arguments,
submodule names and state_dict keys, so reference checkpoints load).

On CUDA tensors every convolution (1x1x1 and 3x3x3) runs in the HIP implicit-GEMM kernels of libstpde_hip
(``stpde_conv3d_fwd`` for forward and input-gradient, ``stpde_conv3d_wgrad`` for the weight gradient) on
channels-last activations [B, T, Z, X, C]; the network output is returned as a channels-last *view* of logical
shape [B, C, T, Z, X], so the ``latent_grid.permute(0, 2, 3, 4, 1)`` of experiments/rb2d/train.py:60 is a free
contiguous view that feeds the local-implicit-grid kernels directly.  BatchNorm (+ residual add) (+ ReLU) run in the
fused HIP kernels ``stpde_bn_fwd`` / ``stpde_bn_bwd``, max pooling and nearest up-sampling in ``stpde_resample3d``;
only the channel concatenation of the skip connections is a torch copy.
``Encoder3d`` of the reference (src/unet3d.py:243-344) is dead code there and is not provided.
"""
import ctypes as C
import math
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib

# pylint: disable=invalid-name, too-many-instance-attributes, arguments-differ, too-many-arguments

_idx_cache = {}


def _pack_indices(co, ci, k, device):
    """Index maps from a flattened Conv3d weight [co, ci, k, k, k] (+ one trailing zero) to the A-operand packs of
    the forward conv and of the input-gradient conv (transposed, taps flipped)."""
    key = (co, ci, k, str(device))
    if key in _idx_cache:
        return _idx_cache[key]
    ntap = k ** 3
    cip, cop = (ci + 15) // 16 * 16, (co + 15) // 16 * 16
    zero = co * ci * ntap
    lane = np.arange(64)
    g, j = lane >> 4, lane & 15
    r = np.arange(4)

    def widx(o, i, t):
        ok = (o < co) & (i < ci)
        return np.where(ok, (np.minimum(o, co - 1) * ci + np.minimum(i, ci - 1)) * ntap + t, zero)

    tap = np.arange(ntap)[:, None, None, None, None]
    # forward: [tap][kt][mt][lane][r] = W[16mt + j][16kt + 4g + r][tap]
    kt = np.arange(cip // 16)[None, :, None, None, None]
    mt = np.arange(cop // 16)[None, None, :, None, None]
    fwd = widx(16 * mt + j[None, None, None, :, None], 16 * kt + 4 * g[None, None, None, :, None] + r, tap)
    # dgrad: out channels = ci, in channels = co: [tap'][kt'][mt'][lane][r] = W[16kt' + 4g + r][16mt' + j][ntap-1-tap']
    ktd = np.arange(cop // 16)[None, :, None, None, None]
    mtd = np.arange(cip // 16)[None, None, :, None, None]
    bwd = widx(16 * ktd + 4 * g[None, None, None, :, None] + r, 16 * mtd + j[None, None, None, :, None] + 0 * r,
               ntap - 1 - tap)
    out = (torch.from_numpy(fwd.reshape(-1)).to(device), torch.from_numpy(bwd.reshape(-1)).to(device), cip, cop)
    _idx_cache[key] = out
    return out


# Deterministic mode (round 6, VERDICT r5 missing #4 / next #6; ``_lib.deterministic``, STPDE_DETERMINISTIC=1).  Every sum the
# U-Net kernels accumulate with atomics -- convolution weight / bias gradients, the BatchNorm statistics of the fused
# convolution epilogues, the BatchNorm-backward sums -- goes to order-independent long accumulators (csrc/common.h: six 64-bit
# integer windows per element, integer atomics), and the deep levels' forward does not split its taps over workgroups: two
# runs of the same step give bit-identical outputs, input gradients and parameter gradients, like the reference's CPU path
# (experiments/rb2d/train.py:77).  Cost: 48 bytes of zero-filled scratch per weight element per step (~450 MB at
# configs[1]) and one finalize pass; timings in DESIGN 7.
def _det():
    return 1 if _lib.deterministic else 0


def _acc_zeros(n, device):
    """zero-filled destination of n accumulated floats: n floats, or -- deterministic mode -- n long accumulators"""
    return torch.zeros(n * (2 * _lib.DET_K if _lib.deterministic else 1), device=device)


def _acc_value(acc, n):
    """fp32 values of a destination made by _acc_zeros (deterministic mode: stpde_det_finalize into a fresh tensor)"""
    if not _lib.deterministic:
        return acc
    out = torch.empty(n, device=acc.device)
    _lib.check(_lib.lib().stpde_det_finalize(_lib.ptr(acc), n, _lib.ptr(out), _lib.stream_ptr()))
    return out


def _desc(x, ci, co, k):
    d = _lib.Conv3dDesc()
    d.B, d.T, d.Z, d.X = x.shape[0], x.shape[1], x.shape[2], x.shape[3]
    d.Ci, d.Co, d.ksize = ci, co, k
    d.det = _det()
    return d


_side_streams = {}


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


class _DeferredGrads:
    """Weight / bias gradients of the U-Net convolutions off the critical path of the backward pass (opt-in:
    ``UNet3d.deferred_weight_grads``; ``train_step.sharded_step`` turns it on).

    The backward of a convolution needs its input gradient for the next node, but nothing downstream needs its weight
    gradient before the optimizer step.  In this mode ``_Conv3dHip.backward`` launches the weight-gradient kernel and the
    bias reduction on a SIDE stream (after an event on the stream that produced its inputs), returns ``None`` for them,
    and one callback queued on the autograd engine -- it runs when the whole backward pass is over -- makes the main
    stream wait for the side stream and assigns every ``.grad`` from the per-step gradient buffer with ONE index gather
    (instead of one permute-copy per convolution).  The deep levels of the U-Net are a chain of 5-50 us kernels that leave
    most of the chip idle; the weight-gradient kernels fill it.  Only ``loss.backward()`` sees the gradients (they are
    accumulated into ``.grad``); ``torch.autograd.grad(..., unet.parameters())`` does not -- hence opt-in."""

    def __init__(self, convs, dwall, sizes, device):
        self.convs, self.dwall, self.device = convs, dwall, device
        self.side = _side_stream(device)
        nb = [c.weight.shape[0] if c.bias is not None else 0 for c in convs]
        self.nbias = max(1, sum(nb))
        self.dball = _acc_zeros(self.nbias, device)
        self.boff = np.concatenate([[0], np.cumsum(nb)]).astype(np.int64)
        self.acc_w = 2 * _lib.DET_K if _lib.deterministic else 1      # floats of storage per accumulated element
        self.used = set()
        self.keep = []
        self.direct_bias = {}        # bias gradients that did not come out of a weight-gradient kernel (frozen weights)
        self.queued = False

    def bias_slice(self, i):
        return self.dball[self.acc_w * int(self.boff[i]):self.acc_w * int(self.boff[i + 1])]

    def enqueue(self, i):
        self.used.add(i)
        if not self.queued:
            self.queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self.finalize)

    @staticmethod
    def unpack_index(convs, sizes, device):
        """position in the flat per-step buffer (tap-major [ntap][co][ci padded] per convolution) of every weight element,
        in the order of the concatenated parameters"""
        chunks, off = [], 0
        for c, n in zip(convs, sizes):
            co, ci, k = c.weight.shape[0], c.weight.shape[1], c.weight.shape[2]
            cip = (ci + 15) // 16 * 16
            o = np.arange(co)[:, None, None]
            i = np.arange(ci)[None, :, None]
            t = np.arange(k ** 3)[None, None, :]
            chunks.append((off + (t * co + o) * cip + i).reshape(-1))
            off += n
        return torch.from_numpy(np.concatenate(chunks)).to(device)

    def finalize(self):
        self.queued = False
        with torch.cuda.device(self.device):
            torch.cuda.current_stream().wait_stream(self.side)
            self.keep = []                           # operands of the side-stream kernels: free for reuse on this stream now
            if self.acc_w != 1:                      # deterministic mode: long accumulators -> fp32, then as usual
                self.dwall = _acc_value(self.dwall, self.dwall.numel() // self.acc_w)
                self.dball = _acc_value(self.dball, self.nbias)
                self.acc_w = 1
            gflat = self.dwall[self.uidx]            # one gather: every weight gradient in parameter layout
            o = 0
            for i, c in enumerate(self.convs):
                n = c.weight.numel()
                if i in self.used:
                    if c.weight.requires_grad:
                        g = gflat[o:o + n].view_as(c.weight)
                        c.weight.grad = g if c.weight.grad is None else c.weight.grad + g
                    if c.bias is not None and c.bias.requires_grad:
                        gb = self.direct_bias[i] if i in self.direct_bias else self.bias_slice(i)
                        c.bias.grad = gb if c.bias.grad is None else c.bias.grad + gb
                o += n
        self.used = set()


class _Conv3dHip(torch.autograd.Function):
    """y = conv3d(x, weight, bias), stride 1, padding (k-1)/2, on channels-last x [B,T,Z,X,Ci].

    packs = (forward A-operand pack, input-gradient pack) prepared by the caller (UNet3d packs all of its convolutions
    with one gather per step), or None: packed here."""

    @staticmethod
    @_lib.guarded
    def forward(ctx, x, weight, bias, packs=None):
        L = _lib.lib()
        co, ci, k = weight.shape[0], weight.shape[1], weight.shape[2]
        cip, cop = (ci + 15) // 16 * 16, (co + 15) // 16 * 16
        if cop != co:
            raise NotImplementedError("HIP conv3d needs out_channels to be a multiple of 16 (got %d)" % co)
        ctx.dwbuf = ctx.defer = None
        if packs is None:
            fidx, bidx, _, _ = _pack_indices(co, ci, k, x.device)
            wflat = torch.cat([weight.detach().reshape(-1), weight.new_zeros(1)])
            fpack, bpack = wflat[fidx], None
        else:
            fpack, bpack, ctx.dwbuf = packs[:3]
            if len(packs) > 3:
                ctx.defer = packs[3:5]              # (_DeferredGrads, index of this convolution in it)
            wflat = bidx = None
        xin = x.detach()
        if cip != ci:
            xin = F.pad(xin, (0, cip - ci))
        xin = xin.contiguous()
        y = torch.empty(x.shape[:-1] + (co,), device=x.device, dtype=torch.float32)
        d = _desc(xin, cip, co, k)
        _lib.check(L.stpde_conv3d_fwd(C.byref(d), _lib.ptr(xin), _lib.ptr(fpack),
                                      _lib.ptr(bias.detach().contiguous()) if bias is not None else None,
                                      _lib.ptr(y), _lib.stream_ptr()))
        ctx.save_for_backward(xin, wflat if bpack is None else bpack)
        ctx.meta = (co, ci, k, cip, bidx, bias is not None)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    @_lib.guarded
    def backward(ctx, gy):
        L = _lib.lib()
        xin, wsaved = ctx.saved_tensors
        co, ci, k, cip, bidx, has_bias = ctx.meta
        gy = gy.contiguous()
        dx = dw = db = None
        ready = torch.cuda.current_stream().record_event() if ctx.defer is not None else None   # gy is complete here
        if ctx.needs_input_grad[0]:
            dxp = torch.empty(xin.shape, device=gy.device, dtype=torch.float32)
            d = _desc(gy, co, cip, k)
            bpack = wsaved if bidx is None else wsaved[bidx]
            _lib.check(L.stpde_conv3d_fwd(C.byref(d), _lib.ptr(gy), _lib.ptr(bpack), None, _lib.ptr(dxp),
                                          _lib.stream_ptr()))
            dx = dxp[..., :ci] if cip != ci else dxp
        if ctx.defer is not None and ctx.dwbuf is not None:
            # weight gradient + bias reduction on the side stream; .grad is assigned when the backward pass is over
            defer, idx = ctx.defer
            # the operands stay referenced until the callback has made the main stream wait for the side stream: no
            # record_stream(), whose event-polled block reuse makes the caching allocator's behaviour timing-dependent
            defer.keep.append((gy, xin))
            defer.side.wait_event(ready)          # not the input-gradient kernel just queued: the two run side by side
            with torch.cuda.stream(defer.side):
                want_b = has_bias and ctx.needs_input_grad[2]
                if ctx.needs_input_grad[1]:
                    # (the bias gradient = column sums of gy comes out of the same kernel: dball is zero-filled per step)
                    dwt, ctx.dwbuf = ctx.dwbuf, None
                    d = _desc(xin, cip, co, k)
                    _lib.check(L.stpde_conv3d_wgrad_bias(C.byref(d), _lib.ptr(xin), _lib.ptr(gy), _lib.ptr(dwt),
                                                         _lib.ptr(defer.bias_slice(idx)) if want_b else None,
                                                         _lib.stream_ptr()))
                elif want_b:
                    defer.direct_bias[idx] = gy.reshape(-1, co).sum(0)     # (a plain torch reduction: no atomics)
            defer.enqueue(idx)
            return dx, None, None, None
        if ctx.needs_input_grad[1]:
            ntap = k ** 3
            dwt, ctx.dwbuf = ctx.dwbuf, None     # zero-filled slice of the per-step buffer (used once), else a fresh one
            if dwt is None:
                dwt = _acc_zeros(ntap * co * cip, gy.device)
            d = _desc(xin, cip, co, k)
            if has_bias and ctx.needs_input_grad[2]:
                db = _acc_zeros(co, gy.device)     # column sums of gy, from the same kernel
            _lib.check(L.stpde_conv3d_wgrad_bias(C.byref(d), _lib.ptr(xin), _lib.ptr(gy), _lib.ptr(dwt), _lib.ptr(db),
                                                 _lib.stream_ptr()))
            dwt = _acc_value(dwt, ntap * co * cip).view(ntap, co, cip)     # (deterministic mode: long accumulators -> fp32)
            if db is not None:
                db = _acc_value(db, co)
            dw = dwt[:, :, :ci].permute(1, 2, 0).reshape(co, ci, k, k, k)
        if has_bias and ctx.needs_input_grad[2] and db is None:
            db = gy.reshape(-1, co).sum(0)
        return dx, dw, db, None


def _hip_conv_ok(x, conv):
    """Envelope of the HIP implicit-GEMM convolution: fp32 CUDA data, out_channels a multiple of 16, cubic 1x1x1 or
    3x3x3 kernel, stride 1, 'same' zero padding, no dilation / groups.  Anything else (e.g. unet_nf = 8, half / double
    models) takes torch's convolution on the permuted tensor -- API compatibility, never the benchmarked path."""
    w = conv.weight
    k = w.shape[2]
    return (x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and w.is_cuda
            and w.shape[0] % 16 == 0 and k in (1, 3) and tuple(w.shape[2:]) == (k, k, k)
            and tuple(conv.stride) == (1, 1, 1) and tuple(conv.padding) == ((k - 1) // 2,) * 3
            and tuple(conv.dilation) == (1, 1, 1) and conv.groups == 1 and conv.padding_mode == "zeros"
            and (conv.bias is None or conv.bias.dtype == torch.float32))


def _conv_cl(x, conv):
    """Apply an nn.Conv3d (1x1x1 or 3x3x3/pad 1, stride 1) to a channels-last tensor [B,T,Z,X,C]."""
    if _hip_conv_ok(x, conv):
        return _Conv3dHip.apply(x, conv.weight, conv.bias, getattr(conv, "_stpde_packs", None))
    y = conv(x.permute(0, 4, 1, 2, 3))
    return y.permute(0, 2, 3, 4, 1).contiguous()


class _ContiguousGrad(torch.autograd.Function):
    """Identity whose backward hands a contiguous gradient upstream: a strided grad (e.g. a loss taken on the
    [B,C,T,Z,X] view in NCHW order) would otherwise send torch's batch-norm backward down its generic, 100x slower
    reduction kernel instead of the channels-last one."""

    @staticmethod
    @_lib.guarded
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    @_lib.guarded
    def backward(ctx, g):
        return g.contiguous()


def _bn_cl(x, bn):
    """nn.BatchNorm3d semantics (batch statistics in training, running-stat update) on channels-last data."""
    shp = x.shape
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None \
            and not getattr(bn, "_stpde_counted", False):
        bn.num_batches_tracked.add_(1)
    y = F.batch_norm(x.reshape(-1, shp[-1]), bn.running_mean, bn.running_var, bn.weight, bn.bias,
                     bn.training or not bn.track_running_stats, bn.momentum, bn.eps)
    return y.reshape(shp)


class _BnActHip(torch.autograd.Function):
    """y = act(batch_norm(x) [+ residual]) on channels-last x, one HIP pass for the statistics and one for the
    normalisation (stpde_bn_fwd); backward = one reduction + one elementwise pass (stpde_bn_bwd)."""

    @staticmethod
    @_lib.guarded
    def forward(ctx, x, residual, weight, bias, running_mean, running_var, training, momentum, eps, relu, scratch=None):
        L = _lib.lib()
        x = x.contiguous()
        residual = residual.contiguous() if residual is not None else None
        c = x.shape[-1]
        d = _lib.BnDesc()
        d.N, d.C, d.training, d.relu, d.eps, d.momentum = x.numel() // c, c, int(training), int(relu), eps, momentum
        # scratch = zero-filled [BN_REP][6][C] slice of the U-Net's per-step buffer (forward + backward sums): no memsets
        ctx.scratch = scratch
        d.scratch_zeroed = int(scratch is not None)
        R = _lib.BN_REP
        ctx.det = _det()
        if ctx.det:     # deterministic mode: statistics in the double format as long accumulators (the first 4RC floats)
            d.det, d.stats_mode = 1, 1
        sums = (scratch[:4 * R * c] if scratch is not None else torch.empty(4 * R * c, device=x.device)) if training else None
        # (scratch layout, shared with _ResBlockHip: [0 : 4RC] forward sums -- this path's float format takes the first 3RC --,
        # [4RC : 6RC] backward sums)
        stat = torch.empty(2 * c, device=x.device)
        y = torch.empty_like(x)
        _lib.check(L.stpde_bn_fwd(C.byref(d), _lib.ptr(x), _lib.ptr(residual),
                                  _lib.ptr(weight.detach() if weight is not None else None),
                                  _lib.ptr(bias.detach() if bias is not None else None),
                                  _lib.ptr(running_mean), _lib.ptr(running_var), _lib.ptr(sums), _lib.ptr(stat),
                                  _lib.ptr(y), _lib.stream_ptr()))
        ctx.save_for_backward(x, y if relu else None, weight, stat)
        ctx.desc = (d.N, c, int(training), int(relu), eps, momentum)
        ctx.has_res = residual is not None
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    @_lib.guarded
    def backward(ctx, gy):
        L = _lib.lib()
        x, y, weight, stat = ctx.saved_tensors
        d = _lib.BnDesc()
        d.N, d.C, d.training, d.relu, d.eps, d.momentum = ctx.desc
        d.det = ctx.det
        c = d.C
        gy = gy.contiguous()
        need_x, need_r, need_w, need_b = ctx.needs_input_grad[:4]
        dx = torch.empty_like(x) if need_x else None
        dr = torch.empty_like(x) if (ctx.has_res and need_r) else None
        dw = torch.empty(c, device=x.device) if (weight is not None and need_w) else None
        db = torch.empty(c, device=x.device) if need_b else None
        scratch, ctx.scratch = ctx.scratch, None          # used once (a second backward gets a fresh buffer + memset)
        d.scratch_zeroed = int(scratch is not None)
        R = _lib.BN_REP
        bsum = scratch[4 * R * c:] if scratch is not None else torch.empty(2 * R * c, device=x.device)
        _lib.check(L.stpde_bn_bwd(C.byref(d), _lib.ptr(x), _lib.ptr(y), _lib.ptr(gy),
                                  _lib.ptr(weight.detach() if weight is not None else None), _lib.ptr(stat),
                                  _lib.ptr(bsum), _lib.ptr(dx), _lib.ptr(dr), _lib.ptr(dw), _lib.ptr(db),
                                  _lib.stream_ptr()))
        return dx, dr, dw, db, None, None, None, None, None, None, None


def _bn_act(x, bn, relu, residual=None):
    """act(bn(x) [+ residual]) with nn.BatchNorm3d semantics on channels-last data: the HIP kernels on CUDA tensors,
    torch ops otherwise (CPU, exotic BatchNorm configurations)."""
    c = x.shape[-1]
    training = bn.training or not bn.track_running_stats
    hip = (x.is_cuda and x.dtype == torch.float32 and 16 <= c <= 512 and (c & (c - 1)) == 0
           and bn.momentum is not None and (training or bn.running_mean is not None)
           and (bn.weight is None) == (bn.bias is None))
    if not hip:
        h = _bn_cl(x, bn)
        if residual is not None:
            h = h + residual
        return F.relu(h) if relu else h
    if training and x.numel() // c == 1:   # same refusal as torch.nn.functional.batch_norm
        raise ValueError("Expected more than 1 value per channel when training, got input size {}".format(tuple(x.shape)))
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None \
            and not getattr(bn, "_stpde_counted", False):
        bn.num_batches_tracked.add_(1)
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    scratch = getattr(bn, "_stpde_scratch", None)
    if scratch is not None:
        bn._stpde_scratch = None              # one use per step
    return _BnActHip.apply(x, residual, bn.weight, bn.bias, rm, rv, training, float(bn.momentum), float(bn.eps),
                           bool(relu), scratch)


class _ResampleHip(torch.autograd.Function):
    """Max pooling (pool=True) or nearest up-sampling by integer factors on channels-last data (stpde_resample3d)."""

    @staticmethod
    @_lib.guarded
    def forward(ctx, x, factors, pool):
        L = _lib.lib()
        x = x.contiguous()
        ft, fz, fx = (int(f) for f in factors)
        b, t, z, xx, c = x.shape
        d = _lib.ResampleDesc()
        if pool:
            d.B, d.T, d.Z, d.X = b, t // ft, z // fz, xx // fx
        else:
            d.B, d.T, d.Z, d.X = b, t, z, xx
        d.C, d.ft, d.fz, d.fx = c, ft, fz, fx
        shape = (b, d.T, d.Z, d.X, c) if pool else (b, t * ft, z * fz, xx * fx, c)
        y = torch.empty(shape, device=x.device, dtype=torch.float32)
        _lib.check(L.stpde_resample3d(C.byref(d), 0 if pool else 2, _lib.ptr(x), None, _lib.ptr(y), _lib.stream_ptr()))
        ctx.desc, ctx.pool, ctx.in_shape = (d.B, d.T, d.Z, d.X, c, ft, fz, fx), pool, x.shape
        ctx.save_for_backward(x if pool else None)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    @_lib.guarded
    def backward(ctx, gy):
        L = _lib.lib()
        (x,) = ctx.saved_tensors
        d = _lib.ResampleDesc()
        d.B, d.T, d.Z, d.X, d.C, d.ft, d.fz, d.fx = ctx.desc
        gy = gy.contiguous()
        gx = torch.empty(ctx.in_shape, device=gy.device, dtype=torch.float32)
        _lib.check(L.stpde_resample3d(C.byref(d), 1 if ctx.pool else 3, _lib.ptr(gy), _lib.ptr(x), _lib.ptr(gx),
                                      _lib.stream_ptr()))
        return gx, None, None


def _hip_resample_ok(x, factors):
    return (x.is_cuda and x.dtype == torch.float32 and x.shape[-1] % 4 == 0 and all(1 <= int(f) <= 4 for f in factors))


def _pool_cl(x, kernel):
    if all(int(k) == 1 for k in kernel):
        return x
    if _hip_resample_ok(x, kernel) and all(x.shape[1 + i] % int(k) == 0 for i, k in enumerate(kernel)):
        return _ResampleHip.apply(x, tuple(kernel), True)
    y = F.max_pool3d(x.permute(0, 4, 1, 2, 3), tuple(int(k) for k in kernel))
    return y.permute(0, 2, 3, 4, 1).contiguous()


def _upsample_cl(x, factors):
    if all(int(f) == 1 for f in factors):
        return x
    if _hip_resample_ok(x, factors):
        return _ResampleHip.apply(x, tuple(factors), False)
    for dim, f in zip((1, 2, 3), factors):
        if int(f) != 1:
            x = x.repeat_interleave(int(f), dim=dim)
    return x


stats = {"fused_resblocks": 0}     # calls that took the fused residual-block path (tests assert it)


def _fused_ok(blk, x):
    """Envelope of the fused residual block (_ResBlockHip): every convolution on the HIP path, every BatchNorm normalising
    with batch statistics.  STPDE_FUSED_RESBLOCK=0 keeps the one-kernel-per-layer path (the A/B switch of the tests)."""
    if os.environ.get("STPDE_FUSED_RESBLOCK", "1") == "0":
        return False
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 5 and x.numel() // x.shape[-1] >= 2):
        return False
    # limits of stpde_conv3d_fused itself: fewer than 2^27 voxels (int32 voxel arithmetic of the fold-ins), all
    # parameters on the input's device -- outside them the layer-wise path runs instead of an error from inside the node
    if x.numel() // x.shape[-1] >= (1 << 27):
        return False
    mods = (blk.conv1, blk.conv2, blk.conv3, blk.shortcut, blk.bn1, blk.bn2, blk.bn3)
    if any(p.device != x.device for m in mods for p in m.parameters(recurse=False)):
        return False
    if not all(_hip_conv_ok(x, c) for c in (blk.conv1, blk.conv2, blk.conv3, blk.shortcut)):
        return False
    if blk.conv1.weight.shape[2] != 1 or blk.conv2.weight.shape[2] != 3 or blk.conv3.weight.shape[2] != 1 \
            or blk.shortcut.weight.shape[2] != 1:
        return False
    for bn in (blk.bn1, blk.bn2, blk.bn3):
        c = bn.num_features
        if not ((bn.training or not bn.track_running_stats) and bn.momentum is not None and 16 <= c <= 512
                and (c & (c - 1)) == 0 and (bn.weight is None) == (bn.bias is None)
                and (bn.weight is None or bn.weight.dtype == torch.float32)):
            return False
    return True


def _ptr(t):
    return _lib.ptr(t) if t is not None else None


class _ResBlockHip(torch.autograd.Function):
    """One ResBlock3D (reference src/unet3d.py:39-56) in training mode as ONE autograd node over the fused kernels of
    csrc/conv3d_fused.hip (round 4):

      forward   conv1 + shortcut in one pass over x (+ bn1 statistics) | bn1 + relu | conv2 (+ bn2 statistics) |
                conv3 with bn2 + relu applied to its operand on load (+ bn3 statistics) | bn3 + shortcut + relu
      backward  bn3 backward | conv3 input gradient (+ relu mask + bn2 backward sums) | bn2 elementwise |
                conv2 input gradient (+ mask + bn1 sums) | bn1 elementwise | conv1 and shortcut input gradients as one
                convolution of two inputs | the four weight (+ bias) gradients (side stream when the U-Net defers them)

    5 + 7 + 4 launches and 11 + 23 tensor passes instead of 10 + 17 + 4 and 18 + 32; the normalised output of bn2 is never
    stored.  Saved for backward: x, the three raw convolution outputs, relu(bn1) and the block output."""

    @staticmethod
    @_lib.guarded
    def forward(ctx, blk, x, w1, b1, w2, b2, w3, b3, ws, bs, g1, be1, g2, be2, g3, be3):
        L = _lib.lib()
        dev = x.device
        ci, cn, co = x.shape[-1], w1.shape[0], w3.shape[0]
        cip = (ci + 15) // 16 * 16
        xin = x.detach()
        if cip != ci:
            xin = F.pad(xin, (0, cip - ci))
        xin = xin.contiguous()
        shp = tuple(xin.shape[:4])
        N = shp[0] * shp[1] * shp[2] * shp[3]
        R = _lib.BN_REP
        convs = (blk.conv1, blk.conv2, blk.conv3, blk.shortcut)
        packs = []
        for conv, w in zip(convs, (w1, w2, w3, ws)):
            p = getattr(conv, "_stpde_packs", None)
            if p is None:
                fidx, bidx, _, _ = _pack_indices(w.shape[0], w.shape[1], w.shape[2], dev)
                wflat = torch.cat([w.detach().reshape(-1), w.new_zeros(1)])
                p = (wflat[fidx], wflat[bidx], None)
            packs.append(p)
        bns = (blk.bn1, blk.bn2, blk.bn3)
        fsum, bsum, stat = [], [], []
        for bn in bns:
            c = bn.num_features
            sc = getattr(bn, "_stpde_scratch", None)
            if sc is not None:
                bn._stpde_scratch = None              # one use per step
            else:
                sc = torch.zeros(6 * R * c, device=dev)
            fsum.append(sc[:4 * R * c])
            bsum.append(sc[4 * R * c:])
            stat.append(torch.empty(2 * c, device=dev))
        rm = [bn.running_mean if bn.track_running_stats else None for bn in bns]
        rv = [bn.running_var if bn.track_running_stats else None for bn in bns]

        def conv_args(ci_, co_, k):
            a = _lib.Conv3dFusedArgs()
            a.d.B, a.d.T, a.d.Z, a.d.X = shp
            a.d.Ci, a.d.Co, a.d.ksize = ci_, co_, k
            a.d.det = _det()
            return a

        def bn_desc(c, relu, bn):
            d = _lib.BnDesc()
            d.N, d.C, d.training, d.relu, d.eps, d.momentum = N, c, 1, int(relu), float(bn.eps), float(bn.momentum)
            d.scratch_zeroed, d.stats_mode = 1, 2
            d.det = _det()
            return d

        st = _lib.stream_ptr()
        new = lambda c: torch.empty(shp + (c,), device=dev, dtype=torch.float32)
        # conv1 + shortcut: one pass over x, bn1's statistics from conv1's accumulator tiles
        y1, sc_out = new(cn), new(co)
        a = conv_args(cip, cn, 1)
        a.x, a.w_pack, a.bias, a.y = _ptr(xin), _ptr(packs[0][0]), _ptr(b1.detach() if b1 is not None else None), _ptr(y1)
        a.y2, a.wo2_pack, a.bias2, a.Co2 = _ptr(sc_out), _ptr(packs[3][0]), _ptr(bs.detach() if bs is not None else None), co
        a.out_sums = _ptr(fsum[0])
        _lib.check(L.stpde_conv3d_fused(C.byref(a), None, st))
        # bn1 + relu (elementwise pass only)
        h1 = new(cn)
        d = bn_desc(cn, True, bns[0])
        _lib.check(L.stpde_bn_fwd(C.byref(d), _ptr(y1), None, _ptr(g1.detach() if g1 is not None else None),
                                  _ptr(be1.detach() if be1 is not None else None), _ptr(rm[0]), _ptr(rv[0]),
                                  _ptr(fsum[0]), _ptr(stat[0]), _ptr(h1), st))
        # conv2 (+ bn2 statistics)
        y2 = new(cn)
        a = conv_args(cn, cn, 3)
        a.x, a.w_pack, a.bias, a.y = _ptr(h1), _ptr(packs[1][0]), _ptr(b2.detach() if b2 is not None else None), _ptr(y2)
        a.out_sums = _ptr(fsum[1])
        _lib.check(L.stpde_conv3d_fused(C.byref(a), None, st))
        # conv3 of relu(bn2(y2)) applied on load (+ bn3 statistics)
        y3 = new(co)
        a = conv_args(cn, co, 1)
        a.x, a.w_pack, a.bias, a.y = _ptr(y2), _ptr(packs[2][0]), _ptr(b3.detach() if b3 is not None else None), _ptr(y3)
        a.in_sums, a.in_gamma, a.in_beta = _ptr(fsum[1]), _ptr(g2.detach() if g2 is not None else None), \
            _ptr(be2.detach() if be2 is not None else None)
        a.in_running_mean, a.in_running_var, a.in_stat = _ptr(rm[1]), _ptr(rv[1]), _ptr(stat[1])
        a.in_eps, a.in_momentum = float(bns[1].eps), float(bns[1].momentum)
        a.out_sums = _ptr(fsum[2])
        _lib.check(L.stpde_conv3d_fused(C.byref(a), None, st))
        # bn3 + shortcut (+ relu)
        out = new(co)
        d = bn_desc(co, blk.final_relu, bns[2])
        _lib.check(L.stpde_bn_fwd(C.byref(d), _ptr(y3), _ptr(sc_out), _ptr(g3.detach() if g3 is not None else None),
                                  _ptr(be3.detach() if be3 is not None else None), _ptr(rm[2]), _ptr(rv[2]),
                                  _ptr(fsum[2]), _ptr(stat[2]), _ptr(out), st))
        ctx.save_for_backward(xin, y1, h1, y2, y3, out if blk.final_relu else None, stat[0], stat[1], stat[2],
                              g1, be1, g2, be2, g3, packs[0][1], packs[1][1], packs[2][1], packs[3][1])
        ctx.meta = (shp, N, ci, cip, cn, co, bool(blk.final_relu), [float(bn.eps) for bn in bns],
                    [float(bn.momentum) for bn in bns], (b1 is not None, b2 is not None, b3 is not None, bs is not None))
        ctx.bsum = bsum
        ctx.dw = [(p[2], p[3:5] if len(p) > 3 else None) for p in packs]
        stats["fused_resblocks"] += 1
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    @_lib.guarded
    def backward(ctx, gout):
        L = _lib.lib()
        (xin, y1, h1, y2, y3, out, st1, st2, st3, g1, be1, g2, be2, g3, bp1, bp2, bp3, bps) = ctx.saved_tensors
        shp, N, ci, cip, cn, co, final_relu, eps, mom, has_b = ctx.meta
        dev = gout.device
        R = _lib.BN_REP
        gout = gout.contiguous()
        bsum, ctx.bsum = ctx.bsum, None               # zero-filled, used once (a second backward gets fresh buffers)
        zeroed = bsum is not None
        if bsum is None:
            bsum = [torch.empty(2 * R * c, device=dev) for c in (cn, cn, co)]
        st = _lib.stream_ptr()
        new = lambda c: torch.empty(shp + (c,), device=dev, dtype=torch.float32)
        vec = lambda t, c: torch.empty(c, device=dev) if t is not None else None

        def conv_args(ci_, co_, k):
            a = _lib.Conv3dFusedArgs()
            a.d.B, a.d.T, a.d.Z, a.d.X = shp
            a.d.Ci, a.d.Co, a.d.ksize = ci_, co_, k
            a.d.det = _det()
            return a

        def bn_desc(c, relu, k, reduce_done):
            d = _lib.BnDesc()
            d.N, d.C, d.training, d.relu, d.eps, d.momentum = N, c, 1, int(relu), eps[k], mom[k]
            d.scratch_zeroed, d.reduce_done = int(zeroed), int(reduce_done)
            d.det = _det()
            return d

        defer = ctx.dw[0][1][0] if ctx.dw[0][1] else None
        # bn3 backward (reduction + elementwise): gradient of conv3's output and of the shortcut branch
        dy3, dsc = new(co), new(co)
        dg3, db3 = vec(g3, co), vec(g3, co)
        d = bn_desc(co, final_relu, 2, False)
        _lib.check(L.stpde_bn_bwd(C.byref(d), _ptr(y3), _ptr(out), _ptr(gout), _ptr(g3), _ptr(st3), _ptr(bsum[2]),
                                  _ptr(dy3), _ptr(dsc), _ptr(dg3), _ptr(db3), st))
        ev3 = torch.cuda.current_stream().record_event() if defer is not None else None
        # conv3 input gradient; its epilogue applies the relu mask of bn2's output and adds bn2's backward sums
        dz2 = new(cn)
        a = conv_args(co, cn, 1)
        a.x, a.w_pack, a.y = _ptr(dy3), _ptr(bp3), _ptr(dz2)
        a.m, a.m_stat, a.m_gamma, a.m_beta, a.m_bsum = _ptr(y2), _ptr(st2), _ptr(g2), _ptr(be2), _ptr(bsum[1])
        done = C.c_int(1)
        if not zeroed:
            bsum[1].zero_()
            bsum[0].zero_()
        _lib.check(L.stpde_conv3d_fused(C.byref(a), C.byref(done), st))
        dy2 = new(cn)
        dg2, db2 = vec(g2, cn), vec(g2, cn)
        d = bn_desc(cn, False, 1, True)
        _lib.check(L.stpde_bn_bwd(C.byref(d), _ptr(y2), None, _ptr(dz2), _ptr(g2), _ptr(st2), _ptr(bsum[1]),
                                  _ptr(dy2), None, _ptr(dg2), _ptr(db2), st))
        ev2 = torch.cuda.current_stream().record_event() if defer is not None else None
        # conv2 input gradient (+ mask of bn1's output + bn1's sums where the kernel can: not the tap-split deep levels)
        dz1 = new(cn)
        a = conv_args(cn, cn, 3)
        a.x, a.w_pack, a.y = _ptr(dy2), _ptr(bp2), _ptr(dz1)
        a.m, a.m_stat, a.m_gamma, a.m_beta, a.m_bsum = _ptr(y1), _ptr(st1), _ptr(g1), _ptr(be1), _ptr(bsum[0])
        done = C.c_int(1)
        _lib.check(L.stpde_conv3d_fused(C.byref(a), C.byref(done), st))
        dy1 = new(cn)
        dg1, db1 = vec(g1, cn), vec(g1, cn)
        if done.value:
            d = bn_desc(cn, False, 0, True)
            _lib.check(L.stpde_bn_bwd(C.byref(d), _ptr(y1), None, _ptr(dz1), _ptr(g1), _ptr(st1), _ptr(bsum[0]),
                                      _ptr(dy1), None, _ptr(dg1), _ptr(db1), st))
        else:
            d = bn_desc(cn, True, 0, False)
            d.scratch_zeroed = 1                                      # zero either way (see above)
            _lib.check(L.stpde_bn_bwd(C.byref(d), _ptr(y1), _ptr(h1), _ptr(dz1), _ptr(g1), _ptr(st1), _ptr(bsum[0]),
                                      _ptr(dy1), None, _ptr(dg1), _ptr(db1), st))
        ev1 = torch.cuda.current_stream().record_event() if defer is not None else None
        # input gradient: conv1's and the shortcut's in one convolution of two inputs
        dx = None
        if ctx.needs_input_grad[1]:
            dxp = new(cip)
            a = conv_args(cn, cip, 1)
            a.x, a.w_pack, a.y = _ptr(dy1), _ptr(bp1), _ptr(dxp)
            a.x2, a.w2_pack, a.Ci2 = _ptr(dsc), _ptr(bps), co
            _lib.check(L.stpde_conv3d_fused(C.byref(a), None, st))
            dx = dxp[..., :ci] if cip != ci else dxp
        # weight (+ bias) gradients: (conv, input, output gradient, kernel size, channels, on-load transform of the input)
        jobs = [(2, y2, dy3, 1, cn, co, ev3, True), (3, xin, dsc, 1, cip, co, ev3, False),
                (1, h1, dy2, 3, cn, cn, ev2, False), (0, xin, dy1, 1, cip, cn, ev1, False)]
        # per convolution: its weight gradient (inputs 2, 4, 6, 8) or its bias gradient (3, 5, 7, 9) is wanted -- a block with
        # frozen weights and trainable biases still gets its bias gradients (the kernel produces both from one pass)
        need = [ctx.needs_input_grad[2 + 2 * k] or (has_b[k] and ctx.needs_input_grad[3 + 2 * k]) for k in range(4)]
        gw = [None] * 4
        gb = [None] * 4
        for k, xi, gy, ks, ci_, co_, ev, onload in jobs:
            if not need[k]:
                continue
            dwt, dfr = ctx.dw[k]
            ctx.dw[k] = (None, dfr)                     # the zero-filled slice of the per-step buffer is used once
            cd = _lib.Conv3dDesc()
            cd.B, cd.T, cd.Z, cd.X = shp
            cd.Ci, cd.Co, cd.ksize = ci_, co_, ks
            cd.det = _det()

            def launch(dwt, dbt):
                if onload:
                    _lib.check(L.stpde_conv3d_wgrad_onload(C.byref(cd), _ptr(xi), _ptr(gy), _ptr(dwt), _ptr(dbt), _ptr(st2),
                                                           _ptr(g2), _ptr(be2), _lib.stream_ptr()))
                else:
                    _lib.check(L.stpde_conv3d_wgrad_bias(C.byref(cd), _ptr(xi), _ptr(gy), _ptr(dwt), _ptr(dbt),
                                                         _lib.stream_ptr()))

            if dfr and dwt is not None:
                dobj, idx = dfr
                dobj.keep.append((gy, xi, st2, g2, be2))
                dobj.side.wait_event(ev)
                with torch.cuda.stream(dobj.side):
                    launch(dwt, dobj.bias_slice(idx) if has_b[k] else None)
                dobj.enqueue(idx)
            else:
                if dwt is None:
                    dwt = _acc_zeros(ks ** 3 * co_ * ci_, dev)
                dbt = _acc_zeros(co_, dev) if has_b[k] else None
                launch(dwt, dbt)
                dwt = _acc_value(dwt, ks ** 3 * co_ * ci_).view(ks ** 3, co_, ci_)     # (deterministic mode: -> fp32)
                if dbt is not None:
                    dbt = _acc_value(dbt, co_)
                cin = ci if k in (0, 3) else ci_
                if ctx.needs_input_grad[2 + 2 * k]:
                    gw[k] = dwt[:, :, :cin].permute(1, 2, 0).reshape(co_, cin, ks, ks, ks)
                gb[k] = dbt if ctx.needs_input_grad[3 + 2 * k] else None
        if _ResBlockHip.debug is not None:
            _ResBlockHip.debug.update(dy3=dy3, dsc=dsc, dz2=dz2, dy2=dy2, dz1=dz1, dy1=dy1, dx=dx, y1=y1, h1=h1, y2=y2, y3=y3,
                                      done=done.value)
        return (None, dx, gw[0], gb[0], gw[1], gb[1], gw[2], gb[2], gw[3], gb[3], dg1, db1, dg2, db2, dg3, db3)

    debug = None     # a dict here makes backward leave its intermediate gradients in it (tools/micro/dbg_resblock.py)


class ResBlock3D(nn.Module):
    """3D convolutional residue block, keeps the resolution (reference :12-56)."""

    def __init__(self, in_channels, neck_channels, out_channels, final_relu=True):
        super().__init__()
        self.in_channels = in_channels
        self.neck_channels = neck_channels
        self.out_channels = out_channels
        self.conv1 = nn.Conv3d(in_channels, neck_channels, kernel_size=1, stride=1)
        self.conv2 = nn.Conv3d(neck_channels, neck_channels, kernel_size=3, stride=1, padding=1)
        self.conv3 = nn.Conv3d(neck_channels, out_channels, kernel_size=1, stride=1)
        self.bn1 = nn.BatchNorm3d(num_features=neck_channels)
        self.bn2 = nn.BatchNorm3d(num_features=neck_channels)
        self.bn3 = nn.BatchNorm3d(num_features=out_channels)
        self.shortcut = nn.Conv3d(in_channels, out_channels, kernel_size=1, stride=1)
        self.final_relu = final_relu

    def forward_cl(self, x):
        """Channels-last in, channels-last out: conv1-bn1-relu-conv2-bn2-relu-conv3-bn3 + shortcut (+relu)."""
        if _fused_ok(self, x):
            for bn in (self.bn1, self.bn2, self.bn3):
                if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None \
                        and not getattr(bn, "_stpde_counted", False):
                    bn.num_batches_tracked.add_(1)
            return _ResBlockHip.apply(self, x, self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias,
                                      self.conv3.weight, self.conv3.bias, self.shortcut.weight, self.shortcut.bias,
                                      self.bn1.weight, self.bn1.bias, self.bn2.weight, self.bn2.bias,
                                      self.bn3.weight, self.bn3.bias)
        h = _bn_act(_conv_cl(x, self.conv1), self.bn1, True)
        h = _bn_act(_conv_cl(h, self.conv2), self.bn2, True)
        # bn3 + shortcut + final ReLU in one pass
        return _bn_act(_conv_cl(h, self.conv3), self.bn3, self.final_relu, residual=_conv_cl(x, self.shortcut))

    def forward(self, x):  # [B, C, T, Z, X] -> [B, C', T, Z, X] (channels-last view)
        h = self.forward_cl(x.permute(0, 2, 3, 4, 1).contiguous())
        if h.is_cuda and h.requires_grad:
            h = _ContiguousGrad.apply(h)
        return h.permute(0, 4, 1, 2, 3)


class UNet3d(nn.Module):  # pylint: disable=too-many-instance-attributes
    """3D U-Net with residual blocks (reference :59-240)."""

    def __init__(self, in_features=4, out_features=32, igres=(4, 32, 32), ogres=None, nf=16, mf=512):
        super().__init__()
        self.igres = igres
        self.nf = nf
        self.mf = mf
        self.in_features = in_features
        self.out_features = out_features
        self.ogres = self.igres if ogres is None else ogres
        if isinstance(self.igres, int):
            self.igres = tuple([self.igres] * 3)
        if isinstance(self.ogres, int):
            self.ogres = tuple([self.ogres] * 3)
        self._check_grid_res()
        fac = np.log2(np.array(self.ogres) / np.array(self.igres))
        if not np.allclose(fac % 1, 0):
            raise ValueError("ogres must be 2^k times greater than igres where k >= 0. "
                             "Instead igres: {}, ogres: {}".format(igres, ogres))
        if not np.all(fac >= 0):
            raise ValueError("ogres must be greater or equal to igres. "
                             "Instead igres: {}, ogres: {}".format(igres, ogres))
        self.exp_fac = fac.astype(np.int32)
        self.expand = bool(np.any(self.exp_fac != 0))
        # weight / bias gradients of the convolutions on a side stream, .grad assigned when loss.backward() is over (see
        # _DeferredGrads); off by default: only .backward() sees these gradients, torch.autograd.grad(...) does not
        self.deferred_weight_grads = False
        self.li = int(round(math.log2(max(self.igres))))   # number of input levels
        self.lo = int(round(math.log2(max(self.ogres))))   # number of output levels
        self._create_layers()

    def _check_grid_res(self):
        if not (hasattr(self.igres, '__len__') and hasattr(self.ogres, '__len__')):
            raise TypeError('igres and ogres must be tuples for grid dimensions')
        if not (len(self.igres) == 3 and len(self.ogres) == 3):
            raise ValueError('igres and ogres must have len = 3, however detected to be'
                             '{} and {}'.format(len(self.igres), len(self.ogres)))
        for d in list(self.igres) + list(self.ogres):
            if not (np.issubdtype(type(d), np.integer) and d > 0 and (int(d) & (int(d) - 1)) == 0):
                raise ValueError('dimensions in igres and ogres must be  integer powers of 2.'
                                 'instead they are {} and {}.'.format(self.igres, self.ogres))

    @staticmethod
    def _get_pool_kernel_size(prev_layer_dims):
        """Halve every dimension that is not already the smallest one (all of them once they are equal)."""
        dims = [int(d) for d in prev_layer_dims]
        lo = min(dims)
        kernel = [2, 2, 2] if all(d == lo for d in dims) else [1 if d == lo else 2 for d in dims]
        return kernel, np.array([d // k for d, k in zip(dims, kernel)])

    @staticmethod
    def _get_exp_kernel_size(prev_exp_fac):
        next_exp_fac = np.clip(prev_exp_fac - 1, 0, None)
        return prev_exp_fac - next_exp_fac + 1, next_exp_fac

    def _create_layers(self):
        down_out = [min(self.nf * (2 ** (i + 1)), self.mf) for i in range(self.li)]
        down_in = [self.nf] + down_out[:-1]
        up_in = [int(n * 2) for n in down_in[::-1][:-1]]
        up_out = down_in[::-1][1:]
        self.conv_in = ResBlock3D(self.in_features, self.nf, self.nf)
        self.conv_out = ResBlock3D(down_in[0] * 2, down_in[0] * 2, self.out_features, final_relu=False)
        self.conv_mid = ResBlock3D(down_out[-1], down_out[-2], down_out[-2])
        down_modules = [ResBlock3D(n_in, int(n / 2), n) for n_in, n in zip(down_in, down_out)]
        up_modules = [ResBlock3D(n_in, n, n) for n_in, n in zip(up_in, up_out)]
        down_pools, up_interps = [], []
        self._pool_kernels = []
        dims = np.array(self.igres)
        for _ in range(len(down_out)):
            kernel, dims = self._get_pool_kernel_size(dims)
            self._pool_kernels.append(kernel)
            down_pools.append(nn.MaxPool3d(kernel))
            up_interps.insert(0, nn.Upsample(scale_factor=tuple(kernel)))   # mirrored order on the way up
        self._up_factors = self._pool_kernels[::-1]
        if self.expand:
            n_exp = int(np.max(self.exp_fac))
            self.exp_modules = nn.ModuleList([ResBlock3D(2 * self.nf, 2 * self.nf, 2 * self.nf) for _ in range(n_exp)])
            interps, self._exp_factors = [], []
            fac = self.exp_fac
            for _ in range(n_exp):
                kernel, fac = self._get_exp_kernel_size(fac)
                self._exp_factors.append([int(k) for k in kernel])
                interps.append(nn.Upsample(scale_factor=tuple(kernel)))
            self.exp_fac = fac
            self.exp_interps = nn.ModuleList(interps)
        self.down_modules = nn.ModuleList(down_modules)
        self.up_modules = nn.ModuleList(up_modules)
        self.down_pools = nn.ModuleList(down_pools)
        self.up_interps = nn.ModuleList(up_interps)

    def _prepare_step(self, device):
        """CUDA path: pack the weights of ALL convolutions with one concatenation + one index gather (instead of two
        small kernels per convolution per pass) and bump all BatchNorm step counters with one foreach op."""
        convs = [m for m in self.modules() if isinstance(m, nn.Conv3d) and m.weight.shape[0] % 16 == 0
                 and m.weight.dtype == torch.float32 and m.weight.shape[2] in (1, 3)]
        plan = getattr(self, "_pack_plan", None)
        if plan is None or plan[0] != str(device):
            total = sum(c.weight.numel() for c in convs)
            chunks, spans, off, pos = [], [], 0, 0
            for c in convs:
                co, ci, k = c.weight.shape[0], c.weight.shape[1], c.weight.shape[2]
                fidx, bidx, _, _ = _pack_indices(co, ci, k, device)
                n = c.weight.numel()
                for idx in (fidx, bidx):
                    chunks.append(torch.where(idx == n, torch.full_like(idx, total), idx + off))
                spans.append((pos, pos + fidx.numel(), pos + fidx.numel() + bidx.numel()))
                pos += fidx.numel() + bidx.numel()
                off += n
            plan = (str(device), torch.cat(chunks), spans)
            self._pack_plan = plan
        theta = torch.cat([c.weight.detach().reshape(-1) for c in convs] + [torch.zeros(1, device=device)])
        packs = theta[plan[1]]
        # one zero-filled buffer for all weight gradients of this step (the kernels accumulate with atomics)
        need_dw = torch.is_grad_enabled() and any(c.weight.requires_grad for c in convs)
        sizes = [c.weight.shape[2] ** 3 * c.weight.shape[0] * ((c.weight.shape[1] + 15) // 16 * 16) for c in convs]
        dwall = _acc_zeros(sum(sizes), device) if need_dw else None
        aw = 2 * _lib.DET_K if _lib.deterministic else 1          # floats of storage per accumulated element
        defer = None
        if need_dw and self.deferred_weight_grads:
            defer = _DeferredGrads(convs, dwall, sizes, device)
            if len(plan) < 4:
                plan = plan + (_DeferredGrads.unpack_index(convs, sizes, device),)
                self._pack_plan = plan
            defer.uidx = plan[3]
        o = 0
        for i, (c, (a, b, e), n) in enumerate(zip(convs, plan[2], sizes)):
            co, ci, k = c.weight.shape[0], c.weight.shape[1], c.weight.shape[2]
            dw = dwall[aw * o:aw * (o + n)] if need_dw else None      # (flat: the kernels index [tap][co][ci padded] themselves)
            c._stpde_packs = (packs[a:b], packs[b:e], dw) + ((defer, i) if defer is not None else ())
            o += n
        bns = []
        if self.training:
            bns = [m for m in self.modules() if isinstance(m, nn.BatchNorm3d) and m.track_running_stats
                   and m.num_batches_tracked is not None]
            if bns:
                torch._foreach_add_([m.num_batches_tracked for m in bns], 1)
                for m in bns:
                    m._stpde_counted = True
            # one zero-filled buffer for the statistics scratch of every BatchNorm of the step (forward + backward sums):
            # one memset instead of two per BatchNorm call
            allbn = [m for m in self.modules() if isinstance(m, nn.BatchNorm3d)]
            per = 6 * _lib.BN_REP
            zero = torch.zeros(per * sum(m.num_features for m in allbn), device=device)
            o = 0
            for m in allbn:
                m._stpde_scratch = zero[o:o + per * m.num_features]
                o += per * m.num_features
        return convs, bns

    def forward(self, x):
        """x [batch, in_features, *igres] -> [batch, out_features, *ogres] (channels-last strides; reference :208-240)."""
        if not x.is_cuda:
            return self._forward_impl(x)
        convs, bns = self._prepare_step(x.device)
        try:
            return self._forward_impl(x)
        finally:
            for c in convs:
                c._stpde_packs = None
            for m in bns:
                m._stpde_counted = False
            for m in self.modules():
                if isinstance(m, nn.BatchNorm3d):
                    m._stpde_scratch = None

    def _forward_impl(self, x):
        h = self.conv_in.forward_cl(x.permute(0, 2, 3, 4, 1).contiguous())
        skips = [h]
        for mod, kernel in zip(self.down_modules, self._pool_kernels):
            h = _pool_cl(mod.forward_cl(skips[-1]), kernel)
            skips.append(h)
        h = skips.pop(-1)
        h = self.conv_mid.forward_cl(_upsample_cl(h, self._up_factors[0]))
        for mod, factors in zip(self.up_modules, self._up_factors[1:]):
            h = torch.cat([h, skips.pop(-1)], dim=-1)
            h = _upsample_cl(mod.forward_cl(h), factors)
        h = torch.cat([h, skips.pop(-1)], dim=-1)
        if self.expand:
            for mod, factors in zip(self.exp_modules, self._exp_factors):
                h = _upsample_cl(mod.forward_cl(h), factors)
        h = self.conv_out.forward_cl(h)
        if h.is_cuda and h.requires_grad:
            h = _ContiguousGrad.apply(h)
        return h.permute(0, 4, 1, 2, 3)
