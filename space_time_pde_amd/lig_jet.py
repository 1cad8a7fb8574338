"""Host driver of the fused LIG + IM-NET jet path (dim = 3) on libstpde_hip.

Computes y = query_local_implicit_grid(imnet, latent, pts) together with its first and selected second
derivatives w.r.t. the query coordinates ("jets") in one forward pass of HIP kernels, and the gradients w.r.t.
the IM-NET parameters and the latent grid in one backward pass.  This replaces, for the hot path,
  * src/local_implicit_grid.py:47-59 + src/regular_nd_grid_interpolation.py:48-76 (gather, weights, rel coords),
  * src/implicit_net.py:48-54 (6 addmm + activations + concats on the [b*p*8, 35] matrix),
  * the 23-25 ``torch.autograd.grad(create_graph=True)`` sweeps of src/pde.py:8-9 and their double backward.

Everything here is plumbing (buffer allocation, weight packing by index gather, launch order); all arithmetic on
query points runs in the HIP library.  There is no fallback: tensors must be CUDA/HIP float32.
"""
import ctypes as C
import os
import threading

import numpy as np
import torch

from . import _lib
from ._lib import XT, GatherDesc, ImNetPlanDesc, JetCfg, LayerDesc, LigWorkspace, XbarDesc, check, ptr, stream_ptr

_FRAG = 256  # floats per 16x16 fragment block
# widest latent the HIP jet path takes: the augmented input [r(3); latent(c); 1] must fit XT = 3 fragment tiles whose third
# one is sparse (4 live slots): 3 + c + 1 <= 36.  Wider latents run the generic composed formulation.
MAX_LATENT_CHANNELS = 16 * (XT - 1) + 4 - 3 - 1

# Optional per-kernel timing (bench.py): set ``profile`` to a dict; every library call then records a pair of
# events on the launch stream under its kernel name.  None = no overhead.
profile = None


class _timed:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if profile is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if profile is not None:
            self.e1.record()
            profile.setdefault(self.name, []).append((self.e0, self.e1))
        return False


# ------------------------------------------------------------------------------------------------------------
# architecture plan: index maps between nn.Linear parameters and MFMA operand packs
# ------------------------------------------------------------------------------------------------------------
class ImNetPlan:
    """Static layout information for one IM-NET architecture (src/implicit_net.py:31-36)."""

    _cache = {}

    @classmethod
    def get(cls, dim, in_features, out_features, nf):
        key = (dim, in_features, out_features, nf)
        if key not in cls._cache:
            cls._cache[key] = cls(*key)
        return cls._cache[key]

    def __init__(self, dim, in_features, out_features, nf):
        if dim != 3:
            raise ValueError("the HIP jet path is built for dim=3 query points")
        if nf % 16 != 0:
            raise ValueError("the HIP jet path needs nf to be a multiple of 16 (hidden widths are MFMA tiles)")
        self.dz = dim + in_features
        if self.dz + 1 > 16 * (XT - 1) + 4:
            raise ValueError("in_features too large for the augmented input of the HIP jet path (max 32 latent channels)")
        if out_features > 16:
            raise ValueError("out_features > 16 not supported by the HIP jet path")
        self.dim, self.cin, self.cout, self.nf = dim, in_features, out_features, nf
        self.xl = (in_features + 15) // 16      # 16-channel tiles of the latent adjoint (k_xbar<XL>)
        # slot of augmented-input feature f (r, latent channels, ones column) in the XT fragment tiles: tiles 0 / 1 in
        # order, tile 2 sparse -- features 32.. sit in register 0 of its fragment (slots 32, 36, 40, 44), so the GEMMs
        # over the augmented input take 9 k-steps instead of 12 (x_live() in csrc/common.h, k_gather)
        f = np.arange(self.dz + 1)
        self.slot = np.where(f < 16 * (XT - 1), f, 16 * (XT - 1) + 4 * (f - 16 * (XT - 1)))
        widths = [16 * nf, 8 * nf, 4 * nf, 2 * nf, nf, out_features]
        self.layers = []
        theta_off = 0
        for l in range(6):
            kh = 0 if l == 0 else widths[l - 1]
            skip = l < 5
            kin = kh + (self.dz if skip else 0)
            m = widths[l]
            lay = dict(M=m, Kh=kh, KT=kh // 16, MT=(m + 15) // 16, skip=skip, Kin=kin, w_off=theta_off,
                       b_off=theta_off + m * kin)
            theta_off += m * kin + m
            self.layers.append(lay)
        self.n_theta = theta_off
        self._build_indices()
        self._dev = {}

    def _build_indices(self):
        zero = self.n_theta  # index of the appended 0.0
        lane = np.arange(64)
        g, j = lane >> 4, lane & 15
        r = np.arange(4)
        pack_chunks, unpack = [], np.zeros(self.n_theta, dtype=np.int64)
        self.pack_off = []
        self.dw_off = []
        pk, dwo = 0, 0
        for lay in self.layers:
            M, Kh, KT, MT, Kin = lay["M"], lay["Kh"], lay["KT"], lay["MT"], lay["Kin"]
            Mp, Ka = 16 * MT, 16 * (KT + XT)
            aug = np.full((Mp, Ka), zero, dtype=np.int64)
            rows = np.arange(M)[:, None]
            if Kh:
                aug[:M, :Kh] = lay["w_off"] + rows * Kin + np.arange(Kh)[None, :]
            slot = self.slot
            if lay["skip"]:
                aug[:M, 16 * KT + slot[:self.dz]] = lay["w_off"] + rows * Kin + Kh + np.arange(self.dz)[None, :]
            aug[:M, 16 * KT + slot[self.dz]] = lay["b_off"] + np.arange(M)
            # unpack map: theta element -> position in the flat dW_aug buffer
            if Kh:
                unpack[lay["w_off"] + rows * Kin + np.arange(Kh)[None, :]] = dwo + rows * Ka + np.arange(Kh)[None, :]
            if lay["skip"]:
                unpack[lay["w_off"] + rows * Kin + Kh + np.arange(self.dz)[None, :]] = \
                    dwo + rows * Ka + 16 * KT + slot[:self.dz][None, :]
            unpack[lay["b_off"] + np.arange(M)] = dwo + np.arange(M) * Ka + 16 * KT + slot[self.dz]
            offs = {}

            def add(name, arr):
                nonlocal pk
                offs[name] = (pk, arr.size)
                pack_chunks.append(arr.reshape(-1))
                pk += arr.size

            mt = np.arange(MT)
            # A operand of W_h: [KT][MT][lane][r] = aug[16mt + j, 16kt + 4g + r]
            if KT:
                kt = np.arange(KT)
                add("Wh", aug[(16 * mt[None, :, None, None] + j[None, None, :, None]),
                              (16 * kt[:, None, None, None] + 4 * g[None, None, :, None] + r[None, None, None, :])])
                # A operand of W_h^T: [MT][KT][lane][r] = aug[16mt + 4g + r, 16kt + j]
                add("WhT", aug[(16 * mt[:, None, None, None] + 4 * g[None, None, :, None] + r[None, None, None, :]),
                               (16 * kt[None, :, None, None] + j[None, None, :, None])])
            xt = np.arange(XT)
            add("Ws", aug[(16 * mt[None, :, None, None] + j[None, None, :, None]),
                          (16 * (KT + xt)[:, None, None, None] + 4 * g[None, None, :, None] + r[None, None, None, :])])
            # A operand of W_s^T restricted to the latent channels (the only columns of the augmented input that receive
            # an adjoint): [MT][XL][lane][r] = aug[16mt + 4g + r, 16KT + dim + 16xl + j], zero beyond channel cin - 1
            xl = np.arange(self.xl)
            ch = 16 * xl[None, :, None, None] + j[None, None, :, None] + 0 * r[None, None, None, :]
            wsl = aug[(16 * mt[:, None, None, None] + 4 * g[None, None, :, None] + r[None, None, None, :]),
                      (16 * KT + slot[self.dim + np.minimum(ch, self.cin - 1)])]
            add("WsL", np.where(ch < self.cin, wsl, zero))
            d = np.arange(3)
            add("tanc", aug[(16 * mt[None, :, None, None] + 4 * g[None, None, :, None] + r[None, None, None, :]),
                            (16 * KT + d)[:, None, None, None] + 0 * mt[None, :, None, None]])
            self.pack_off.append(offs)
            self.dw_off.append((dwo, Mp, Ka))
            dwo += Mp * Ka
        self.pack_index_np = np.concatenate(pack_chunks)
        self.unpack_index_np = unpack
        self.n_pack = pk
        self.n_dw = dwo

    def device_indices(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = (torch.from_numpy(self.pack_index_np).to(device),
                              torch.from_numpy(self.unpack_index_np).to(device))
        return self._dev[key]

    def pack(self, params):
        """params: 12 tensors (w0, b0, ..., w5, b5) -> flat pack buffer (one index gather)."""
        dev = params[0].device
        pidx, _ = self.device_indices(dev)
        theta = torch.cat([p.detach().reshape(-1).float() for p in params] + [torch.zeros(1, device=dev)])
        return theta[pidx]

    def pack_bf16(self, packs, nsplit=1):
        """bf16 A-operand packs of the hidden-to-hidden weights: block (q, mt) = the fp32 blocks (2q, mt) and (2q+1, mt)
        lane by lane.  nsplit = 1 (config-4 path): rounded to bf16 -- except the forward packs of the three narrow layers
        fc3 ... fc5, which carry TWO terms hi + lo stacked [2][...] for the two-term products of k_tail_fwd_bf (a kernel that
        takes one term reads the hi part, which comes first).  nsplit = 3 ("fp32x3"): each weight split exactly into three
        bf16 terms hi + mid + lo (8 + 8 + 8 mantissa bits), stacked [3][...] -- the weight side of the fp32-accurate
        six-product scheme of k_layer_coop<..., SPL = 3>.  Returns {(l, "Wh"|"WhT"): tensor}."""
        out = {}
        for l in range(1, 6):
            kt, mt = self.layers[l]["KT"], self.layers[l]["MT"]
            for name, (ka, ma) in (("Wh", (kt, mt)), ("WhT", (mt, kt))):
                if ka % 2 == 0:
                    w = self.pack_view(packs, l, name).view(ka // 2, 2, ma, 64, 4).permute(0, 2, 3, 1, 4)
                    nterm = 2 if (nsplit == 1 and l >= 3 and name == "Wh") else nsplit
                    terms, r = [], w
                    for t in range(nterm):
                        h = r.to(torch.bfloat16)
                        terms.append(h)
                        if t + 1 < nterm:
                            r = r - h.float()
                    out[(l, name)] = (torch.stack(terms, 0) if nterm > 1 else terms[0]).contiguous()
        return out

    def pack_view(self, packs, l, name):
        off, n = self.pack_off[l][name]
        return packs[off:off + n]

    def unpack_grads(self, dw_flat, params):
        _, uidx = self.device_indices(dw_flat.device)
        gtheta = dw_flat[uidx]
        out, o = [], 0
        for p in params:
            n = p.numel()
            out.append(gtheta[o:o + n].view_as(p))
            o += n
        return out


# ------------------------------------------------------------------------------------------------------------
# stream configuration
# ------------------------------------------------------------------------------------------------------------
CANON_PAIRS = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]
# stream sets (S1, S2) the fused fc3 -> fc5 kernels (csrc/jet_tail.hip) and the dgrad-first two-phase backward are compiled for
TAIL_SETS = ((0, 0), (3, 0), (3, 1), (3, 2), (3, 4))


def make_cfg(act, act_param, first, pairs, combo=None):
    """Map a derivative request onto a compiled stream configuration; returns (JetCfg, S, padded_pairs).

    combo = {pair: alpha}: ONE combined second-order stream sum_k alpha_k d2/dq_a dq_b instead of one per pair."""
    if combo:
        cfg = JetCfg()
        cfg.S1, cfg.S2, cfg.combo = 3, 1, 1
        for k, pr in enumerate(CANON_PAIRS):
            cfg.alpha[k] = float(combo.get(pr, 0.0))
        cfg.act = _lib.ACT_CODES[act]
        cfg.act_param = float(act_param)
        return cfg, 5, ["combo"]
    pairs = [tuple(sorted(p)) for p in pairs]
    if pairs and not first:
        first = True
    if not first:
        s1, s2, pp = 0, 0, []
    elif not pairs:
        s1, s2, pp = 3, 0, []
    elif len(pairs) <= 2:
        s1, s2, pp = 3, 2, pairs + [pairs[-1]] * (2 - len(pairs))
    elif len(pairs) <= 4 and os.environ.get("STPDE_S34", "1") != "0":
        # (round 5: BASELINE configs[4] names four second derivatives and ran padded to six; STPDE_S34=0 = the padded set)
        s1, s2, pp = 3, 4, pairs + [pairs[-1]] * (4 - len(pairs))
    elif len(pairs) <= 6:
        s1, s2, pp = 3, 6, pairs + [pairs[-1]] * (6 - len(pairs))
    else:
        raise NotImplementedError("more than 6 second-order derivative pairs")
    cfg = JetCfg()
    cfg.S1, cfg.S2 = s1, s2
    for k, (a, b) in enumerate(pp):
        cfg.pair0[k], cfg.pair1[k] = a, b
    cfg.act = _lib.ACT_CODES[act]
    cfg.act_param = float(act_param)
    return cfg, 1 + s1 + s2, pp


def box_constants(shape3, xmin, xmax):
    """lo_c, hi_c, cube with the reference's own fp32 expression sequence
    (src/regular_nd_grid_interpolation.py:40-51), evaluated on the host."""
    dim = len(shape3)

    def as_vec(v):
        if isinstance(v, (int, float)):
            return float(v) * torch.ones([dim], dtype=torch.float32)
        if torch.is_tensor(v):
            return v.detach().to("cpu", torch.float32).reshape(-1)
        return torch.tensor(np.asarray(v)).to(torch.float32).reshape(-1)

    lo, hi = as_vec(xmin), as_vec(xmax)
    if lo.numel() != dim or hi.numel() != dim:
        raise ValueError("xmin/xmax must have one entry per grid dimension")
    if bool((lo != 0).any()):
        # quirk a-Q1: the reference computes ind0 = floor(q / cubesize) with no xmin offset, so xmin != 0
        # silently mis-indexes there; the HIP path refuses instead of reproducing or "fixing" it.
        raise ValueError("xmin must be 0 (the reference's cell index ignores xmin)")
    size = torch.tensor(list(shape3)).float()
    eps = 1e-6 * (hi - lo)
    cube = (hi - lo) / (size - 1)
    return (lo + eps).tolist(), (hi - eps).tolist(), cube.tolist()


_box_cache = {}
_box_tensor_cache = {}     # id(xmin) -> (weakref(xmin), weakref(xmax), key, constants)
_box_lock = threading.RLock()   # re-entrant: a weakref callback (_drop) may fire from a GC pass inside the lock


def cached_box_constants(shape3, xmin, xmax):
    """box_constants with a cache, so that device-tensor bounds (train.py:48-49 passes CUDA tensors) do not cost a
    device sync per call.  Tensor entries are keyed on id(xmin) and validated through weak references to BOTH tensor
    objects plus their version counters (an id alone could alias a freed tensor; a WeakKeyDictionary cannot be used
    because its lookups compare keys with ``==``, which is elementwise for tensors); python scalars / sequences are
    keyed by value."""
    shape3 = tuple(shape3)
    if torch.is_tensor(xmin) and torch.is_tensor(xmax):
        import weakref
        key = (shape3, xmin._version, xmax._version)
        with _box_lock:
            ent = _box_tensor_cache.get(id(xmin))
            if ent is not None and ent[0]() is xmin and ent[1]() is xmax and ent[2] == key:
                return ent[3]
        val = box_constants(shape3, xmin, xmax)
        ident = id(xmin)

        def _drop(_ref, ident=ident):
            with _box_lock:
                cur = _box_tensor_cache.get(ident)
                if cur is not None and cur[0] is _ref:
                    del _box_tensor_cache[ident]

        with _box_lock:
            if len(_box_tensor_cache) > 256:
                _box_tensor_cache.clear()
            _box_tensor_cache[ident] = (weakref.ref(xmin, _drop), weakref.ref(xmax), key, val)
        return val
    if torch.is_tensor(xmin) or torch.is_tensor(xmax):
        return box_constants(shape3, xmin, xmax)
    key = (shape3, repr(xmin), repr(xmax))
    with _box_lock:
        if key not in _box_cache:
            if len(_box_cache) > 64:
                _box_cache.clear()
            _box_cache[key] = box_constants(shape3, xmin, xmax)
        return _box_cache[key]


# ------------------------------------------------------------------------------------------------------------
# the autograd function
# ------------------------------------------------------------------------------------------------------------
class _Meta:
    pass


def _AW():
    """floats of storage per accumulated dW element: 1, or -- deterministic mode -- a long accumulator (_lib.DET_K int64)"""
    return 2 * _lib.DET_K if _lib.deterministic else 1


def _layer_desc(ntiles, lay, cfg, first_hidden, bf16=0, packed=0):
    d = LayerDesc()
    d.ntiles, d.KT, d.MT, d.first_hidden, d.cfg = ntiles, lay["KT"], lay["MT"], int(first_hidden), cfg
    d.mfma_bf16 = int(bf16)
    d.packed = int(packed)
    d.det = int(_lib.deterministic)      # (read by the weight-gradient entry points only)
    return d


def _buf_floats(meta, l, nt):
    """floats of the layer buffer of layer l's output rows for nt row tiles: [S][MT] fp32 blocks, or the PACKED form (bf16
    mode: value stream fp32, derivative streams bf16; include/stpde_hip.h, stpde_layer_desc.packed)."""
    mt = meta.plan.layers[l]["MT"]
    if (meta.packed_mask >> l) & 1:
        return nt * mt * (256 + (meta.S - 1) * 128)
    return nt * meta.S * mt * _FRAG


def _adj_floats(meta, l, nt):
    """floats of the adjoint buffer of layer l's output rows: the size of the layer buffer, or -- packed mode -- a packed
    ADJOINT buffer, every stream bf16 (layer 0: its value stream only)."""
    mt = meta.plan.layers[l]["MT"]
    if (meta.packed_mask >> l) & 1:
        return nt * (meta.S if l else 1) * mt * 128
    return nt * mt * _FRAG if l == 0 else _buf_floats(meta, l, nt)


def _pk(meta, l, writes):
    """stpde_layer_desc.packed of a call on layer l that writes the buffer of layer `writes` (-1: none)."""
    m = meta.packed_mask
    return (1 if (l >= 2 and (m >> (l - 1)) & 1) else 0) | (2 if (writes >= 0 and (m >> writes) & 1) else 0) | \
        (4 if (m >> l) & 1 else 0)


# ------------------------------------------------------------------------------------------------------------
# one C call per direction (include/stpde_hip.h: stpde_lig_imnet_jet_fwd / _bwd); the per-kernel functions below stay as
# the profiling path (bench.py's per-kernel event timings) and are what ``use_pipeline = False`` selects (tests)
# ------------------------------------------------------------------------------------------------------------
use_pipeline = True


def _dp(t):
    return None if t is None else t.data_ptr()


def _plan_desc(meta, packs):
    plan = meta.plan
    d = ImNetPlanDesc()
    d.nlayers, d.cin, d.cout, d.nf16 = 6, plan.cin, plan.cout, plan.nf // 16
    pv = plan.pack_view
    for l, lay in enumerate(plan.layers):
        d.KT[l], d.MT[l] = lay["KT"], lay["MT"]
        if lay["KT"]:
            d.Wh[l], d.WhT[l] = pv(packs, l, "Wh").data_ptr(), pv(packs, l, "WhT").data_ptr()
        d.Ws[l], d.WsL[l], d.tanc[l] = (pv(packs, l, n).data_ptr() for n in ("Ws", "WsL", "tanc"))
        if meta.packs16:
            d.Wh16[l], d.WhT16[l] = _dp(meta.packs16.get((l, "Wh"))), _dp(meta.packs16.get((l, "WhT")))
        d.dw_off[l] = plan.dw_off[l][0]
    d.mfma_bf16 = meta.nsplit if meta.packs16 else 0
    d.packed_mask = meta.packed_mask
    return d


def _gather_desc(meta, latent, Pc, p0):
    gd = GatherDesc()
    gd.P, gd.N, gd.B = Pc, meta.N, meta.B
    gd.n0, gd.n1, gd.n2, gd.C = latent.shape[1], latent.shape[2], latent.shape[3], latent.shape[4]
    for k in range(3):
        gd.lo_c[k], gd.hi_c[k], gd.cube[k] = meta.lo_c[k], meta.hi_c[k], meta.cube[k]
    gd.p_base = p0
    for k in range(6):
        gd.alpha[k] = meta.cfg_out.alpha[k]
    return gd


def _flags(meta, need_grad):
    f = 0
    if need_grad:
        f |= _lib.F_STASH
    if value_tiles:
        f |= _lib.F_VALUE_TILES
    if fused_tail:
        f |= _lib.F_FUSED_TAIL
    if tan0_rowsum:
        f |= _lib.F_TAN0_ROWSUM
    if deterministic_dlatent:
        f |= _lib.F_DETERMINISTIC
    if meta.need_wgrad:
        f |= _lib.F_WGRAD
    if not wgrad_split:
        f |= _lib.F_WGRAD_FP32
    if not fc1_fused_enabled():
        f |= _lib.F_NO_FC1_FUSED
    if _lib.deterministic:
        f |= _lib.F_DET
    return f


def _forward_chunk_c(meta, packs, latent, pts_c, jets, p0, need_grad=True):
    """_forward_chunk through ONE library call: this function only allocates the chunk's buffers."""
    plan, S = meta.plan, meta.S
    Pc = pts_c.shape[0]
    nt = Pc // 2
    dev = pts_c.device
    ws = LigWorkspace()
    s = dict(p0=p0, Pc=Pc, ws=ws)
    s["X"] = torch.empty(nt * XT * _FRAG, device=dev)
    s["XR"] = None      # (round 5: no row-major copy of X any more -- the weight-gradient kernels transpose the fragments they need)
    s["cw"] = torch.empty(Pc * 8, device=dev) if meta.cfg_out.combo else None
    s["coef"] = torch.empty(Pc * 16, device=dev)
    s["cell"] = torch.empty(Pc, device=dev, dtype=torch.int32)
    s["bufs"] = [None] + [torch.empty(_buf_floats(meta, l, nt), device=dev) for l in range(1, 6)]
    s["z0"] = torch.empty(nt * plan.layers[0]["MT"] * _FRAG, device=dev) if need_grad else None
    ws.X, ws.XR, ws.coef, ws.cw, ws.cell = (_dp(s[k]) for k in ("X", "XR", "coef", "cw", "cell"))
    ws.pre[0] = _dp(s["z0"])
    for l in range(1, 6):
        ws.pre[l] = s["bufs"][l].data_ptr()
    pd = _plan_desc(meta, packs)
    gd = _gather_desc(meta, latent, Pc, p0)
    check(_lib.lib().stpde_lig_imnet_jet_fwd(C.byref(pd), C.byref(meta.cfg), C.byref(meta.cfg_out), C.byref(gd), ptr(pts_c),
                                             ptr(latent), C.byref(ws), C.c_void_p(jets.data_ptr() + 4 * p0), jets.shape[2],
                                             _flags(meta, need_grad), stream_ptr()))
    s["gd"] = gd
    return s


def _backward_chunk_c(meta, packs, saved, jets_bar, dw_flat, dlatent, pbar=None, after_dlatent=None):
    """_backward_chunk through ONE library call (+ allocation of its scratch).

    after_dlatent (callable or None): "dgrad-first" order in TWO calls -- phase A runs the whole input-gradient chain into
    fresh adjoint buffers and finishes d latent, after_dlatent() is called (the point-sharded step starts the all-reduce of
    d latent there), phase B computes the remaining weight gradients while that all-reduce is in flight.  (Both phases on the
    current stream.  Round 4's variant with phase B on a side stream, so that the U-Net backward ran beside it, measured slower
    -- profiles/r4_overlap_timeline.txt -- and was deleted in round 6.)"""
    plan, S = meta.plan, meta.S
    Pc, ws, gd = saved["Pc"], saved["ws"], saved["gd"]
    nt = Pc // 2
    dev = saved["X"].device
    keep = []

    def buf(n, dtype=torch.float32):
        t = torch.empty(n, device=dev, dtype=dtype)
        keep.append(t)
        return t.data_ptr()

    MT0 = plan.layers[0]["MT"]
    SP0 = 1 + meta.cfg.S1
    ws.abar2x, ws.abar3x = buf(_adj_floats(meta, 2, nt)), buf(_adj_floats(meta, 3, nt))
    if meta.packed_mask:        # packed ADJOINT buffers are a format of their own: none of them goes over a stash
        ws.abar1x, ws.abar4x = buf(_adj_floats(meta, 1, nt)), buf(_adj_floats(meta, 4, nt))
        if meta.packed_mask & 1:
            ws.abar0x = buf(_adj_floats(meta, 0, nt))
    ws.tan0 = buf(nt * MT0 * 48) if (SP0 == 4 and tan0_rowsum) else None
    ws.abar0 = buf(nt * SP0 * MT0 * _FRAG) if (SP0 == 4 and not tan0_rowsum) else None
    if dlatent is not None and deterministic_dlatent:
        cp = (plan.cin + 3) // 4 * 4
        n_nodes = meta.B * meta.grid_shape[0] * meta.grid_shape[1] * meta.grid_shape[2]
        ws.xrows = buf(Pc * 8 * cp)
        ws.perm, ws.start = buf(Pc, torch.int32), buf(n_nodes + 1, torch.int32)
        nb = int(_lib.lib().stpde_lig_sort_tmp_bytes(Pc, n_nodes))
        ws.sort_tmp, ws.sort_tmp_bytes = buf(nb, torch.uint8), nb
    pd = _plan_desc(meta, packs)
    flags = _flags(meta, True)
    two_phase = (after_dlatent is not None and fused_tail and tan0_rowsum and plan.nf in (16, 32) and SP0 in (1, 4)
                 and (meta.cfg.S1, meta.cfg.S2) in TAIL_SETS and (meta.cfg.S2 != 1 or saved["cw"] is not None))

    def call(fl):
        check(_lib.lib().stpde_lig_imnet_jet_bwd(C.byref(pd), C.byref(meta.cfg), C.byref(meta.cfg_out), C.byref(meta.cfg_val),
                                                 C.byref(gd), C.byref(ws), C.c_void_p(jets_bar.data_ptr() + 4 * saved["p0"]),
                                                 jets_bar.shape[2], ptr(dw_flat), ptr(dlatent), ptr(pbar), fl, stream_ptr()))

    if two_phase:
        if not meta.packed_mask:
            ws.abar1x = buf(saved["bufs"][1].numel())
        if not meta.packed_mask & 1:
            ws.abar0x = buf(nt * MT0 * _FRAG)
        call(flags | _lib.F_PHASE_A)
        after_dlatent()
        call(flags | _lib.F_PHASE_B)
    else:
        call(flags)
        if after_dlatent is not None:
            after_dlatent()
    return None


def _forward_chunk(meta, packs, latent, pts_c, jets, p0, need_grad=True):
    """Run gather + layers 1..5 + reduce for points [p0, p0+Pc) ; returns the buffers backward needs."""
    if use_pipeline and profile is None:
        return _forward_chunk_c(meta, packs, latent, pts_c, jets, p0, need_grad)
    plan, cfg, S = meta.plan, meta.cfg, meta.S
    L = _lib.lib()
    st = stream_ptr()
    Pc = pts_c.shape[0]
    nt = Pc // 2
    dev = pts_c.device
    X = torch.empty(nt * XT * _FRAG, device=dev)
    XR = None           # (round 5: no row-major copy of X; the weight-gradient kernels read X)
    cw = torch.empty(Pc * 8, device=dev) if meta.cfg_out.combo else None
    coef = torch.empty(Pc * 16, device=dev)
    cell = torch.empty(Pc, device=dev, dtype=torch.int32)
    gd = GatherDesc()
    gd.P, gd.N, gd.B = Pc, meta.N, meta.B
    gd.n0, gd.n1, gd.n2, gd.C = latent.shape[1], latent.shape[2], latent.shape[3], latent.shape[4]
    for k in range(3):
        gd.lo_c[k], gd.hi_c[k], gd.cube[k] = meta.lo_c[k], meta.hi_c[k], meta.cube[k]
    gd.p_base = p0
    for k in range(6):
        gd.alpha[k] = meta.cfg_out.alpha[k]
    with _timed("gather"):
        check(L.stpde_lig_gather(C.byref(gd), ptr(pts_c), ptr(latent), ptr(X), ptr(XR), ptr(coef), ptr(cell),
                                 ptr(cw), st))
    bufs = [None]
    pv = plan.pack_view
    prev = None
    z0 = None
    # forward-only value queries (inference): VALUE-TILE kernels -- four consecutive row tiles share one pass over the
    # weights (the weight operand is what bounds a one-stream pass); same buffers, a quarter of the "tiles"
    vt = S == 1 and not need_grad and (not meta.packs16 or meta.nsplit == 3) and nt % 4 == 0 and value_tiles
    lcfg, lnt = cfg, nt
    if vt:
        lcfg = JetCfg()
        lcfg.S1, lcfg.S2, lcfg.act, lcfg.act_param = 0, 3, cfg.act, cfg.act_param
        lnt = nt // 4
    # fc3 -> fc4 -> fc5 in one kernel (inter-layer data in registers) for the reference widths and the training stream sets
    tail = (fused_tail and plan.nf in (16, 32) and (vt or ((cfg.S1, cfg.S2) in TAIL_SETS
                                                           and (cfg.S2 != 1 or cw is not None))))
    for l in range(1, 6):
        lay = plan.layers[l]
        if tail and l == 3:
            outs = [torch.empty(_buf_floats(meta, k, nt), device=dev) for k in (3, 4, 5)]
            arr = lambda ts: (C.c_void_p * 3)(*[t.data_ptr() for t in ts])
            pk_tail = (meta.packed_mask >> 2) & 1         # bf16 mode: packed buffers on both sides, bf16-operand kernel
            w16s = arr([meta.packs16[(k, "Wh")] for k in (3, 4, 5)]) if pk_tail else None
            with _timed("tail_fwd"):
                check(L.stpde_jet_tail_fwd_p(C.byref(lcfg), lnt, plan.nf // 16, ptr(prev), ptr(X),
                                             arr([pv(packs, k, "Wh") for k in (3, 4, 5)]),
                                             arr([pv(packs, k, "Ws") for k in (3, 4, 5)]),
                                             arr([pv(packs, k, "tanc") for k in (3, 4, 5)]), arr(outs), ptr(cw),
                                             3 if pk_tail else 0, w16s, st))
            bufs += outs
            break
        out = torch.empty(_buf_floats(meta, l, nt), device=dev)
        w16 = meta.packs16.get((l, "Wh")) if meta.packs16 else None
        d = _layer_desc(lnt, lay, lcfg, l == 1, meta.nsplit if w16 is not None else 0, _pk(meta, l, l) & 3)
        if l == 1 and need_grad:
            # value stream of the layer-0 pre-activations, kept for the layer-1 input-gradient kernel (which then writes
            # the layer-0 adjoint over it): reading 2 KB per row back is cheaper than regenerating it on the fp32 MFMA
            z0 = torch.empty(nt * plan.layers[0]["MT"] * _FRAG, device=dev)
        with _timed("layer%d_fwd" % l):
            check(L.stpde_jet_layer_fwd(C.byref(d), ptr(prev), ptr(X), ptr(pv(packs, l, "Wh")),
                                        ptr(pv(packs, l, "Ws")), ptr(pv(packs, l, "tanc")), ptr(pv(packs, 0, "Ws")),
                                        ptr(pv(packs, 0, "tanc")), ptr(out), ptr(cw), ptr(w16),
                                        ptr(z0) if l == 1 else None, st))
        bufs.append(out)
        prev = out
    with _timed("reduce_fwd"):
        check(L.stpde_lig_reduce_fwd(C.byref(meta.cfg_out), S, Pc, plan.cout, ptr(bufs[5]), ptr(coef),
                                     C.c_void_p(jets.data_ptr() + 4 * p0), jets.shape[2], st))
    return dict(X=X, XR=X, coef=coef, cell=cell, bufs=bufs, p0=p0, Pc=Pc, cw=cw, z0=z0)


def _backward_chunk(meta, packs, saved, jets_bar, dw_flat, dlatent, pbar=None, after_dlatent=None):
    """reduce_bwd -> for l = 5..1: wgrad_l (reads abar_l and the still intact pre-activations of layer l-1), then
    dgrad_l (overwrites them with abar_{l-1}) -> wgrad_0 -> xbar/scatter."""
    if "ws" in saved:
        return _backward_chunk_c(meta, packs, saved, jets_bar, dw_flat, dlatent, pbar, after_dlatent)
    plan, cfg, S = meta.plan, meta.cfg, meta.S
    L = _lib.lib()
    st = stream_ptr()
    Pc, p0 = saved["Pc"], saved["p0"]
    nt = Pc // 2
    X, XR, coef, cell, bufs = saved["X"], saved["XR"], saved["coef"], saved["cell"], saved["bufs"]
    cw = saved["cw"]
    dev = X.device
    pv = plan.pack_view
    SP0 = 1 + cfg.S1
    # adjoint of the fc5 output rows (overwrites the forward's out_pre buffer)
    with _timed("reduce_bwd"):
        check(L.stpde_lig_reduce_bwd(C.byref(meta.cfg_out), S, Pc, plan.cout, C.c_void_p(jets_bar.data_ptr() + 4 * p0),
                                     jets_bar.shape[2], ptr(coef), ptr(bufs[5]), st))
    # layer-0 adjoint: the value stream as fragment blocks; the three tangent streams only as per-tile row sums (their
    # layer-0 "input" is the constant column W0[:, d], so only sum_rows matters): 32 + 6 KB per tile instead of 128 KB
    MT0 = plan.layers[0]["MT"]
    split0 = SP0 == 4 and tan0_rowsum
    z0 = saved["z0"]
    # the value-stream-only layer-0 adjoint goes over the z0 stash (same shape; each lane reads before it writes)
    abar0 = z0 if (split0 or SP0 == 1) else torch.empty(nt * SP0 * MT0 * _FRAG, device=dev)
    if meta.packed_mask & 1:       # bf16 mode: packed ADJOINT buffer (bf16 blocks) of its own
        abar0 = torch.empty(_adj_floats(meta, 0, nt), device=dev)
    tan0 = torch.empty(nt * MT0 * 48, device=dev) if split0 else None
    # fc5 -> fc4 -> fc3 input gradients in one kernel (adjoints of layers 4 and 3 feed the next GEMM from the registers).
    # Order: wgrad_5 (needs the pre-activations of fc4's output intact), the chain (abar4 in place; abar3 / abar2 into
    # fresh buffers because wgrad_4 / wgrad_3 still need the pre-activations they would overwrite), wgrad_4, wgrad_3.
    tail = (fused_tail and plan.nf in (16, 32) and (cfg.S1, cfg.S2) in TAIL_SETS
            and (cfg.S2 != 1 or cw is not None))
    abar = {l: bufs[l] for l in range(1, 6)}      # where the adjoint of layer l's output rows lives once it exists
    if meta.packed_mask:           # packed ADJOINT buffers are a format of their own: none of them goes over a stash
        abar[1], abar[4] = (torch.empty(_adj_floats(meta, k, nt), device=dev) for k in (1, 4))
    for l in range(5, 0, -1):
        lay = plan.layers[l]
        w16 = meta.packs16.get((l, "WhT")) if meta.packs16 else None
        d = _layer_desc(nt, lay, cfg, l == 1, meta.nsplit if w16 is not None else 0, _pk(meta, l, l - 1))
        # weight gradient: same operand mode as the layer kernels (``wgrad_split = False`` keeps it on exact-fp32 MFMA in
        # "fp32x3" mode, for A/B timing); only the wide layers, MT >= 8, have bf16-pipe weight-gradient kernels --
        # flagging a narrow layer would take it off its per-wave kernel
        dwg = _layer_desc(nt, lay, cfg, l == 1, (meta.nsplit if wgrad_split or meta.nsplit == 1 else 0)
                          if (w16 is not None and lay["MT"] >= 8) else 0, _pk(meta, l, -1) & 5)
        off, mp, ka = plan.dw_off[l]
        if (l == 1 and meta.need_wgrad and split0 and w16 is not None and abar0 is not z0
                and fc1_fused_enabled() and L.stpde_jet_fc1_bwd_supported(C.byref(d))):
            # bf16 mode, reference width (round 5): weight gradient + input gradient of the first hidden layer in ONE kernel
            # (csrc/jet_fc1_bwd.hip: one read of the adjoint tile, one activation-jet evaluation per z0 element)
            with _timed("layer1_bwd"):
                check(L.stpde_jet_fc1_bwd(C.byref(d), ptr(abar[1]), ptr(w16), ptr(z0), ptr(pv(packs, 0, "tanc")), ptr(cw),
                                          ptr(XR), ptr(abar0), ptr(tan0), ptr(dw_flat[_AW() * off:_AW() * (off + mp * ka)]), ptr(pbar), st))
            continue
        if meta.need_wgrad:
            with _timed("layer%d_wgrad" % l):
                check(L.stpde_jet_wgrad(C.byref(dwg), S, ptr(abar[l]), ptr(bufs[l - 1]) if l > 1 else ptr(saved["z0"]),
                                        ptr(XR), ptr(pv(packs, 0, "tanc")), ptr(dw_flat[_AW() * off:_AW() * (off + mp * ka)]), ptr(cw), st))
        if tail and l >= 3:
            if l == 5:
                abar[3], abar[2] = (torch.empty(_adj_floats(meta, k, nt), device=dev) for k in (3, 2))
                arr = lambda ts: (C.c_void_p * 3)(*[t.data_ptr() for t in ts])
                pk_tail = (meta.packed_mask >> 2) & 1
                w16s = (C.c_void_p * 3)(meta.packs16[(3, "WhT")].data_ptr(), meta.packs16[(4, "WhT")].data_ptr(), None) \
                    if pk_tail else None
                with _timed("tail_dgrad"):
                    check(L.stpde_jet_tail_bwd_p(C.byref(cfg), nt, plan.nf // 16, ptr(bufs[5]),
                                                 arr([pv(packs, k, "WhT") for k in (3, 4, 5)]),
                                                 arr([bufs[2], bufs[3], bufs[4]]), arr([abar[2], abar[3], abar[4]]),
                                                 ptr(cw), ptr(pbar), 3 if pk_tail else 0, w16s, st))
            continue
        if l > 1 and abar[l - 1] is not bufs[l - 1]:       # into a fresh buffer: the stash stays intact
            with _timed("layer%d_dgrad" % l):
                check(L.stpde_jet_layer_bwd_to(C.byref(d), ptr(abar[l]), ptr(pv(packs, l, "WhT")), ptr(bufs[l - 1]),
                                               ptr(abar[l - 1]), ptr(cw), ptr(pbar), ptr(w16), st))
            continue
        with _timed("layer%d_dgrad" % l):
            check(L.stpde_jet_layer_bwd(C.byref(d), ptr(abar[l]), ptr(pv(packs, l, "WhT")),
                                        ptr(bufs[l - 1]) if l > 1 else None, ptr(X), ptr(pv(packs, 0, "Ws")),
                                        ptr(pv(packs, 0, "tanc")), ptr(abar0), ptr(cw), ptr(pbar), ptr(w16),
                                        ptr(tan0) if l == 1 else None, ptr(z0) if l == 1 else None, st))
    if meta.need_wgrad:
        lay = plan.layers[0]
        # (bf16 mode: the adjoint is a packed ADJOINT buffer and the contraction over the rows runs on the bf16 MFMA)
        d = _layer_desc(nt, lay, cfg, False, meta.packed_mask & 1, _pk(meta, 0, -1) & 4)
        off, mp, ka = plan.dw_off[0]
        with _timed("layer0_wgrad"):
            if split0:
                d.cfg = meta.cfg_val      # value stream x raw input (the S = 1 weight-gradient kernels)
                check(L.stpde_jet_wgrad(C.byref(d), 1, ptr(abar0), None, ptr(XR), None,
                                        ptr(dw_flat[_AW() * off:_AW() * (off + mp * ka)]), None, st))
                check(L.stpde_jet_tan0_reduce(nt, MT0, ptr(tan0), ptr(dw_flat[_AW() * off:_AW() * (off + mp * ka)]), ka,
                                              int(_lib.deterministic), st))
            else:
                check(L.stpde_jet_wgrad(C.byref(d), SP0, ptr(abar0), None, ptr(XR), None,
                                        ptr(dw_flat[_AW() * off:_AW() * (off + mp * ka)]), ptr(cw), st))
    if dlatent is not None:
        xd = XbarDesc()
        xd.ntiles, xd.nlayers, xd.C = nt, 5, plan.cin
        xd.n1, xd.n2 = meta.grid_shape[1], meta.grid_shape[2]
        ab = (C.c_void_p * 5)()
        wt = (C.c_void_p * 5)()
        for l in range(5):
            xd.MT[l] = plan.layers[l]["MT"]
            xd.SP[l] = (1 if split0 else SP0) if l == 0 else S
            xd.packed[l], xd.S[l] = (meta.packed_mask >> l) & 1, (S if l else 1)
            ab[l] = (abar0 if l == 0 else abar[l]).data_ptr()
            wt[l] = pv(packs, l, "WsL").data_ptr()
        if not deterministic_dlatent:
            with _timed("xbar_scatter"):
                check(L.stpde_lig_xbar_scatter(C.byref(xd), ab, wt, ptr(cell), ptr(dlatent), st))
        else:
            # deterministic scatter: per-row adjoints, then a per-node gather in fixed order over the cell-sorted points
            # (the same three library calls the one-call path makes, scratch allocated OUTSIDE the timed region: with torch.sort /
            # index_add_ / cumsum in here the caching allocator's stalls showed up as a 60-75 ms "kernel" in bench.py's table)
            cp = (plan.cin + 3) // 4 * 4
            xrows = torch.empty(Pc * 8 * cp, device=dev)
            n_nodes = meta.B * meta.grid_shape[0] * meta.grid_shape[1] * meta.grid_shape[2]
            perm = torch.empty(Pc, device=dev, dtype=torch.int32)
            start = torch.empty(n_nodes + 1, device=dev, dtype=torch.int32)
            nb = int(L.stpde_lig_sort_tmp_bytes(Pc, n_nodes))
            tmp = torch.empty(nb, device=dev, dtype=torch.uint8)
            with _timed("xbar_scatter"):
                check(L.stpde_lig_xbar_rows(C.byref(xd), ab, wt, ptr(xrows), st))
                check(L.stpde_lig_cell_sort(Pc, n_nodes, ptr(cell), ptr(perm), ptr(start), ptr(tmp), nb, st))
                check(L.stpde_lig_dlatent_reduce(meta.B, meta.grid_shape[0], meta.grid_shape[1], meta.grid_shape[2],
                                                 plan.cin, ptr(xrows), ptr(perm), ptr(start), ptr(dlatent), st))
    if after_dlatent is not None:      # per-kernel (profiling) path: weight gradients first, nothing left to overlap
        after_dlatent()
    return None


# {"dlatent": f(tensor) -> work | None, "dw": f(tensor) -> work | None}: collectives of the point-sharded step, started from
# inside the backward (set / cleared by train_step.sharded_step; the first LigJetFunction.backward of the step consumes them).
sync_hooks = None

# True while train_step.sharded_step runs the FORWARD of a step whose backward will install ``sync_hooks`` (the hooks themselves
# are set only around loss.backward(), i.e. after LigJetFunction.forward has made its memory plan): the plan then budgets the
# dgrad-first scratch of the last chunk (with the test on ``sync_hooks`` alone the term was never counted and a
# step near the limit failed with an out-of-memory error in the backward instead of shrinking its chunks).
expect_two_phase = False

def _chunk_ranges(meta, P):
    """[(first point, points)] of the launch chunks: ``meta.chunk`` points each."""
    return [(p0, min(meta.chunk, P - p0)) for p0 in range(0, P, meta.chunk)]

stats = {"recompute_steps": 0}     # calls whose backward rebuilt the stash chunk by chunk (memory plan below)


def _free_bytes(device):
    """Device memory this process can still get: free on the device + what torch's caching allocator holds unused."""
    free, _ = torch.cuda.mem_get_info(device)
    return free + torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)


def _per_point_bytes(meta):
    """(forward stash, backward scratch) bytes per query point of the jet path, from the SAME size functions the chunk
    allocators use (``_buf_floats`` / ``_adj_floats``: fp32 blocks or the packed formats of the bf16 mode), plus the
    fragment images of the augmented input (X, XR), the interpolation coefficients, the per-row weights of a combined
    second-order stream and the cell index.  Tile = 2 points."""
    plan = meta.plan
    mt0 = plan.layers[0]["MT"]
    cp = (plan.cin + 3) // 4 * 4
    fwd_tile = 4 * (sum(_buf_floats(meta, l, 1) for l in range(1, 6)) + mt0 * _FRAG + XT * _FRAG)
    fwd = fwd_tile // 2 + 4 * 16 + (4 * 8 if meta.cfg_out.combo else 0) + 4
    bwd_tile = 4 * (_adj_floats(meta, 2, 1) + _adj_floats(meta, 3, 1) + mt0 * 48)
    if meta.packed_mask:
        bwd_tile += 4 * (_adj_floats(meta, 1, 1) + _adj_floats(meta, 4, 1) + _adj_floats(meta, 0, 1))
    bwd = bwd_tile // 2 + 8 * cp * 4 + 4 + 24          # per-row latent adjoints, permutation, radix-sort scratch
    return fwd, bwd


def _two_phase_bytes(meta):
    """extra bytes per point of the dgrad-first order (last chunk of a sharded_step backward): two more adjoint buffers"""
    mt0 = meta.plan.layers[0]["MT"]
    return 2 * ((0 if meta.packed_mask else _buf_floats(meta, 1, 1)) + (0 if meta.packed_mask & 1 else mt0 * _FRAG))


# Device-memory budget of ONE jet call in bytes (None = whatever the device has free): a call whose forward stash + backward
# scratch would exceed it keeps no stash and rebuilds it chunk by chunk in the backward (+1 forward of compute), with the
# chunk sized so that one chunk's stash + scratch fits half of the budget.  ``STPDE_MEM_BUDGET_GB`` / ``set_memory_budget``
# / ``lig_jets(..., memory_budget=bytes)``: how a rank with less than the 142 GB the default 2^20-point step peaks at
#  runs the same configuration.
memory_budget = (lambda v: int(float(v) * 2 ** 30) if v else None)(os.environ.get("STPDE_MEM_BUDGET_GB"))


def set_memory_budget(nbytes):
    """Budget in bytes for the stash + scratch of one jet call (None: the free device memory); returns the previous one."""
    global memory_budget
    prev, memory_budget = memory_budget, (None if nbytes is None else int(nbytes))
    return prev


def _avail_bytes(meta, device):
    free = _free_bytes(device)
    return free if meta.budget is None else min(free, meta.budget)


def _stash_bytes(meta, P):
    fwd, bwd = _per_point_bytes(meta)
    last = _chunk_ranges(meta, P)[-1][1]
    # (the dgrad-first scratch of the last chunk only when the point-sharded step's hooks are / will be installed: ADVICE r4, r5)
    two = _two_phase_bytes(meta) if (sync_hooks or expect_two_phase) else 0
    return P * fwd + min(P, meta.chunk) * bwd + last * two


def _recompute_chunk(meta, device):
    """Largest power-of-two chunk whose stash + backward scratch takes at most half of the available memory."""
    fwd, bwd = _per_point_bytes(meta)
    two = _two_phase_bytes(meta) if (sync_hooks or expect_two_phase) else 0
    n = max(1, int(0.5 * _avail_bytes(meta, device) / (fwd + bwd + two)))
    c = 1 << (n.bit_length() - 1)
    mult = 8 if meta.S == 1 else 2
    return max(mult, min(c, DEFAULT_CHUNK))


class LigJetFunction(torch.autograd.Function):
    """jets[S, n_out, P] of the LIG+IM-NET composite; differentiable w.r.t. latent grid and IM-NET parameters.

    Reproducibility.  Run to run on the same inputs: the forward (jets, and therefore a stash rebuilt by recomputation) and
    ``d latent`` (``deterministic_dlatent``, the default: per-node gather in a fixed order) are bit-identical.  The IM-NET
    WEIGHT gradients by default are NOT: every weight-gradient kernel (k_wgrad_coop / k_wgrad_quad / k_wgrad_wave,
    csrc/jet_wgrad_impl.h) finishes with fp32 atomic adds of its workgroups' partial sums into the flat dW buffer, whose order
    varies -- they agree to fp32 summation-order rounding (a few 1e-6 relative).  ``_lib.deterministic`` (STPDE_DETERMINISTIC=1,
    round 6) sends those partial sums -- and the U-Net's, unet3d.py -- to order-independent long accumulators instead: the
    gradients are then bit-identical from run to run (exception: the adjoint of a learnable swish beta)."""

    @staticmethod
    @_lib.guarded
    def forward(ctx, meta, latent, pts, act_param, *params):
        # act_param: the learnable swish beta (a tensor input so that autograd routes its gradient) or None
        packs = meta.plan.pack(params)
        meta.packs16 = meta.plan.pack_bf16(packs, meta.nsplit) if meta.bf16 else None
        P = pts.shape[0]
        jets = torch.empty(meta.S_out, meta.plan.cout, P, device=pts.device)
        # grad mode is always off inside Function.forward and ctx.needs_input_grad ignores torch.no_grad(): whether a
        # backward can follow was decided by lig_jets() before apply()
        need_grad = meta.need_grad and any(ctx.needs_input_grad)
        chunk = meta.chunk
        # Memory plan (VERDICT r2 #9 / ADVICE r2): the activation stash of ALL chunks lives until the backward.  When it
        # would not fit the free device memory (or an allocation fails half way), the forward keeps NO stash and the
        # backward re-runs the forward kernels chunk by chunk right before each chunk's backward (the same kernels on the
        # same inputs rebuild the same stash bit for bit): +1 forward of compute, memory bounded by one chunk.
        limit = 0.85 * _avail_bytes(meta, pts.device)
        if need_grad and not meta.recompute and _stash_bytes(meta, P) > limit:
            # the stash of all chunks is what it is, the backward scratch is per launch chunk: smaller chunks (down to 2^16
            # points: below that the launches get short) before giving up the stash (configs[4], S = 10 at 2^20 points: 187 GB
            # of stash + 69 GB of scratch per 2^20-point chunk, 17 GB per 2^18-point chunk)
            c = chunk
            while c > (1 << 16) and _stash_bytes(meta, P) > limit:
                c //= 2
                meta.chunk = c
            if _stash_bytes(meta, P) <= limit:
                chunk = c
            else:
                meta.chunk = chunk
        recompute = need_grad and (meta.recompute or _stash_bytes(meta, P) > limit)
        if recompute:
            chunk = meta.chunk = min(chunk, _recompute_chunk(meta, pts.device))
        saved = []
        oom = False
        try:
            for p0, n in _chunk_ranges(meta, P):
                s = _forward_chunk(meta, packs, latent, pts[p0:p0 + n], jets, p0, need_grad and not recompute)
                if need_grad and not recompute:
                    saved.append(s)
        except torch.OutOfMemoryError:
            if not need_grad or recompute:
                raise
            oom = True
        if oom:
            # OUTSIDE the except block: while it runs, the exception's traceback pins the failed _forward_chunk
            # frame and its half-allocated buffers, so empty_cache() could not release them and the retry was sized against
            # less memory than there is
            saved, s = [], None
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            recompute = True
            chunk = meta.chunk = min(chunk, _recompute_chunk(meta, pts.device))
            for p0, n in _chunk_ranges(meta, P):
                _forward_chunk(meta, packs, latent, pts[p0:p0 + n], jets, p0, False)
        stats["recompute_steps"] += int(recompute)
        ctx.recompute = recompute
        ctx.meta, ctx.packs, ctx.saved = meta, packs, saved
        ctx.inputs = (latent, pts) if need_grad else None   # to rebuild the stash if backward runs again (retain_graph)
        ctx.n_params = len(params)
        ctx.prm_shape = act_param.shape if act_param is not None else None
        ctx.lat_shape = latent.shape
        ctx.params = params
        ctx.used = recompute     # nothing stashed: the backward rebuilds every chunk's stash itself
        return jets

    @staticmethod
    @_lib.guarded
    def backward(ctx, jets_bar):
        meta = ctx.meta
        rebuild = ctx.used
        if rebuild and ctx.inputs is None:
            raise RuntimeError("LigJetFunction.backward: no stash was kept for this call")
        # rebuild: either the memory plan kept no stash (ctx.recompute), or this is a second backward after
        # backward(retain_graph=True) -- the dgrad kernels overwrote the stash in place.  The forward kernels of a chunk are
        # run again right before its backward (same kernels, same inputs -> the same stash, bit for bit).
        ctx.used = True
        jets_bar = jets_bar.contiguous()
        dev = jets_bar.device
        meta.need_wgrad = any(ctx.needs_input_grad[4:])
        need_lat = ctx.needs_input_grad[1]
        # (deterministic mode: long accumulators -- order-independent integer sums -- turned into fp32 after the last chunk)
        dw_flat = torch.zeros(_AW() * meta.plan.n_dw, device=dev) if meta.need_wgrad else None
        det_dw = bool(_lib.deterministic) and meta.need_wgrad
        dlatent = torch.zeros(ctx.lat_shape, device=dev) if need_lat else None
        pbar = torch.zeros(_lib.PBAR_SLOTS, device=dev) if ctx.needs_input_grad[3] else None
        # Point-sharded multi-GPU step (train_step.sharded_step sets ``sync_hooks``): the all-reduce of the partial d latent is
        # started as soon as the LAST chunk has finished it and runs behind that chunk's weight gradients (dgrad-first order);
        # the IM-NET gradients are all-reduced in place in their flat buffer before they are unpacked.
        hooks = sync_hooks if (sync_hooks and not sync_hooks.get("used")) else None
        works = []

        def start_dlatent_sync():
            if hooks and need_lat and hooks.get("dlatent"):
                works.append(hooks["dlatent"](dlatent))
                hooks["dlatent_done"] = dlatent      # WHICH tensor is summed over ranks (train_step._SumGradAcrossRanks)

        if rebuild:
            latent, pts = ctx.inputs
            scratch = torch.empty(meta.S_out, meta.plan.cout, pts.shape[0], device=pts.device)
            ranges = _chunk_ranges(meta, pts.shape[0])
            for p0, n in ranges:
                last = hooks and p0 == ranges[-1][0]
                s = _forward_chunk(meta, ctx.packs, latent, pts[p0:p0 + n], scratch, p0, True)
                _backward_chunk(meta, ctx.packs, s, jets_bar, dw_flat, dlatent, pbar, start_dlatent_sync if last else None)
                s = None
        else:
            for k, s in enumerate(ctx.saved):
                last = hooks and k == len(ctx.saved) - 1
                _backward_chunk(meta, ctx.packs, s, jets_bar, dw_flat, dlatent, pbar, start_dlatent_sync if last else None)
                s["bufs"] = s["z0"] = None  # release the stash chunk by chunk
        ctx.saved = []
        grads = [None] * ctx.n_params
        if det_dw:
            acc, dw_flat = dw_flat, torch.empty(meta.plan.n_dw, device=dev)
            check(_lib.lib().stpde_det_finalize(ptr(acc), meta.plan.n_dw, ptr(dw_flat), stream_ptr()))
            del acc
        if hooks:
            hooks["used"] = True
        if hooks and meta.need_wgrad and hooks.get("dw"):
            works.append(hooks["dw"](dw_flat))
            if pbar is not None:                      # adjoint of the learnable swish beta: same treatment
                works.append(hooks["dw"](pbar))
            hooks["dw_done"] = True
        for wk in works:
            if wk is not None:
                wk.wait()
        if meta.need_wgrad:
            g = meta.plan.unpack_grads(dw_flat, ctx.params)
            grads = [gi if need else None for gi, need in zip(g, ctx.needs_input_grad[4:])]
        dprm = pbar.sum().reshape(ctx.prm_shape) if pbar is not None else None
        return (None, dlatent, None, dprm) + tuple(grads)


def activation_name(module):
    """Map an activation module to the library's enum name; returns (name, param) or None if unknown."""
    import torch.nn as nn
    from . import nonlinearities
    if isinstance(module, nonlinearities.Swish):
        return "swish", module.beta
    table = [(nn.Tanh, "tanh"), (nn.ReLU, "relu"), (nn.Softplus, "softplus"), (nn.ELU, "elu"),
             (nn.LeakyReLU, "leakyrelu")]
    for cls, name in table:
        if type(module) is cls:
            if name == "softplus" and (module.beta != 1 or module.threshold != 20):
                return None
            if name == "elu" and module.alpha != 1.0:
                return None
            if name == "leakyrelu" and module.negative_slope != 0.01:
                return None
            return name, 0.0
    return None


# Module-level settings (plain attributes: tests set them with monkeypatch; none of them is read from the environment since
# round 6 -- the environment selects only the operand mode, the memory budget and the A/B switch of the fused fc1 backward).
# d latent: per-node gather in a fixed order (bit-reproducible, like the reference's CPU index_put_ accumulate); False = the
# fp32-atomic scatter.
deterministic_dlatent = True
# forward-only value queries use the value-tile kernels (4 row tiles per weight pass); False = one tile per pass
value_tiles = True
# layer-0 tangent-stream adjoints as per-tile row sums (False: full fragment blocks)
tan0_rowsum = True
wgrad_split = True
packed_stash = True
# forward of fc3 -> fc4 -> fc5 in one kernel (False: three per-layer kernels)
fused_tail = True

# ``force_recompute = True``: never keep the stash, always rebuild it in the backward (tests; also what the memory plan of
# LigJetFunction.forward switches to on its own when the stash does not fit)
force_recompute = False


def fc1_fused_enabled():
    """bf16 mode: the fused backward of the first hidden layer (csrc/jet_fc1_bwd.hip) is used where the library serves it.
    STPDE_FC1_FUSED=0 (read per call: tests switch it inside one process) runs the input gradient and the weight gradient of
    that layer as two kernels -- the A/B reference of tests/test_gpu_bf16_kernel_variants.py."""
    return os.environ.get("STPDE_FC1_FUSED", "1") != "0"

DEFAULT_CHUNK = 1 << 20   # query points per launch chunk (per-chunk backward scratch: ~50 GB at 2^20; measured on
                          # MI355X: 2^17 / 2^18 / 2^19 / 2^20 points per chunk -> 459.8 / 458.7 / 455.5 / 454.3 ms per step)

# MFMA operand precision of the hidden-to-hidden GEMMs of the wide layers: "fp32" (exact-fp32 MFMA, the default and
# the parity path) or "bf16" (BASELINE config 4: bf16 operands, fp32 accumulation, everything else fp32).
mlp_precision = os.environ.get("STPDE_MLP_PRECISION", "fp32")


def set_mlp_precision(precision):
    """Select "fp32" or "bf16" MFMA operands for subsequent HIP jet calls; returns the previous setting."""
    global mlp_precision
    if precision not in ("fp32", "bf16", "fp32x3"):
        raise ValueError("mlp precision must be 'fp32', 'bf16' or 'fp32x3'")
    prev, mlp_precision = mlp_precision, precision
    return prev


def lig_jets(imnet, latent_grid, query_pts, xmin, xmax, first=True, pairs=(), chunk_points=None, combo=None,
             precision=None, memory_budget=None):
    """HIP evaluation of y and its coordinate derivatives.

    imnet: implicit_net.ImNet (dim=3); latent_grid [b, n0, n1, n2, c]; query_pts [b, p, 3].
    Returns (jets [S, n_out, b*p] with S = 1 + 3*first + len(padded pairs), padded_pairs).
    Stream order: value, d/dq_0, d/dq_1, d/dq_2, then d2/dq_a dq_b per pair.
    combo = {(a, b): alpha}: instead of one stream per pair, ONE combined second-order stream
    sum alpha_ab d2y/dq_a dq_b is carried through the network (S = 5); returned pairs = ["combo"].
    precision: "fp32" | "bf16" MFMA operands of the wide layers (None = module setting ``mlp_precision``).
    memory_budget: bytes the stash + backward scratch of this call may take (None = module setting ``memory_budget``, whose
    None means the free device memory); above it the backward recomputes the forward chunk by chunk (``_recompute_chunk``).
    """
    if not (latent_grid.is_cuda and query_pts.is_cuda):
        raise RuntimeError("the HIP jet path needs CUDA/HIP tensors (no CPU fallback)")
    if latent_grid.dim() != 5 or query_pts.dim() != 3 or query_pts.shape[-1] != 3:
        raise ValueError("lig_jets expects latent_grid [b,n0,n1,n2,c] and query_pts [b,p,3]")
    if latent_grid.dtype != torch.float32 or query_pts.dtype != torch.float32:
        raise ValueError("lig_jets is fp32 only")
    an = activation_name(imnet.activ)
    if an is None:
        raise NotImplementedError("activation %r is not implemented in the HIP jet path" % (imnet.activ,))
    act, prm = an
    prm_tensor = None
    if act == "swish":
        prm_tensor = prm          # learnable beta: a tensor input of the autograd function (its adjoint comes from
        prm = float(prm.detach())  # the dgrad epilogues); the kernels take the current value as a launch constant
    B, N = query_pts.shape[0], query_pts.shape[1]
    if latent_grid.shape[0] != B:
        raise ValueError("batch mismatch between latent_grid and query_pts")
    plan = ImNetPlan.get(imnet.dim, imnet.in_features, imnet.out_features, imnet.nf)
    if latent_grid.shape[-1] != plan.cin:
        raise ValueError("latent channels != imnet.in_features")
    meta = _Meta()
    meta.plan = plan
    precision = precision or mlp_precision
    if precision not in ("fp32", "bf16", "fp32x3"):
        raise ValueError("mlp precision must be 'fp32', 'bf16' or 'fp32x3'")
    meta.bf16 = precision in ("bf16", "fp32x3")
    meta.nsplit = 3 if precision == "fp32x3" else 1
    meta.packs16 = None
    # bf16 mode, reference width: the stashes of the hidden layers' output rows and their adjoints in the PACKED form -- all
    # of their consumers are bf16-operand kernels (layers 1-2: k_layer_coop / k_wgrad_coop; fc3 -> fc5: k_tail_fwd_bf /
    # k_tail_bwd_bf and the bf16 variant of k_wgrad_wave).  STPDE_PACKED_STASH=0: fp32 blocks, fp32 MFMA in fc3 -> fc5.
    meta.packed_mask = 0
    # output streams (what the caller gets) vs MLP streams (what the layer kernels carry): for piecewise-linear
    # activations sigma'' = 0 makes every second-order MLP stream identically zero, so only value + gradient streams
    # go through the network and the reduction supplies the second derivatives from the weight cross terms
    meta.cfg_out, meta.S_out, ppairs = make_cfg(act, prm, first, list(pairs), combo)
    if act in ("relu", "leakyrelu") and ppairs:
        meta.cfg, meta.S, _ = make_cfg(act, prm, True, [])
    else:
        meta.cfg, meta.S = meta.cfg_out, meta.S_out
    # (S > 6 is not served by the bf16 kernels; the packed buffer of fc2's rows is read by the fused fc3 -> fc5 kernels)
    if precision == "bf16" and packed_stash and imnet.nf == 32 and meta.S <= 6:
        tail_ok = fused_tail and meta.cfg.S1 == 3 and (meta.cfg.S2 != 1 or bool(meta.cfg_out.combo))
        # the buffers of fc1 ... fc4's rows (the bf16 kernels are compiled for "every boundary packed" or none)
        # bits 1-4: the layer buffers of fc1 ... fc4's rows; bit 0: the value-stream adjoint of layer 0 as bf16 blocks (needs
        # the tangent row sums: value-only adjoint).  The library serves "all of them" or none.
        meta.packed_mask = 31 if (tail_ok and tan0_rowsum) else 0
    meta.B, meta.N = B, N
    meta.grid_shape = tuple(latent_grid.shape[1:4])
    meta.lo_c, meta.hi_c, meta.cube = cached_box_constants(meta.grid_shape, xmin, xmax)
    meta.need_wgrad = True
    meta.recompute = force_recompute
    meta.budget = memory_budget if memory_budget is not None else globals()["memory_budget"]
    meta.cfg_val = make_cfg(act, prm, False, [])[0]
    P = B * N
    if P == 0:   # empty query set: nothing to launch (the reference returns an empty [b, 0, o] tensor as well)
        return torch.zeros(meta.S_out, plan.cout, 0, device=query_pts.device), ppairs
    pts = query_pts.detach().reshape(P, 3).contiguous()
    # tiles hold 2 points; value-only queries are padded to 8 points (4 tiles) for the value-tile kernels
    mult = 8 if meta.S == 1 else 2
    pad = (-P) % mult
    if pad:
        pts = torch.cat([pts, pts[-1:].expand(pad, 3)], 0)
    meta.P_pad = P + pad
    meta.chunk = max(mult, (chunk_points or DEFAULT_CHUNK) // mult * mult)
    lat = latent_grid.contiguous()
    params = []
    for k in range(6):
        params += [imnet.fc[k].weight, imnet.fc[k].bias]
    meta.need_grad = torch.is_grad_enabled() and (lat.requires_grad or any(p.requires_grad for p in params)
                                                  or (prm_tensor is not None and prm_tensor.requires_grad))
    jets = LigJetFunction.apply(meta, lat, pts, prm_tensor, *params)
    if pad:
        jets = jets[:, :, :P]
    return jets, ppairs
