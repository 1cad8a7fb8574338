"""ctypes binding of libstpde_hip.so (C ABI declared in include/stpde_hip.h).

The library is built in-tree by ``build_library()`` (``hipcc --offload-arch=gfx950``) and is the ONLY device
path of this package: there is no CPU or eager fallback for the operators it implements, and loading fails
loudly when the shared object is missing.
"""
import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
# STPDE_LIB: load another build of the library (A/B timing of kernel variants on one box; never set by tests or the driver)
LIB_PATH = os.environ.get("STPDE_LIB") or os.path.join(_HERE, "libstpde_hip.so")
_SOURCES = ["jet_layer.hip", "jet_layer_s00.hip", "jet_layer_s03.hip", "jet_layer_s30.hip", "jet_layer_s31.hip", "jet_layer_s32.hip", "jet_layer_s34.hip", "jet_layer_s36.hip", "jet_tail.hip", "jet_wgrad.hip", "jet_wgrad_s00.hip", "jet_wgrad_s30.hip", "jet_wgrad_s31.hip", "jet_wgrad_s32.hip", "jet_wgrad_s34.hip", "jet_wgrad_s36.hip", "jet_fc1_bwd.hip", "lig_gather_reduce.hip", "lig_pipeline.hip", "interp_nd.hip", "conv3d.hip", "conv3d_fused.hip", "optim.hip", "residual.hip", "bn.hip", "resample.hip", "api.cpp"]
# --offload-compress: the gfx950 code objects are stored zstd-compressed in the fat binary (the HIP runtime inflates them at
# module load): libstpde_hip.so 68 MB -> ~1/4; the instruction bytes are the same (tools/check_dpp_hazard.py scans them)
_HIPFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics",
             "--offload-compress"]

ABI_VERSION = 314   # == stpde_version() of the library these ctypes signatures were written for (csrc/api.cpp)

ACT_CODES = {"tanh": 0, "relu": 1, "softplus": 2, "elu": 3, "swish": 4, "leakyrelu": 5}
PBAR_SLOTS = 64   # STPDE_PBAR_SLOTS: accumulation slots of the swish-beta adjoint
XT = 3


class JetCfg(C.Structure):
    _fields_ = [("S1", C.c_int), ("S2", C.c_int), ("pair0", C.c_int * 6), ("pair1", C.c_int * 6),
                ("act", C.c_int), ("act_param", C.c_float), ("combo", C.c_int), ("alpha", C.c_float * 6)]


class GatherDesc(C.Structure):
    _fields_ = [("P", C.c_int), ("N", C.c_int), ("B", C.c_int), ("n0", C.c_int), ("n1", C.c_int), ("n2", C.c_int),
                ("C", C.c_int), ("p_base", C.c_int), ("lo_c", C.c_float * 3), ("hi_c", C.c_float * 3),
                ("cube", C.c_float * 3), ("alpha", C.c_float * 6)]


class LayerDesc(C.Structure):
    _fields_ = [("ntiles", C.c_int), ("KT", C.c_int), ("MT", C.c_int), ("first_hidden", C.c_int), ("cfg", JetCfg),
                ("mfma_bf16", C.c_int), ("packed", C.c_int), ("det", C.c_int)]


class XbarDesc(C.Structure):
    _fields_ = [("ntiles", C.c_int), ("nlayers", C.c_int), ("C", C.c_int), ("n1", C.c_int), ("n2", C.c_int),
                ("MT", C.c_int * 8), ("SP", C.c_int * 8), ("packed", C.c_int * 8), ("S", C.c_int * 8)]


class ImNetPlanDesc(C.Structure):       # stpde_imnet_plan
    _fields_ = [("nlayers", C.c_int), ("cin", C.c_int), ("cout", C.c_int), ("nf16", C.c_int), ("KT", C.c_int * 8),
                ("MT", C.c_int * 8), ("Wh", C.c_void_p * 8), ("WhT", C.c_void_p * 8), ("Ws", C.c_void_p * 8),
                ("WsL", C.c_void_p * 8), ("tanc", C.c_void_p * 8), ("Wh16", C.c_void_p * 8), ("WhT16", C.c_void_p * 8),
                ("mfma_bf16", C.c_int), ("packed_mask", C.c_int), ("dw_off", C.c_long * 8)]


class LigWorkspace(C.Structure):        # stpde_lig_workspace
    _fields_ = [("X", C.c_void_p), ("XR", C.c_void_p), ("coef", C.c_void_p), ("cw", C.c_void_p), ("cell", C.c_void_p),
                ("pre", C.c_void_p * 8), ("abar2x", C.c_void_p), ("abar3x", C.c_void_p), ("tan0", C.c_void_p),
                ("abar0", C.c_void_p), ("abar1x", C.c_void_p), ("abar0x", C.c_void_p), ("xrows", C.c_void_p), ("perm", C.c_void_p), ("start", C.c_void_p),
                ("sort_tmp", C.c_void_p), ("sort_tmp_bytes", C.c_ulong), ("abar4x", C.c_void_p)]


F_STASH, F_VALUE_TILES, F_FUSED_TAIL, F_TAN0_ROWSUM, F_DETERMINISTIC, F_WGRAD, F_WGRAD_FP32 = 1, 2, 4, 8, 16, 32, 64
F_PHASE_A, F_PHASE_B, F_NO_FC1_FUSED, F_DET = 128, 256, 512, 1024
LOSS_DET = 16      # STPDE_LOSS_DET: stpde_loss_sum into a long accumulator


class Conv3dDesc(C.Structure):
    _fields_ = [("B", C.c_int), ("T", C.c_int), ("Z", C.c_int), ("X", C.c_int), ("Ci", C.c_int), ("Co", C.c_int),
                ("ksize", C.c_int), ("det", C.c_int)]


class Conv3dFusedArgs(C.Structure):     # stpde_conv3d_fused_args
    _fields_ = [("d", Conv3dDesc), ("x", C.c_void_p), ("w_pack", C.c_void_p), ("bias", C.c_void_p), ("y", C.c_void_p),
                ("x2", C.c_void_p), ("w2_pack", C.c_void_p), ("y2", C.c_void_p), ("wo2_pack", C.c_void_p),
                ("bias2", C.c_void_p), ("in_sums", C.c_void_p), ("in_gamma", C.c_void_p), ("in_beta", C.c_void_p),
                ("in_running_mean", C.c_void_p), ("in_running_var", C.c_void_p), ("in_stat", C.c_void_p),
                ("out_sums", C.c_void_p), ("m", C.c_void_p), ("m_stat", C.c_void_p), ("m_gamma", C.c_void_p),
                ("m_beta", C.c_void_p), ("m_bsum", C.c_void_p), ("Ci2", C.c_int), ("Co2", C.c_int),
                ("in_eps", C.c_float), ("in_momentum", C.c_float)]


class ResampleDesc(C.Structure):
    _fields_ = [("B", C.c_int), ("T", C.c_int), ("Z", C.c_int), ("X", C.c_int), ("C", C.c_int), ("ft", C.c_int),
                ("fz", C.c_int), ("fx", C.c_int)]


BN_REP = 16        # STPDE_BN_REP: replicas of the BatchNorm reduction scratch (include/stpde_hip.h)
DET_K = 6          # STPDE_DET_K: 64-bit windows of a deterministic-mode long accumulator (12 floats of storage per element)
# Deterministic mode (round 6; STPDE_DETERMINISTIC=1 or ``_lib.deterministic = True``): every sum the step accumulates with
# fp32 / fp64 atomics -- U-Net convolution weight / bias gradients, BatchNorm statistics and backward sums, IM-NET weight
# gradients, the loss sums -- goes to order-independent long accumulators (csrc/common.h); with the deterministic d latent
# (the default) a whole training step is then bit-identical from run to run, like the reference's CPU path
# (experiments/rb2d/train.py:58-77).  The one exception: the adjoint of a learnable swish beta keeps its fp32 atomics.
deterministic = os.environ.get("STPDE_DETERMINISTIC", "0") == "1"


class BnDesc(C.Structure):
    _fields_ = [("N", C.c_long), ("C", C.c_int), ("training", C.c_int), ("relu", C.c_int), ("eps", C.c_float),
                ("momentum", C.c_float), ("scratch_zeroed", C.c_int), ("stats_mode", C.c_int), ("reduce_done", C.c_int),
                ("det", C.c_int)]


class AdamDesc(C.Structure):
    _fields_ = [("n", C.c_long), ("clip", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("weight_decay", C.c_float), ("step_size", C.c_float), ("bias2_sqrt", C.c_float)]


class InterpDesc(C.Structure):
    _fields_ = [("P", C.c_int), ("N", C.c_int), ("B", C.c_int), ("dim", C.c_int), ("C", C.c_int),
                ("n", C.c_int * 4), ("lo_c", C.c_float * 4), ("hi_c", C.c_float * 4), ("cube", C.c_float * 4)]


def _sources():
    return [os.path.join(_CSRC, s) for s in _SOURCES if os.path.exists(os.path.join(_CSRC, s))]


def _source_hash(paths):
    import hashlib
    h = hashlib.sha256()
    for p in sorted(paths):
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(_HIPFLAGS).encode())
    return h.hexdigest()


def _includes(path, seen=None):
    """Project headers (``#include "..."``) a source file pulls in, transitively."""
    import re
    seen = set() if seen is None else seen
    out = []
    try:
        text = open(path).read()
    except OSError:
        return out
    for m in re.finditer(r'^\s*#\s*include\s+"([^"]+)"', text, re.M):
        h = os.path.normpath(os.path.join(os.path.dirname(path), m.group(1)))
        if h not in seen and os.path.exists(h):
            seen.add(h)
            out.append(h)
            out += _includes(h, seen)
    return out


def build_library(force=False, verbose=False):
    """Compile every HIP source for gfx950 into libstpde_hip.so (cross-compiles without a GPU).

    Staleness is decided by a content hash of all sources / headers / flags stored next to the library (not by file
    times, which a copy of the tree to another machine may not preserve)."""
    srcs = _sources()
    inc = os.path.join(_HERE, "..", "include", "stpde_hip.h")
    deps = sorted(set(srcs) | {h for s_ in srcs for h in _includes(s_)})
    stamp = LIB_PATH + ".srchash"
    want = _source_hash([p for p in deps if os.path.exists(p)])
    if not force and os.path.exists(LIB_PATH) and os.path.exists(stamp) and open(stamp).read().strip() == want:
        return LIB_PATH
    objdir = os.path.join(_CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    objs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(o)
        ostamp = o + ".srchash"
        owant = _source_hash([s] + _includes(s))        # the source and exactly the headers it (transitively) includes
        if not force and os.path.exists(o) and os.path.exists(ostamp) and open(ostamp).read().strip() == owant:
            continue
        flags = _HIPFLAGS if s.endswith(".hip") else ["-O3", "-std=c++17", "-fPIC"]
        cmd = ["hipcc"] + flags + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, ostamp, owant, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, ostamp, owant, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out.decode()))
        with open(ostamp, "w") as f:
            f.write(owant)
    cmd = ["hipcc", "--offload-arch=gfx950", "--offload-compress", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed: %s\n%s" % (" ".join(cmd), r.stdout.decode()))
    with open(stamp, "w") as f:
        f.write(want)
    return LIB_PATH


_lib = None
_lock = threading.Lock()
_VP = C.c_void_p

_SIGNATURES = {
    "stpde_version": ([], C.c_int),
    "stpde_last_error": ([C.c_char_p, C.c_ulong], C.c_int),
    "stpde_tune": ([C.c_char_p, C.c_int], C.c_int),
    "stpde_trace_enable": ([C.c_int], C.c_int),
    "stpde_trace_read": ([C.c_char_p, C.c_ulong], C.c_long),
    "stpde_lig_gather": ([C.POINTER(GatherDesc), _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP], C.c_int),
    "stpde_jet_layer_fwd": ([C.POINTER(LayerDesc)] + [_VP] * 12, C.c_int),
    "stpde_jet_layer_bwd": ([C.POINTER(LayerDesc)] + [_VP] * 13, C.c_int),
    "stpde_jet_layer_bwd_to": ([C.POINTER(LayerDesc)] + [_VP] * 8, C.c_int),
    "stpde_jet_tail_fwd": ([C.POINTER(JetCfg), C.c_int, C.c_int, _VP, _VP, C.POINTER(_VP), C.POINTER(_VP), C.POINTER(_VP),
                            C.POINTER(_VP), _VP, _VP], C.c_int),
    "stpde_jet_tail_bwd": ([C.POINTER(JetCfg), C.c_int, C.c_int, _VP, C.POINTER(_VP), C.POINTER(_VP), C.POINTER(_VP), _VP,
                            _VP, _VP], C.c_int),
    "stpde_jet_tail_fwd_p": ([C.POINTER(JetCfg), C.c_int, C.c_int, _VP, _VP, C.POINTER(_VP), C.POINTER(_VP), C.POINTER(_VP),
                              C.POINTER(_VP), _VP, C.c_int, C.POINTER(_VP), _VP], C.c_int),
    "stpde_jet_tail_bwd_p": ([C.POINTER(JetCfg), C.c_int, C.c_int, _VP, C.POINTER(_VP), C.POINTER(_VP), C.POINTER(_VP), _VP,
                              _VP, C.c_int, C.POINTER(_VP), _VP], C.c_int),
    "stpde_jet_tan0_reduce": ([C.c_int, C.c_int, _VP, _VP, C.c_int, C.c_int, _VP], C.c_int),
    "stpde_jet_wgrad": ([C.POINTER(LayerDesc), C.c_int] + [_VP] * 7, C.c_int),
    "stpde_jet_fc1_bwd_supported": ([C.POINTER(LayerDesc)], C.c_int),
    "stpde_jet_fc1_bwd": ([C.POINTER(LayerDesc)] + [_VP] * 11, C.c_int),
    "stpde_lig_reduce_fwd": ([C.POINTER(JetCfg), C.c_int, C.c_int, C.c_int, _VP, _VP, _VP, C.c_long, _VP], C.c_int),
    "stpde_lig_reduce_bwd": ([C.POINTER(JetCfg), C.c_int, C.c_int, C.c_int, _VP, C.c_long, _VP, _VP, _VP], C.c_int),
    "stpde_lig_xbar_scatter": ([C.POINTER(XbarDesc), C.POINTER(_VP), C.POINTER(_VP), _VP, _VP, _VP], C.c_int),
    "stpde_lig_xbar_rows": ([C.POINTER(XbarDesc), C.POINTER(_VP), C.POINTER(_VP), _VP, _VP], C.c_int),
    "stpde_lig_dlatent_reduce": ([C.c_int] * 5 + [_VP] * 5, C.c_int),
    "stpde_lig_imnet_jet_fwd": ([C.POINTER(ImNetPlanDesc), C.POINTER(JetCfg), C.POINTER(JetCfg), C.POINTER(GatherDesc), _VP,
                                 _VP, C.POINTER(LigWorkspace), _VP, C.c_long, C.c_int, _VP], C.c_int),
    "stpde_lig_imnet_jet_bwd": ([C.POINTER(ImNetPlanDesc), C.POINTER(JetCfg), C.POINTER(JetCfg), C.POINTER(JetCfg),
                                 C.POINTER(GatherDesc), C.POINTER(LigWorkspace), _VP, C.c_long, _VP, _VP, _VP, C.c_int, _VP],
                                C.c_int),
    "stpde_lig_sort_tmp_bytes": ([C.c_int, C.c_long], C.c_ulong),
    "stpde_lig_cell_sort": ([C.c_int, C.c_long, _VP, _VP, _VP, _VP, C.c_ulong, _VP], C.c_int),
    "stpde_interp_fwd": ([C.POINTER(InterpDesc)] + [_VP] * 7, C.c_int),
    "stpde_interp_bwd_grid": ([C.POINTER(InterpDesc)] + [_VP] * 5, C.c_int),
    "stpde_conv3d_fwd": ([C.POINTER(Conv3dDesc)] + [_VP] * 5, C.c_int),
    "stpde_conv3d_wgrad": ([C.POINTER(Conv3dDesc)] + [_VP] * 4, C.c_int),
    "stpde_conv3d_wgrad_bias": ([C.POINTER(Conv3dDesc)] + [_VP] * 5, C.c_int),
    "stpde_conv3d_fused": ([C.POINTER(Conv3dFusedArgs), C.POINTER(C.c_int), _VP], C.c_int),
    "stpde_conv3d_wgrad_onload": ([C.POINTER(Conv3dDesc)] + [_VP] * 8, C.c_int),
    "stpde_residual_fwd": ([_VP, C.c_int, C.c_int, C.c_int, C.c_int, _VP, C.c_long, C.c_long, _VP, _VP, _VP], C.c_int),
    "stpde_residual_bwd": ([_VP, C.c_int, C.c_int, C.c_int, C.c_int, _VP, C.c_long, C.c_long, _VP, _VP, _VP, _VP],
                           C.c_int),
    "stpde_resample3d": ([C.POINTER(ResampleDesc), C.c_int, _VP, _VP, _VP, _VP], C.c_int),
    "stpde_det_finalize": ([_VP, C.c_long, _VP, _VP], C.c_int),
    "stpde_bn_fwd": ([C.POINTER(BnDesc)] + [_VP] * 10, C.c_int),
    "stpde_bn_bwd": ([C.POINTER(BnDesc)] + [_VP] * 11, C.c_int),
    "stpde_loss_sum": ([C.c_int, C.c_long, _VP, _VP, _VP, _VP], C.c_int),
    "stpde_loss_grad": ([C.c_int, C.c_long, _VP, _VP, _VP, _VP, _VP], C.c_int),
    "stpde_clip_adam": ([C.POINTER(AdamDesc)] + [_VP] * 5, C.c_int),
    "stpde_clip_adam_multi": ([C.POINTER(AdamDesc), _VP, _VP, C.c_int, C.c_int, _VP], C.c_int),
}


def exported_symbols():
    return sorted(_SIGNATURES)


def register_signatures(sigs):
    """Let sibling modules (e.g. the conv3d binding) add their entry points before the first load."""
    _SIGNATURES.update(sigs)


def lib():
    """Load (once) and return the ctypes handle; raises if the shared object is not built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        "libstpde_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "-- this package has no CPU/eager fallback for its HIP operators." % LIB_PATH)
                # PyTorch-ROCm ships its own libamdhip64; import torch FIRST so that this library binds to the same,
                # already initialised HIP runtime (a second runtime instance in the process sees no device)
                import torch  # noqa: F401
                h = C.CDLL(LIB_PATH)
                for name, (argtypes, restype) in _SIGNATURES.items():
                    fn = getattr(h, name)          # AttributeError if the symbol is missing
                    fn.argtypes = argtypes
                    fn.restype = restype
                if h.stpde_version() != ABI_VERSION:
                    raise RuntimeError("libstpde_hip.so at %s has ABI version %d, this package needs %d: rebuild it "
                                       "(python -c 'import __graft_entry__ as g; g.build()')"
                                       % (LIB_PATH, h.stpde_version(), ABI_VERSION))
                _lib = h
    return _lib


_EXC = {1: ValueError, 2: NotImplementedError, 3: RuntimeError}


def check(rc):
    if rc != 0:
        buf = C.create_string_buffer(512)
        lib().stpde_last_error(buf, 512)
        raise _EXC.get(rc, RuntimeError)("libstpde_hip: " + buf.value.decode())


def tune(name, value):
    """stpde_tune: launch-geometry override for tests (0 = the library's own choice); returns the previous value."""
    prev = lib().stpde_tune(name.encode(), int(value))
    if prev < 0:
        raise KeyError(name)
    return prev


class tuned:
    """``with tuned(conv3_lds_gx=24, conv3_lds_minblk=1): ...`` -- overrides restored on exit."""

    def __init__(self, **kv):
        self.kv, self.prev = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.prev[k] = tune(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.prev.items():
            tune(k, v)
        return False


class dispatch_trace:
    """``with dispatch_trace() as tr: ...; tr.kernels`` = the kernel template instantiations launched inside
    (tests use it to assert that a parity case really exercised the kernels the benchmark times)."""

    def __enter__(self):
        lib().stpde_trace_enable(1)
        self.kernels = []
        return self

    def __exit__(self, *exc):
        L = lib()
        n = L.stpde_trace_read(None, 0)
        buf = C.create_string_buffer(int(n) + 1)
        L.stpde_trace_read(buf, n + 1)
        L.stpde_trace_enable(0)
        self.kernels = [k for k in buf.value.decode().split("\n") if k]
        return False

    def has(self, *needles):
        """True if one traced entry contains every needle."""
        return any(all(n in k for n in needles) for k in self.kernels)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    """torch's current stream of the CURRENT device.  Launch sites run under ``device_of(tensor)`` (see ``guarded``), so
    the current device is the one the tensors live on."""
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class device_of:
    """``with device_of(t):`` makes t's device the current HIP device for the launches inside (and restores the
    previous one); a no-op when it already is.  Without it a model on ``cuda:1`` driven from a process whose current
    device is 0 -- e.g. the replicas of the reference's multi-device nn.DataParallel (experiments/rb2d/train.py:352-355)
    -- would be launched on device 0's stream."""
    __slots__ = ("idx", "prev")

    def __init__(self, t):
        self.idx = t.device.index if (t is not None and t.is_cuda) else None
        self.prev = None

    def __enter__(self):
        if self.idx is not None:
            import torch
            cur = torch.cuda.current_device()
            if cur != self.idx:
                self.prev = cur
                torch.cuda.set_device(self.idx)
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            import torch
            torch.cuda.set_device(self.prev)
        return False


def guarded(fn):
    """Decorator for the static forward / backward of an autograd Function (or any function whose arguments include the
    tensors it launches on): runs it under ``device_of`` the first CUDA tensor argument."""
    import functools

    @functools.wraps(fn)
    def wrap(*args, **kw):
        t = None
        for a in args:
            if hasattr(a, "is_cuda") and a.is_cuda:
                t = a
                break
        with device_of(t):
            return fn(*args, **kw)
    return wrap
