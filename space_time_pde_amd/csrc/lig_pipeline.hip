// One call per direction for the whole query path (lig_imnet_jet_fwd / lig_imnet_jet_bwd):
//   forward   gather -> fc1 (fc0 on the fly) -> fc2 -> fc3..fc5 (fused tail or per layer) -> corner reduction
//   backward  corner-reduction adjoint -> weight gradients / input gradients fc5..fc1 -> fc0 weight gradient ->
//             latent-gradient GEMM -> deterministic per-node sums (cell sort on the device) or fp32-atomic scatter
// i.e. what src/local_implicit_grid.py:47-59 + src/implicit_net.py:48-54 + the reverse sweeps of src/pde.py:8-9 and
// loss.backward() (experiments/rb2d/train.py:77) do for one chunk of query points.  Pure launch sequencing: every kernel is
// the one behind the per-layer entry points of include/stpde_hip.h (same order as the Python host used to issue them), the
// caller owns every buffer (stpde_lig_workspace), nothing is allocated or synchronised here.
#include <hipcub/hipcub.hpp>

#include "common.h"

namespace {

struct Seq {                       // first error wins, later launches are skipped
  int rc = STPDE_OK;
  template <class F>
  void operator()(F&& f) {
    if (rc == STPDE_OK) rc = f();
  }
};

bool is_packed(const stpde_imnet_plan* p, int l) { return l >= 0 && ((p->packed_mask >> l) & 1); }

// packed flags of a call on layer l (stpde_layer_desc.packed): 1 = its hidden input pre[l-1], 2 = the buffer it writes,
// 4 = its abar_out; `writes` = index of the layer buffer the call writes (fwd: l, bwd: l - 1), -1 = none
int packed_flags(const stpde_imnet_plan* p, int l, int writes) {
  return ((l >= 2 && is_packed(p, l - 1)) ? 1 : 0) | ((writes >= 0 && is_packed(p, writes)) ? 2 : 0) | (is_packed(p, l) ? 4 : 0);
}

stpde_layer_desc layer_desc(int ntiles, const stpde_imnet_plan* p, int l, const stpde_jet_cfg& cfg, int bf16) {
  stpde_layer_desc d{};
  d.ntiles = ntiles;
  d.KT = p->KT[l];
  d.MT = p->MT[l];
  d.first_hidden = l == 1;
  d.cfg = cfg;
  d.mfma_bf16 = bf16;
  return d;
}

bool tail_ok(const stpde_imnet_plan* p, const stpde_jet_cfg& c, const float* cw, bool value_tiles) {
  if (p->nlayers != 6 || (p->nf16 != 1 && p->nf16 != 2)) return false;
  if (value_tiles) return true;
  const bool set = (c.S1 == 0 && c.S2 == 0) || (c.S1 == 3 && (c.S2 == 0 || c.S2 == 1 || c.S2 == 2 || c.S2 == 4));
  return set && (c.S2 != 1 || cw);
}

__global__ __launch_bounds__(256) void k_cell_count(const int* cell, int P, int* counts) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p < P) atomicAdd(counts + cell[p] + 1, 1);
}
__global__ __launch_bounds__(256) void k_iota(int* v, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) v[i] = i;
}

int key_bits(long n_nodes) {
  int b = 1;
  while ((1L << b) < n_nodes + 1) ++b;
  return b;
}

}  // namespace

// Scratch of stpde_lig_cell_sort for P points on a grid of n_nodes nodes: the radix sort's / scan's temporary storage plus
// two key arrays, one value array and the histogram.
extern "C" unsigned long stpde_lig_sort_tmp_bytes(int P, long n_nodes) {
  if (P <= 0 || n_nodes <= 0) return 0;
  size_t a = 0, b = 0;
  hipcub::DeviceRadixSort::SortPairs(nullptr, a, (const int*)nullptr, (int*)nullptr, (const int*)nullptr, (int*)nullptr, P, 0,
                                     key_bits(n_nodes));
  hipcub::DeviceScan::InclusiveSum(nullptr, b, (const int*)nullptr, (int*)nullptr, (int)(n_nodes + 1));
  const size_t t = (a > b ? a : b);
  return (unsigned long)((t + 255) / 256 * 256 + 2 * (((size_t)P * 4 + 255) / 256 * 256));
}

// perm [P] = point indices in STABLE cell order (ties keep ascending point index: LSD radix sort), start [n_nodes + 1] =
// number of points in cells < c.  Replaces torch.sort + index_add_ + cumsum of the round-2 host (rocprim / ATen kernels).
extern "C" int stpde_lig_cell_sort(int P, long n_nodes, const int* cell, int* perm, int* start, void* tmp,
                                   unsigned long tmp_bytes, void* stream) {
  if (P <= 0 || n_nodes <= 0 || !cell || !perm || !start || !tmp || tmp_bytes < stpde_lig_sort_tmp_bytes(P, n_nodes)) {
    stpde_set_error("lig_cell_sort: bad argument / scratch too small");
    return STPDE_E_BADARG;
  }
  hipStream_t st = (hipStream_t)stream;
  const size_t arr = ((size_t)P * 4 + 255) / 256 * 256;
  char* base = (char*)tmp;
  int* keys_out = (int*)base;
  int* iota = (int*)(base + arr);
  void* cub = base + 2 * arr;
  size_t cub_bytes = tmp_bytes - 2 * arr;
  (void)hipGetLastError();
  hipLaunchKernelGGL(k_iota, dim3((P + 255) / 256), dim3(256), 0, st, iota, P);
  if (hipcub::DeviceRadixSort::SortPairs(cub, cub_bytes, cell, keys_out, (const int*)iota, perm, P, 0, key_bits(n_nodes), st) !=
      hipSuccess) {
    stpde_set_error("lig_cell_sort: radix sort failed");
    return STPDE_E_LAUNCH;
  }
  if (hipMemsetAsync(start, 0, (size_t)(n_nodes + 1) * sizeof(int), st) != hipSuccess) return STPDE_E_LAUNCH;
  hipLaunchKernelGGL(k_cell_count, dim3((P + 255) / 256), dim3(256), 0, st, cell, P, start);
  cub_bytes = tmp_bytes - 2 * arr;
  if (hipcub::DeviceScan::InclusiveSum(cub, cub_bytes, (const int*)start, start, (int)(n_nodes + 1), st) != hipSuccess) {
    stpde_set_error("lig_cell_sort: scan failed");
    return STPDE_E_LAUNCH;
  }
  return stpde_check_launch("lig_cell_sort");
}

extern "C" int stpde_lig_imnet_jet_fwd(const stpde_imnet_plan* p, const stpde_jet_cfg* cfg_mlp, const stpde_jet_cfg* cfg_out,
                                       const stpde_gather_desc* gd, const float* pts, const float* latent,
                                       const stpde_lig_workspace* ws, float* jets, long ldp, int flags, void* stream) {
  if (!p || !cfg_mlp || !cfg_out || !gd || !pts || !latent || !ws || !jets || p->nlayers < 2 || p->nlayers > 8 || (gd->P & 1)) {
    stpde_set_error("lig_imnet_jet_fwd: bad argument");
    return STPDE_E_BADARG;
  }
  const int nt = gd->P / 2, NL = p->nlayers;
  const bool stash = flags & STPDE_F_STASH;
  const int S = 1 + cfg_mlp->S1 + cfg_mlp->S2;
  Seq seq;
  seq([&] { return stpde_lig_gather(gd, pts, latent, ws->X, nullptr, ws->coef, ws->cell, ws->cw, stream); });
  // forward-only value queries: VALUE-TILE kernels (four consecutive row tiles share one pass over the weights)
  const bool vt = S == 1 && !stash && (flags & STPDE_F_VALUE_TILES) && nt % 4 == 0 && (!p->mfma_bf16 || p->mfma_bf16 == 3);
  stpde_jet_cfg lcfg = *cfg_mlp;
  int lnt = nt;
  if (vt) {
    lcfg.S1 = 0;
    lcfg.S2 = 3;
    lnt = nt / 4;
  }
  const bool tail = (flags & STPDE_F_FUSED_TAIL) && tail_ok(p, *cfg_mlp, ws->cw, vt);
  const float* prev = nullptr;
  for (int l = 1; l < NL; ++l) {
    if (tail && l == 3) {
      const float* Wh[3] = {p->Wh[3], p->Wh[4], p->Wh[5]};
      const float* Wsk[3] = {p->Ws[3], p->Ws[4], p->Ws[5]};
      const float* tc[3] = {p->tanc[3], p->tanc[4], p->tanc[5]};
      float* outs[3] = {ws->pre[3], ws->pre[4], ws->pre[5]};
      seq([&] {
        const void* w16[3] = {p->Wh16[3], p->Wh16[4], p->Wh16[5]};
        const bool pk = is_packed(p, 2);      // bf16 mode: packed buffers on both sides, bf16-operand kernel
        return stpde_jet_tail_fwd_p(&lcfg, lnt, p->nf16, prev, ws->X, Wh, Wsk, tc, outs, ws->cw, pk ? 3 : 0, pk ? w16 : nullptr,
                                    stream);
      });
      break;
    }
    const void* w16 = p->mfma_bf16 ? p->Wh16[l] : nullptr;
    stpde_layer_desc d = layer_desc(lnt, p, l, lcfg, w16 ? p->mfma_bf16 : 0);
    d.packed = packed_flags(p, l, l) & 3;
    seq([&] {
      return stpde_jet_layer_fwd(&d, prev, ws->X, p->Wh[l], p->Ws[l], p->tanc[l], p->Ws[0], p->tanc[0], ws->pre[l], ws->cw, w16,
                                 (l == 1 && stash) ? ws->pre[0] : nullptr, stream);
    });
    prev = ws->pre[l];
  }
  seq([&] { return stpde_lig_reduce_fwd(cfg_out, S, gd->P, p->cout, ws->pre[NL - 1], ws->coef, jets, ldp, stream); });
  return seq.rc;
}

extern "C" int stpde_lig_imnet_jet_bwd(const stpde_imnet_plan* p, const stpde_jet_cfg* cfg_mlp, const stpde_jet_cfg* cfg_out,
                                       const stpde_jet_cfg* cfg_val, const stpde_gather_desc* gd, const stpde_lig_workspace* ws,
                                       const float* jets_bar, long ldp, float* dW_flat, float* dlatent, float* act_param_bar,
                                       int flags, void* stream) {
  if (!p || !cfg_mlp || !cfg_out || !cfg_val || !gd || !ws || !jets_bar || p->nlayers != 6 || (gd->P & 1) || !ws->pre[0]) {
    stpde_set_error("lig_imnet_jet_bwd: bad argument (needs the stash of a forward call with STPDE_F_STASH)");
    return STPDE_E_BADARG;
  }
  const int nt = gd->P / 2, NL = p->nlayers;
  const stpde_jet_cfg& cfg = *cfg_mlp;
  const int S = 1 + cfg.S1 + cfg.S2, SP0 = 1 + cfg.S1;
  const bool wgrad = (flags & STPDE_F_WGRAD) && dW_flat;
  // deterministic mode: dW_flat holds long accumulators (STPDE_F_DET; 2 * STPDE_DET_K floats of storage per element)
  const int det = (flags & STPDE_F_DET) ? 1 : 0;
  const long aw = det ? 2 * STPDE_DET_K : 1;
  const bool phaseA = flags & STPDE_F_PHASE_A, phaseB = flags & STPDE_F_PHASE_B;
  const bool dfirst = phaseA || phaseB;                 // dgrad-first order, two calls
  Seq seq;
  const int MT0 = p->MT[0];
  const bool split0 = SP0 == 4 && (flags & STPDE_F_TAN0_ROWSUM) && ws->tan0;
  if (!split0 && SP0 != 1 && !ws->abar0) {
    stpde_set_error("lig_imnet_jet_bwd: workspace.abar0 needed without the tangent row sums");
    return STPDE_E_BADARG;
  }
  const bool tail = (flags & STPDE_F_FUSED_TAIL) && tail_ok(p, cfg, ws->cw, false) && ws->abar2x && ws->abar3x;
  if (dfirst && !(tail && (split0 || SP0 == 1) && ws->abar1x && ws->abar0x)) {
    stpde_set_error("lig_imnet_jet_bwd: the dgrad-first phases need the fused tail, the tangent row sums and workspace.abar1x / abar0x");
    return STPDE_E_BADARG;
  }
  float* z0 = ws->pre[0];
  // value-stream-only layer-0 adjoint: over the z0 stash (each lane reads before it writes), or -- dgrad-first -- into a
  // fresh buffer, because the weight gradient of the first hidden layer still needs z0 afterwards
  // bf16 mode with packed buffers: the adjoints are packed ADJOINT buffers (every stream bf16), a format of their own, so
  // none of them goes over the stash it belongs to
  const bool pkadj = p->packed_mask != 0;
  if (pkadj && !(tail && ws->abar1x && ws->abar4x && (!is_packed(p, 0) || (split0 && ws->abar0x)))) {
    stpde_set_error("lig_imnet_jet_bwd: packed layer buffers need the fused tail and workspace.abar1x / abar4x (and abar0x + the tangent row sums with packed_mask bit 0)");
    return STPDE_E_BADARG;
  }
  float* abar0 = (dfirst || is_packed(p, 0)) ? ws->abar0x : ((split0 || SP0 == 1) ? z0 : ws->abar0);
  float* abar[8];
  for (int l = 1; l < NL; ++l) abar[l] = ws->pre[l];      // where the adjoint of layer l's output rows lives once it exists
  if (tail) {
    abar[3] = ws->abar3x;
    abar[2] = ws->abar2x;
  }
  if (dfirst || pkadj) abar[1] = ws->abar1x;
  if (pkadj) abar[4] = ws->abar4x;

  auto wgrad_l = [&](int l) {
    const void* w16 = p->mfma_bf16 ? p->WhT16[l] : nullptr;
    // same operand mode as the layer kernels; only the wide layers (MT >= 8) have bf16-pipe weight-gradient kernels
    stpde_layer_desc dwg = layer_desc(nt, p, l, cfg, (w16 && p->MT[l] >= 8 && (p->mfma_bf16 == 1 || !(flags & STPDE_F_WGRAD_FP32))) ? p->mfma_bf16 : 0);
    dwg.det = det;
    dwg.packed = packed_flags(p, l, -1) & 5;
    seq([&] {
      return stpde_jet_wgrad(&dwg, S, abar[l], l > 1 ? ws->pre[l - 1] : z0, ws->X, p->tanc[0], dW_flat + aw * p->dw_off[l], ws->cw,
                             stream);
    });
  };
  auto wgrad_0 = [&] {
    stpde_layer_desc d = layer_desc(nt, p, 0, cfg, is_packed(p, 0) ? 1 : 0);    // (bf16 mode: bf16 contraction over the rows)
    d.first_hidden = 0;
    d.packed = packed_flags(p, 0, -1) & 4;
    d.det = det;
    float* dw0 = dW_flat + aw * p->dw_off[0];
    if (split0) {
      d.cfg = *cfg_val;       // value stream x raw input (the S = 1 weight-gradient kernels)
      seq([&] { return stpde_jet_wgrad(&d, 1, abar0, nullptr, ws->X, nullptr, dw0, nullptr, stream); });
      seq([&] { return stpde_jet_tan0_reduce(nt, MT0, ws->tan0, dw0, 16 * STPDE_XT, det, stream); });
    } else {
      seq([&] { return stpde_jet_wgrad(&d, SP0, abar0, nullptr, ws->X, nullptr, dw0, ws->cw, stream); });
    }
  };
  auto tail_bwd = [&] {
    const float* WhT[3] = {p->WhT[3], p->WhT[4], p->WhT[5]};
    const float* pre[3] = {ws->pre[2], ws->pre[3], ws->pre[4]};
    float* outs[3] = {abar[2], abar[3], abar[4]};
    seq([&] {
      const void* w16[3] = {p->WhT16[3], p->WhT16[4], nullptr};
      const bool pk = is_packed(p, 2);
      return stpde_jet_tail_bwd_p(&cfg, nt, p->nf16, ws->pre[5], WhT, pre, outs, ws->cw, act_param_bar, pk ? 3 : 0,
                                  pk ? w16 : nullptr, stream);
    });
  };
  auto dgrad_l = [&](int l) {
    const void* w16 = p->mfma_bf16 ? p->WhT16[l] : nullptr;
    stpde_layer_desc d = layer_desc(nt, p, l, cfg, w16 ? p->mfma_bf16 : 0);
    d.packed = packed_flags(p, l, l - 1);
    if (l > 1 && abar[l - 1] != ws->pre[l - 1]) {       // into a fresh buffer: the pre-activations stay intact
      seq([&] {
        return stpde_jet_layer_bwd_to(&d, abar[l], p->WhT[l], ws->pre[l - 1], abar[l - 1], ws->cw, act_param_bar, w16, stream);
      });
      return;
    }
    seq([&] {
      return stpde_jet_layer_bwd(&d, abar[l], p->WhT[l], l > 1 ? ws->pre[l - 1] : nullptr, ws->X, p->Ws[0], p->tanc[0], abar0,
                                 ws->cw, act_param_bar, w16, (l == 1 && split0) ? ws->tan0 : nullptr, l == 1 ? z0 : nullptr,
                                 stream);
    });
  };
  // bf16 mode, reference width: weight gradient + input gradient of the first hidden layer in one kernel (jet_fc1_bwd.hip)
  // Decided ONCE per call, from the plan and the flags only (it used to be re-evaluated, with a getenv, at every use;
  // phase B skips fc1's weight gradient exactly when phase A's fused kernel produced it, so both phases of a dgrad-first
  // backward MUST be called with the same flags apart from the phase bits -- STPDE_F_WGRAD and STPDE_F_NO_FC1_FUSED included)
  const bool fc1_fused_on = [&]() -> bool {
    if (!wgrad || !split0 || (flags & STPDE_F_NO_FC1_FUSED)) return false;
    stpde_layer_desc d = layer_desc(nt, p, 1, cfg, p->mfma_bf16 ? p->mfma_bf16 : 0);
    d.packed = packed_flags(p, 1, 0);
    return p->WhT16[1] && stpde_jet_fc1_bwd_supported(&d) != 0 && abar0 != z0;
  }();
  auto fc1_fused_ok = [&]() -> bool { return fc1_fused_on; };
  auto fc1_fused = [&] {
    stpde_layer_desc d = layer_desc(nt, p, 1, cfg, p->mfma_bf16);
    d.packed = packed_flags(p, 1, 0);
    d.det = det;
    seq([&] {
      return stpde_jet_fc1_bwd(&d, abar[1], p->WhT16[1], z0, p->tanc[0], ws->cw, ws->X, abar0, ws->tan0,
                               dW_flat + aw * p->dw_off[1], act_param_bar, stream);
    });
  };
  auto dlatent_part = [&] {
    if (!dlatent) return;
    stpde_xbar_desc xd{};
    xd.ntiles = nt;
    xd.nlayers = 5;
    xd.C = p->cin;
    xd.n1 = gd->n1;
    xd.n2 = gd->n2;
    const float* ab[5];
    const float* wt[5];
    for (int l = 0; l < 5; ++l) {
      xd.MT[l] = p->MT[l];
      xd.SP[l] = l == 0 ? (split0 ? 1 : SP0) : S;
      xd.packed[l] = is_packed(p, l) ? 1 : 0;
      xd.S[l] = l == 0 ? 1 : S;
      ab[l] = l == 0 ? abar0 : abar[l];
      wt[l] = p->WsL[l];
    }
    if (!(flags & STPDE_F_DETERMINISTIC)) {
      seq([&] { return stpde_lig_xbar_scatter(&xd, ab, wt, ws->cell, dlatent, stream); });
    } else {
      const long n_nodes = (long)gd->B * gd->n0 * gd->n1 * gd->n2;
      seq([&] { return stpde_lig_xbar_rows(&xd, ab, wt, ws->xrows, stream); });
      seq([&] { return stpde_lig_cell_sort(gd->P, n_nodes, ws->cell, ws->perm, ws->start, ws->sort_tmp, ws->sort_tmp_bytes, stream); });
      seq([&] {
        return stpde_lig_dlatent_reduce(gd->B, gd->n0, gd->n1, gd->n2, p->cin, ws->xrows, ws->perm, ws->start, dlatent, stream);
      });
    }
  };

  if (dfirst) {
    if (phaseA) {
      // adjoint of the fc5 output rows (overwrites the forward's output buffer); fc5's weight gradient needs the
      // pre-activations of fc4's output, which the fused chain overwrites with abar4
      seq([&] { return stpde_lig_reduce_bwd(cfg_out, S, gd->P, p->cout, jets_bar, ldp, ws->coef, ws->pre[NL - 1], stream); });
      if (wgrad) wgrad_l(5);
      tail_bwd();
      dgrad_l(2);
      if (fc1_fused_ok())
        fc1_fused();          // (fc1's weight gradient comes with its input gradient: phase B skips it)
      else
        dgrad_l(1);
      dlatent_part();
    }
    if (phaseB && wgrad) {
      for (int l = 4; l >= (fc1_fused_ok() ? 2 : 1); --l) wgrad_l(l);
      wgrad_0();
    }
    return seq.rc;
  }
  // one call: weight gradient of layer l, then its input gradient (which overwrites what the weight gradient read)
  seq([&] { return stpde_lig_reduce_bwd(cfg_out, S, gd->P, p->cout, jets_bar, ldp, ws->coef, ws->pre[NL - 1], stream); });
  for (int l = NL - 1; l >= 1; --l) {
    if (l == 1 && fc1_fused_ok()) {
      fc1_fused();
      continue;
    }
    if (wgrad) wgrad_l(l);
    if (tail && l >= 3) {
      if (l == 5) tail_bwd();
      continue;
    }
    dgrad_l(l);
  }
  if (wgrad) wgrad_0();
  dlatent_part();
  return seq.rc;
}
