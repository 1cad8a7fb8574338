// C-ABI entry points of the layer kernels (implementation: jet_layer_impl.h, one TU per stream configuration).
#include "jet_layer_impl.h"

static int dispatch_streams(const LayerArgs& a, int mode, hipStream_t stream) {
  const int S1 = a.cfg.S1, S2 = a.cfg.S2;
  if (S1 == 0 && S2 == 0) return stpde_layer_launch_0_0(a, mode, stream);
  if (S1 == 0 && S2 == 3) return stpde_layer_launch_0_3(a, mode, stream);   // value-tile mode (4 row tiles per pass)
  if (S1 == 3 && S2 == 0) return stpde_layer_launch_3_0(a, mode, stream);
  if (S1 == 3 && S2 == 1 && a.cfg.combo && a.cw) return stpde_layer_launch_3_1(a, mode, stream);
  if (S1 == 3 && S2 == 2) return stpde_layer_launch_3_2(a, mode, stream);
  if (S1 == 3 && S2 == 4) return stpde_layer_launch_3_4(a, mode, stream);
  if (S1 == 3 && S2 == 6) return stpde_layer_launch_3_6(a, mode, stream);
  stpde_set_error("stream configuration S1=%d S2=%d not compiled (supported: (0,0) (0,3 = four value tiles, forward only) (3,0) (3,1 combined, needs cw) (3,2) (3,4) (3,6))", S1, S2);
  return STPDE_E_UNSUPPORTED;
}

static int check_cfg(const stpde_layer_desc* d) {
  if (!d || d->ntiles <= 0 || d->MT <= 0 || d->KT < 0) {
    stpde_set_error("layer desc: bad sizes");
    return STPDE_E_BADARG;
  }
  if (d->cfg.act < 0 || d->cfg.act > 5) {
    stpde_set_error("layer desc: unknown activation %d", d->cfg.act);
    return STPDE_E_BADARG;
  }
  for (int p = 0; p < d->cfg.S2 && p < 6; ++p)
    if (d->cfg.pair0[p] < 0 || d->cfg.pair0[p] > 2 || d->cfg.pair1[p] < 0 || d->cfg.pair1[p] > 2) {
      stpde_set_error("layer desc: bad second-order pair %d", p);
      return STPDE_E_BADARG;
    }
  return STPDE_OK;
}

extern "C" int stpde_jet_layer_fwd(const stpde_layer_desc* d, const float* in_pre, const float* X,
                                   const float* Wh_pack, const float* Ws_pack, const float* tanc,
                                   const float* W0s_pack, const float* tanc0, float* out_pre, const float* cw,
                                   const void* Wh_pack_bf16, float* z0, void* stream) {
  int rc = check_cfg(d);
  if (rc) return rc;
  LayerArgs a{};
  a.Bin = in_pre;
  a.Wp = Wh_pack;
  a.X = X;
  a.W0s = W0s_pack;
  a.tanc0 = tanc0;
  a.Wsp = Ws_pack;
  a.tanc = tanc;
  a.Out = out_pre;
  a.cw = cw;
  a.Wp16 = d->mfma_bf16 ? Wh_pack_bf16 : nullptr;
  a.nsplit = d->mfma_bf16 == 3 ? 3 : 1;
  a.KT = d->KT;
  a.MT = d->MT;
  a.ntiles = d->ntiles;
  a.cfg = d->cfg;
  a.pk = d->packed & 3;      // 1: in_pre packed, 2: out_pre packed
  if (!X || !Ws_pack || !out_pre || (d->cfg.S1 && !tanc)) {
    stpde_set_error("jet_layer_fwd: null pointer");
    return STPDE_E_BADARG;
  }
  if (d->first_hidden) {
    if (!W0s_pack || (d->cfg.S1 && !tanc0)) {
      stpde_set_error("jet_layer_fwd: first_hidden needs W0s_pack/tanc0");
      return STPDE_E_BADARG;
    }
    a.Z0 = z0;
    return dispatch_streams(a, 1, (hipStream_t)stream);
  }
  if (d->KT > 0 && (!in_pre || !Wh_pack)) {
    stpde_set_error("jet_layer_fwd: null hidden input");
    return STPDE_E_BADARG;
  }
  return dispatch_streams(a, 0, (hipStream_t)stream);
}

static int layer_bwd(const stpde_layer_desc* d, const float* abar_out, const float* WhT_pack, const float* in_pre,
                     float* abar_in, const float* X, const float* W0s_pack, const float* tanc0, float* abar0,
                     const float* cw, float* act_param_bar, const void* WhT_pack_bf16, float* abar0_tan, const float* z0,
                     void* stream);

extern "C" int stpde_jet_layer_bwd(const stpde_layer_desc* d, const float* abar_out, const float* WhT_pack,
                                   float* in_pre, const float* X, const float* W0s_pack, const float* tanc0,
                                   float* abar0, const float* cw, float* act_param_bar,
                                   const void* WhT_pack_bf16, float* abar0_tan, const float* z0, void* stream) {
  return layer_bwd(d, abar_out, WhT_pack, in_pre, in_pre, X, W0s_pack, tanc0, abar0, cw, act_param_bar, WhT_pack_bf16,
                   abar0_tan, z0, stream);
}

extern "C" int stpde_jet_layer_bwd_to(const stpde_layer_desc* d, const float* abar_out, const float* WhT_pack,
                                      const float* in_pre, float* abar_in, const float* cw, float* act_param_bar,
                                      const void* WhT_pack_bf16, void* stream) {
  if (!d || d->first_hidden || !abar_in) {
    stpde_set_error("jet_layer_bwd_to: hidden layers only (the first hidden layer already writes to abar0), abar_in required");
    return STPDE_E_BADARG;
  }
  return layer_bwd(d, abar_out, WhT_pack, in_pre, abar_in, nullptr, nullptr, nullptr, nullptr, cw, act_param_bar,
                   WhT_pack_bf16, nullptr, nullptr, stream);
}

static int layer_bwd(const stpde_layer_desc* d, const float* abar_out, const float* WhT_pack, const float* in_pre,
                     float* abar_in, const float* X, const float* W0s_pack, const float* tanc0, float* abar0,
                     const float* cw, float* act_param_bar, const void* WhT_pack_bf16, float* abar0_tan, const float* z0,
                     void* stream) {
  int rc = check_cfg(d);
  if (rc) return rc;
  // GEMM roles swap: contraction over this layer's MT output tiles, result over its KT input tiles.
  LayerArgs a{};
  a.Bin = abar_out;
  a.Wp = WhT_pack;
  a.X = X;
  a.W0s = W0s_pack;
  a.tanc0 = tanc0;
  a.cw = cw;
  a.pbar = act_param_bar;
  a.Wp16 = d->mfma_bf16 ? WhT_pack_bf16 : nullptr;
  a.nsplit = d->mfma_bf16 == 3 ? 3 : 1;
  a.KT = d->MT;
  a.MT = d->KT;
  a.ntiles = d->ntiles;
  a.cfg = d->cfg;
  // abar_out is this kernel's B operand, in_pre what the adjoint is taken against, abar_in where it goes
  a.pk = ((d->packed & 4) ? 1 : 0) | ((d->packed & 2) ? 2 : 0) | ((d->packed & 1) ? 4 : 0);
  if (!abar_out || !WhT_pack || d->KT <= 0) {
    stpde_set_error("jet_layer_bwd: null pointer / no hidden input");
    return STPDE_E_BADARG;
  }
  if (d->first_hidden) {
    if (!z0 || !abar0 || (d->cfg.S1 && !tanc0)) {
      stpde_set_error("jet_layer_bwd: first_hidden needs z0/tanc0/abar0");
      return STPDE_E_BADARG;
    }
    if (z0 == abar0 && d->cfg.S1 && !abar0_tan) {
      stpde_set_error("jet_layer_bwd: z0 may alias abar0 only when abar0 holds the value stream alone");
      return STPDE_E_BADARG;
    }
    a.Out = abar0;
    a.Tan0 = abar0_tan;
    a.Z0 = const_cast<float*>(z0);
    return dispatch_streams(a, 3, (hipStream_t)stream);
  }
  if (!in_pre || !abar_in) {
    stpde_set_error("jet_layer_bwd: null in_pre");
    return STPDE_E_BADARG;
  }
  a.Out = abar_in;
  a.Pre = in_pre;
  return dispatch_streams(a, 2, (hipStream_t)stream);
}
