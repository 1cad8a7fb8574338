// C-ABI entry point of the weight-gradient kernels (implementation: jet_wgrad_impl.h, one TU per stream config).
#include "jet_wgrad_impl.h"

static int dispatch_streams(const WgradArgs& a, int mode, hipStream_t stream) {
  const int S1 = a.cfg.S1, S2 = a.cfg.S2;
  if (S1 == 0 && S2 == 0) return stpde_wgrad_launch_0_0(a, mode, stream);
  if (S1 == 3 && S2 == 0) return stpde_wgrad_launch_3_0(a, mode, stream);
  if (S1 == 3 && S2 == 1 && a.cfg.combo && a.cw) return stpde_wgrad_launch_3_1(a, mode, stream);
  if (S1 == 3 && S2 == 2) return stpde_wgrad_launch_3_2(a, mode, stream);
  if (S1 == 3 && S2 == 6) return stpde_wgrad_launch_3_6(a, mode, stream);
  stpde_set_error("stream configuration S1=%d S2=%d not compiled", S1, S2);
  return STPDE_E_UNSUPPORTED;
}

extern "C" int stpde_jet_wgrad(const stpde_layer_desc* d, int SP, const float* abar_out, const float* in_pre,
                               const float* X, const float* XR, const float* W0s_pack, const float* tanc0R,
                               float* dW_aug, const float* cw, void* stream) {
  if (!d || d->ntiles <= 0 || d->MT <= 0 || d->KT < 0 || !abar_out || !XR || !dW_aug || SP < 1 ||
      SP > 1 + d->cfg.S1 + d->cfg.S2) {
    stpde_set_error("jet_wgrad: bad argument");
    return STPDE_E_BADARG;
  }
  WgradArgs a{};
  a.P = abar_out;
  a.Q = in_pre;
  a.X = X;
  a.XR = XR;
  a.W0s = W0s_pack;
  a.tancR = tanc0R;
  a.dW = dW_aug;
  a.cw = cw;
  a.SP = SP;
  a.KT = d->KT;
  a.MT = d->MT;
  a.ntiles = d->ntiles;
  a.cfg = d->cfg;
  a.bf16 = d->mfma_bf16;
  if (d->first_hidden) {
    if (!X || !W0s_pack || (d->cfg.S1 && !tanc0R)) {
      stpde_set_error("jet_wgrad: first_hidden needs X/W0s_pack/tanc0R");
      return STPDE_E_BADARG;
    }
    return dispatch_streams(a, 1, (hipStream_t)stream);
  }
  if (d->KT > 0 && !in_pre) {
    stpde_set_error("jet_wgrad: null in_pre");
    return STPDE_E_BADARG;
  }
  return dispatch_streams(a, 0, (hipStream_t)stream);
}
