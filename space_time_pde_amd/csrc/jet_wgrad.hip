// C-ABI entry point of the weight-gradient kernels (implementation: jet_wgrad_impl.h, one TU per stream config).
#include "jet_wgrad_impl.h"

static int dispatch_streams(const WgradArgs& a, int mode, hipStream_t stream) {
  const int S1 = a.cfg.S1, S2 = a.cfg.S2;
  if (S1 == 0 && S2 == 0) return stpde_wgrad_launch_0_0(a, mode, stream);
  if (S1 == 3 && S2 == 0) return stpde_wgrad_launch_3_0(a, mode, stream);
  if (S1 == 3 && S2 == 1 && a.cfg.combo && a.cw) return stpde_wgrad_launch_3_1(a, mode, stream);
  if (S1 == 3 && S2 == 2) return stpde_wgrad_launch_3_2(a, mode, stream);
  if (S1 == 3 && S2 == 4) return stpde_wgrad_launch_3_4(a, mode, stream);
  if (S1 == 3 && S2 == 6) return stpde_wgrad_launch_3_6(a, mode, stream);
  stpde_set_error("stream configuration S1=%d S2=%d not compiled", S1, S2);
  return STPDE_E_UNSUPPORTED;
}

extern "C" int stpde_jet_wgrad(const stpde_layer_desc* d, int SP, const float* abar_out, const float* in_pre,
                               const float* X, const float* tanc0, float* dW_aug, const float* cw, void* stream) {
  if (!d || d->ntiles <= 0 || d->MT <= 0 || d->KT < 0 || !abar_out || !X || !dW_aug || SP < 1 ||
      SP > 1 + d->cfg.S1 + d->cfg.S2) {
    stpde_set_error("jet_wgrad: bad argument");
    return STPDE_E_BADARG;
  }
  WgradArgs a{};
  a.P = abar_out;
  a.Q = in_pre;
  a.X = X;
  a.tanc0 = tanc0;
  a.dW = dW_aug;
  a.cw = cw;
  a.SP = SP;
  a.KT = d->KT;
  a.MT = d->MT;
  a.ntiles = d->ntiles;
  a.cfg = d->cfg;
  a.bf16 = d->mfma_bf16;
  a.pk = d->packed & 5;      // 1: in_pre (Q) packed, 4: abar_out (P) packed
  a.det = d->det;
  if (d->KT > 0 && !in_pre) {
    stpde_set_error("jet_wgrad: null in_pre");
    return STPDE_E_BADARG;
  }
  if (d->first_hidden) {
    if (d->cfg.S1 && !tanc0) {
      stpde_set_error("jet_wgrad: first_hidden needs tanc0");
      return STPDE_E_BADARG;
    }
    return dispatch_streams(a, 1, (hipStream_t)stream);
  }
  return dispatch_streams(a, 0, (hipStream_t)stream);
}


// d W0[:, d] += sum over tiles of the row-reduced tangent-stream adjoints of layer 0 (written by the layer-1 dgrad
// epilogue): [tile][MT][3][16] -> one column per d.  Grid-stride partial sums, one atomic per (block, element).
__global__ __launch_bounds__(256) void k_tan0_reduce(const float* tan, float* dW, int ntiles, int MT, int ldw, int det) {
  const int n = MT * 48;
  const int e = blockIdx.y * 256 + threadIdx.x;     // grid.y covers the n elements of a tile, grid.x strides the tiles
  if (e >= n) return;
  // four independent partial sums: four loads in flight per thread (a single running sum serialises on the HBM latency)
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const int G = gridDim.x;
  int t = blockIdx.x;
  for (; t + 3 * G < ntiles; t += 4 * G) {
    s0 += tan[(size_t)t * n + e];
    s1 += tan[(size_t)(t + G) * n + e];
    s2 += tan[(size_t)(t + 2 * G) * n + e];
    s3 += tan[(size_t)(t + 3 * G) * n + e];
  }
  for (; t < ntiles; t += G) s0 += tan[(size_t)t * n + e];
  const int mt = e / 48, d = (e % 48) / 16, f = e % 16;
  acc_add_f32(dW, (size_t)(16 * mt + f) * ldw + d, (s0 + s1) + (s2 + s3), det);
}

extern "C" int stpde_jet_tan0_reduce(int ntiles, int MT, const float* abar0_tan, float* dW_aug, int ldw, int det, void* stream) {
  if (ntiles <= 0 || MT <= 0 || !abar0_tan || !dW_aug || ldw < 3) {
    stpde_set_error("jet_tan0_reduce: bad argument");
    return STPDE_E_BADARG;
  }
  const int gy = (MT * 48 + 255) / 256;
  int gx = 4096 / gy;                        // ~16 blocks per CU
  if (gx > ntiles) gx = ntiles;
  STPDE_LAUNCH(k_tan0_reduce, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, abar0_tan, dW_aug, ntiles, MT, ldw, det);
  return stpde_check_launch("k_tan0_reduce");
}
