// Weight gradient of one IM-NET layer over all derivative streams:
//   dW_aug[m][k] += sum_{rows, streams} abar[row][m] * hin[row][k],   hin = [act_jet(in_pre) ; X_aug]
// (the contribution loss.backward() -- experiments/rb2d/train.py:77 -- makes to fc_l.weight / fc_l.bias through
// src/implicit_net.py:48-54, including the second-order terms of the src/pde.py:8-9 sweeps).
//
// The contraction runs over corner rows, so both operands are read in the ROW-MAJOR fragment image ("R layout":
// lane 16g+c holds rows 4g..4g+3 of feature c), which is exactly the A/B register image of
// v_mfma_f32_16x16x4_f32 for dW = P^T Q.  The dgrad kernels emit R-layout copies of abar and of the activated
// layer input next to the column-major images they chain on, so this kernel is a pure load -> MFMA stream
// (no LDS, no re-layout).  For the first hidden layer the activated input is regenerated on the fly from the raw
// input.  Each wave owns an MCW x KCW block of 16x16 output tiles and walks a strided subset of the row tiles;
// partial sums are merged with fp32 atomics at the end.
#include "common.h"

struct WgradArgs {
  const float* P;      // R(abar_out) [tile][SP][MT][256]
  const float* Q;      // R(act_jet(in_pre)) [tile][S][KT][256]   (MODE 0)
  const float* X;      // D-layout augmented input [tile][XT][256]   (MODE 1: A operand of the layer-0 regeneration)
  const float* XR;     // R-layout augmented input [tile][XT][256]
  const float* W0s;    // [XT][KT][256]  (MODE 1)
  const float* tancR;  // [3][KT][256] layer-0 tangent constants in R layout (MODE 1)
  float* dW;           // [16*MT][16*(KT+XT)]
  int SP, KT, MT, ntiles;
  stpde_jet_cfg cfg;
};

template <int S1, int S2, int MODE, int ACT, int MCW, int KCW>
__global__ __launch_bounds__(256) void k_wgrad(WgradArgs a) {
  constexpr int S = 1 + S1 + S2;
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int lo = lane * 4;
  const int KT = a.KT, MT = a.MT, SP = a.SP;
  const int mt0 = blockIdx.y * MCW;
  const int kq0 = blockIdx.z * KCW;
  const int g = lane >> 4, c = lane & 15;

  f32x4 acc[MCW][KCW];
#pragma unroll
  for (int mi = 0; mi < MCW; ++mi)
#pragma unroll
    for (int ki = 0; ki < KCW; ++ki) acc[mi][ki] = f32x4{0.f, 0.f, 0.f, 0.f};

  // unit vectors e_d in R layout (feature column d of the first raw-input tile), used by the tangent streams
  f32x4 ed[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float v = c == d ? 1.f : 0.f;
    ed[d] = f32x4{v, v, v, v};
  }

  for (int tile = blockIdx.x * 4 + wv; tile < a.ntiles; tile += gridDim.x * 4) {
    const float* pbase = a.P + (size_t)tile * SP * MT * 256 + lo;
    if (MODE == 1) {
      // all streams of P stay in registers; the activated input block is regenerated per k-tile
      f32x4 pa[S][MCW];
#pragma unroll
      for (int st = 0; st < S; ++st)
#pragma unroll
        for (int mi = 0; mi < MCW; ++mi) {
          const int mt = mt0 + mi < MT ? mt0 + mi : MT - 1;
          pa[st][mi] = st < SP ? ld4(pbase + ((size_t)st * MT + mt) * 256) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      f32x4 xd[XT], xr[XT];
#pragma unroll
      for (int xt = 0; xt < XT; ++xt) {
        xd[xt] = ld4(a.X + ((size_t)tile * XT + xt) * 256 + lo);
        xr[xt] = ld4(a.XR + ((size_t)tile * XT + xt) * 256 + lo);
      }
#pragma unroll
      for (int ki = 0; ki < KCW; ++ki) {
        const int kq = kq0 + ki;
        if (kq < KT) {
          f32x4 pre[S], H[S];
          f32x4 part[XT];
#pragma unroll
          for (int xt = 0; xt < XT; ++xt) {
            f32x4 w = ld4(a.W0s + ((size_t)xt * KT + kq) * 256 + lo);
            f32x4 cc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) cc = mfma4(xd[xt][r], w[r], cc);   // rows x features: R image
            part[xt] = cc;
          }
          pre[0] = (part[0] + part[1]) + part[2];
          if (S1 == 3) {
#pragma unroll
            for (int d = 0; d < 3; ++d) pre[1 + d] = ld4(a.tancR + ((size_t)d * KT + kq) * 256 + lo);
#pragma unroll
            for (int p = 0; p < S2; ++p) pre[4 + p] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
          act_jet_fwd<S1, S2, ACT>(a.cfg, pre, H);
#pragma unroll
          for (int mi = 0; mi < MCW; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
              for (int st = 0; st < S; ++st) acc[mi][ki] = mfma4(pa[st][mi][r], H[st][r], acc[mi][ki]);
        } else if (kq < KT + XT) {
          const int xt = kq - KT;
          const f32x4 q0 = xt == 0 ? xr[0] : (xt == 1 ? xr[1] : xr[2]);
#pragma unroll
          for (int mi = 0; mi < MCW; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mi][ki] = mfma4(pa[0][mi][r], q0[r], acc[mi][ki]);
          if (S1 == 3 && xt == 0) {
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
              for (int mi = 0; mi < MCW; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[mi][ki] = mfma4(pa[1 + d][mi][r], ed[d][r], acc[mi][ki]);
          }
        }
      }
    } else {
      const float* qbase = a.Q + (size_t)tile * S * KT * 256 + lo;
      const float* xrb = a.XR + (size_t)tile * XT * 256 + lo;
#pragma unroll
      for (int st = 0; st < S; ++st) {
        if (st >= SP) continue;
        f32x4 pa[MCW], qb[KCW];
#pragma unroll
        for (int mi = 0; mi < MCW; ++mi) {
          const int mt = mt0 + mi < MT ? mt0 + mi : MT - 1;
          pa[mi] = ld4(pbase + ((size_t)st * MT + mt) * 256);
        }
#pragma unroll
        for (int ki = 0; ki < KCW; ++ki) {
          const int kq = kq0 + ki;
          f32x4 q = f32x4{0.f, 0.f, 0.f, 0.f};
          if (kq < KT) {
            q = ld4(qbase + ((size_t)st * KT + kq) * 256);
          } else if (kq < KT + XT) {
            const int xt = kq - KT;
            if (st == 0) q = ld4(xrb + (size_t)xt * 256);
            if (S1 == 3 && xt == 0 && st >= 1 && st <= 3) q = st == 1 ? ed[0] : (st == 2 ? ed[1] : ed[2]);
          }
          qb[ki] = q;
        }
#pragma unroll
        for (int mi = 0; mi < MCW; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int ki = 0; ki < KCW; ++ki) acc[mi][ki] = mfma4(pa[mi][r], qb[ki][r], acc[mi][ki]);
      }
    }
  }

  const int ldw = 16 * (KT + XT);
#pragma unroll
  for (int mi = 0; mi < MCW; ++mi) {
    const int mt = mt0 + mi;
    if (mt >= MT) continue;
#pragma unroll
    for (int ki = 0; ki < KCW; ++ki) {
      const int kq = kq0 + ki;
      if (kq >= KT + XT) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        atomicAdd(a.dW + (size_t)(16 * mt + 4 * g + r) * ldw + 16 * kq + c, acc[mi][ki][r]);
    }
  }
}

template <int S1, int S2, int MODE, int ACT>
static int launch_wgrad(const WgradArgs& a, hipStream_t stream) {
  constexpr int MCW = 4, KCW = 8;
  const int gy = (a.MT + MCW - 1) / MCW, gz = (a.KT + XT + KCW - 1) / KCW;
  int gx = 2048 / (gy * gz);  // ~2048 workgroups in flight overall
  if (gx < 1) gx = 1;
  const int maxx = (a.ntiles + 3) / 4;
  if (gx > maxx) gx = maxx;
  hipLaunchKernelGGL((k_wgrad<S1, S2, MODE, ACT, MCW, KCW>), dim3(gx, gy, gz), dim3(256), 0, stream, a);
  return stpde_check_launch("k_wgrad");
}

template <int S1, int S2>
static int launch_mode(const WgradArgs& a, int mode, hipStream_t stream) {
  if (mode == 0) return launch_wgrad<S1, S2, 0, -1>(a, stream);
  switch (a.cfg.act) {
    case STPDE_ACT_TANH: return launch_wgrad<S1, S2, 1, STPDE_ACT_TANH>(a, stream);
    case STPDE_ACT_RELU: return launch_wgrad<S1, S2, 1, STPDE_ACT_RELU>(a, stream);
    case STPDE_ACT_SOFTPLUS: return launch_wgrad<S1, S2, 1, STPDE_ACT_SOFTPLUS>(a, stream);
    case STPDE_ACT_ELU: return launch_wgrad<S1, S2, 1, STPDE_ACT_ELU>(a, stream);
    case STPDE_ACT_LEAKYRELU: return launch_wgrad<S1, S2, 1, STPDE_ACT_LEAKYRELU>(a, stream);
    default: return launch_wgrad<S1, S2, 1, STPDE_ACT_SWISH>(a, stream);
  }
}

static int dispatch_streams(const WgradArgs& a, int mode, hipStream_t stream) {
  const int S1 = a.cfg.S1, S2 = a.cfg.S2;
  if (S1 == 0 && S2 == 0) return launch_mode<0, 0>(a, mode, stream);
  if (S1 == 3 && S2 == 0) return launch_mode<3, 0>(a, mode, stream);
  if (S1 == 3 && S2 == 2) return launch_mode<3, 2>(a, mode, stream);
  if (S1 == 3 && S2 == 6) return launch_mode<3, 6>(a, mode, stream);
  stpde_set_error("stream configuration S1=%d S2=%d not compiled", S1, S2);
  return STPDE_E_UNSUPPORTED;
}

extern "C" int stpde_jet_wgrad(const stpde_layer_desc* d, int SP, const float* abar_out_R, const float* hin_R,
                               const float* X, const float* XR, const float* W0s_pack, const float* tanc0R,
                               float* dW_aug, void* stream) {
  if (!d || d->ntiles <= 0 || d->MT <= 0 || d->KT < 0 || !abar_out_R || !XR || !dW_aug || SP < 1 ||
      SP > 1 + d->cfg.S1 + d->cfg.S2) {
    stpde_set_error("jet_wgrad: bad argument");
    return STPDE_E_BADARG;
  }
  WgradArgs a{};
  a.P = abar_out_R;
  a.Q = hin_R;
  a.X = X;
  a.XR = XR;
  a.W0s = W0s_pack;
  a.tancR = tanc0R;
  a.dW = dW_aug;
  a.SP = SP;
  a.KT = d->KT;
  a.MT = d->MT;
  a.ntiles = d->ntiles;
  a.cfg = d->cfg;
  if (d->first_hidden) {
    if (!X || !W0s_pack || (d->cfg.S1 && !tanc0R)) {
      stpde_set_error("jet_wgrad: first_hidden needs X/W0s_pack/tanc0R");
      return STPDE_E_BADARG;
    }
    return dispatch_streams(a, 1, (hipStream_t)stream);
  }
  if (d->KT > 0 && !hin_R) {
    stpde_set_error("jet_wgrad: null hin_R");
    return STPDE_E_BADARG;
  }
  return dispatch_streams(a, 0, (hipStream_t)stream);
}
