// Weight gradient of one IM-NET layer over all derivative streams:
//   dW_aug[m][k] += sum_{rows, streams} abar[row][m] * hin[row][k],   hin = [act_jet(in_pre) ; X_aug]
// (the contribution loss.backward() -- experiments/rb2d/train.py:77 -- makes to fc_l.weight / fc_l.bias through
// src/implicit_net.py:48-54, including the second-order terms of the src/pde.py:8-9 sweeps).
//
// Contraction is over rows, but fragment blocks hold rows on lanes, so both operands are transposed through a
// wave-private LDS patch (16x16 floats, row stride 17).  Each wave owns a MCW x KCW block of 16x16 output tiles
// and walks a strided subset of the row tiles; partial sums are merged with fp32 atomics at the end.
#include "common.h"

struct WgradArgs {
  const float* P;      // abar_out [tile][SP][MT][256]
  const float* Q;      // in_pre   [tile][S][KT][256]   (MODE 0)
  const float* X;      // [tile][XT][256]
  const float* W0s;    // [XT][KT][256]  (MODE 1)
  const float* tanc0;  // [3][KT][256]   (MODE 1)
  float* dW;           // [16*MT][16*(KT+XT)]
  int SP, KT, MT, ntiles;
  stpde_jet_cfg cfg;
};

constexpr int TS = 17;  // padded row stride of the transpose patch

// D-image block v (lane 16g+j: row j, features 4g..4g+3) -> operand image (lane 16k+i, step s: row 4s+k, feature i)
__device__ __forceinline__ void transpose_block(float* patch, int lane, f32x4 v, float* out) {
  const int g = lane >> 4, j = lane & 15;
  float* w = patch + j * TS + 4 * g;
  w[0] = v[0];
  w[1] = v[1];
  w[2] = v[2];
  w[3] = v[3];
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int s = 0; s < 4; ++s) out[s] = patch[(4 * s + g) * TS + j];
  __builtin_amdgcn_wave_barrier();
}

template <int S1, int S2, int MODE, int MCW, int KCW>
__global__ __launch_bounds__(256) void k_wgrad(WgradArgs a) {
  constexpr int S = 1 + S1 + S2;
  __shared__ float lds[4][S][16 * TS];
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

  for (int tile = blockIdx.x * 4 + wv; tile < a.ntiles; tile += gridDim.x * 4) {
    // A operands: abar blocks of this wave's m-tiles, all streams, transposed
    float pa[S][MCW][4];
#pragma unroll
    for (int st = 0; st < S; ++st) {
#pragma unroll
      for (int mi = 0; mi < MCW; ++mi) {
        const int mt = mt0 + mi;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (mt < MT && st < SP) v = ld4(a.P + (((size_t)tile * SP + st) * MT + mt) * 256 + lo);
        transpose_block(lds[wv][st], lane, v, pa[st][mi]);
      }
    }
    f32x4 xb[XT];
#pragma unroll
    for (int xt = 0; xt < XT; ++xt) xb[xt] = ld4(a.X + ((size_t)tile * XT + xt) * 256 + lo);

#pragma unroll
    for (int ki = 0; ki < KCW; ++ki) {
      const int kq = kq0 + ki;
      if (kq >= KT + XT) continue;
      f32x4 H[S];
      int nst = S;  // streams with a non-zero input block
      if (kq < KT) {
        f32x4 pre[S];
        if (MODE == 0) {
#pragma unroll
          for (int st = 0; st < S; ++st) pre[st] = ld4(a.Q + (((size_t)tile * S + st) * KT + kq) * 256 + lo);
        } else {
          f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int xt = 0; xt < XT; ++xt) {
            f32x4 w = ld4(a.W0s + ((size_t)xt * KT + kq) * 256 + lo);
#pragma unroll
            for (int r = 0; r < 4; ++r) a0 = mfma4(w[r], xb[xt][r], a0);
          }
          pre[0] = a0;
          if (S1 == 3) {
#pragma unroll
            for (int d = 0; d < 3; ++d) pre[1 + d] = ld4(a.tanc0 + ((size_t)d * KT + kq) * 256 + lo);
#pragma unroll
            for (int p = 0; p < S2; ++p) pre[4 + p] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
        act_jet_fwd<S1, S2, -1>(a.cfg, pre, H);
      } else {
        // raw-input part: value stream = X_aug block, tangent stream d = unit vector e_d, second order = 0
        const int xt = kq - KT;
        H[0] = xt == 0 ? xb[0] : (xt == 1 ? xb[1] : xb[2]);
        nst = 1;
        if (S1 == 3 && xt == 0) {
          nst = 4;
#pragma unroll
          for (int d = 0; d < 3; ++d) {
            f32x4 e = f32x4{0.f, 0.f, 0.f, 0.f};
            if (g == 0) e[d] = 1.f;
            H[1 + d] = e;
          }
        }
      }
#pragma unroll
      for (int st = 0; st < S; ++st) {
        if (st < nst) {
          float qb[4];
          transpose_block(lds[wv][st], lane, H[st], qb);
#pragma unroll
          for (int mi = 0; mi < MCW; ++mi)
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[mi][ki] = mfma4(pa[st][mi][s], qb[s], acc[mi][ki]);
        }
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

template <int S1, int S2, int MODE>
static int launch_wgrad(const WgradArgs& a, hipStream_t stream) {
  constexpr int MCW = 4, KCW = 8;
  const int gy = (a.MT + MCW - 1) / MCW, gz = (a.KT + XT + KCW - 1) / KCW;
  int gx = 2048 / (gy * gz);  // ~2048 workgroups in flight overall
  if (gx < 1) gx = 1;
  const int maxx = (a.ntiles + 3) / 4;
  if (gx > maxx) gx = maxx;
  hipLaunchKernelGGL((k_wgrad<S1, S2, MODE, MCW, KCW>), dim3(gx, gy, gz), dim3(256), 0, stream, a);
  return stpde_check_launch("k_wgrad");
}

template <int MODE>
static int dispatch_streams(const WgradArgs& a, hipStream_t stream) {
  const int S1 = a.cfg.S1, S2 = a.cfg.S2;
  if (S1 == 0 && S2 == 0) return launch_wgrad<0, 0, MODE>(a, stream);
  if (S1 == 3 && S2 == 0) return launch_wgrad<3, 0, MODE>(a, stream);
  if (S1 == 3 && S2 == 2) return launch_wgrad<3, 2, MODE>(a, stream);
  if (S1 == 3 && S2 == 6) return launch_wgrad<3, 6, MODE>(a, stream);
  stpde_set_error("stream configuration S1=%d S2=%d not compiled", S1, S2);
  return STPDE_E_UNSUPPORTED;
}

extern "C" int stpde_jet_wgrad(const stpde_layer_desc* d, int SP, const float* abar_out, const float* in_pre,
                               const float* X, const float* W0s_pack, const float* tanc0, float* dW_aug,
                               void* stream) {
  if (!d || d->ntiles <= 0 || d->MT <= 0 || d->KT < 0 || !abar_out || !X || !dW_aug || SP < 1 ||
      SP > 1 + d->cfg.S1 + d->cfg.S2) {
    stpde_set_error("jet_wgrad: bad argument");
    return STPDE_E_BADARG;
  }
  WgradArgs a{};
  a.P = abar_out;
  a.Q = in_pre;
  a.X = X;
  a.W0s = W0s_pack;
  a.tanc0 = tanc0;
  a.dW = dW_aug;
  a.SP = SP;
  a.KT = d->KT;
  a.MT = d->MT;
  a.ntiles = d->ntiles;
  a.cfg = d->cfg;
  if (d->first_hidden) {
    if (!W0s_pack || (d->cfg.S1 && !tanc0)) {
      stpde_set_error("jet_wgrad: first_hidden needs W0s_pack/tanc0");
      return STPDE_E_BADARG;
    }
    return dispatch_streams<1>(a, (hipStream_t)stream);
  }
  if (d->KT > 0 && !in_pre) {
    stpde_set_error("jet_wgrad: null in_pre");
    return STPDE_E_BADARG;
  }
  return dispatch_streams<0>(a, (hipStream_t)stream);
}
