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
#pragma once
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
  int gx, gy, gz;      // logical grid: tile splits (multiple of 8), m-blocks, k-blocks
  int kz0;             // first k-block of this launch (k_wgrad_first is launched per k-block class)
  stpde_jet_cfg cfg;
};

template <int S1, int S2, int MODE, int ACT, int MCW, int KCW>
__global__ __launch_bounds__(256, (MODE == 0 && KCW == 8 && S1 + S2 <= 5) ? 2 : 1) void k_wgrad(WgradArgs a) {
  constexpr int S = 1 + S1 + S2;
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int lo = lane * 4;
  const int KT = a.KT, MT = a.MT, SP = a.SP;
  // XCD-aware block order: the dispatcher places block b on XCD b % 8.  All k-blocks of one (tile range, m-block)
  // get consecutive slots of the SAME XCD, so they stream the same abar rows at the same time and share them in
  // that XCD's L2 instead of each pulling them from HBM.
  const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
  const int kz = slot % a.gz, tq = slot / a.gz;
  const int my = tq % a.gy, bx = (tq / a.gy) * 8 + xcd;
  const int mt0 = my * MCW;
  const int kq0 = kz * KCW;
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

  for (int tile = bx * 4 + wv; tile < a.ntiles; tile += a.gx * 4) {
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
      // activated input block of k-tile kq (hidden tiles: layer 0 regenerated from the raw input; raw-input tiles:
      // value stream = X_aug, tangent stream d = e_d in the first tile, everything else zero)
      // The layer-0 operand blocks do not depend on the tile: launder the pointers so that LICM does not hoist all
      // KCW * 6 loads out of the tile loop and pin > 200 registers (spills); they are L1/L2 hits anyway.
      const float* w0s = a.W0s;
      const float* tcr = a.tancR;
      asm volatile("" : "+s"(w0s), "+s"(tcr));
      auto make_H = [&](int kq, f32x4* H) {
        if (kq < KT) {
          f32x4 pre[S];
          f32x4 part[XT];
#pragma unroll
          for (int xt = 0; xt < XT; ++xt) {
            f32x4 w = ld4(w0s + ((size_t)xt * KT + kq) * 256 + lo);
            f32x4 cc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) cc = mfma4(xd[xt][r], w[r], cc);   // rows x features: R image
            part[xt] = cc;
          }
          pre[0] = (part[0] + part[1]) + part[2];
          if (S1 == 3) {
#pragma unroll
            for (int d = 0; d < 3; ++d) pre[1 + d] = ld4(tcr + ((size_t)d * KT + kq) * 256 + lo);
#pragma unroll
            for (int p = 0; p < S2; ++p) pre[4 + p] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
          act_jet_fwd<S1, S2, ACT>(a.cfg, pre, H);
        } else {
          const int xt = kq - KT;
#pragma unroll
          for (int st = 0; st < S; ++st) H[st] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (xt < XT) {
            H[0] = xt == 0 ? xr[0] : (xt == 1 ? xr[1] : xr[2]);
            if (S1 == 3 && xt == 0) {
#pragma unroll
              for (int d = 0; d < 3; ++d) H[1 + d] = ed[d];
            }
          }
        }
      };
      // software pipeline over the k-tiles of the block: the regeneration + activation jet of tile ki+1 is issued
      // next to the MFMAs of tile ki; the scheduling barrier keeps later tiles' loads from being hoisted (spills)
      f32x4 Hc[S];
      make_H(kq0, Hc);
#pragma unroll
      for (int ki = 0; ki < KCW; ++ki) {
        f32x4 Hn[S];
        if (ki + 1 < KCW) make_H(kq0 + ki + 1, Hn);
        const bool hidden = kq0 + ki < KT;
        if (hidden) {
#pragma unroll
          for (int mi = 0; mi < MCW; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
              for (int st = 0; st < S; ++st) acc[mi][ki] = mfma4(pa[st][mi][r], Hc[st][r], acc[mi][ki]);
        } else {
          constexpr int SX = S1 == 3 ? 4 : 1;   // raw-input tiles only feed the value and tangent streams
#pragma unroll
          for (int mi = 0; mi < MCW; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
              for (int st = 0; st < SX; ++st) acc[mi][ki] = mfma4(pa[st][mi][r], Hc[st][r], acc[mi][ki]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ki + 1 < KCW) {
#pragma unroll
          for (int st = 0; st < S; ++st) Hc[st] = Hn[st];
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


// ------------------------------------------------------------------------------------------------------------
// First hidden layer (its activated input is regenerated from the raw input): workgroup-cooperative variant.
// The 4 waves of a workgroup own the 4 m-blocks (16 output tiles) of ONE k-block and walk the same row tiles in
// lock step.  The regenerated + activated input blocks of a tile (KCW k-tiles x S streams, R image) are produced
// ONCE per workgroup -- wave w produces k-tiles w, w+4, .. -- into LDS and consumed by all four waves, so the
// activation VALU work and the layer-0 MFMAs are paid once per 16 output tiles instead of once per 4.  Production
// for tile t+1 is issued in the same basic block as the MFMAs of tile t (double-buffered LDS, one barrier per tile).
// ------------------------------------------------------------------------------------------------------------
template <int S1, int S2, int ACT, int KCW, bool HASX>
__global__ __launch_bounds__(512, 2) void k_wgrad_first(WgradArgs a) {
  // 8 waves (2 per SIMD) x 2 output tiles each = the 16 output tiles of one k-block
  constexpr int S = 1 + S1 + S2, MCW = 2, NW = 8;
  constexpr int NBUF = (2 * KCW * S * 1024 <= 112 * 1024) ? 2 : 1;
  constexpr int NP = (KCW + NW - 1) / NW;  // k-tiles produced per wave per row tile
  __shared__ __attribute__((aligned(16))) float hl[NBUF][KCW][S][256];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int lo = lane * 4;
  const int KT = a.KT, MT = a.MT, SP = a.SP;
  const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
  const int kz = (slot % a.gz) + a.kz0, tq = slot / a.gz;
  const int mg = tq % a.gy, bx = (tq / a.gy) * 8 + xcd;
  const int mt0 = (mg * NW + wv) * MCW;
  const int kq0 = kz * KCW;
  const int g = lane >> 4, c = lane & 15;
  constexpr bool has_x = HASX;  // this k-block contains raw-input tiles (the launcher splits the grid)

  f32x4 acc[MCW][KCW];
#pragma unroll
  for (int mi = 0; mi < MCW; ++mi)
#pragma unroll
    for (int ki = 0; ki < KCW; ++ki) acc[mi][ki] = f32x4{0.f, 0.f, 0.f, 0.f};

  const float* w0s = a.W0s;
  const float* tcr = a.tancR;

  // produce this wave's share of the activated input blocks of `tile` into buffer `buf`
  auto produce = [&](int tile, int buf) {
    f32x4 xd[XT];
#pragma unroll
    for (int xt = 0; xt < XT; ++xt) xd[xt] = ld4(a.X + ((size_t)tile * XT + xt) * 256 + lo);
#pragma unroll
    for (int pi = 0; pi < NP; ++pi) {
      const int ki = wv + NW * pi;
      if (KCW % NW != 0 && ki >= KCW) continue;
      const int kq = kq0 + ki;
      f32x4 H[S];
      if (!has_x || kq < KT) {
        f32x4 pre[S], part[XT];
#pragma unroll
        for (int xt = 0; xt < XT; ++xt) {
          f32x4 w = ld4(w0s + ((size_t)xt * KT + kq) * 256 + lo);
          f32x4 cc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int r = 0; r < 4; ++r) cc = mfma4(xd[xt][r], w[r], cc);   // rows x features: R image
          part[xt] = cc;
        }
        pre[0] = (part[0] + part[1]) + part[2];
        if (S1 == 3) {
#pragma unroll
          for (int d = 0; d < 3; ++d) pre[1 + d] = ld4(tcr + ((size_t)d * KT + kq) * 256 + lo);
#pragma unroll
          for (int p = 0; p < S2; ++p) pre[4 + p] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        act_jet_fwd<S1, S2, ACT>(a.cfg, pre, H);
      } else {
        const int xt = kq - KT;
#pragma unroll
        for (int st = 0; st < S; ++st) H[st] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (xt < XT) {
          H[0] = ld4(a.XR + ((size_t)tile * XT + xt) * 256 + lo);
          if (S1 == 3 && xt == 0) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
              const float v = c == d ? 1.f : 0.f;
              H[1 + d] = f32x4{v, v, v, v};
            }
          }
        }
      }
#pragma unroll
      for (int st = 0; st < S; ++st) st4(&hl[buf][ki][st][lo], H[st]);
    }
  };
  auto load_p = [&](int tile, f32x4 (*pa)[MCW]) {
    const float* pbase = a.P + (size_t)tile * SP * MT * 256 + lo;
#pragma unroll
    for (int st = 0; st < S; ++st)
#pragma unroll
      for (int mi = 0; mi < MCW; ++mi) {
        const int mt = mt0 + mi < MT ? mt0 + mi : MT - 1;
        pa[st][mi] = st < SP ? ld4(pbase + ((size_t)st * MT + mt) * 256) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
  };

  const int stride = a.gx;
  int tile = bx;
  f32x4 pa[S][MCW];
  if (tile < a.ntiles) {
    produce(tile, 0);
    load_p(tile, pa);
  }
  __syncthreads();
  int buf = 0;
  for (; tile < a.ntiles; tile += stride) {
    const int next = tile + stride;
    const bool more = next < a.ntiles;
    // consume: all k-tiles of this row tile from LDS against the resident abar blocks
#pragma unroll
    for (int ki = 0; ki < KCW; ++ki) {
      f32x4 H[S];
#pragma unroll
      for (int st = 0; st < S; ++st) H[st] = ld4(&hl[buf][ki][st][lo]);
      if (!has_x || kq0 + ki < KT) {
#pragma unroll
        for (int mi = 0; mi < MCW; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int st = 0; st < S; ++st) acc[mi][ki] = mfma4(pa[st][mi][r], H[st][r], acc[mi][ki]);
      } else {
        constexpr int SX = S1 == 3 ? 4 : 1;   // raw-input tiles only feed the value and tangent streams
#pragma unroll
        for (int mi = 0; mi < MCW; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int st = 0; st < SX; ++st) acc[mi][ki] = mfma4(pa[st][mi][r], H[st][r], acc[mi][ki]);
      }
    }
    // Branch-free tail (the last iteration harmlessly re-produces its own tile): keeping the whole iteration in ONE
    // basic block lets the scheduler interleave the produce stage's VALU work with the MFMAs above.
    const int nx = more ? next : tile;
    load_p(nx, pa);   // the next tile's abar blocks land while the produce stage below runs
    if (NBUF == 2) {
      produce(nx, buf ^ 1);
      __syncthreads();
      buf ^= 1;
    } else {
      __syncthreads();
      produce(nx, 0);
      __syncthreads();
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

template <int S1, int S2, int ACT>
static int launch_wgrad_first(const WgradArgs& a0, hipStream_t stream) {
  constexpr int KCW = (1 + S1 + S2) > 6 ? 4 : 8;   // LDS: KCW * S KiB per buffer
  WgradArgs a = a0;
  a.gy = (a.MT + 15) / 16;                 // groups of 16 output tiles (2 per wave)
  const int nkb = (a.KT + XT + KCW - 1) / KCW;
  const int nhid = a.KT / KCW;             // k-blocks made of hidden tiles only
  for (int part = 0; part < 2; ++part) {
    a.kz0 = part == 0 ? 0 : nhid;
    a.gz = part == 0 ? nhid : nkb - nhid;
    if (a.gz <= 0) continue;
    int gx = 512 / (a.gy * a.gz);          // ~2 rounds of two workgroups per CU
    if (gx > a.ntiles) gx = a.ntiles;
    gx = (gx + 7) / 8 * 8;
    if (gx < 8) gx = 8;
    a.gx = gx;
    if (part == 0)
      STPDE_LAUNCH((k_wgrad_first<S1, S2, ACT, KCW, false>), dim3(gx * a.gy * a.gz), dim3(512), 0, stream, a);
    else
      STPDE_LAUNCH((k_wgrad_first<S1, S2, ACT, KCW, true>), dim3(gx * a.gy * a.gz), dim3(512), 0, stream, a);
    int rc = stpde_check_launch("k_wgrad_first");
    if (rc) return rc;
  }
  return STPDE_OK;
}

template <int S1, int S2, int MODE, int ACT, int KCW>
static int launch_wgrad(const WgradArgs& a0, hipStream_t stream) {
  constexpr int MCW = 4;
  WgradArgs a = a0;
  a.gy = (a.MT + MCW - 1) / MCW;
  a.gz = (a.KT + XT + KCW - 1) / KCW;
  int gx = 2048 / (a.gy * a.gz);  // ~2048 workgroups overall (8 rounds of one workgroup per CU)
  const int maxx = (a.ntiles + 3) / 4;
  if (gx > maxx) gx = maxx;
  gx = (gx + 7) / 8 * 8;  // one slice per XCD
  if (gx < 8) gx = 8;
  a.gx = gx;
  STPDE_LAUNCH((k_wgrad<S1, S2, MODE, ACT, MCW, KCW>), dim3(gx * a.gy * a.gz), dim3(256), 0, stream, a);
  return stpde_check_launch("k_wgrad");
}

// k-block width with the least padding of the KT + XT input tiles (ties -> wider)
static inline int pick_kcw(int ktot) {
  int best = 8, waste = 1 << 30;
  for (int k = 8; k <= 10; ++k) {
    const int w = (ktot + k - 1) / k * k - ktot;
    if (w <= waste) {
      waste = w;
      best = k;
    }
  }
  return best;
}

template <int S1, int S2, int MODE, int ACT>
static int launch_kcw(const WgradArgs& a, hipStream_t stream) {
  switch (pick_kcw(a.KT + XT)) {
    case 8: return launch_wgrad<S1, S2, MODE, ACT, 8>(a, stream);
    case 9: return launch_wgrad<S1, S2, MODE, ACT, 9>(a, stream);
    default: return launch_wgrad<S1, S2, MODE, ACT, 10>(a, stream);
  }
}

template <int S1, int S2>
static int launch_mode(const WgradArgs& a, int mode, hipStream_t stream) {
  if (mode == 0) return launch_kcw<S1, S2, 0, -1>(a, stream);
  switch (a.cfg.act) {
    case STPDE_ACT_TANH: return launch_wgrad_first<S1, S2, STPDE_ACT_TANH>(a, stream);
    case STPDE_ACT_RELU: return launch_wgrad_first<S1, S2, STPDE_ACT_RELU>(a, stream);
    case STPDE_ACT_SOFTPLUS: return launch_wgrad_first<S1, S2, STPDE_ACT_SOFTPLUS>(a, stream);
    case STPDE_ACT_ELU: return launch_wgrad_first<S1, S2, STPDE_ACT_ELU>(a, stream);
    case STPDE_ACT_LEAKYRELU: return launch_wgrad_first<S1, S2, STPDE_ACT_LEAKYRELU>(a, stream);
    default: return launch_wgrad_first<S1, S2, STPDE_ACT_SWISH>(a, stream);
  }
}

#define STPDE_DEFINE_WGRAD_TU(S1, S2) \
  int stpde_wgrad_launch_##S1##_##S2(const WgradArgs& a, int mode, hipStream_t stream) { return launch_mode<S1, S2>(a, mode, stream); }

int stpde_wgrad_launch_0_0(const WgradArgs& a, int mode, hipStream_t stream);
int stpde_wgrad_launch_3_0(const WgradArgs& a, int mode, hipStream_t stream);
int stpde_wgrad_launch_3_2(const WgradArgs& a, int mode, hipStream_t stream);
int stpde_wgrad_launch_3_6(const WgradArgs& a, int mode, hipStream_t stream);
