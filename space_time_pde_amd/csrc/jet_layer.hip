// One IM-NET layer on all derivative streams, forward and input-gradient (dgrad), as a per-wave MFMA GEMM.
//
// Each wave owns one tile of 16 corner rows (2 query points x 8 corners) and MC output feature tiles, for all
// S streams:   out^T[16*MC x 16] (per stream) = W[16*MC x K] * in^T[K x 16]
// A operand = packed weights (one float4 per lane per (k-tile, m-tile) block = the 4 k-steps of the block),
// B operand = the fragment block of the previous layer (C/D image == B image, so nothing is re-laid out),
// accumulators = MC*S float4 per lane.  No LDS, no barriers: waves are independent and the weights stream
// from L2 (every wave reads the same blocks).
//
// Replaces (reference): src/implicit_net.py:48-54 on the rows of src/local_implicit_grid.py:53, and the reverse
// sweeps of src/pde.py:8-9 (streams carry d/dr and d2/dr2 forward instead).
#include "common.h"

enum { PRO_NONE = 0, PRO_ACT = 1, PRO_L0 = 2 };
enum { EPI_FWD = 0, EPI_ADJ = 1, EPI_ADJ_L0 = 2 };

struct LayerArgs {
  const float* Bin;    // [tile][S][KT][256] B-operand source (pre-activations or adjoints)
  const float* Wp;     // [KT][MT][256] packed A operand
  const float* X;      // [tile][XT][256] augmented raw input
  const float* W0s;    // [XT][KT or MT][256] packed layer-0 weights (PRO_L0 / EPI_ADJ_L0)
  const float* tanc0;  // [3][KT or MT][256]   layer-0 tangent constants W0[:, d]
  const float* Wsp;    // [XT][MT][256] packed skip weights (EPI_FWD)
  const float* tanc;   // [3][MT][256]  skip tangent constants (EPI_FWD)
  float* Out;          // EPI_FWD: [tile][S][MT][256]; EPI_ADJ: in place over pre-activations; EPI_ADJ_L0: [tile][1+S1][MT][256]
  int KT, MT, ntiles;
  stpde_jet_cfg cfg;
};

template <int S1, int S2, int MC, int PRO, int EPI>
__global__ __launch_bounds__(256) void k_layer(LayerArgs a) {
  constexpr int S = 1 + S1 + S2;
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= a.ntiles) return;
  const int mt0 = blockIdx.y * MC;
  const int KT = a.KT, MT = a.MT;
  const int lo = lane * 4;

  f32x4 acc[MC][S];
#pragma unroll
  for (int mi = 0; mi < MC; ++mi)
#pragma unroll
    for (int st = 0; st < S; ++st) acc[mi][st] = f32x4{0.f, 0.f, 0.f, 0.f};

  f32x4 xb[XT];
  if (PRO == PRO_L0 || EPI == EPI_FWD || EPI == EPI_ADJ_L0) {
#pragma unroll
    for (int xt = 0; xt < XT; ++xt) xb[xt] = ld4(a.X + ((size_t)tile * XT + xt) * 256 + lo);
  }

  const float* bin = a.Bin + (size_t)tile * S * KT * 256 + lo;
  for (int kt = 0; kt < KT; ++kt) {
    f32x4 B[S];
    if (PRO == PRO_NONE) {
#pragma unroll
      for (int st = 0; st < S; ++st) B[st] = ld4(bin + ((size_t)st * KT + kt) * 256);
    } else {
      f32x4 pre[S];
      if (PRO == PRO_ACT) {
#pragma unroll
        for (int st = 0; st < S; ++st) pre[st] = ld4(bin + ((size_t)st * KT + kt) * 256);
      } else {  // PRO_L0: regenerate layer 0's pre-activation block kt from the raw input
        f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int xt = 0; xt < XT; ++xt) {
          f32x4 w = ld4(a.W0s + ((size_t)xt * KT + kt) * 256 + lo);
#pragma unroll
          for (int r = 0; r < 4; ++r) a0 = mfma4(w[r], xb[xt][r], a0);
        }
        pre[0] = a0;
        if (S1 == 3) {
#pragma unroll
          for (int d = 0; d < 3; ++d) pre[1 + d] = ld4(a.tanc0 + ((size_t)d * KT + kt) * 256 + lo);
#pragma unroll
          for (int p = 0; p < S2; ++p) pre[4 + p] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      act_jet_fwd<S1, S2>(a.cfg, pre, B);
    }
#pragma unroll
    for (int mi = 0; mi < MC; ++mi) {
      const int mt = mt0 + mi;
      if (mt < MT) {
        f32x4 w = ld4(a.Wp + ((size_t)kt * MT + mt) * 256 + lo);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int st = 0; st < S; ++st) acc[mi][st] = mfma4(w[r], B[st][r], acc[mi][st]);
      }
    }
  }

#pragma unroll
  for (int mi = 0; mi < MC; ++mi) {
    const int mt = mt0 + mi;
    if (mt >= MT) continue;
    if (EPI == EPI_FWD) {
#pragma unroll
      for (int xt = 0; xt < XT; ++xt) {
        f32x4 w = ld4(a.Wsp + ((size_t)xt * MT + mt) * 256 + lo);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[mi][0] = mfma4(w[r], xb[xt][r], acc[mi][0]);
      }
      if (S1 == 3) {
#pragma unroll
        for (int d = 0; d < 3; ++d) acc[mi][1 + d] += ld4(a.tanc + ((size_t)d * MT + mt) * 256 + lo);
      }
#pragma unroll
      for (int st = 0; st < S; ++st) st4(a.Out + (((size_t)tile * S + st) * MT + mt) * 256 + lo, acc[mi][st]);
    } else {
      f32x4 pre[S], ab[S];
      if (EPI == EPI_ADJ) {
#pragma unroll
        for (int st = 0; st < S; ++st) pre[st] = ld4(a.Out + (((size_t)tile * S + st) * MT + mt) * 256 + lo);
      } else {
        f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int xt = 0; xt < XT; ++xt) {
          f32x4 w = ld4(a.W0s + ((size_t)xt * MT + mt) * 256 + lo);
#pragma unroll
          for (int r = 0; r < 4; ++r) a0 = mfma4(w[r], xb[xt][r], a0);
        }
        pre[0] = a0;
        if (S1 == 3) {
#pragma unroll
          for (int d = 0; d < 3; ++d) pre[1 + d] = ld4(a.tanc0 + ((size_t)d * MT + mt) * 256 + lo);
#pragma unroll
          for (int p = 0; p < S2; ++p) pre[4 + p] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      act_jet_adj<S1, S2>(a.cfg, pre, acc[mi], ab);
      constexpr int SO = (EPI == EPI_ADJ) ? S : 1 + S1;
#pragma unroll
      for (int st = 0; st < SO; ++st) st4(a.Out + (((size_t)tile * SO + st) * MT + mt) * 256 + lo, ab[st]);
    }
  }
}

template <int S1, int S2, int PRO, int EPI>
static int launch_layer(const LayerArgs& a, hipStream_t stream) {
  constexpr int MC = 4;
  dim3 grid((a.ntiles + 3) / 4, (a.MT + MC - 1) / MC);
  hipLaunchKernelGGL((k_layer<S1, S2, MC, PRO, EPI>), grid, dim3(256), 0, stream, a);
  return stpde_check_launch("k_layer");
}

template <int PRO, int EPI>
static int dispatch_streams(const LayerArgs& a, hipStream_t stream) {
  const int S1 = a.cfg.S1, S2 = a.cfg.S2;
  if (S1 == 0 && S2 == 0) return launch_layer<0, 0, PRO, EPI>(a, stream);
  if (S1 == 3 && S2 == 0) return launch_layer<3, 0, PRO, EPI>(a, stream);
  if (S1 == 3 && S2 == 2) return launch_layer<3, 2, PRO, EPI>(a, stream);
  if (S1 == 3 && S2 == 6) return launch_layer<3, 6, PRO, EPI>(a, stream);
  stpde_set_error("stream configuration S1=%d S2=%d not compiled (supported: (0,0) (3,0) (3,2) (3,6))", S1, S2);
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
                                   const float* W0s_pack, const float* tanc0, float* out_pre, void* stream) {
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
  a.KT = d->KT;
  a.MT = d->MT;
  a.ntiles = d->ntiles;
  a.cfg = d->cfg;
  if (!X || !Ws_pack || !out_pre || (d->cfg.S1 && !tanc)) {
    stpde_set_error("jet_layer_fwd: null pointer");
    return STPDE_E_BADARG;
  }
  if (d->first_hidden) {
    if (!W0s_pack || (d->cfg.S1 && !tanc0)) {
      stpde_set_error("jet_layer_fwd: first_hidden needs W0s_pack/tanc0");
      return STPDE_E_BADARG;
    }
    return dispatch_streams<PRO_L0, EPI_FWD>(a, (hipStream_t)stream);
  }
  if (d->KT > 0 && (!in_pre || !Wh_pack)) {
    stpde_set_error("jet_layer_fwd: null hidden input");
    return STPDE_E_BADARG;
  }
  return dispatch_streams<PRO_ACT, EPI_FWD>(a, (hipStream_t)stream);
}

extern "C" int stpde_jet_layer_bwd(const stpde_layer_desc* d, const float* abar_out, const float* WhT_pack,
                                   float* in_pre, const float* X, const float* W0s_pack, const float* tanc0,
                                   float* abar0, void* stream) {
  int rc = check_cfg(d);
  if (rc) return rc;
  // GEMM roles swap: contraction over this layer's MT output tiles, result over its KT input tiles.
  LayerArgs a{};
  a.Bin = abar_out;
  a.Wp = WhT_pack;
  a.X = X;
  a.W0s = W0s_pack;
  a.tanc0 = tanc0;
  a.KT = d->MT;
  a.MT = d->KT;
  a.ntiles = d->ntiles;
  a.cfg = d->cfg;
  if (!abar_out || !WhT_pack || d->KT <= 0) {
    stpde_set_error("jet_layer_bwd: null pointer / no hidden input");
    return STPDE_E_BADARG;
  }
  if (d->first_hidden) {
    if (!X || !W0s_pack || !abar0 || (d->cfg.S1 && !tanc0)) {
      stpde_set_error("jet_layer_bwd: first_hidden needs X/W0s_pack/tanc0/abar0");
      return STPDE_E_BADARG;
    }
    a.Out = abar0;
    return dispatch_streams<PRO_NONE, EPI_ADJ_L0>(a, (hipStream_t)stream);
  }
  if (!in_pre) {
    stpde_set_error("jet_layer_bwd: null in_pre");
    return STPDE_E_BADARG;
  }
  a.Out = in_pre;
  return dispatch_streams<PRO_NONE, EPI_ADJ>(a, (hipStream_t)stream);
}
