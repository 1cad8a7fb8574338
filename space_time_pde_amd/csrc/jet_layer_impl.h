// One IM-NET layer on all derivative streams, forward and input-gradient (dgrad), as MFMA GEMMs
//   out^T[16*MT x 16 rows] (per stream) = W[16*MT x K] * in^T[K x 16 rows]
// over row tiles of 16 corner rows (2 query points x 8 corners).  A operand = packed weights (one float4 per lane per
// (k-tile, m-tile) block = the 4 k-steps of the block, streamed from L2), B operand = the fragment block of the
// previous layer (C/D image == B image, so nothing is re-laid out), accumulators = float4 per (output tile, stream).
// Kernel variants (dispatch at the bottom of this file):
//   k_layer        per wave, no LDS / barriers: narrow layers (fewer than 8 output tiles) -- HBM-bound on the stash
//   k_layer_coop   4 / 8 waves share ONE row tile and split its output tiles; the B operand of every k-tile (stash load
//                  + activation jet, or layer-0 regeneration) is produced once per workgroup into a double-buffered
//                  LDS ring; bf16-operand variant for BASELINE configs[3]
// Replaces (reference): src/implicit_net.py:48-54 on the rows of src/local_implicit_grid.py:53, and the reverse
// sweeps of src/pde.py:8-9 (streams carry d/dr and d2/dr2 forward instead).
#pragma once
#include <cstdlib>

#include "common.h"

enum { PRO_NONE = 0, PRO_ACT = 1, PRO_L0 = 2 };
enum { EPI_FWD = 0, EPI_ADJ = 1, EPI_ADJ_L0 = 2 };
// Timing-only ablations of the cooperative kernel (tools/micro/ablate_layer.py builds a private library with
// -DSTPDE_ABLATE=n; results are WRONG by construction): 1 = no activation jet in the produce stage, 2 = no barrier in the
// main loop, 3 = weight fragments always from k-tile 0 (L1-resident), 4 = no epilogue, 5 = no produce stage in the loop,
// 6 = (three-term split mode) one of the six partial products only.
#ifndef STPDE_ABLATE
#define STPDE_ABLATE 0
#endif

// Phase timing (tools/micro/ablate_layer.py stamp, private builds with -DSTPDE_STAMP=1): s_memtime stamps of every wave of
// 256 mid-launch workgroups at the phase boundaries of k_layer_coop, read back with stpde_stamp_read (jet_layer_s31.hip).
#ifndef STPDE_STAMP
#define STPDE_STAMP 0
#endif
#ifndef STPDE_FWD2_OCC3
#define STPDE_FWD2_OCC3 1    // fp32 forward of the 2-output-tiles-per-wave shapes (second hidden layer), S <= 5: 168-register budget
#endif
#ifndef STPDE_FWD2_OCC_N
#define STPDE_FWD2_OCC_N 3    // (4: 128 registers -- see profiles/r5_ablate_first_barrier.txt)
#endif
#ifndef STPDE_EARLY_L0
#define STPDE_EARLY_L0 1
#endif
#ifndef STPDE_EPI_PF
#define STPDE_EPI_PF 1
#endif
#ifndef STPDE_EPI_PF_BF
#define STPDE_EPI_PF_BF 1
#endif
#ifndef STPDE_EPI_PF_DEPTH
#define STPDE_EPI_PF_DEPTH 1     // output tiles whose stashed pre-activation blocks are in flight ahead of the adjoint being computed
                                 // (2: +-0, 3 / 4: fc2 dgrad +7 % -- the epilogue is not waiting for these loads; same file)
#endif
#ifndef STPDE_EPI_OPF
#define STPDE_EPI_OPF 3          // cooperative kernel, tile-independent-of-the-rows operands of the epilogues requested ahead: bit 0 = first
                                 // hidden layer's input gradient (z0 blocks where they are not fetched at the start of the pass,
                                 // layer-0 tangent constants), bit 1 = forward (skip weights / tangent constants one output tile ahead),
                                 // bit 2 = also in the three-term split mode (measured slower there: profiles/r5_ablate_fc1_fp32x3.txt).
                                 // (value-tile mode of the lattice inference: not offered -- its kernels run at 3-4 waves per SIMD and the
                                 // extra fragments would cost one of them.)
                                 // Stream sets with S <= 5 only: at S = 8 the extra fragments spill (12 -> 68 B in the fc2 forward)
#endif
#ifndef STPDE_X3_STREAM_OUTER
#define STPDE_X3_STREAM_OUTER 1  // three-term split mode: one stream's B fragments live at a time in the MFMA loop (see there)
#endif
#ifndef STPDE_X3_EARLY
#define STPDE_X3_EARLY 0         // three-term split mode, input-gradient kernels: stash loads of the next group issued before the MFMAs
#endif
#ifndef STPDE_X3_Z0P
#define STPDE_X3_Z0P 0           // three-term split mode, first hidden layer's input gradient: z0 blocks requested at the start of the pass
#endif
#ifndef STPDE_SKIP_FIRST_BAR
#define STPDE_SKIP_FIRST_BAR 1   // no "ring free" barrier in front of a workgroup's FIRST pass (nobody has read the ring yet)
#endif
#if STPDE_STAMP
#define STPDE_STAMP_B0 8192
static __device__ unsigned long long g_stamp[256 * 8 * 16];
#define STAMP(i)                                                                                          \
  do {                                                                                                    \
    if (blockIdx.x >= STPDE_STAMP_B0 && blockIdx.x < STPDE_STAMP_B0 + 256 && (threadIdx.x & 63) == 0)     \
      g_stamp[((blockIdx.x - STPDE_STAMP_B0) * 8 + (threadIdx.x >> 6)) * 16 + (i)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define STAMP(i)
#endif

struct LayerArgs {
  const float* Bin;    // [tile][S][KT][256] B-operand source (pre-activations or adjoints)
  const float* Wp;     // [KT][MT][256] packed A operand
  const float* X;      // [tile][XT][256] augmented raw input
  const float* W0s;    // [XT][KT or MT][256] packed layer-0 weights (PRO_L0 / EPI_ADJ_L0)
  const float* tanc0;  // [3][KT or MT][256]   layer-0 tangent constants W0[:, d]
  const float* Wsp;    // [XT][MT][256] packed skip weights (EPI_FWD)
  const float* tanc;   // [3][MT][256]  skip tangent constants (EPI_FWD)
  float* Out;          // EPI_FWD: [tile][S][MT][256]; EPI_ADJ: the adjoints (== Pre: in place); EPI_ADJ_L0: [tile][1+S1][MT][256]
  const float* Pre;    // EPI_ADJ: stashed pre-activations [tile][S][MT][256] the adjoint is taken against (may be Out)
  const float* cw;     // [P][8] per-point weights of the combined second-order stream (S2 == 1), else unused
  const void* Wp16;    // [nsplit][KT/2][MT][64] x 8 bf16: A operand of the bf16-MFMA variants (two k-tiles per block), or null
  int nsplit;          // 1: operands rounded to bf16 (configs[3]);  3: fp32 operands split into three bf16 terms each
  float* pbar;         // dgrad, swish only: [STPDE_PBAR_SLOTS] accumulators of the adjoint of beta (nullable)
  float* Tan0;         // EPI_ADJ_L0, nullable: [tile][MT][3][16] row sums of the tangent-stream adjoints of layer 0; when
                       // given, Out holds the VALUE stream only: [tile][MT][256]
  float* Z0;           // [tile][KT or MT][256] value stream of the layer-0 pre-activations: written by the PRO_L0 forward
                       // (nullable there: the stores are dropped), read by EPI_ADJ_L0 instead of regenerating it from X
                       // (may alias Out when Out holds the value stream only: each lane reads its element before it writes it)
  int KT, MT, ntiles;
  int split;           // cooperative kernel: > 0 = number of output passes, each run by its own workgroup
  int pk;              // packed-buffer flags (common.h: ld_blk / st_blk): 1 = Bin, 2 = Out, 4 = Pre -- a forward kernel reads /
                       // writes packed STASH buffers (mode 1); an input-gradient kernel reads its Pre as a stash, its Bin and
                       // Out are packed ADJOINT buffers (mode 2)
  stpde_jet_cfg cfg;
};

// layer-0 pre-activation block `blk` (of nblk) regenerated from the raw input: a0 = W0 x (+ bias via the ones column)
__device__ __forceinline__ f32x4 layer0_block(const float* W0s, int nblk, int blk, int lo, const f32x4* xb) {
  f32x4 part[XT];
#pragma unroll
  for (int xt = 0; xt < XT; ++xt) {
    f32x4 w = ld4x(W0s + ((size_t)xt * nblk + blk) * 256 + lo, xt);
    f32x4 c = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < x_live(xt); ++r) c = mfma4(w[r], xb[xt][r], c);
    part[xt] = c;
  }
  return (part[0] + part[1]) + part[2];
}

// Epilogue of one output tile `mt` held in acc[S] (D image): forward = skip GEMM + tangent constants + store of the
// pre-activations; dgrad = activation-jet adjoint against the stored pre-activations (first hidden layer: the z0 stash of
// the forward kernel + the constant tangent columns; its tangent-stream adjoints leave as per-tile row sums).
// PKM: compile-time packed-buffer mask (LayerArgs.pk: 2 = Out, 4 = Pre); a run-time flag would put a branch in front of every
// block load / store and split the MFMA basic blocks (measured: +40 % on the kernels that took it)
template <int S1, int S2, int EPI, int ACT, int PKM = 0>
__device__ __forceinline__ void layer_epilogue(const LayerArgs& a, int tile, int mt, int MT, int lane, f32x4* accm,
                                               const f32x4 (*xbv)[XT], const float* cq, float& pacc,
                                               const f32x4* z0pre = nullptr, const f32x4* prepf = nullptr,
                                               const f32x4* wspf = nullptr, const f32x4* tcpf = nullptr) {
  constexpr int S = 1 + S1 + S2;
  constexpr bool VT = S1 == 0 && S2 > 0;     // value-tile mode: every stream is the value stream of its own row tile
  const int lo = lane * 4;
  f32x4 (&acc)[1][S] = *reinterpret_cast<f32x4 (*)[1][S]>(accm);
  constexpr int mi = 0;
    if (EPI == EPI_FWD) {
#pragma unroll
      for (int xt = 0; xt < XT; ++xt) {
        f32x4 w = wspf ? wspf[xt] : ld4x(a.Wsp + ((size_t)xt * MT + mt) * 256 + lo, xt);
#pragma unroll
        for (int sv = 0; sv < (VT ? S : 1); ++sv)
#pragma unroll
          for (int r = 0; r < x_live(xt); ++r) acc[mi][sv] = mfma4(w[r], xbv[sv][xt][r], acc[mi][sv]);
      }
      if (S1 == 3) {
#pragma unroll
        for (int d = 0; d < 3; ++d) acc[mi][1 + d] += tcpf ? tcpf[d] : ld4(a.tanc + ((size_t)d * MT + mt) * 256 + lo);
      }
#pragma unroll
      for (int st = 0; st < S; ++st) st_blk(a.Out, (PKM & 2) ? 1 : 0, tile, S, MT, st, mt, lane, acc[mi][st]);
    } else {
      f32x4 pre[S], ab[S];
      if (EPI == EPI_ADJ) {
#pragma unroll
        for (int st = 0; st < S; ++st)
          pre[st] = prepf ? prepf[st] : ld_blk(a.Pre, (PKM & 4) ? 1 : 0, tile, S, MT, st, mt, lane);
      } else {
        pre[0] = z0pre ? *z0pre : ld4(a.Z0 + ((size_t)tile * MT + mt) * 256 + lo);
        if (S1 == 3) {
#pragma unroll
          for (int d = 0; d < 3; ++d) pre[1 + d] = tcpf ? tcpf[d] : ld4(a.tanc0 + ((size_t)d * MT + mt) * 256 + lo);
#pragma unroll
          for (int p = 0; p < S2; ++p) pre[4 + p] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      act_jet_adj<S1, S2, ACT>(a.cfg, pre, acc[mi], ab, cq);
      if ((ACT == STPDE_ACT_SWISH || (ACT < 0 && a.cfg.act == STPDE_ACT_SWISH)) && a.pbar)
        pacc += swish_beta_adj<S1, S2>(a.cfg, pre, acc[mi], cq);
      constexpr int SO = (EPI == EPI_ADJ) ? S : 1 + S1;
      if (EPI == EPI_ADJ_L0 && S1 == 3 && a.Tan0) {
        // Layer 0's tangent streams are the constant columns W0[:, d], so their adjoints enter d W0[:, d] only through
        // their sum over rows: reduce over the 16 rows of the tile here (DPP row reduction) and write 48 floats per
        // (tile, output tile) instead of three 1 KiB blocks -- the layer-0 adjoint shrinks to its value stream.
        st_blk(a.Out, (PKM & 2) ? 2 : 0, tile, 1, MT, 0, mt, lane, ab[0]);      // (bf16 mode: the layer-0 adjoint as bf16 blocks)
        f32x4 ts[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) ts[d] = row_sum16x4(ab[1 + d]);
        if ((lane & 15) == 15) {
          float* tp = a.Tan0 + ((size_t)tile * MT + mt) * 48 + 4 * (lane >> 4);
#pragma unroll
          for (int d = 0; d < 3; ++d) st4(tp + 16 * d, ts[d]);
        }
      } else {
#pragma unroll
        for (int st = 0; st < SO; ++st) st_blk(a.Out, (PKM & 2) ? 2 : 0, tile, SO, MT, st, mt, lane, ab[st]);
      }
    }
}

// one atomic per wave: the per-lane partial sums of the swish-beta adjoint go to slot (block % STPDE_PBAR_SLOTS)
template <int EPI, int ACT>
__device__ __forceinline__ void flush_pbar(const LayerArgs& a, float pacc, int lane) {
  if (EPI != EPI_FWD && (ACT == STPDE_ACT_SWISH || (ACT < 0 && a.cfg.act == STPDE_ACT_SWISH)) && a.pbar) {
    const float v = wave_sum(pacc);
    if (lane == 0) atomicAdd(a.pbar + (blockIdx.x % STPDE_PBAR_SLOTS), v);
  }
}

template <int S1, int S2, int MC, int PRO, int EPI, int ACT, bool GUARD>
__global__ __launch_bounds__(256) void k_layer(LayerArgs a) {
  constexpr int S = 1 + S1 + S2;
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= a.ntiles) return;
  const int KT = a.KT, MT = a.MT;
  const int lo = lane * 4;
  float cq[6];
  load_cq<S2>(a.cw, tile * 2 + ((lane & 15) >> 3), cq);
  float pacc = 0.f;

  constexpr bool VT = S1 == 0 && S2 > 0;
  constexpr int NX = VT ? S : 1;             // value-tile mode: one raw-input tile per stream
  f32x4 xb[NX][XT];
  if (PRO == PRO_L0 || EPI == EPI_FWD) {
#pragma unroll
    for (int sv = 0; sv < NX; ++sv)
#pragma unroll
      for (int xt = 0; xt < XT; ++xt) xb[sv][xt] = ld4x(a.X + (((size_t)tile * NX + sv) * XT + xt) * 256 + lo, xt);
  }
  const auto z0r = opt_store_rsrc(a.Z0 ? a.Z0 + (size_t)tile * KT * 256 : nullptr, (unsigned)KT * 1024u);

  // the wave walks all output chunks of its tile: the B blocks it re-reads stay hot in this CU's L1/L2
  for (int mt0 = 0; mt0 < MT; mt0 += MC) {
  f32x4 acc[MC][S];
#pragma unroll
  for (int mi = 0; mi < MC; ++mi)
#pragma unroll
    for (int st = 0; st < S; ++st) acc[mi][st] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* wp = a.Wp + (size_t)mt0 * 256 + lo;

  // raw (un-activated) B block and weight blocks of k-tile kt
  auto load_raw = [&](int kt, f32x4* raw) {
    if (PRO == PRO_L0) {
#pragma unroll
      for (int sv = 0; sv < NX; ++sv) raw[sv] = layer0_block(a.W0s, KT, kt, lo, xb[sv]);
      if (!VT) opt_st4(z0r, kt * 1024 + lane * 16, raw[0]);      // stash for the dgrad epilogue (dropped when Z0 is null)
      if (S1 == 3) {
#pragma unroll
        for (int d = 0; d < 3; ++d) raw[1 + d] = ld4(a.tanc0 + ((size_t)d * KT + kt) * 256 + lo);
#pragma unroll
        for (int p = 0; p < S2; ++p) raw[4 + p] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    } else {
#pragma unroll
      for (int st = 0; st < S; ++st) raw[st] = ld_blk(a.Bin, 0, tile, S, KT, st, kt, lane);
    }
  };
  auto load_w = [&](int kt, f32x4* w) {
#pragma unroll
    for (int mi = 0; mi < MC; ++mi) {
      const int mi_c = GUARD ? (mt0 + mi < MT ? mi : 0) : mi;   // clamp instead of branching
      w[mi] = ld4(wp + ((size_t)kt * MT + mi_c) * 256);
    }
  };

  if (KT > 0) {
    f32x4 Bc[S], wc[MC];
    {
      f32x4 raw[S];
      load_raw(0, raw);
      if (PRO == PRO_NONE) {
#pragma unroll
        for (int st = 0; st < S; ++st) Bc[st] = raw[st];
      } else {
        act_jet_fwd<S1, S2, ACT>(a.cfg, raw, Bc, cq);
      }
      load_w(0, wc);
    }
    for (int kt = 0; kt < KT; ++kt) {
      const int kn = kt + 1 < KT ? kt + 1 : kt;   // last iteration re-fetches its own block (harmless)
      f32x4 rawn[S], wn[MC];
      load_raw(kn, rawn);
      load_w(kn, wn);
#pragma unroll
      for (int mi = 0; mi < MC; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int st = 0; st < S; ++st) acc[mi][st] = mfma4(wc[mi][r], Bc[st][r], acc[mi][st]);
      if (PRO == PRO_NONE) {
#pragma unroll
        for (int st = 0; st < S; ++st) Bc[st] = rawn[st];
      } else {
        act_jet_fwd<S1, S2, ACT>(a.cfg, rawn, Bc, cq);
      }
#pragma unroll
      for (int mi = 0; mi < MC; ++mi) wc[mi] = wn[mi];
    }
  }

#pragma unroll
  for (int mi = 0; mi < MC; ++mi) {
    const int mt = mt0 + mi;
    if (GUARD && mt >= MT) continue;
    layer_epilogue<S1, S2, EPI, ACT>(a, tile, mt, MT, lane, acc[mi], xb, cq, pacc);
  }
  }  // chunk loop
  flush_pbar<EPI, ACT>(a, pacc, lane);
}


// ------------------------------------------------------------------------------------------------------------
// Workgroup-cooperative variant: the 4 waves of a workgroup share ONE row tile and split its output tiles
// (MCg per wave per pass).  The B operand of every k-tile (S blocks) is produced ONCE per workgroup -- wave kt%4
// loads / regenerates / activates block kt -- into a double-buffered LDS ring of 2 x 4 k-tiles and is consumed by
// all four waves, so the stash reads, the layer-0 regeneration MFMAs and the activation-jet VALU work are shared
// 4 ways.  One barrier per 4 k-tiles; production of group g+1 sits in the same basic block as the MFMAs of group g.
// Requires KT % 4 == 0 and MT % (4*MCg) == 0 (otherwise the per-wave kernel above is used).
// ------------------------------------------------------------------------------------------------------------
// BF: the hidden-to-hidden GEMM runs on v_mfma_f32_16x16x32_bf16 (operands rounded to bf16 in the produce stage /
// the bf16 weight pack, fp32 accumulation); the ring then holds 8-byte bf16 fragments and one MFMA contracts over TWO
// k-tiles.  Layer-0 regeneration, skip GEMM, activation jets and epilogues stay fp32.
// PK (bf16 variant only): k-tiles produced per wave and group -- the bf16 MFMAs of a k-tile take 1/16 of the fp32
// time, so twice the k-tiles per barrier halve the number of exposed load -> activation -> LDS -> barrier chains.
// SPL = 3 (with BF): fp32-ACCURATE product on the bf16 pipe.  Every fp32 operand is split exactly into three bf16 terms
// (x = hi + mid + lo, 8 + 8 + 8 mantissa bits: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid), the residuals
// are exact in fp32) and the six partial products of weight 2^0, 2^-8, 2^-16 are accumulated in fp32
// (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid; the dropped ones are below 2^-24 of the product = below fp32 rounding).
// bf16 x bf16 products are exact in fp32, so the result carries fp32 accuracy, at 6 bf16 MFMAs (6 x 16 cycles for K = 32)
// instead of 8 fp32 MFMAs (8 x 33 cycles): the fp32 matrix pipe of gfx950 runs at the VECTOR fp32 rate, the bf16 pipe 16 x
// faster.  The split of the activations is done once per workgroup in the produce stage, that of the weights on the host.
template <int S1, int S2, int MCg, int PRO, int EPI, int ACT, int NW, bool BF = false, int PK = 1, bool WRING = false,
          int SPL = 1, int PKM = 0>
__global__ __launch_bounds__(64 * NW, ((!BF || SPL == 1) && PRO == PRO_ACT && EPI == EPI_FWD && MCg == 2 && NW == 4 && !WRING && S1 + S2 <= 4 && STPDE_FWD2_OCC3) ? STPDE_FWD2_OCC_N : 2) void k_layer_coop(LayerArgs a) {
  constexpr int S = 1 + S1 + S2, GK = NW * PK;
  static_assert(!WRING || (!BF && GK == 4), "the weight ring is written for 4 k-tiles per group, fp32");
  static_assert(SPL == 1 || (BF && SPL == 3), "operand splitting is a bf16-pipe mode");
  // format of the B operand's buffer: a forward kernel reads a packed STASH, an input-gradient kernel a packed ADJOINT
  constexpr int BMODE = (PKM & 1) ? (EPI == EPI_FWD ? 1 : 2) : 0;
  __shared__ __attribute__((aligned(16))) float hb[2][GK][S][BF ? 128 * SPL : 256];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int KT = a.KT, MT = a.MT;
  const int lo = lane * 4;
  // Pass splitting (dgrad of wide layers): the output passes of one row tile are run by different workgroups that
  // sit in consecutive slots of the SAME XCD (block b runs on XCD b % 8), i.e. at the same time on the same L2, so
  // the tile's B operand is fetched from HBM once instead of once per pass.
  STAMP(0);
  int tile = blockIdx.x, pass0 = 0, pstep = 1;
  if (a.split > 0) {
    const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
    pass0 = slot % a.split;
    pstep = a.split;
    tile = (slot / a.split) * 8 + xcd;
    if (tile >= a.ntiles) return;
  }
  float cq[6];
  load_cq<S2>(a.cw, tile * 2 + ((lane & 15) >> 3), cq);
  float pacc = 0.f;

  constexpr bool VT = S1 == 0 && S2 > 0;
  constexpr int NX = VT ? S : 1;             // value-tile mode: one raw-input tile per stream
  f32x4 xb[NX][XT];
  if (PRO == PRO_L0 || EPI == EPI_FWD) {
#pragma unroll
    for (int sv = 0; sv < NX; ++sv)
#pragma unroll
      for (int xt = 0; xt < XT; ++xt) xb[sv][xt] = ld4x(a.X + (((size_t)tile * NX + sv) * XT + xt) * 256 + lo, xt);
  }
  const auto z0r = opt_store_rsrc(a.Z0 ? a.Z0 + (size_t)tile * KT * 256 : nullptr, (unsigned)KT * 1024u);

  // produce the B block of k-tile kt (this wave's turn) into ring slot (buf, slot)
  auto produce = [&](int kt, int buf, int slot) {
    f32x4 raw[S], B[S];
    if (PRO == PRO_L0) {
#pragma unroll
      for (int sv = 0; sv < NX; ++sv) raw[sv] = layer0_block(a.W0s, KT, kt, lo, xb[sv]);
      if (!VT) opt_st4(z0r, kt * 1024 + lane * 16, raw[0]);      // stash for the dgrad epilogue (dropped when Z0 is null)
      if (S1 == 3) {
#pragma unroll
        for (int d = 0; d < 3; ++d) raw[1 + d] = ld4(a.tanc0 + ((size_t)d * KT + kt) * 256 + lo);
#pragma unroll
        for (int p = 0; p < S2; ++p) raw[4 + p] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    } else {
#pragma unroll
      for (int st = 0; st < S; ++st) raw[st] = ld_blk(a.Bin, BMODE, tile, S, KT, st, kt, lane);
    }
    if (PRO == PRO_NONE || STPDE_ABLATE == 1) {
#pragma unroll
      for (int st = 0; st < S; ++st) B[st] = raw[st];
    } else {
      act_jet_fwd<S1, S2, ACT>(a.cfg, raw, B, cq);
    }
#pragma unroll
    for (int st = 0; st < S; ++st) {
      if constexpr (BF) {
        f32x4 v = B[st];
#pragma unroll
        for (int t = 0; t < SPL; ++t) {
          const bf16x4 h = to_bf4(v);
          *reinterpret_cast<bf16x4*>(&hb[buf][slot][st][128 * t + lane * 2]) = h;
          if (t + 1 < SPL) v -= bf4_to_f32(h);      // exact: the residual of a round-to-nearest bf16 fits in fp32
        }
      } else {
        st4(&hb[buf][slot][st][lo], B[st]);
      }
    }
  };
  // PRO_NONE (input-gradient kernels: the B blocks are plain loads from the stash): the loads are ISSUED before the MFMAs of
  // the current group and converted / written to the ring after them, so their HBM latency hides behind the MFMAs instead of
  // sitting in front of the barrier (round 3; first-layer dgrad 92.1 -> 89.5 ms per step in fp32, 26.9 -> 23.6 ms with bf16
  // operands).  The same split for PRO_ACT (loads early, activation late) was measured and dropped: fp32 +-0, bf16 layer-2
  // forward 12.0 -> 16.6 ms.
  auto store_block = [&](const f32x4* B, int buf, int slot) {
#pragma unroll
    for (int st = 0; st < S; ++st) {
      if constexpr (BF) {
        f32x4 v = B[st];
#pragma unroll
        for (int t = 0; t < SPL; ++t) {
          const bf16x4 h = to_bf4(v);
          *reinterpret_cast<bf16x4*>(&hb[buf][slot][st][128 * t + lane * 2]) = h;
          if (t + 1 < SPL) v -= bf4_to_f32(h);
        }
      } else {
        st4(&hb[buf][slot][st][lo], B[st]);
      }
    }
  };
  auto produce_group = [&](int g, int buf) {      // this wave's PK k-tiles of group g
#pragma unroll
    for (int k = 0; k < PK; ++k) produce(GK * g + NW * k + wv, buf, NW * k + wv);
  };

  const int ngroups = KT / GK;
  bool first_pass = true;
  for (int mt0 = (pass0 * NW + wv) * MCg; mt0 < MT; mt0 += pstep * NW * MCg) {
    f32x4 acc[MCg][S];
#pragma unroll
    for (int mi = 0; mi < MCg; ++mi)
#pragma unroll
      for (int st = 0; st < S; ++st) acc[mi][st] = f32x4{0.f, 0.f, 0.f, 0.f};
    // first hidden layer's input gradient: the z0 blocks its epilogue takes the adjoint against, fetched now
    constexpr bool Z0P = EPI == EPI_ADJ_L0 && (SPL == 1 || STPDE_X3_Z0P);      // (three-term split mode: behind a switch)
    f32x4 z0p[Z0P ? MCg : 1];
    if constexpr (Z0P) {
#pragma unroll
      for (int mi = 0; mi < MCg; ++mi) z0p[mi] = ld4(a.Z0 + ((size_t)tile * MT + mt0 + mi) * 256 + lo);
    }
    const float* wp = a.Wp + (size_t)mt0 * 256 + lo;
    f32x4 wr[WRING ? 4 : 1][MCg];
    if constexpr (WRING) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int mi = 0; mi < MCg; ++mi) wr[q][mi] = ld4(wp + ((size_t)q * MT + mi) * 256);
    }
    // ring free (previous pass fully consumed).  Not in front of the FIRST pass (round 5): __syncthreads() carries a
    // vmcnt(0), so the barrier made the first group's operand loads wait for the round trip of the loads above (raw input,
    // combination weights, weight ring, z0 blocks) instead of travelling with them.  Measured (profiles/r5_ablate_first_barrier.txt,
    // 2^18 points): exact-fp32 fc1 forward 22.19 -> 22.01 ms, every other kernel of the template within +-0.3 %.
    if (!STPDE_SKIP_FIRST_BAR || !first_pass) __syncthreads();
    first_pass = false;
    STAMP(1);
    produce_group(0, 0);
    STAMP(2);
    __syncthreads();
    STAMP(3);
    for (int gi = 0; gi < ngroups; ++gi) {
      const int buf = gi & 1;
      const int gnext = gi + 1 < ngroups ? gi + 1 : gi;
      // B blocks are plain loads from the stash: issue them now (not in the three-term split mode: the registers the
      // early blocks occupy cost that mode more than the hidden latency gives, fp32x3 dgrad 64.5 -> 71.3 ms)
      constexpr bool EARLY = PRO == PRO_NONE && (SPL == 1 || STPDE_X3_EARLY);
      constexpr bool EARLY0 = PRO == PRO_L0 && !VT && STPDE_EARLY_L0;   // layer 0 on the fly: its weight / tangent fragments
      f32x4 rawn[EARLY ? PK : 1][S];
      if constexpr (EARLY) {
#pragma unroll
        for (int k = 0; k < PK; ++k)
#pragma unroll
          for (int st = 0; st < S; ++st)
            rawn[k][st] = ld_blk(a.Bin, BMODE, tile, S, KT, st, GK * gnext + NW * k + wv, lane);
      }
      f32x4 w0n[EARLY0 ? PK : 1][XT], tcn[EARLY0 ? PK : 1][3];
      if constexpr (EARLY0) {
#pragma unroll
        for (int k = 0; k < PK; ++k) {
          const int kt = GK * gnext + NW * k + wv;
#pragma unroll
          for (int xt = 0; xt < XT; ++xt) w0n[k][xt] = ld4x(a.W0s + ((size_t)xt * KT + kt) * 256 + lo, xt);
          if (S1 == 3) {
#pragma unroll
            for (int d = 0; d < 3; ++d) tcn[k][d] = ld4(a.tanc0 + ((size_t)d * KT + kt) * 256 + lo);
          }
        }
      }
      if constexpr (BF && SPL == 3 && STPDE_X3_STREAM_OUTER) {
        // Three-term split mode, stream-outer (round 5, last session).  The loop below holds the B fragments of ALL S streams
        // (S x 3 terms x 4 registers = 60) next to the 48 weight-fragment registers and the 80 accumulators: 256 registers
        // and 76-108 B of scratch inside the loop.  Here one stream's three terms are live at a time (the next stream's
        // requested before the 6 x MCg MFMAs of the current one, order pinned); every accumulator still receives its six
        // partial products smallest first, k-pair by k-pair: bit-identical sums.
        const bf16x8* wp16 = reinterpret_cast<const bf16x8*>(a.Wp16) + (size_t)mt0 * 64 + lane;
        constexpr int PIN = 0x2 | 0x4 | 0x10 | 0x20 | 0x40 | 0x200;    // VALU, SALU, VMEM, DS writes may cross; MFMA / DS reads not
        auto ldB = [&](int q, int st, bf16x8* B) {
#pragma unroll
          for (int t = 0; t < SPL; ++t)
            B[t] = cat8(*reinterpret_cast<const bf16x4*>(&hb[buf][2 * q][st][128 * t + lane * 2]),
                        *reinterpret_cast<const bf16x4*>(&hb[buf][2 * q + 1][st][128 * t + lane * 2]));
        };
        constexpr int TW[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};
        bf16x8 Bc[SPL], Bn[SPL];
        ldB(0, 0, Bc);
#pragma unroll
        for (int q = 0; q < GK / 2; ++q) {
          const int kp = GK / 2 * gi + q;
          bf16x8 w8[MCg][SPL];
#pragma unroll
          for (int mi = 0; mi < MCg; ++mi)
#pragma unroll
            for (int t = 0; t < SPL; ++t)
              w8[mi][t] = wp16[(((size_t)t * (KT / 2) + (STPDE_ABLATE == 3 ? q : kp)) * MT + mi) * 64];
#pragma unroll
          for (int st = 0; st < S; ++st) {
            if (st + 1 < S) ldB(q, st + 1, Bn);
            else if (q + 1 < GK / 2) ldB(q + 1, 0, Bn);
            __builtin_amdgcn_sched_barrier(PIN);
#pragma unroll
            for (int c = 0; c < (STPDE_ABLATE == 6 ? 1 : 6); ++c)
#pragma unroll
              for (int mi = 0; mi < MCg; ++mi) acc[mi][st] = mfma_bf(w8[mi][TW[c]], Bc[TB[c]], acc[mi][st]);
            __builtin_amdgcn_sched_barrier(PIN);
#pragma unroll
            for (int t = 0; t < SPL; ++t) Bc[t] = Bn[t];
          }
        }
      } else if constexpr (BF) {
        const bf16x8* wp16 = reinterpret_cast<const bf16x8*>(a.Wp16) + (size_t)mt0 * 64 + lane;
#pragma unroll
        for (int q = 0; q < GK / 2; ++q) {
          const int kp = GK / 2 * gi + q;       // pair of k-tiles (2 kp, 2 kp + 1)
          bf16x8 B8[S][SPL], w8[MCg][SPL];
#pragma unroll
          for (int st = 0; st < S; ++st)
#pragma unroll
            for (int t = 0; t < SPL; ++t)
              B8[st][t] = cat8(*reinterpret_cast<const bf16x4*>(&hb[buf][2 * q][st][128 * t + lane * 2]),
                               *reinterpret_cast<const bf16x4*>(&hb[buf][2 * q + 1][st][128 * t + lane * 2]));
#pragma unroll
          for (int mi = 0; mi < MCg; ++mi)
#pragma unroll
            for (int t = 0; t < SPL; ++t)
              w8[mi][t] = wp16[(((size_t)t * (KT / 2) + (STPDE_ABLATE == 3 ? q : kp)) * MT + mi) * 64];
          if constexpr (SPL == 3) {
            // six partial products, smallest first: (weight term, activation term)
            constexpr int TW[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
            for (int c = 0; c < (STPDE_ABLATE == 6 ? 1 : 6); ++c)
#pragma unroll
              for (int mi = 0; mi < MCg; ++mi)
#pragma unroll
                for (int st = 0; st < S; ++st) acc[mi][st] = mfma_bf(w8[mi][TW[c]], B8[st][TB[c]], acc[mi][st]);
          } else {
#pragma unroll
            for (int mi = 0; mi < MCg; ++mi)
#pragma unroll
              for (int st = 0; st < S; ++st) acc[mi][st] = mfma_bf(w8[mi][0], B8[st][0], acc[mi][st]);
          }
        }
      } else if constexpr (WRING) {
        // weight fragments through a register ring three k-tiles deep that runs straight across the group barriers
        // (GK == 4 == ring length, so the slot of k-tile kt is the compile-time q): an L2 round trip then has
        // ~3 x (MCg * S * 4) MFMAs of cover instead of the one k-tile the scheduler arranges on its own
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int kt = 4 * gi + q;
          const int ktn = STPDE_ABLATE == 3 ? (q & 1) : (kt + 3 < KT ? kt + 3 : KT - 1);
          f32x4 B[S];
#pragma unroll
          for (int st = 0; st < S; ++st) B[st] = ld4(&hb[buf][q][st][lo]);
#pragma unroll
          for (int mi = 0; mi < MCg; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
              for (int st = 0; st < S; ++st) acc[mi][st] = mfma4(wr[q][mi][r], B[st][r], acc[mi][st]);
#pragma unroll
          for (int mi = 0; mi < MCg; ++mi) wr[(q + 3) & 3][mi] = ld4(wp + ((size_t)ktn * MT + mi) * 256);
        }
      } else {
#pragma unroll
      for (int q = 0; q < GK; ++q) {
        const int kt = GK * gi + q;
        f32x4 B[S], w[MCg];
#pragma unroll
        for (int st = 0; st < S; ++st) B[st] = ld4(&hb[buf][q][st][lo]);
#pragma unroll
        for (int mi = 0; mi < MCg; ++mi) w[mi] = ld4(wp + ((size_t)kt * MT + mi) * 256);
#pragma unroll
        for (int mi = 0; mi < MCg; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int st = 0; st < S; ++st) acc[mi][st] = mfma4(w[mi][r], B[st][r], acc[mi][st]);
      }
      }
      // branch-free (the last group re-produces one of its own blocks): one basic block per group, so the
      // produce stage's loads / regeneration / activation VALU interleave with the MFMAs above
      if constexpr (EARLY) {
#pragma unroll
        for (int k = 0; k < PK; ++k) store_block(rawn[k], buf ^ 1, NW * k + wv);
      } else if constexpr (EARLY0) {
#pragma unroll
        for (int k = 0; k < PK; ++k) {
          const int kt = GK * gnext + NW * k + wv;
          f32x4 raw[S], B[S], part[XT];
#pragma unroll
          for (int xt = 0; xt < XT; ++xt) {
            f32x4 c = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < x_live(xt); ++r) c = mfma4(w0n[k][xt][r], xb[0][xt][r], c);
            part[xt] = c;
          }
          raw[0] = (part[0] + part[1]) + part[2];
          opt_st4(z0r, kt * 1024 + lane * 16, raw[0]);
          if (S1 == 3) {
#pragma unroll
            for (int d = 0; d < 3; ++d) raw[1 + d] = tcn[k][d];
#pragma unroll
            for (int p = 0; p < S2; ++p) raw[4 + p] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
          act_jet_fwd<S1, S2, ACT>(a.cfg, raw, B, cq);
          store_block(B, buf ^ 1, NW * k + wv);
        }
      } else {
        if (STPDE_ABLATE != 5) produce_group(gnext, buf ^ 1);
      }
      if (gi < 8) STAMP(4 + gi);
      if (STPDE_ABLATE != 2) __syncthreads();
    }
    STAMP(12);
    if (STPDE_ABLATE == 4) {     // keep the accumulators alive without the epilogue
      f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int mi = 0; mi < MCg; ++mi)
#pragma unroll
        for (int st = 0; st < S; ++st) sum += acc[mi][st];
      if (sum[0] == 12345.678f) st4(a.Out + lo, sum);
      continue;
    }
    // exact-fp32 input-gradient kernels of the hidden layers (round 4): the stashed pre-activation blocks of output tile
    // mi + 1 are requested before the adjoint of tile mi is computed (they were loaded tile by tile right in front of their
    // use: MCg exposed HBM round trips per pass with one co-resident workgroup to cover them).  STPDE_EPI_PF=0: as before.
    // (round 4: also the plain bf16 kernels with their packed stash, STPDE_EPI_PF_BF)
    constexpr bool EPF = EPI == EPI_ADJ && (!BF || (SPL == 1 && STPDE_EPI_PF_BF)) && MCg > 1 && STPDE_EPI_PF;
    if constexpr (EPF) {
      constexpr int PMD = (PKM & 4) ? 1 : 0;
      // PFD tiles ahead (STPDE_EPI_PF_DEPTH, clamped to what the pass has; 1 = round 4's one-ahead order)
      constexpr int PFD = STPDE_EPI_PF_DEPTH < 1 ? 1 : (STPDE_EPI_PF_DEPTH > MCg ? MCg : STPDE_EPI_PF_DEPTH);
      f32x4 pr[MCg][S];
#pragma unroll
      for (int mi = 0; mi < PFD; ++mi)
#pragma unroll
        for (int st = 0; st < S; ++st) pr[mi][st] = ld_blk(a.Pre, PMD, tile, S, MT, st, mt0 + mi, lane);
#pragma unroll
      for (int mi = 0; mi < MCg; ++mi) {
        if (mi + PFD < MCg) {
#pragma unroll
          for (int st = 0; st < S; ++st) pr[mi + PFD][st] = ld_blk(a.Pre, PMD, tile, S, MT, st, mt0 + mi + PFD, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        layer_epilogue<S1, S2, EPI, ACT, PKM>(a, tile, mt0 + mi, MT, lane, acc[mi], xb, cq, pacc, nullptr, pr[mi]);
      }
    } else if constexpr (EPI == EPI_ADJ_L0 && S1 == 3 && S <= 5 && (STPDE_EPI_OPF & 1) && (SPL == 1 || (STPDE_EPI_OPF & 4))) {
      // first hidden layer's input gradient (round 5): the layer-0 tangent constants (and, where they are not requested at
      // the start of the pass, the z0 blocks) of the output tiles are requested ahead of the adjoint that needs them -- they
      // were loaded tile by tile right in front of their use, one exposed L2 (+ HBM) round trip per output tile.  SPL == 1:
      // all tiles of the pass in one batch; three-term split mode (behind bit 2 of the switch): one tile ahead, pinned.
      if constexpr (SPL == 1) {
        f32x4 z0l[MCg], tcl[MCg][3];
#pragma unroll
        for (int mi = 0; mi < MCg; ++mi) {
          z0l[mi] = Z0P ? z0p[Z0P ? mi : 0] : ld4(a.Z0 + ((size_t)tile * MT + mt0 + mi) * 256 + lo);
#pragma unroll
          for (int d = 0; d < 3; ++d) tcl[mi][d] = ld4(a.tanc0 + ((size_t)d * MT + mt0 + mi) * 256 + lo);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < MCg; ++mi)
          layer_epilogue<S1, S2, EPI, ACT, PKM>(a, tile, mt0 + mi, MT, lane, acc[mi], xb, cq, pacc, &z0l[mi], nullptr, nullptr,
                                                  tcl[mi]);
      } else {
        f32x4 zc, zn, tcc[3], tcn2[3];
        auto fetch = [&](int mt, f32x4& z, f32x4* t) {
          z = ld4(a.Z0 + ((size_t)tile * MT + mt) * 256 + lo);
#pragma unroll
          for (int d = 0; d < 3; ++d) t[d] = ld4(a.tanc0 + ((size_t)d * MT + mt) * 256 + lo);
        };
        __builtin_amdgcn_sched_barrier(0);
        fetch(mt0, zc, tcc);
#pragma unroll
        for (int mi = 0; mi < MCg; ++mi) {
          if (mi + 1 < MCg) fetch(mt0 + mi + 1, zn, tcn2);
          __builtin_amdgcn_sched_barrier(0);
          layer_epilogue<S1, S2, EPI, ACT, PKM>(a, tile, mt0 + mi, MT, lane, acc[mi], xb, cq, pacc, &zc, nullptr, nullptr, tcc);
          __builtin_amdgcn_sched_barrier(0);
          zc = zn;
#pragma unroll
          for (int d = 0; d < 3; ++d) tcc[d] = tcn2[d];
        }
      }
    } else if constexpr (EPI == EPI_FWD && S1 == 3 && !VT && S <= 5 && (STPDE_EPI_OPF & 2) && (SPL == 1 || (STPDE_EPI_OPF & 4))) {
      // forward: skip weights and tangent constants of output tile mi + 1 requested before the epilogue of tile mi (one
      // exposed L2 round trip per output tile otherwise)
      f32x4 wsc[XT], tcc[3], wsn[XT], tcn2[3];
      auto fetch = [&](int mt, f32x4* w, f32x4* t) {
#pragma unroll
        for (int xt = 0; xt < XT; ++xt) w[xt] = ld4x(a.Wsp + ((size_t)xt * MT + mt) * 256 + lo, xt);
#pragma unroll
        for (int d = 0; d < 3; ++d) t[d] = ld4(a.tanc + ((size_t)d * MT + mt) * 256 + lo);
      };
      if (SPL != 1) __builtin_amdgcn_sched_barrier(0);
      fetch(mt0, wsc, tcc);
#pragma unroll
      for (int mi = 0; mi < MCg; ++mi) {
        if (mi + 1 < MCg) fetch(mt0 + mi + 1, wsn, tcn2);
        __builtin_amdgcn_sched_barrier(0);
        layer_epilogue<S1, S2, EPI, ACT, PKM>(a, tile, mt0 + mi, MT, lane, acc[mi], xb, cq, pacc, nullptr, nullptr, wsc, tcc);
        if (SPL != 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int xt = 0; xt < XT; ++xt) wsc[xt] = wsn[xt];
#pragma unroll
        for (int d = 0; d < 3; ++d) tcc[d] = tcn2[d];
      }
    } else {
#pragma unroll
    for (int mi = 0; mi < MCg; ++mi)
      layer_epilogue<S1, S2, EPI, ACT, PKM>(a, tile, mt0 + mi, MT, lane, acc[mi], xb, cq, pacc,
                                              Z0P ? &z0p[mi] : nullptr);
    }
    STAMP(13);
  }
  flush_pbar<EPI, ACT>(a, pacc, lane);
}

#include "jet_spec_bf16.h"

template <int S1, int S2, int MCg, int PRO, int EPI, int ACT, int NW>
static int launch_layer_coop(const LayerArgs& a0, hipStream_t stream) {
  LayerArgs a = a0;
  // bf16 operands, first hidden layer of the reference width: the wave-specialised persistent kernel (jet_spec_bf16.h)
  if constexpr (PRO == PRO_L0 && EPI == EPI_FWD && NW == 4 && MCg == 4 && S1 == 3 && S2 <= 2) {
    if (a.Wp16 && a.nsplit == 1 && a.MT == 16 && (a.KT == 32 || a.KT == 16))
      return launch_fc1_fwd_spec<S1, S2, ACT>(a, stream);
  }
  // ... and the forward of the second hidden layer (round 5): k_fc2_fwd_bf.  (The input gradient of the first hidden layer
  // in that mode is the fused kernel of jet_fc1_bwd.hip, sequenced by lig_pipeline.hip; this cooperative kernel serves a
  // backward without weight gradients and the stream sets the fused kernel is not compiled for.)
  if constexpr (PRO == PRO_ACT && EPI == EPI_FWD && NW == 4 && MCg == 2 && S1 == 3 && S2 <= 1) {
    if (a.Wp16 && a.nsplit == 1 && a.pk == 3 && a.KT == 16 && a.MT == 8)
      return launch_fc2_fwd_bf<S1, S2, ACT>(a, stream);
  }
  const int npass = a.MT / (NW * MCg);
  // kernels that stream their B operand from the stash and need several output passes: one workgroup per pass
  a.split = (PRO != PRO_L0 && npass > 1) ? npass : 0;
  const int nblocks = a.split ? (a.ntiles + 7) / 8 * 8 * a.split : a.ntiles;
  if (a.pk && !(a.Wp16 && a.nsplit == 1)) {
    stpde_set_error("packed layer buffers are a bf16-operand mode (stpde_layer_desc.mfma_bf16 == 1)");
    return STPDE_E_UNSUPPORTED;
  }
  // "fp32x3" is a contract on accuracy (fp32), not on the pipe: the split kernels are compiled for the wide-layer shapes of the
  // training stream sets; every other stream set / workgroup shape takes the exact-fp32 kernels (round 4; used to be refused)
  if (a.Wp16 && a.nsplit == 3 && !(S1 + S2 <= 4 && NW == 4)) a.Wp16 = nullptr;
  if (a.Wp16 && a.nsplit == 3) {
    if constexpr (S1 + S2 <= 4 && NW == 4)
      STPDE_LAUNCH((k_layer_coop<S1, S2, MCg, PRO, EPI, ACT, NW, true, 1, false, 3>), dim3(nblocks), dim3(64 * NW), 0, stream, a);
  } else if (a.Wp16) {
    // the packed-buffer mask this (PRO, EPI) kind may be launched with (stpde_layer_desc.packed, mapped by jet_layer.hip)
    // (bf16 mode packs the buffers of fc1's AND fc2's rows: fc1 forward writes one, fc2 forward reads one and writes one, ...)
    constexpr int PKA = (PRO == PRO_L0 && EPI == EPI_FWD) ? 2 : (PRO == PRO_ACT ? 3 : (EPI == EPI_ADJ ? 7 : 3));
    if (a.pk != 0 && a.pk != PKA) {
      stpde_set_error("packed layer buffers: combination %d not compiled for this kernel kind (expects %d)", a.pk, PKA);
      return STPDE_E_UNSUPPORTED;
    }
    if (a.KT % (2 * NW) == 0) {
      if (a.pk)
        STPDE_LAUNCH((k_layer_coop<S1, S2, MCg, PRO, EPI, ACT, NW, true, 2, false, 1, PKA>), dim3(nblocks), dim3(64 * NW), 0, stream, a);
      else
        STPDE_LAUNCH((k_layer_coop<S1, S2, MCg, PRO, EPI, ACT, NW, true, 2>), dim3(nblocks), dim3(64 * NW), 0, stream, a);
    } else {
      if (a.pk)
        STPDE_LAUNCH((k_layer_coop<S1, S2, MCg, PRO, EPI, ACT, NW, true, 1, false, 1, PKA>), dim3(nblocks), dim3(64 * NW), 0, stream, a);
      else
        STPDE_LAUNCH((k_layer_coop<S1, S2, MCg, PRO, EPI, ACT, NW, true>), dim3(nblocks), dim3(64 * NW), 0, stream, a);
    }
  }
  else {
    // 4 output tiles per wave: weight fragments through the 3-deep register ring (-2.5 % on the forward of the widest
    // layer; no gain for the 2-tile-per-wave shapes).  (The cooperative kernels are launched with KT >= 8 only.)
    if constexpr (NW == 4 && MCg == 4 && S1 + S2 <= 5) {      // (S = 8: the ring's 48 fragment registers spill)
      STPDE_LAUNCH((k_layer_coop<S1, S2, MCg, PRO, EPI, ACT, NW, false, 1, true>), dim3(nblocks), dim3(64 * NW), 0, stream, a);
    } else {
      STPDE_LAUNCH((k_layer_coop<S1, S2, MCg, PRO, EPI, ACT, NW, false>), dim3(nblocks), dim3(64 * NW), 0, stream, a);
    }
  }
  return stpde_check_launch("k_layer_coop");
}

template <int S1, int S2, int PRO, int EPI, int MCg, int NW>
static int launch_coop_act(const LayerArgs& a, hipStream_t stream) {
  switch (a.cfg.act) {
    case STPDE_ACT_TANH: return launch_layer_coop<S1, S2, MCg, PRO, EPI, STPDE_ACT_TANH, NW>(a, stream);
    case STPDE_ACT_RELU: return launch_layer_coop<S1, S2, MCg, PRO, EPI, STPDE_ACT_RELU, NW>(a, stream);
    case STPDE_ACT_SOFTPLUS: return launch_layer_coop<S1, S2, MCg, PRO, EPI, STPDE_ACT_SOFTPLUS, NW>(a, stream);
    case STPDE_ACT_ELU: return launch_layer_coop<S1, S2, MCg, PRO, EPI, STPDE_ACT_ELU, NW>(a, stream);
    case STPDE_ACT_LEAKYRELU: return launch_layer_coop<S1, S2, MCg, PRO, EPI, STPDE_ACT_LEAKYRELU, NW>(a, stream);
    default: return launch_layer_coop<S1, S2, MCg, PRO, EPI, STPDE_ACT_SWISH, NW>(a, stream);
  }
}

template <int S1, int S2, int MC, int PRO, int EPI, int ACT, bool GUARD>
static int launch_layer(const LayerArgs& a, hipStream_t stream) {
  if (a.pk) {
    stpde_set_error("packed layer buffers need the cooperative bf16 kernels (shape not served by them)");
    return STPDE_E_UNSUPPORTED;
  }
  dim3 grid((a.ntiles + 3) / 4);
  STPDE_LAUNCH((k_layer<S1, S2, MC, PRO, EPI, ACT, GUARD>), grid, dim3(256), 0, stream, a);
  return stpde_check_launch("k_layer");
}

template <int S1, int S2, int PRO, int EPI, int MC>
static int launch_fwd_act_mc(const LayerArgs& a, hipStream_t stream) {
  switch (a.cfg.act) {
    case STPDE_ACT_TANH: return launch_layer<S1, S2, MC, PRO, EPI, STPDE_ACT_TANH, false>(a, stream);
    case STPDE_ACT_RELU: return launch_layer<S1, S2, MC, PRO, EPI, STPDE_ACT_RELU, false>(a, stream);
    case STPDE_ACT_SOFTPLUS: return launch_layer<S1, S2, MC, PRO, EPI, STPDE_ACT_SOFTPLUS, false>(a, stream);
    case STPDE_ACT_ELU: return launch_layer<S1, S2, MC, PRO, EPI, STPDE_ACT_ELU, false>(a, stream);
    case STPDE_ACT_LEAKYRELU: return launch_layer<S1, S2, MC, PRO, EPI, STPDE_ACT_LEAKYRELU, false>(a, stream);
    default: return launch_layer<S1, S2, MC, PRO, EPI, STPDE_ACT_SWISH, false>(a, stream);
  }
}

template <int S1, int S2, int PRO, int EPI>
static int launch_fwd_act(const LayerArgs& a, hipStream_t stream) {
  // the two narrowest layers (fc4: 2 output tiles, fc5: 1): exact tile count and compile-time activation
  if (a.MT == 2 && S1 + S2 <= 5) return launch_fwd_act_mc<S1, S2, PRO, EPI, 2>(a, stream);
  if (a.MT == 1 && S1 + S2 <= 5) return launch_fwd_act_mc<S1, S2, PRO, EPI, 1>(a, stream);
  if (a.MT % 4 != 0) return launch_layer<S1, S2, 4, PRO, EPI, -1, true>(a, stream);
  // workgroup-cooperative variant: B operand produced once per 4 waves (S = 10 would not fit two workgroups of LDS)
  if (a.KT % 4 == 0 && a.KT >= 8 && S1 + S2 <= 5) {
    if (EPI == EPI_FWD) {
      if (a.MT % 16 == 0) return launch_coop_act<S1, S2, PRO, EPI, 4, 4>(a, stream);
      if (a.MT % 8 == 0) return launch_coop_act<S1, S2, PRO, EPI, 2, 4>(a, stream);
    } else {
      // dgrad epilogues are VALU heavy: 2 output tiles per wave keeps two waves per SIMD.  Measured on MI355X: the
      // 8-wave workgroup (one pass over 16 output tiles) wins when it makes the kernel single-pass (MT == 16);
      // for MT == 32 four passes of the 4-wave workgroup are faster than two passes of the 8-wave one.
      // and 4 output tiles per wave with the weight ring beat both for MT % 16 == 0 (layers 1 and 2 of the reference net)
      if (a.MT % 16 == 0 && S1 + S2 <= 4) return launch_coop_act<S1, S2, PRO, EPI, 4, 4>(a, stream);
      if (a.KT % 8 == 0 && a.MT == 16) return launch_coop_act<S1, S2, PRO, EPI, 2, 8>(a, stream);
      if (a.MT % 8 == 0) return launch_coop_act<S1, S2, PRO, EPI, 2, 4>(a, stream);
    }
  }
  // many-stream sets (S = 10, configs[4]): the cooperative kernel with two output tiles per wave, one workgroup per CU (its
  // ring takes 80 KB) instead of the per-wave kernel below (round 4: configs[4] step 994 -> 892 ms, first-layer forward 251 -> 183 ms)
  if constexpr (S1 + S2 > 5) {
    // S = 8 (round 5, the (3,4) set of configs[4]): its forward kernels also fit four output tiles per wave without scratch
    // (250 registers, 64 KB ring: two workgroups per CU) -- half the ring reads per MFMA (configs[4] 690.0 -> 682.4 ms)
    if constexpr (S1 + S2 == 7 && EPI == EPI_FWD) {
      if (a.KT % 4 == 0 && a.KT >= 8 && a.MT % 16 == 0) return launch_coop_act<S1, S2, PRO, EPI, 4, 4>(a, stream);
    }
    if (a.KT % 4 == 0 && a.KT >= 8 && a.MT % 8 == 0) return launch_coop_act<S1, S2, PRO, EPI, 2, 4>(a, stream);
  }
  // kernels that stream their B operand from memory (everything except the layer-0-on-the-fly forward) halve that
  // traffic with 8 output tiles per pass; S=10 would not fit the register file
  if (PRO != PRO_L0 && S1 + S2 <= 5 && a.MT % 8 == 0) return launch_fwd_act_mc<S1, S2, PRO, EPI, 8>(a, stream);
  return launch_fwd_act_mc<S1, S2, PRO, EPI, 4>(a, stream);
}

// mode: 0 = fwd (hidden input from stash), 1 = fwd first hidden (layer 0 on the fly), 2 = dgrad, 3 = dgrad into layer 0
template <int S1, int S2>
static int launch_mode(const LayerArgs& a, int mode, hipStream_t stream) {
  if constexpr (S1 == 0 && S2 > 0) {   // value-tile mode is forward-only (inference)
    if (mode == 0) return launch_fwd_act<S1, S2, PRO_ACT, EPI_FWD>(a, stream);
    if (mode == 1) return launch_fwd_act<S1, S2, PRO_L0, EPI_FWD>(a, stream);
    stpde_set_error("value-tile stream configuration (S1 = 0, S2 > 0) has no backward kernels");
    return STPDE_E_UNSUPPORTED;
  } else {
    switch (mode) {
      case 0: return launch_fwd_act<S1, S2, PRO_ACT, EPI_FWD>(a, stream);
      case 1: return launch_fwd_act<S1, S2, PRO_L0, EPI_FWD>(a, stream);
      case 2: return launch_fwd_act<S1, S2, PRO_NONE, EPI_ADJ>(a, stream);
      default: return launch_fwd_act<S1, S2, PRO_NONE, EPI_ADJ_L0>(a, stream);
    }
  }
}

#define STPDE_DEFINE_LAYER_TU(S1, S2) \
  int stpde_layer_launch_##S1##_##S2(const LayerArgs& a, int mode, hipStream_t stream) { return launch_mode<S1, S2>(a, mode, stream); }

int stpde_layer_launch_0_0(const LayerArgs& a, int mode, hipStream_t stream);
int stpde_layer_launch_3_0(const LayerArgs& a, int mode, hipStream_t stream);
int stpde_layer_launch_3_1(const LayerArgs& a, int mode, hipStream_t stream);
int stpde_layer_launch_3_2(const LayerArgs& a, int mode, hipStream_t stream);
int stpde_layer_launch_3_4(const LayerArgs& a, int mode, hipStream_t stream);
int stpde_layer_launch_3_6(const LayerArgs& a, int mode, hipStream_t stream);
int stpde_layer_launch_0_3(const LayerArgs& a, int mode, hipStream_t stream);   // value-tile mode: 4 row tiles per pass
