// Weight gradient of one IM-NET layer over all derivative streams:
//   dW_aug[m][k] += sum_{rows, streams} abar[row][m] * hin[row][k],   hin = [act_jet(in_pre) ; X_aug]
// (the contribution loss.backward() -- experiments/rb2d/train.py:77 -- makes to fc_l.weight / fc_l.bias through
// src/implicit_net.py:48-54, including the second-order terms of the src/pde.py:8-9 sweeps).
//
// The contraction runs over corner rows, but the stash holds column-major fragment images (rows on lanes), so both
// operands are turned into the row-major image (lane 16g+c: rows 4g..4g+3 of feature c = the A/B register image of
// v_mfma_f32_16x16x4_f32 for dW = P^T Q) through LDS, cooperatively:
//   * a workgroup = 8 waves (2 per SIMD) walks a strided set of row tiles in lock step and owns 8 k-tiles x
//     (2*KC) m-tiles of dW: wave w = (m-slot w % KC, k-slot w / KC) accumulates a 2 x KC block of 16x16 tiles;
//   * produce stage: wave w builds ring slot w of the NEXT row tile -- loads the S pre-activation blocks of its
//     k-tile, applies the activation jet ONCE for the whole workgroup, and writes them transposed (padded
//     feature-major LDS blocks); for the first hidden layer the value stream of the pre-activations comes from the z0
//     stash the forward kernel wrote and the tangent streams are the constant columns W0[:, d];
//   * consume stage: every wave reads the ring slots of its k-slot and runs 2*KC*4*S MFMAs against its own abar
//     blocks (transposed once per tile through a private patch);
//   * double-buffered ring, one barrier per row tile; partial sums are merged with fp32 atomics at the end.
// Block order is XCD-aware (all k-groups of a tile range share one XCD's L2).
#pragma once
#include <cstdlib>

#include "common.h"

struct WgradArgs {
  const float* P;      // abar_out, column-major image [tile][SP][MT][256]
  const float* Q;      // pre-activations of the layer input, column-major image [tile][S][KT][256]   (MODE 0);
                       // MODE 1 (first hidden layer): the z0 stash [tile][KT][256] = value stream of layer 0's pre-activations
  const float* X;      // augmented raw input [tile][XT][256], COLUMN-major image (round 5: the row-major copy XR the gather kernel
                       // used to write -- 1.5 KiB per point, re-read by six kernels -- is gone; every consumer turns the fragment
                       // it needs into the row-major image through an LDS patch, as it does for every other operand)
  const float* tanc0;  // [3][KT][256] layer-0 tangent constants W0[:, d], column-major image (MODE 1)
  float* dW;           // [16*MT][16*(KT+XT)]
  const float* cw;     // [P][8] weights of the combined second-order stream (S2 == 1)
  int SP, KT, MT, ntiles;
  int gx, gy, gz;      // logical grid: tile splits (multiple of 8), m-groups, k-groups of 8 tiles
  int kz0;             // first k-group of this launch (hidden-only groups and raw-input groups are launched separately)
  int det;             // dW addresses long accumulators (stpde_layer_desc.det, common.h: acc_add_f32)
  int bf16;            // != 0: contract with bf16-rounded operands (v_mfma_f32_16x16x32_bf16), fp32 accumulation
  int pk;              // packed-buffer flags (common.h: ld_blk): 1 = Q (the layer input's pre-activations), 4 = P (abar_out)
  int xfold;           // fp32 hidden-group launches: the (k-group, k-slot) pairs 0 .. XT-1 also contract their abar blocks with
                       // raw-input tile 0 .. XT-1 (one extra 16x16 tile per wave, the XR fragment straight from memory), and all
                       // of them keep the row sums of the tangent-stream adjoints (the tangent "input" of a skip connection is the
                       // unit vector e_d): no separate launch for the three raw-input k-tiles, which re-read and re-transposed every
                       // abar block of the layer for 3 / 35 of the MFMA work (9.3 + 5.9 ms per step for layers 1 and 2)
  stpde_jet_cfg cfg;
};

// Timing-only ablations (tools/micro/ablate_wgrad.py, private builds with -DSTPDE_ABLATE_W=n; results are WRONG):
// 1 = one partial product instead of six, 2 = abar blocks transposed / split for the first tile only, 3 = no produce stage
// in the loop, 4 = consumer operands not read from the LDS ring, 5 = no barrier in the loop.
#ifndef STPDE_QUAD_PIPE
#define STPDE_QUAD_PIPE 1
#endif
#ifndef STPDE_ABLATE_W
#define STPDE_ABLATE_W 0
#endif
#ifndef STPDE_X3_XFOLD
#define STPDE_X3_XFOLD 1     // fp32x3: raw-input k-tiles folded into the hidden-group launch of the split kernel (round 5)
#endif
#ifndef STPDE_WG_RPIPE
#define STPDE_WG_RPIPE 1     // exact-fp32 cooperative kernel: ring blocks read ONE BLOCK AHEAD of their MFMAs (round 5)
#endif
#ifndef STPDE_X3_EARLYP
#define STPDE_X3_EARLYP 0
#endif
// Padded row length TP (floats) of a transposed (feature-major) LDS block.  Under the per-instruction banking of gfx950
// (MI355X_MICROARCH.md, LDS): TP = 24 makes the ds_read_b128 of a block conflict-free and its four ds_write_b32 2-way
// conflicted, which a ds_write_b32 hides (its cost is the VGPR transfer); TP = 20 is the other way round (writes free, reads 2-way:
// SQ_LDS_BANK_CONFLICT was 38 % of the LDS-array cycles of the first-layer kernel) and is kept where 24 would not fit two ring
// buffers in LDS.
template <int TP>
__device__ __forceinline__ void lds_put_T(float* blk, int lane, f32x4 v) {   // column-major image -> feature-major block
  const int g = lane >> 4, j = lane & 15;
  float* w = blk + (4 * g) * TP + j;
  w[0] = v[0];
  w[TP] = v[1];
  w[2 * TP] = v[2];
  w[3 * TP] = v[3];
}
template <int TP>
__device__ __forceinline__ void lds_put_R(float* blk, int lane, f32x4 v) {   // row-major image -> the same block format
  *reinterpret_cast<f32x4*>(blk + (lane & 15) * TP + 4 * (lane >> 4)) = v;
}
template <int TP>
__device__ __forceinline__ f32x4 lds_get_R(const float* blk, int lane) {     // lane 16g+c: rows 4g..4g+3 of feature c
  return *reinterpret_cast<const f32x4*>(blk + (lane & 15) * TP + 4 * (lane >> 4));
}

// BF: one v_mfma_f32_16x16x32_bf16 contracts over the 16 rows of the tile for TWO derivative streams (k-slot e of a
// lane = stream parity e >> 2, row 4g + (e & 3)), operands rounded to bf16 when they leave LDS / the patch.
// SPL = 3 (with BF): fp32-accurate contraction on the bf16 pipe -- both operands are split exactly into three bf16 terms
// (hi + mid + lo) and the six partial products of weight >= 2^-16 are accumulated (see k_layer_coop); the abar blocks are
// split once per tile when they are packed, the activated-input blocks when they leave the LDS ring (up to two VALU
// instructions issue for free behind every bf16 MFMA).
// PKM: compile-time packed-buffer mask (WgradArgs.pk: 1 = Q is a packed STASH, 4 = P is a packed ADJOINT buffer; common.h)
template <int S1, int S2, int MODE, int ACT, int KC, bool HASX, bool BF = false, int SPL = 1, int PKM = 0>
__global__ __launch_bounds__(512, 2) void k_wgrad_coop(WgradArgs a) {
  static_assert(SPL == 1 || (BF && SPL == 3), "operand splitting is a bf16-pipe mode");
  constexpr int PMODE = (PKM & 4) ? 2 : 0;      // format of the adjoint operand's buffer (common.h)
  constexpr int S = 1 + S1 + S2, MCW = 2, NW = 8, RS = 8;
  constexpr int NM = KC;                    // m-slots per workgroup; k-slots = NW / NM, each KC ring slots wide
  // SPLP (split mode): the ring holds the three bf16 terms of every activated-input fragment, split ONCE by the producing
  // wave: block = [term][feature][16 rows] bf16 (1536 B); a consumer lane (g, c) reads rows 4g..4g+3 of feature c of term t
  // with one ds_read_b64 -- the same layout serves the column-major producers (transposing ds_write_b16) and the
  // row-major ones (ds_write_b64).
  constexpr bool SPLP = BF && SPL == 3;
  // bf16-pipe modes: every LDS block (ring slots and the private abar patches) holds the SPL bf16 terms of a fragment as
  // [term][16][16] bf16 (512 B per term), written with ONE ds_write_b64 per lane and term at (lane & 15) * 16 + 4 * (lane >> 4)
  // -- which is [row][feature] for a column-major source and [feature][row] for a row-major one -- and read back as the
  // row-major operand fragment with the hardware transpose read ds_read_b64_tr_b16 (column-major sources) or a plain
  // ds_read_b64 (row-major sources).  (First half of round 2: fp32 blocks transposed with four ds_write_b32 per lane and
  // converted / split by every consumer; in plain bf16 mode those transposes were 44 % of the kernel.)
  constexpr int TP = (2 * RS * S * 16 * 24 * 4 + NW * 2 * 16 * 24 * 4 <= 150 * 1024) ? 24 : 20;
  constexpr int TBLK = 16 * TP;
  constexpr int BLK = BF ? 128 * SPL : TBLK;
  constexpr int NBUF = (2 * RS * S * BLK * 4 + NW * 2 * BLK * 4 <= 150 * 1024) ? 2 : 1;
  constexpr int SX = S1 == 3 ? 4 : 1;       // raw-input tiles only feed the value and tangent streams
  __shared__ __attribute__((aligned(16))) float hl[NBUF][RS][S][BLK];
  // plain bf16 mode: one patch per adjoint block of a wave (S * MCW of them) -- all blocks of a tile are written, then all
  // are read back transposed: ONE LDS round trip per tile and wave instead of one per block (the waves of this kernel are
  // bound by their dependent chains, not by pipe throughput: profiles/r4_wgrad_bf16_steps.txt)
  constexpr int NPP = (BF && SPL == 1 && 2 * RS * S * 512 + NW * S * MCW * 512 <= 100 * 1024) ? S * MCW : 2;
  __shared__ __attribute__((aligned(16))) float pp[NW][NPP][BLK];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int lo = lane * 4;
  const int KT = a.KT, MT = a.MT, SP = a.SP;
  const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
  const int kz = (slot % a.gz) + a.kz0, tq = slot / a.gz;
  const int mg = tq % a.gy, bx = (tq / a.gy) * 8 + xcd;
  const int ms = wv % NM, ks = wv / NM;
  const int mt0 = (mg * NM + ms) * MCW;
  const int kq0 = kz * RS;
  const int g = lane >> 4, c = lane & 15;

  f32x4 acc[MCW][KC];
#pragma unroll
  for (int mi = 0; mi < MCW; ++mi)
#pragma unroll
    for (int ki = 0; ki < KC; ++ki) acc[mi][ki] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto put = [&](float* blk, f32x4 v, bool row_major) {
    if constexpr (BF) {
      __bf16* hb16 = reinterpret_cast<__bf16*>(blk);
#pragma unroll
      for (int t = 0; t < SPL; ++t) {
        const bf16x4 h = to_bf4(v);
        *reinterpret_cast<bf16x4*>(hb16 + (t * 16 + (lane & 15)) * 16 + 4 * (lane >> 4)) = h;
        if (t + 1 < SPL) v -= bf4_to_f32(h);   // exact: the residual of a round-to-nearest bf16 fits in fp32
      }
    } else if (row_major) {
      lds_put_R<TP>(blk, lane, v);
    } else {
      lds_put_T<TP>(blk, lane, v);
    }
  };
  auto get = [&](const float* blk) -> f32x4 { return lds_get_R<TP>(blk, lane); };
  // bf16-pipe modes: term t of this lane's row-major operand fragment (rows 4g..4g+3 of feature c) out of a block written
  // by put(); cm = the block came from a column-major source ([row][feature] image -> transpose read)
  auto get16 = [&](const float* blk, int t, bool cm = true) -> bf16x4 {
    const __bf16* b16 = reinterpret_cast<const __bf16*>(blk) + t * 256;
    if (cm) return lds_read_tr16(b16 + (4 * (lane >> 4) + ((lane & 15) >> 2)) * 16 + 4 * (lane & 3));
    return *reinterpret_cast<const bf16x4*>(b16 + (lane & 15) * 16 + 4 * (lane >> 4));
  };

  // folded raw-input fragment (XF / XB / XS): the column-major block `v` of X -> this lane's row-major operand fragment (rows
  // 4g .. 4g+3 of feature c) through the wave's private patch; bf16-pipe modes: SPL bf16 terms
  auto x_rowmajor = [&](f32x4 v) -> f32x4 {
    float* patch = pp[wv][0];
    __builtin_amdgcn_wave_barrier();
    lds_put_T<TP>(patch, lane, v);
    __builtin_amdgcn_wave_barrier();
    const f32x4 r = lds_get_R<TP>(patch, lane);
    __builtin_amdgcn_wave_barrier();
    return r;
  };
  auto x_rowmajor16 = [&](f32x4 v, bf16x4* t) {
    float* patch = pp[wv][0];
    __builtin_amdgcn_wave_barrier();
    put(patch, v, false);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < (BF ? SPL : 1); ++k) t[k] = get16(patch, k, true);
    __builtin_amdgcn_wave_barrier();
  };
  // produce ring slot `wv` (k-tile kq0 + wv) of row tile `tile` into buffer `buf`, in two halves: load_q ISSUES the stash
  // loads (before the MFMAs of the current tile), finish_q applies the activation jet and writes the ring slot (after them),
  // so the loads' latency hides behind the MFMAs instead of sitting in front of the barrier
  auto load_q = [&](int tile, f32x4* pre, float* cq) {
    const int kq = kq0 + wv;
    if (!HASX || kq < KT) {
      // rows of this lane (column-major image): row = lane & 15
      load_cq<S2>(a.cw, tile * 2 + (c >> 3), cq);
      if (MODE == 1) {
        // first hidden layer: value stream from the z0 stash (written by the forward; round 2 regenerated it here with
        // 12 MFMAs per k-tile), tangent streams = the constant columns W0[:, d], second-order streams = 0
        pre[0] = ld4(a.Q + ((size_t)tile * KT + kq) * 256 + lo);
        if (S1 == 3) {
#pragma unroll
          for (int d = 0; d < 3; ++d) pre[1 + d] = ld4(a.tanc0 + ((size_t)d * KT + kq) * 256 + lo);
#pragma unroll
          for (int p = 0; p < S2; ++p) pre[4 + p] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      } else {
#pragma unroll
        for (int st = 0; st < S; ++st) pre[st] = ld_blk(a.Q, (PKM & 1) ? 1 : 0, tile, S, KT, st, kq, lane);
      }
    } else if (kq < KT + XT) {
      pre[0] = ld4(a.X + ((size_t)tile * XT + (kq - KT)) * 256 + lo);
    }
  };
  auto finish_q = [&](const f32x4* pre, const float* cq, int buf) {
    const int kq = kq0 + wv;
    if (!HASX || kq < KT) {
      f32x4 H[S];
      act_jet_fwd<S1, S2, ACT>(a.cfg, pre, H, cq);
#pragma unroll
      for (int st = 0; st < S; ++st) put(&hl[buf][wv][st][0], H[st], false);
    } else if (kq < KT + XT) {
      const int xt = kq - KT;
      put(&hl[buf][wv][0][0], pre[0], false);          // (column-major source, like every hidden k-tile)
      if (S1 == 3) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          // tangent stream d sees the unit vector e_d: in the column-major image lane (g, row) holds features 4g .. 4g+3
          f32x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (xt == 0 && 4 * g + r == d) ? 1.f : 0.f;
          put(&hl[buf][wv][1 + d][0], v, false);
        }
      }
    }
  };
  auto produce = [&](int tile, int buf) {
    f32x4 pre[S];
    float cq[6];
    load_q(tile, pre, cq);
    finish_q(pre, cq, buf);
  };
  // this wave's abar blocks of row tile `tile`: column-major loads, transposed through the private patch
  auto load_p_raw = [&](int tile, f32x4 (*raw)[MCW]) {
#pragma unroll
    for (int st = 0; st < S; ++st)
#pragma unroll
      for (int mi = 0; mi < MCW; ++mi) {
        const int mt = mt0 + mi < MT ? mt0 + mi : MT - 1;
        raw[st][mi] = st < SP ? ld_blk_raw(a.P, PMODE, tile, SP, MT, st, mt, lane) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
  };
  auto transpose_p = [&](f32x4 (*raw)[MCW], f32x4 (*pa)[MCW]) {
    if constexpr (BF) return;      // bf16-pipe modes: pack_p transposes the bf16 terms (hardware transpose read)
#pragma unroll
    for (int st = 0; st < S; ++st)
#pragma unroll
      for (int mi = 0; mi < MCW; ++mi) {
        float* patch = pp[wv][(st * MCW + mi) & 1];
        lds_put_T<TP>(patch, lane, blk_val(raw[st][mi], st < SP ? PMODE : 0, st));
        __builtin_amdgcn_wave_barrier();
        pa[st][mi] = lds_get_R<TP>(patch, lane);
        __builtin_amdgcn_wave_barrier();
      }
  };

  constexpr int SH = (S + 1) / 2, SXH = (SX + 1) / 2;
  bf16x8 pa8[BF ? SPL : 1][BF ? SH : 1][MCW];
  const bf16x4 zero4 = to_bf4(f32x4{0.f, 0.f, 0.f, 0.f});
  // bf16-pipe modes: the column-major abar blocks `raw_` of the tile -> split into bf16 terms, transposed through this
  // wave's private patch (one ds_write_b64 + one transpose read per term), packed two streams per MFMA operand
  auto pack_p = [&](f32x4 (*raw_)[MCW]) {
    if constexpr (BF && NPP > 2) {
#pragma unroll
      for (int st = 0; st < S; ++st)
#pragma unroll
        for (int mi = 0; mi < MCW; ++mi) put(pp[wv][st * MCW + mi], blk_val(raw_[st][mi], st < SP ? PMODE : 0, st), false);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int sp = 0; sp < SH; ++sp)
#pragma unroll
        for (int mi = 0; mi < MCW; ++mi)
          pa8[0][sp][mi] = cat8(get16(pp[wv][2 * sp * MCW + mi], 0, true),
                                2 * sp + 1 < S ? get16(pp[wv][(2 * sp + 1 < S ? 2 * sp + 1 : 0) * MCW + mi], 0, true) : zero4);
      __builtin_amdgcn_wave_barrier();
    } else if constexpr (BF) {
      int flip = 0;
      auto terms = [&](f32x4 v, bf16x4* t) {
        float* patch = pp[wv][flip];
        flip ^= 1;
        put(patch, v, false);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < SPL; ++k) t[k] = get16(patch, k, true);
        __builtin_amdgcn_wave_barrier();
      };
#pragma unroll
      for (int sp = 0; sp < SH; ++sp)
#pragma unroll
        for (int mi = 0; mi < MCW; ++mi) {
          bf16x4 t0[SPL], t1[SPL];
          terms(blk_val(raw_[2 * sp][mi], 2 * sp < SP ? PMODE : 0, 2 * sp), t0);
          if (2 * sp + 1 < S) terms(blk_val(raw_[2 * sp + 1][mi], 2 * sp + 1 < SP ? PMODE : 0, 2 * sp + 1), t1);
#pragma unroll
          for (int k = 0; k < SPL; ++k) pa8[k][sp][mi] = cat8(t0[k], 2 * sp + 1 < S ? t1[k] : zero4);
        }
    }
  };
  // acc += P^T H for one pair of streams: one bf16 MFMA (SPL = 1) or the six partial products, smallest first
  auto mma16 = [&](f32x4& c, int sp, int mi, const bf16x8* H8) {
    if constexpr (SPL == 3) {
      constexpr int TP[6] = {1, 0, 2, 0, 1, 0}, TH[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
      for (int q6 = 0; q6 < 6; ++q6) c = mfma_bf(pa8[TP[q6]][sp][mi], H8[TH[q6]], c);
    } else {
      c = mfma_bf(pa8[0][sp][mi], H8[0], c);
    }
  };

  const int stride = a.gx;
  int tile = bx;
  f32x4 pa[S][MCW];
  // folded raw-input tile (xfold): slot index of this wave, its accumulators, the XR fragment of the current tile
  constexpr bool XF = !HASX && !BF;
  // the plain bf16 mode folds the same way (XB): the value product as ONE bf16 MFMA per output tile (raw-input fragment rounded
  // to bf16 like every operand of the mode, the second stream slot of the MFMA zero), the tangent columns as fp32 sums of the
  // column-major abar blocks (per lane and feature over all tiles; folded over the 16 rows at the end).  Not the split mode:
  // its kernel has no registers left (248 VGPRs).
  constexpr bool XB = !HASX && BF && SPL == 1;
  // ... and the three-term split mode (round 5, XS): the raw-input fragment split into three bf16 terms like every operand of
  // that mode (six partial products per output tile: fp32-accurate), the tangent columns through the pattern operand with all
  // three terms of the adjoint.  It used to keep a separate launch for the three raw-input k-tiles that re-read and re-split
  // every adjoint block of the layer (15.8 ms per 2^20 points for the first hidden layer); the registers come from loading
  // the next tile's adjoint blocks right in front of their split instead of a tile ahead (EARLYP below).
  constexpr bool XS = !HASX && BF && SPL == 3 && STPDE_X3_XFOLD;
  const int xslot = (kz - a.kz0) * (NW / NM) + ks;
  const int xsel = xslot < XT ? xslot : XT - 1;
  f32x4 accx[(XF || XB || XS) ? MCW : 1];
  float acct[XF && S1 == 3 ? 3 : 1][XF ? MCW : 1];
  // XB, wave of raw-input tile 0: the tangent columns (the tangent "input" of a skip connection is the unit vector e_d) go
  // through the same accumulator tile as the value product -- stream d of the second operand is the pattern [feature == d]
  // -- two more MFMAs per row tile instead of 24 accumulator registers and 48 VALU instructions (round 4; those registers
  // are what the double-buffered ring reads below needed)
  const bf16x4 one4 = to_bf4(f32x4{1.f, 1.f, 1.f, 1.f});
  const bool tanw = (XB || XS) && S1 == 3 && xslot == 0;
  const bf16x4 e0 = (tanw && c == 0) ? one4 : zero4;
  const bf16x8 x8t = cat8((tanw && c == 1) ? one4 : zero4, (tanw && c == 2) ? one4 : zero4);
  f32x4 xr = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (XF || XB || XS) {
#pragma unroll
    for (int mi = 0; mi < MCW; ++mi) accx[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if constexpr (XF && S1 == 3) {
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int mi = 0; mi < MCW; ++mi) acct[d][mi] = 0.f;
  }
  // value product with the raw-input tile + tangent columns (XB): part of the MFMA phase of a row tile
  auto xb_mma = [&]() {
    if constexpr (XB) {
      bf16x4 xt1[1];
      x_rowmajor16(xr, xt1);
      const bf16x8 x8 = cat8(xt1[0], S1 == 3 ? e0 : zero4);
#pragma unroll
      for (int mi = 0; mi < MCW; ++mi) accx[mi] = mfma_bf(pa8[0][0][mi], x8, accx[mi]);
      if constexpr (S1 == 3) {
#pragma unroll
        for (int mi = 0; mi < MCW; ++mi) accx[mi] = mfma_bf(pa8[0][1][mi], x8t, accx[mi]);
      }
    }
  };
  auto xs_mma = [&]() {
    if constexpr (XS) {
      bf16x4 xt3[3];
      x_rowmajor16(xr, xt3);           // (put() splits into the three terms, the split commutes with the transposition)
      constexpr int TP[6] = {1, 0, 2, 0, 1, 0}, TH[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
      for (int q6 = 0; q6 < 6; ++q6) {
        // second stream slot of the operand: tangent stream 0 against the pattern [feature == 0] (exact in bf16: only the
        // products with the pattern's single term, i.e. TH == 0, carry it -- all three terms of the adjoint)
        const bf16x8 x8 = cat8(xt3[TH[q6]], (S1 == 3 && TH[q6] == 0) ? e0 : zero4);
#pragma unroll
        for (int mi = 0; mi < MCW; ++mi) accx[mi] = mfma_bf(pa8[TP[q6]][0][mi], x8, accx[mi]);
      }
      if constexpr (S1 == 3) {
#pragma unroll
        for (int k = 2; k >= 0; --k)
#pragma unroll
          for (int mi = 0; mi < MCW; ++mi) accx[mi] = mfma_bf(pa8[k][1][mi], x8t, accx[mi]);
      }
    }
  };
  // plain bf16 mode, hidden k-tiles: the MFMAs of a row tile with the ring reads of k-tile ki + 1 requested BEFORE the MFMAs
  // of k-tile ki (the compiler's own schedule re-used ONE operand register set: read, s_waitcnt lgkmcnt(0), two MFMAs, 24
  // times per tile -- ~3,000 cycles of exposed LDS latency per wave and tile, the largest single item of this kernel)
  auto ring_mma = [&](int rb) {
    auto rd = [&](int ki, bf16x8* H8) {
      const int q = ks * KC + ki;
#pragma unroll
      for (int sp = 0; sp < SH; ++sp)
        H8[sp] = cat8(get16(&hl[rb][q][2 * sp][0], 0, true),
                      2 * sp + 1 < S ? get16(&hl[rb][q][2 * sp + 1 < S ? 2 * sp + 1 : 0][0], 0, true) : zero4);
    };
    bf16x8 Hn[SH];
    rd(0, Hn);
#pragma unroll
    for (int ki = 0; ki < KC; ++ki) {
      bf16x8 Hc[SH];
#pragma unroll
      for (int sp = 0; sp < SH; ++sp) Hc[sp] = Hn[sp];
      if (ki + 1 < KC) rd(ki + 1, Hn);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int sp = 0; sp < SH; ++sp)
#pragma unroll
        for (int mi = 0; mi < MCW; ++mi) acc[mi][ki] = mfma_bf(pa8[0][sp][mi], Hc[sp], acc[mi][ki]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (tile < a.ntiles) {
    f32x4 raw[S][MCW];
    produce(tile, 0);
    load_p_raw(tile, raw);
    transpose_p(raw, pa);
    pack_p(raw);
    if constexpr (XF || XB || XS) {
      if (a.xfold) xr = ld4(a.X + ((size_t)tile * XT + xsel) * 256 + lo);
    }
  }
  __syncthreads();
  // Round 6: every load of the prologue is waited for HERE, with a counter wait the compiler's own wait insertion sees.  The
  // raw-input fragment `xr` loaded above is first used at the head of the loop below, behind the loop's early loads of the next
  // tile's adjoint blocks -- and those sit in `st < SP` branches, so the number of loads issued after xr's is path-dependent
  // and the only statically safe wait in front of xr's use is "all but the last one": the listing had `s_waitcnt vmcnt(1)`
  // right behind the ten 16-byte loads that were meant to land during the MFMAs, i.e. every wave sat through an HBM round trip
  // per row tile in front of its matrix phase (since round 4).  Exact fp32 23.56 -> 22.68 ms per 2^18 points, the plain bf16 variant
  // 7.07 -> 6.22 ms.  (Tried with it and dropped: the adjoint fragments loaded already transposed -- four dword loads per block
  // instead of the LDS round trip, 23.43 ms -- and the next tile's z0 blocks requested before the MFMAs as well, 23.7-24.3 ms.)
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
  int buf = 0;
  // Phase swap (round 4, plain bf16 mode, hidden k-groups): the two waves of a SIMD meet at one barrier per row tile, and in
  // the loop below both run the same phases at the same time -- ring reads + bf16 MFMAs (the vector ALU idle), then the
  // preparation of the next tile: adjoint blocks through the patch, ring slot, tangent sums (the matrix pipe idle).  Within
  // an iteration the order of the two phases is free (the ring buffer being written is not the one being read, the prepared
  // adjoint operand goes to a staging set), so waves 4 .. 7 -- the second wave of every SIMD -- run them the other way
  // round: while one wave of a SIMD issues MFMAs the other one is in its VALU / LDS phase.  Their operand loads are
  // requested one iteration ahead (behind the preparation, in flight during the MFMAs and across the barrier).
  // (compiled for the packed-buffer variants: with fp32 blocks on both sides the second loop body spills)
  constexpr bool SWAPOK = BF && SPL == 1 && NBUF == 2 && !HASX && PKM != 0 && MODE == 1;
  if constexpr (SWAPOK) {
    if (wv >= NW / 2) {
      auto pack_to = [&](f32x4 (*raw_)[MCW], bf16x8 (*dst)[MCW]) {
        if constexpr (NPP > 2) {
#pragma unroll
          for (int st = 0; st < S; ++st)
#pragma unroll
            for (int mi = 0; mi < MCW; ++mi) put(pp[wv][st * MCW + mi], blk_val(raw_[st][mi], st < SP ? PMODE : 0, st), false);
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int sp = 0; sp < SH; ++sp)
#pragma unroll
            for (int mi = 0; mi < MCW; ++mi)
              dst[sp][mi] = cat8(get16(pp[wv][2 * sp * MCW + mi], 0, true),
                                 2 * sp + 1 < S ? get16(pp[wv][(2 * sp + 1 < S ? 2 * sp + 1 : 0) * MCW + mi], 0, true) : zero4);
          __builtin_amdgcn_wave_barrier();
          return;
        }
        int flip = 0;
        auto term = [&](f32x4 v) -> bf16x4 {
          float* patch = pp[wv][flip];
          flip ^= 1;
          put(patch, v, false);
          __builtin_amdgcn_wave_barrier();
          const bf16x4 t = get16(patch, 0, true);
          __builtin_amdgcn_wave_barrier();
          return t;
        };
#pragma unroll
        for (int sp = 0; sp < SH; ++sp)
#pragma unroll
          for (int mi = 0; mi < MCW; ++mi) {
            const bf16x4 t0 = term(blk_val(raw_[2 * sp][mi], 2 * sp < SP ? PMODE : 0, 2 * sp));
            bf16x4 t1 = zero4;
            if (2 * sp + 1 < S) t1 = term(blk_val(raw_[2 * sp + 1][mi], 2 * sp + 1 < SP ? PMODE : 0, 2 * sp + 1));
            dst[sp][mi] = cat8(t0, t1);
          }
      };
      f32x4 raw[S][MCW];
      f32x4 preq[S];
      float cqq[6];
      f32x4 xrn = f32x4{0.f, 0.f, 0.f, 0.f};
      if (tile < a.ntiles) {
        const int nx0 = tile + stride < a.ntiles ? tile + stride : tile;
        load_p_raw(nx0, raw);
        load_q(nx0, preq, cqq);
        if constexpr (XB) xrn = ld4(a.X + ((size_t)nx0 * XT + xsel) * 256 + lo);
      }
      for (; tile < a.ntiles; tile += stride) {
        const int next = tile + stride;
        const int nx = next < a.ntiles ? next : tile;
        // ---- preparation of tile nx (operands requested during the previous iteration)
        bf16x8 tmp8[SH][MCW];
        finish_q(preq, cqq, buf ^ 1);
        pack_to(raw, tmp8);
        const f32x4 xr_next = xrn;
        // ---- request the tile after it
        const int nn = nx + stride < a.ntiles ? nx + stride : nx;
        load_p_raw(nn, raw);
        load_q(nn, preq, cqq);
        if constexpr (XB) xrn = ld4(a.X + ((size_t)nn * XT + xsel) * 256 + lo);
        // ---- the MFMAs of the current tile
        xb_mma();
        ring_mma(buf);
#pragma unroll
        for (int sp = 0; sp < SH; ++sp)
#pragma unroll
          for (int mi = 0; mi < MCW; ++mi) pa8[0][sp][mi] = tmp8[sp][mi];
        if constexpr (XB) xr = xr_next;
        __syncthreads();
        buf ^= 1;
      }
    }
  }
  for (; tile < a.ntiles; tile += stride) {
    const int next = tile + stride;
    const int nx = next < a.ntiles ? next : tile;   // branch-free tail: the last iteration re-produces its own tile
    f32x4 raw[S][MCW];
    // (many-stream sets, S >= 8 = configs[4]'s (3,6): the S * MCW in-flight blocks do not fit next to the S * MCW transposed
    // ones -- 216 ... 432 bytes of scratch per lane -- so those load right in front of the transposes, round 4)
    constexpr bool EARLYP = S < 8 && !(XS && !STPDE_X3_EARLYP);
    if (EARLYP) load_p_raw(nx, raw);               // lands while the MFMAs below run
    f32x4 preq[S];
    float cqq[6];
    // bf16-pipe modes: likewise the next tile's activated-input source blocks (first-layer weight gradient 29.3 -> 27.0 ms
    // per step; with fp32 operands the early loads cost 1.5 %, so the fp32 kernels keep them behind the MFMAs)
    constexpr bool EARLYQ = BF && SPL == 1;     // (three-term split mode: 77.9 -> 93.2 ms with the early loads)
    if (EARLYQ && NBUF == 2 && STPDE_ABLATE_W != 3) load_q(nx, preq, cqq);
    // exact fp32, first hidden layer (round 6): only the z0 block -- the one load of the produce stage that comes from HBM --
    // before the MFMAs: 22.78 -> 22.45 ms per 2^18 points.  (All of load_q there: slower, 23.7 ms.  And without the
    // sched_barrier behind the MFMA loop the compiler hoisted the jets' first compares on z0 INTO the loop -- with a vmcnt(1)
    // behind the twelve early loads in front of them: 23.5 ms.)
    constexpr bool EARLYZ = !BF && MODE == 1 && !HASX && NBUF == 2;
    f32x4 z0e = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (EARLYZ) z0e = ld4(a.Q + ((size_t)nx * KT + kq0 + wv) * 256 + lo);
    f32x4 xrn = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (XB) {
      xrn = ld4(a.X + ((size_t)nx * XT + xsel) * 256 + lo);
      xb_mma();
    }
    if constexpr (XS) {
      xrn = ld4(a.X + ((size_t)nx * XT + xsel) * 256 + lo);
      xs_mma();
    }
    if constexpr (XF) {
      // branch-free: without xfold the pointer is this launch's own XR anyway and the results are simply not written
      xrn = ld4(a.X + ((size_t)nx * XT + xsel) * 256 + lo);
      const f32x4 xrr = x_rowmajor(xr);
#pragma unroll
      for (int mi = 0; mi < MCW; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) accx[mi] = mfma4(pa[0][mi][r], xrr[r], accx[mi]);
      if constexpr (S1 == 3) {
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
          for (int mi = 0; mi < MCW; ++mi)
            acct[d][mi] += (pa[1 + d][mi][0] + pa[1 + d][mi][1]) + (pa[1 + d][mi][2] + pa[1 + d][mi][3]);
      }
    }
    if constexpr (SPLP && !HASX && KC % 2 == 0) {
      // split mode, hidden k-tiles: two k-tiles at a time and the six partial products outermost, so that consecutive
      // MFMAs go to 2 x MCW different accumulators (six back-to-back MFMAs into ONE accumulator stall on each other)
#pragma unroll
      for (int ki = 0; ki < KC; ki += 2) {
        const int q = ks * KC + ki;
#pragma unroll
        for (int sp = 0; sp < SH; ++sp) {
          bf16x8 H8[2][3];
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int k = 0; k < 3; ++k)
              H8[kk][k] = STPDE_ABLATE_W == 4 ? pa8[k][sp][kk] :
                          cat8(get16(&hl[buf][q + kk][2 * sp][0], k),
                               2 * sp + 1 < S ? get16(&hl[buf][q + kk][2 * sp + 1 < S ? 2 * sp + 1 : 0][0], k) : zero4);
          constexpr int TP[6] = {1, 0, 2, 0, 1, 0}, TH[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
          for (int q6 = (STPDE_ABLATE_W == 1 ? 5 : 0); q6 < 6; ++q6)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
              for (int mi = 0; mi < MCW; ++mi)
                acc[mi][ki + kk] = mfma_bf(pa8[TP[q6]][sp][mi], H8[kk][TH[q6]], acc[mi][ki + kk]);
        }
      }
    } else if constexpr (BF && SPL == 1 && !HASX && STPDE_ABLATE_W == 0) {
      ring_mma(buf);
    } else if constexpr (!BF && !HASX && STPDE_WG_RPIPE) {
      // Exact fp32, hidden k-groups (round 5).  The loop below reads the S ring blocks of a k-slot and issues its MCW * 4 * S
      // MFMAs stream-innermost, and the compiler emitted `2 x ds_read_b128 ; s_waitcnt lgkmcnt(1) ; 4 MFMAs ; s_waitcnt
      // lgkmcnt(0) ; ...` -- 19 waits to zero per iteration, each with at most one 32-cycle MFMA in flight to cover an LDS round
      // trip (tools/micro/isa_waits.py; both waves of a SIMD are in this phase at the same time).  Here the block of
      // (k-slot, stream) n + 1 is requested before the MCW * 4 MFMAs of block n (256 cycles of matrix work per block), the
      // order of ring reads and MFMAs pinned; everything else may move across the pins.  (The sums of a dW element are
      // accumulated stream by stream instead of k-step by k-step: same terms, another fp32 rounding order.)
      constexpr int PIN = 0x2 | 0x4 | 0x10 | 0x20 | 0x40 | 0x200;      // VALU, SALU, VMEM and DS writes may cross; MFMA / DS reads not
      f32x4 Hc = get(&hl[buf][ks * KC][0][0]);
#pragma unroll
      for (int ki = 0; ki < KC; ++ki) {
        const int q = ks * KC + ki;
#pragma unroll
        for (int st = 0; st < S; ++st) {
          f32x4 Hn = Hc;
          if (st + 1 < S) Hn = get(&hl[buf][q][st + 1][0]);
          else if (ki + 1 < KC) Hn = get(&hl[buf][q + 1][0][0]);
          __builtin_amdgcn_sched_barrier(PIN);
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mi = 0; mi < MCW; ++mi) acc[mi][ki] = mfma4(pa[st][mi][r], Hc[r], acc[mi][ki]);
          __builtin_amdgcn_sched_barrier(PIN);
          Hc = Hn;
        }
      }
    } else {
#pragma unroll
    for (int ki = 0; ki < KC; ++ki) {
      const int q = ks * KC + ki;
      const int kq = kq0 + q;
      if (!HASX || kq < KT) {
        f32x4 H[S];
        if constexpr (!BF) {
#pragma unroll
          for (int st = 0; st < S; ++st) H[st] = get(&hl[buf][q][st][0]);
        }
        if constexpr (BF) {
#pragma unroll
          for (int sp = 0; sp < SH; ++sp) {
            bf16x4 t0[SPL], t1[SPL];
            bf16x8 H8[SPL];
#pragma unroll
            for (int k = 0; k < SPL; ++k) {
              t0[k] = get16(&hl[buf][q][2 * sp][0], k, true);
              if (2 * sp + 1 < S) t1[k] = get16(&hl[buf][q][2 * sp + 1 < S ? 2 * sp + 1 : 0][0], k, true);
            }
#pragma unroll
            for (int k = 0; k < SPL; ++k) H8[k] = cat8(t0[k], 2 * sp + 1 < S ? t1[k] : zero4);
#pragma unroll
            for (int mi = 0; mi < MCW; ++mi) mma16(acc[mi][ki], sp, mi, H8);
          }
        } else {
#pragma unroll
        for (int mi = 0; mi < MCW; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int st = 0; st < S; ++st) acc[mi][ki] = mfma4(pa[st][mi][r], H[st][r], acc[mi][ki]);
        }
      } else if (kq < KT + XT) {
        f32x4 H[SX];
        if constexpr (!BF) {
#pragma unroll
          for (int st = 0; st < SX; ++st) H[st] = get(&hl[buf][q][st][0]);
        }
        if constexpr (BF) {
#pragma unroll
          for (int sp = 0; sp < SXH; ++sp) {
            bf16x4 t0[SPL], t1[SPL];
            bf16x8 H8[SPL];
#pragma unroll
            for (int k = 0; k < SPL; ++k) {         // (raw-input k-tiles come from the column-major image as well: transpose read)
              t0[k] = get16(&hl[buf][q][2 * sp][0], k, true);
              if (2 * sp + 1 < SX) t1[k] = get16(&hl[buf][q][2 * sp + 1 < SX ? 2 * sp + 1 : 0][0], k, true);
            }
#pragma unroll
            for (int k = 0; k < SPL; ++k) H8[k] = cat8(t0[k], 2 * sp + 1 < SX ? t1[k] : zero4);
#pragma unroll
            for (int mi = 0; mi < MCW; ++mi) mma16(acc[mi][ki], sp, mi, H8);
          }
        } else {
#pragma unroll
        for (int mi = 0; mi < MCW; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int st = 0; st < SX; ++st) acc[mi][ki] = mfma4(pa[st][mi][r], H[st][r], acc[mi][ki]);
        }
      }
    }
    }
    if constexpr (XF || XB || XS) xr = xrn;
    if (NBUF == 2) {
      if (STPDE_ABLATE_W != 3) {
        if constexpr (EARLYZ) __builtin_amdgcn_sched_barrier(0);   // (the jets' first compares were hoisted into the MFMA loop: vmcnt(1) behind the early loads)
        if (!EARLYQ) load_q(nx, preq, cqq);
        if constexpr (EARLYZ) preq[0] = z0e;
        finish_q(preq, cqq, buf ^ 1);
      }
      if (!EARLYP) load_p_raw(nx, raw);
      if (STPDE_ABLATE_W != 2) {
        transpose_p(raw, pa);
        pack_p(raw);
      }
      if (STPDE_ABLATE_W != 5) __syncthreads();
      buf ^= 1;
    } else {
      __syncthreads();
      produce(nx, 0);
      if (!EARLYP) load_p_raw(nx, raw);
      transpose_p(raw, pa);
      pack_p(raw);
      __syncthreads();
    }
  }

  const int ldw = 16 * (KT + XT);
#pragma unroll
  for (int mi = 0; mi < MCW; ++mi) {
    const int mt = mt0 + mi;
    if (mt >= MT) continue;
#pragma unroll
    for (int ki = 0; ki < KC; ++ki) {
      const int kq = kq0 + ks * KC + ki;
      if (kq >= KT + XT) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        acc_add_f32(a.dW, (size_t)(16 * mt + 4 * g + r) * ldw + 16 * kq + c, acc[mi][ki][r], a.det);
    }
  }
  if constexpr (XB || XS) {
    if (a.xfold && xslot < XT) {
#pragma unroll
      for (int mi = 0; mi < MCW; ++mi) {
        const int mt = mt0 + mi;
        if (mt >= MT) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          acc_add_f32(a.dW, (size_t)(16 * mt + 4 * g + r) * ldw + 16 * (KT + xslot) + c, accx[mi][r], a.det);
      }
    }
  }
  if constexpr (XF) {
    if (a.xfold && xslot < XT) {
#pragma unroll
      for (int mi = 0; mi < MCW; ++mi) {
        const int mt = mt0 + mi;
        if (mt >= MT) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          acc_add_f32(a.dW, (size_t)(16 * mt + 4 * g + r) * ldw + 16 * (KT + xslot) + c, accx[mi][r], a.det);
        if constexpr (S1 == 3) {
          if (xslot == 0) {
            // column d of the raw-input block: sum over all rows of the tangent-stream adjoint d (lane (g, c) holds the rows
            // 4g..4g+3 of output feature c: fold the four lane groups, lanes of group 0 add)
#pragma unroll
            for (int d = 0; d < 3; ++d) {
              float v = acct[d][mi];
              v += __shfl_xor(v, 16, 64);
              v += __shfl_xor(v, 32, 64);
              if (g == 0) acc_add_f32(a.dW, (size_t)(16 * mt + c) * ldw + 16 * KT + d, v, a.det);
            }
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Narrow layers (MT <= 4 output tiles, KT = 2 / 4 / 8 hidden k-tiles: fc3..fc5 of the reference IM-NET): per-wave
// variant.  The cooperative kernel above keeps one 8-wave workgroup per CU (its ring takes most of the LDS) and pays
// one barrier + one exposed HBM latency per row tile, which is all there is when a tile carries only a few hundred
// MFMAs; here every wave owns whole row tiles (MCW output tiles x all KT + XT k-tiles), transposes its operands through
// a private 2.5 KB patch with wave-local barriers only, and 3-4 waves per SIMD overlap each other's latencies.  The
// four waves of a block are summed through LDS before the atomics.
// ------------------------------------------------------------------------------------------------------------
// PKW (bf16 mode): 1 = Q (the layer input's pre-activations) is a packed STASH, 2 = P is a packed ADJOINT buffer; the
// contraction over the rows then runs on v_mfma_f32_16x16x16_bf16 (BFM): the four fp32 k-steps of a transposed block pair
// are one bf16 MFMA on the rounded blocks (same (lane group, element) -> row map on both operands, so the sum is the same).
// The raw-input columns (exact skip operand) stay on the fp32 MFMA.
// (occupancy stated: with the packed jets of round 5 the allocator took 294 registers for the fc4 shape -- one wave per SIMD
// instead of two, 2.2 -> 3.4 ms in bf16 mode -- where 219 had done)
template <int S1, int S2, int ACT, int MCW, int KTT, int PKW = 0, bool BFM = false>
__global__ __launch_bounds__(256, (MCW * (KTT + XT) <= 8) ? 3 : ((MCW * (KTT + XT) <= 14) ? 2 : 1)) void k_wgrad_wave(WgradArgs a) {
  constexpr int S = 1 + S1 + S2, NK = KTT + XT;
  constexpr int TP = 24, TBLK = 16 * TP;
  __shared__ __attribute__((aligned(16))) float pp[4][2][TBLK];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int lo = lane * 4;
  const int MT = a.MT;
  const int mt0 = blockIdx.y * MCW;
  const int g = lane >> 4, c = lane & 15;

  f32x4 acc[MCW][NK];
#pragma unroll
  for (int mi = 0; mi < MCW; ++mi)
#pragma unroll
    for (int ki = 0; ki < NK; ++ki) acc[mi][ki] = f32x4{0.f, 0.f, 0.f, 0.f};

  float acct[S1 == 3 ? 3 : 1][MCW];
#pragma unroll
  for (int d = 0; d < (S1 == 3 ? 3 : 1); ++d)
#pragma unroll
    for (int mi = 0; mi < MCW; ++mi) acct[d][mi] = 0.f;
  int flip = 0;
  auto transpose = [&](f32x4 v) -> f32x4 {   // column-major image -> row-major image (alternating patches)
    float* patch = pp[wv][flip];
    flip ^= 1;
    lds_put_T<TP>(patch, lane, v);
    __builtin_amdgcn_wave_barrier();
    const f32x4 r = lds_get_R<TP>(patch, lane);
    __builtin_amdgcn_wave_barrier();
    return r;
  };

  // packed adjoint blocks (8 bytes per lane and block): the next tile's are requested a tile ahead (round 4)
  constexpr bool PFP = (PKW & 2) != 0;
  f32x4 rawn[PFP ? S : 1][PFP ? MCW : 1];
  auto load_p = [&](int t, f32x4 (*raw)[PFP ? MCW : 1]) {
    if constexpr (PFP) {
#pragma unroll
      for (int st = 0; st < S; ++st)
#pragma unroll
        for (int mi = 0; mi < MCW; ++mi) {
          const int mt = mt0 + mi < MT ? mt0 + mi : MT - 1;
          raw[st][mi] = ld_blk_raw(a.P, 2, t, S, MT, st, mt, lane);
        }
    }
  };
  if (PFP && (int)(blockIdx.x * 4 + wv) < a.ntiles) load_p(blockIdx.x * 4 + wv, rawn);
#pragma unroll 1
  for (int tile = blockIdx.x * 4 + wv; tile < a.ntiles; tile += gridDim.x * 4) {
    float cq[6];
    load_cq<S2>(a.cw, tile * 2 + (c >> 3), cq);
    f32x4 pa[S][MCW];
    if constexpr (PFP) {
#pragma unroll
      for (int st = 0; st < S; ++st)
#pragma unroll
        for (int mi = 0; mi < MCW; ++mi) pa[st][mi] = transpose(blk_val(rawn[st][mi], 2, st));
      const int tn = tile + (int)gridDim.x * 4 < a.ntiles ? tile + (int)gridDim.x * 4 : tile;
      load_p(tn, rawn);
    } else {
      f32x4 raw[S][MCW];
#pragma unroll
      for (int st = 0; st < S; ++st)
#pragma unroll
        for (int mi = 0; mi < MCW; ++mi) {
          const int mt = mt0 + mi < MT ? mt0 + mi : MT - 1;
          raw[st][mi] = ld_blk_raw(a.P, (PKW & 2) ? 2 : 0, tile, S, MT, st, mt, lane);
        }
#pragma unroll
      for (int st = 0; st < S; ++st)
#pragma unroll
        for (int mi = 0; mi < MCW; ++mi) pa[st][mi] = transpose(blk_val(raw[st][mi], (PKW & 2) ? 2 : 0, st));
    }
    // (round 4) the input blocks of k-tile ki + 1 and the raw-input fragments are requested before the jets / MFMAs of k-tile
    // ki: the loop used to load every block right in front of its use -- KTT + 1 exposed HBM round trips per row tile and
    // wave (tools/micro/isa_waits.py), which three or four waves per SIMD do not cover
    f32x4 qn[S], xrp[XT];
#pragma unroll
    for (int st = 0; st < S; ++st) qn[st] = ld_blk_raw(a.Q, (PKW & 1) ? 1 : 0, tile, S, KTT, st, 0, lane);
#pragma unroll
    for (int xt = 0; xt < XT; ++xt) xrp[xt] = ld4(a.X + ((size_t)tile * XT + xt) * 256 + lo);   // column-major: transposed at its use
#pragma unroll
    for (int ki = 0; ki < KTT; ++ki) {
      f32x4 pre[S], H[S];
#pragma unroll
      for (int st = 0; st < S; ++st) pre[st] = blk_val(qn[st], (PKW & 1) ? 1 : 0, st);
      if (ki + 1 < KTT) {
#pragma unroll
        for (int st = 0; st < S; ++st) qn[st] = ld_blk_raw(a.Q, (PKW & 1) ? 1 : 0, tile, S, KTT, st, ki + 1, lane);
      }
      act_jet_fwd<S1, S2, ACT>(a.cfg, pre, H, cq);
#pragma unroll
      for (int st = 0; st < S; ++st) {
        const f32x4 hr = transpose(H[st]);
        if constexpr (BFM) {
          const bf16x4 h16 = to_bf4(hr);
#pragma unroll
          for (int mi = 0; mi < MCW; ++mi) acc[mi][ki] = mfma_bf16k(to_bf4(pa[st][mi]), h16, acc[mi][ki]);
        } else {
#pragma unroll
          for (int mi = 0; mi < MCW; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mi][ki] = mfma4(pa[st][mi][r], hr[r], acc[mi][ki]);
        }
      }
    }
#pragma unroll
    for (int xt = 0; xt < XT; ++xt) {
      const f32x4 xr = transpose(xrp[xt]);
#pragma unroll
      for (int mi = 0; mi < MCW; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[mi][KTT + xt] = mfma4(pa[0][mi][r], xr[r], acc[mi][KTT + xt]);
    }
    if (S1 == 3) {
      // tangent stream d of a skip connection sees the unit vector e_d: its contribution to column d of the raw-input block
      // is the sum over rows of the adjoint -- per-lane partial sums here, folded once at the end (12 VALU instead of the 24
      // MFMAs against unit vectors of the first version)
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int mi = 0; mi < MCW; ++mi)
          acct[d][mi] += (pa[1 + d][mi][0] + pa[1 + d][mi][1]) + (pa[1 + d][mi][2] + pa[1 + d][mi][3]);
    }
  }

  // block reduction (4 waves) through the patches, then one set of atomics per block
  float* red = &pp[0][0][0];        // 4 x 256 floats needed, 8 x 384 available
  const int ldw = 16 * (KTT + XT);
#pragma unroll
  for (int mi = 0; mi < MCW; ++mi)
#pragma unroll
    for (int ki = 0; ki < NK; ++ki) {
      __syncthreads();
      st4(red + wv * 256 + lo, acc[mi][ki]);
      __syncthreads();
      const int mt = mt0 + mi;
      if (mt >= MT) continue;      // block-uniform
      const float sum = (red[lo + wv] + red[256 + lo + wv]) + (red[512 + lo + wv] + red[768 + lo + wv]);
      acc_add_f32(a.dW, (size_t)(16 * mt + 4 * g + wv) * ldw + 16 * ki + c, sum, a.det);
    }
  if constexpr (S1 == 3) {
    // tangent columns: lane (g, c) holds the partial row sums (rows 4g..4g+3 of every tile this wave walked) of output
    // feature c; fold the four lane groups, then one atomic per wave, feature and d
#pragma unroll
    for (int mi = 0; mi < MCW; ++mi) {
      const int mt = mt0 + mi;
      if (mt >= MT) continue;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        float v = acct[d][mi];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (g == 0) acc_add_f32(a.dW, (size_t)(16 * mt + c) * ldw + 16 * KTT + d, v, a.det);
      }
    }
  }
}

// fc3 of the reference width (4 output tiles, 8 hidden k-tiles + the raw-input tiles) with the four waves of a workgroup
// on ONE row tile (round 3).  The per-wave kernel above needs the accumulators of all 4 x 11 output blocks in one wave
// (368 - 480 registers: one wave per SIMD, every load latency in the open: 9.6 ms fp32 / 7.1 ms bf16 per step for 32 / 19 GB).
// Here wave w transposes the adjoint blocks of output tile w into a patch all four waves read, and owns hidden k-tiles
// 2w, 2w+1 and raw-input tile w (wave 3: the tangent column sums instead): 12 accumulator blocks and ~130 registers per
// wave, three workgroups per CU.  Every wave adds its own columns of dW with atomics at the end (no cross-wave reduction).
// NWV = 8 (round 3): the same scheme for the second hidden layer of the reference width (8 output tiles, 16 hidden k-tiles):
// 24 accumulator blocks per wave, one workgroup of 8 waves per CU; the adjoint blocks of a row tile are transposed ONCE (the
// ring kernel k_wgrad_coop does it once per k-group) and the waves meet at two barriers per row tile.
template <int S1, int S2, int ACT, int PKW, bool BFM, int NWV = 4>
__global__ __launch_bounds__(64 * NWV, NWV == 4 ? 3 : 2) void k_wgrad_quad(WgradArgs a) {
  constexpr int S = 1 + S1 + S2, KTT = 2 * NWV, MCW = NWV, KW = 2;
  constexpr int TP = 24, TBLK = 16 * TP;
  __shared__ __attribute__((aligned(16))) float pshare[S][MCW][TBLK];    // transposed adjoint blocks of the row tile
  __shared__ __attribute__((aligned(16))) float ppriv[NWV][2][TBLK];     // private patches (activated-input blocks)
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lo = lane * 4;
  const int g = lane >> 4, c = lane & 15;
  constexpr int PM = (PKW & 2) ? 2 : 0, QM = (PKW & 1) ? 1 : 0;

  f32x4 acc[MCW][KW + 1];
#pragma unroll
  for (int mi = 0; mi < MCW; ++mi)
#pragma unroll
    for (int k = 0; k <= KW; ++k) acc[mi][k] = f32x4{0.f, 0.f, 0.f, 0.f};
  float acct[3][MCW];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int mi = 0; mi < MCW; ++mi) acct[d][mi] = 0.f;
  int flip = 0;
  auto transpose = [&](f32x4 v) -> f32x4 {
    float* patch = ppriv[wv][flip];
    flip ^= 1;
    lds_put_T<TP>(patch, lane, v);
    __builtin_amdgcn_wave_barrier();
    const f32x4 r = lds_get_R<TP>(patch, lane);
    __builtin_amdgcn_wave_barrier();
    return r;
  };

  // the blocks of a row tile are requested one iteration ahead, into the registers the current tile has just released: the
  // adjoint blocks right after they went into the patch, the input blocks of k-tile k after its activation jets
  f32x4 praw[S], qraw[KW][S];
  f32x4 xr = f32x4{0.f, 0.f, 0.f, 0.f};
  auto load_p = [&](int t) {
#pragma unroll
    for (int st = 0; st < S; ++st) praw[st] = ld_blk_raw(a.P, PM, t, S, MCW, st, wv, lane);
    if (wv < XT) xr = ld4(a.X + ((size_t)t * XT + wv) * 256 + lo);        // column-major image: transposed at its use
  };
  auto load_qk = [&](int t, int k) {
#pragma unroll
    for (int st = 0; st < S; ++st) qraw[k][st] = ld_blk_raw(a.Q, QM, t, S, KTT, st, KW * wv + k, lane);
  };
  constexpr bool PF = NWV == 8;       // (the 4-wave variant runs three workgroups per CU on a 168-register budget: no room)
  if (PF && (int)blockIdx.x < a.ntiles) {
    load_p(blockIdx.x);
#pragma unroll
    for (int k = 0; k < KW; ++k) load_qk(blockIdx.x, k);
  }
#pragma unroll 1
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int tnext = tile + (int)gridDim.x < a.ntiles ? tile + (int)gridDim.x : tile;     // (last one: re-read, unused)
    float cq[6];
    load_cq<S2>(a.cw, tile * 2 + (c >> 3), cq);
    if (!PF) {
      load_p(tile);
#pragma unroll
      for (int k = 0; k < KW; ++k) load_qk(tile, k);
    }
#pragma unroll
    for (int st = 0; st < S; ++st) lds_put_T<TP>(&pshare[st][wv][0], lane, blk_val(praw[st], PM, st));
    const f32x4 xcur = xr;
    if (PF) load_p(tnext);
    __syncthreads();
    // (the transposed adjoint blocks are read from the shared patch where they are used: 4 instead of 20 blocks in registers)
#pragma unroll
    for (int k = 0; k < KW; ++k) {
      f32x4 pre[S], H[S];
#pragma unroll
      for (int st = 0; st < S; ++st) pre[st] = blk_val(qraw[k][st], QM, st);
      if (PF) load_qk(tnext, k);
      act_jet_fwd<S1, S2, ACT>(a.cfg, pre, H, cq);
#pragma unroll
      for (int st = 0; st < S; ++st) {
        const f32x4 hr = transpose(H[st]);
        if constexpr (!BFM && STPDE_QUAD_PIPE && NWV == 8) {
          // exact fp32 (round 4): the adjoint fragment of output tile mi + 1 is requested before the four MFMAs of tile mi (the
          // compiler read them in pairs and waited for each pair right in front of its MFMAs)
          f32x4 pc = lds_get_R<TP>(&pshare[st][0][0], lane);
#pragma unroll
          for (int mi = 0; mi < MCW; ++mi) {
            f32x4 pn = pc;
            if (mi + 1 < MCW) pn = lds_get_R<TP>(&pshare[st][mi + 1][0], lane);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mi][k] = mfma4(pc[r], hr[r], acc[mi][k]);
            pc = pn;
          }
          continue;
        }
        f32x4 pa[MCW];
#pragma unroll
        for (int mi = 0; mi < MCW; ++mi) pa[mi] = lds_get_R<TP>(&pshare[st][mi][0], lane);
        if constexpr (BFM) {
          const bf16x4 h16 = to_bf4(hr);
#pragma unroll
          for (int mi = 0; mi < MCW; ++mi) acc[mi][k] = mfma_bf16k(to_bf4(pa[mi]), h16, acc[mi][k]);
        } else {
#pragma unroll
          for (int mi = 0; mi < MCW; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mi][k] = mfma4(pa[mi][r], hr[r], acc[mi][k]);
        }
      }
    }
    if (wv < XT) {
      const f32x4 xrr = transpose(xcur);
#pragma unroll
      for (int mi = 0; mi < MCW; ++mi) {
        const f32x4 p0 = lds_get_R<TP>(&pshare[0][mi][0], lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[mi][KW] = mfma4(p0[r], xrr[r], acc[mi][KW]);
      }
    } else if (S1 == 3) {
      // tangent stream d of a skip connection sees the unit vector e_d: column d of the raw-input block gets the sum over the
      // rows of the adjoint (per-lane partial sums, folded once at the end)
      // (round 4: the output tiles are dealt to the NWV - XT waves without a raw-input tile -- every one of them used to sum
      // all of them, and only wave XT's sums were added at the end)
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int mi = 0; mi < MCW; ++mi) {
          if (mi % (NWV - XT) != wv - XT) continue;          // wave-uniform
          const f32x4 pd = lds_get_R<TP>(&pshare[1 + d][mi][0], lane);
          acct[d][mi] += (pd[0] + pd[1]) + (pd[2] + pd[3]);
        }
    }
    __syncthreads();        // every wave is done with pshare before the next tile's blocks go in
  }
  const int ldw = 16 * (KTT + XT);
#pragma unroll
  for (int mi = 0; mi < MCW; ++mi) {
#pragma unroll
    for (int k = 0; k <= KW; ++k) {
      if (k == KW && wv >= XT) continue;
      const int col = k < KW ? 16 * (KW * wv + k) : 16 * (KTT + wv);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc_add_f32(a.dW, (size_t)(16 * mi + 4 * g + r) * ldw + col + c, acc[mi][k][r], a.det);
    }
  }
  if (S1 == 3 && wv >= XT) {
    // lane (g, c): partial row sums (rows 4g..4g+3 of every tile) of output feature c; fold the four lane groups
#pragma unroll
    for (int mi = 0; mi < MCW; ++mi)
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        if (mi % (NWV - XT) != wv - XT) continue;
        float v = acct[d][mi];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (g == 0) acc_add_f32(a.dW, (size_t)(16 * mi + c) * ldw + 16 * KTT + d, v, a.det);
      }
  }
}

// Second hidden layer, plain bf16 mode with packed buffers (round 4): k_wgrad_quad<..., 8> above keeps fp32 blocks in LDS --
// every adjoint block goes bf16 -> fp32 -> four transposing ds_write_b32 -> ds_read_b128 -> v_cvt back to bf16 in EVERY wave
// and for EVERY k-tile that uses it: 970 VALU instructions per wave and row tile for 80 MFMAs (tools/micro/isa_waits.py), the
// kernel ran at its VALU issue time (13.3 ms per step for 6.2 ms of HBM traffic).  Same work split here (wave w: adjoint
// blocks of output tile w into the shared patch, hidden k-tiles 2w / 2w + 1, raw-input tile w), but
//   * the patches hold bf16 [row][feature] blocks: the adjoint blocks go in as the bytes they are stored as (one
//     ds_write_b64 per lane, no conversion), the activated blocks after ONE rounding; every operand fragment comes back
//     through the hardware transpose read ds_read_b64_tr_b16 -- no conversion at the consumers,
//   * an adjoint fragment is read once per stream and feeds the MFMAs of BOTH k-tiles of the wave,
//   * the tangent columns of the skip connection (sum over the rows of the tangent-stream adjoints) come out of wave 0's
//     raw-input accumulator tile: stream d against the pattern [feature == d] (three bf16 MFMAs per output tile) instead
//     of 72 VALU adds in five waves.
// NWV = 4: the same kernel for fc3 (4 output tiles, 8 hidden k-tiles; was k_wgrad_quad<..., 3, true, 4>: 613 VALU per wave and tile).
template <int S1, int S2, int ACT, int NWV = 8>
__global__ __launch_bounds__(64 * NWV, NWV == 4 ? 3 : 2) void k_wgrad_oct_bf(WgradArgs a) {
  constexpr int S = 1 + S1 + S2, KTT = 2 * NWV, MCW = NWV, KW = 2;
  static_assert(S1 == 3, "tangent streams expected");
  __shared__ __attribute__((aligned(16))) __bf16 pshare[S][MCW][256];        // adjoint blocks of the row tile
  __shared__ __attribute__((aligned(16))) __bf16 ppriv[NWV][KW][S][256];     // activated-input blocks of each wave's k-tiles
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lo = lane * 4;
  const int g = lane >> 4, c = lane & 15;
  const int wofs = (lane & 15) * 16 + 4 * (lane >> 4);                                   // this lane's 4 features of its row
  const int rofs = (4 * (lane >> 4) + ((lane & 15) >> 2)) * 16 + 4 * (lane & 3);         // transpose read (common.h)
  const bf16x4 zero4 = to_bf4(f32x4{0.f, 0.f, 0.f, 0.f}), one4 = to_bf4(f32x4{1.f, 1.f, 1.f, 1.f});
  f32x4 acc[MCW][KW + 1];
#pragma unroll
  for (int mi = 0; mi < MCW; ++mi)
#pragma unroll
    for (int k = 0; k <= KW; ++k) acc[mi][k] = f32x4{0.f, 0.f, 0.f, 0.f};
  float2 praw[S];
  f32x4 qraw[KW][S];
  f32x4 xr = f32x4{0.f, 0.f, 0.f, 0.f};
  auto load_p = [&](int t) {
    const char* pb = reinterpret_cast<const char*>(a.P) + (size_t)t * (S * MCW * 512) + (size_t)wv * 512 + lane * 8;
#pragma unroll
    for (int st = 0; st < S; ++st) praw[st] = *reinterpret_cast<const float2*>(pb + (size_t)st * MCW * 512);
    if (wv < XT) xr = ld4(a.X + ((size_t)t * XT + wv) * 256 + lo);
  };
  auto load_qk = [&](int t, int k) {
#pragma unroll
    for (int st = 0; st < S; ++st) qraw[k][st] = ld_blk_raw(a.Q, 1, t, S, KTT, st, KW * wv + k, lane);
  };
  if ((int)blockIdx.x < a.ntiles) {
    load_p(blockIdx.x);
#pragma unroll
    for (int k = 0; k < KW; ++k) load_qk(blockIdx.x, k);
  }
#pragma unroll 1
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int tnext = tile + (int)gridDim.x < a.ntiles ? tile + (int)gridDim.x : tile;
    float cq[6];
    load_cq<S2>(a.cw, tile * 2 + (c >> 3), cq);
#pragma unroll
    for (int st = 0; st < S; ++st) *reinterpret_cast<float2*>(&pshare[st][wv][wofs]) = praw[st];
    // raw-input tile of this wave (exact fp32 skip operand): column-major block -> row-major fragment through the wave's
    // own patch area, before the activated blocks of this iteration go into it
    f32x4 xcur = xr;
    if (wv < XT) {
      float* patch = reinterpret_cast<float*>(&ppriv[wv][0][0][0]);          // 5 KB per wave: room for a padded fp32 block
      lds_put_T<20>(patch, lane, xr);
      __builtin_amdgcn_wave_barrier();
      xcur = lds_get_R<20>(patch, lane);
      __builtin_amdgcn_wave_barrier();
    }
    load_p(tnext);
#pragma unroll
    for (int k = 0; k < KW; ++k) {
      f32x4 pre[S], H[S];
#pragma unroll
      for (int st = 0; st < S; ++st) pre[st] = blk_val(qraw[k][st], 1, st);
      load_qk(tnext, k);
      act_jet_fwd<S1, S2, ACT>(a.cfg, pre, H, cq);
#pragma unroll
      for (int st = 0; st < S; ++st) *reinterpret_cast<bf16x4*>(&ppriv[wv][k][st][wofs]) = to_bf4(H[st]);
    }
    __syncthreads();
    bf16x4 h16[KW][S];
#pragma unroll
    for (int k = 0; k < KW; ++k)
#pragma unroll
      for (int st = 0; st < S; ++st) h16[k][st] = lds_read_tr16(&ppriv[wv][k][st][rofs]);
#pragma unroll
    for (int st = 0; st < S; ++st) {
#pragma unroll
      for (int mi = 0; mi < MCW; ++mi) {
        const bf16x4 pa = lds_read_tr16(&pshare[st][mi][rofs]);
#pragma unroll
        for (int k = 0; k < KW; ++k) acc[mi][k] = mfma_bf16k(pa, h16[k][st], acc[mi][k]);
        if (st == 0) {
          if (wv < XT) {                                   // raw-input tile wv: exact fp32 skip operand (wave-uniform branch)
            const f32x4 p0 = bf4_to_f32(pa);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mi][KW] = mfma4(p0[r], xcur[r], acc[mi][KW]);
          }
        } else if (st <= 3) {
          if (wv == 0) acc[mi][KW] = mfma_bf16k(pa, c == st - 1 ? one4 : zero4, acc[mi][KW]);   // tangent column st - 1
        }
      }
    }
    __syncthreads();        // every wave is done with pshare before the next tile's blocks go in
  }
  const int ldw = 16 * (KTT + XT);
#pragma unroll
  for (int mi = 0; mi < MCW; ++mi) {
#pragma unroll
    for (int k = 0; k <= KW; ++k) {
      if (k == KW && wv >= XT) continue;
      const int col = k < KW ? 16 * (KW * wv + k) : 16 * (KTT + wv);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc_add_f32(a.dW, (size_t)(16 * mi + 4 * g + r) * ldw + col + c, acc[mi][k][r], a.det);
    }
  }
}

template <int S1, int S2, int ACT, int MCW, int KTT>
static int launch_wgrad_wave(const WgradArgs& a, hipStream_t stream) {
  const int gy = (a.MT + MCW - 1) / MCW;
  int gx = 768 / gy;                       // ~3 waves per SIMD over the chip
  if (gx > (a.ntiles + 3) / 4) gx = (a.ntiles + 3) / 4;
  if (gx < 1) gx = 1;
  if (a.pk) {
    // bf16 mode, fc3 / fc4 / fc5 of the reference width: packed operands (WgradArgs.pk: 1 = Q, 4 = P), bf16 contraction
    if constexpr (S1 == 3 && ((KTT == 8 && (MCW == 4 || MCW == 2)) || (KTT == 4 && MCW == 2) || (KTT == 2 && MCW == 1))) {
      if (a.pk == 5) {
        STPDE_LAUNCH((k_wgrad_wave<S1, S2, ACT, MCW, KTT, 3, true>), dim3(gx, gy), dim3(256), 0, stream, a);
        return stpde_check_launch("k_wgrad_wave");
      }
      if (a.pk == 1) {
        STPDE_LAUNCH((k_wgrad_wave<S1, S2, ACT, MCW, KTT, 1, true>), dim3(gx, gy), dim3(256), 0, stream, a);
        return stpde_check_launch("k_wgrad_wave");
      }
    }
    stpde_set_error("jet_wgrad: packed operands (%d) not compiled for this narrow-layer shape", a.pk);
    return STPDE_E_UNSUPPORTED;
  }
  STPDE_LAUNCH((k_wgrad_wave<S1, S2, ACT, MCW, KTT>), dim3(gx, gy), dim3(256), 0, stream, a);
  return stpde_check_launch("k_wgrad_wave");
}

// returns -1 when the shape is not served by the per-wave kernel
template <int S1, int S2, int ACT>
static int try_wgrad_wave(const WgradArgs& a, hipStream_t stream) {
  if (a.MT > 4 || a.SP != 1 + S1 + S2 || a.bf16) return -1;
  if constexpr (S1 + S2 > 5) {
    return -1;                             // S = 10: the accumulators + operands would not fit the register file
  } else {
    if (a.MT == 1) {
      if (a.KT == 2) return launch_wgrad_wave<S1, S2, ACT, 1, 2>(a, stream);
      if (a.KT == 4) return launch_wgrad_wave<S1, S2, ACT, 1, 4>(a, stream);
    } else {
      if (a.KT == 2) return launch_wgrad_wave<S1, S2, ACT, 2, 2>(a, stream);
      if (a.KT == 4) return launch_wgrad_wave<S1, S2, ACT, 2, 4>(a, stream);
      // fc3 of the reference net (4 output tiles, 8 hidden k-tiles): all four output tiles in ONE pass halves the stash reads
      // of this HBM-bound kernel (the two-pass grid reads abar3 / pre2 twice: 11.4 -> 9.6 ms per step)
      if constexpr (S1 == 3) {
        if (a.KT == 8 && a.MT == 4 && XT == 3 && (a.pk == 0 || a.pk == 5)) {
          int gx = 768;                        // three workgroups per CU, persistent
          if (gx > a.ntiles) gx = a.ntiles;
          if (a.pk) {                          // bf16 mode, packed buffers: bf16 operand blocks in LDS
            STPDE_LAUNCH((k_wgrad_oct_bf<S1, S2, ACT, 4>), dim3(gx), dim3(256), 0, stream, a);
            return stpde_check_launch("k_wgrad_oct_bf");
          }
          STPDE_LAUNCH((k_wgrad_quad<S1, S2, ACT, 0, false>), dim3(gx), dim3(256), 0, stream, a);
          return stpde_check_launch("k_wgrad_quad");
        }
      }
      if (a.KT == 8 && a.MT == 4) return launch_wgrad_wave<S1, S2, ACT, 4, 8>(a, stream);
      if (a.KT == 8) return launch_wgrad_wave<S1, S2, ACT, 2, 8>(a, stream);
    }
    return -1;
  }
}

template <int S1, int S2, int MODE, int ACT, int KC>
static int launch_wgrad_kc(const WgradArgs& a0, hipStream_t stream) {
  WgradArgs a = a0;
  if (a.KT == 0 && a.pk == 4 && a.bf16 == 1) a.pk = 5;      // raw-input layer: no Q operand, the mask bit of Q is free
  a.gy = (a.MT + 2 * KC - 1) / (2 * KC);    // groups of 2*KC output tiles
  const int ngr = (a.KT + XT + 7) / 8;      // k-groups (ring = 8 k-tiles)
  const int nhid = a.KT / 8;                // groups made of hidden tiles only
  // fp32, every hidden k-tile in a full group and enough (k-group, k-slot) pairs for the XT raw-input tiles: those are folded
  // into the hidden-group launch and the second launch is dropped
  const int kslots = 8 / KC;                 // k-slots per workgroup (NW / NM)
  // "fp32x3" (a.bf16 == 3) is a contract on ACCURACY (fp32), not on the pipe: stream sets / widths its split kernels are not
  // compiled for take the exact-fp32 kernels (round 4; they used to be refused)
  constexpr bool SPLIT_OK = KC >= 4 && S1 + S2 <= 4;
  const int bfm = (a.bf16 == 3 && !SPLIT_OK) ? 0 : a.bf16;
  const bool xfold = (!bfm || (bfm == 1 && KC >= 4) || (bfm == 3 && KC >= 4 && STPDE_X3_XFOLD)) && a.KT > 0 && a.KT % 8 == 0 && nhid * kslots >= XT && a.X;
  for (int part = 0; part < 2; ++part) {
    a.kz0 = part == 0 ? 0 : nhid;
    a.gz = part == 0 ? nhid : ngr - nhid;
    a.xfold = (xfold && part == 0) ? 1 : 0;
    if (a.gz <= 0 || (xfold && part == 1)) continue;
    int gx = 512 / (a.gy * a.gz);           // ~2 rounds of one 8-wave workgroup per CU
    if (gx > a.ntiles) gx = a.ntiles;
    gx = (gx + 7) / 8 * 8;
    if (gx < 8) gx = 8;
    a.gx = gx;
    constexpr bool HAS_BF = KC >= 4;          // bf16 variant compiled for the wide layers only (MT >= 8)
    const dim3 grid(gx * a.gy * a.gz);
    if (HAS_BF && bfm == 3) {
      if constexpr (SPLIT_OK) {
        if (part == 0)
          STPDE_LAUNCH((k_wgrad_coop<S1, S2, MODE, ACT, KC, false, true, 3>), grid, dim3(512), 0, stream, a);
        else
          STPDE_LAUNCH((k_wgrad_coop<S1, S2, MODE, ACT, KC, true, true, 3>), grid, dim3(512), 0, stream, a);
      }
    } else if (HAS_BF && bfm) {
      if constexpr (HAS_BF) {
        constexpr int PKA = MODE == 1 ? 4 : 5;      // first hidden layer: packed abar; layer 2: packed input stash AND packed abar
        if (a.pk != 0 && a.pk != PKA) {
          stpde_set_error("packed layer buffers: combination %d not compiled for this weight-gradient kind", a.pk);
          return STPDE_E_UNSUPPORTED;
        }
        if (part == 0 && a.pk)
          STPDE_LAUNCH((k_wgrad_coop<S1, S2, MODE, ACT, KC, false, true, 1, PKA>), grid, dim3(512), 0, stream, a);
        else if (part == 0)
          STPDE_LAUNCH((k_wgrad_coop<S1, S2, MODE, ACT, KC, false, true>), grid, dim3(512), 0, stream, a);
        else if (a.pk)
          STPDE_LAUNCH((k_wgrad_coop<S1, S2, MODE, ACT, KC, true, true, 1, PKA>), grid, dim3(512), 0, stream, a);
        else
          STPDE_LAUNCH((k_wgrad_coop<S1, S2, MODE, ACT, KC, true, true>), grid, dim3(512), 0, stream, a);
      }
    } else if (a.pk) {
      // bf16 mode: the value-stream weight gradient of the raw-input layer reads its adjoint as bf16 blocks (fp32 MFMA)
      if constexpr (S1 == 0 && S2 == 0 && MODE == 0 && KC == 8) {
        if (a.pk == 4 && part != 0) {
          STPDE_LAUNCH((k_wgrad_coop<S1, S2, MODE, ACT, KC, true, false, 1, 4>), grid, dim3(512), 0, stream, a);
          int rc = stpde_check_launch("k_wgrad_coop");
          if (rc) return rc;
          continue;
        }
      }
      stpde_set_error("packed layer buffers: combination %d not compiled for the fp32 weight-gradient kernels", a.pk);
      return STPDE_E_UNSUPPORTED;
    } else if (part == 0) {
      STPDE_LAUNCH((k_wgrad_coop<S1, S2, MODE, ACT, KC, false>), grid, dim3(512), 0, stream, a);
    } else {
      STPDE_LAUNCH((k_wgrad_coop<S1, S2, MODE, ACT, KC, true>), grid, dim3(512), 0, stream, a);
    }
    int rc = stpde_check_launch("k_wgrad_coop");
    if (rc) return rc;
  }
  return STPDE_OK;
}

template <int S1, int S2, int MODE, int ACT>
static int launch_wgrad_act(const WgradArgs& a, hipStream_t stream) {
  if constexpr (MODE == 0) {
    const int rc = try_wgrad_wave<S1, S2, ACT>(a, stream);
    if (rc >= 0) return rc;
    // second hidden layer of the reference width: eight waves on one row tile (k_wgrad_quad<..., 8>); exact-fp32 operands, or
    // bf16 operands with packed buffers (the three-term split mode keeps the ring kernel)
    if constexpr (S1 == 3 && S1 + S2 <= 5) {
      // "fp32x3" (a.bf16 == 3, round 5): the exact-fp32 eight-wave kernel as well -- 25.7 ms per 2^20 points against 26.0 + 8.0
      // for the split ring kernel and its separate raw-input launch; the mode is a contract on accuracy, not on the pipe
      if (a.KT == 16 && a.MT == 8 && XT == 3 && a.SP == 1 + S1 + S2 && a.X &&
          ((a.bf16 == 0 && a.pk == 0) || (a.bf16 == 1 && a.pk == 5) || (a.bf16 == 3 && a.pk == 0))) {
        int gx = 256;                          // one workgroup per CU, persistent
        if (gx > a.ntiles) gx = a.ntiles;
        if (a.pk) {                            // bf16 mode, packed buffers: bf16 operand blocks in LDS
          STPDE_LAUNCH((k_wgrad_oct_bf<S1, S2, ACT>), dim3(gx), dim3(512), 0, stream, a);
          return stpde_check_launch("k_wgrad_oct_bf");
        }
        STPDE_LAUNCH((k_wgrad_quad<S1, S2, ACT, 0, false, 8>), dim3(gx), dim3(512), 0, stream, a);
        return stpde_check_launch("k_wgrad_quad");
      }
    }
  }
  if (a.MT >= 16) return launch_wgrad_kc<S1, S2, MODE, ACT, 8>(a, stream);
  if constexpr (MODE == 1) {
    return launch_wgrad_kc<S1, S2, MODE, ACT, 4>(a, stream);   // the first hidden layer is never narrower than 8 tiles
  } else {
    if (a.MT >= 8) return launch_wgrad_kc<S1, S2, MODE, ACT, 4>(a, stream);
    if (a.MT >= 4) return launch_wgrad_kc<S1, S2, MODE, ACT, 2>(a, stream);
    return launch_wgrad_kc<S1, S2, MODE, ACT, 1>(a, stream);
  }
}

template <int S1, int S2>
static int launch_mode(const WgradArgs& a, int mode, hipStream_t stream) {
  if (mode == 0) {
    // the produce stage (one activation jet per ring slot) is the VALU-heavy part of the narrow layers:
    // compile-time activation (branch-free) there as well
    switch (a.cfg.act) {
      case STPDE_ACT_TANH: return launch_wgrad_act<S1, S2, 0, STPDE_ACT_TANH>(a, stream);
      case STPDE_ACT_RELU: return launch_wgrad_act<S1, S2, 0, STPDE_ACT_RELU>(a, stream);
      case STPDE_ACT_SOFTPLUS: return launch_wgrad_act<S1, S2, 0, STPDE_ACT_SOFTPLUS>(a, stream);
      case STPDE_ACT_ELU: return launch_wgrad_act<S1, S2, 0, STPDE_ACT_ELU>(a, stream);
      case STPDE_ACT_LEAKYRELU: return launch_wgrad_act<S1, S2, 0, STPDE_ACT_LEAKYRELU>(a, stream);
      default: return launch_wgrad_act<S1, S2, 0, STPDE_ACT_SWISH>(a, stream);
    }
  }
  switch (a.cfg.act) {
    case STPDE_ACT_TANH: return launch_wgrad_act<S1, S2, 1, STPDE_ACT_TANH>(a, stream);
    case STPDE_ACT_RELU: return launch_wgrad_act<S1, S2, 1, STPDE_ACT_RELU>(a, stream);
    case STPDE_ACT_SOFTPLUS: return launch_wgrad_act<S1, S2, 1, STPDE_ACT_SOFTPLUS>(a, stream);
    case STPDE_ACT_ELU: return launch_wgrad_act<S1, S2, 1, STPDE_ACT_ELU>(a, stream);
    case STPDE_ACT_LEAKYRELU: return launch_wgrad_act<S1, S2, 1, STPDE_ACT_LEAKYRELU>(a, stream);
    default: return launch_wgrad_act<S1, S2, 1, STPDE_ACT_SWISH>(a, stream);
  }
}

#define STPDE_DEFINE_WGRAD_TU(S1, S2) \
  int stpde_wgrad_launch_##S1##_##S2(const WgradArgs& a, int mode, hipStream_t stream) { return launch_mode<S1, S2>(a, mode, stream); }

int stpde_wgrad_launch_0_0(const WgradArgs& a, int mode, hipStream_t stream);
int stpde_wgrad_launch_3_0(const WgradArgs& a, int mode, hipStream_t stream);
int stpde_wgrad_launch_3_1(const WgradArgs& a, int mode, hipStream_t stream);
int stpde_wgrad_launch_3_2(const WgradArgs& a, int mode, hipStream_t stream);
int stpde_wgrad_launch_3_4(const WgradArgs& a, int mode, hipStream_t stream);
int stpde_wgrad_launch_3_6(const WgradArgs& a, int mode, hipStream_t stream);
