// Fused forward of the three narrowest IM-NET layers (fc3 -> fc4 -> fc5 of src/implicit_net.py:48-54) on all derivative
// streams: one wave owns a row tile, runs fc3 from the stashed layer-2 pre-activations, and then feeds the accumulator
// tiles of each layer -- whose C/D register image is the B-operand image -- through the activation jet straight into the
// next layer's MFMAs: the inter-layer data never leaves the registers.  The pre-activations of all three layers are still
// written (the backward needs them), but each is written once and never read back in the forward pass: 78 KB per tile
// instead of 114 KB for the three separate kernels, two launches fewer.
// Compiled for the reference widths nf = 32 (NFT = 2: 128 -> 64 -> 32 -> out) and nf = 16 (NFT = 1) and for the training
// stream sets; every other shape keeps the per-layer kernels (jet_layer_impl.h).
#include "jet_layer_impl.h"

// nothing may be scheduled across this point: keeps the operand requests of the NEXT stage above the MFMAs of the current one
// (the machine scheduler of a fully unrolled one-wave-per-tile kernel otherwise sinks every load down to its first use)
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

struct TailArgs {
  const float* in2;              // [tile][S][KT3][256] pre-activations of the layer in front (layer 2)
  const float* X;                // [tile][XT][256]
  const float *Wh[3], *Ws[3], *tanc[3];
  float* out[3];                 // [tile][S][MT_l][256]
  const float* cw;
  int ntiles;
  const void* Wh16[3];           // bf16 A-operand packs of the three layers (k_tail_fwd_bf) or null
  int packed;                    // k_tail_fwd_bf: 1 = in2 is a packed layer buffer, 2 = out[0] / out[1] are written packed
  stpde_jet_cfg cfg;
};

template <int S1, int S2, int ACT, int NFT>
__global__ __launch_bounds__(256) void k_tail_fwd(TailArgs a) {
  constexpr int S = 1 + S1 + S2;
  constexpr int KT3 = 4 * NFT, MT3 = 2 * NFT, MT4 = NFT, MT5 = 1;
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= a.ntiles) return;
  const int lo = lane * 4;
  float cq[6];
  constexpr bool VT = S1 == 0 && S2 > 0;     // value-tile mode (forward-only queries): the S "streams" are the value
  constexpr int NX = VT ? S : 1;             // streams of S consecutive row tiles, each with its own raw-input tile
  if (!VT) load_cq<S2>(a.cw, tile * 2 + ((lane & 15) >> 3), cq);
  f32x4 xb[NX][XT];
#pragma unroll
  for (int sv = 0; sv < NX; ++sv)
#pragma unroll
    for (int xt = 0; xt < XT; ++xt) xb[sv][xt] = ld4x(a.X + (((size_t)tile * NX + sv) * XT + xt) * 256 + lo, xt);

  // epilogue of one layer: skip GEMM with the raw input (bias through its ones column), tangent constants, store
  auto finish = [&](int l, int MT, int mt, f32x4* acc) {
#pragma unroll
    for (int xt = 0; xt < XT; ++xt) {
      const f32x4 w = ld4x(a.Ws[l] + ((size_t)xt * MT + mt) * 256 + lo, xt);
#pragma unroll
      for (int sv = 0; sv < NX; ++sv)
#pragma unroll
        for (int r = 0; r < x_live(xt); ++r) acc[sv] = mfma4(w[r], xb[sv][xt][r], acc[sv]);
    }
    if (S1 == 3) {
#pragma unroll
      for (int d = 0; d < 3; ++d) acc[1 + d] += ld4(a.tanc[l] + ((size_t)d * MT + mt) * 256 + lo);
    }
#pragma unroll
    for (int st = 0; st < S; ++st) st4(a.out[l] + (((size_t)tile * S + st) * MT + mt) * 256 + lo, acc[st]);
  };

  // ---- fc3: B operand from the stash (one k-tile prefetched ahead), all MT3 output tiles in registers
  f32x4 acc3[MT3][S];
#pragma unroll
  for (int mi = 0; mi < MT3; ++mi)
#pragma unroll
    for (int st = 0; st < S; ++st) acc3[mi][st] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    f32x4 raw[S], w[MT3];
#pragma unroll
    for (int st = 0; st < S; ++st) raw[st] = ld4(a.in2 + (((size_t)tile * S + st) * KT3) * 256 + lo);
#pragma unroll
    for (int mi = 0; mi < MT3; ++mi) w[mi] = ld4(a.Wh[0] + ((size_t)mi) * 256 + lo);
#pragma unroll
    for (int kt = 0; kt < KT3; ++kt) {
      const int kn = kt + 1 < KT3 ? kt + 1 : kt;
      f32x4 rawn[S], wn[MT3], B[S];
#pragma unroll
      for (int st = 0; st < S; ++st) rawn[st] = ld4(a.in2 + (((size_t)tile * S + st) * KT3 + kn) * 256 + lo);
#pragma unroll
      for (int mi = 0; mi < MT3; ++mi) wn[mi] = ld4(a.Wh[0] + ((size_t)kn * MT3 + mi) * 256 + lo);
      SCHED_FENCE();
      act_jet_fwd<S1, S2, ACT>(a.cfg, raw, B, cq);
#pragma unroll
      for (int mi = 0; mi < MT3; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int st = 0; st < S; ++st) acc3[mi][st] = mfma4(w[mi][r], B[st][r], acc3[mi][st]);
#pragma unroll
      for (int st = 0; st < S; ++st) raw[st] = rawn[st];
#pragma unroll
      for (int mi = 0; mi < MT3; ++mi) w[mi] = wn[mi];
    }
  }
  // weights of fc4 and fc5 (L2-resident, MT3 * MT4 + MT4 fragments): requested before fc3's epilogue (round 4; they were
  // loaded right in front of their MFMAs, one exposed L2 round trip per k-tile with two waves per SIMD to cover it)
  constexpr bool PFW = !VT && S <= 5;          // (S = 6: the prefetched fragments would cost the second wave per SIMD)
  f32x4 w4[PFW ? MT3 : 1][PFW ? MT4 : 1], w5[PFW ? MT4 : 1];
  if constexpr (PFW) {
#pragma unroll
    for (int kt = 0; kt < MT3; ++kt)
#pragma unroll
      for (int mi = 0; mi < MT4; ++mi) w4[kt][mi] = ld4(a.Wh[1] + ((size_t)kt * MT4 + mi) * 256 + lo);
#pragma unroll
    for (int kt = 0; kt < MT4; ++kt) w5[kt] = ld4(a.Wh[2] + ((size_t)kt * MT5) * 256 + lo);
    SCHED_FENCE();
  }
#pragma unroll
  for (int mi = 0; mi < MT3; ++mi) finish(0, MT3, mi, acc3[mi]);

  // ---- fc4: the k-tiles are the accumulator tiles of fc3 (C/D image == B image)
  f32x4 acc4[MT4][S];
#pragma unroll
  for (int mi = 0; mi < MT4; ++mi)
#pragma unroll
    for (int st = 0; st < S; ++st) acc4[mi][st] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kt = 0; kt < MT3; ++kt) {
    f32x4 B[S];
    act_jet_fwd<S1, S2, ACT>(a.cfg, acc3[kt], B, cq);
#pragma unroll
    for (int mi = 0; mi < MT4; ++mi) {
      const f32x4 w = PFW ? w4[PFW ? kt : 0][PFW ? mi : 0] : ld4(a.Wh[1] + ((size_t)kt * MT4 + mi) * 256 + lo);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int st = 0; st < S; ++st) acc4[mi][st] = mfma4(w[r], B[st][r], acc4[mi][st]);
    }
  }
#pragma unroll
  for (int mi = 0; mi < MT4; ++mi) finish(1, MT4, mi, acc4[mi]);

  // ---- fc5 (no activation after it, src/implicit_net.py:54)
  f32x4 acc5[S];
#pragma unroll
  for (int st = 0; st < S; ++st) acc5[st] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kt = 0; kt < MT4; ++kt) {
    f32x4 B[S];
    act_jet_fwd<S1, S2, ACT>(a.cfg, acc4[kt], B, cq);
    const f32x4 w = PFW ? w5[PFW ? kt : 0] : ld4(a.Wh[2] + ((size_t)kt * MT5) * 256 + lo);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int st = 0; st < S; ++st) acc5[st] = mfma4(w[r], B[st][r], acc5[st]);
  }
  finish(2, MT5, 0, acc5);
}

// ------------------------------------------------------------------------------------------------------------
// bf16 mode (BASELINE configs[3]), reference width: the same chain on the bf16 matrix pipe, every layer buffer in the PACKED
// form (common.h: value stream fp32, derivative streams bf16).
// These three layers are narrow (128 -> 64 -> 32 -> out): a rounding error of an operand is averaged over few terms, and the
// oracle puts plain bf16 operands here at 2e-2 ... 5e-2 on the second-derivative streams (against 1e-3 for the two wide
// layers together).  So both operands are split into TWO bf16 terms (x = hi + lo, 16 mantissa bits) and a product is three
// v_mfma_f32_16x16x32_bf16 (lo x hi, hi x lo, hi x hi; fp32 accumulation): 315 MFMAs of 16 cycles per row tile instead of 840
// of 33, at 2^-16 relative accuracy -- the mode's error stays that of the wide layers.  Two accumulator tiles of a layer,
// activated and split, are lane by lane the B operands of one K = 32 step of the next layer -- the k-order of the bf16 weight
// packs (lig_jet.ImNetPlan.pack_bf16: fp32 blocks (2q, mt) and (2q + 1, mt) lane by lane; [2][KT/2][MT][64] = hi, lo).  The
// skip GEMM with the raw input stays on the exact fp32 MFMA (63 per tile), as in the wide bf16 layers.  The packed stores
// round the derivative streams for the BACKWARD pass only: the chain itself runs on the unrounded accumulators.
// LDSW (round 5): one wave per row tile with two waves per SIMD is a chain of ~15 exposed L2 round trips per tile -- the
// skip-weight / tangent-constant fragments of the seven epilogues and the bf16 weight fragments are loaded right where they
// are used (a tile took 64k cycles per wave for ~20k cycles of issued work; 3.9 TB/s of HBM traffic).  All of these operands
// are the same for every row tile: 84 KB.  The LDSW kernel is persistent (one workgroup of 8 waves per CU, every wave walks
// over row tiles), copies them into LDS once and reads them from there (~100 cycles instead of an L2 round trip).
template <int S1, int S2, int ACT, bool LDSW>
__global__ __launch_bounds__(LDSW ? 512 : 256, LDSW ? 1 : 2) void k_tail_fwd_bf(TailArgs a) {
  constexpr int S = 1 + S1 + S2;
  constexpr int KT3 = 8, MT3 = 4, MT4 = 2;
  // LDS image of the tile-independent operands (floats): Ws[3], tanc[3] (XT or 3 x MT_l x 256 each), then the bf16 packs
  constexpr int NWS = XT * (MT3 + MT4 + 1) * 256, NTC = 3 * (MT3 + MT4 + 1) * 256;
  constexpr int NW3 = 2 * (KT3 / 2) * MT3 * 64 * 4, NW4 = 2 * (MT3 / 2) * MT4 * 64 * 4, NW5 = 2 * 64 * 4;     // in floats (16 B per lane)
  __shared__ __attribute__((aligned(16))) float sm[LDSW ? NWS + NTC + NW3 + NW4 + NW5 : 4];
  constexpr int OWS[3] = {0, XT * MT3 * 256, XT * (MT3 + MT4) * 256};
  constexpr int OTC[3] = {NWS, NWS + 3 * MT3 * 256, NWS + 3 * (MT3 + MT4) * 256};
  constexpr int OW3 = NWS + NTC, OW4 = OW3 + NW3, OW5 = OW4 + NW4;
  const int lane = threadIdx.x & 63;
  if constexpr (LDSW) {
    auto copy = [&](int off, const void* src, int nfloat) {
      for (int i = threadIdx.x * 4; i < nfloat; i += 512 * 4) st4(sm + off + i, ld4(reinterpret_cast<const float*>(src) + i));
    };
    const int mts[3] = {MT3, MT4, 1};
#pragma unroll
    for (int l = 0; l < 3; ++l) {
      copy(OWS[l], a.Ws[l], XT * mts[l] * 256);
      copy(OTC[l], a.tanc[l], 3 * mts[l] * 256);
    }
    copy(OW3, a.Wh16[0], NW3);
    copy(OW4, a.Wh16[1], NW4);
    copy(OW5, a.Wh16[2], NW5);
    __syncthreads();
  }
  // one row tile (lo = lane * 4, handed in: the persistent loop launders it per iteration, see below)
  auto run_tile = [&](const int tile, const int lo) {
  float cq[6];
  load_cq<S2>(a.cw, tile * 2 + ((lane & 15) >> 3), cq);
  f32x4 xb[XT];
#pragma unroll
  for (int xt = 0; xt < XT; ++xt) xb[xt] = ld4x(a.X + ((size_t)tile * XT + xt) * 256 + lo, xt);

  // epilogue of one layer: skip GEMM (fp32), tangent constants, store (packed: derivative streams rounded in place)
  auto finish = [&](int l, int MT, int mt, f32x4* acc, bool pk) {
#pragma unroll
    for (int xt = 0; xt < XT; ++xt) {
      const f32x4 w = LDSW ? ld4(sm + OWS[l] + (xt * MT + mt) * 256 + lo) : ld4x(a.Ws[l] + ((size_t)xt * MT + mt) * 256 + lo, xt);
#pragma unroll
      for (int r = 0; r < x_live(xt); ++r) acc[0] = mfma4(w[r], xb[xt][r], acc[0]);
    }
#pragma unroll
    for (int d = 0; d < 3; ++d)
      acc[1 + d] += LDSW ? ld4(sm + OTC[l] + (d * MT + mt) * 256 + lo) : ld4(a.tanc[l] + ((size_t)d * MT + mt) * 256 + lo);
    if (!pk) {
#pragma unroll
      for (int st = 0; st < S; ++st) st4(a.out[l] + (((size_t)tile * S + st) * MT + mt) * 256 + lo, acc[st]);
      return;
    }
    char* t = reinterpret_cast<char*>(a.out[l]) + (size_t)tile * packed_tile_bytes(S, MT);
    st4(reinterpret_cast<float*>(t + (size_t)mt * 1024) + lo, acc[0]);
#pragma unroll
    for (int st = 1; st < S; ++st)
      *reinterpret_cast<bf16x4*>(t + (size_t)MT * 1024 + ((size_t)(st - 1) * MT + mt) * 512 + lane * 8) = to_bf4(acc[st]);
  };
  // two activated blocks -> the B operands (hi, lo) of one K = 32 step
  auto pair_b = [&](const f32x4* p0, const f32x4* p1, bf16x8 (*B8)[2]) {
    f32x4 h0[S], h1[S];
    act_jet_fwd<S1, S2, ACT>(a.cfg, p0, h0, cq);
    act_jet_fwd<S1, S2, ACT>(a.cfg, p1, h1, cq);
#pragma unroll
    for (int st = 0; st < S; ++st) {
      const bf16x4 a0 = to_bf4(h0[st]), a1 = to_bf4(h1[st]);
      B8[st][0] = cat8(a0, a1);
      B8[st][1] = cat8(to_bf4(h0[st] - bf4_to_f32(a0)), to_bf4(h1[st] - bf4_to_f32(a1)));   // residuals: exact in fp32
    }
  };
  // one K = 32 step of the two-term product, smallest partial products first
  auto mma3 = [&](const bf16x8* w8, const bf16x8* b8, f32x4 c) -> f32x4 {
    c = mfma_bf(w8[1], b8[0], c);
    c = mfma_bf(w8[0], b8[1], c);
    return mfma_bf(w8[0], b8[0], c);
  };

  // ---- fc3: B operands from the packed stash of fc2's rows (the next pair of k-tiles in flight)
  f32x4 acc3[MT3][S];
#pragma unroll
  for (int mi = 0; mi < MT3; ++mi)
#pragma unroll
    for (int st = 0; st < S; ++st) acc3[mi][st] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    const bf16x8* W3 = (LDSW ? reinterpret_cast<const bf16x8*>(sm + OW3) : reinterpret_cast<const bf16x8*>(a.Wh16[0])) + lane;      // [2][KT3 / 2][MT3][64]
    f32x4 raw[2][S];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int st = 0; st < S; ++st) raw[e][st] = ld_blk_raw(a.in2, true, tile, S, KT3, st, e, lane);
#pragma unroll
    for (int q = 0; q < KT3 / 2; ++q) {
      const int qn = q + 1 < KT3 / 2 ? q + 1 : q;
      f32x4 rawn[2][S];
      bf16x8 w8[MT3][2], B8[S][2];
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int st = 0; st < S; ++st) rawn[e][st] = ld_blk_raw(a.in2, true, tile, S, KT3, st, 2 * qn + e, lane);
#pragma unroll
      for (int mi = 0; mi < MT3; ++mi)
#pragma unroll
        for (int t = 0; t < 2; ++t) w8[mi][t] = W3[(((size_t)t * (KT3 / 2) + q) * MT3 + mi) * 64];
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int st = 0; st < S; ++st) raw[e][st] = blk_val(raw[e][st], true, st);
      pair_b(raw[0], raw[1], B8);
#pragma unroll
      for (int mi = 0; mi < MT3; ++mi)
#pragma unroll
        for (int st = 0; st < S; ++st) acc3[mi][st] = mma3(w8[mi], B8[st], acc3[mi][st]);
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int st = 0; st < S; ++st) raw[e][st] = rawn[e][st];
    }
  }
#pragma unroll
  for (int mi = 0; mi < MT3; ++mi) finish(0, MT3, mi, acc3[mi], true);

  // ---- fc4
  f32x4 acc4[MT4][S];
#pragma unroll
  for (int mi = 0; mi < MT4; ++mi)
#pragma unroll
    for (int st = 0; st < S; ++st) acc4[mi][st] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    const bf16x8* W4 = (LDSW ? reinterpret_cast<const bf16x8*>(sm + OW4) : reinterpret_cast<const bf16x8*>(a.Wh16[1])) + lane;      // [2][MT3 / 2][MT4][64]
#pragma unroll
    for (int q = 0; q < MT3 / 2; ++q) {
      bf16x8 B8[S][2];
      pair_b(acc3[2 * q], acc3[2 * q + 1], B8);
#pragma unroll
      for (int mi = 0; mi < MT4; ++mi) {
        bf16x8 w8[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) w8[t] = W4[(((size_t)t * (MT3 / 2) + q) * MT4 + mi) * 64];
#pragma unroll
        for (int st = 0; st < S; ++st) acc4[mi][st] = mma3(w8, B8[st], acc4[mi][st]);
      }
    }
  }
#pragma unroll
  for (int mi = 0; mi < MT4; ++mi) finish(1, MT4, mi, acc4[mi], true);

  // ---- fc5 (no activation after it; its rows go to the reduction as fp32 blocks)
  f32x4 acc5[S];
#pragma unroll
  for (int st = 0; st < S; ++st) acc5[st] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    bf16x8 B8[S][2];
    pair_b(acc4[0], acc4[1], B8);
    const bf16x8* W5 = LDSW ? reinterpret_cast<const bf16x8*>(sm + OW5) : reinterpret_cast<const bf16x8*>(a.Wh16[2]);
    const bf16x8 w8[2] = {W5[lane], W5[64 + lane]};                                 // [2][1][1][64]
#pragma unroll
    for (int st = 0; st < S; ++st) acc5[st] = mma3(w8, B8[st], acc5[st]);
  }
  finish(2, 1, 0, acc5, false);
  };  // run_tile
  if constexpr (!LDSW) {
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile < a.ntiles) run_tile(tile, lane * 4);
  } else {
#pragma unroll 1
    for (int tile = blockIdx.x * 8 + (threadIdx.x >> 6); tile < a.ntiles; tile += gridDim.x * 8) {
      // an opaque zero per iteration: without it every tile-independent address of the body (dozens of 64-bit pointers)
      // is hoisted out of the loop and the body spills ~1 KB
      int zero;
      asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
      run_tile(tile, lane * 4 + zero);
    }
  }
}

template <int S1, int S2, int ACT>
static int launch_tail_nft(const TailArgs& a, int nft, hipStream_t stream) {
  const dim3 grid((a.ntiles + 3) / 4);
  if (a.packed || a.Wh16[0]) {
    if constexpr (S1 == 3) {       // bf16 mode: the training stream sets, reference width, every layer buffer packed
      if (nft == 2 && a.packed == 3 && a.Wh16[0] && a.Wh16[1] && a.Wh16[2]) {
        // persistent, tile-independent operands in LDS
        int gx = 256;
        if (gx > (a.ntiles + 7) / 8) gx = (a.ntiles + 7) / 8;
        STPDE_LAUNCH((k_tail_fwd_bf<S1, S2, ACT, true>), dim3(gx), dim3(512), 0, stream, a);
        return stpde_check_launch("k_tail_fwd_bf");
      }
    }
    stpde_set_error("jet_tail_fwd: the bf16-operand kernel is compiled for nf = 32, S1 = 3, packed = 3 with all three bf16 packs");
    return STPDE_E_UNSUPPORTED;
  }
  if (nft == 2)
    STPDE_LAUNCH((k_tail_fwd<S1, S2, ACT, 2>), grid, dim3(256), 0, stream, a);
  else
    STPDE_LAUNCH((k_tail_fwd<S1, S2, ACT, 1>), grid, dim3(256), 0, stream, a);
  return stpde_check_launch("k_tail_fwd");
}

template <int S1, int S2>
static int launch_tail_act(const TailArgs& a, int nft, hipStream_t stream) {
  switch (a.cfg.act) {
    case STPDE_ACT_TANH: return launch_tail_nft<S1, S2, STPDE_ACT_TANH>(a, nft, stream);
    case STPDE_ACT_RELU: return launch_tail_nft<S1, S2, STPDE_ACT_RELU>(a, nft, stream);
    case STPDE_ACT_SOFTPLUS: return launch_tail_nft<S1, S2, STPDE_ACT_SOFTPLUS>(a, nft, stream);
    case STPDE_ACT_ELU: return launch_tail_nft<S1, S2, STPDE_ACT_ELU>(a, nft, stream);
    case STPDE_ACT_LEAKYRELU: return launch_tail_nft<S1, S2, STPDE_ACT_LEAKYRELU>(a, nft, stream);
    default: return launch_tail_nft<S1, S2, STPDE_ACT_SWISH>(a, nft, stream);
  }
}

static int tail_fwd(const stpde_jet_cfg* cfg, int ntiles, int nf16, const float* in_pre2, const float* X,
                    const float* const* Wh_pack, const float* const* Ws_pack, const float* const* tanc,
                    float* const* out_pre, const float* cw, int packed, const void* const* Wh16_pack, void* stream);

extern "C" int stpde_jet_tail_fwd(const stpde_jet_cfg* cfg, int ntiles, int nf16, const float* in_pre2, const float* X,
                                  const float* const* Wh_pack, const float* const* Ws_pack, const float* const* tanc,
                                  float* const* out_pre, const float* cw, void* stream) {
  return tail_fwd(cfg, ntiles, nf16, in_pre2, X, Wh_pack, Ws_pack, tanc, out_pre, cw, 0, nullptr, stream);
}
extern "C" int stpde_jet_tail_fwd_p(const stpde_jet_cfg* cfg, int ntiles, int nf16, const float* in_pre2, const float* X,
                                    const float* const* Wh_pack, const float* const* Ws_pack, const float* const* tanc,
                                    float* const* out_pre, const float* cw, int packed, const void* const* Wh16_pack,
                                    void* stream) {
  return tail_fwd(cfg, ntiles, nf16, in_pre2, X, Wh_pack, Ws_pack, tanc, out_pre, cw, packed, Wh16_pack, stream);
}

static int tail_fwd(const stpde_jet_cfg* cfg, int ntiles, int nf16, const float* in_pre2, const float* X,
                    const float* const* Wh_pack, const float* const* Ws_pack, const float* const* tanc,
                    float* const* out_pre, const float* cw, int packed, const void* const* Wh16_pack, void* stream) {
  if (!cfg || ntiles <= 0 || (nf16 != 1 && nf16 != 2) || !in_pre2 || !X || !Wh_pack || !Ws_pack || !tanc || !out_pre ||
      cfg->act < 0 || cfg->act > 5) {
    stpde_set_error("jet_tail_fwd: bad argument (nf must be 16 or 32)");
    return STPDE_E_BADARG;
  }
  TailArgs a{};
  a.in2 = in_pre2;
  a.X = X;
  for (int l = 0; l < 3; ++l) {
    if (!Wh_pack[l] || !Ws_pack[l] || !out_pre[l] || (cfg->S1 && !tanc[l])) {
      stpde_set_error("jet_tail_fwd: null pointer for layer %d", 3 + l);
      return STPDE_E_BADARG;
    }
    a.Wh[l] = Wh_pack[l];
    a.Ws[l] = Ws_pack[l];
    a.tanc[l] = tanc ? tanc[l] : nullptr;
    a.out[l] = out_pre[l];
  }
  a.cw = cw;
  a.ntiles = ntiles;
  a.packed = packed;
  for (int l = 0; l < 3; ++l) a.Wh16[l] = Wh16_pack ? Wh16_pack[l] : nullptr;
  a.cfg = *cfg;
  const int S1 = cfg->S1, S2 = cfg->S2;
  if (S1 == 0 && S2 == 0) return launch_tail_act<0, 0>(a, nf16, (hipStream_t)stream);
  if (S1 == 3 && S2 == 0) return launch_tail_act<3, 0>(a, nf16, (hipStream_t)stream);
  if (S1 == 3 && S2 == 1 && cfg->combo && cw) return launch_tail_act<3, 1>(a, nf16, (hipStream_t)stream);
  if (S1 == 3 && S2 == 2) return launch_tail_act<3, 2>(a, nf16, (hipStream_t)stream);
  if (S1 == 3 && S2 == 4) return launch_tail_act<3, 4>(a, nf16, (hipStream_t)stream);
  if (S1 == 0 && S2 == 3) return launch_tail_act<0, 3>(a, nf16, (hipStream_t)stream);   // value tiles: ntiles = row tiles / 4
  stpde_set_error("jet_tail_fwd: stream configuration S1=%d S2=%d not compiled", S1, S2);
  return STPDE_E_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------------------
// Fused input-gradient chain of the same three layers (what loss.backward(), experiments/rb2d/train.py:77, does through
// fc5 -> fc4 -> fc3): abar_5 (from the corner-reduction adjoint) -> W5^T -> activation-jet adjoint with the stashed
// pre-activations of fc4's output -> abar_4 -> W4^T -> ... -> abar_2, written over the layer-2 stash.  The adjoints of
// layers 4 and 3 are the B operand of the next GEMM straight from the registers (C/D image == B image); they are still
// written once (weight gradients and the latent-gradient GEMM read them), but not read back by this chain.
// ------------------------------------------------------------------------------------------------------------
struct TailBwdArgs {
  const float* abar5;            // [tile][S][1][256]
  const float* WhT[3];           // layers 3, 4, 5: [MT_l][KT_l][256] transposed packs
  const float* pre[3];           // stashed pre-activations of the outputs of layers 2, 3, 4
  float* out[3];                 // their adjoints (may alias pre[l]: each block is read before it is written)
  const float* cw;
  float* pbar;
  int ntiles;
  const void* WhT16[3];          // bf16 packs of the transposed weights of layers 3, 4 (k_tail_bwd_bf; [2] unused) or null
  int packed;                    // k_tail_bwd_bf: 3 = every pre[] / out[] is a packed layer buffer
  stpde_jet_cfg cfg;
};

// Round 4: every stash block and weight fragment is REQUESTED one stage ahead of its use.  The first version loaded the five
// stash blocks of an output tile right in front of their activation-jet adjoint and every weight fragment right in front of
// its MFMAs (compiler listing: ~30 `global_load ; s_waitcnt vmcnt(0..1)` pairs per row tile), with two waves per SIMD (252
// registers) to cover HBM / L2 round trips of 1-2 us each: 0.45 of the fp32 pipe.
template <int S1, int S2, int ACT, int NFT>
__global__ __launch_bounds__(256, (1 + S1 + S2 >= 8 && NFT == 2) ? 1 : 2) void k_tail_bwd(TailBwdArgs a) {   // (S = 8: 552-700 B of scratch at two waves per SIMD)
  constexpr int S = 1 + S1 + S2;
  constexpr int T2 = 4 * NFT, T3 = 2 * NFT, T4 = NFT;     // feature tiles of the outputs of layers 2, 3, 4
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= a.ntiles) return;
  const int lo = lane * 4;
  float cq[6];
  load_cq<S2>(a.cw, tile * 2 + ((lane & 15) >> 3), cq);
  float pacc = 0.f;
  const bool swish = (ACT == STPDE_ACT_SWISH) && a.pbar;

  auto ldpre = [&](int l, int MT, int mt, f32x4* pre) {
#pragma unroll
    for (int st = 0; st < S; ++st) pre[st] = ld4(a.pre[l] + (((size_t)tile * S + st) * MT + mt) * 256 + lo);
  };
  // hbar (accumulator tile) + stored pre-activations (already in registers) -> adjoint, written to out[l] and returned in acc
  auto adjoint = [&](int l, int MT, int mt, const f32x4* pre, f32x4* acc) {
    float* buf = a.out[l];
    f32x4 ab[S];
    act_jet_adj<S1, S2, ACT>(a.cfg, pre, acc, ab, cq);
    if (swish) pacc += swish_beta_adj<S1, S2>(a.cfg, pre, acc, cq);
#pragma unroll
    for (int st = 0; st < S; ++st) {
      st4(buf + (((size_t)tile * S + st) * MT + mt) * 256 + lo, ab[st]);
      acc[st] = ab[st];
    }
  };

  // ---- head: everything the fc5 stage needs, and the fc4 stage's weights + its first stash blocks
  f32x4 b5[S], p4[T4][S], w5[T4], w4[T4][T3];
  constexpr int P3 = T3 < 2 ? T3 : 2;          // stash blocks of fc3's output rows: a ring of two output tiles
  f32x4 p3[P3][S];
#pragma unroll
  for (int st = 0; st < S; ++st) b5[st] = ld4(a.abar5 + ((size_t)tile * S + st) * 256 + lo);
#pragma unroll
  for (int mi = 0; mi < T4; ++mi) {
    w5[mi] = ld4(a.WhT[2] + ((size_t)mi) * 256 + lo);                  // [MT5 = 1][KT = T4]: block (0, mi)
    ldpre(2, T4, mi, p4[mi]);
  }
#pragma unroll
  for (int kt = 0; kt < T4; ++kt)
#pragma unroll
    for (int mi = 0; mi < T3; ++mi) w4[kt][mi] = ld4(a.WhT[1] + ((size_t)kt * T3 + mi) * 256 + lo);   // [MT4 = T4][KT = T3]
#pragma unroll
  for (int e = 0; e < P3; ++e) ldpre(1, T3, e, p3[e]);
  SCHED_FENCE();      // (the machine scheduler otherwise sinks every request down to its first use)

  // ---- through fc5: abar_5 [1 tile] -> hbar_4 [T4 tiles] -> abar_4
  f32x4 a4[T4][S];
#pragma unroll
  for (int mi = 0; mi < T4; ++mi) {
#pragma unroll
    for (int st = 0; st < S; ++st) a4[mi][st] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int st = 0; st < S; ++st) a4[mi][st] = mfma4(w5[mi][r], b5[st][r], a4[mi][st]);
    adjoint(2, T4, mi, p4[mi], a4[mi]);
  }
  // first weight slab (k-tile 0, output tiles 0 / 1) of the fc3 stage: requested before the fc4 stage's MFMAs
  f32x4 w3c[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) w3c[e] = ld4(a.WhT[0] + ((size_t)e) * 256 + lo);                          // [MT3 = T3][KT = T2]
  SCHED_FENCE();
  // ---- through fc4: abar_4 -> hbar_3 [T3 tiles] -> abar_3
  f32x4 a3[T3][S];
#pragma unroll
  for (int mi = 0; mi < T3; ++mi) {
#pragma unroll
    for (int st = 0; st < S; ++st) a3[mi][st] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < T4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int st = 0; st < S; ++st) a3[mi][st] = mfma4(w4[kt][mi][r], a4[kt][st][r], a3[mi][st]);
    adjoint(1, T3, mi, p3[mi % P3], a3[mi]);
    if (mi + P3 < T3) ldpre(1, T3, mi + P3, p3[mi % P3]);           // ring: the block two tiles ahead
    SCHED_FENCE();
  }
  // ---- through fc3: abar_3 -> hbar_2 [T2 tiles] -> abar_2, two output tiles at a time; the stash blocks of a pair are
  // requested at the head of the pair's MFMAs (2 x T3 x 4 x S of them), the weight slabs one k-tile ahead
#pragma unroll
  for (int m0 = 0; m0 < T2; m0 += 2) {
    f32x4 a2[2][S], p2[2][S];
#pragma unroll
    for (int e = 0; e < 2; ++e) ldpre(0, T2, m0 + e, p2[e]);
    SCHED_FENCE();
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int st = 0; st < S; ++st) a2[e][st] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < T3; ++kt) {
      // next slab: k-tile kt + 1 of this pair, or k-tile 0 of the next pair (the last one re-reads its own: harmless)
      const int ktn = kt + 1 < T3 ? kt + 1 : 0;
      const int mn = kt + 1 < T3 ? m0 : (m0 + 2 < T2 ? m0 + 2 : m0);
      f32x4 w3n[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) w3n[e] = ld4(a.WhT[0] + ((size_t)ktn * T2 + mn + e) * 256 + lo);
      SCHED_FENCE();
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int st = 0; st < S; ++st) a2[e][st] = mfma4(w3c[e][r], a3[kt][st][r], a2[e][st]);
#pragma unroll
      for (int e = 0; e < 2; ++e) w3c[e] = w3n[e];
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) adjoint(0, T2, m0 + e, p2[e], a2[e]);
  }
  if (swish) {
    const float v = wave_sum(pacc);
    if (lane == 0) atomicAdd(a.pbar + (blockIdx.x % STPDE_PBAR_SLOTS), v);
  }
}

// ------------------------------------------------------------------------------------------------------------
// bf16 mode: the input-gradient chain on the bf16 pipe; pre[] are packed STASH buffers, out[] packed ADJOINT buffers (common.h:
// every stream bf16 -- out[l] must not alias pre[l]).  The adjoint blocks of a layer, as they are stored, are pairwise the B
// operand of the transposed product in front (one bf16 term: a rounding error of an adjoint is averaged over the rows);
// the narrow product through fc5 (K = one 16-feature tile) stays on the fp32 MFMA.  All stash blocks of layers 3 / 4 are
// requested at the head of the kernel, those of layer 2 one pair of tiles ahead of their use.
template <int S1, int S2, int ACT>
__global__ __launch_bounds__(256, 2) void k_tail_bwd_bf(TailBwdArgs a) {
  constexpr int S = 1 + S1 + S2;
  constexpr int T2 = 8, T3 = 4, T4 = 2;
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= a.ntiles) return;
  const int lo = lane * 4;
  float cq[6];
  load_cq<S2>(a.cw, tile * 2 + ((lane & 15) >> 3), cq);
  float pacc = 0.f;
  const bool swish = (ACT == STPDE_ACT_SWISH) && a.pbar;

  f32x4 b5[S], p4[T4][S], p3[T3][S], p2[2][S];
#pragma unroll
  for (int st = 0; st < S; ++st) b5[st] = ld4(a.abar5 + ((size_t)tile * S + st) * 256 + lo);
#pragma unroll
  for (int mi = 0; mi < T4; ++mi)
#pragma unroll
    for (int st = 0; st < S; ++st) p4[mi][st] = ld_blk_raw(a.pre[2], true, tile, S, T4, st, mi, lane);
#pragma unroll
  for (int mi = 0; mi < T3; ++mi)
#pragma unroll
    for (int st = 0; st < S; ++st) p3[mi][st] = ld_blk_raw(a.pre[1], true, tile, S, T3, st, mi, lane);
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int st = 0; st < S; ++st) p2[e][st] = ld_blk_raw(a.pre[0], true, tile, S, T2, st, e, lane);

  // hbar (accumulator tile) + raw stash blocks -> adjoint: stored packed, returned as the bf16 blocks of the next B operand
  auto adjoint = [&](int l, int MT, int mt, const f32x4* praw, const f32x4* hbar, bf16x4* ob) {
    f32x4 pre[S], ab[S];
#pragma unroll
    for (int st = 0; st < S; ++st) pre[st] = blk_val(praw[st], true, st);
    act_jet_adj<S1, S2, ACT>(a.cfg, pre, hbar, ab, cq);
    if (swish) pacc += swish_beta_adj<S1, S2>(a.cfg, pre, hbar, cq);
    char* t = reinterpret_cast<char*>(a.out[l]) + (size_t)tile * blk_tile_bytes(2, S, MT);     // packed ADJOINT buffer
#pragma unroll
    for (int st = 0; st < S; ++st) {
      ob[st] = to_bf4(ab[st]);
      *reinterpret_cast<bf16x4*>(t + ((size_t)st * MT + mt) * 512 + lane * 8) = ob[st];
    }
  };

  // ---- through fc5 (fp32: K = 16): abar_5 -> hbar_4 [T4 tiles] -> abar_4
  bf16x4 o4[T4][S];
#pragma unroll
  for (int mi = 0; mi < T4; ++mi) {
    const f32x4 w = ld4(a.WhT[2] + ((size_t)mi) * 256 + lo);
    f32x4 hb[S];
#pragma unroll
    for (int st = 0; st < S; ++st) hb[st] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int st = 0; st < S; ++st) hb[st] = mfma4(w[r], b5[st][r], hb[st]);
    adjoint(2, T4, mi, p4[mi], hb, o4[mi]);
  }
  // ---- through fc4: abar_4 (one K = 32 step) -> hbar_3 [T3 tiles] -> abar_3
  bf16x4 o3[T3][S];
  {
    const bf16x8* W4T = reinterpret_cast<const bf16x8*>(a.WhT16[1]) + lane;      // [T4 / 2][T3][64]
    bf16x8 B8[S];
#pragma unroll
    for (int st = 0; st < S; ++st) B8[st] = cat8(o4[0][st], o4[1][st]);
#pragma unroll
    for (int mi = 0; mi < T3; ++mi) {
      const bf16x8 w8 = W4T[(size_t)mi * 64];
      f32x4 hb[S];
#pragma unroll
      for (int st = 0; st < S; ++st) hb[st] = mfma_bf(w8, B8[st], f32x4{0.f, 0.f, 0.f, 0.f});
      adjoint(1, T3, mi, p3[mi], hb, o3[mi]);
    }
  }
  // ---- through fc3: abar_3 (two K = 32 steps) -> hbar_2 [T2 tiles] -> abar_2, two output tiles at a time
  {
    const bf16x8* W3T = reinterpret_cast<const bf16x8*>(a.WhT16[0]) + lane;      // [T3 / 2][T2][64]
    bf16x8 B8[2][S];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int st = 0; st < S; ++st) B8[q][st] = cat8(o3[2 * q][st], o3[2 * q + 1][st]);
#pragma unroll
    for (int m0 = 0; m0 < T2; m0 += 2) {
      const int mn = m0 + 2 < T2 ? m0 + 2 : m0;
      f32x4 p2n[2][S];
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int st = 0; st < S; ++st) p2n[e][st] = ld_blk_raw(a.pre[0], true, tile, S, T2, st, mn + e, lane);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        f32x4 hb[S];
        bf16x4 o2[S];
#pragma unroll
        for (int st = 0; st < S; ++st) hb[st] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const bf16x8 w8 = W3T[((size_t)q * T2 + m0 + e) * 64];
#pragma unroll
          for (int st = 0; st < S; ++st) hb[st] = mfma_bf(w8, B8[q][st], hb[st]);
        }
        adjoint(0, T2, m0 + e, p2[e], hb, o2);
      }
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int st = 0; st < S; ++st) p2[e][st] = p2n[e][st];
    }
  }
  if (swish) {
    const float v = wave_sum(pacc);
    if (lane == 0) atomicAdd(a.pbar + (blockIdx.x % STPDE_PBAR_SLOTS), v);
  }
}

template <int S1, int S2, int ACT>
static int launch_tailb_nft(const TailBwdArgs& a, int nft, hipStream_t stream) {
  const dim3 grid((a.ntiles + 3) / 4);
  if (a.packed || a.WhT16[0]) {
    if constexpr (S1 == 3) {
      if (nft == 2 && a.packed == 3 && a.WhT16[0] && a.WhT16[1]) {
        STPDE_LAUNCH((k_tail_bwd_bf<S1, S2, ACT>), grid, dim3(256), 0, stream, a);
        return stpde_check_launch("k_tail_bwd_bf");
      }
    }
    stpde_set_error("jet_tail_bwd: the bf16-operand kernel is compiled for nf = 32, S1 = 3, packed = 3 with the bf16 packs of fc3 / fc4");
    return STPDE_E_UNSUPPORTED;
  }
  if (nft == 2)
    STPDE_LAUNCH((k_tail_bwd<S1, S2, ACT, 2>), grid, dim3(256), 0, stream, a);
  else
    STPDE_LAUNCH((k_tail_bwd<S1, S2, ACT, 1>), grid, dim3(256), 0, stream, a);
  return stpde_check_launch("k_tail_bwd");
}

template <int S1, int S2>
static int launch_tailb_act(const TailBwdArgs& a, int nft, hipStream_t stream) {
  switch (a.cfg.act) {
    case STPDE_ACT_TANH: return launch_tailb_nft<S1, S2, STPDE_ACT_TANH>(a, nft, stream);
    case STPDE_ACT_RELU: return launch_tailb_nft<S1, S2, STPDE_ACT_RELU>(a, nft, stream);
    case STPDE_ACT_SOFTPLUS: return launch_tailb_nft<S1, S2, STPDE_ACT_SOFTPLUS>(a, nft, stream);
    case STPDE_ACT_ELU: return launch_tailb_nft<S1, S2, STPDE_ACT_ELU>(a, nft, stream);
    case STPDE_ACT_LEAKYRELU: return launch_tailb_nft<S1, S2, STPDE_ACT_LEAKYRELU>(a, nft, stream);
    default: return launch_tailb_nft<S1, S2, STPDE_ACT_SWISH>(a, nft, stream);
  }
}

static int tail_bwd(const stpde_jet_cfg* cfg, int ntiles, int nf16, const float* abar5, const float* const* WhT_pack,
                    const float* const* pre, float* const* abar_out, const float* cw, float* act_param_bar, int packed,
                    const void* const* WhT16_pack, void* stream);

extern "C" int stpde_jet_tail_bwd(const stpde_jet_cfg* cfg, int ntiles, int nf16, const float* abar5,
                                  const float* const* WhT_pack, const float* const* pre, float* const* abar_out,
                                  const float* cw, float* act_param_bar, void* stream) {
  return tail_bwd(cfg, ntiles, nf16, abar5, WhT_pack, pre, abar_out, cw, act_param_bar, 0, nullptr, stream);
}
extern "C" int stpde_jet_tail_bwd_p(const stpde_jet_cfg* cfg, int ntiles, int nf16, const float* abar5,
                                    const float* const* WhT_pack, const float* const* pre, float* const* abar_out,
                                    const float* cw, float* act_param_bar, int packed, const void* const* WhT16_pack,
                                    void* stream) {
  return tail_bwd(cfg, ntiles, nf16, abar5, WhT_pack, pre, abar_out, cw, act_param_bar, packed, WhT16_pack, stream);
}

static int tail_bwd(const stpde_jet_cfg* cfg, int ntiles, int nf16, const float* abar5, const float* const* WhT_pack,
                    const float* const* pre, float* const* abar_out, const float* cw, float* act_param_bar, int packed,
                    const void* const* WhT16_pack, void* stream) {
  if (!cfg || ntiles <= 0 || (nf16 != 1 && nf16 != 2) || !abar5 || !WhT_pack || !pre || !abar_out || cfg->act < 0 ||
      cfg->act > 5) {
    stpde_set_error("jet_tail_bwd: bad argument (nf must be 16 or 32)");
    return STPDE_E_BADARG;
  }
  TailBwdArgs a{};
  a.abar5 = abar5;
  for (int l = 0; l < 3; ++l) {
    if (!WhT_pack[l] || !pre[l] || !abar_out[l]) {
      stpde_set_error("jet_tail_bwd: null pointer for layer %d", 3 + l);
      return STPDE_E_BADARG;
    }
    a.WhT[l] = WhT_pack[l];
    a.pre[l] = pre[l];
    a.out[l] = abar_out[l];
  }
  a.cw = cw;
  a.pbar = act_param_bar;
  a.ntiles = ntiles;
  a.packed = packed;
  for (int l = 0; l < 3; ++l) a.WhT16[l] = WhT16_pack ? WhT16_pack[l] : nullptr;
  a.cfg = *cfg;
  const int S1 = cfg->S1, S2 = cfg->S2;
  if (S1 == 0 && S2 == 0) return launch_tailb_act<0, 0>(a, nf16, (hipStream_t)stream);
  if (S1 == 3 && S2 == 0) return launch_tailb_act<3, 0>(a, nf16, (hipStream_t)stream);
  if (S1 == 3 && S2 == 1 && cfg->combo && cw) return launch_tailb_act<3, 1>(a, nf16, (hipStream_t)stream);
  if (S1 == 3 && S2 == 2) return launch_tailb_act<3, 2>(a, nf16, (hipStream_t)stream);
  if (S1 == 3 && S2 == 4) return launch_tailb_act<3, 4>(a, nf16, (hipStream_t)stream);
  stpde_set_error("jet_tail_bwd: stream configuration S1=%d S2=%d not compiled", S1, S2);
  return STPDE_E_UNSUPPORTED;
}
