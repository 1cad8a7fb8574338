// Wave-specialised, persistent bf16 variants of the first-hidden-layer kernels (BASELINE configs[3], round 3).
//
// Why: with bf16 operands the MFMAs of a row tile take 1/16 of their fp32 time and what is left of k_layer_coop is a chain
// of latencies -- s_memtime stamps (tools/micro/ablate_layer.py stamp bf16, profiles/r3_stamp_bf16.txt) show a workgroup
// living 49 k cycles per row tile for 5 k cycles of MFMAs: 8 k before the first MFMA (raw-input loads from HBM, first
// produce stage), 7-9 k per group of 8 k-tiles (produce stage and MFMAs of one wave run back to back, every group waits
// for an L2 round trip of its weight fragments) and 8 k of epilogue.  The activation jets (VALU) and the bf16 MFMAs use
// different pipes of a SIMD, but one wave issues them one after the other.
//
// Here a workgroup has 8 waves with FIXED ROLES and stays resident (one per CU, walking row tiles with a grid stride):
//   waves 0-3  producers: layer 0 on the fly (fp32 MFMA) + activation jet + bf16 rounding of 2 k-tiles each per step into
//              an LDS ring, the z0 stash store; they run ONE STEP AHEAD of the consumers, across row-tile boundaries
//   waves 4-7  consumers: the bf16 MFMAs of 4 output tiles x S streams each, weight fragments through a register ring
//              4 k-tile pairs deep that runs across steps and row tiles (the weights do not depend on the tile), then the
//              skip GEMM / tangent constants / stores of the finished tile
// Every SIMD hosts one producer and one consumer wave, so its VALU and MFMA pipes work at the same time; one barrier per
// step.  Arithmetic: operand rounding and the accumulation order over the k-tile pairs are those of
// k_layer_coop<..., BF = true>; the skip GEMM is added during the steps instead of after the last one (fp32 rounding only).
// Measured (tools/micro/ablate_layer.py stamp_spec, profiles/r3_stamp_fc1_fwd_spec.txt): a step takes 3.0-3.5 k cycles for
// either role against 2.2 k cycles of MFMA-pipe time per SIMD (80 bf16 + 18-27 fp32 MFMAs) and 1.5 k of VALU time: 29.8 ->
// 23.2 ms per 2^20 points.  Tried without effect: s_setprio on either role, 8 waves x 2 output tiles at 4 waves per SIMD for
// the cooperative kernel (slower), persistent workgroups for the cooperative kernel (+-0).
#pragma once
#include <type_traits>


// NST = steps (groups of 8 k-tiles) per row tile = KT / 8, compile-time so that every ring / register-array index is static.
// Nothing in the producers' loop reads global memory except the prefetch of the next tile's raw input: their layer-0 weight
// fragments (always the same 2 * NST k-tiles) live in registers, the tangent constants W0[:, d] in LDS.
#if STPDE_STAMP
#define SPEC_STAMP(it, i)                                                                                         \
  do {                                                                                                            \
    if ((it) == 64 && blockIdx.x < 256 && (threadIdx.x & 63) == 0)                                                \
      g_stamp[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (i)] = __builtin_readcyclecounter();                   \
  } while (0)
#else
#define SPEC_STAMP(it, i)
#endif
template <int S1, int S2, int ACT, int NST>
__global__ __launch_bounds__(512, 2) void k_fc1_fwd_spec(LayerArgs a) {
  constexpr int S = 1 + S1 + S2, MCg = 4, GK = 8, KT = GK * NST, KP = KT / 2, ST = S1 == 3 ? 3 : 1;
  // bf16 fragment blocks (512 B each); the two k-tiles of a pair sit next to each other per stream, so that a consumer's
  // K = 32 operand is ONE ds_read2st64_b64 into four consecutive registers (round 4; with [slot][stream] the compiler paired
  // neighbouring streams instead and re-arranged the registers with 18 v_mov per k-tile pair)
  __shared__ __attribute__((aligned(16))) float hb[2][GK / 2][S][2][128];
  __shared__ __attribute__((aligned(16))) float tcl[ST][KT][256];      // tangent constants of layer 0 (S1 == 3)
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const bool consumer = wv >= 4;
  const int w = wv & 3;
  const int MT = a.MT;
  const int lo = lane * 4;
  const int G = gridDim.x;
  const int ntl = a.ntiles > (int)blockIdx.x ? (a.ntiles - (int)blockIdx.x + G - 1) / G : 0;   // row tiles of this workgroup

  if (S1 == 3) {
    for (int i = threadIdx.x; i < 3 * KT * 64; i += 512)
      st4(&tcl[0][0][0] + 4 * i, ld4(a.tanc0 + 4 * (size_t)i));
  }
  // wave-uniform by construction; told to the compiler (scalar offsets of the buffer loads / stores below)
  const int mt0 = __builtin_amdgcn_readfirstlane(w * MCg);
  __syncthreads();

  // The two roles are separate code paths with their own loops (and the same sequence of barriers: one after step 0 of
  // every iteration, one after each of the steps 1 .. NST-1), so that the register allocator does not keep the producers'
  // weight fragments and the consumers' accumulators / weight ring alive at the same time.
  if (!consumer) {
    // =========================================== producers ===========================================
    f32x4 w0r[NST][2][XT];
    f32x4 xb[XT], xn[XT];
    float cq[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    CqRaw cqn{};
#pragma unroll
    for (int g = 0; g < NST; ++g)
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int xt = 0; xt < XT; ++xt) w0r[g][k][xt] = ld4x(a.W0s + ((size_t)xt * KT + GK * g + 4 * k + w) * 256 + lo, xt);
    if (ntl > 0) {
#pragma unroll
      for (int xt = 0; xt < XT; ++xt) xn[xt] = ld4x(a.X + ((size_t)blockIdx.x * XT + xt) * 256 + lo, xt);
      load_cq_raw<S2>(a.cw, blockIdx.x * 2 + ((lane & 15) >> 3), cqn);
    }
    auto produce = [&](auto gc, int tile) {
      constexpr int g = decltype(gc)::value;
      const auto z0r = opt_store_rsrc(a.Z0 ? a.Z0 + (size_t)tile * KT * 256 : nullptr, (unsigned)KT * 1024u);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int slot = 4 * k + w, kt = GK * g + slot;
        f32x4 raw[S], B[S], part[XT];
#pragma unroll
        for (int xt = 0; xt < XT; ++xt) {
          f32x4 c = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int r = 0; r < x_live(xt); ++r) c = mfma4(w0r[g][k][xt][r], xb[xt][r], c);
          part[xt] = c;
        }
        raw[0] = (part[0] + part[1]) + part[2];
        opt_st4(z0r, kt * 1024 + lane * 16, raw[0]);
        if (S1 == 3) {
#pragma unroll
          for (int d = 0; d < 3; ++d) raw[1 + d] = ld4(&tcl[d][kt][lo]);
#pragma unroll
          for (int p = 0; p < S2; ++p) raw[4 + p] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        act_jet_fwd<S1, S2, ACT>(a.cfg, raw, B, cq);
#pragma unroll
        for (int st = 0; st < S; ++st) {
          const bf16x4 b4 = to_bf4(B[st]);
          *reinterpret_cast<bf16x4*>(&hb[g & 1][slot >> 1][st][slot & 1][lane * 2]) = b4;
        }
      }
    };
    for (int it = 0; it <= ntl; ++it) {
      const int tile = (int)blockIdx.x + it * G;
      if (it < ntl) {
#pragma unroll
        for (int xt = 0; xt < XT; ++xt) xb[xt] = xn[xt];
        unpack_cq<S2>(cqn, cq);
        SPEC_STAMP(it, 0);
        produce(std::integral_constant<int, 0>{}, tile);
      }
      SPEC_STAMP(it, 1);
      __syncthreads();
      SPEC_STAMP(it, 2);
      if (it == ntl) break;
      auto step = [&](auto gc) {
        constexpr int g = decltype(gc)::value;
        if (g == NST - 1) {           // prefetch the next tile's raw input (HBM latency behind this step)
          // Branch-free (round 6; the last tile re-reads itself): under `it + 1 < ntl` the number of loads issued here was
          // path-dependent, so every counter wait behind them -- the write-after-read waits on the z0 store data inside
          // produce() -- had to assume the path WITHOUT them: the listing had vmcnt(4) right behind the five requests, i.e. the
          // step that was to hide their HBM round trip waited for it.
          const int tn = it + 1 < ntl ? tile + G : tile;
#pragma unroll
          for (int xt = 0; xt < XT; ++xt) xn[xt] = ld4x(a.X + ((size_t)tn * XT + xt) * 256 + lo, xt);
          load_cq_raw<S2>(a.cw, tn * 2 + ((lane & 15) >> 3), cqn);
        }
        produce(gc, tile);
        SPEC_STAMP(it, 1 + 2 * g);
        __syncthreads();
        SPEC_STAMP(it, 2 + 2 * g);
      };
      if constexpr (NST > 1) step(std::integral_constant<int, 1>{});
      if constexpr (NST > 2) step(std::integral_constant<int, 2>{});
      if constexpr (NST > 3) step(std::integral_constant<int, 3>{});
    }
  } else {
    // =========================================== consumers ===========================================
    f32x4 acc[MCg][S];
    bf16x8 wr[4][MCg];
    f32x4 xb[1][XT];
    // weight fragments through a buffer descriptor (block offsets are instruction immediates / SGPRs, one lane-offset VGPR)
    const auto wrs = load_rsrc(a.Wp16, (unsigned)KP * MT * 1024u);
    const auto srs = load_rsrc(a.Wsp, (unsigned)XT * MT * 1024u);
    const auto trs = load_rsrc(a.tanc, (unsigned)3 * MT * 1024u);
    const int wlane = lane * 16;
    auto wload = [&](int kp, int mi) -> bf16x8 {
      return __builtin_bit_cast(bf16x8, buf_ld16(wrs, wlane, ((kp * MT + mt0 + mi) * 64) * 16));
    };
    // The skip GEMM with the raw input (bias through its ones column) and the tangent constants are ADDED to the accumulators
    // during the steps of the tile, so that little work is left when the last group is done.  Every address
    // is descriptor + scalar offset + the lane offset (no per-block address registers).
    // Round 4: the fragments of an output tile (3 skip-weight blocks + 3 tangent constants) are REQUESTED before the k-tile
    // pairs of a step and used after them -- the first version loaded each one right in front of its use and the compiler
    // waited for every single one (s_waitcnt vmcnt(0): six exposed L2 round trips per output tile, tools/micro/isa_waits.py);
    // one output tile per step-slot (the slot in front of the epilogue takes the last one), so one set of registers.
    constexpr int PER = MCg / NST, NSK = XT + (S1 == 3 ? 3 : 0);
    f32x4 sk[PER][NSK];
    auto skip_load = [&](int slot) {
#pragma unroll
      for (int p = 0; p < PER; ++p) {
        const int mi = slot * PER + p;
#pragma unroll
        for (int xt = 0; xt < XT; ++xt)
          sk[p][xt] = xt == XT - 1 ? f32x4{buf_ld4(srs, wlane, (xt * MT + mt0 + mi) * 1024), 0.f, 0.f, 0.f}      // (sparse tile: common.h ld4x)
                               : __builtin_bit_cast(f32x4, buf_ld16(srs, wlane, (xt * MT + mt0 + mi) * 1024));
        if (S1 == 3) {
#pragma unroll
          for (int d = 0; d < 3; ++d) sk[p][XT + d] = __builtin_bit_cast(f32x4, buf_ld16(trs, wlane, (d * MT + mt0 + mi) * 1024));
        }
      }
    };
    auto skip_apply = [&](auto slotc) {
      constexpr int slot = decltype(slotc)::value;
#pragma unroll
      for (int p = 0; p < PER; ++p) {
        constexpr int mi0 = slot * PER;
#pragma unroll
        for (int xt = 0; xt < XT; ++xt)
#pragma unroll
          for (int r = 0; r < x_live(xt); ++r) acc[mi0 + p][0] = mfma4(sk[p][xt][r], xb[0][xt][r], acc[mi0 + p][0]);
        if (S1 == 3) {
#pragma unroll
          for (int d = 0; d < 3; ++d) acc[mi0 + p][1 + d] += sk[p][XT + d];
        }
      }
    };
    auto epilogue = [&](int tile) {
      if (a.pk & 2) {      // packed output buffer (common.h): value stream fp32, derivative streams bf16
        const unsigned tb = (unsigned)packed_tile_bytes(S, MT);
        const auto ors = load_rsrc(reinterpret_cast<const char*>(a.Out) + (size_t)tile * tb, tb);
#pragma unroll
        for (int mi = 0; mi < MCg; ++mi) {
          buf_st16(ors, wlane, (mt0 + mi) * 1024, acc[mi][0]);
#pragma unroll
          for (int st = 1; st < S; ++st)
            buf_st8(ors, lane * 8, MT * 1024 + ((st - 1) * MT + mt0 + mi) * 512, to_bf4(acc[mi][st]));
        }
        return;
      }
      const auto ors = load_rsrc(a.Out + (size_t)tile * S * MT * 256, (unsigned)S * MT * 1024u);
#pragma unroll
      for (int mi = 0; mi < MCg; ++mi)
#pragma unroll
        for (int st = 0; st < S; ++st) buf_st16(ors, wlane, (st * MT + mt0 + mi) * 1024, acc[mi][st]);
    };
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int mi = 0; mi < MCg; ++mi) wr[q][mi] = wload(q, mi);
    // the 4 k-tile pairs of group gg (compile-time) out of ring buffer (gg & 1)
    auto consume = [&](auto ggc) {
      constexpr int gg = decltype(ggc)::value;
      // stream-major MFMA order (round 4): the operand of stream st is dead after its MCg MFMAs and is re-read for the NEXT
      // k-tile pair right there, (S - 1) * MCg MFMAs ahead of its use -- the compiler's own schedule read all S operands after
      // the last MFMA of a pair and waited for them (one exposed LDS latency + 18 v_mov per pair, the MFMA pipe idle)
      auto rd = [&](int q, int st) -> bf16x8 {
        return cat8(*reinterpret_cast<const bf16x4*>(&hb[gg & 1][q][st][0][lane * 2]),
                    *reinterpret_cast<const bf16x4*>(&hb[gg & 1][q][st][1][lane * 2]));
      };
      bf16x8 B8[S];
#pragma unroll
      for (int st = 0; st < S; ++st) B8[st] = rd(0, st);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int st = 0; st < S; ++st) {
#pragma unroll
          for (int mi = 0; mi < MCg; ++mi) acc[mi][st] = mfma_bf(wr[q][mi], B8[st], acc[mi][st]);
          if (q + 1 < 4) B8[st] = rd(q + 1, st);
          __builtin_amdgcn_sched_barrier(0);
        }
        // ring: this slot now fetches the pair four ahead (wraps into the next row tile: the weights are the same)
        const int kpn = (4 * gg + q + 4) % KP;          // compile-time after unrolling
#pragma unroll
        for (int mi = 0; mi < MCg; ++mi) wr[q][mi] = wload(kpn, mi);
      }
    };
    for (int it = 0; it <= ntl; ++it) {
      const int tile = (int)blockIdx.x + it * G;        // the producers' tile; step 0 finishes tile - G here
      SPEC_STAMP(it, 0);
      if (it >= 1) {
        skip_load(NST - 1);
        consume(std::integral_constant<int, NST - 1>{});
        skip_apply(std::integral_constant<int, NST - 1>{});
        SPEC_STAMP(it, 9);
        epilogue(tile - G);
      }
      SPEC_STAMP(it, 1);
      __syncthreads();
      SPEC_STAMP(it, 2);
      if (it == ntl) break;
      auto step = [&](auto gc) {
        constexpr int g = decltype(gc)::value;
        if (g == 1) {
#pragma unroll
          for (int mi = 0; mi < MCg; ++mi)
#pragma unroll
            for (int st = 0; st < S; ++st) acc[mi][st] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int xt = 0; xt < XT; ++xt) xb[0][xt] = ld4x(a.X + ((size_t)tile * XT + xt) * 256 + lo, xt);   // for the epilogue
        }
        skip_load(g - 1);
        consume(std::integral_constant<int, g - 1>{});
        // this step's share of the skip GEMM: output tile(s) (g - 1) * PER ... (the last share runs in front of the epilogue)
        skip_apply(std::integral_constant<int, g - 1>{});
        SPEC_STAMP(it, 1 + 2 * g);
        __syncthreads();
        SPEC_STAMP(it, 2 + 2 * g);
      };
      if constexpr (NST > 1) step(std::integral_constant<int, 1>{});
      if constexpr (NST > 2) step(std::integral_constant<int, 2>{});
      if constexpr (NST > 3) step(std::integral_constant<int, 3>{});
    }
  }
  static_assert(NST <= 4 && NST >= 2 && NST % 2 == 0, "ring parity needs an even number of steps per tile");
}

// (Round 4's wave-specialised input gradient of this layer, k_fc1_dgrad_spec, was superseded in round 5 by the fused backward
// k_fc1_bwd_fused, csrc/jet_fc1_bwd.hip, and deleted in round 6; a backward that wants no weight gradients, or a stream set
// the fused kernel is not compiled for, takes k_layer_coop<..., EPI_ADJ_L0, BF>.)

// Workgroup barrier WITHOUT the vector-memory drain: __syncthreads() of a wave that has a global_load_lds in flight is
// compiled to `s_waitcnt vmcnt(0) lgkmcnt(0) ; s_barrier`, i.e. every step would wait for the HBM round trip of the piece it
// has just requested.  LDS writes of this wave are complete at lgkmcnt(0); pieces in flight are waited for explicitly where
// their data is needed.
#define SPEC_BARRIER()                                     \
  do {                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_s_barrier();                          \
    asm volatile("" ::: "memory");                         \
  } while (0)
// 16 bytes per lane from global memory straight into LDS (no registers, no wait at a use; common.h: glds16)
#define STPDE_GLDS16(gptr, lptr) glds16((gptr), (lptr))

template <int S1, int S2, int ACT>
static int launch_fc1_fwd_spec(const LayerArgs& a, hipStream_t stream) {
  int dev = 0, ncu = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  int grid = ncu < a.ntiles ? ncu : a.ntiles;
  if (a.KT == 32)
    STPDE_LAUNCH((k_fc1_fwd_spec<S1, S2, ACT, 4>), dim3(grid), dim3(512), 0, stream, a);
  else
    STPDE_LAUNCH((k_fc1_fwd_spec<S1, S2, ACT, 2>), dim3(grid), dim3(512), 0, stream, a);
  return stpde_check_launch("k_fc1_fwd_spec");
}

// ------------------------------------------------------------------------------------------------------------
// bf16 mode, SECOND hidden layer forward (reference width: KT = 16 input tiles, MT = 8 output tiles, packed stash in and out;
// round 5).  In k_layer_coop<..., BF> a row tile is 6k cycles of issued work per wave inside 41k cycles of wave life: phase
// stamps (tools/micro/ablate_layer.py run2 bf16) show the two 8-k-tile groups at 6-7k cycles each for 640 cycles of MFMAs
// -- stash loads, activation jets, ring stores and a barrier in series per group -- and the timing ablations put a third of the
// launch on the loads and a quarter on the stores: the kernel streams 41 GB at 3.5 TB/s, not because it computes.
// Here ONE persistent workgroup of 8 waves per CU:
//   * the packed stash tile of a row tile (48 KB, contiguous in HBM) lands in LDS by global_load_lds, 6 x 1 KiB pieces per
//     wave, requested when the buffer's previous tile has been consumed -- a tile time ahead of its use (double buffer): no
//     registers, no wait at a use;
//   * wave w activates input tiles 2w, 2w + 1 (one activation-jet evaluation per element and workgroup) into a 40 KB bf16 block
//     image, barrier, then owns OUTPUT tile w: 8 k-tile pairs x S bf16 MFMAs with its 8 weight fragments, skip-weight and
//     tangent-constant fragments resident in registers for the whole launch (no weight traffic in the loop);
//   * combined-stream weights and the raw-input tile of a row tile arrive the same way, together with its pieces.
// Two barriers per row tile; the only vector-memory wait is one counted s_waitcnt per tile for requests made a whole tile
// earlier (the tile's stores leave a tile late, behind that wait).  Operand rounding, accumulation order over the k-tile pairs and epilogue are those of
// k_layer_coop<..., BF = true, PK = 2, PKM = 3>: bit-identical output.
// ------------------------------------------------------------------------------------------------------------
template <int S1, int S2, int ACT>
__global__ __launch_bounds__(512, 1) void k_fc2_fwd_bf(LayerArgs a) {
  constexpr int S = 1 + S1 + S2, KT = 16, MT = 8, KP = KT / 2;
  constexpr int TILE_IN = KT * (1024 + (S - 1) * 512);                     // bytes of a packed stash tile (49152 at S = 5)
  constexpr int NQ = TILE_IN / 1024 / 8;                                   // 1 KiB pieces per wave and row tile
  static_assert(TILE_IN % 8192 == 0, "whole pieces per wave");
  __shared__ __attribute__((aligned(16))) float raw[2][TILE_IN / 4];       // the tile as it lies in HBM
  __shared__ __attribute__((aligned(16))) float act[KT][S][128];           // activated blocks: [kt][stream][lane][4 bf16]
  __shared__ __attribute__((aligned(16))) float xs[3][XT][256];            // raw-input tiles (skip GEMM of the epilogue)
  __shared__ float cqs[3][16];                                             // combined-stream weights of the two points of a tile
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lo = lane * 4;
  const int ntl = ((int)blockIdx.x < a.ntiles) ? (a.ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  if (ntl == 0) return;
  // launch-resident operands of output tile w
  bf16x8 wk[KP];
#pragma unroll
  for (int q = 0; q < KP; ++q) wk[q] = reinterpret_cast<const bf16x8*>(a.Wp16)[((size_t)q * MT + w) * 64 + lane];
  f32x4 wsk[XT], tck[3];
#pragma unroll
  for (int xt = 0; xt < XT; ++xt) wsk[xt] = ld4x(a.Wsp + ((size_t)xt * MT + w) * 256 + lo, xt);
#pragma unroll
  for (int d = 0; d < 3; ++d) tck[d] = S1 == 3 ? ld4(a.tanc + ((size_t)d * MT + w) * 256 + lo) : f32x4{0.f, 0.f, 0.f, 0.f};
  auto stage = [&](int tile, int buf) {
    const char* src = reinterpret_cast<const char*>(a.Bin) + (size_t)tile * TILE_IN + lane * 16;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int c = __builtin_amdgcn_readfirstlane(w + 8 * q);
      STPDE_GLDS16(src + (size_t)c * 1024, reinterpret_cast<char*>(&raw[buf][0]) + c * 1024);
    }
  };
  // requests of one row tile, all by LDS-DMA: its stash pieces (every wave), its raw-input tile (waves 0-2, one KiB each), its
  // 16 combined-stream weights (wave 3, lanes 0-15, 4 bytes each).  The loop has NO ordinary global load and the requests are
  // invisible to the compiler (glds16: until round 6 it put a vmcnt(0) in front of the first read of raw[buf], which waited
  // for the acknowledgements of the stores issued right above it), so the only vector-memory wait of a tile is the counted one
  // below; the raw-input / weight slots rotate through three buffers (a slot is rewritten
  // two tiles after its last reader, with two barriers in between).
  auto request = [&](int tile, int buf, int b3) {
    stage(tile, buf);
    if (w < XT) STPDE_GLDS16(a.X + ((size_t)tile * XT + w) * 256 + lo, &xs[b3][w][0]);
    if (S2 == 1 && w == 3 && lane < 16)
      glds4(a.cw + (size_t)tile * 16 + lane, &cqs[b3][0]);
  };
  request(blockIdx.x, 0, 0);
  if (ntl > 1) request(blockIdx.x + gridDim.x, 1, 1);
  f32x4 hold[S];                                                           // output blocks of the previous tile, stored a tile late
  int b3 = 0;                                                              // i % 3
#pragma unroll 1
  for (int i = 0; i < ntl; ++i) {
    const int tile = blockIdx.x + i * gridDim.x, buf = i & 1;
    // The ONE vector-memory wait of the tile.  Vector-memory loads return in order, so "at most NREQ outstanding" (NREQ = this
    // wave's requests per tile) means: everything older than the requests of tile i + 1 -- i.e. the requests of tile i -- has
    // arrived.  The previous tile's stores are issued BEHIND this point (from `hold`), so nothing younger than a tile time
    // is ever waited for; stores still in flight can only make the count larger.
    if (i + 1 < ntl) {
      if (w < XT || (S2 == 1 && w == 3))
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NQ + 1) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NQ) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    SPEC_BARRIER();                                                        // all requests of tile i landed; act free
    if (i > 0) {
#pragma unroll
      for (int st = 0; st < S; ++st) st_blk(a.Out, 1, tile - gridDim.x, S, MT, st, w, lane, hold[st]);
    }
    float cq[6];
    if (S2 == 1) {
#pragma unroll
      for (int k = 0; k < 6; ++k) cq[k] = cqs[b3][8 * ((lane & 15) >> 3) + k];
    }
    // ---- activation jets of input tiles 2w, 2w + 1
    const char* rb = reinterpret_cast<const char*>(&raw[buf][0]);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int kt = 2 * w + e;
      f32x4 rv[S], B[S];
      rv[0] = *reinterpret_cast<const f32x4*>(rb + kt * 1024 + lane * 16);
#pragma unroll
      for (int st = 1; st < S; ++st)
        rv[st] = bf4_to_f32(*reinterpret_cast<const bf16x4*>(rb + KT * 1024 + ((st - 1) * KT + kt) * 512 + lane * 8));
      act_jet_fwd<S1, S2, ACT>(a.cfg, rv, B, cq);
#pragma unroll
      for (int st = 0; st < S; ++st) *reinterpret_cast<bf16x4*>(&act[kt][st][lane * 2]) = to_bf4(B[st]);
    }
    SPEC_BARRIER();                                                        // block image complete; raw[buf] consumed
    // ---- tile i + 2 into the stash buffer this tile has consumed
    if (i + 2 < ntl) request(tile + 2 * gridDim.x, buf, b3 == 0 ? 2 : b3 - 1);
    // ---- output tile w
    f32x4 acc[S];
#pragma unroll
    for (int st = 0; st < S; ++st) acc[st] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < KP; ++q) {
#pragma unroll
      for (int st = 0; st < S; ++st) {
        const bf16x8 B8 = cat8(*reinterpret_cast<const bf16x4*>(&act[2 * q][st][lane * 2]),
                               *reinterpret_cast<const bf16x4*>(&act[2 * q + 1][st][lane * 2]));
        acc[st] = mfma_bf(wk[q], B8, acc[st]);
      }
    }
#pragma unroll
    for (int xt = 0; xt < XT; ++xt) {
      const f32x4 xv = ld4(&xs[b3][xt][lo]);
#pragma unroll
      for (int r = 0; r < x_live(xt); ++r) acc[0] = mfma4(wsk[xt][r], xv[r], acc[0]);
    }
    if (S1 == 3) {
#pragma unroll
      for (int d = 0; d < 3; ++d) acc[1 + d] += tck[d];
    }
#pragma unroll
    for (int st = 0; st < S; ++st) hold[st] = acc[st];
    b3 = b3 == 2 ? 0 : b3 + 1;
  }
#pragma unroll
  for (int st = 0; st < S; ++st) st_blk(a.Out, 1, blockIdx.x + (ntl - 1) * gridDim.x, S, MT, st, w, lane, hold[st]);
}

template <int S1, int S2, int ACT>
static int launch_fc2_fwd_bf(const LayerArgs& a, hipStream_t stream) {
  int dev = 0, ncu = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  const int grid = ncu < a.ntiles ? ncu : a.ntiles;
  STPDE_LAUNCH((k_fc2_fwd_bf<S1, S2, ACT>), dim3(grid), dim3(512), 0, stream, a);
  return stpde_check_launch("k_fc2_fwd_bf");
}
