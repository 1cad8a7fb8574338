// Backward of the first hidden layer in ONE kernel (bf16 mode, packed layer buffers; round 5):
//   input gradient   hbar0 = W1h^T abar1 ; abar0 = act_jet_adjoint(hbar0 ; z0)          (was k_fc1_dgrad_spec)
//   weight gradient  dW1h += abar1^T act_jet(z0)                                          (was k_wgrad_coop<.., MODE 1, BF>)
// i.e. what loss.backward() (experiments/rb2d/train.py:77) does through fc1 of src/implicit_net.py:48-54 on all derivative
// streams of the src/pde.py:8-9 sweeps.
//
// Why one kernel: both halves evaluate sigma ... sigma''' at the SAME z0 elements (the forward did it a first time) and both
// stream the same adjoint tile; profiles/r4_bf16_pmc_sq_counters.txt: 5.9 non-MFMA VALU instructions per MFMA in the weight
// gradient, ~60 % of them the activation jet, matrix pipe busy 26 %; the input gradient's adjoint waves were its long pole.
// Here the packed adjoint tile of a row tile lands in LDS ONCE per workgroup (global_load_lds, the bytes as they lie in HBM)
// and is read from there as both operands it is -- the B operand of W1h^T abar1 (plain 8-byte reads: the column-major image)
// and, through the hardware transpose read ds_read_b64_tr_b16, the A operand of abar1^T h0 -- and every z0 element goes through
// act_eval ONCE, feeding the layer-0 adjoint and the bf16 operand blocks h0 of the weight gradient (which never leave the
// wave: a private LDS patch turns them into the row-major image).
//
// Work split: the 512 x 256 block of dW1h cannot stay resident in one workgroup's registers, so TWO workgroups share a row
// tile: each owns 16 of the 32 feature tiles of layer 0 -- input gradient of those 16 tiles (contraction over all of abar1),
// weight gradient dW1h[:, its 16 tiles].  A workgroup = 4 waves, ONE per SIMD, 512 registers each: wave w owns 4 feature
// tiles, keeps dW1h[16 output tiles][its 4 tiles] = 64 accumulator blocks (256 AGPRs) for the whole launch (persistent, grid
// stride over row tiles, atomics at the end), streams W1h^T from L2 through a register ring and the adjoint tile from LDS.
// The two halves of a row tile sit on the same XCD (blocks b and b + 8), so HBM delivers the adjoint tile once.
// The raw-input columns of dW1 (skip connection / bias: 3 k-tiles, value stream + tangent column sums) are dealt over the
// eight waves of a row tile's two workgroups (two output tiles each).
// Arithmetic: operand rounding and the accumulation order of the input gradient are those of k_fc1_dgrad_spec (bit-identical
// abar0 / tangent row sums); the weight gradient sums the same bf16 products in another order (fp32 summation rounding).
#include "jet_wgrad_impl.h"

#ifndef STPDE_FC1F_ABL
#define STPDE_FC1F_ABL 0     // timing-only ablations (results WRONG): 1 = no activation jets, 2 = no weight-gradient MFMAs, 3 = no input-gradient MFMAs
#endif

// Phase timing (tools/micro/ablate_fc1_fused.py, private build with -DSTPDE_FC1F_STAMP=1): s_memtime stamps of every wave of the
// first 8 workgroups at the phase boundaries of their 5th row tile, read back with stpde_fc1f_stamp_read.
#ifndef STPDE_FC1F_STAMP
#define STPDE_FC1F_STAMP 0
#endif
#if STPDE_FC1F_STAMP
static __device__ unsigned long long g_fc1f_stamp[8 * 4 * 8];
#define FSTAMP(i)                                                                                   \
  do {                                                                                              \
    if (it == 4 && blockIdx.x < 8 && lane == 0) g_fc1f_stamp[(blockIdx.x * 4 + w) * 8 + (i)] = __builtin_readcyclecounter(); \
  } while (0)
extern "C" int stpde_fc1f_stamp_read(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_fc1f_stamp), sizeof(g_fc1f_stamp));
}
#else
#define FSTAMP(i)
#endif

struct Fc1BwdArgs {
  const float* abar1;   // packed ADJOINT buffer of fc1's rows: [tile][S][16][64 lanes][4 bf16]
  const void* WT16;     // bf16 pack of W1h^T: [8 k-tile pairs][32 output tiles][64 lanes] x 8 bf16
  const float* Z0;      // [tile][32][256] value stream of layer 0's pre-activations
  const float* tanc0;   // [3][32][256] tangent constants W0[:, d], column-major image
  const float* cw;      // [P][8] weights of the combined second-order stream (S2 == 1)
  const float* X;       // [tile][XT][256] augmented raw input, column-major image
  float* abar0;         // out: value-stream adjoint of layer 0, packed ADJOINT blocks [tile][32][64][4 bf16]
  float* Tan0;          // out: [tile][32][3][16] row sums of the tangent-stream adjoints of layer 0
  float* dW;            // [256][16 * (32 + XT)] fp32, atomically accumulated
  float* pbar;          // swish: [STPDE_PBAR_SLOTS] (nullable)
  int ntiles;
  int det;              // dW addresses long accumulators (stpde_layer_desc.det)
  stpde_jet_cfg cfg;
};

#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#ifndef STPDE_FC1F_RING
#define STPDE_FC1F_RING 3    // k-tile pairs of W1h^T in flight per feature-tile pair
#endif
#ifndef STPDE_FC1F_VPM
#define STPDE_FC1F_VPM 10    // vector instructions placed behind every MFMA of the mixed phases
#endif
// row tiles are staged by LDS-DMA through common.h's glds16 (an asm statement: the compiler's bookkeeping of the builtin put
// vmcnt(0) in front of the first read of the OTHER staging buffer, i.e. right behind the request)
#define FC1F_GLDS16(gptr, lptr) glds16((gptr), (lptr))
// workgroup barrier without the compiler's vector-memory drain in front of it (LDS traffic of this wave complete)
#define FC1F_BARRIER()                                     \
  do {                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_s_barrier();                          \
    asm volatile("" ::: "memory");                         \
  } while (0)

// DET (compile time, not a run-time branch: with one the 256 accumulator registers of dW1h went to scratch -- 1040 bytes per
// lane, configs[3] 145 -> 182 ms): the final sums go to long accumulators (deterministic mode, common.h) instead of fp32 atomics
template <int S2, int ACT, bool DET = false>
__global__ __launch_bounds__(256) void k_fc1_bwd_fused(Fc1BwdArgs a) {
  constexpr int S1 = 3, S = 1 + S1 + S2, MT = 32, KT = 16, KP = KT / 2, NQ = 4;
  constexpr int NCH = S * KT / 2;          // 1 KiB chunks of a packed adjoint tile (40 at S = 5)
  constexpr int NP = (S + 1) / 2;          // stream pairs of the weight-gradient MFMAs (K = 32 = 2 streams x 16 rows)
  static_assert(NCH % 4 == 0, "chunks are dealt to four waves");
  __shared__ __attribute__((aligned(16))) float bst[2][NCH * 256];     // two row tiles as they lie in HBM
  __shared__ __attribute__((aligned(16))) float hp[4][NQ][S][128];     // h0 operand blocks of each wave's own feature tiles (bf16)
  __shared__ __attribute__((aligned(16))) float tcl[4][NQ][3][16];     // tangent constants: features 4g .. 4g+3 at [4 g + r]
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lo = lane * 4;
  const int b = blockIdx.x, G = gridDim.x;
  // the two halves of a row tile on one XCD (block -> XCD is b % 8) when the grid allows it
  const bool xa = (G % 16) == 0;
  const int half = xa ? ((b >> 3) & 1) : (b & 1);
  const int pair = xa ? ((b & 7) + 8 * (b >> 4)) : (b >> 1);
  const int npairs = G / 2;
  const int ntl = a.ntiles > pair ? (a.ntiles - pair + npairs - 1) / npairs : 0;
  const int kt0 = 16 * half + w;           // feature tile of local index q: kt0 + 4 q
  // transpose-read offset (bytes) inside a 512-byte bf16 block stored lane by lane ([g][row][4 features]): lane i of group g'
  // passes the 8 bytes of lane (i % 4, 4 g' + i / 4) and receives rows 4g' .. 4g'+3 of feature i (common.h: lds_read_tr16)
  const int trofs = (16 * (lane & 3) + 4 * (lane >> 4) + ((lane & 15) >> 2)) * 8;
  const bf16x4 zero4 = to_bf4(f32x4{0.f, 0.f, 0.f, 0.f});

  f32x4 dw[KT][NQ];
#pragma unroll
  for (int m = 0; m < KT; ++m)
#pragma unroll
    for (int q = 0; q < NQ; ++q) dw[m][q] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Raw-input columns of dW1 (skip connection / bias: XT k-tiles; value stream x raw input, tangent columns = row sums of the
  // tangent-stream adjoints through the pattern operand [feature == d], as k_wgrad_coop's XB path): the 16 output tiles are
  // dealt two to each of the eight waves of a row tile's two workgroups -- 6 accumulator blocks per wave, 8 MFMAs per row tile
  // (the first version kept a separate launch of the ring kernel for them: 5.5 ms per 2^20 points to re-read every adjoint)
  const int mx = 2 * (4 * half + w);
  f32x4 dwx[2][XT];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int xt = 0; xt < XT; ++xt) dwx[mi][xt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bf16x4 one4 = to_bf4(f32x4{1.f, 1.f, 1.f, 1.f});
  const bf16x4 pat0 = (lane & 15) == 0 ? one4 : zero4;
  const bf16x8 pat12 = cat8((lane & 15) == 1 ? one4 : zero4, (lane & 15) == 2 ? one4 : zero4);

  // tangent constants W0[:, d] of this wave's feature tiles: in the column-major image a block holds the same four values in
  // every row, so 16 floats per (tile, d) in LDS and one broadcast ds_read_b128 per use (48 registers otherwise: the 256
  // accumulator registers of dW1h leave 256 for everything else)
  if (lane < 16) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int d = 0; d < 3; ++d)
        tcl[w][q][d][lane] = a.tanc0[((size_t)d * MT + kt0 + 4 * q) * 256 + 64 * (lane >> 2) + (lane & 3)];
  }

  // W1h^T fragments: descriptor + scalar offsets, a ring two k-tile pairs deep that runs across row tiles
  const auto wrs = load_rsrc(a.WT16, (unsigned)KP * MT * 1024u);
  const int wlane = lane * 16 + kt0 * 1024;
  auto wload = [&](int kp, int q) -> bf16x8 {
    return __builtin_bit_cast(bf16x8, buf_ld16(wrs, wlane, (kp * MT + 4 * q) * 1024));
  };
  // (ring depth: a stage is 10 MFMAs = 170 cycles; two stages ahead did not cover an L2 round trip under load -- the input
  // gradient ran at 2.5x its MFMA time)
  constexpr int RD = STPDE_FC1F_RING;
  bf16x8 wr[RD][2];
#pragma unroll
  for (int s = 0; s < RD; ++s)
#pragma unroll
    for (int q = 0; q < 2; ++q) wr[s][q] = wload(s, q);

  auto stage = [&](int tile, int bb) {      // this wave's chunks (w + 4 c) of row tile `tile` into buffer bb
    const char* src = reinterpret_cast<const char*>(a.abar1) + (size_t)tile * (NCH * 1024) + lane * 16;
    char* dst = reinterpret_cast<char*>(&bst[bb][0]);
#pragma unroll
    for (int c = 0; c < NCH / 4; ++c) {
      const int ch = __builtin_amdgcn_readfirstlane(w + 4 * c);
      FC1F_GLDS16(src + (size_t)ch * 1024, dst + ch * 1024);
    }
  };
  f32x4 z0n[NQ];
  CqRaw cqn{};         // combination weights of the NEXT row tile as loaded (common.h: unpacked where the tile starts)
  float cq[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](int tile) {               // z0 blocks and combination weights of row tile `tile`
#pragma unroll
    for (int q = 0; q < NQ; ++q) z0n[q] = ld4(a.Z0 + ((size_t)tile * MT + kt0 + 4 * q) * 256 + lo);
    load_cq_raw<S2>(a.cw, tile * 2 + ((lane & 15) >> 3), cqn);
  };
  if (ntl > 0) {
    stage(pair, 0);
    fetch(pair);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  FC1F_BARRIER();
  float pacc = 0.f;

  for (int it = 0; it < ntl; ++it) {
    const int tile = pair + it * npairs;
    const int buf = it & 1;
    const bool more = it + 1 < ntl;
    const int tnext = more ? tile + npairs : tile;
    // (the next row tile's request: see below, behind the second input-gradient phase)
    unpack_cq<S2>(cqn, cq);

    // Schedule of a row tile.  One wave per SIMD: nothing but this wave's own instruction stream hides a latency, so the matrix
    // phases request every LDS operand one step ahead of its MFMAs and a step carries enough MFMAs (20 / 12 x 17 cycles) to
    // cover an LDS round trip; the first version read each operand right in front of its use and spent 20 k cycles per row
    // tile on 6 k cycles of MFMAs (tools/micro/isa_trace.py: `s_waitcnt ; 2 MFMAs` all the way).
    //   1  input gradient of feature tiles 0, 1 (80 MFMAs, B fragments of k-tile pair kp + 1 in flight), their adjoints + h0 blocks
    //   2  the same for feature tiles 2, 3
    //   3  weight gradient of the four tiles (192 MFMAs, adjoint fragments of output tile m + 1 in flight)
    f32x4 acc[2][S];          // the input gradient runs two feature tiles at a time: 40 accumulator registers instead of 80
    const bf16x4* bs = reinterpret_cast<const bf16x4*>(&bst[buf][0]) + lane;
    // hbar0 of feature tiles q0, q0 + 1 = W1h^T abar1; wr[.][0..1] hold the first two k-tile pairs of their weights on entry
    auto dgrad2 = [&](int q0) {
#pragma unroll
      for (int qq = 0; qq < 2; ++qq)
#pragma unroll
        for (int st = 0; st < S; ++st) acc[qq][st] = f32x4{0.f, 0.f, 0.f, 0.f};
      bf16x8 Bn[S];
#pragma unroll
      for (int st = 0; st < S; ++st) Bn[st] = cat8(bs[(st * KT) * 64], bs[(st * KT + 1) * 64]);
#pragma unroll
      for (int kp = 0; kp < KP; ++kp) {
        bf16x8 B8[S];
#pragma unroll
        for (int st = 0; st < S; ++st) B8[st] = Bn[st];
        if (kp + 1 < KP && STPDE_FC1F_ABL != 5) {
#pragma unroll
          for (int st = 0; st < S; ++st) Bn[st] = cat8(bs[(st * KT + 2 * kp + 2) * 64], bs[(st * KT + 2 * kp + 3) * 64]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
#pragma unroll
          for (int st = 0; st < S; ++st)
            if (STPDE_FC1F_ABL != 3) acc[qq][st] = mfma_bf(wr[kp % RD][qq], B8[st], acc[qq][st]);
          if (kp + RD < KP && STPDE_FC1F_ABL != 4) wr[kp % RD][qq] = wload(kp + RD, q0 + qq);      // ring: RD pairs ahead
        }
      }
    };
    auto wring = [&](int q0) {     // first two k-tile pairs of the weights of feature tiles q0, q0 + 1
#pragma unroll
      for (int s2 = 0; s2 < RD; ++s2)
#pragma unroll
        for (int qq = 0; qq < 2; ++qq)
          if (STPDE_FC1F_ABL != 4) wr[s2][qq] = wload(s2, q0 + qq);
    };

    // the tangent row sums leave through a buffer descriptor over this row tile's [32][48] floats: lanes that hold no sum pass
    // an offset outside it and their stores are dropped by the hardware -- no branch in the block
    const auto tanr = load_rsrc(a.Tan0 + (size_t)tile * MT * 48, (unsigned)MT * 48u * 4u);
    const int tanlane = (lane & 15) == 15 ? 16 * (lane >> 4) : 0x7fffff00;
    // ---------------- per feature tile: one activation-jet evaluation -> layer-0 adjoint + h0 operand blocks
    auto epi = [&](int q) {
      const int kt = kt0 + 4 * q;
      f32x4 pre[S], ab[S], h[S];
      pre[0] = z0n[q];
#pragma unroll
      for (int d = 0; d < 3; ++d) pre[1 + d] = ld4(&tcl[w][q][d][4 * (lane >> 4)]);
#pragma unroll
      for (int p = 0; p < S2; ++p) pre[4 + p] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (STPDE_FC1F_ABL == 1) {
#pragma unroll
        for (int st = 0; st < S; ++st) {
          ab[st] = acc[q & 1][st] + pre[st];
          h[st] = pre[st];
        }
      } else {
        act_jet_adj<S1, S2, ACT>(a.cfg, pre, acc[q & 1], ab, cq);
        act_jet_fwd<S1, S2, ACT>(a.cfg, pre, h, cq);
      }
      if (ACT == STPDE_ACT_SWISH && a.pbar) pacc += swish_beta_adj<S1, S2>(a.cfg, pre, acc[q & 1], cq);
      *reinterpret_cast<bf16x4*>(reinterpret_cast<char*>(a.abar0) + ((size_t)tile * MT + kt) * 512 + lane * 8) = to_bf4(ab[0]);
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const f32x4 ts = row_sum16x4(ab[1 + d]);      // (hazard-safe form: this wave has its SIMD to itself, common.h)
        buf_st16(tanr, tanlane, (__builtin_amdgcn_readfirstlane(kt) * 48 + 16 * d) * 4, ts);
        // (round 6: wait states between the 16-byte store and the next write of its data registers.  The compiler re-uses them
        // for the next row sum's inputs, its hazard model exempts buffer stores with a scalar offset, and a two-waves-per-SIMD
        // variant of this kernel -- four workgroups per row tile, measured slower and not kept, DESIGN 8 -- stored the NEW value
        // of the first data dword now and then when the overwrite was the very next instruction; tools/check_dpp_hazard.py rule 2)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 2");
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int st = 0; st < S; ++st) *reinterpret_cast<bf16x4*>(&hp[w][q][st][lane * 2]) = to_bf4(h[st]);
    };
    // (one tile at a time: interleaving the evaluations multiplies their temporaries past the 256 registers the accumulators of
    // dW1h leave, and every spilled register is a scratch access inside the loop)
    FSTAMP(0);
    dgrad2(0);
    __builtin_amdgcn_sched_barrier(0);
    FSTAMP(1);
    wring(2);
    epi(0);
    __builtin_amdgcn_sched_barrier(0);
    epi(1);
    __builtin_amdgcn_sched_barrier(0);
    FSTAMP(2);
    dgrad2(2);
    __builtin_amdgcn_sched_barrier(0);
    FSTAMP(3);
    // The next row tile's adjoint goes to the other buffer from HERE (round 6; it used to be requested at the top of the
    // iteration): vector-memory results return in order, so every load issued behind these ten requests waits for them too --
    // and the input-gradient phases consume weight fragments a few hundred cycles after they ask for them.  From this point on
    // the iteration issues stores and loads that are not needed before its last MFMAs (next tile's z0 blocks / weight ring,
    // the raw-input fragments): ~8,000 cycles for an HBM round trip.
    if (more && STPDE_FC1F_ABL != 6) stage(tnext, buf ^ 1);
    epi(2);
    __builtin_amdgcn_sched_barrier(0);
    epi(3);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_sched_barrier(0);
    FSTAMP(4);
    // next row tile's z0 blocks / combination weights and the first stages of its weight ring (the same weights for every row
    // tile): they land while the weight-gradient MFMAs run
    fetch(tnext);
    wring(0);
    // ---------------- weight gradient: dW1h[:, this wave's tiles] += abar1^T h0
    {
      bf16x8 H8[NQ][NP];
#pragma unroll
      for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const char* h0 = reinterpret_cast<const char*>(&hp[w][q][2 * p][0]) + trofs;
          const char* h1 = reinterpret_cast<const char*>(&hp[w][q][2 * p + 1 < S ? 2 * p + 1 : 0][0]) + trofs;
          H8[q][p] = cat8(lds_read_tr16(reinterpret_cast<const __bf16*>(h0)),
                          2 * p + 1 < S ? lds_read_tr16(reinterpret_cast<const __bf16*>(h1)) : zero4);
        }
      const char* ab = reinterpret_cast<const char*>(&bst[buf][0]) + trofs;
      auto rdA = [&](int m, bf16x8* A8) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const bf16x4 a0 = lds_read_tr16(reinterpret_cast<const __bf16*>(ab + ((2 * p) * KT + m) * 512));
          const bf16x4 a1 = 2 * p + 1 < S ? lds_read_tr16(reinterpret_cast<const __bf16*>(ab + ((2 * p + 1 < S ? 2 * p + 1 : 0) * KT + m) * 512))
                                          : zero4;
          A8[p] = cat8(a0, a1);
        }
      };
      // raw-input columns: this wave's two output tiles against the row tile's XT raw-input fragments (requested here, used
      // after the hidden k-tiles' MFMAs)
      f32x4 xr[XT];
#pragma unroll
      for (int xt = 0; xt < XT; ++xt) xr[xt] = ld4(a.X + ((size_t)tile * XT + xt) * 256 + lo);
      bf16x8 An[NP];
      rdA(0, An);
#pragma unroll
      for (int m = 0; m < KT; ++m) {
        bf16x8 A8[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) A8[p] = An[p];
        if (m + 1 < KT) rdA(m + 1, An);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            if (STPDE_FC1F_ABL != 2) dw[m][q] = mfma_bf(A8[p], H8[q][p], dw[m][q]);
      }
      // the raw-input fragments as row-major bf16 operands: through this wave's h0 patch (its blocks are in registers by now)
      bf16x4 xb4[XT];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int xt = 0; xt < XT; ++xt) *reinterpret_cast<bf16x4*>(&hp[w][0][xt][lane * 2]) = to_bf4(xr[xt]);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int xt = 0; xt < XT; ++xt)
        xb4[xt] = lds_read_tr16(reinterpret_cast<const __bf16*>(reinterpret_cast<const char*>(&hp[w][0][xt][0]) + trofs));
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const char* am = ab + (size_t)(mx + mi) * 512;
        const bf16x8 A01 = cat8(lds_read_tr16(reinterpret_cast<const __bf16*>(am)),
                                lds_read_tr16(reinterpret_cast<const __bf16*>(am + KT * 512)));
        const bf16x8 A23 = cat8(lds_read_tr16(reinterpret_cast<const __bf16*>(am + 2 * KT * 512)),
                                lds_read_tr16(reinterpret_cast<const __bf16*>(am + 3 * KT * 512)));
#pragma unroll
        for (int xt = 0; xt < XT; ++xt)
          dwx[mi][xt] = mfma_bf(A01, cat8(xb4[xt], xt == 0 ? pat0 : zero4), dwx[mi][xt]);
        dwx[mi][0] = mfma_bf(A23, pat12, dwx[mi][0]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    FSTAMP(5);
    // the staged row tile has landed (the only vector-memory wait of an iteration), every wave is done with this buffer
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FSTAMP(6);
    FC1F_BARRIER();
    FSTAMP(7);
  }

  const int g = lane >> 4, c = lane & 15;
  const int ldw = 16 * (MT + XT);
  if constexpr (DET) {
    // Deterministic mode: 1024 long-accumulator adds per lane.  Unrolled they are ~60k instructions -- the compiler gives up on
    // the unroll, indexes dw dynamically and moves the whole accumulator array to scratch (every MFMA of the launch through
    // scratch).  So: 16 blocks at a time through this wave's quarter of the (now idle) staging buffers, written with static
    // register indices, read back by a ROLLED loop.
    __syncthreads();
    float* stg = &bst[0][0] + w * 4096;
    long long* accb = reinterpret_cast<long long*>(a.dW);
#pragma unroll
    for (int m0 = 0; m0 < KT; m0 += 4) {
#pragma unroll
      for (int mm = 0; mm < 4; ++mm)
#pragma unroll
        for (int q = 0; q < NQ; ++q) *reinterpret_cast<f32x4*>(stg + ((mm * NQ + q) * 64 + lane) * 4) = dw[m0 + mm][q];
      __builtin_amdgcn_wave_barrier();
#pragma unroll 1
      for (int i = 0; i < 4 * NQ * 4; ++i) {
        const int blk = i >> 2, r = i & 3, mm = blk / NQ, q = blk % NQ;
        const float v = stg[(blk * 64 + lane) * 4 + r];
        det_add_f32(accb + ((size_t)(16 * (m0 + mm) + 4 * g + r) * ldw + 16 * (kt0 + 4 * q) + c) * STPDE_DET_K, v);
      }
      __builtin_amdgcn_wave_barrier();
    }
  } else {
#pragma unroll
    for (int m = 0; m < KT; ++m)
#pragma unroll
      for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(a.dW + (size_t)(16 * m + 4 * g + r) * ldw + 16 * (kt0 + 4 * q) + c, dw[m][q][r]);
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int xt = 0; xt < XT; ++xt)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc_add_f32(a.dW, (size_t)(16 * (mx + mi) + 4 * g + r) * ldw + 16 * (MT + xt) + c, dwx[mi][xt][r], DET ? 1 : 0);
  if (ACT == STPDE_ACT_SWISH && a.pbar) {
    const float v = wave_sum(pacc);
    if (lane == 0) atomicAdd(a.pbar + (blockIdx.x % STPDE_PBAR_SLOTS), v);
  }
}

template <int S2, int ACT>
static int launch_fused(const Fc1BwdArgs& a, hipStream_t stream) {
  int dev = 0, ncu = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  int pairs = ncu / 2;
  if (pairs > a.ntiles) pairs = a.ntiles;
  if (pairs < 1) pairs = 1;
  if (a.det)
    STPDE_LAUNCH((k_fc1_bwd_fused<S2, ACT, true>), dim3(2 * pairs), dim3(256), 0, stream, a);
  else
    STPDE_LAUNCH((k_fc1_bwd_fused<S2, ACT, false>), dim3(2 * pairs), dim3(256), 0, stream, a);
  return stpde_check_launch("k_fc1_bwd_fused");
}

template <int S2>
static int launch_fused_act(const Fc1BwdArgs& a, hipStream_t stream) {
  switch (a.cfg.act) {
    case STPDE_ACT_TANH: return launch_fused<S2, STPDE_ACT_TANH>(a, stream);
    case STPDE_ACT_RELU: return launch_fused<S2, STPDE_ACT_RELU>(a, stream);
    case STPDE_ACT_SOFTPLUS: return launch_fused<S2, STPDE_ACT_SOFTPLUS>(a, stream);
    case STPDE_ACT_ELU: return launch_fused<S2, STPDE_ACT_ELU>(a, stream);
    case STPDE_ACT_LEAKYRELU: return launch_fused<S2, STPDE_ACT_LEAKYRELU>(a, stream);
    default: return launch_fused<S2, STPDE_ACT_SWISH>(a, stream);
  }
}

// 1 when stpde_jet_fc1_bwd serves this layer description (callers fall back to stpde_jet_wgrad + stpde_jet_layer_bwd otherwise)
extern "C" int stpde_jet_fc1_bwd_supported(const stpde_layer_desc* d) {
  if (!d) return 0;
  const bool combo = d->cfg.S2 == 1 && d->cfg.combo;
  return d->first_hidden && d->mfma_bf16 == 1 && (d->packed & 6) == 6 && d->KT == 32 && d->MT == 16 && d->cfg.S1 == 3 &&
         (d->cfg.S2 == 0 || combo) && XT == 3 && d->cfg.act >= 0 && d->cfg.act <= 5;
}

extern "C" int stpde_jet_fc1_bwd(const stpde_layer_desc* d, const float* abar1, const void* WhT_pack_bf16, const float* z0,
                                 const float* tanc0, const float* cw, const float* X, float* abar0, float* abar0_tan,
                                 float* dW_aug, float* act_param_bar, void* stream) {
  if (!stpde_jet_fc1_bwd_supported(d)) {
    stpde_set_error("jet_fc1_bwd: bf16 mode with packed buffers only (first hidden layer of the reference width: KT = 32, MT = 16, "
                    "S1 = 3, S2 = 0 or the combined stream, STPDE_FC1_FUSED != 0)");
    return STPDE_E_UNSUPPORTED;
  }
  if (d->ntiles <= 0 || !abar1 || !WhT_pack_bf16 || !z0 || !tanc0 || !X || !abar0 || !abar0_tan || !dW_aug || (d->cfg.S2 && !cw)) {
    stpde_set_error("jet_fc1_bwd: bad argument");
    return STPDE_E_BADARG;
  }
  Fc1BwdArgs a{};
  a.abar1 = abar1;
  a.WT16 = WhT_pack_bf16;
  a.Z0 = z0;
  a.tanc0 = tanc0;
  a.cw = cw;
  a.X = X;
  a.abar0 = abar0;
  a.Tan0 = abar0_tan;
  a.dW = dW_aug;
  a.pbar = act_param_bar;
  a.ntiles = d->ntiles;
  a.det = d->det;
  a.cfg = d->cfg;
  int rc = d->cfg.S2 == 1 ? launch_fused_act<1>(a, (hipStream_t)stream) : launch_fused_act<0>(a, (hipStream_t)stream);
  if (rc) return rc;
  return STPDE_OK;
}
