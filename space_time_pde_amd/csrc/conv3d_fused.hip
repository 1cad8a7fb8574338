// The convolutions of one ResBlock3D (src/unet3d.py:39-56 of the reference: conv1-bn1-relu-conv2-bn2-relu-conv3-bn3 + shortcut,
// relu) with the BatchNorm work next to them folded in (round 4).  The implicit GEMM is that of conv3d.hip (a wave owns VT tiles of
// 16 consecutive voxels, A operand = packed weights, B operand = one float4 per lane from channels-last memory,
// v_mfma_f32_16x16x4_f32); what is new sits in front of and behind it:
//   ONLOAD  (1x1x1)  x is the RAW output of the previous convolution; its training-mode BatchNorm + ReLU are applied to every
//                    operand fragment as it is loaded (per-channel mean / scale / beta from an LDS table every workgroup derives
//                    from the double-format sums): the normalised activation is never written or read (conv3 of a block)
//   DUAL 1  (1x1x1)  two convolutions of the same input in one pass over it (conv1 + shortcut)
//   DUAL 2  (1x1x1)  one convolution of two inputs, y = W x + W2 x2 (the input gradient of a block: conv1's + the shortcut's,
//                    no separate add pass)
//   EPI 1            per-channel sum and sum of squares of the output from the accumulator tiles (the statistics pass of the
//                    following BatchNorm): every wave sums around ITS first voxel in fp32 (|y - shift| is of the order of the
//                    deviation), converts to plain sums in fp64, the four waves of a block meet in LDS and add one set of fp64
//                    atomics to one of STPDE_BN_REP replicas
//   EPI 2            input-gradient convolutions: the output is the gradient of relu(bn(m)); the epilogue reads m, recomputes the
//                    activation exactly as the forward pass did, stores dz = y * [act > 0] and adds sum(dz), sum(dz * xhat) to the
//                    BatchNorm-backward sums (the reduction pass of stpde_bn_bwd, and the mask read of its elementwise pass)
// Per ResBlock3D: forward 5 launches and 11 tensor passes instead of 10 and 18, backward 7 + 4 launches instead of 17 + 4.
#include <type_traits>
#include "common.h"
#include "conv_common.h"

struct FusedArgs {
  stpde_conv3d_fused_args f;
  int nvox;
};

// MC output tiles per pass, VT voxel tiles per wave, K3: 3x3x3 (else 1x1x1)
template <int MC, int VT, bool K3, int DUAL, bool ONLOAD, int EPI>
__global__ __launch_bounds__(256) void k_conv_fused(FusedArgs a) {
  static_assert(!(K3 && (DUAL || ONLOAD)), "dual / on-load variants are 1x1x1 only");
  __shared__ float tab[ONLOAD ? 3 * 512 : 1];           // mean, rstd * gamma, beta of the input channels
  __shared__ double red[EPI ? 4 * MC * 16 * 2 : 1];     // per-wave channel sums of the pass
  const stpde_conv3d_desc& d = a.f.d;
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int g = lane >> 4, j = lane & 15;
  const int lo = lane * 4;
  const int Ci = d.Ci, Co = d.Co, KT = Ci / 16, MT1 = Co / 16;
  const int Co2 = DUAL == 1 ? a.f.Co2 : 0, MT2 = Co2 / 16, MT = MT1 + MT2;
  const int KT2 = DUAL == 2 ? a.f.Ci2 / 16 : 0;
  const int ntap = K3 ? 27 : 1;
  const long N = a.nvox;
  if (ONLOAD) {
    for (int c = threadIdx.x; c < Ci; c += 256) {
      float mean, var, rstd;
      bn_stat_f64(a.f.in_sums, Ci, c, N, a.f.in_eps, a.f.d.det, mean, var, rstd);
      const float gm = a.f.in_gamma ? a.f.in_gamma[c] : 1.f, bt = a.f.in_beta ? a.f.in_beta[c] : 0.f;
      tab[c] = mean;
      tab[512 + c] = rstd * gm;
      tab[1024 + c] = bt;
      if (blockIdx.x == 0 && blockIdx.y == 0) {
        a.f.in_stat[c] = mean;
        a.f.in_stat[Ci + c] = rstd;
        if (a.f.in_running_mean) {                   // torch: running = (1 - momentum) * running + momentum * batch (unbiased)
          const float unb = N > 1 ? var * ((float)N / (float)(N - 1)) : var;
          a.f.in_running_mean[c] = (1.f - a.f.in_momentum) * a.f.in_running_mean[c] + a.f.in_momentum * mean;
          a.f.in_running_var[c] = (1.f - a.f.in_momentum) * a.f.in_running_var[c] + a.f.in_momentum * unb;
        }
      }
    }
    __syncthreads();
  }
  const int tile0 = (blockIdx.x * 4 + wv) * VT;
  const bool active = tile0 * 16 < a.nvox;              // inactive waves still meet the block at the epilogue barriers
  int v[VT];
  bool vin[VT];
  VoxN vn[VT];
#pragma unroll
  for (int t = 0; t < VT; ++t) {
    v[t] = (tile0 + t) * 16 + j;
    vin[t] = v[t] < a.nvox;
    vn[t] = vox_prepare(d, vox_coords(d, vin[t] ? v[t] : 0));
    if (!vin[t]) vn[t].ok = 0u;
  }
  const int nvalid = active ? (a.nvox - tile0 * 16 < VT * 16 ? a.nvox - tile0 * 16 : VT * 16) : 0;
  for (int mt0 = blockIdx.y * MC; mt0 < MT; mt0 += gridDim.y * MC) {
    f32x4 acc[VT][MC];
#pragma unroll
    for (int t = 0; t < VT; ++t)
#pragma unroll
      for (int mi = 0; mi < MC; ++mi) acc[t][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
    // weight blocks of this pass: tile mt < MT1 from the first pack, the others (DUAL 1) from the second output's pack
    const float* wbase[MC];
    int wstride[MC];                                    // floats between consecutive k-tiles of that pack
#pragma unroll
    for (int mi = 0; mi < MC; ++mi) {
      const int mt = mt0 + mi < MT ? mt0 + mi : mt0;
      if (DUAL == 1 && mt >= MT1) {
        wbase[mi] = a.f.wo2_pack + (size_t)(mt - MT1) * 256 + lo;
        wstride[mi] = MT2 * 256;
      } else {
        wbase[mi] = a.f.w_pack + (size_t)mt * 256 + lo;
        wstride[mi] = MT1 * 256;
      }
    }
    if (active) {
      for (int tap = 0; tap < ntap; ++tap) {
        int nb[VT];
        const float* src[VT];
        unsigned tmask;
        int toff;
        tap_uniform(d, tap, tmask, toff);
#pragma unroll
        for (int t = 0; t < VT; ++t) {
          nb[t] = vin[t] ? tap_nb(vn[t], tmask, toff) : -1;
          src[t] = a.f.x + (size_t)(nb[t] < 0 ? 0 : nb[t]) * Ci + 4 * g;
        }
        for (int kt = 0; kt < KT; ++kt) {
          f32x4 B[VT];
          f32x4 mn, sc, bt;
          if (ONLOAD) {
            mn = *reinterpret_cast<const f32x4*>(tab + 16 * kt + 4 * g);
            sc = *reinterpret_cast<const f32x4*>(tab + 512 + 16 * kt + 4 * g);
            bt = *reinterpret_cast<const f32x4*>(tab + 1024 + 16 * kt + 4 * g);
          }
#pragma unroll
          for (int t = 0; t < VT; ++t) {
            B[t] = ld4(src[t] + 16 * kt);
            if (ONLOAD) {
              B[t] = (B[t] - mn) * sc + bt;             // exactly k_bn_apply's expression (the backward recomputes this mask)
#pragma unroll
              for (int r = 0; r < 4; ++r) B[t][r] = B[t][r] > 0.f ? B[t][r] : 0.f;
            }
            if (nb[t] < 0) B[t] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
#pragma unroll
          for (int mi = 0; mi < MC; ++mi) {
            const f32x4 w = ld4(wbase[mi] + (size_t)(tap * KT + kt) * wstride[mi]);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
              for (int t = 0; t < VT; ++t) acc[t][mi] = mfma4(w[r], B[t][r], acc[t][mi]);
          }
        }
      }
      if (DUAL == 2) {
        const int C2 = a.f.Ci2;
        for (int kt = 0; kt < KT2; ++kt) {
          f32x4 B[VT];
#pragma unroll
          for (int t = 0; t < VT; ++t) {
            B[t] = ld4(a.f.x2 + (size_t)(vin[t] ? v[t] : 0) * C2 + 4 * g + 16 * kt);
            if (!vin[t]) B[t] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
#pragma unroll
          for (int mi = 0; mi < MC; ++mi) {
            const int mt = mt0 + mi < MT ? mt0 + mi : mt0;
            const f32x4 w = ld4(a.f.w2_pack + ((size_t)kt * MT1 + mt) * 256 + lo);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
              for (int t = 0; t < VT; ++t) acc[t][mi] = mfma4(w[r], B[t][r], acc[t][mi]);
          }
        }
      }
    }
    // ---- epilogue: bias, stores, channel sums ---------------------------------------------------------------
#pragma unroll
    for (int mi = 0; mi < MC; ++mi) {
      const int mt = mt0 + mi;
      const bool live = mt < MT;                         // block-uniform
      const bool second = DUAL == 1 && mt >= MT1;
      float* ybase = second ? a.f.y2 : a.f.y;
      const int ystride = second ? Co2 : Co;
      const int ch = second ? 16 * (mt - MT1) + 4 * g : 16 * mt + 4 * g;
      const float* bias = second ? a.f.bias2 : a.f.bias;
      f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
      if (live && bias) bv = ld4(bias + ch);
      f32x4 s1 = f32x4{0.f, 0.f, 0.f, 0.f}, s2 = s1, sh = s1;
      if (EPI == 1) {
        // shift of this wave and channel = the value of its first voxel (lane 16 g of tile 0)
        const f32x4 o0 = acc[0][mi] + bv;
#pragma unroll
        for (int r = 0; r < 4; ++r) sh[r] = __shfl(o0[r], lane & 48, 64);
      }
      f32x4 mmean, mrstd, mscale, mbeta;
      if (EPI == 2 && live) {
        mmean = ld4(a.f.m_stat + ch);
        mrstd = ld4(a.f.m_stat + Co + ch);
        const f32x4 one = f32x4{1.f, 1.f, 1.f, 1.f}, zero = f32x4{0.f, 0.f, 0.f, 0.f};
        mscale = mrstd * (a.f.m_gamma ? ld4(a.f.m_gamma + ch) : one);
        mbeta = a.f.m_beta ? ld4(a.f.m_beta + ch) : zero;
      }
#pragma unroll
      for (int t = 0; t < VT; ++t) {
        if (!live || !vin[t] || !active) continue;
        f32x4 o = acc[t][mi] + bv;
        float* yp = ybase + (size_t)v[t] * ystride + ch;
        if (EPI == 1 && !second) {
          const f32x4 dd = o - sh;
          s1 += dd;
          s2 += dd * dd;
        }
        if (EPI == 2) {
          const f32x4 xm = ld4(a.f.m + (size_t)v[t] * Co + ch) - mmean;
          const f32x4 h = xm * mscale + mbeta;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (!(h[r] > 0.f)) o[r] = 0.f;
          s1 += o;
          s2 += o * xm * mrstd;
        }
        st4(yp, o);
      }
      if (EPI) {
        s1 = row_sum16x4(s1);
        s2 = row_sum16x4(s2);
        if (j == 15) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            double t1 = s1[r], t2 = s2[r];
            if (EPI == 1) {                              // sums around the wave's shift -> plain sums, in fp64
              const double s = sh[r], n = nvalid;
              t2 = t2 + 2. * s * t1 + n * s * s;
              t1 = t1 + n * s;
            }
            if (!active || (EPI == 1 && second)) t1 = t2 = 0.;
            red[((wv * MC + mi) * 16 + 4 * g + r) * 2] = t1;
            red[((wv * MC + mi) * 16 + 4 * g + r) * 2 + 1] = t2;
          }
        }
      }
    }
    if (EPI) {
      __syncthreads();
      if (threadIdx.x < MC * 16) {
        const int mi = threadIdx.x >> 4, c = threadIdx.x & 15, mt = mt0 + mi;
        if (mt < MT1) {
          double t1 = 0., t2 = 0.;
          for (int w = 0; w < 4; ++w) {
            t1 += red[((w * MC + mi) * 16 + c) * 2];
            t2 += red[((w * MC + mi) * 16 + c) * 2 + 1];
          }
          const int rep = blockIdx.x % STPDE_BN_REP;
          if (a.f.d.det) {      // order-independent long accumulators, one replica [2][Co][STPDE_DET_K] (common.h)
            long long* acc = reinterpret_cast<long long*>(EPI == 1 ? (void*)a.f.out_sums : (void*)a.f.m_bsum);
            if (EPI == 1) {
              det_add_f64(acc + (size_t)(16 * mt + c) * STPDE_DET_K, t1);
              det_add_f64(acc + ((size_t)Co + 16 * mt + c) * STPDE_DET_K, t2);
            } else {
              det_add_f32(acc + (size_t)(16 * mt + c) * STPDE_DET_K, (float)t1);
              det_add_f32(acc + ((size_t)Co + 16 * mt + c) * STPDE_DET_K, (float)t2);
            }
          } else if (EPI == 1) {
            atomicAdd(a.f.out_sums + (size_t)(2 * rep) * Co + 16 * mt + c, t1);
            atomicAdd(a.f.out_sums + (size_t)(2 * rep + 1) * Co + 16 * mt + c, t2);
          } else {
            atomicAdd(a.f.m_bsum + (size_t)(2 * rep) * Co + 16 * mt + c, (float)t1);
            atomicAdd(a.f.m_bsum + (size_t)(2 * rep + 1) * Co + 16 * mt + c, (float)t2);
          }
        }
      }
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// 3x3x3 convolutions of the wide levels from LDS halo tiles (round 5).  k_conv_fused above fetches every B fragment of every
// tap from global memory: a voxel's channels are read 27 times through the 32 KB L1 (TA-bound: 74 TFLOP/s on 16 -> 16 channels,
// 100 on 32 -> 32).  Here a persistent workgroup of 4 waves owns a block of output voxels at a time (KT = channel tiles of the
// square convolution: 2 x 4 x 32 voxels for 16 channels, 2 x 2 x 32 for 32, 2 x 2 x 16 for 64); the block's input WITH its
// one-voxel halo goes to LDS once (zero outside the volume), 16-byte loads software-pipelined through registers so that block
// n + 1 is in flight during the MFMAs of block n, and all 27 taps read their B fragments from there with one ds_read_b128 per
// tile and tap at an immediate offset (voxel rows padded by 4 floats: at most 2-way bank conflicts).  A wave owns 4 voxel
// tiles x ONE output tile (wave = voxel group x output tile); weight fragments come from L1 / L2 as above (27 of them in
// registers for the whole launch was tried for 16 channels: 108 registers next to the staging registers spill).
// Same MFMA order per output element as k_conv_fused (tap, k-tile, k-step): results are bit-identical.
// EPI 1 / 2 (statistics, mask + BatchNorm-backward sums): per block as above, but the per-wave sums stay in fp64 registers
// across the blocks of the launch and meet in LDS once at the end: one set of atomics per workgroup and launch.
// ------------------------------------------------------------------------------------------------------------------------
template <int KT, int EPI>
__global__ __launch_bounds__(256, 2) void k_conv3_lds(FusedArgs a) {
  constexpr int TT = 2, TZ = KT == 1 ? 4 : 2, TX = KT == 4 ? 16 : 32;
  constexpr int HZ = TZ + 2, HX = TX + 2, NH = (TT + 2) * HZ * HX, NV = TT * TZ * TX, NXH = TX / 16;
  constexpr int C = 16 * KT, SV = C + 4, Q = C / 4;
  static_assert(NV == 64 * (4 / KT), "a block = (4 / KT) voxel groups of 4 tiles");
  __shared__ __attribute__((aligned(16))) float xs[NH * SV];
  __shared__ double red[EPI ? 4 * 16 * 2 : 1];
  const stpde_conv3d_desc& d = a.f.d;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, j = lane & 15;
  const int mt = wv % KT, vg = wv / KT;
  const int T = d.T, Z = d.Z, X = d.X;
  const int nbx = X / TX, nbz = Z / TZ, nbt = T / TT;
  const int nblk = d.B * nbt * nbz * nbx;
  const int ch = 16 * mt + 4 * g;                        // this lane's 4 output channels
  // tile t of this wave inside the block, and the LDS address of its lane's centre voxel (tap offsets are immediates)
  int tt_[4], zz_[4], xo_[4];
  const float* xc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int ti = vg * 4 + t;
    tt_[t] = ti / (NXH * TZ);
    zz_[t] = (ti / NXH) % TZ;
    xo_[t] = 16 * (ti % NXH);
    xc[t] = xs + ((tt_[t] * HZ + zz_[t]) * HX + xo_[t] + j) * SV + 4 * g;     // tap (-1, -1, -1) of this lane's voxel
  }
  const float* wp = a.f.w_pack + (size_t)mt * 256 + lane * 4;
  f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
  if (a.f.bias) bv = ld4(a.f.bias + ch);
  f32x4 mmean, mrstd, mscale, mbeta;
  if constexpr (EPI == 2) {
    mmean = ld4(a.f.m_stat + ch);
    mrstd = ld4(a.f.m_stat + C + ch);
    mscale = a.f.m_gamma ? mrstd * ld4(a.f.m_gamma + ch) : mrstd;
    mbeta = a.f.m_beta ? ld4(a.f.m_beta + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // lanes j == 15: this wave's channel sums over its blocks, in fp64 for both epilogues (per-flush partial sums of at most
  // FLUSH blocks are fp32 row sums, everything above them is fp64 until the final atomic; ADVICE r5: the fp32 running sum of
  // the mask epilogue made the BatchNorm-backward sums depend on how many blocks a persistent workgroup processed)
  using SumT = double;
  SumT sd1[4] = {0, 0, 0, 0}, sd2[4] = {0, 0, 0, 0};
  constexpr int FLUSH = 16;
  f32x4 s1 = f32x4{0.f, 0.f, 0.f, 0.f}, s2 = s1, sh = s1;
  int nacc = 0;
  bool have_sh = false;
  auto flush = [&]() {
    const f32x4 r1 = row_sum16x4(s1), r2 = row_sum16x4(s2);      // (hazard-safe DPP row sums, common.h)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      SumT t1 = r1[rr], t2 = r2[rr];
      if constexpr (EPI == 1) {                          // sums around the wave's shift -> plain sums, in fp64
        const double s = sh[rr], n = 64. * nacc;
        t2 = t2 + 2. * s * t1 + n * s * s;
        t1 = t1 + n * s;
      }
      sd1[rr] += t1;
      sd2[rr] += t2;
    }
    s1 = s2 = f32x4{0.f, 0.f, 0.f, 0.f};
    nacc = 0;
  };
  // output (and mask operand) through buffer descriptors: 32-bit byte offsets per lane
  const unsigned ybytes = (unsigned)((size_t)d.B * T * Z * X * C * 4);
  const auto yr = opt_store_rsrc(a.f.y, ybytes);
  const auto mr = load_rsrc(EPI == 2 ? (const void*)a.f.m : (const void*)a.f.x, ybytes);

  // Staging by halo ROWS (fixed t, z: HX voxels x C channels, contiguous in memory): a wave owns RPW rows, a lane the quads
  // lane + 64 k of a row.  Row validity and the row's base are wave-uniform (scalar), what a lane adds -- its byte offset in
  // the row, its LDS address, whether its voxel is the row's first / last one (outside the volume on the x faces) -- does
  // not depend on the block and is computed once.  A request costs ~5 vector instructions (the voxel-indexed version spent
  // ~60, 64-bit multiplies among them, per 16-byte load: 1000 instructions per block and wave in front of the MFMA loop);
  // invalid requests go to the buffer descriptor's out-of-range offset and return zeros.
  constexpr int NR = (TT + 2) * HZ, RPW = NR / 4, RQ = HX * Q, NK = (RQ + 63) / 64;
  static_assert(NR % 4 == 0, "halo rows: whole rounds of the 4 waves");
  const auto xr = load_rsrc(a.f.x, (unsigned)((size_t)d.B * T * Z * X * C * 4));
  int loff[NK];                       // lane's byte offset relative to voxel x0 of the row (negative for the x0 - 1 halo voxel)
  const float* lds_k[NK];             // where it goes: row wv * RPW of the tile, (+ i rows: immediate)
  unsigned f_first = 0u, f_last = 0u, f_none = 0u;
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int kq = lane + 64 * k, hx = kq / Q, q = kq % Q;
    loff[k] = (kq - Q) * 16;
    lds_k[k] = xs + ((wv * RPW) * HX + hx) * SV + 4 * q;
    f_first |= (hx == 0 ? 1u : 0u) << k;
    f_last |= (hx == HX - 1 ? 1u : 0u) << k;
    f_none |= (kq >= RQ ? 1u : 0u) << k;
  }
  f32x4 px[RPW][NK];
  auto fetch = [&](int bi) {
    int r = bi;
    const int x0 = (r % nbx) * TX;
    r /= nbx;
    const int z0 = (r % nbz) * TZ;
    r /= nbz;
    const int t0 = (r % nbt) * TT;
    const int b = r / nbt;
    const unsigned bad = f_none | (x0 == 0 ? f_first : 0u) | (x0 + TX == X ? f_last : 0u);
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int row = wv * RPW + i;                      // wave-uniform
      const int t = t0 + row / HZ - 1, z = z0 + row % HZ - 1;
      const bool rowok = t >= 0 && t < T && z >= 0 && z < Z;
      const int rbase = ((((b * T + t) * Z + z) * X + x0) * C) * 4;
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        const bool ok = rowok && !((bad >> k) & 1u);
        px[i][k] = __builtin_bit_cast(f32x4, buf_ld16(xr, ok ? rbase + loff[k] : (int)0x80000000, 0));
      }
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < RPW; ++i)
#pragma unroll
      for (int k = 0; k < NK; ++k)
        if (!((f_none >> k) & 1u)) st4(const_cast<float*>(lds_k[k]) + i * HX * SV, px[i][k]);
  };
  if ((int)blockIdx.x < nblk) fetch(blockIdx.x);
#pragma unroll 1
  for (int bi = blockIdx.x; bi < nblk; bi += gridDim.x) {
    __syncthreads();                                     // the readers of the previous block are done
    stage();
    __syncthreads();
    if (bi + (int)gridDim.x < nblk) fetch(bi + gridDim.x);
    int r = bi;
    const int x0 = (r % nbx) * TX;
    r /= nbx;
    const int z0 = (r % nbz) * TZ;
    r /= nbz;
    const int t0 = (r % nbt) * TT;
    const int b = r / nbt;
    int vo[4];                                           // byte offset of this lane's output voxel / channels
#pragma unroll
    for (int t = 0; t < 4; ++t)
      vo[t] = (((((b * T + t0 + tt_[t]) * Z + z0 + zz_[t]) * X + x0 + xo_[t] + j) * C) + ch) * 4;
    // mask epilogue: the pre-activation values it needs are requested behind the MFMAs (in front of them: 16 more live
    // registers than the staging registers leave room for)
    constexpr bool MV_EARLY = false;
    f32x4 mv[EPI == 2 ? 4 : 1];
    if constexpr (EPI == 2 && MV_EARLY) {
#pragma unroll
      for (int t = 0; t < 4; ++t) mv[t] = __builtin_bit_cast(f32x4, buf_ld16(mr, vo[t], 0));
    }
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      // 9 (dt, dz) rows at run time, the NS = 3 dx taps x KT k-tiles of a row unrolled (immediate offsets).  Explicit software
      // pipeline across the steps AND the rows: the B fragments of step s + 1 and the weight fragment of step s + 2 are
      // requested in front of the 16 MFMAs of step s (left to the scheduler inside one row, the first steps of every row
      // waited for their LDS / L2 round trips)
      constexpr int NS = 3 * KT;
      auto ldB = [&](f32x4* B, int roff, int s) {
#pragma unroll
        for (int t = 0; t < 4; ++t) B[t] = ld4(xc[t] + roff + (s / KT) * SV + 16 * (s % KT));
      };
      auto roff_of = [&](int row) { return ((row / 3) * HZ + row % 3) * HX * SV; };
      f32x4 Bc[4], Bn[4], wc, wn, wnn;
      ldB(Bc, roff_of(0), 0);
      wc = ld4(wp);
      wn = ld4(wp + (size_t)1 * (KT * 256));
#pragma unroll 1
      for (int row = 0; row < 9; ++row) {
        const int roff = roff_of(row);
        const int rown = row < 8 ? row + 1 : 8;        // (the last row requests its own fragments again: branch-free)
        const int roffn = roff_of(rown);
        const float* wrow = wp + (size_t)(row * NS) * (KT * 256);
        const float* wrown = wp + (size_t)(rown * NS) * (KT * 256);
#pragma unroll
        for (int sI = 0; sI < NS; ++sI) {
          if (sI + 1 < NS)
            ldB(Bn, roff, sI + 1);
          else
            ldB(Bn, roffn, 0);
          if (sI + 2 < NS)
            wnn = ld4(wrow + (size_t)(sI + 2) * (KT * 256));
          else
            wnn = ld4(wrown + (size_t)(sI + 2 - NS) * (KT * 256));
          __builtin_amdgcn_sched_barrier(0);             // (else the scheduler sinks the requests below the MFMAs to reuse
#pragma unroll                                           //  the registers of the current fragments: no prefetch at all)
          for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma4(wc[rr], Bc[t][rr], acc[t]);
#pragma unroll
          for (int t = 0; t < 4; ++t) Bc[t] = Bn[t];
          wc = wn;
          wn = wnn;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    // ---- epilogue of the block ----------------------------------------------------------------------------
    // (statistics / mask sums: per LANE in fp32 across FLUSH blocks -- 4 * FLUSH terms, around a shift the wave takes from its
    //  first block -- then the 16-lane row sums and the fp64 arithmetic once per FLUSH blocks, not per block: the epilogue
    //  runs while the matrix pipe of this wave idles)
    if constexpr (EPI == 2 && !MV_EARLY) {
#pragma unroll
      for (int t = 0; t < 4; ++t) mv[t] = __builtin_bit_cast(f32x4, buf_ld16(mr, vo[t], 0));
    }
    if constexpr (EPI == 1) {
      if (!have_sh) {
        const f32x4 o0 = acc[0] + bv;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) sh[rr] = __shfl(o0[rr], lane & 48, 64);
        have_sh = true;
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x4 o = acc[t] + bv;
      if constexpr (EPI == 1) {
        const f32x4 dd = o - sh;
        s1 += dd;
        s2 += dd * dd;
      }
      if constexpr (EPI == 2) {
        const f32x4 xm = mv[t] - mmean;
        const f32x4 h = xm * mscale + mbeta;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
          if (!(h[rr] > 0.f)) o[rr] = 0.f;
        s1 += o;
        s2 += o * xm * mrstd;
      }
      buf_st16(yr, vo[t], 0, o);
    }
    if constexpr (EPI != 0) {
      if (++nacc == FLUSH) flush();
    }
  }
  if constexpr (EPI != 0) {
    if (nacc) flush();
  }
  if constexpr (EPI != 0) {
    if (j == 15) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        red[(wv * 16 + 4 * g + rr) * 2] = sd1[rr];
        red[(wv * 16 + 4 * g + rr) * 2 + 1] = sd2[rr];
      }
    }
    __syncthreads();
    if (threadIdx.x < KT * 16) {                         // the (4 / KT) waves that share an output tile: waves mt, mt + KT, ...
      const int m = threadIdx.x >> 4, c = threadIdx.x & 15;
      double t1 = 0., t2 = 0.;
      for (int w = m; w < 4; w += KT) {
        t1 += red[(w * 16 + c) * 2];
        t2 += red[(w * 16 + c) * 2 + 1];
      }
      const int rep = blockIdx.x % STPDE_BN_REP;
      if (a.f.d.det) {
        long long* acc = reinterpret_cast<long long*>(EPI == 1 ? (void*)a.f.out_sums : (void*)a.f.m_bsum);
        if (EPI == 1) {
          det_add_f64(acc + (size_t)(16 * m + c) * STPDE_DET_K, t1);
          det_add_f64(acc + ((size_t)C + 16 * m + c) * STPDE_DET_K, t2);
        } else {
          det_add_f32(acc + (size_t)(16 * m + c) * STPDE_DET_K, (float)t1);
          det_add_f32(acc + ((size_t)C + 16 * m + c) * STPDE_DET_K, (float)t2);
        }
      } else if (EPI == 1) {
        atomicAdd(a.f.out_sums + (size_t)(2 * rep) * C + 16 * m + c, t1);
        atomicAdd(a.f.out_sums + (size_t)(2 * rep + 1) * C + 16 * m + c, t2);
      } else {
        atomicAdd(a.f.m_bsum + (size_t)(2 * rep) * C + 16 * m + c, (float)t1);
        atomicAdd(a.f.m_bsum + (size_t)(2 * rep + 1) * C + 16 * m + c, (float)t2);
      }
    }
  }
}

// the LDS-tile kernel serves square 16 / 32 / 64-channel 3x3x3 convolutions on volumes made of whole blocks, enough of them
template <int EPI>
static bool launch_conv3_lds(const FusedArgs& a0, hipStream_t st) {
  // (test overrides through stpde_tune, read per call: tests/test_gpu_conv_fused.py compares the two kernels in one run)
  const stpde_conv3d_desc& d = a0.f.d;
  if (stpde_tune_get(STPDE_TUNE_CONV3_LDS_OFF) || d.ksize != 3 || d.Ci != d.Co || (d.Ci != 16 && d.Ci != 32 && d.Ci != 64)) return false;
  const int KT = d.Ci / 16;
  const int TZ = KT == 1 ? 4 : 2, TX = KT == 4 ? 16 : 32;
  if (d.T % 2 || d.Z % TZ || d.X % TX) return false;
  if ((size_t)d.B * d.T * d.Z * d.X * d.Ci * 4 >= (1u << 31)) return false;      // 32-bit byte offsets into x / y / m
  const int nblk = d.B * (d.T / 2) * (d.Z / TZ) * (d.X / TX);
  const int minblk_t = stpde_tune_get(STPDE_TUNE_CONV3_LDS_MINBLK);
  if (nblk < (minblk_t > 0 ? minblk_t : 1024)) return false;
  const int gx_env = stpde_tune_get(STPDE_TUNE_CONV3_LDS_GX);
  int gx = gx_env > 0 ? gx_env : 512;                    // two persistent workgroups (64 - 77 KB of LDS) per CU
  if (gx > nblk) gx = nblk;
  const FusedArgs& a = a0;
#define STPDE_C3L(K)                                                                  \
  if (KT == K) {                                                                     \
    if constexpr (EPI == 0) STPDE_LAUNCH((k_conv3_lds<K, 0>), dim3(gx), dim3(256), 0, st, a); \
    if constexpr (EPI == 1) STPDE_LAUNCH((k_conv3_lds<K, 1>), dim3(gx), dim3(256), 0, st, a); \
    if constexpr (EPI == 2) STPDE_LAUNCH((k_conv3_lds<K, 2>), dim3(gx), dim3(256), 0, st, a); \
  }
  STPDE_C3L(1) STPDE_C3L(2) STPDE_C3L(4)
#undef STPDE_C3L
  return true;
}

template <bool K3, int DUAL, bool ONLOAD, int EPI>
static void launch_fused(const FusedArgs& a, int MT, bool big, dim3 grid, hipStream_t st) {
  if (big) {
    if (MT == 1)
      STPDE_LAUNCH((k_conv_fused<1, 4, K3, DUAL, ONLOAD, EPI>), grid, dim3(256), 0, st, a);
    else if (MT == 2)
      STPDE_LAUNCH((k_conv_fused<2, 4, K3, DUAL, ONLOAD, EPI>), grid, dim3(256), 0, st, a);
    else
      STPDE_LAUNCH((k_conv_fused<4, 4, K3, DUAL, ONLOAD, EPI>), grid, dim3(256), 0, st, a);
  } else {
    if (MT == 1)
      STPDE_LAUNCH((k_conv_fused<1, 1, K3, DUAL, ONLOAD, EPI>), grid, dim3(256), 0, st, a);
    else if (MT == 2)
      STPDE_LAUNCH((k_conv_fused<2, 1, K3, DUAL, ONLOAD, EPI>), grid, dim3(256), 0, st, a);
    else
      STPDE_LAUNCH((k_conv_fused<4, 1, K3, DUAL, ONLOAD, EPI>), grid, dim3(256), 0, st, a);
  }
}

extern "C" int stpde_conv3d_fused(const stpde_conv3d_fused_args* f, int* epilogue_done, void* stream) {
  if (!f) {
    stpde_set_error("conv3d_fused: null arguments");
    return STPDE_E_BADARG;
  }
  const stpde_conv3d_desc* d = &f->d;
  if (d->B < 1 || d->T < 1 || d->Z < 1 || d->X < 1 || d->Ci < 16 || d->Co < 16 || (d->Ci & 15) || (d->Co & 15) ||
      (d->ksize != 1 && d->ksize != 3) || (size_t)d->B * d->T * d->Z * d->X >= (1u << 31) / 16) {
    stpde_set_error("conv3d_fused: bad descriptor (channels must be multiples of 16, ksize 1 or 3)");
    return STPDE_E_BADARG;
  }
  const bool dual_out = f->y2 != nullptr, dual_in = f->x2 != nullptr, onload = f->in_sums != nullptr;
  const bool stats = f->out_sums != nullptr, mask = f->m != nullptr;
  if (!f->x || !f->w_pack || !f->y || (dual_out && (!f->wo2_pack || f->Co2 < 16 || (f->Co2 & 15))) ||
      (dual_in && (!f->w2_pack || f->Ci2 < 16 || (f->Ci2 & 15))) || (onload && (!f->in_stat || d->Ci > 512)) ||
      (mask && (!f->m_stat || !f->m_bsum)) || (d->ksize == 3 && (dual_out || dual_in || onload)) ||
      (dual_out + dual_in + onload + mask > 1) || (stats && (dual_in || mask))) {
    stpde_set_error("conv3d_fused: inconsistent arguments");
    return STPDE_E_BADARG;
  }
  if (epilogue_done) *epilogue_done = 1;
  FusedArgs a{};
  a.f = *f;
  a.nvox = d->B * d->T * d->Z * d->X;
  const hipStream_t st = (hipStream_t)stream;
  const int ntiles = (a.nvox + 15) / 16;
  const int MT = d->Co / 16 + (dual_out ? f->Co2 / 16 : 0);
  const int nchunks = (MT + 3) / 4;
  const bool big = ntiles >= 16384;
  dim3 grid;
  if (big) {
    grid = dim3((ntiles + 15) / 16, 1);
  } else {
    const int gx = (ntiles + 3) / 4;
    const int gy = gx >= 1024 ? 1 : nchunks;
    if (d->ksize == 3 && gx * gy < 256 && !d->det) {
      // deep levels: the tap-split kernel of conv3d.hip fills the chip (not in deterministic mode: partial outputs by atomics); its output holds partial sums until the last atomic,
      // so the statistics take a pass of their own and a mask epilogue is left to the caller (stpde_bn_bwd)
      int rc = stpde_conv3d_fwd(d, f->x, f->w_pack, f->bias, f->y, stream);
      if (rc) return rc;
      if (stats) rc = stpde_bn_stats_f64(f->y, a.nvox, d->Co, f->out_sums, st);
      if (mask && epilogue_done) *epilogue_done = 0;
      if (mask && !epilogue_done) {
        stpde_set_error("conv3d_fused: tap-split volume needs epilogue_done");
        return STPDE_E_BADARG;
      }
      return rc;
    }
    grid = dim3(gx, gy);
  }
  if (d->ksize == 3) {
    if (stats ? launch_conv3_lds<1>(a, st) : (mask ? launch_conv3_lds<2>(a, st) : launch_conv3_lds<0>(a, st)))
      return stpde_check_launch("k_conv3_lds");
    if (stats)
      launch_fused<true, 0, false, 1>(a, MT, big, grid, st);
    else if (mask)
      launch_fused<true, 0, false, 2>(a, MT, big, grid, st);
    else
      launch_fused<true, 0, false, 0>(a, MT, big, grid, st);
  } else if (dual_out) {
    if (stats)
      launch_fused<false, 1, false, 1>(a, MT, big, grid, st);
    else
      launch_fused<false, 1, false, 0>(a, MT, big, grid, st);
  } else if (dual_in) {
    launch_fused<false, 2, false, 0>(a, MT, big, grid, st);
  } else if (onload) {
    if (stats)
      launch_fused<false, 0, true, 1>(a, MT, big, grid, st);
    else
      launch_fused<false, 0, true, 0>(a, MT, big, grid, st);
  } else if (mask) {
    launch_fused<false, 0, false, 2>(a, MT, big, grid, st);
  } else if (stats) {
    launch_fused<false, 0, false, 1>(a, MT, big, grid, st);
  } else {
    launch_fused<false, 0, false, 0>(a, MT, big, grid, st);
  }
  return stpde_check_launch("k_conv_fused");
}
