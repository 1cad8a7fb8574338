// 3-D convolutions of the U-Net encoder (src/unet3d.py:39-56 of the reference: Conv3d 1x1x1 and 3x3x3, padding 1,
// stride 1) as implicit GEMMs on v_mfma_f32_16x16x4_f32 over channels-last activations [B][T][Z][X][C].
//
// Forward / input-gradient: each wave owns 16 consecutive voxels and walks all output-channel tiles:
//   y^T[16co x 16 vox] += W_tap[16co x 16ci] * x_tap^T[16ci x 16 vox]
// A operand = packed weights (float4 per lane per (tap, ci-tile, co-tile)), B operand = one float4 per lane straight
// from channels-last memory (lane 16g+j: channels 4g..4g+3 of voxel j, zero outside the volume), D = float4 store
// back to channels-last memory.  No im2col buffer, no LDS.
// Weight gradient: contraction over voxels, operands are dword loads (lane 16k+i: voxel 4s+k, channel i).
#include "common.h"
#include "conv_common.h"

// VT voxel tiles (16 voxels each) per wave: every weight block that is loaded feeds VT*4 MFMAs per output tile.
// SPLIT (small volumes = the deep U-Net levels, where a handful of waves would otherwise walk 27 taps x all channel
// tiles serially): blockIdx.z owns a slice of the taps and adds its partial sums to the zero-filled output with fp32
// atomics; taps whose 16 neighbours all fall outside the volume are skipped.
template <int MC, int VT, bool SPLIT = false>
__global__ __launch_bounds__(256) void k_conv3d_fwd(ConvArgs a) {
  const int lane = threadIdx.x & 63;
  const int tile0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * VT;
  if (tile0 * 16 >= a.nvox) return;
  const int g = lane >> 4, j = lane & 15;
  const int lo = lane * 4;
  const int Ci = a.d.Ci, Co = a.d.Co, KT = Ci / 16, MT = Co / 16;
  const int ntap = a.d.ksize == 3 ? 27 : 1;
  int v[VT];
  bool vin[VT];
  VoxN vn[VT];
#pragma unroll
  for (int t = 0; t < VT; ++t) {
    v[t] = (tile0 + t) * 16 + j;
    vin[t] = v[t] < a.nvox;
    vn[t] = vox_prepare(a.d, vox_coords(a.d, vin[t] ? v[t] : 0));
    if (!vin[t]) vn[t].ok = 0u;          // voxels past the end: no tap is valid (ksize 1 handled below)
  }
  // large volumes: one wave walks all output-channel chunks of its voxels (inputs stay hot in L1);
  // small volumes (deep U-Net levels): the chunks are spread over blockIdx.y so that the chip is not idle
  for (int mt0 = blockIdx.y * MC; mt0 < MT; mt0 += gridDim.y * MC) {
    f32x4 acc[VT][MC];
#pragma unroll
    for (int t = 0; t < VT; ++t)
#pragma unroll
      for (int mi = 0; mi < MC; ++mi) acc[t][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int tps = SPLIT ? (ntap + (int)gridDim.z - 1) / (int)gridDim.z : ntap;
    const int tap_lo = SPLIT ? (int)blockIdx.z * tps : 0;
    const int tap_hi = tap_lo + tps < ntap ? tap_lo + tps : ntap;
    for (int tap = tap_lo; tap < tap_hi; ++tap) {
      int nb[VT];
      const float* src[VT];
      unsigned tmask;
      int toff;
      tap_uniform(a.d, tap, tmask, toff);
#pragma unroll
      for (int t = 0; t < VT; ++t) {
        nb[t] = vin[t] ? tap_nb(vn[t], tmask, toff) : -1;
        src[t] = a.x + (size_t)(nb[t] < 0 ? 0 : nb[t]) * Ci + 4 * g;
      }
      if (SPLIT && VT == 1 && __ballot(nb[0] >= 0) == 0) continue;   // wave-uniform: nothing to add for this tap
      for (int kt = 0; kt < KT; ++kt) {
        f32x4 B[VT];
#pragma unroll
        for (int t = 0; t < VT; ++t) {
          B[t] = ld4(src[t] + 16 * kt);
          if (nb[t] < 0) B[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const float* wp = a.w + ((size_t)(tap * KT + kt) * MT + mt0) * 256 + lo;
#pragma unroll
        for (int mi = 0; mi < MC; ++mi) {
          const int mic = mt0 + mi < MT ? mi : 0;
          f32x4 w = ld4(wp + (size_t)mic * 256);
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < VT; ++t) acc[t][mi] = mfma4(w[r], B[t][r], acc[t][mi]);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < VT; ++t) {
      if (!vin[t]) continue;
#pragma unroll
      for (int mi = 0; mi < MC; ++mi) {
        const int mt = mt0 + mi;
        if (mt >= MT) continue;
        f32x4 o = acc[t][mi];
        float* yp = a.y + (size_t)v[t] * Co + 16 * mt + 4 * g;
        if (SPLIT) {
          if (a.bias && blockIdx.z == 0) o += ld4(a.bias + 16 * mt + 4 * g);
#pragma unroll
          for (int r = 0; r < 4; ++r) atomicAdd(yp + r, o[r]);
        } else {
          if (a.bias) o += ld4(a.bias + 16 * mt + 4 * g);
          st4(yp, o);
        }
      }
    }
  }
}

// Weight gradient: a wave owns (tap group of TG taps, MCW co-tiles, KCW ci-tiles) and strides over voxel tiles; the
// ybar fragment of a tile is loaded once and reused for the TG taps (the shifted x fragments mostly hit L1), and the
// voxel coordinates are decoded once per tile.  The four waves of a block are summed through LDS before ONE set of
// fp32 atomics per block, so at most gridDim.x atomics hit any dW address.
// ONLOAD (round 4, 1x1x1): x is the raw output of the previous convolution of a residual block and the convolution consumed
// max(0, bn(x)) applied on load (k_conv_fused); the same per-channel transform on the operand dwords here.
template <int MCW, int KCW, int TG, bool ONLOAD = false>
__global__ __launch_bounds__(256) void k_conv3d_wgrad(ConvArgs a) {
  __shared__ float red[4][256];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int kk = lane >> 4, i = lane & 15;
  const int Ci = a.d.Ci, Co = a.d.Co, KT = Ci / 16, MT = Co / 16;
  const int nmb = (MT + MCW - 1) / MCW, nkb = (KT + KCW - 1) / KCW;
  int id = blockIdx.y;
  const int kb = id % nkb;
  id /= nkb;
  const int mb = id % nmb;
  const int tap0 = (id / nmb) * TG;
  f32x4 acc[TG][MCW][KCW];
#pragma unroll
  for (int tg = 0; tg < TG; ++tg)
#pragma unroll
    for (int mi = 0; mi < MCW; ++mi)
#pragma unroll
      for (int ki = 0; ki < KCW; ++ki) acc[tg][mi][ki] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ntiles = (a.nvox + 15) / 16;
  // Operand fragments of one voxel tile: 4 k-steps (4 voxels each) x (MCW ybar + TG * KCW shifted-x dwords).
  struct Frag {
    float pa[4][MCW];
    float qb[4][TG][KCW];
  };
  unsigned tmask[TG];
  int toff[TG];
#pragma unroll
  for (int tg = 0; tg < TG; ++tg) tap_uniform(a.d, tap0 + tg, tmask[tg], toff[tg]);
  // bias gradient = column sums of ybar: taken from the ybar fragments by the blocks of tap group 0 / ci block 0 (every
  // element of ybar passes through exactly one of them) -- saves the separate reduction pass over ybar
  float omean[KCW], oscale[KCW], obeta[KCW];
  if (ONLOAD) {
#pragma unroll
    for (int ki = 0; ki < KCW; ++ki) {
      const int ch = 16 * (kb * KCW + ki) + i;
      const bool okc = kb * KCW + ki < KT;
      omean[ki] = okc ? a.in_stat[ch] : 0.f;
      oscale[ki] = okc ? a.in_stat[Ci + ch] * (a.in_gamma ? a.in_gamma[ch] : 1.f) : 0.f;
      obeta[ki] = okc && a.in_beta ? a.in_beta[ch] : 0.f;
    }
  }
  const bool dob = a.dbias && kb == 0 && tap0 == 0;
  float bs[MCW];
#pragma unroll
  for (int mi = 0; mi < MCW; ++mi) bs[mi] = 0.f;
  auto load = [&](int tile, Frag& f) {
    // coordinates of this lane's voxel of k-step 0 by division, of the following k-steps (+4 voxels each) by carry
    const int v0 = tile * 16 + kk;
    Vox c = vox_coords(a.d, v0 < a.nvox ? v0 : 0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int v = v0 + 4 * s;
      const bool vin = v < a.nvox;
      if (s > 0) {
        c.x += 4;
        while (c.x >= a.d.X) {          // X may be smaller than 4 on the deepest levels
          c.x -= a.d.X;
          if (++c.z >= a.d.Z) {
            c.z = 0;
            if (++c.t >= a.d.T) {
              c.t = 0;
              ++c.b;
            }
          }
        }
      }
      const VoxN vn = vox_prepare(a.d, c);
#pragma unroll
      for (int mi = 0; mi < MCW; ++mi) {
        const int mt = mb * MCW + mi;
        f.pa[s][mi] = (vin && mt < MT) ? a.ybar[(size_t)v * Co + 16 * mt + i] : 0.f;
      }
#pragma unroll
      for (int tg = 0; tg < TG; ++tg) {
        const int nb = vin ? tap_nb(vn, tmask[tg], toff[tg]) : -1;
#pragma unroll
        for (int ki = 0; ki < KCW; ++ki) {
          const int kt = kb * KCW + ki;
          f.qb[s][tg][ki] = (nb >= 0 && kt < KT) ? a.x[(size_t)nb * Ci + 16 * kt + i] : 0.f;
          if (ONLOAD) {
            const float h = (f.qb[s][tg][ki] - omean[ki]) * oscale[ki] + obeta[ki];
            f.qb[s][tg][ki] = (nb >= 0 && kt < KT && h > 0.f) ? h : 0.f;
          }
        }
      }
    }
  };
  // (prefetching the next tile's fragments while the MFMAs of the current one run was measured: no gain, one wave per
  // SIMD less -- the kernel is bound by its index / address arithmetic and, for 32 channels, by re-reading the activations
  // once per tap group)
#pragma unroll 1
  for (int tile = blockIdx.x * 4 + wv; tile < ntiles; tile += gridDim.x * 4) {
    Frag cur;
    load(tile, cur);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int tg = 0; tg < TG; ++tg)
#pragma unroll
        for (int mi = 0; mi < MCW; ++mi)
#pragma unroll
          for (int ki = 0; ki < KCW; ++ki) acc[tg][mi][ki] = mfma4(cur.pa[s][mi], cur.qb[s][tg][ki], acc[tg][mi][ki]);
    if (dob) {
#pragma unroll
      for (int mi = 0; mi < MCW; ++mi) bs[mi] += (cur.pa[0][mi] + cur.pa[1][mi]) + (cur.pa[2][mi] + cur.pa[3][mi]);
    }
  }
  if (dob) {
#pragma unroll
    for (int mi = 0; mi < MCW; ++mi) {
      float v = bs[mi];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      const int mt = mb * MCW + mi;
      if (kk == 0 && mt < MT) acc_add_f32(a.dbias, 16 * mt + i, v, a.d.det);
    }
  }
  const int g = lane >> 4, c = lane & 15;
#pragma unroll
  for (int tg = 0; tg < TG; ++tg)
#pragma unroll
    for (int mi = 0; mi < MCW; ++mi)
#pragma unroll
      for (int ki = 0; ki < KCW; ++ki) {
        __syncthreads();
        st4(&red[wv][lane * 4], acc[tg][mi][ki]);
        __syncthreads();
        const int mt = mb * MCW + mi, kt = kb * KCW + ki;
        if (mt >= MT || kt >= KT) continue;          // block-uniform
        // wave w sums and adds register r == w of every lane's fragment
        const float sum = (red[0][lane * 4 + wv] + red[1][lane * 4 + wv]) + (red[2][lane * 4 + wv] + red[3][lane * 4 + wv]);
        acc_add_f32(a.dW, ((size_t)(tap0 + tg) * Co + 16 * mt + 4 * g + wv) * Ci + 16 * kt + c, sum, a.d.det);
      }
}

// Weight gradient of the 3x3x3 convolutions of the full-resolution U-Net levels (16 / 32 channels, millions of voxels) from
// LDS tiles (round 3).  The per-wave kernel above loads every MFMA operand element with its own dword load and re-reads the
// activations once per tap group from L1 / L2: 32-48 TFLOP/s.  Here a workgroup of 8 waves owns a 2 x 4 x 32 block of output
// voxels at a time (persistent, grid-stride): the activations of the block WITH its one-voxel halo (4 x 6 x 34 voxels,
// zero outside the volume) and the output gradients of the block go to LDS once, as coalesced 16-byte loads, and all 27 taps
// read their operand fragments from there -- one ds_read_b32 per lane and k-step (lane 16k+i: voxel 4s+k of the 16-voxel
// tile, channel i), the tap being a constant offset into the halo tile.  The 27 taps are dealt to the 8 waves (4 4 4 3 3 3 3 3:
// the two waves of a SIMD carry 7 7 7 6 taps), every wave keeps the accumulators of its taps for the whole launch and adds them
// with one set of fp32 atomics at the end.  32-channel tiles: the two 16-channel halves of a voxel are swapped on odd voxels,
// so that the two voxels a half-wave reads in one ds_read_b32 land in different banks.
// TX (round 5): 16-voxel rows for the 64-channel level (a 32-voxel halo tile of 64 channels would not fit), whose output
// channels are additionally split over blockIdx.y in groups of COT tiles (COS = all of them): 4 taps x 4 x 4 accumulator
// tiles would not fit the register file; the workgroups of one block stage the same halo tile (from L2).
// Address arithmetic (round 5, after an SQ counter pass: 3 scalar + 1.7 vector instructions per MFMA in the 16-channel
// instantiation, matrix pipe 61 % busy): staging goes by halo ROWS (fixed t, z: contiguous in memory) through buffer loads --
// row base and validity are wave-uniform, a lane's offset in the row, its LDS address and its swizzle do not depend on the
// block and are computed once; out-of-volume requests hit the descriptor's range check and return zeros.  The fragment reads
// of the MFMA loop use per-tap lane addresses computed once per launch + one add per (t, z) row of the block + immediates.
template <int CIT, int COT, int TX = 32, int COS = COT>
__global__ __launch_bounds__(512) void k_conv3d_wgrad_lds(ConvArgs a) {
  constexpr int TT = 2, TZ = 4, HT = TT + 2, HZ = TZ + 2, HX = TX + 2, NXH = TX / 16;
  constexpr int CoF = 16 * COS;                        // channels of a voxel of ybar in memory
  const int coh = COS == COT ? 0 : blockIdx.y * COT;   // first output tile of this workgroup
  constexpr int Ci = 16 * CIT, Co = 16 * COT, NH = HT * HZ * HX, NV = TT * TZ * TX;
  constexpr bool SWX = CIT % 2 == 0, SWY = COT % 2 == 0;
  __shared__ __attribute__((aligned(16))) float xs[NH * Ci];
  __shared__ __attribute__((aligned(16))) float ys[NV * Co];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, j = lane & 15;
  const int ntap_w = wv < 3 ? 4 : 3;
  const int tap0 = wv < 3 ? 4 * wv : 12 + 3 * (wv - 3);
  f32x4 acc[4][COT][CIT];
#pragma unroll
  for (int ti = 0; ti < 4; ++ti)
#pragma unroll
    for (int co = 0; co < COT; ++co)
#pragma unroll
      for (int ci = 0; ci < CIT; ++ci) acc[ti][co][ci] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int T = a.d.T, Z = a.d.Z, X = a.d.X;
  const int nbx = X / TX, nbz = Z / TZ, nbt = T / TT;
  const int nblk = a.d.B * nbt * nbz * nbx;
  // bias gradient (column sums of ybar) from the staged block: thread -> (channel, every (512 / Co)-th voxel)
  const int bc = threadIdx.x % Co, bv0 = threadIdx.x / Co;
  float bsum = 0.f;

  // ---- staging: 3 halo rows of x and (waves 0 .. TT * TZ - 1) one row of ybar per wave ----------------------------------
  constexpr int QX = Ci / 4, QY = Co / 4, RPW = HT * HZ / 8, RQX = HX * QX, NKX = (RQX + 63) / 64, RQY = TX * QY,
                NKY = (RQY + 63) / 64;
  static_assert(HT * HZ == 8 * RPW && TT * TZ == 8, "24 halo rows / 8 output rows over 8 waves");
  const size_t nvox = (size_t)a.d.B * T * Z * X;
  const auto xr = load_rsrc(a.x, (unsigned)(nvox * Ci * 4));
  const auto yr = load_rsrc(a.ybar, (unsigned)(nvox * CoF * 4));
  int xoff[NKX], yoff[NKY];
  float* xl[NKX];
  float* yl[NKY];
  unsigned f_first = 0u, f_last = 0u, f_none = 0u, fy_none = 0u;
#pragma unroll
  for (int k = 0; k < NKX; ++k) {
    const int kq = lane + 64 * k, hx = kq / QX, q = kq % QX;
    xoff[k] = (kq - QX) * 16;                          // relative to voxel x0 of the row (the halo voxel x0 - 1: negative)
    xl[k] = xs + ((wv * RPW) * HX + hx) * Ci + 4 * (SWX ? (q ^ ((hx & 1) << 2)) : q);      // (HX is even: parity of hv = of hx)
    f_first |= (hx == 0 ? 1u : 0u) << k;
    f_last |= (hx == HX - 1 ? 1u : 0u) << k;
    f_none |= (kq >= RQX ? 1u : 0u) << k;
  }
#pragma unroll
  for (int k = 0; k < NKY; ++k) {
    const int kq = lane + 64 * k, xx = kq / QY, q = kq % QY;
    yoff[k] = (xx * CoF + 16 * coh + 4 * q) * 4;
    yl[k] = ys + (wv * TX + xx) * Co + 4 * (SWY ? (q ^ ((xx & 1) << 2)) : q);
    fy_none |= (kq >= RQY ? 1u : 0u) << k;
  }
  f32x4 px[RPW][NKX], py[NKY];
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
      const int row = wv * RPW + i;                    // wave-uniform
      const int t = t0 + row / HZ - 1, z = z0 + row % HZ - 1;
      const bool rowok = t >= 0 && t < T && z >= 0 && z < Z;
      const int rbase = ((((b * T + t) * Z + z) * X + x0) * Ci) * 4;
#pragma unroll
      for (int k = 0; k < NKX; ++k) {
        const bool ok = rowok && !((bad >> k) & 1u);
        px[i][k] = __builtin_bit_cast(f32x4, buf_ld16(xr, ok ? rbase + xoff[k] : (int)0x80000000, 0));
      }
    }
    const int ybase = ((((b * T + t0 + wv / TZ) * Z + z0 + wv % TZ) * X + x0) * CoF) * 4;
#pragma unroll
    for (int k = 0; k < NKY; ++k)
      py[k] = __builtin_bit_cast(f32x4, buf_ld16(yr, ((fy_none >> k) & 1u) ? (int)0x80000000 : ybase + yoff[k], 0));
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < RPW; ++i)
#pragma unroll
      for (int k = 0; k < NKX; ++k)
        if (!((f_none >> k) & 1u)) st4(xl[k] + i * HX * Ci, px[i][k]);
#pragma unroll
    for (int k = 0; k < NKY; ++k)
      if (!((fy_none >> k) & 1u)) st4(yl[k], py[k]);
  };

  // ---- fragment addresses: lane part per tap, computed once ---------------------------------------------------------------
  // x fragment of tap (dt, dz, dx), 16-voxel tile at (tt, zz, xh), k-step sk, input tile ci:
  //   xs[(((tt + 1 + dt) * HZ + zz + 1 + dz) * HX + 16 xh + 1 + dx + g + 4 sk) * Ci + 16 (ci ^ parity) + j],  parity = (1 + dx + g) & 1
  const float* ax[4][2];
#pragma unroll
  for (int ti = 0; ti < 4; ++ti) {
    const int tap = tap0 + (ti < ntap_w ? ti : 0);
    const int dt = tap / 9 - 1, dz = (tap / 3) % 3 - 1, dx = tap % 3 - 1;
    const int par = SWX ? ((1 + dx + g) & 1) : 0;
#pragma unroll
    for (int p2 = 0; p2 < 2; ++p2) ax[ti][p2] = xs + ((dt * HZ + dz) * HX + 1 + dx + g) * Ci + 16 * (p2 ^ par) + j;
  }
  const float* ay[2];
#pragma unroll
  for (int p2 = 0; p2 < 2; ++p2) ay[p2] = ys + g * Co + 16 * (p2 ^ (SWY ? (g & 1) : 0)) + j;

  if ((int)blockIdx.x < nblk) fetch(blockIdx.x);
#pragma unroll 1
  for (int bi = blockIdx.x; bi < nblk; bi += gridDim.x) {
    __syncthreads();                                   // the readers of the previous block are done
    stage();
    __syncthreads();
    if (bi + (int)gridDim.x < nblk) fetch(bi + gridDim.x);
    if (a.dbias) {
      for (int vv = bv0; vv < NV; vv += 512 / Co) bsum += ys[vv * Co + (SWY ? (bc ^ ((vv & 1) << 4)) : bc)];
    }
#pragma unroll 1
    for (int rz = 0; rz < TT * TZ; ++rz) {             // (t, z) rows of the block
      const int tt = rz / TZ, zz = rz % TZ;
      const int rowx = ((tt + 1) * HZ + zz + 1) * HX * Ci, rowy = rz * TX * Co;
      const float* axr[4][2];
#pragma unroll
      for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int p2 = 0; p2 < (CIT > 1 ? 2 : 1); ++p2) axr[ti][p2] = ax[ti][p2] + rowx;
      const float* ayr[2];
#pragma unroll
      for (int p2 = 0; p2 < (COT > 1 ? 2 : 1); ++p2) ayr[p2] = ay[p2] + rowy;
#pragma unroll
      for (int xh = 0; xh < NXH; ++xh) {
        float pa[COT][4];
#pragma unroll
        for (int co = 0; co < COT; ++co)
#pragma unroll
          for (int sk = 0; sk < 4; ++sk) pa[co][sk] = ayr[co & 1][(16 * xh + 4 * sk) * Co + 16 * (co & ~1)];
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
          if (ti < ntap_w) {                           // wave-uniform
#pragma unroll
            for (int sk = 0; sk < 4; ++sk) {
              float qb[CIT];
#pragma unroll
              for (int ci = 0; ci < CIT; ++ci) qb[ci] = axr[ti][ci & 1][(16 * xh + 4 * sk) * Ci + 16 * (ci & ~1)];
#pragma unroll
              for (int co = 0; co < COT; ++co)
#pragma unroll
                for (int ci = 0; ci < CIT; ++ci) acc[ti][co][ci] = mfma4(pa[co][sk], qb[ci], acc[ti][co][ci]);
            }
          }
        }
      }
    }
  }
  if (a.dbias) {            // (block-uniform) one atomic per channel and workgroup: the 512 / Co partial sums of a channel meet
    __syncthreads();        // in LDS first (round 5: 512 same-address atomics per workgroup serialised in L2 -- with the
    xs[threadIdx.x] = bsum; // pipelined staging, 16 -> 16 channels on 4.2 M voxels: 1199 -> 591 us, 48 -> 98 TFLOP/s)
    __syncthreads();
    if (threadIdx.x < Co) {
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < 512 / Co; ++k) sum += xs[threadIdx.x + Co * k];
      acc_add_f32(a.dbias, 16 * coh + threadIdx.x, sum, a.d.det);
    }
  }
#pragma unroll
  for (int ti = 0; ti < 4; ++ti) {
    if (ti >= ntap_w) continue;
#pragma unroll
    for (int co = 0; co < COT; ++co)
#pragma unroll
      for (int ci = 0; ci < CIT; ++ci)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
          acc_add_f32(a.dW, ((size_t)(tap0 + ti) * CoF + 16 * (coh + co) + 4 * g + rr) * Ci + 16 * ci + j, acc[ti][co][ci][rr], a.d.det);
  }
}

// Weight gradient of the 1x1x1 convolutions of the wide levels (round 5): dW[co][ci] = sum over voxels of ybar[v][co] x[v][ci]
// is a GEMM whose contraction runs over the voxels, i.e. over the SLOW index of both channels-last operands.  The per-wave kernel
// above fetches every MFMA operand element with a dword load (64-byte segments, ~2.5 TB/s on 16-channel tensors: these launches
// are HBM bound and reached 40-50 % of the stream rate).  Here a persistent workgroup of 8 waves stages 256 consecutive voxels of
// both operands with 16-byte loads -- software-pipelined through registers like the 3x3x3 kernel above, so a block's loads are
// in flight while the previous block's MFMAs run -- and the waves read their fragments from LDS (ds_read_b32: lane 16 k + i ->
// voxel 4 s + k, channel i; channel bit 4 is flipped on odd voxels when a voxel's row is a multiple of 32 banks long).
// ONLOAD: x' = max(0, bn(x)) applied between the registers and LDS (conv3 of a residual block, see k_conv3d_wgrad).
// Every wave owns two of the block's 16 voxel tiles and all COT x CIT accumulator tiles; the 8 waves meet in LDS at the end.
template <int CIT, int COT, bool ONLOAD>
__global__ __launch_bounds__(512) void k_conv1_wgrad_lds(ConvArgs a) {
  constexpr int NV = 256, Ci = 16 * CIT, Co = 16 * COT;
  constexpr bool SWX = CIT % 2 == 0, SWY = COT % 2 == 0;
  __shared__ __attribute__((aligned(16))) float xs[NV * Ci];
  __shared__ __attribute__((aligned(16))) float ys[NV * Co];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, j = lane & 15;
  f32x4 acc[COT][CIT];
#pragma unroll
  for (int co = 0; co < COT; ++co)
#pragma unroll
    for (int ci = 0; ci < CIT; ++ci) acc[co][ci] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nblk = (a.nvox + NV - 1) / NV;
  const int bc = threadIdx.x % Co, bv0 = threadIdx.x / Co;
  float bsum = 0.f;
  constexpr int NXI = NV * (Ci / 4) / 512, NYI = NV * (Co / 4) / 512;      // = 2 CIT, 2 COT
  // 512 is a multiple of Ci / 4 and Co / 4: a thread stages the same channel quad in every round
  const int qx = threadIdx.x % (Ci / 4), vx0 = threadIdx.x / (Ci / 4);
  const int qy = threadIdx.x % (Co / 4), vy0 = threadIdx.x / (Co / 4);
  f32x4 omean, oscale, obeta;
  if (ONLOAD) {
    omean = ld4(a.in_stat + 4 * qx);
    oscale = ld4(a.in_stat + Ci + 4 * qx);
    if (a.in_gamma) oscale = oscale * ld4(a.in_gamma + 4 * qx);
    obeta = a.in_beta ? ld4(a.in_beta + 4 * qx) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  f32x4 px[NXI], py[NYI];
  auto fetch = [&](int bi) {
#pragma unroll
    for (int it = 0; it < NXI; ++it) {
      const int v = bi * NV + vx0 + (512 / (Ci / 4)) * it;
      px[it] = v < a.nvox ? ld4(a.x + (size_t)v * Ci + 4 * qx) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int it = 0; it < NYI; ++it) {
      const int v = bi * NV + vy0 + (512 / (Co / 4)) * it;
      py[it] = v < a.nvox ? ld4(a.ybar + (size_t)v * Co + 4 * qy) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto stage = [&](int bi) {
#pragma unroll
    for (int it = 0; it < NXI; ++it) {
      const int vv = vx0 + (512 / (Ci / 4)) * it;
      f32x4 v = px[it];
      if (ONLOAD) {
        v = (v - omean) * oscale + obeta;
        const bool in = bi * NV + vv < a.nvox;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (in && v[r] > 0.f) ? v[r] : 0.f;
      }
      st4(xs + vv * Ci + 4 * (SWX ? (qx ^ ((vv & 1) << 2)) : qx), v);
    }
#pragma unroll
    for (int it = 0; it < NYI; ++it) {
      const int vv = vy0 + (512 / (Co / 4)) * it;
      st4(ys + vv * Co + 4 * (SWY ? (qy ^ ((vv & 1) << 2)) : qy), py[it]);
    }
  };
  if ((int)blockIdx.x < nblk) fetch(blockIdx.x);
#pragma unroll 1
  for (int bi = blockIdx.x; bi < nblk; bi += gridDim.x) {
    __syncthreads();                                   // the readers of the previous block are done
    stage(bi);
    __syncthreads();
    if (bi + (int)gridDim.x < nblk) fetch(bi + gridDim.x);
    if (a.dbias) {
      for (int vv = bv0; vv < NV; vv += 512 / Co) bsum += ys[vv * Co + (SWY ? (bc ^ ((vv & 1) << 4)) : bc)];
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int sk = 0; sk < 4; ++sk) {
        const int vv = 16 * (2 * wv + t) + 4 * sk + g;
        const int sw = (vv & 1) << 4;
        float pa[COT], qb[CIT];
#pragma unroll
        for (int co = 0; co < COT; ++co) pa[co] = ys[vv * Co + ((16 * co + j) ^ (SWY ? sw : 0))];
#pragma unroll
        for (int ci = 0; ci < CIT; ++ci) qb[ci] = xs[vv * Ci + ((16 * ci + j) ^ (SWX ? sw : 0))];
#pragma unroll
        for (int co = 0; co < COT; ++co)
#pragma unroll
          for (int ci = 0; ci < CIT; ++ci) acc[co][ci] = mfma4(pa[co], qb[ci], acc[co][ci]);
      }
    }
  }
  // the 8 waves' partial tiles meet in LDS (tile by tile, xs is free now): one set of atomics per workgroup
  float* red = xs;                                     // [8][256] floats = 8 KB <= NV * 16 * 4
  if (a.dbias) {                                       // (block-uniform) the 512 / Co partial column sums of a channel first:
    __syncthreads();                                   // 512 same-address atomics per workgroup serialise in L2
    red[threadIdx.x] = bsum;
    __syncthreads();
    if (threadIdx.x < Co) {
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < 512 / Co; ++k) sum += red[threadIdx.x + Co * k];
      acc_add_f32(a.dbias, threadIdx.x, sum, a.d.det);
    }
  }
#pragma unroll
  for (int co = 0; co < COT; ++co)
#pragma unroll
    for (int ci = 0; ci < CIT; ++ci) {
      __syncthreads();
      st4(red + wv * 256 + lane * 4, acc[co][ci]);
      __syncthreads();
      if (threadIdx.x < 256) {
        const int l = threadIdx.x >> 2, rr = threadIdx.x & 3;      // element rr of lane l's fragment
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) sum += red[w * 256 + threadIdx.x];
        acc_add_f32(a.dW, ((size_t)(16 * co + 4 * (l >> 4) + rr)) * Ci + 16 * ci + (l & 15), sum, a.d.det);
      }
    }
}

template <bool ONLOAD>
static bool launch_conv1_wgrad_lds(const ConvArgs& a, hipStream_t st) {
  const int Ci = a.d.Ci, Co = a.d.Co;
  if (a.d.ksize != 1 || (Ci != 16 && Ci != 32 && Ci != 64) || (Co != 16 && Co != 32 && Co != 64) || a.nvox < 65536)
    return false;
  const int nblk = (a.nvox + 255) / 256;
  // persistent workgroups: as many per CU as their LDS tiles (1 KB per channel of x and ybar) and 32 wave slots allow
  int per_cu = 160 / (Ci + Co + 1);
  if (per_cu > 4) per_cu = 4;
  const int gx_env = stpde_tune_get(STPDE_TUNE_CONV1_WGRAD_LDS_GX);      // (test override, 0 = own choice)
  int gx = gx_env > 0 ? gx_env : 256 * per_cu;
  if (gx > nblk) gx = nblk;
#define STPDE_C1W(CIT, COT)                                                                              \
  if (Ci == 16 * CIT && Co == 16 * COT) {                                                               \
    if constexpr (ONLOAD)                                                                               \
      STPDE_LAUNCH((k_conv1_wgrad_lds<CIT, COT, true>), dim3(gx), dim3(512), 0, st, a);                  \
    else                                                                                                \
      STPDE_LAUNCH((k_conv1_wgrad_lds<CIT, COT, false>), dim3(gx), dim3(512), 0, st, a);                 \
    return true;                                                                                        \
  }
  STPDE_C1W(1, 1) STPDE_C1W(1, 2) STPDE_C1W(1, 4) STPDE_C1W(2, 1) STPDE_C1W(2, 2) STPDE_C1W(2, 4) STPDE_C1W(4, 1)
  STPDE_C1W(4, 2) STPDE_C1W(4, 4)
#undef STPDE_C1W
  return false;
}

static int check_conv(const stpde_conv3d_desc* d) {
  if (!d || d->B < 1 || d->T < 1 || d->Z < 1 || d->X < 1 || d->Ci < 16 || d->Co < 16 || (d->Ci & 15) || (d->Co & 15) ||
      (d->ksize != 1 && d->ksize != 3)) {
    stpde_set_error("conv3d: bad descriptor (channels must be multiples of 16, ksize 1 or 3)");
    return STPDE_E_BADARG;
  }
  if ((size_t)d->B * d->T * d->Z * d->X >= (1u << 31) / 16) {
    stpde_set_error("conv3d: volume too large for int32 voxel indices");
    return STPDE_E_BADARG;
  }
  return STPDE_OK;
}

extern "C" int stpde_conv3d_fwd(const stpde_conv3d_desc* d, const float* x, const float* w_pack, const float* bias,
                                float* y, void* stream) {
  int rc = check_conv(d);
  if (rc) return rc;
  if (!x || !w_pack || !y) {
    stpde_set_error("conv3d_fwd: null pointer");
    return STPDE_E_BADARG;
  }
  ConvArgs a{};
  a.d = *d;
  a.x = x;
  a.w = w_pack;
  a.bias = bias;
  a.y = y;
  a.nvox = d->B * d->T * d->Z * d->X;
  const int ntiles = (a.nvox + 15) / 16;
  const int nchunks = (d->Co / 16 + 3) / 4;
  const int MT = d->Co / 16;
  const hipStream_t st = (hipStream_t)stream;
  // MC = output-channel tiles per pass: never more than the layer has (a clamped tile would redo real MFMA work)
  if (ntiles >= 16384) {   // big volumes: 4 voxel tiles per wave (4x weight reuse), one wave walks all chunks
    const dim3 grid((ntiles + 15) / 16, 1);
    if (MT == 1)
      STPDE_LAUNCH((k_conv3d_fwd<1, 4>), grid, dim3(256), 0, st, a);
    else if (MT == 2)
      STPDE_LAUNCH((k_conv3d_fwd<2, 4>), grid, dim3(256), 0, st, a);
    else
      STPDE_LAUNCH((k_conv3d_fwd<4, 4>), grid, dim3(256), 0, st, a);
  } else {
    const int gx = (ntiles + 3) / 4;
    const int gy = gx >= 1024 ? 1 : nchunks;
    // 3x3x3 convs of the deep levels: too few (voxel tile, channel chunk) pairs to fill 256 CUs -> split the taps
    int gz = 1;
    if (d->ksize == 3 && gx * gy < 256 && !d->det) {      // (the tap split adds partial outputs with atomics: not in deterministic mode)
      gz = (256 + gx * gy - 1) / (gx * gy);
      if (gz > 27) gz = 27;
    }
    if (gz > 1) {
      (void)hipMemsetAsync(y, 0, (size_t)a.nvox * d->Co * sizeof(float), st);
      STPDE_LAUNCH((k_conv3d_fwd<4, 1, true>), dim3(gx, gy, gz), dim3(256), 0, st, a);
    } else if (MT == 1) {
      STPDE_LAUNCH((k_conv3d_fwd<1, 1>), dim3(gx, gy), dim3(256), 0, st, a);
    } else if (MT == 2) {
      STPDE_LAUNCH((k_conv3d_fwd<2, 1>), dim3(gx, gy), dim3(256), 0, st, a);
    } else {
      STPDE_LAUNCH((k_conv3d_fwd<4, 1>), dim3(gx, gy), dim3(256), 0, st, a);
    }
  }
  return stpde_check_launch("k_conv3d_fwd");
}

static int conv3d_wgrad(const stpde_conv3d_desc* d, const float* x, const float* ybar, float* dW, float* dbias, void* stream);

extern "C" int stpde_conv3d_wgrad_onload(const stpde_conv3d_desc* d, const float* x, const float* ybar, float* dW, float* dbias,
                                         const float* in_stat, const float* in_gamma, const float* in_beta, void* stream) {
  int rc = check_conv(d);
  if (rc) return rc;
  if (!x || !ybar || !dW || !in_stat || d->ksize != 1) {
    stpde_set_error("conv3d_wgrad_onload: null pointer or ksize != 1");
    return STPDE_E_BADARG;
  }
  ConvArgs a{};
  a.d = *d;
  a.x = x;
  a.ybar = ybar;
  a.dW = dW;
  a.dbias = dbias;
  a.in_stat = in_stat;
  a.in_gamma = in_gamma;
  a.in_beta = in_beta;
  a.nvox = d->B * d->T * d->Z * d->X;
  if (launch_conv1_wgrad_lds<true>(a, (hipStream_t)stream)) return stpde_check_launch("k_conv1_wgrad_lds");
  const int ntiles = (a.nvox + 15) / 16;
  const int KT = d->Ci / 16, MT = d->Co / 16;
  const int gy = ((MT + 1) / 2) * ((KT + 1) / 2);
  int gx = 2048 / gy;
  if (gx > 1024) gx = 1024;
  if (gx > (ntiles + 3) / 4) gx = (ntiles + 3) / 4;
  if (gx < 1) gx = 1;
  STPDE_LAUNCH((k_conv3d_wgrad<2, 2, 1, true>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, a);
  return stpde_check_launch("k_conv3d_wgrad");
}

extern "C" int stpde_conv3d_wgrad(const stpde_conv3d_desc* d, const float* x, const float* ybar, float* dW,
                                  void* stream) {
  return conv3d_wgrad(d, x, ybar, dW, nullptr, stream);
}
extern "C" int stpde_conv3d_wgrad_bias(const stpde_conv3d_desc* d, const float* x, const float* ybar, float* dW, float* dbias,
                                       void* stream) {
  return conv3d_wgrad(d, x, ybar, dW, dbias, stream);
}

static int conv3d_wgrad(const stpde_conv3d_desc* d, const float* x, const float* ybar, float* dW, float* dbias, void* stream) {
  int rc = check_conv(d);
  if (rc) return rc;
  if (!x || !ybar || !dW) {
    stpde_set_error("conv3d_wgrad: null pointer");
    return STPDE_E_BADARG;
  }
  ConvArgs a{};
  a.d = *d;
  a.x = x;
  a.ybar = ybar;
  a.dW = dW;
  a.dbias = dbias;
  a.nvox = d->B * d->T * d->Z * d->X;
  if (launch_conv1_wgrad_lds<false>(a, (hipStream_t)stream)) return stpde_check_launch("k_conv1_wgrad_lds");
  const int ntiles = (a.nvox + 15) / 16;
  const int KT = d->Ci / 16, MT = d->Co / 16;
  // (tap group, co block, ci block) triples on blockIdx.y; voxel-tile stripes on blockIdx.x: ~2048 blocks in total
  // (8 waves per SIMD hide the dword-load latency), at most 1024 stripes (= atomic adds any dW element receives)
  auto stripes = [&](int gy) {
    int gx = 2048 / gy;
    if (gx > 1024) gx = 1024;
    if (gx > (ntiles + 3) / 4) gx = (ntiles + 3) / 4;
    return gx < 1 ? 1 : gx;
  };
  // full-resolution levels (16 / 32 channels, volume made of whole 2 x 4 x 32 blocks, enough of them): the LDS-tile kernel
  const int nblk = d->B * (d->T / 2) * (d->Z / 4) * (d->X / 32);
  // (the LDS kernels address x / ybar with 32-bit byte offsets)
  const bool off32 = (size_t)a.nvox * (d->Ci > d->Co ? d->Ci : d->Co) * 4 < (1u << 31);
  if (off32 && d->ksize == 3 && KT <= 2 && MT <= 2 && d->T % 2 == 0 && d->Z % 4 == 0 && d->X % 32 == 0 && nblk >= 256) {
    const int wide = (KT == 2 || MT == 2);
    int gx = wide ? 256 : 512;               // persistent workgroups: one (137 KB of LDS) or two (68 KB) per CU
    // Volumes below ~2 M voxels: the training step runs this kernel on a side stream next to the input-gradient chain
    // (unet3d._DeferredGrads), which at that size is a chain of short launches -- a grid that fills every CU with persistent
    // MFMA-bound workgroups stretches that chain by more than the weight gradients shrink (2^17-point step 63.2 -> 64.6 ms);
    // on half of the CUs it is neutral there and still 2x the per-wave kernel when it runs alone.  (stpde_tune
    // "conv_wgrad_lds_gx": test override of the grid.)
    const int gx_env = stpde_tune_get(STPDE_TUNE_CONV_WGRAD_LDS_GX);
    const int half_below = 8192;
    if (gx_env > 0)
      gx = gx_env;
    else if (nblk < half_below)
      gx = 128;
    if (gx > nblk) gx = nblk;
    if (KT == 1 && MT == 1)
      STPDE_LAUNCH((k_conv3d_wgrad_lds<1, 1>), dim3(gx), dim3(512), 0, (hipStream_t)stream, a);
    else if (KT == 2 && MT == 2)
      STPDE_LAUNCH((k_conv3d_wgrad_lds<2, 2>), dim3(gx), dim3(512), 0, (hipStream_t)stream, a);
    else if (KT == 1)
      STPDE_LAUNCH((k_conv3d_wgrad_lds<1, 2>), dim3(gx), dim3(512), 0, (hipStream_t)stream, a);
    else
      STPDE_LAUNCH((k_conv3d_wgrad_lds<2, 1>), dim3(gx), dim3(512), 0, (hipStream_t)stream, a);
    return stpde_check_launch("k_conv3d_wgrad_lds");
  }
  // 64 -> 64 channels (third level): 2 x 4 x 16 blocks, one 16-channel output tile per workgroup on blockIdx.y (round 5; two
  // tiles per workgroup spill: 4 taps x 2 x 4 accumulator tiles + the staging registers)
  const int nblk16 = d->B * (d->T / 2) * (d->Z / 4) * (d->X / 16);
  if (off32 && d->ksize == 3 && KT == 4 && MT == 4 && d->T % 2 == 0 && d->Z % 4 == 0 && d->X % 16 == 0 &&
      nblk16 >= 256) {
    int gx = 64;                             // x 4 output tiles: one workgroup (116 KB of LDS) per CU
    if (gx > nblk16) gx = nblk16;
    STPDE_LAUNCH((k_conv3d_wgrad_lds<4, 1, 16, 4>), dim3(gx, 4), dim3(512), 0, (hipStream_t)stream, a);
    return stpde_check_launch("k_conv3d_wgrad_lds");
  }
  if (d->ksize == 3 && KT == 1 && MT == 1) {
    STPDE_LAUNCH((k_conv3d_wgrad<1, 1, 9>), dim3(stripes(3), 3), dim3(256), 0, (hipStream_t)stream, a);
  } else if (d->ksize == 3) {
    const int gy = 9 * ((MT + 1) / 2) * ((KT + 1) / 2);
    STPDE_LAUNCH((k_conv3d_wgrad<2, 2, 3>), dim3(stripes(gy), gy), dim3(256), 0, (hipStream_t)stream, a);
  } else {
    const int gy = ((MT + 1) / 2) * ((KT + 1) / 2);
    STPDE_LAUNCH((k_conv3d_wgrad<2, 2, 1>), dim3(stripes(gy), gy), dim3(256), 0, (hipStream_t)stream, a);
  }
  return stpde_check_launch("k_conv3d_wgrad");
}
