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
      bn_stat_f64(a.f.in_sums, Ci, c, N, a.f.in_eps, mean, var, rstd);
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
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s1[r] = row_sum16(s1[r]);
          s2[r] = row_sum16(s2[r]);
        }
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
          if (EPI == 1) {
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
    if (d->ksize == 3 && gx * gy < 256) {
      // deep levels: the tap-split kernel of conv3d.hip fills the chip; its output holds partial sums until the last atomic,
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
