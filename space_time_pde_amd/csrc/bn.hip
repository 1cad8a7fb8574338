// BatchNorm3d (+ optional residual add, + optional ReLU) of the U-Net residual blocks (src/unet3d.py:39-56 of the
// reference: conv-bn-relu, conv-bn-relu, conv-bn, + shortcut, relu) on channels-last activations x[N][C]
// (N = B*T*Z*X voxels, C a power of two, 16 <= C <= 512).
//   forward (training): k_bn_stats  -> per-channel shifted sums (shift = first voxel: no cancellation in the variance)
//                        k_bn_apply  -> y = act((x - mean) * rstd * gamma + beta [+ r]); block 0 also updates the
//                                       running statistics with torch's convention (unbiased variance, momentum)
//   backward:            k_bn_bwd_reduce -> sum(dz), sum(dz * xhat) with dz = dy * [y > 0]
//                        k_bn_bwd_apply  -> dx = gamma * rstd * (dz - mean(dz) - xhat * mean(dz * xhat)), dr = dz
// Evaluation mode uses the running statistics (no batch terms in dx).  All four are single HBM passes.
#include "common.h"

struct BnArgs {
  stpde_bn_desc d;
  const float* x;
  const float* r;        // residual input or null
  const float* gamma;
  const float* beta;
  float* running_mean;   // updated in training mode (may be null)
  float* running_var;
  float* sums;           // [3][C]: shift, sum(x - shift), sum((x - shift)^2)      (training forward)
  float* stat;           // [2][C]: mean, rstd (written by apply, read by backward)
  float* y;
  const float* dy;
  float* bsum;           // [2][C]: sum(dz), sum(dz * xhat)
  float* dx;
  float* dr;             // gradient of the residual input or null
  float* dgamma;
  float* dbeta;
};

// thread -> (channel slot, row offset): C <= 256: one channel per thread, 256 / C rows in flight per block;
// C == 512: two channels per thread (c and c + 256), one row in flight
struct BnMap {
  int c, nc, row0, rstep;
};
__device__ __forceinline__ BnMap bn_map(int C) {
  BnMap m;
  if (C <= 256) {
    m.c = threadIdx.x % C;
    m.nc = 1;
    m.row0 = threadIdx.x / C;
    m.rstep = 256 / C;
  } else {
    m.c = threadIdx.x;
    m.nc = C / 256;
    m.row0 = 0;
    m.rstep = 1;
  }
  return m;
}

// sums over the threads of a block that share a channel slot; result valid for threads with row0 == 0
__device__ __forceinline__ float bn_block_sum(float v, float* sh, const BnMap& m, int C) {
  __syncthreads();
  sh[threadIdx.x] = v;
  __syncthreads();
  float s = 0.f;
  if (m.row0 == 0) {
    const int stride = C <= 256 ? C : 256;
    for (int t = threadIdx.x; t < 256; t += stride) s += sh[t];
  }
  return s;
}

__global__ __launch_bounds__(256) void k_bn_stats(BnArgs a) {
  __shared__ float sh[256];
  const int C = a.d.C;
  const long N = a.d.N;
  const BnMap m = bn_map(C);
  const long rows_per_block = (N + gridDim.x - 1) / gridDim.x;
  const long lo = (long)blockIdx.x * rows_per_block;
  const long hi = lo + rows_per_block < N ? lo + rows_per_block : N;
  for (int k = 0; k < m.nc; ++k) {
    const int c = m.c + 256 * k;
    const float shift = a.x[c];                    // voxel 0 of this channel
    float s1 = 0.f, s2 = 0.f;
    for (long row = lo + m.row0; row < hi; row += m.rstep) {
      const float v = a.x[row * C + c] - shift;
      s1 += v;
      s2 += v * v;
    }
    s1 = bn_block_sum(s1, sh, m, C);
    s2 = bn_block_sum(s2, sh, m, C);
    if (m.row0 == 0) {
      atomicAdd(a.sums + C + c, s1);
      atomicAdd(a.sums + 2 * C + c, s2);
      if (blockIdx.x == 0) a.sums[c] = shift;
    }
  }
}

__global__ __launch_bounds__(256) void k_bn_apply(BnArgs a) {
  const int C = a.d.C;
  const long N = a.d.N;
  const BnMap m = bn_map(C);
  const long rows_per_block = (N + gridDim.x - 1) / gridDim.x;
  const long lo = (long)blockIdx.x * rows_per_block;
  const long hi = lo + rows_per_block < N ? lo + rows_per_block : N;
  for (int k = 0; k < m.nc; ++k) {
    const int c = m.c + 256 * k;
    float mean, var;
    if (a.d.training) {
      const float shift = a.sums[c], e1 = a.sums[C + c] / (float)N, e2 = a.sums[2 * C + c] / (float)N;
      mean = shift + e1;
      var = fmaxf(e2 - e1 * e1, 0.f);               // biased variance of the batch
    } else {
      mean = a.running_mean[c];
      var = a.running_var[c];
    }
    const float rstd = 1.f / sqrtf(var + a.d.eps);
    if (blockIdx.x == 0 && m.row0 == 0) {
      a.stat[c] = mean;
      a.stat[C + c] = rstd;
      if (a.d.training && a.running_mean) {         // torch: running = (1 - momentum) * running + momentum * batch
        const float unb = N > 1 ? var * ((float)N / (float)(N - 1)) : var;
        a.running_mean[c] = (1.f - a.d.momentum) * a.running_mean[c] + a.d.momentum * mean;
        a.running_var[c] = (1.f - a.d.momentum) * a.running_var[c] + a.d.momentum * unb;
      }
    }
    const float g = a.gamma ? a.gamma[c] : 1.f, b = a.beta ? a.beta[c] : 0.f;
    const float scale = rstd * g;
    for (long row = lo + m.row0; row < hi; row += m.rstep) {
      const long i = row * C + c;
      float v = (a.x[i] - mean) * scale + b;
      if (a.r) v += a.r[i];
      if (a.d.relu) v = v > 0.f ? v : 0.f;
      a.y[i] = v;
    }
  }
}

__global__ __launch_bounds__(256) void k_bn_bwd_reduce(BnArgs a) {
  __shared__ float sh[256];
  const int C = a.d.C;
  const long N = a.d.N;
  const BnMap m = bn_map(C);
  const long rows_per_block = (N + gridDim.x - 1) / gridDim.x;
  const long lo = (long)blockIdx.x * rows_per_block;
  const long hi = lo + rows_per_block < N ? lo + rows_per_block : N;
  for (int k = 0; k < m.nc; ++k) {
    const int c = m.c + 256 * k;
    const float mean = a.stat[c], rstd = a.stat[C + c];
    float s1 = 0.f, s2 = 0.f;
    for (long row = lo + m.row0; row < hi; row += m.rstep) {
      const long i = row * C + c;
      float dz = a.dy[i];
      if (a.d.relu && !(a.y[i] > 0.f)) dz = 0.f;
      s1 += dz;
      s2 += dz * (a.x[i] - mean) * rstd;
    }
    s1 = bn_block_sum(s1, sh, m, C);
    s2 = bn_block_sum(s2, sh, m, C);
    if (m.row0 == 0) {
      atomicAdd(a.bsum + c, s1);
      atomicAdd(a.bsum + C + c, s2);
    }
  }
}

__global__ __launch_bounds__(256) void k_bn_bwd_apply(BnArgs a) {
  const int C = a.d.C;
  const long N = a.d.N;
  const BnMap m = bn_map(C);
  const long rows_per_block = (N + gridDim.x - 1) / gridDim.x;
  const long lo = (long)blockIdx.x * rows_per_block;
  const long hi = lo + rows_per_block < N ? lo + rows_per_block : N;
  for (int k = 0; k < m.nc; ++k) {
    const int c = m.c + 256 * k;
    const float mean = a.stat[c], rstd = a.stat[C + c];
    const float sdz = a.bsum[c], sdzx = a.bsum[C + c];
    if (blockIdx.x == 0 && m.row0 == 0) {
      if (a.dbeta) a.dbeta[c] = sdz;
      if (a.dgamma) a.dgamma[c] = sdzx;
    }
    const float g = a.gamma ? a.gamma[c] : 1.f;
    const float k1 = a.d.training ? sdz / (float)N : 0.f, k2 = a.d.training ? sdzx / (float)N : 0.f;
    for (long row = lo + m.row0; row < hi; row += m.rstep) {
      const long i = row * C + c;
      float dz = a.dy[i];
      if (a.d.relu && !(a.y[i] > 0.f)) dz = 0.f;
      if (a.dr) a.dr[i] = dz;
      if (a.dx) a.dx[i] = g * rstd * (dz - k1 - (a.x[i] - mean) * rstd * k2);
    }
  }
}

static int bn_check(const stpde_bn_desc* d) {
  if (!d || d->N <= 0 || d->C < 16 || d->C > 512 || (d->C & (d->C - 1))) {
    stpde_set_error("batchnorm: C must be a power of two in [16, 512], N > 0");
    return STPDE_E_BADARG;
  }
  return STPDE_OK;
}

static unsigned bn_grid(const stpde_bn_desc* d) {
  const long rows_per_pass = d->C <= 256 ? 256 / d->C : 1;
  long blocks = (d->N + rows_per_pass * 8 - 1) / (rows_per_pass * 8);   // >= 8 row passes per block
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

extern "C" int stpde_bn_fwd(const stpde_bn_desc* d, const float* x, const float* residual, const float* gamma,
                            const float* beta, float* running_mean, float* running_var, float* sums, float* stat,
                            float* y, void* stream) {
  int rc = bn_check(d);
  if (rc) return rc;
  if (!x || !y || !stat || (d->training && !sums) || (!d->training && (!running_mean || !running_var))) {
    stpde_set_error("bn_fwd: null pointer");
    return STPDE_E_BADARG;
  }
  BnArgs a{};
  a.d = *d;
  a.x = x;
  a.r = residual;
  a.gamma = gamma;
  a.beta = beta;
  a.running_mean = running_mean;
  a.running_var = running_var;
  a.sums = sums;
  a.stat = stat;
  a.y = y;
  const unsigned grid = bn_grid(d);
  if (d->training) {
    if (!d->scratch_zeroed) (void)hipMemsetAsync(sums, 0, 3 * d->C * sizeof(float), (hipStream_t)stream);
    STPDE_LAUNCH(k_bn_stats, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    rc = stpde_check_launch("k_bn_stats");
    if (rc) return rc;
  }
  STPDE_LAUNCH(k_bn_apply, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  return stpde_check_launch("k_bn_apply");
}

extern "C" int stpde_bn_bwd(const stpde_bn_desc* d, const float* x, const float* y, const float* dy,
                            const float* gamma, const float* stat, float* bsum, float* dx, float* dresidual,
                            float* dgamma, float* dbeta, void* stream) {
  int rc = bn_check(d);
  if (rc) return rc;
  if (!x || !dy || !stat || !bsum || (d->relu && !y)) {
    stpde_set_error("bn_bwd: null pointer");
    return STPDE_E_BADARG;
  }
  BnArgs a{};
  a.d = *d;
  a.x = x;
  a.y = const_cast<float*>(y);
  a.dy = dy;
  a.gamma = gamma;
  a.stat = const_cast<float*>(stat);
  a.bsum = bsum;
  a.dx = dx;
  a.dr = dresidual;
  a.dgamma = dgamma;
  a.dbeta = dbeta;
  const unsigned grid = bn_grid(d);
  if (!d->scratch_zeroed) (void)hipMemsetAsync(bsum, 0, 2 * d->C * sizeof(float), (hipStream_t)stream);
  STPDE_LAUNCH(k_bn_bwd_reduce, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  rc = stpde_check_launch("k_bn_bwd_reduce");
  if (rc) return rc;
  STPDE_LAUNCH(k_bn_bwd_apply, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  return stpde_check_launch("k_bn_bwd_apply");
}
