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
  float* sums;           // [STPDE_BN_REP][3][C]: shift (replica 0), sum(x - shift), sum((x - shift)^2)   (training forward)
  float* stat;           // [2][C]: mean, rstd (written by apply, read by backward)
  float* y;
  const float* dy;
  float* bsum;           // [STPDE_BN_REP][2][C]: sum(dz), sum(dz * xhat)
  float* dx;
  float* dr;             // gradient of the residual input or null
  float* dgamma;
  float* dbeta;
  int il;                // backward kernels: rows dealt to the blocks in interleaved 16 KiB chunks instead of one range per block
};

// thread -> (channel quad q = 4 consecutive channels, row offset): C / 4 quads, 256 / (C / 4) rows in flight per block; every
// access is one 16-byte load / store per thread and a block iteration covers 4 KiB of contiguous memory.  (First version: one
// channel and 4 bytes per thread -- 1.15 TB/s on the 524,288-voxel level; the four kernels were 4.0 ms of the step.)
struct BnMap {
  int q, row0, rstep, nq;
};
__device__ __forceinline__ BnMap bn_map(int C) {
  BnMap m;
  m.nq = C >> 2;                 // 4 .. 128
  m.q = threadIdx.x % m.nq;
  m.row0 = threadIdx.x / m.nq;
  m.rstep = 256 / m.nq;
  return m;
}

// sums over the threads of a block that share a channel quad; result valid for threads with row0 == 0.
// Lanes of a wave that share a quad are nq apart: butterfly over the lane bits above log2(nq), then the four waves through LDS.
__device__ __forceinline__ f32x4 bn_block_sum(f32x4 v, f32x4* sh, const BnMap& m) {
  for (int off = 32; off >= m.nq; off >>= 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] += __shfl_xor(v[i], off, 64);
  }
  __syncthreads();
  sh[threadIdx.x] = v;
  __syncthreads();
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
  if (m.row0 == 0) {
    if (m.nq <= 64) {
      for (int w = 0; w < 4; ++w) s += sh[64 * w + m.q];       // lane q of every wave holds that wave's sum
    } else {
      s = sh[m.q] + sh[128 + m.q];                             // nq == 128: two rows in flight, no lanes to fold
    }
  }
  return s;
}

__device__ __forceinline__ void atomic_add4(float* p, f32x4 v) {
#pragma unroll
  for (int i = 0; i < 4; ++i) atomicAdd(p + i, v[i]);
}

__global__ __launch_bounds__(256) void k_bn_stats(BnArgs a) {
  __shared__ f32x4 sh[256];
  const int C = a.d.C;
  const long N = a.d.N;
  const BnMap m = bn_map(C);
  const long rows_per_block = (N + gridDim.x - 1) / gridDim.x;
  const long lo = (long)blockIdx.x * rows_per_block;
  const long hi = lo + rows_per_block < N ? lo + rows_per_block : N;
  const int c = 4 * m.q;
  const f32x4 shift = ld4(a.x + c);                    // voxel 0 of these channels
  f32x4 s1 = f32x4{0.f, 0.f, 0.f, 0.f}, s2 = s1;
  long row = lo + m.row0;
  const long st = m.rstep;
  for (; row + 3 * st < hi; row += 4 * st) {          // four independent 16-byte loads in flight per thread
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = ld4(a.x + (row + u * st) * C + c);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const f32x4 d = v[u] - shift;
      s1 += d;
      s2 += d * d;
    }
  }
  for (; row < hi; row += st) {
    const f32x4 v = ld4(a.x + row * C + c) - shift;
    s1 += v;
    s2 += v * v;
  }
  s1 = bn_block_sum(s1, sh, m);
  s2 = bn_block_sum(s2, sh, m);
  if (m.row0 == 0 && a.d.stats_mode) {
    // double format (round 4, shared with the convolution epilogues of conv3d_fused.hip): sum of x and of x^2, the shifted
    // block sums converted in fp64 (exact to fp32 rounding of the shifted sums whatever the mean / deviation ratio is)
    double* rep = reinterpret_cast<double*>(a.sums) + (size_t)(blockIdx.x % STPDE_BN_REP) * 2 * C;
    const double n = hi > lo ? (double)(hi - lo) : 0.;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const double sh = shift[i], t1 = s1[i], t2 = s2[i];
      if (a.d.det) {       // order-independent long accumulators, one replica (common.h)
        long long* acc = reinterpret_cast<long long*>(a.sums);
        det_add_f64(acc + (size_t)(c + i) * STPDE_DET_K, t1 + n * sh);
        det_add_f64(acc + ((size_t)C + c + i) * STPDE_DET_K, t2 + 2. * sh * t1 + n * sh * sh);
      } else {
        atomicAdd(rep + c + i, t1 + n * sh);
        atomicAdd(rep + C + c + i, t2 + 2. * sh * t1 + n * sh * sh);
      }
    }
  } else if (m.row0 == 0) {
    // the partial sums of a block go to replica blockIdx % STPDE_BN_REP: with one copy, ~1000 blocks queue up on the same 2 C
    // addresses in the L2 atomic units (that, not HBM, bounded this kernel: 62 us for 67 MB)
    float* rep = a.sums + (size_t)(blockIdx.x % STPDE_BN_REP) * 3 * C;
    atomic_add4(rep + C + c, s1);
    atomic_add4(rep + 2 * C + c, s2);
    if (blockIdx.x == 0) st4(a.sums + c, shift);
  }
}

__global__ __launch_bounds__(256) void k_bn_apply(BnArgs a) {
  const int C = a.d.C;
  const long N = a.d.N;
  const BnMap m = bn_map(C);
  const long rows_per_block = (N + gridDim.x - 1) / gridDim.x;
  const long lo = (long)blockIdx.x * rows_per_block;
  const long hi = lo + rows_per_block < N ? lo + rows_per_block : N;
  const int c = 4 * m.q;
  f32x4 mean, var;
  if (a.d.training && a.d.stats_mode) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float mi, vi, ri;
      bn_stat_f64(reinterpret_cast<const double*>(a.sums), C, c + i, N, a.d.eps, a.d.det, mi, vi, ri);
      mean[i] = mi;
      var[i] = vi;
    }
  } else if (a.d.training) {
    f32x4 t1 = f32x4{0.f, 0.f, 0.f, 0.f}, t2 = t1;
    for (int r = 0; r < STPDE_BN_REP; ++r) {          // fixed order: every block gets the same sums
      t1 += ld4(a.sums + ((size_t)r * 3 + 1) * C + c);
      t2 += ld4(a.sums + ((size_t)r * 3 + 2) * C + c);
    }
    const f32x4 shift = ld4(a.sums + c), e1 = t1 / (float)N, e2 = t2 / (float)N;
    mean = shift + e1;
    var = e2 - e1 * e1;                               // biased variance of the batch
#pragma unroll
    for (int i = 0; i < 4; ++i) var[i] = fmaxf(var[i], 0.f);
  } else {
    mean = ld4(a.running_mean + c);
    var = ld4(a.running_var + c);
  }
  f32x4 rstd;
#pragma unroll
  for (int i = 0; i < 4; ++i) rstd[i] = 1.f / sqrtf(var[i] + a.d.eps);
  if (blockIdx.x == 0 && m.row0 == 0) {
    st4(a.stat + c, mean);
    st4(a.stat + C + c, rstd);
    if (a.d.training && a.running_mean) {           // torch: running = (1 - momentum) * running + momentum * batch
      const f32x4 unb = N > 1 ? var * ((float)N / (float)(N - 1)) : var;
      st4(a.running_mean + c, (1.f - a.d.momentum) * ld4(a.running_mean + c) + a.d.momentum * mean);
      st4(a.running_var + c, (1.f - a.d.momentum) * ld4(a.running_var + c) + a.d.momentum * unb);
    }
  }
  const f32x4 one = f32x4{1.f, 1.f, 1.f, 1.f}, zero = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 g = a.gamma ? ld4(a.gamma + c) : one, b = a.beta ? ld4(a.beta + c) : zero;
  const f32x4 scale = rstd * g;
  for (long row = lo + m.row0; row < hi; row += m.rstep) {
    const long i = row * C + c;
    f32x4 v = (ld4(a.x + i) - mean) * scale + b;
    if (a.r) v += ld4(a.r + i);
    if (a.d.relu) {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : 0.f;
    }
    st4(a.y + i, v);
  }
}

__global__ __launch_bounds__(256) void k_bn_bwd_reduce(BnArgs a) {
  __shared__ f32x4 sh[256];
  const int C = a.d.C;
  const long N = a.d.N;
  const BnMap m = bn_map(C);
  const long rows_per_block = (N + gridDim.x - 1) / gridDim.x;
  const long lo = (long)blockIdx.x * rows_per_block;
  const long hi = lo + rows_per_block < N ? lo + rows_per_block : N;
  const int c = 4 * m.q;
  const f32x4 mean = ld4(a.stat + c), rstd = ld4(a.stat + C + c);
  f32x4 s1 = f32x4{0.f, 0.f, 0.f, 0.f}, s2 = s1;
  const bool relu = a.d.relu != 0;
  auto term = [&](f32x4 dz, f32x4 y, f32x4 x) {
    if (relu) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (!(y[k] > 0.f)) dz[k] = 0.f;
    }
    s1 += dz;
    s2 += dz * (x - mean) * rstd;
  };
  const long st = m.rstep;
  auto four = [&](long row) {                         // 8-12 independent 16-byte loads in flight per thread
    f32x4 dz[4], y[4], x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long i = (row + u * st) * C + c;
      dz[u] = ld4(a.dy + i);
      x[u] = ld4(a.x + i);
      y[u] = relu ? ld4(a.y + i) : dz[u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) term(dz[u], y[u], x[u]);
  };
  auto one_row = [&](long row) {
    const long i = row * C + c;
    const f32x4 dz = ld4(a.dy + i);
    term(dz, relu ? ld4(a.y + i) : dz, ld4(a.x + i));
  };
  if (a.il) {
    // chunk k = rows [4 st k, 4 st (k + 1)): consecutive blocks read consecutive 16 KiB of every array
    for (long base = (long)blockIdx.x * 4 * st; base < N; base += (long)gridDim.x * 4 * st) {
      if (base + 4 * st <= N) {
        four(base + m.row0);
      } else {
        for (long row = base + m.row0; row < N; row += st) one_row(row);
      }
    }
  } else {
    long row = lo + m.row0;
    for (; row + 3 * st < hi; row += 4 * st) four(row);
    for (; row < hi; row += st) one_row(row);
  }
  s1 = bn_block_sum(s1, sh, m);
  s2 = bn_block_sum(s2, sh, m);
  if (m.row0 == 0 && a.d.det) {
    long long* acc = reinterpret_cast<long long*>(a.bsum);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      det_add_f32(acc + (size_t)(c + i) * STPDE_DET_K, s1[i]);
      det_add_f32(acc + ((size_t)C + c + i) * STPDE_DET_K, s2[i]);
    }
  } else if (m.row0 == 0) {
    float* rep = a.bsum + (size_t)(blockIdx.x % STPDE_BN_REP) * 2 * C;
    atomic_add4(rep + c, s1);
    atomic_add4(rep + C + c, s2);
  }
}

__global__ __launch_bounds__(256) void k_bn_bwd_apply(BnArgs a) {
  const int C = a.d.C;
  const long N = a.d.N;
  const BnMap m = bn_map(C);
  const long rows_per_block = (N + gridDim.x - 1) / gridDim.x;
  const long lo = (long)blockIdx.x * rows_per_block;
  const long hi = lo + rows_per_block < N ? lo + rows_per_block : N;
  const int c = 4 * m.q;
  const f32x4 mean = ld4(a.stat + c), rstd = ld4(a.stat + C + c);
  f32x4 sdz = f32x4{0.f, 0.f, 0.f, 0.f}, sdzx = sdz;
  if (a.d.det) {
    const long long* acc = reinterpret_cast<const long long*>(a.bsum);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      sdz[i] = (float)det_value(acc + (size_t)(c + i) * STPDE_DET_K);
      sdzx[i] = (float)det_value(acc + ((size_t)C + c + i) * STPDE_DET_K);
    }
  } else {
    for (int r = 0; r < STPDE_BN_REP; ++r) {
      sdz += ld4(a.bsum + ((size_t)r * 2) * C + c);
      sdzx += ld4(a.bsum + ((size_t)r * 2 + 1) * C + c);
    }
  }
  if (blockIdx.x == 0 && m.row0 == 0) {
    if (a.dbeta) st4(a.dbeta + c, sdz);
    if (a.dgamma) st4(a.dgamma + c, sdzx);
  }
  const f32x4 one = f32x4{1.f, 1.f, 1.f, 1.f}, zero = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 g = a.gamma ? ld4(a.gamma + c) : one;
  const f32x4 k1 = a.d.training ? sdz / (float)N : zero, k2 = a.d.training ? sdzx / (float)N : zero;
  const long st = m.rstep;
  const bool relu = a.d.relu != 0;
  auto finish = [&](long i, f32x4 dz, f32x4 y, f32x4 x) {
    if (relu) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (!(y[k] > 0.f)) dz[k] = 0.f;
    }
    if (a.dr) st4(a.dr + i, dz);
    if (a.dx) st4(a.dx + i, g * rstd * (dz - k1 - (x - mean) * rstd * k2));
  };
  auto four = [&](long row) {
    f32x4 dz[4], y[4], x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long i = (row + u * st) * C + c;
      dz[u] = ld4(a.dy + i);
      y[u] = relu ? ld4(a.y + i) : dz[u];
      x[u] = a.dx ? ld4(a.x + i) : dz[u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) finish((row + u * st) * C + c, dz[u], y[u], x[u]);
  };
  auto one_row = [&](long row) {
    const long i = row * C + c;
    const f32x4 dz = ld4(a.dy + i);
    finish(i, dz, relu ? ld4(a.y + i) : dz, a.dx ? ld4(a.x + i) : dz);
  };
  if (a.il) {
    for (long base = (long)blockIdx.x * 4 * st; base < N; base += (long)gridDim.x * 4 * st) {
      if (base + 4 * st <= N) {
        four(base + m.row0);
      } else {
        for (long row = base + m.row0; row < N; row += st) one_row(row);
      }
    }
  } else {
    long row = lo + m.row0;
    for (; row + 3 * st < hi; row += 4 * st) four(row);
    for (; row < hi; row += st) one_row(row);
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
  const long rows_per_pass = 256 / (d->C / 4);
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
  if (d->det && d->training && d->stats_mode == 0) {
    stpde_set_error("bn_fwd: the deterministic mode takes its statistics in the double format (stats_mode 1 or 2)");
    return STPDE_E_UNSUPPORTED;
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
  // the reduction kernels end in 8 atomics per channel quad and block: a quarter of the blocks (>= 4 waves per SIMD still)
  const unsigned rgrid = grid > 1024 ? 1024 : grid;
  if (d->training && d->stats_mode != 2) {
    const size_t bytes = d->stats_mode ? (size_t)STPDE_BN_REP * 2 * d->C * sizeof(double) : (size_t)STPDE_BN_REP * 3 * d->C * sizeof(float);
    if (!d->scratch_zeroed) (void)hipMemsetAsync(sums, 0, bytes, (hipStream_t)stream);
    STPDE_LAUNCH(k_bn_stats, dim3(rgrid), dim3(256), 0, (hipStream_t)stream, a);
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
  const int rg_env = 1024;      // grid of the reduction pass
  a.il = 1;                     // interleaved apply pass (-5 ... -8 % on the 4.2 M-voxel levels)
  const unsigned grid = bn_grid(d);
  if (!d->reduce_done) {
    if (!d->scratch_zeroed) (void)hipMemsetAsync(bsum, 0, (size_t)STPDE_BN_REP * 2 * d->C * sizeof(float), (hipStream_t)stream);
    STPDE_LAUNCH(k_bn_bwd_reduce, dim3(grid > (unsigned)rg_env ? (unsigned)rg_env : grid), dim3(256), 0, (hipStream_t)stream, a);
    rc = stpde_check_launch("k_bn_bwd_reduce");
    if (rc) return rc;
  }
  STPDE_LAUNCH(k_bn_bwd_apply, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  return stpde_check_launch("k_bn_bwd_apply");
}

// statistics of x in the double format, for stpde_conv3d_fused when the convolution ran in the tap-split kernel (its epilogue
// holds partial sums only); sums zero-filled by the caller
int stpde_bn_stats_f64(const float* x, long N, int C, double* sums, hipStream_t stream) {
  stpde_bn_desc d{};
  d.N = N;
  d.C = C;
  d.training = 1;
  d.stats_mode = 1;
  int rc = bn_check(&d);
  if (rc) return rc;
  BnArgs a{};
  a.d = d;
  a.x = x;
  a.sums = reinterpret_cast<float*>(sums);
  const unsigned grid = bn_grid(&d);
  STPDE_LAUNCH(k_bn_stats, dim3(grid > 1024 ? 1024 : grid), dim3(256), 0, stream, a);
  return stpde_check_launch("k_bn_stats");
}


// ---- deterministic mode: long accumulators -> fp32 (include/stpde_hip.h: stpde_det_finalize) -------------------------------
__global__ __launch_bounds__(256) void k_det_finalize(const long long* acc, long n, float* out) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    out[i] = (float)det_value(acc + (size_t)i * STPDE_DET_K);
}

extern "C" int stpde_det_finalize(const void* acc, long n, float* out, void* stream) {
  if (!acc || !out || n < 0) {
    stpde_set_error("det_finalize: bad argument");
    return STPDE_E_BADARG;
  }
  if (n == 0) return STPDE_OK;
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  STPDE_LAUNCH(k_det_finalize, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const long long*>(acc), n, out);
  return stpde_check_launch("k_det_finalize");
}
