// Plain multilinear interpolation on a regular grid for dim = 1..4 (any channel count):
// regular_nd_grid_interpolation / ..._coefficients (src/regular_nd_grid_interpolation.py:14-104) and the
// scatter-add backward of its gather (:65-66).  HBM/L2-bound gather: one thread per (point, channel), lanes
// run along the channel axis so each corner read is a contiguous C*4-byte segment.
#include "common.h"

struct InterpArgs {
  stpde_interp_desc d;
  const float* grid;
  const float* pts;
  float* out;
  float* cv;
  float* wts;
  float* rel;
  const float* out_bar;
  const float* cv_bar;
  float* dgrid;
};

struct GeomN {
  float om[2][4], rl[2][4];
  int i0[4];
};

__device__ __forceinline__ GeomN geom_nd(const InterpArgs& a, int p) {
  GeomN gm;
  const int dim = a.d.dim;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (k < dim) {
      const float x = a.pts[(size_t)p * dim + k];
      const float q = fmaxf(fminf(x, a.d.hi_c[k]), a.d.lo_c[k]);
      const float cs = a.d.cube[k];
      int i0 = (int)floorf(q / cs);
      i0 = i0 < 0 ? 0 : (i0 > a.d.n[k] - 2 ? a.d.n[k] - 2 : i0);
      const float i0f = (float)i0;
      const float p0 = i0f * cs, p1 = (i0f + 1.f) * cs;
      gm.i0[k] = i0;
      gm.om[0][k] = fabsf(q - p1) / cs;
      gm.om[1][k] = fabsf(q - p0) / cs;
      gm.rl[0][k] = (q - p0) / cs;
      gm.rl[1][k] = (q - p1) / cs;
    } else {
      gm.i0[k] = 0;
      gm.om[0][k] = gm.om[1][k] = 1.f;
      gm.rl[0][k] = gm.rl[1][k] = 0.f;
    }
  }
  return gm;
}

template <bool BWD>
__global__ __launch_bounds__(256) void k_interp(InterpArgs a) {
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int C = a.d.C, dim = a.d.dim;
  if (gid >= (size_t)a.d.P * C) return;
  const int p = gid / C, c = gid % C;
  const GeomN gm = geom_nd(a, p);
  const int b = p / a.d.N;
  const int nc = 1 << dim;
  float acc = 0.f;
  const float ob = (BWD && a.out_bar) ? a.out_bar[(size_t)p * C + c] : 0.f;
  for (int j = 0; j < nc; ++j) {
    size_t node = b;
    float w = 1.f;
    float rl[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < dim) {
        const int bit = (j >> (dim - 1 - k)) & 1;  // first dim most significant (:55-56)
        node = node * a.d.n[k] + gm.i0[k] + bit;
        const float o = bit ? gm.om[1][k] : gm.om[0][k];
        w = (k == 0) ? o : w * o;
        rl[k] = bit ? gm.rl[1][k] : gm.rl[0][k];
      }
    }
    if (!BWD) {
      const float v = a.grid[node * C + c];
      acc += v * w;
      if (a.cv) a.cv[((size_t)p * nc + j) * C + c] = v;
      if (c == 0) {
        if (a.wts) a.wts[(size_t)p * nc + j] = w;
        if (a.rel)
          for (int k = 0; k < dim; ++k) a.rel[((size_t)p * nc + j) * dim + k] = rl[k];
      }
    } else {
      float gsum = ob * w;
      if (a.cv_bar) gsum += a.cv_bar[((size_t)p * nc + j) * C + c];
      atomicAdd(a.dgrid + node * C + c, gsum);
    }
  }
  if (!BWD && a.out) a.out[(size_t)p * C + c] = acc;
}

static int check_interp(const stpde_interp_desc* d, const float* pts) {
  if (!d || !pts || d->P <= 0 || d->N <= 0 || d->B <= 0 || d->dim < 1 || d->dim > 4 || d->C < 1) {
    stpde_set_error("interp: bad argument (dim 1..4)");
    return STPDE_E_BADARG;
  }
  for (int k = 0; k < d->dim; ++k)
    if (d->n[k] < 2) {
      stpde_set_error("interp: grid needs >= 2 nodes per dim");
      return STPDE_E_BADARG;
    }
  return STPDE_OK;
}

extern "C" int stpde_interp_fwd(const stpde_interp_desc* d, const float* grid, const float* pts, float* out,
                                float* corner_values, float* weights, float* x_relative, void* stream) {
  int rc = check_interp(d, pts);
  if (rc) return rc;
  if (!grid) {
    stpde_set_error("interp_fwd: null grid");
    return STPDE_E_BADARG;
  }
  InterpArgs a{};
  a.d = *d;
  a.grid = grid;
  a.pts = pts;
  a.out = out;
  a.cv = corner_values;
  a.wts = weights;
  a.rel = x_relative;
  const size_t n = (size_t)d->P * d->C;
  STPDE_LAUNCH(k_interp<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return stpde_check_launch("k_interp_fwd");
}

extern "C" int stpde_interp_bwd_grid(const stpde_interp_desc* d, const float* pts, const float* out_bar,
                                     const float* corner_bar, float* dgrid, void* stream) {
  int rc = check_interp(d, pts);
  if (rc) return rc;
  if (!dgrid || (!out_bar && !corner_bar)) {
    stpde_set_error("interp_bwd_grid: null pointer");
    return STPDE_E_BADARG;
  }
  InterpArgs a{};
  a.d = *d;
  a.pts = pts;
  a.out_bar = out_bar;
  a.cv_bar = corner_bar;
  a.dgrid = dgrid;
  const size_t n = (size_t)d->P * d->C;
  STPDE_LAUNCH(k_interp<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return stpde_check_launch("k_interp_bwd");
}
