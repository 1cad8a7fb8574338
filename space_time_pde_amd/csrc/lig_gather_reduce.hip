// Local-implicit-grid glue kernels for dim = 3:
//   k_gather      clip + cell index + corner gather + relative coords  -> augmented MLP input X (fragment layout)
//                 and per-point interpolation coefficients               (src/regular_nd_grid_interpolation.py:48-76,
//                                                                         src/local_implicit_grid.py:49)
//   k_reduce_fwd  corner-weighted sum of the MLP output on every derivative stream (src/local_implicit_grid.py:59
//                 and what the src/pde.py:8-9 sweeps differentiate through)
//   k_reduce_bwd  its adjoint
//   k_xbar        d latent: xbar = sum_l W_s,l^T abar_l, scatter-added at the 8 corner nodes (backward of the
//                 advanced-index gather at src/regular_nd_grid_interpolation.py:65-66)
// All of these are HBM/L2-bound byte movers; no MFMA except the small xbar GEMM.
#include "common.h"

struct PointGeom {
  float om[2][3];   // omega[b][d] = |q_d - pos_opposite| / cube_d
  float dom[2][3];  // d omega / d q_d (sign * s_d / cube_d)
  float rel[2][3];  // (q_d - pos) / cube_d
  float kap[3];     // s_d / cube_d
  int i0[3];
};

// Same fp32 expression sequence as the reference (division, no reciprocal, no contraction).
__device__ __forceinline__ PointGeom point_geom(const float* p, const float* lo_c, const float* hi_c,
                                                const float* cube, const int* n) {
  PointGeom gm;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float x = p[d];
    const float m = fminf(x, hi_c[d]);
    const float q = fmaxf(m, lo_c[d]);
    // derivative of the clip: min/max backward split ties evenly (quirk a-Q2)
    const float s = (x < hi_c[d] ? 1.f : (x == hi_c[d] ? 0.5f : 0.f)) * (m > lo_c[d] ? 1.f : (m == lo_c[d] ? 0.5f : 0.f));
    const float cs = cube[d];
    const float fl = floorf(q / cs);
    int i0 = (int)fl;
    i0 = i0 < 0 ? 0 : (i0 > n[d] - 2 ? n[d] - 2 : i0);  // only NaN / out-of-box inputs ever hit the clamp
    const float i0f = (float)i0;
    const float p0 = i0f * cs;
    const float p1 = (i0f + 1.f) * cs;
    gm.i0[d] = i0;
    gm.kap[d] = s / cs;
    // bit 0: pos = p0, opposite = p1 ; bit 1: pos = p1, opposite = p0
    const float t0 = q - p1, t1 = q - p0;
    gm.om[0][d] = fabsf(t0) / cs;
    gm.om[1][d] = fabsf(t1) / cs;
    gm.dom[0][d] = (t0 > 0.f ? 1.f : (t0 < 0.f ? -1.f : 0.f)) / cs * s;
    gm.dom[1][d] = (t1 > 0.f ? 1.f : (t1 < 0.f ? -1.f : 0.f)) / cs * s;
    gm.rel[0][d] = (q - p0) / cs;
    gm.rel[1][d] = (q - p1) / cs;
  }
  return gm;
}

struct GatherArgs {
  stpde_gather_desc d;
  const float* pts;
  const float* latent;
  float* X;
  float* XR;
  float* coef;
  int* cell;
  float* cw;
};

// Clipped coordinate and cell index of one dimension (the part of point_geom every lane needs): the same fp32 expression
// sequence, so the index is bit-identical.
__device__ __forceinline__ int cell_index_1d(float x, float lo, float hi, float cs, int n, float& q) {
  q = fmaxf(fminf(x, hi), lo);
  int i0 = (int)floorf(q / cs);
  return i0 < 0 ? 0 : (i0 > n - 2 ? n - 2 : i0);        // only NaN / out-of-box inputs ever hit the clamp
}

// one thread per (tile, xt, lane): writes one float4 of X.
// Round 3: a lane evaluates only what ITS slot needs -- the three IEEE divisions of the cell index always, the relative
// coordinates only in the 16 lanes that hold them (tile 0, lane group 0), the full geometry (weights, weight derivatives,
// clip derivative: 21 more divisions) only in the one lane per point that writes the coefficients (the first version ran
// point_geom in all 192 lanes of a row tile).  Same results bit for bit; the launch time did not move (1.45 ms per 2^20
// points either way): the kernel is bound by its 6 KiB of X / XR stores per point, not by the divisions.
__global__ __launch_bounds__(256) void k_gather(GatherArgs a) {
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int ntiles = a.d.P / 2;
  if (gid >= (size_t)ntiles * XT * 64) return;
  const int lane = gid & 63;
  const int xt = (gid >> 6) % XT;
  const int tile = (gid >> 6) / XT;
  const int g = lane >> 4, j = lane & 15;
  const int p = tile * 2 + (j >> 3);
  const int corner = j & 7;
  const int bit[3] = {(corner >> 2) & 1, (corner >> 1) & 1, corner & 1};
  const int n[3] = {a.d.n0, a.d.n1, a.d.n2};
  float pt[3] = {a.pts[(size_t)p * 3], a.pts[(size_t)p * 3 + 1], a.pts[(size_t)p * 3 + 2]};
  float q[3];
  int i0[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) i0[d] = cell_index_1d(pt[d], a.d.lo_c[d], a.d.hi_c[d], a.d.cube[d], n[d], q[d]);
  int b = (a.d.p_base + p) / a.d.N;
  b = b > a.d.B - 1 ? a.d.B - 1 : b;
  const size_t node0 = (((size_t)b * n[0] + i0[0]) * n[1] + i0[1]) * n[2] + i0[2];
  const size_t node = node0 + ((size_t)bit[0] * n[1] + bit[1]) * n[2] + bit[2];
  const int C = a.d.C;
  f32x4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    // feature held by slot (xt, g, r): tiles 0 / 1 in order, tile 2 sparse (register 0 only: features 32 + g)
    const int f = xt < XT - 1 ? 16 * xt + 4 * g + r : (r == 0 ? 16 * xt + g : 16 * XT);
    float val = 0.f;
    if (f < 3) {
      // relative coordinate of this corner along dimension f: (q - pos) / cube, pos = (i0 + bit) * cube (reference :69-76)
      const float cs = sel3(f, a.d.cube[0], a.d.cube[1], a.d.cube[2]);
      const float qq = sel3(f, q[0], q[1], q[2]);
      const int ii = f == 0 ? i0[0] : (f == 1 ? i0[1] : i0[2]);
      const int bb = f == 0 ? bit[0] : (f == 1 ? bit[1] : bit[2]);
      const float i0f = (float)ii;
      const float pos = bb ? (i0f + 1.f) * cs : i0f * cs;
      val = (qq - pos) / cs;
    } else if (f < 3 + C)
      val = a.latent[node * C + (f - 3)];
    else if (f == 3 + C)
      val = 1.f;  // bias column
    v[r] = val;
  }
  st4(a.X + gid * 4, v);
  if (a.XR) st_R(a.XR + (gid >> 6) * 256, lane, v);
  if (xt == 0 && g == 0 && corner == 0) {
    const PointGeom gm = point_geom(pt, a.d.lo_c, a.d.hi_c, a.d.cube, n);
    float* cf = a.coef + (size_t)p * 16;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      cf[d] = gm.om[0][d];
      cf[3 + d] = gm.om[1][d];
      cf[6 + d] = gm.dom[0][d];
      cf[9 + d] = gm.dom[1][d];
      cf[12 + d] = gm.kap[d];
    }
    cf[15] = 0.f;
    a.cell[p] = (int)node0;
    if (a.cw) {  // weights of the combined second-order stream: alpha_k * kappa_a * kappa_b over the canonical pairs
      float* cwp = a.cw + (size_t)p * 8;
      const float k0 = gm.kap[0], k1 = gm.kap[1], k2 = gm.kap[2];
      cwp[0] = a.d.alpha[0] * k0 * k0;
      cwp[1] = a.d.alpha[1] * k0 * k1;
      cwp[2] = a.d.alpha[2] * k0 * k2;
      cwp[3] = a.d.alpha[3] * k1 * k1;
      cwp[4] = a.d.alpha[4] * k1 * k2;
      cwp[5] = a.d.alpha[5] * k2 * k2;
      cwp[6] = 0.f;
      cwp[7] = 0.f;
    }
  }
}

// The same kernel for the reference's 32 latent channels, one WAVE per row tile (round 3): a lane (g, j) fetches channels
// 8g .. 8g+7 of the grid node of corner row j as two aligned 16-byte loads (the first version: four dword loads per lane,
// unaligned by the three coordinate features in front of the channels), the 16 x 36 feature rows of the tile go through an
// LDS patch, and BOTH fragment images of the three input tiles leave as coalesced 16-byte stores (the row-major image used
// to be four strided dword stores per lane).  Same expressions for the cell index / relative coordinates, bit for bit.
__global__ __launch_bounds__(256) void k_gather_tile(GatherArgs a) {
  constexpr int RS = 36;                       // floats per row: features 0..35 = rel(3), 32 channels, the ones column
  __shared__ __attribute__((aligned(16))) float rows[4][16 * RS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ntiles = a.d.P / 2;
  const int tile = blockIdx.x * 4 + wv;
  const bool live = tile < ntiles;
  const int g = lane >> 4, j = lane & 15;
  float* rw = rows[wv];
  if (live) {
    const int p = tile * 2 + (j >> 3);
    const int corner = j & 7;
    const int bit[3] = {(corner >> 2) & 1, (corner >> 1) & 1, corner & 1};
    const int n[3] = {a.d.n0, a.d.n1, a.d.n2};
    float pt[3] = {a.pts[(size_t)p * 3], a.pts[(size_t)p * 3 + 1], a.pts[(size_t)p * 3 + 2]};
    float q[3];
    int i0[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) i0[d] = cell_index_1d(pt[d], a.d.lo_c[d], a.d.hi_c[d], a.d.cube[d], n[d], q[d]);
    int b = (a.d.p_base + p) / a.d.N;
    b = b > a.d.B - 1 ? a.d.B - 1 : b;
    const size_t node0 = (((size_t)b * n[0] + i0[0]) * n[1] + i0[1]) * n[2] + i0[2];
    const size_t node = node0 + ((size_t)bit[0] * n[1] + bit[1]) * n[2] + bit[2];
    const f32x4 c0 = ld4(a.latent + node * 32 + 8 * g), c1 = ld4(a.latent + node * 32 + 8 * g + 4);
    float* rj = rw + j * RS;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      rj[3 + 8 * g + k] = c0[k];
      rj[3 + 8 * g + 4 + k] = c1[k];
    }
    if (g == 0) {
#pragma unroll
      for (int f = 0; f < 3; ++f) {
        // relative coordinate of this corner along dimension f: (q - pos) / cube, pos = (i0 + bit) * cube (reference :69-76)
        const float cs = a.d.cube[f];
        const float i0f = (float)i0[f];
        const float pos = bit[f] ? (i0f + 1.f) * cs : i0f * cs;
        rj[f] = (q[f] - pos) / cs;
      }
      rj[35] = 1.f;                            // bias column
      if (corner == 0) {
        const PointGeom gm = point_geom(pt, a.d.lo_c, a.d.hi_c, a.d.cube, n);
        float* cf = a.coef + (size_t)p * 16;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          cf[d] = gm.om[0][d];
          cf[3 + d] = gm.om[1][d];
          cf[6 + d] = gm.dom[0][d];
          cf[9 + d] = gm.dom[1][d];
          cf[12 + d] = gm.kap[d];
        }
        cf[15] = 0.f;
        a.cell[p] = (int)node0;
        if (a.cw) {  // weights of the combined second-order stream: alpha_k * kappa_a * kappa_b over the canonical pairs
          float* cwp = a.cw + (size_t)p * 8;
          const float k0 = gm.kap[0], k1 = gm.kap[1], k2 = gm.kap[2];
          cwp[0] = a.d.alpha[0] * k0 * k0;
          cwp[1] = a.d.alpha[1] * k0 * k1;
          cwp[2] = a.d.alpha[2] * k0 * k2;
          cwp[3] = a.d.alpha[3] * k1 * k1;
          cwp[4] = a.d.alpha[4] * k1 * k2;
          cwp[5] = a.d.alpha[5] * k2 * k2;
          cwp[6] = 0.f;
          cwp[7] = 0.f;
        }
      }
    }
  }
  __syncthreads();
  if (!live) return;
  float* xo = a.X + (size_t)tile * XT * 256 + lane * 4;
  // column-major images: lane (g, j) = features 4g..4g+3 of row j; tile 2 is sparse (register 0: feature 32 + g)
  st4(xo, ld4(rw + j * RS + 4 * g));
  st4(xo + 256, ld4(rw + j * RS + 16 + 4 * g));
  st4(xo + 512, f32x4{rw[j * RS + 32 + g], 0.f, 0.f, 0.f});
  if (a.XR) {
    // row-major images: lane (g, c) = rows 4g..4g+3 of slot c
    float* ro = a.XR + (size_t)tile * XT * 256 + lane * 4;
    const float* r0 = rw + 4 * g * RS;
    st4(ro, f32x4{r0[j], r0[RS + j], r0[2 * RS + j], r0[3 * RS + j]});
    st4(ro + 256, f32x4{r0[16 + j], r0[RS + 16 + j], r0[2 * RS + 16 + j], r0[3 * RS + 16 + j]});
    const int fs = 32 + (j >> 2);
    const bool has = (j & 3) == 0;
    st4(ro + 512, has ? f32x4{r0[fs], r0[RS + fs], r0[2 * RS + fs], r0[3 * RS + fs]} : f32x4{0.f, 0.f, 0.f, 0.f});
  }
}

extern "C" int stpde_lig_gather(const stpde_gather_desc* d, const float* pts, const float* latent, float* X,
                                float* XR, float* coef, int* cell, float* cw, void* stream) {
  if (!d || d->P <= 0 || (d->P & 1) || d->N <= 0 || d->p_base < 0 || d->B <= 0 || d->n0 < 2 || d->n1 < 2 || d->n2 < 2 || d->C < 1 ||
      3 + d->C + 1 > 16 * (XT - 1) + 4 || !pts || !latent || !X || !coef || !cell) {
    stpde_set_error("lig_gather: bad argument (P even, grid >= 2 per dim, C <= %d)", 16 * (XT - 1));
    return STPDE_E_BADARG;
  }
  if ((size_t)d->B * d->n0 * d->n1 * d->n2 >= (1u << 31)) {
    stpde_set_error("lig_gather: latent grid too large for int32 node index");
    return STPDE_E_BADARG;
  }
  GatherArgs a{*d, pts, latent, X, XR, coef, cell, cw};
  if (d->C == 32) {       // the reference's latent width: one wave per row tile, 16-byte loads and stores
    STPDE_LAUNCH(k_gather_tile, dim3((unsigned)((d->P / 2 + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
    return stpde_check_launch("k_gather_tile");
  }
  const size_t nthreads = (size_t)(d->P / 2) * XT * 64;
  STPDE_LAUNCH(k_gather, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return stpde_check_launch("k_gather");
}

// ---------------------------------------------------------------------------------------------------------
// corner reduction
// ---------------------------------------------------------------------------------------------------------
struct CornerW {
  float w, dw[3], ddw[3];  // ddw index: pair (0,1)->0, (0,2)->1, (1,2)->2
};

__device__ __forceinline__ CornerW corner_weights(const float* cf, int corner) {
  const int bit[3] = {(corner >> 2) & 1, (corner >> 1) & 1, corner & 1};
  float om[3], dm[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    om[d] = bit[d] ? cf[3 + d] : cf[d];
    dm[d] = bit[d] ? cf[9 + d] : cf[6 + d];
  }
  CornerW c;
  c.w = om[0] * om[1] * om[2];
  c.dw[0] = dm[0] * om[1] * om[2];
  c.dw[1] = om[0] * dm[1] * om[2];
  c.dw[2] = om[0] * om[1] * dm[2];
  c.ddw[0] = dm[0] * dm[1] * om[2];
  c.ddw[1] = dm[0] * om[1] * dm[2];
  c.ddw[2] = om[0] * dm[1] * dm[2];
  return c;
}

__device__ __forceinline__ float pair_ddw(const CornerW& c, int d, int e) {
  if (d == e) return 0.f;
  const int s = d + e;  // 1 -> (0,1), 2 -> (0,2), 3 -> (1,2)
  return s == 1 ? c.ddw[0] : (s == 2 ? c.ddw[1] : c.ddw[2]);
}

struct ReduceArgs {
  stpde_jet_cfg cfg;
  int P, n_out;
  int S_mlp;   // streams present in the layer buffers (piecewise-linear activations carry no second-order streams)
  long ldp;
  const float* src;  // fwd: out_pre [tile][S][1][256]; bwd: jets_bar [S][n_out][P]
  const float* coef;
  float* dst;  // fwd: jets [S][n_out][P]; bwd: abar_out [tile][S][1][256]
};

// one thread per (point, channel)
__global__ __launch_bounds__(256) void k_reduce_fwd(ReduceArgs a) {
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (size_t)a.P * a.n_out) return;
  const int p = gid / a.n_out, ch = gid % a.n_out;
  const int S1 = a.cfg.S1, S2 = a.cfg.S2, S = 1 + S1 + S2;
  float cf[16];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f32x4 v = ld4(a.coef + (size_t)p * 16 + 4 * i);
    cf[4 * i] = v[0];
    cf[4 * i + 1] = v[1];
    cf[4 * i + 2] = v[2];
    cf[4 * i + 3] = v[3];
  }
  const float* kap = cf + 12;
  float y[10];
#pragma unroll
  for (int s = 0; s < 10; ++s) y[s] = 0.f;
  const int tile = p >> 1;
  for (int corner = 0; corner < 8; ++corner) {
    const int j = ((p & 1) << 3) | corner;
    const int lane = ((ch >> 2) << 4) | j;
    const float* base = a.src + (size_t)tile * a.S_mlp * 256 + lane * 4 + (ch & 3);
    float f[10];
#pragma unroll
    for (int s = 0; s < 10; ++s) f[s] = s < a.S_mlp ? base[(size_t)s * 256] : 0.f;
    CornerW c = corner_weights(cf, corner);
    y[0] += c.w * f[0];
    if (S1 == 3) {
#pragma unroll
      for (int d = 0; d < 3; ++d) y[1 + d] += c.dw[d] * f[0] + c.w * kap[d] * f[1 + d];
      if (a.cfg.combo) {
        // combined stream: L y = sum_k alpha_k d2y/dq_a dq_b ; the MLP stream f[4] already carries
        // sum_k alpha_k kappa_a kappa_b d2f/dr_a dr_b
        const float* al = a.cfg.alpha;
        float acc = c.w * f[4];
        acc += (al[1] * c.ddw[0] + al[2] * c.ddw[1] + al[4] * c.ddw[2]) * f[0];
        acc += 2.f * (al[0] * c.dw[0] * kap[0] * f[1] + al[3] * c.dw[1] * kap[1] * f[2] + al[5] * c.dw[2] * kap[2] * f[3]);
        acc += al[1] * (c.dw[0] * kap[1] * f[2] + c.dw[1] * kap[0] * f[1]);
        acc += al[2] * (c.dw[0] * kap[2] * f[3] + c.dw[2] * kap[0] * f[1]);
        acc += al[4] * (c.dw[1] * kap[2] * f[3] + c.dw[2] * kap[1] * f[2]);
        y[4] += acc;
      } else {
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        if (k < S2) {
          const int d = a.cfg.pair0[k], e = a.cfg.pair1[k];
          const float kd = sel3(d, kap[0], kap[1], kap[2]), ke = sel3(e, kap[0], kap[1], kap[2]);
          const float dwd = sel3(d, c.dw[0], c.dw[1], c.dw[2]), dwe = sel3(e, c.dw[0], c.dw[1], c.dw[2]);
          const float fd = sel3(d, f[1], f[2], f[3]), fe = sel3(e, f[1], f[2], f[3]);
          y[4 + k] += pair_ddw(c, d, e) * f[0] + dwd * ke * fe + dwe * kd * fd + c.w * kd * ke * f[4 + k];
        }
      }
      }
    }
  }
#pragma unroll
  for (int s = 0; s < 10; ++s)
    if (s < S) a.dst[((size_t)s * a.n_out + ch) * a.ldp + p] = y[s];
}

// one thread per (row tile, lane of its fragment block): lane (g, j) holds output features 4g..4g+3 of corner row j and
// writes ONE 16-byte store per stream (zeros for features >= n_out) -- the first version had one thread per feature and
// 4-byte stores scattered over the block (1.5 ms per 2^20-point launch for 2.7 GB)
__global__ __launch_bounds__(256) void k_reduce_bwd(ReduceArgs a) {
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (size_t)a.P * 32) return;
  const int lane = gid & 63;
  const int tile = gid >> 6;
  const int g = lane >> 4, j = lane & 15;
  const int p = tile * 2 + (j >> 3), corner = j & 7;
  const int S1 = a.cfg.S1, S2 = a.cfg.S2, S = 1 + S1 + S2;
  float* base = a.dst + (size_t)tile * a.S_mlp * 256 + lane * 4;
  f32x4 fbv[10];
#pragma unroll
  for (int s = 0; s < 10; ++s) fbv[s] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (4 * g < a.n_out) {
    float cf[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 v = ld4(a.coef + (size_t)p * 16 + 4 * i);
      cf[4 * i] = v[0];
      cf[4 * i + 1] = v[1];
      cf[4 * i + 2] = v[2];
      cf[4 * i + 3] = v[3];
    }
    const float* kap = cf + 12;
    const CornerW c = corner_weights(cf, corner);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ch = 4 * g + r;
      if (ch >= a.n_out) continue;
      float fb[10];
#pragma unroll
      for (int s = 0; s < 10; ++s) fb[s] = 0.f;
      float yb[10];
#pragma unroll
      for (int s = 0; s < 10; ++s) yb[s] = s < S ? a.src[((size_t)s * a.n_out + ch) * a.ldp + p] : 0.f;
      fb[0] = c.w * yb[0];
      if (S1 == 3) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          fb[0] += c.dw[d] * yb[1 + d];
          fb[1 + d] = c.w * kap[d] * yb[1 + d];
        }
        if (a.cfg.combo) {
          const float* al = a.cfg.alpha;
          const float gg = yb[4];
          fb[0] += (al[1] * c.ddw[0] + al[2] * c.ddw[1] + al[4] * c.ddw[2]) * gg;
          fb[1] += (2.f * al[0] * c.dw[0] * kap[0] + al[1] * c.dw[1] * kap[0] + al[2] * c.dw[2] * kap[0]) * gg;
          fb[2] += (2.f * al[3] * c.dw[1] * kap[1] + al[1] * c.dw[0] * kap[1] + al[4] * c.dw[2] * kap[1]) * gg;
          fb[3] += (2.f * al[5] * c.dw[2] * kap[2] + al[2] * c.dw[0] * kap[2] + al[4] * c.dw[1] * kap[2]) * gg;
          fb[4] = c.w * gg;
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          if (k < S2 && !a.cfg.combo) {
            const int d = a.cfg.pair0[k], e = a.cfg.pair1[k];
            const float kd = sel3(d, kap[0], kap[1], kap[2]), ke = sel3(e, kap[0], kap[1], kap[2]);
            const float dwd = sel3(d, c.dw[0], c.dw[1], c.dw[2]), dwe = sel3(e, c.dw[0], c.dw[1], c.dw[2]);
            const float gg = yb[4 + k];
            fb[0] += pair_ddw(c, d, e) * gg;
            const float te = dwd * ke * gg;  // -> f_{1+e}
            const float td = dwe * kd * gg;  // -> f_{1+d}
            fb[1] += (e == 0 ? te : 0.f) + (d == 0 ? td : 0.f);
            fb[2] += (e == 1 ? te : 0.f) + (d == 1 ? td : 0.f);
            fb[3] += (e == 2 ? te : 0.f) + (d == 2 ? td : 0.f);
            fb[4 + k] = c.w * kd * ke * gg;
          }
        }
      }
#pragma unroll
      for (int s = 0; s < 10; ++s) fbv[s][r] = fb[s];
    }
  }
#pragma unroll
  for (int s = 0; s < 10; ++s)
    if (s < a.S_mlp) st4(base + (size_t)s * 256, fbv[s]);
}

static int check_reduce(const stpde_jet_cfg* cfg, int P, int n_out, const void* a, const void* b, const void* c) {
  if (!cfg || P <= 0 || (P & 1) || n_out < 1 || n_out > 16 || !a || !b || !c || (cfg->S1 != 0 && cfg->S1 != 3) ||
      cfg->S2 < 0 || cfg->S2 > 6 || (cfg->S1 == 0 && cfg->S2 != 0)) {
    stpde_set_error("lig_reduce: bad argument");
    return STPDE_E_BADARG;
  }
  return STPDE_OK;
}

extern "C" int stpde_lig_reduce_fwd(const stpde_jet_cfg* cfg, int S_mlp, int P, int n_out, const float* out_pre,
                                    const float* coef, float* jets, long ldp, void* stream) {
  int rc = check_reduce(cfg, P, n_out, out_pre, coef, jets);
  if (rc) return rc;
  if (ldp < P || S_mlp < 1 || S_mlp > 1 + cfg->S1 + cfg->S2) {
    stpde_set_error("lig_reduce_fwd: ldp < P or bad S_mlp");
    return STPDE_E_BADARG;
  }
  ReduceArgs a{*cfg, P, n_out, S_mlp, ldp, out_pre, coef, jets};
  const size_t n = (size_t)P * n_out;
  STPDE_LAUNCH(k_reduce_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return stpde_check_launch("k_reduce_fwd");
}

extern "C" int stpde_lig_reduce_bwd(const stpde_jet_cfg* cfg, int S_mlp, int P, int n_out, const float* jets_bar,
                                    long ldp, const float* coef, float* abar_out, void* stream) {
  int rc = check_reduce(cfg, P, n_out, jets_bar, coef, abar_out);
  if (rc) return rc;
  if (ldp < P || S_mlp < 1 || S_mlp > 1 + cfg->S1 + cfg->S2) {
    stpde_set_error("lig_reduce_bwd: ldp < P or bad S_mlp");
    return STPDE_E_BADARG;
  }
  ReduceArgs a{*cfg, P, n_out, S_mlp, ldp, jets_bar, coef, abar_out};
  const size_t n = (size_t)P * 32;      // one thread per (row tile, lane): P / 2 tiles x 64 lanes
  STPDE_LAUNCH(k_reduce_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return stpde_check_launch("k_reduce_bwd");
}

// ---------------------------------------------------------------------------------------------------------
// xbar = sum_l W_s,l^T abar_l (value stream)  ->  scatter-add into d latent
// ---------------------------------------------------------------------------------------------------------
struct XbarArgs {
  stpde_xbar_desc d;
  const float* abar[8];
  const float* wsl[8];  // [MT_l][XL][256]: A operand of W_s,l^T restricted to the latent channels (row m = channel 16 xl + m)
  const int* cell;
  float* dlatent;
  float* xrows;  // != null: per-row latent adjoints [16 * ntiles][CP] instead of atomics (deterministic path)
  int CP;        // row length of xrows = C rounded up to a multiple of 4
};

constexpr int XPAD = 49;   // padded row length (floats) of the 16 x 48 transpose patch: odd -> conflict-free columns
#ifndef STPDE_XBAR_R
#define STPDE_XBAR_R 4
#endif
constexpr int XR = STPDE_XBAR_R;      // row tiles per wave: every weight fragment fetched from L2 feeds XR * XL * 4 MFMAs

// XL = number of 16-channel output tiles (C <= 16 XL).  The coordinate / bias columns of the augmented input get no
// adjoint (query points carry no gradient), so only the latent channels are contracted.  Round 2: one row tile per wave
// and all 3 augmented-input tiles made the kernel L2-bound on the weight fragments (3 KB of weights per 1 KB of abar).
template <int XL>
__global__ __launch_bounds__(256) void k_xbar(XbarArgs a) {
  const int lane = threadIdx.x & 63;
  const int tile0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * XR;
  if (tile0 >= a.d.ntiles) return;
  const int lo = lane * 4;
  int tl[XR];
#pragma unroll
  for (int t = 0; t < XR; ++t) tl[t] = tile0 + t < a.d.ntiles ? tile0 + t : a.d.ntiles - 1;   // clamp, stores are guarded
  f32x4 acc[XR][XL];
#pragma unroll
  for (int t = 0; t < XR; ++t)
#pragma unroll
    for (int xl = 0; xl < XL; ++xl) acc[t][xl] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int l = 0; l < a.d.nlayers; ++l) {
    const int MT = a.d.MT[l], SP = a.d.SP[l];
    const float* ab = a.abar[l] + lo;  // stream 0 of tile t starts at (size_t)t * SP * MT * 256
    const float* w = a.wsl[l] + lo;
    if (a.d.packed[l]) {
      // packed ADJOINT buffer (bf16 mode, common.h): bf16 blocks, stream 0 first -- two blocks are one K = 32 bf16 MFMA
      // against the weight blocks of the same two output tiles, rounded here
      const char* ab16 = reinterpret_cast<const char*>(a.abar[l]) + lane * 8;
      const size_t tb = blk_tile_bytes(2, a.d.S[l], MT);
      const bf16x4 z4 = to_bf4(f32x4{0.f, 0.f, 0.f, 0.f});
      for (int mt = 0; mt < MT; mt += 2) {
        const bool two = mt + 1 < MT;
        bf16x8 B8[XR];
#pragma unroll
        for (int t = 0; t < XR; ++t) {
          const char* bp = ab16 + tl[t] * tb + (size_t)mt * 512;
          B8[t] = cat8(*reinterpret_cast<const bf16x4*>(bp), two ? *reinterpret_cast<const bf16x4*>(bp + 512) : z4);
        }
#pragma unroll
        for (int xl = 0; xl < XL; ++xl) {
          const f32x4 w0 = ld4(w + ((size_t)mt * XL + xl) * 256);
          const f32x4 w1 = two ? ld4(w + ((size_t)(mt + 1) * XL + xl) * 256) : f32x4{0.f, 0.f, 0.f, 0.f};
          const bf16x8 A8 = cat8(to_bf4(w0), to_bf4(w1));
#pragma unroll
          for (int t = 0; t < XR; ++t) acc[t][xl] = mfma_bf(A8, B8[t], acc[t][xl]);
        }
      }
      continue;
    }
    const size_t tstride = (size_t)SP * MT * 256;     // floats between the value streams of consecutive tiles
    // two output tiles per iteration (MT is even for every hidden layer): their loads are issued together
    for (int mt = 0; mt + 1 < MT; mt += 2) {
      f32x4 B0[XR], B1[XR], w0[XL], w1[XL];
#pragma unroll
      for (int t = 0; t < XR; ++t) {
        B0[t] = ld4(ab + tl[t] * tstride + (size_t)mt * 256);
        B1[t] = ld4(ab + tl[t] * tstride + (size_t)(mt + 1) * 256);
      }
#pragma unroll
      for (int xl = 0; xl < XL; ++xl) {
        w0[xl] = ld4(w + ((size_t)mt * XL + xl) * 256);
        w1[xl] = ld4(w + ((size_t)(mt + 1) * XL + xl) * 256);
      }
#pragma unroll
      for (int t = 0; t < XR; ++t)
#pragma unroll
        for (int xl = 0; xl < XL; ++xl) {
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[t][xl] = mfma4(w0[xl][r], B0[t][r], acc[t][xl]);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[t][xl] = mfma4(w1[xl][r], B1[t][r], acc[t][xl]);
        }
    }
    if (MT & 1) {
      const int mt = MT - 1;
#pragma unroll
      for (int xl = 0; xl < XL; ++xl) {
        f32x4 wv = ld4(w + ((size_t)mt * XL + xl) * 256);
#pragma unroll
        for (int t = 0; t < XR; ++t) {
          f32x4 B = ld4(ab + tl[t] * tstride + (size_t)mt * 256);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[t][xl] = mfma4(wv[r], B[r], acc[t][xl]);
        }
      }
    }
  }
  const int g = lane >> 4, j = lane & 15;
  if (a.xrows) {
    // deterministic path: rows x channels through a per-wave LDS patch, then row-major, fully coalesced float4 stores;
    // the per-node sums are taken by k_dlat_reduce in a fixed order
    __shared__ float patch[4][16 * XPAD];
    float* pt = patch[threadIdx.x >> 6];
    const int q4 = a.CP >> 2;                       // float4 groups per row
#pragma unroll
    for (int t = 0; t < XR; ++t) {
      if (tile0 + t >= a.d.ntiles) break;
#pragma unroll
      for (int xl = 0; xl < XL; ++xl)
#pragma unroll
        for (int r = 0; r < 4; ++r) pt[j * XPAD + 16 * xl + 4 * g + r] = acc[t][xl][r];
      __builtin_amdgcn_wave_barrier();
      float* dst = a.xrows + (size_t)(tile0 + t) * 16 * a.CP;
      for (int idx = lane; idx < 16 * q4; idx += 64) {
        const int row = idx / q4, c4 = idx - row * q4;
        const float* src = pt + row * XPAD + 4 * c4;
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (4 * c4 + r < a.d.C) ? src[r] : 0.f;
        st4(dst + (size_t)row * a.CP + 4 * c4, v);
      }
      __builtin_amdgcn_wave_barrier();
    }
    return;
  }
  const int n1 = a.d.n1, n2 = a.d.n2, C = a.d.C;
#pragma unroll
  for (int t = 0; t < XR; ++t) {
    if (tile0 + t >= a.d.ntiles) break;
    const int p = (tile0 + t) * 2 + (j >> 3), corner = j & 7;
    const size_t node =
        (size_t)a.cell[p] + ((size_t)((corner >> 2) & 1) * n1 + ((corner >> 1) & 1)) * n2 + (corner & 1);
    float* dst = a.dlatent + node * C;
#pragma unroll
    for (int xl = 0; xl < XL; ++xl)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ch = 16 * xl + 4 * g + r;
        if (ch < C) atomicAdd(dst + ch, acc[t][xl][r]);
      }
  }
}

static int launch_xbar(const stpde_xbar_desc* d, const float* const* abar, const float* const* WsL_pack,
                       const int* cell, float* dlatent, float* xrows, void* stream);

extern "C" int stpde_lig_xbar_scatter(const stpde_xbar_desc* d, const float* const* abar,
                                      const float* const* WsL_pack, const int* cell, float* dlatent, void* stream) {
  if (!cell || !dlatent) {
    stpde_set_error("lig_xbar_scatter: bad argument");
    return STPDE_E_BADARG;
  }
  return launch_xbar(d, abar, WsL_pack, cell, dlatent, nullptr, stream);
}

extern "C" int stpde_lig_xbar_rows(const stpde_xbar_desc* d, const float* const* abar, const float* const* WsL_pack,
                                   float* xrows, void* stream) {
  if (!xrows) {
    stpde_set_error("lig_xbar_rows: bad argument");
    return STPDE_E_BADARG;
  }
  return launch_xbar(d, abar, WsL_pack, nullptr, nullptr, xrows, stream);
}

static int launch_xbar(const stpde_xbar_desc* d, const float* const* abar, const float* const* WsL_pack,
                       const int* cell, float* dlatent, float* xrows, void* stream) {
  if (!d || d->ntiles <= 0 || d->nlayers < 1 || d->nlayers > 8 || d->C < 1 || 3 + d->C + 1 > 16 * (XT - 1) + 4 || !abar ||
      !WsL_pack) {
    stpde_set_error("lig_xbar: bad argument");
    return STPDE_E_BADARG;
  }
  XbarArgs a{};
  a.d = *d;
  for (int l = 0; l < d->nlayers; ++l) {
    if (!abar[l] || !WsL_pack[l] || d->MT[l] < 1 || d->SP[l] < 1) {
      stpde_set_error("lig_xbar_scatter: bad layer %d", l);
      return STPDE_E_BADARG;
    }
    a.abar[l] = abar[l];
    a.wsl[l] = WsL_pack[l];
  }
  a.cell = cell;
  a.dlatent = dlatent;
  a.xrows = xrows;
  a.CP = (d->C + 3) / 4 * 4;
  const dim3 grid((d->ntiles + 4 * XR - 1) / (4 * XR));
  if (d->C <= 16)
    STPDE_LAUNCH(k_xbar<1>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else      // C <= 32 (checked above: the sparse third tile of the augmented input holds features 32..35)
    STPDE_LAUNCH(k_xbar<2>, grid, dim3(256), 0, (hipStream_t)stream, a);
  return stpde_check_launch("k_xbar");
}

// ---------------------------------------------------------------------------------------------------------
// Deterministic d latent: per-NODE gather of the per-row adjoints written by k_xbar (xrows), in a fixed order
// (corner 0..7, then the points of the owning cell in ascending point index), instead of fp32 atomics.
// The backward of the advanced-index gather at src/regular_nd_grid_interpolation.py:65-66 is an
// index_put_(accumulate=True), which is deterministic on the reference's CPU path; so is this.
//   perm[q]   point index of the q-th point in (stable) cell order
//   start[c]  first q of the points whose cell (= linear index of the cell's corner-0 node, incl. batch) is c;
//             start has n_nodes + 1 entries
// One group of 16 lanes per node (lane = float4 of channels), 4 nodes per wave.
// ---------------------------------------------------------------------------------------------------------
struct DlatArgs {
  int B, n0, n1, n2, C, CP;
  const float* xrows;   // [8 * P][CP]   row = 8 * point + corner
  const int* perm;
  const int* start;
  float* dlatent;       // [B][n0][n1][n2][C], accumulated into (+=)
};

__global__ __launch_bounds__(256) void k_dlat_reduce(DlatArgs a) {
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int c4 = gid & 15;
  const size_t node = gid >> 4;
  const size_t nnodes = (size_t)a.B * a.n0 * a.n1 * a.n2;
  if (node >= nnodes || 4 * c4 >= a.CP) return;
  const int i2 = node % a.n2, i1 = (node / a.n2) % a.n1, i0 = (node / ((size_t)a.n2 * a.n1)) % a.n0;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  bool any = false;
#pragma unroll
  for (int corner = 0; corner < 8; ++corner) {
    const int b0 = (corner >> 2) & 1, b1 = (corner >> 1) & 1, b2 = corner & 1;
    const int c0 = i0 - b0, c1 = i1 - b1, c2 = i2 - b2;     // the cell for which this node is corner `corner`
    if (c0 < 0 || c1 < 0 || c2 < 0 || c0 > a.n0 - 2 || c1 > a.n1 - 2 || c2 > a.n2 - 2) continue;
    const size_t cell = node - ((size_t)b0 * a.n1 + b1) * a.n2 - b2;
    const int q0 = a.start[cell], q1 = a.start[cell + 1];
    for (int q = q0; q < q1; ++q) {
      const size_t row = (size_t)a.perm[q] * 8 + corner;
      acc += ld4(a.xrows + row * a.CP + 4 * c4);
      any = true;
    }
  }
  if (!any) return;
  float* dst = a.dlatent + node * a.C + 4 * c4;
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (4 * c4 + r < a.C) dst[r] += acc[r];
}

extern "C" int stpde_lig_dlatent_reduce(int B, int n0, int n1, int n2, int C, const float* xrows, const int* perm,
                                        const int* start, float* dlatent, void* stream) {
  if (B < 1 || n0 < 2 || n1 < 2 || n2 < 2 || C < 1 || C > 64 || !xrows || !perm || !start || !dlatent) {
    stpde_set_error("lig_dlatent_reduce: bad argument (C <= 64)");
    return STPDE_E_BADARG;
  }
  DlatArgs a{B, n0, n1, n2, C, (C + 3) / 4 * 4, xrows, perm, start, dlatent};
  const size_t n = (size_t)B * n0 * n1 * n2 * 16;
  STPDE_LAUNCH(k_dlat_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return stpde_check_launch("k_dlat_reduce");
}
