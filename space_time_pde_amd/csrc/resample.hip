// Max pooling (kernel = stride in {1,2} per dimension) and nearest-neighbour up-sampling of the U-Net levels
// (nn.MaxPool3d / nn.Upsample of src/unet3d.py:163-176, 216-238) on channels-last activations [B][T][Z][X][C],
// C % 4 == 0: one float4 of channels per thread, pure HBM streaming.  Pooling ties go to the FIRST maximum of the
// window in (t, z, x) order like torch's max_pool3d backward.
#include "common.h"

struct ResampleArgs {
  stpde_resample_desc d;   // B, T, Z, X = dimensions of the SMALL (pooled / low-resolution) tensor, C, ft, fz, fx
  const float* in;
  float* out;
  const float* aux;        // pooling backward: the forward input
};

__device__ __forceinline__ void small_coords(const stpde_resample_desc& d, long v, int& b, int& t, int& z, int& x) {
  x = (int)(v % d.X);
  v /= d.X;
  z = (int)(v % d.Z);
  v /= d.Z;
  t = (int)(v % d.T);
  b = (int)(v / d.T);
}
__device__ __forceinline__ size_t big_index(const stpde_resample_desc& d, int b, int t, int z, int x) {
  return ((((size_t)b * d.T * d.ft + t) * d.Z * d.fz + z) * d.X * d.fx + x);
}

// MODE 0: pool forward (in = big, out = small)     MODE 1: pool backward (in = d small, aux = big x, out = d big)
// MODE 2: upsample forward (in = small, out = big) MODE 3: upsample backward (in = d big, out = d small)
template <int MODE>
__global__ __launch_bounds__(256) void k_resample(ResampleArgs a) {
  const int C4 = a.d.C / 4;
  const long n = (long)a.d.B * a.d.T * a.d.Z * a.d.X * C4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4;
    int b, t, z, x;
    small_coords(a.d, i / C4, b, t, z, x);
    const size_t so = (size_t)(i / C4) * a.d.C + c;
    if (MODE == 0 || MODE == 1) {
      const float* src = MODE == 0 ? a.in : a.aux;
      f32x4 best = f32x4{0.f, 0.f, 0.f, 0.f};
      int arg[4] = {0, 0, 0, 0};
      int w = 0;
      for (int dt = 0; dt < a.d.ft; ++dt)
        for (int dz = 0; dz < a.d.fz; ++dz)
          for (int dx = 0; dx < a.d.fx; ++dx, ++w) {
            const f32x4 v = ld4(src + big_index(a.d, b, t * a.d.ft + dt, z * a.d.fz + dz, x * a.d.fx + dx) * a.d.C + c);
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (w == 0 || v[r] > best[r] || v[r] != v[r]) {     // first maximum wins; NaN propagates (torch)
                best[r] = v[r];
                arg[r] = w;
              }
          }
      if (MODE == 0) {
        st4(a.out + so, best);
      } else {
        const f32x4 g = ld4(a.in + so);
        w = 0;
        for (int dt = 0; dt < a.d.ft; ++dt)
          for (int dz = 0; dz < a.d.fz; ++dz)
            for (int dx = 0; dx < a.d.fx; ++dx, ++w) {
              f32x4 o;
#pragma unroll
              for (int r = 0; r < 4; ++r) o[r] = arg[r] == w ? g[r] : 0.f;
              st4(a.out + big_index(a.d, b, t * a.d.ft + dt, z * a.d.fz + dz, x * a.d.fx + dx) * a.d.C + c, o);
            }
      }
    } else if (MODE == 2) {
      const f32x4 v = ld4(a.in + so);
      for (int dt = 0; dt < a.d.ft; ++dt)
        for (int dz = 0; dz < a.d.fz; ++dz)
          for (int dx = 0; dx < a.d.fx; ++dx)
            st4(a.out + big_index(a.d, b, t * a.d.ft + dt, z * a.d.fz + dz, x * a.d.fx + dx) * a.d.C + c, v);
    } else {
      f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int dt = 0; dt < a.d.ft; ++dt)
        for (int dz = 0; dz < a.d.fz; ++dz)
          for (int dx = 0; dx < a.d.fx; ++dx)
            s += ld4(a.in + big_index(a.d, b, t * a.d.ft + dt, z * a.d.fz + dz, x * a.d.fx + dx) * a.d.C + c);
      st4(a.out + so, s);
    }
  }
}

extern "C" int stpde_resample3d(const stpde_resample_desc* d, int mode, const float* in, const float* aux, float* out,
                                void* stream) {
  if (!d || d->B < 1 || d->T < 1 || d->Z < 1 || d->X < 1 || d->C < 4 || (d->C & 3) || d->ft < 1 || d->fz < 1 ||
      d->fx < 1 || d->ft > 4 || d->fz > 4 || d->fx > 4 || mode < 0 || mode > 3 || !in || !out || (mode == 1 && !aux)) {
    stpde_set_error("resample3d: bad argument (C %% 4 == 0, factors 1..4, mode 0..3)");
    return STPDE_E_BADARG;
  }
  ResampleArgs a{*d, in, out, aux};
  const long n = (long)d->B * d->T * d->Z * d->X * (d->C / 4);
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  const dim3 grid((unsigned)blocks), block(256);
  const hipStream_t st = (hipStream_t)stream;
  switch (mode) {
    case 0: STPDE_LAUNCH(k_resample<0>, grid, block, 0, st, a); break;
    case 1: STPDE_LAUNCH(k_resample<1>, grid, block, 0, st, a); break;
    case 2: STPDE_LAUNCH(k_resample<2>, grid, block, 0, st, a); break;
    default: STPDE_LAUNCH(k_resample<3>, grid, block, 0, st, a); break;
  }
  return stpde_check_launch("k_resample");
}
