// Shared pieces of the U-Net convolution kernels (conv3d.hip, conv3d_fused.hip): kernel arguments and voxel / tap indexing.
#pragma once
#include "common.h"

struct ConvArgs {
  stpde_conv3d_desc d;
  const float* x;
  const float* w;
  const float* bias;
  float* y;
  const float* ybar;
  float* dW;
  float* dbias;      // weight-gradient kernels: sum over the voxels of ybar, added with atomics (nullable)
  const float* in_stat;   // ONLOAD weight gradient (1x1x1): x' = max(0, (x - mean) * (rstd * gamma) + beta) on the operand
  const float* in_gamma;
  const float* in_beta;
  int nvox, gx;
};

struct Vox {
  int b, t, z, x;
};

__device__ __forceinline__ Vox vox_coords(const stpde_conv3d_desc& d, int v) {
  Vox c;
  c.x = v % d.X;
  int r = v / d.X;
  c.z = r % d.Z;
  r /= d.Z;
  c.t = r % d.T;
  c.b = r / d.T;
  return c;
}

// flattened index of the tap neighbour of voxel c, or -1 outside the volume
__device__ __forceinline__ int tap_neighbour(const stpde_conv3d_desc& d, const Vox& c, int tap) {
  if (d.ksize == 1) return ((c.b * d.T + c.t) * d.Z + c.z) * d.X + c.x;
  const int dt = tap / 9 - 1, dz = (tap / 3) % 3 - 1, dx = tap % 3 - 1;
  const int t = c.t + dt, z = c.z + dz, x = c.x + dx;
  if (t < 0 || t >= d.T || z < 0 || z >= d.Z || x < 0 || x >= d.X) return -1;
  return ((c.b * d.T + t) * d.Z + z) * d.X + x;
}

// Cheap neighbour indexing (round 3): the linear index of a voxel and nine validity bits (bit 3 * dim + delta + 1: the
// neighbour at delta = -1 / 0 / +1 along dim exists) are computed ONCE per voxel; a tap then costs an AND, a compare, an add
// and a select instead of the ~15 integer instructions of tap_neighbour -- on the fp32 MFMA every VALU instruction is paid in
// MFMA time, and the 16-channel 3x3x3 convolutions of the full-resolution levels spent more cycles on indices than on MFMAs.
struct VoxN {
  int lin;
  unsigned ok;
};
__device__ __forceinline__ VoxN vox_prepare(const stpde_conv3d_desc& d, const Vox& c) {
  VoxN v;
  v.lin = ((c.b * d.T + c.t) * d.Z + c.z) * d.X + c.x;
  v.ok = (c.t > 0 ? 1u : 0u) | 2u | (c.t + 1 < d.T ? 4u : 0u) | (c.z > 0 ? 8u : 0u) | 16u | (c.z + 1 < d.Z ? 32u : 0u) |
         (c.x > 0 ? 64u : 0u) | 128u | (c.x + 1 < d.X ? 256u : 0u);
  return v;
}
// wave-uniform part of a tap: validity mask and linear offset
__device__ __forceinline__ void tap_uniform(const stpde_conv3d_desc& d, int tap, unsigned& mask, int& off) {
  if (d.ksize == 1) {
    mask = 0u;
    off = 0;
    return;
  }
  const int dt = tap / 9 - 1, dz = (tap / 3) % 3 - 1, dx = tap % 3 - 1;
  mask = (1u << (dt + 1)) | (1u << (3 + dz + 1)) | (1u << (6 + dx + 1));
  off = (dt * d.Z + dz) * d.X + dx;
}
__device__ __forceinline__ int tap_nb(const VoxN& v, unsigned mask, int off) {
  return (v.ok & mask) == mask ? v.lin + off : -1;
}
