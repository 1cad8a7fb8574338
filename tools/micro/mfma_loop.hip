// Micro-benchmark of the cooperative layer kernel's main loop (fp32 MFMA 16x16x4, B operand from an LDS ring,
// A operand = weight fragments streamed from L2), without produce stage / epilogue.  Variants isolate what keeps the
// MFMA pipe below peak:  hipcc --offload-arch=gfx950 -O3 mfma_loop.hip -o mfma_loop && ./mfma_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// MODE 0: weights from global (L2), B from LDS, barrier per group
// MODE 1: as 0 without barriers
// MODE 2: weights held in registers (no global loads in the loop), B from LDS
// MODE 3: pure MFMA (operands in registers)
// MODE 4: as 0 plus a synthetic "produce" (global load -> exp/rcp VALU -> LDS store) per group
template <int MCg, int S, int NW, int MODE>
__global__ __launch_bounds__(64 * NW, 2) void k_loop(const float* W, const float* Bsrc, float* out, int KT, int MT, int tiles_per_wg) {
  __shared__ __attribute__((aligned(16))) float hb[2][NW][S][256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, lo = lane * 4;
  for (int i = threadIdx.x; i < 2 * NW * S * 256; i += 64 * NW) (&hb[0][0][0][0])[i] = 1e-3f * (i & 255);
  __syncthreads();
  const float* wp = W + (size_t)(wv * MCg) * 256 + lo;
  f32x4 acc[MCg][S];
  for (int mi = 0; mi < MCg; ++mi) for (int st = 0; st < S; ++st) acc[mi][st] = f32x4{0, 0, 0, 0};
  f32x4 wreg[MCg];
  for (int mi = 0; mi < MCg; ++mi) wreg[mi] = ld4(wp + mi * 256);
  const int ngroups = KT / NW;
  for (int t = 0; t < tiles_per_wg; ++t) {
    for (int gi = 0; gi < ngroups; ++gi) {
      const int buf = gi & 1;
#pragma unroll
      for (int q = 0; q < NW; ++q) {
        const int kt = NW * gi + q;
        f32x4 B[S], w[MCg];
#pragma unroll
        for (int st = 0; st < S; ++st) B[st] = MODE == 3 ? wreg[st % MCg] : ld4(&hb[buf][q][st][lo]);
#pragma unroll
        for (int mi = 0; mi < MCg; ++mi) w[mi] = MODE >= 2 && MODE != 4 ? wreg[mi] : ld4(wp + ((size_t)kt * MT + mi) * 256);
#pragma unroll
        for (int mi = 0; mi < MCg; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int st = 0; st < S; ++st) acc[mi][st] = mfma4(w[mi][r], B[st][r], acc[mi][st]);
      }
      if (MODE == 4) {
        f32x4 raw[S];
#pragma unroll
        for (int st = 0; st < S; ++st) raw[st] = ld4(Bsrc + ((((size_t)(blockIdx.x * tiles_per_wg + t) * S + st) * KT + NW * gi + wv) % ((size_t)4 << 20)) * 256 + lo);   // 4 GiB window
#pragma unroll
        for (int st = 0; st < S; ++st) {
          f32x4 v = raw[st];
#pragma unroll
          for (int r = 0; r < 4; ++r) { float e = __expf(-fabsf(v[r])); float inv = __builtin_amdgcn_rcpf(1.f + e); v[r] = v[r] * inv + e * inv * inv * raw[(st + 1) % S][r]; }
          *reinterpret_cast<f32x4*>(&hb[buf ^ 1][wv][st][lo]) = v;
        }
      }
      if (MODE == 0 || MODE == 4) __syncthreads();
    }
  }
  f32x4 s = f32x4{0, 0, 0, 0};
  for (int mi = 0; mi < MCg; ++mi) for (int st = 0; st < S; ++st) s += acc[mi][st];
  if (s[0] == 12345.678f) out[threadIdx.x] = s[1];
}

template <int MCg, int S, int NW, int MODE>
__global__ __launch_bounds__(64 * NW, 2) void k_loop2(const float* W, const float* Bsrc, float* out, int KT, int MT, int tiles_per_wg) {
  __shared__ __attribute__((aligned(16))) float hb[2][NW][S][256];
  __shared__ __attribute__((aligned(16))) float wl[NW][2][MCg][256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, lo = lane * 4;
  for (int i = threadIdx.x; i < 2 * NW * S * 256; i += 64 * NW) (&hb[0][0][0][0])[i] = 1e-3f * (i & 255);
  __syncthreads();
  const float* wp = W + (size_t)(wv * MCg) * 256 + lo;
  f32x4 acc[MCg][S];
  for (int mi = 0; mi < MCg; ++mi) for (int st = 0; st < S; ++st) acc[mi][st] = f32x4{0, 0, 0, 0};
  const int ngroups = KT / NW;
  auto issue = [&](int kt, int slot) {
#pragma unroll
    for (int mi = 0; mi < MCg; ++mi)
      __builtin_amdgcn_global_load_lds(wp + ((size_t)kt * MT + mi) * 256, (__attribute__((address_space(3))) void*)&wl[wv][slot][mi][0], 16, 0, 0);
  };
  f32x4 wn[MCg];
  if (MODE == 5) issue(0, 0);
  if (MODE == 6) for (int mi = 0; mi < MCg; ++mi) wn[mi] = ld4(wp + mi * 256);
  int it = 0;
  for (int t = 0; t < tiles_per_wg; ++t) {
    for (int gi = 0; gi < ngroups; ++gi) {
      const int buf = gi & 1;
#pragma unroll
      for (int q = 0; q < NW; ++q, ++it) {
        const int kt = NW * gi + q;
        const int ktn = (kt + 1) % KT;
        f32x4 B[S], w[MCg];
        if (MODE == 5) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_wave_barrier();
          issue(ktn, (it + 1) & 1);
#pragma unroll
          for (int mi = 0; mi < MCg; ++mi) w[mi] = ld4(&wl[wv][it & 1][mi][lo]);
        } else {
#pragma unroll
          for (int mi = 0; mi < MCg; ++mi) { w[mi] = wn[mi]; }
#pragma unroll
          for (int mi = 0; mi < MCg; ++mi) wn[mi] = ld4(wp + ((size_t)ktn * MT + mi) * 256);
        }
#pragma unroll
        for (int st = 0; st < S; ++st) B[st] = ld4(&hb[buf][q][st][lo]);
#pragma unroll
        for (int mi = 0; mi < MCg; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int st = 0; st < S; ++st) acc[mi][st] = mfma4(w[mi][r], B[st][r], acc[mi][st]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  }
  f32x4 s = f32x4{0, 0, 0, 0};
  for (int mi = 0; mi < MCg; ++mi) for (int st = 0; st < S; ++st) s += acc[mi][st];
  if (s[0] == 12345.678f) out[threadIdx.x] = s[1];
}

template <int MCg, int S, int NW, int MODE>
void run(const char* name, const float* W, const float* B, float* out, int KT, int wgs_per_cu) {
  const int MT = NW * MCg, tiles = 64;
  const int grid = 256 * wgs_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  if constexpr (MODE >= 5) hipLaunchKernelGGL((k_loop2<MCg, S, NW, MODE>), dim3(grid), dim3(64 * NW), 0, 0, W, B, out, KT, MT, 4);
  else hipLaunchKernelGGL((k_loop<MCg, S, NW, MODE>), dim3(grid), dim3(64 * NW), 0, 0, W, B, out, KT, MT, 4);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  if constexpr (MODE >= 5) hipLaunchKernelGGL((k_loop2<MCg, S, NW, MODE>), dim3(grid), dim3(64 * NW), 0, 0, W, B, out, KT, MT, tiles);
  else hipLaunchKernelGGL((k_loop<MCg, S, NW, MODE>), dim3(grid), dim3(64 * NW), 0, 0, W, B, out, KT, MT, tiles);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = 2.0 * 16 * 16 * 4 * (double)grid * NW * tiles * KT * MCg * S * 4;
  printf("%-46s MCg=%d S=%d NW=%d KT=%d wg/cu=%d : %7.2f ms  %6.1f TFLOP/s (%.0f%% of 157.3)\n", name, MCg, S, NW, KT, wgs_per_cu, ms,
         flop / ms / 1e9, 100 * flop / ms / 1e9 / 157.3);
}

int main() {
  float *W, *B, *out;
  hipMalloc(&W, 64 << 20); hipMalloc(&B, (size_t)6 << 30); hipMalloc(&out, 1 << 20);
  hipMemset(W, 0, 64 << 20); hipMemset(B, 0, (size_t)6 << 30);
  for (int wg = 1; wg <= 3; ++wg) {
    run<4, 5, 4, 3>("pure MFMA (registers)", W, B, out, 32, wg);
    run<4, 5, 4, 2>("B from LDS, weights in registers", W, B, out, 32, wg);
    run<4, 5, 4, 1>("B from LDS, weights from L2, no barrier", W, B, out, 32, wg);
    run<4, 5, 4, 0>("B from LDS, weights from L2, barrier/group", W, B, out, 32, wg);
    run<4, 5, 4, 4>("... + synthetic produce stage", W, B, out, 32, wg);
    if (wg == 1) {
      run<2, 10, 8, 0>("2 row tiles/WG: MCg=2 S=10 NW=8 (ring 160 KB)", W, B, out, 32, wg);
      run<2, 10, 8, 4>("2 row tiles/WG: ... + synthetic produce", W, B, out, 32, wg);
    }
    run<4, 10, 4, 0>("2 row tiles/WG: MCg=4 S=10 NW=4 (80 KB)", W, B, out, 32, wg > 1 ? 1 : 1);
    run<2, 10, 4, 0>("2 row tiles/WG: MCg=2 S=10 NW=4 (80 KB)", W, B, out, 32, wg > 1 ? 1 : 1);
    run<2, 10, 4, 4>("2 row tiles/WG: MCg=2 S=10 NW=4 + produce", W, B, out, 32, wg > 1 ? 1 : 1);
    run<4, 5, 4, 5>("weights via global_load_lds, 1 k-tile ahead", W, B, out, 32, wg);
    run<4, 5, 4, 6>("weights via explicit register double buffer", W, B, out, 32, wg);
    run<2, 5, 4, 5>("MCg=2: weights via global_load_lds", W, B, out, 16, wg);
    run<2, 5, 4, 6>("MCg=2: explicit register double buffer", W, B, out, 16, wg);
    run<2, 5, 4, 0>("MCg=2: LDS + L2 + barrier", W, B, out, 16, wg);
    run<2, 5, 4, 4>("MCg=2: ... + synthetic produce", W, B, out, 16, wg);
  }
  return 0;
}
