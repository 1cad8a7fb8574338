// Does a VALU instruction issue "for free" in the shadow of an fp32 MFMA on gfx950?  Each wave runs a loop of
// 16 x v_mfma_f32_16x16x4_f32 (independent accumulators) + K independent VALU instructions (v_fma_f32 on private
// registers) interleaved one after each MFMA; W waves per SIMD.  If the matrix pipe and the vector ALU co-execute, the time
// does not depend on K (up to ~7 VALU per MFMA); if fp32 MFMA shares the vector FMA lanes, time grows by 4 cycles per VALU.
//   hipcc --offload-arch=gfx950 -O3 mfma_valu_coissue.hip -o mfma_valu_coissue && ./mfma_valu_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int K, int KIND>   // KIND 0: v_fma_f32, 1: v_pk_fma_f32, 2: v_exp_f32 (transcendental), 3: bf16 MFMA + v_fma_f32
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  bf16x8 a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(a + i); b8[i] = (__bf16)(b + i); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      if (KIND == 3) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[m], 0, 0, 0);
      else acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < K; ++q) {
        const int r = (m * K + q) & 7;
        if (KIND == 0 || KIND == 3) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[r]) : "v"(b));
        if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&v[r & 6]) : "v"(*(double*)&v[(r & 6) ^ 2]));
        if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(v[r]));
      }
    }
  }
  f32x4 s = f32x4{0, 0, 0, 0};
  for (int i = 0; i < 16; ++i) s += acc[i];
  float t = 0;
  for (int i = 0; i < 8; ++i) t += v[i];
  if (s[0] + t == 12345.678f) out[threadIdx.x] = s[1];
}

template <int K, int KIND>
void run(const char* name, int wgs_per_cu, float* out) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<K, KIND>), dim3(256 * wgs_per_cu), dim3(256), 0, 0, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<K, KIND>), dim3(256 * wgs_per_cu), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double nm = 16.0 * iters * wgs_per_cu;                 // MFMAs per SIMD
  const double cyc = ms * 1e-3 * 2.4e9 / nm;                     // cycles per MFMA at a nominal 2.4 GHz
  printf("%-34s K=%d waves/SIMD=%d : %8.3f ms  %6.1f cycles per MFMA (+%d VALU)\n", name, K, wgs_per_cu, ms, cyc, K);
}

int main() {
  float* out; hipMalloc(&out, 1 << 20);
  for (int w = 1; w <= 2; ++w) {
    run<0, 0>("fp32 MFMA only", w, out);
    run<1, 0>("fp32 MFMA + v_fma_f32", w, out);
    run<2, 0>("fp32 MFMA + v_fma_f32", w, out);
    run<4, 0>("fp32 MFMA + v_fma_f32", w, out);
    run<7, 0>("fp32 MFMA + v_fma_f32", w, out);
    run<2, 1>("fp32 MFMA + v_pk_fma_f32", w, out);
    run<4, 1>("fp32 MFMA + v_pk_fma_f32", w, out);
    run<1, 2>("fp32 MFMA + v_exp_f32", w, out);
    run<2, 2>("fp32 MFMA + v_exp_f32", w, out);
    run<0, 3>("bf16 MFMA 16x16x32 only", w, out);
    run<2, 3>("bf16 MFMA + v_fma_f32", w, out);
    run<4, 3>("bf16 MFMA + v_fma_f32", w, out);
  }
  return 0;
}
