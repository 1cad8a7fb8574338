// Probe of ds_read_b64_tr_b16 on gfx950: which LDS elements does lane l receive for a given per-lane address?
// LDS is filled with lds[i] = i (16-bit); every lane of a 16-lane group passes the address of 4 contiguous elements.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/micro/tr16_probe.hip -o /tmp/tr16 && /tmp/tr16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out, int rowstride) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  // group g reads a 4 x 16 block whose rows are `rowstride` elements apart; lane i supplies row i / 4, columns 4 (i % 4)..+3
  const short* p = lds + g * 4 * rowstride + (i / 4) * rowstride + (i % 4) * 4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = v[e];
}
int main() {
  short* d;
  hipMalloc(&d, 64 * 4 * sizeof(short));
  for (int rs : {16, 20}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, rs);
    short h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("rowstride %d\n", rs);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d:", l);
      for (int e = 0; e < 4; ++e) printf(" %4d", h[l * 4 + e]);
      printf("%s", (l % 4 == 3) ? "\n" : "   ");
    }
  }
  return 0;
}
