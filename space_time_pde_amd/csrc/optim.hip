// Train-step tail: gradient value clipping + Adam update fused into one pass over a
// parameter tensor.  Replaces torch.nn.utils.clip_grad_value_ + optimizer.step() of the reference
// (experiments/rb2d/train.py:79-83, optim.Adam at :330-333).  Pure HBM streaming: 16 B read (p, g, m, v) and 12 B
// written (p, m, v) per element, float4 accesses, grid-stride.
#include "common.h"

struct AdamArgs {
  stpde_adam_desc d;
  float* p;
  const float* g;
  float* m;
  float* v;
};

__device__ __forceinline__ void adam_elem(const stpde_adam_desc& d, float& p, float g, float& m, float& v) {
  if (d.clip > 0.f) g = fminf(fmaxf(g, -d.clip), d.clip);        // clip_grad_value_
  if (d.weight_decay != 0.f) g = g + d.weight_decay * p;
  m = m + (1.f - d.beta1) * (g - m);                             // torch: exp_avg.lerp_(grad, 1 - beta1)
  v = d.beta2 * v + (1.f - d.beta2) * g * g;                     // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
  const float denom = sqrtf(v) / d.bias2_sqrt + d.eps;
  p = p - d.step_size * (m / denom);                             // addcdiv_(exp_avg, denom, value=-lr/bias1)
}

__global__ __launch_bounds__(256) void k_clip_adam(AdamArgs a) {
  const long n4 = a.d.n / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 p = ld4(a.p + 4 * i), g = ld4(a.g + 4 * i), m = ld4(a.m + 4 * i), v = ld4(a.v + 4 * i);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float pp = p[r], mm = m[r], vv = v[r];
      adam_elem(a.d, pp, g[r], mm, vv);
      p[r] = pp;
      m[r] = mm;
      v[r] = vv;
    }
    st4(a.p + 4 * i, p);
    st4(a.m + 4 * i, m);
    st4(a.v + 4 * i, v);
  }
  if (blockIdx.x == 0) {  // tail (n not a multiple of 4)
    const long i = n4 * 4 + threadIdx.x;
    if (i < a.d.n) {
      float pp = a.p[i], mm = a.m[i], vv = a.v[i];
      adam_elem(a.d, pp, a.g[i], mm, vv);
      a.p[i] = pp;
      a.m[i] = mm;
      a.v[i] = vv;
    }
  }
}

// Multi-tensor variant: ONE launch updates every parameter tensor of the model.  Block b works on chunk b of the
// chunk table (tensor index, element offset); the tensor table holds the four pointers, the length and the
// bias-correction factors (tensors may have different step counts).
__global__ __launch_bounds__(256) void k_clip_adam_multi(stpde_adam_desc d, const stpde_adam_tensor* tensors,
                                                         const stpde_adam_chunk* chunks, int chunk_elems) {
  const stpde_adam_chunk c = chunks[blockIdx.x];
  const stpde_adam_tensor t = tensors[c.tensor];
  d.step_size = t.step_size;
  d.bias2_sqrt = t.bias2_sqrt;
  const long lo = c.offset;
  const long hi = lo + chunk_elems < t.n ? lo + chunk_elems : t.n;
  const long n4 = (hi - lo) / 4;           // offsets are multiples of 4 and the pointers 16-byte aligned
  for (long i = threadIdx.x; i < n4; i += 256) {
    const long e = lo + 4 * i;
    f32x4 p = ld4(t.p + e), g = ld4(t.g + e), m = ld4(t.m + e), v = ld4(t.v + e);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float pp = p[r], mm = m[r], vv = v[r];
      adam_elem(d, pp, g[r], mm, vv);
      p[r] = pp;
      m[r] = mm;
      v[r] = vv;
    }
    st4(t.p + e, p);
    st4(t.m + e, m);
    st4(t.v + e, v);
  }
  const long e = lo + 4 * n4 + threadIdx.x;
  if (e < hi) {
    float pp = t.p[e], mm = t.m[e], vv = t.v[e];
    adam_elem(d, pp, t.g[e], mm, vv);
    t.p[e] = pp;
    t.m[e] = mm;
    t.v[e] = vv;
  }
}

extern "C" int stpde_clip_adam_multi(const stpde_adam_desc* d, const stpde_adam_tensor* tensors_dev,
                                     const stpde_adam_chunk* chunks_dev, int nchunks, int chunk_elems, void* stream) {
  if (!d || !tensors_dev || !chunks_dev || nchunks <= 0 || chunk_elems <= 0 || (chunk_elems & 3)) {
    stpde_set_error("clip_adam_multi: bad argument (chunk_elems must be a positive multiple of 4)");
    return STPDE_E_BADARG;
  }
  STPDE_LAUNCH(k_clip_adam_multi, dim3((unsigned)nchunks), dim3(256), 0, (hipStream_t)stream, *d, tensors_dev,
               chunks_dev, chunk_elems);
  return stpde_check_launch("k_clip_adam_multi");
}

extern "C" int stpde_clip_adam(const stpde_adam_desc* d, float* param, const float* grad, float* exp_avg,
                               float* exp_avg_sq, void* stream) {
  if (!d || d->n <= 0 || !param || !grad || !exp_avg || !exp_avg_sq || !(d->bias2_sqrt > 0.f)) {
    stpde_set_error("clip_adam: bad argument");
    return STPDE_E_BADARG;
  }
  if (((size_t)param | (size_t)grad | (size_t)exp_avg | (size_t)exp_avg_sq) & 15) {
    stpde_set_error("clip_adam: pointers must be 16-byte aligned");
    return STPDE_E_BADARG;
  }
  AdamArgs a{*d, param, grad, exp_avg, exp_avg_sq};
  long blocks = (d->n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  STPDE_LAUNCH(k_clip_adam, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  return stpde_check_launch("k_clip_adam");
}
