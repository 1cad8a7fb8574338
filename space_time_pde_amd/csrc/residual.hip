// PDE residuals from the jet streams (the algebra of src/pde.py:115-143 after the derivatives are known):
// every equation of a PDELayer is compiled on the host into a straight-line SSA program over the jet atoms
// (y_c, dy_c/dq_d, d2y_c/dq_a dq_b or the combined second-order stream), the query coordinates and constants;
// one thread evaluates all equations of one query point (coalesced [stream][channel][point] reads), and the adjoint
// kernel replays the program and sweeps it backwards (the reverse-mode tape autograd would build from ~100 elementwise
// torch kernels per step).  Pure streaming work: ~S*n_out*4 B read and n_eq*4 B written per point.
#include "common.h"

#define RES_MAX STPDE_RES_MAX_INS

struct ResArgs {
  const stpde_res_ins* prog;   // device
  int nins, n_eq, n_out, P;
  const float* jets;           // [(s * n_out + c) * ld + p]
  long ld_s, ld_c;             // element strides of the stream and channel dimensions
  const float* x;              // [P][3] or null
  float* res;                  // [n_eq][P]
  const float* res_bar;        // [n_eq][P]
  float* jets_bar;             // same layout as jets, zero-filled by the caller
};

__device__ __forceinline__ float res_eval(const stpde_res_ins& in, const float* v, const ResArgs& a, int p) {
  switch (in.op) {
    case STPDE_RES_JET: return a.jets[(size_t)in.a * a.ld_s + (size_t)in.b * a.ld_c + p];
    case STPDE_RES_X: return a.x[(size_t)p * 3 + in.a];
    case STPDE_RES_CONST: return in.c;
    case STPDE_RES_ADD: return v[in.a] + v[in.b];
    case STPDE_RES_SUB: return v[in.a] - v[in.b];
    case STPDE_RES_MUL: return v[in.a] * v[in.b];
    case STPDE_RES_DIV: return v[in.a] / v[in.b];
    case STPDE_RES_NEG: return -v[in.a];
    case STPDE_RES_POWI: {
      float base = v[in.a], r = 1.f;
      int e = in.b < 0 ? -in.b : in.b;
      while (e) {
        if (e & 1) r *= base;
        base *= base;
        e >>= 1;
      }
      return in.b < 0 ? 1.f / r : r;
    }
    case STPDE_RES_SIN: return sinf(v[in.a]);
    case STPDE_RES_COS: return cosf(v[in.a]);
    case STPDE_RES_EXP: return expf(v[in.a]);
    case STPDE_RES_LOG: return logf(v[in.a]);
    case STPDE_RES_SQRT: return sqrtf(v[in.a]);
    case STPDE_RES_TANH: return tanhf(v[in.a]);
    case STPDE_RES_ABS: return fabsf(v[in.a]);
    default: return v[in.a];   // STPDE_RES_OUT: residual b = value a
  }
}

__global__ __launch_bounds__(256) void k_residual_fwd(ResArgs a) {
  __shared__ stpde_res_ins prog[RES_MAX];
  for (int i = threadIdx.x; i < a.nins; i += 256) prog[i] = a.prog[i];
  __syncthreads();
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= a.P) return;
  float v[RES_MAX];
  for (int i = 0; i < a.nins; ++i) {
    const stpde_res_ins in = prog[i];
    v[i] = res_eval(in, v, a, p);
    if (in.op == STPDE_RES_OUT) a.res[(size_t)in.b * a.P + p] = v[i];
  }
}

__global__ __launch_bounds__(256) void k_residual_bwd(ResArgs a) {
  __shared__ stpde_res_ins prog[RES_MAX];
  for (int i = threadIdx.x; i < a.nins; i += 256) prog[i] = a.prog[i];
  __syncthreads();
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= a.P) return;
  float v[RES_MAX], g[RES_MAX];
  for (int i = 0; i < a.nins; ++i) {
    v[i] = res_eval(prog[i], v, a, p);
    g[i] = 0.f;
  }
  for (int i = a.nins - 1; i >= 0; --i) {
    const stpde_res_ins in = prog[i];
    float gi = g[i];
    switch (in.op) {
      case STPDE_RES_OUT: g[in.a] += a.res_bar[(size_t)in.b * a.P + p]; break;
      case STPDE_RES_JET: {
        float* dst = a.jets_bar + (size_t)in.a * a.ld_s + (size_t)in.b * a.ld_c + p;   // this thread owns point p
        *dst += gi;
        break;
      }
      case STPDE_RES_ADD: g[in.a] += gi; g[in.b] += gi; break;
      case STPDE_RES_SUB: g[in.a] += gi; g[in.b] -= gi; break;
      case STPDE_RES_MUL: g[in.a] += gi * v[in.b]; g[in.b] += gi * v[in.a]; break;
      case STPDE_RES_DIV: g[in.a] += gi / v[in.b]; g[in.b] -= gi * v[i] / v[in.b]; break;
      case STPDE_RES_NEG: g[in.a] -= gi; break;
      case STPDE_RES_POWI: {   // d(x^n)/dx = n x^(n-1), evaluated without dividing by x (x = 0 is a regular point for n >= 1)
        if (in.b != 0) {
          const int m = in.b - 1;
          float base = v[in.a], r = 1.f;
          int e = m < 0 ? -m : m;
          while (e) {
            if (e & 1) r *= base;
            base *= base;
            e >>= 1;
          }
          g[in.a] += gi * (float)in.b * (m < 0 ? 1.f / r : r);
        }
        break;
      }
      case STPDE_RES_SIN: g[in.a] += gi * cosf(v[in.a]); break;
      case STPDE_RES_COS: g[in.a] -= gi * sinf(v[in.a]); break;
      case STPDE_RES_EXP: g[in.a] += gi * v[i]; break;
      case STPDE_RES_LOG: g[in.a] += gi / v[in.a]; break;
      case STPDE_RES_SQRT: g[in.a] += gi * 0.5f / v[i]; break;
      case STPDE_RES_TANH: g[in.a] += gi * (1.f - v[i] * v[i]); break;
      case STPDE_RES_ABS: g[in.a] += v[in.a] > 0.f ? gi : (v[in.a] < 0.f ? -gi : 0.f); break;   // torch: sign(0) = 0
      default: break;   // X, CONST: the coordinates are not differentiated on this path
    }
  }
}

static int res_check(const stpde_res_ins* prog_dev, int nins, int n_eq, int n_out, int P, const float* jets) {
  if (!prog_dev || nins <= 0 || nins > RES_MAX || n_eq <= 0 || n_out <= 0 || P <= 0 || !jets) {
    stpde_set_error("residual: bad argument (at most %d program instructions)", RES_MAX);
    return STPDE_E_BADARG;
  }
  return STPDE_OK;
}

extern "C" int stpde_residual_fwd(const stpde_res_ins* prog_dev, int nins, int n_eq, int n_out, int P,
                                  const float* jets, long ld_stream, long ld_channel, const float* x, float* res,
                                  void* stream) {
  int rc = res_check(prog_dev, nins, n_eq, n_out, P, jets);
  if (rc) return rc;
  if (!res) {
    stpde_set_error("residual_fwd: null output");
    return STPDE_E_BADARG;
  }
  ResArgs a{prog_dev, nins, n_eq, n_out, P, jets, ld_stream, ld_channel, x, res, nullptr, nullptr};
  STPDE_LAUNCH(k_residual_fwd, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
  return stpde_check_launch("k_residual_fwd");
}

extern "C" int stpde_residual_bwd(const stpde_res_ins* prog_dev, int nins, int n_eq, int n_out, int P,
                                  const float* jets, long ld_stream, long ld_channel, const float* x,
                                  const float* res_bar, float* jets_bar, void* stream) {
  int rc = res_check(prog_dev, nins, n_eq, n_out, P, jets);
  if (rc) return rc;
  if (!res_bar || !jets_bar) {
    stpde_set_error("residual_bwd: null pointer");
    return STPDE_E_BADARG;
  }
  ResArgs a{prog_dev, nins, n_eq, n_out, P, jets, ld_stream, ld_channel, x, nullptr, res_bar, jets_bar};
  STPDE_LAUNCH(k_residual_bwd, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
  return stpde_check_launch("k_residual_bwd");
}

// ---- loss reductions of the train step (experiments/rb2d/train.py:69-76: l1 / mse / smooth_l1 of prediction vs target
// and of the stacked residuals vs 0), as SUMS (the caller divides by the global element count, so that the same code
// serves the point-sharded multi-GPU step).  One pass, block reduction, one atomic per block.
struct LossArgs {
  int kind;
  int det;             // forward: out is a long accumulator (common.h), not a float
  long n;
  const float* a;
  const float* b;      // null: compare against 0
  float* out;          // forward: scalar accumulator (zero-filled by the caller)
  const float* gscale; // backward: device scalar = d loss / d sum
  float* ga;           // backward: d loss / d a
};

__device__ __forceinline__ float loss_elem(int kind, float d) {
  const float ad = fabsf(d);
  if (kind == STPDE_LOSS_L1) return ad;
  if (kind == STPDE_LOSS_L2) return d * d;
  return ad < 1.f ? 0.5f * d * d : ad - 0.5f;          // smooth_l1, beta = 1 (torch default)
}
__device__ __forceinline__ float loss_grad(int kind, float d) {
  if (kind == STPDE_LOSS_L1) return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
  if (kind == STPDE_LOSS_L2) return 2.f * d;
  return fabsf(d) < 1.f ? d : (d > 0.f ? 1.f : -1.f);
}

__global__ __launch_bounds__(256) void k_loss_sum(LossArgs a) {
  __shared__ float part[4];
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long)gridDim.x * 256)
    s += loss_elem(a.kind, a.a[i] - (a.b ? a.b[i] : 0.f));
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) acc_add_f32(a.out, 0, (part[0] + part[1]) + (part[2] + part[3]), a.det);
}

__global__ __launch_bounds__(256) void k_loss_grad(LossArgs a) {
  const float g = *a.gscale;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long)gridDim.x * 256)
    a.ga[i] = g * loss_grad(a.kind, a.a[i] - (a.b ? a.b[i] : 0.f));
}

extern "C" int stpde_loss_sum(int kind, long n, const float* a, const float* b, float* out_sum, void* stream) {
  const int det = (kind & STPDE_LOSS_DET) ? 1 : 0;
  kind &= ~STPDE_LOSS_DET;
  if (kind < 0 || kind > 2 || n <= 0 || !a || !out_sum) {
    stpde_set_error("loss_sum: bad argument");
    return STPDE_E_BADARG;
  }
  LossArgs args{kind, det, n, a, b, out_sum, nullptr, nullptr};
  long blocks = (n + 1023) / 1024;
  if (blocks > 1024) blocks = 1024;
  STPDE_LAUNCH(k_loss_sum, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, args);
  return stpde_check_launch("k_loss_sum");
}

extern "C" int stpde_loss_grad(int kind, long n, const float* a, const float* b, const float* grad_sum_dev,
                               float* grad_a, void* stream) {
  if (kind < 0 || kind > 2 || n <= 0 || !a || !grad_sum_dev || !grad_a) {
    stpde_set_error("loss_grad: bad argument");
    return STPDE_E_BADARG;
  }
  LossArgs args{kind, 0, n, a, b, nullptr, grad_sum_dev, grad_a};
  long blocks = (n + 1023) / 1024;
  if (blocks > 2048) blocks = 2048;
  STPDE_LAUNCH(k_loss_grad, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, args);
  return stpde_check_launch("k_loss_grad");
}
