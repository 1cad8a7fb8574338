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
      case STPDE_RES_POWI: g[in.a] += gi * (float)in.b * v[i] / v[in.a]; break;   // d(x^n) = n x^n / x (x != 0)
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
