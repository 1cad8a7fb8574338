// Shared device helpers for libstpde_hip (gfx950 / CDNA4 only: wave64, v_mfma_f32_16x16x4_f32).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/stpde_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define XT STPDE_XT
// Augmented raw input of a corner row (coordinates, latent channels, the ones column of the bias): XT = 3 fragment tiles of
// 16 slots.  Tiles 0 and 1 hold features 0..31 in order; tile 2 is SPARSE -- only register 0 of its fragment is populated
// (slots 32, 36, 40, 44 = features 32..35) -- so a GEMM over the augmented input takes 9 k-steps instead of 12 (3 + 32 + 1
// = 36 features is exactly what the reference's 32 latent channels need; wider latents take the generic path).
__host__ __device__ constexpr int x_live(int xt) { return xt == XT - 1 ? 1 : 4; }

// Fragment of tile xt (compile time) of an augmented-input operand -- the raw-input image X or a weight pack against it: the
// sparse last tile is loaded as the ONE dword that is used (round 6).  As a 16-byte load its three dead result registers were
// handed out again at once -- to the address of the next load, to the next load's own result -- and every such write has to
// wait for the load: `global_load_dwordx4 v[24:27] ; s_waitcnt vmcnt(0) ; v_lshl_add_u64 v[26:27] ; global_load_dwordx4
// v[26:29]` in the listing of k_layer_coop's layer-0 prefetch, i.e. serialised L2 round trips inside a burst of loads that was
// written to be in flight together.
__device__ __forceinline__ f32x4 ld4x(const float* p, int xt) {
  if (xt == XT - 1) return f32x4{p[0], 0.f, 0.f, 0.f};
  return *reinterpret_cast<const f32x4*>(p);
}

// One v_mfma_f32_16x16x4_f32: D[i][j] += sum_k A[i][k] B[k][j]; lane l holds A[l&15][l>>4], B[l>>4][l&15],
// D rows 4*(l>>4)+r (r = register), column l&15.  Exact fp32 (k-ordered fma chain).
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// bf16-operand MFMA (config-4 path): v_mfma_f32_16x16x32_bf16, lane l holds A[l&15][8*(l>>4)+e], B[8*(l>>4)+e][l&15]
// (e = 0..7), D as above; operands are rounded to nearest-even (v_cvt_pk_bf16_f32), accumulation stays fp32.
// The 8 k-slots of a lane carry TWO of the fp32 path's 4-element fragments (two k-tiles, or two streams).
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x4 to_bf4(f32x4 v) {
  bf16x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = (__bf16)v[i];
  return r;
}
__device__ __forceinline__ f32x4 bf4_to_f32(bf16x4 v) {
  f32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = (float)v[i];
  return r;
}
// ds_read_b64_tr_b16 (gfx950): hardware 4 x 16 -> 16 x 4 transpose of 16-bit elements inside a 16-lane group.  Lane i of a
// group passes the address of 4 contiguous elements -- row i / 4, columns 4 (i % 4) .. +3 of a [4][16] block with a free row
// stride -- and receives column i of the block (rows 0..3).  Measured: tools/micro/tr16_probe.hip.
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x4 lds_read_tr16(const __bf16* p) {
  return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                         (__attribute__((address_space(3))) s16x4*)(const_cast<__bf16*>(p))));
}
__device__ __forceinline__ bf16x8 cat8(bf16x4 a, bf16x4 b) {
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
// v_mfma_f32_16x16x16_bf16: lane l holds A[l&15][4*(l>>4)+e], B[4*(l>>4)+e][l&15], e = 0..3
__device__ __forceinline__ f32x4 mfma_bf16k(bf16x4 a, bf16x4 b, f32x4 c) {
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_bf(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// PACKED layer buffers (round 3, bf16 mode).  `mode` (wave-uniform, a compile-time constant at every call site):
//   0  fp32 blocks           tile t: [S][MT][64 lanes][4] fp32
//   1  packed STASH          tile t: [MT][64][4] fp32 (stream 0)  then  [S - 1][MT][64][4] bf16 (streams 1 .. S-1)
//      a buffer of pre-activations: the value stream is the argument of the activation jets and stays fp32, the derivative
//      streams are only ever multiplied into bf16 MFMA operands.  MT * (1024 + (S - 1) * 512) bytes per row tile
//   2  packed ADJOINT        tile t: [S][MT][64][4] bf16
//      a buffer of adjoints: every consumer rounds every stream to a bf16 MFMA operand anyway (input-gradient and
//      weight-gradient products; the d-latent reduction reads the value stream).  MT * S * 512 bytes per row tile
// against MT * S * 1024: -40 % / -50 % of the bytes of the HBM-bound kernels.  `st` is a compile-time index at every call site.
__host__ __device__ inline size_t packed_tile_bytes(int S, int MT) { return (size_t)MT * (1024 + (S - 1) * 512); }
__host__ __device__ inline size_t blk_tile_bytes(int mode, int S, int MT) {
  return mode == 0 ? (size_t)S * MT * 1024 : (mode == 1 ? packed_tile_bytes(S, MT) : (size_t)S * MT * 512);
}
// byte offset of block (st, mt) inside its tile
__device__ __forceinline__ size_t blk_off(int mode, int MT, int st, int mt) {
  if (mode == 0) return ((size_t)st * MT + mt) * 1024;
  if (mode == 2) return ((size_t)st * MT + mt) * 512;
  return st == 0 ? (size_t)mt * 1024 : (size_t)MT * 1024 + ((size_t)(st - 1) * MT + mt) * 512;
}
__device__ __forceinline__ bool blk_is16(int mode, int st) { return mode == 2 || (mode == 1 && st > 0); }
__device__ __forceinline__ f32x4 ld_blk(const float* buf, int mode, size_t tile, int S, int MT, int st, int mt, int lane) {
  const char* b = reinterpret_cast<const char*>(buf) + tile * blk_tile_bytes(mode, S, MT) + blk_off(mode, MT, st, mt);
  if (!blk_is16(mode, st)) return ld4(reinterpret_cast<const float*>(b) + lane * 4);
  return bf4_to_f32(*reinterpret_cast<const bf16x4*>(b + lane * 8));
}
// The same load WITHOUT the bf16 -> fp32 conversion of a bf16 block: the 8 bytes land in the first two registers of the
// result and blk_val() converts them where the value is used -- for loads that are issued a whole row tile ahead of their
// use (weight-gradient P operand), where a conversion at the load site would wait for the load at once.
__device__ __forceinline__ f32x4 ld_blk_raw(const float* buf, int mode, size_t tile, int S, int MT, int st, int mt, int lane) {
  if (!blk_is16(mode, st)) return ld_blk(buf, mode, tile, S, MT, st, mt, lane);
  const char* b = reinterpret_cast<const char*>(buf) + tile * blk_tile_bytes(mode, S, MT) + blk_off(mode, MT, st, mt);
  const float2 v = *reinterpret_cast<const float2*>(b + lane * 8);
  return f32x4{v.x, v.y, 0.f, 0.f};
}
__device__ __forceinline__ bf16x4 raw_bf4(f32x4 raw) {
  float2 v;
  v.x = raw[0];
  v.y = raw[1];
  return __builtin_bit_cast(bf16x4, v);
}
__device__ __forceinline__ f32x4 blk_val(f32x4 raw, int mode, int st) {
  if (!blk_is16(mode, st)) return raw;
  return bf4_to_f32(raw_bf4(raw));
}
__device__ __forceinline__ void st_blk(float* buf, int mode, size_t tile, int S, int MT, int st, int mt, int lane, f32x4 v) {
  char* b = reinterpret_cast<char*>(buf) + tile * blk_tile_bytes(mode, S, MT) + blk_off(mode, MT, st, mt);
  if (!blk_is16(mode, st))
    st4(reinterpret_cast<float*>(b) + lane * 4, v);
  else
    *reinterpret_cast<bf16x4*>(b + lane * 8) = to_bf4(v);
}

// Optional 16-byte stores without a branch: a raw buffer descriptor over [base, base + bytes) -- built from wave-uniform
// values only -- drops every store that falls outside it, so bytes = 0 switches the stores off in hardware and the
// surrounding basic block stays in one piece (a uniform `if (ptr)` would split the MFMA / VALU interleaving of the loop).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t opt_store_rsrc(float* base, unsigned bytes) {
  const unsigned long long u = (unsigned long long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0,
                                           __builtin_amdgcn_readfirstlane(base ? bytes : 0u), 0x00020000);
}
// 16-byte loads through a buffer descriptor: wave-uniform base in SGPRs, ONE per-lane offset register, the block offset as
// the instruction's scalar offset -- kernels that keep many weight fragments in flight would otherwise hold a 64-bit
// per-lane address pair for every one of them.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t load_rsrc(const void* base, unsigned bytes) {
  const unsigned long long u = (unsigned long long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 buf_ld16(__amdgpu_buffer_rsrc_t r, int lane_byte_off, int uniform_byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, lane_byte_off, uniform_byte_off, 0);
}
__device__ __forceinline__ float buf_ld4(__amdgpu_buffer_rsrc_t r, int lane_byte_off, int uniform_byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, lane_byte_off, uniform_byte_off, 0));
}
__device__ __forceinline__ void buf_st16(__amdgpu_buffer_rsrc_t r, int lane_byte_off, int uniform_byte_off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, lane_byte_off, uniform_byte_off, 0);
}
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void buf_st8(__amdgpu_buffer_rsrc_t r, int lane_byte_off, int uniform_byte_off, bf16x4 v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, lane_byte_off, uniform_byte_off, 0);
}
__device__ __forceinline__ void opt_st4(__amdgpu_buffer_rsrc_t r, int byte_off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, 0, 0);
}

// Global memory -> LDS without registers (LDS-DMA): 16 (4) bytes per lane to the wave-uniform LDS address `lptr` + 16 (4) * lane.
// Issued from an asm statement, NOT through __builtin_amdgcn_global_load_lds (round 6).  The compiler books the builtin as a
// pending LDS write and puts `s_waitcnt vmcnt(0)` in front of the first LDS read it cannot prove disjoint from the target --
// with double buffers picked by a run-time index that is every read of the OTHER buffer.  Found in the listings of both kernels
// that stage row tiles this way: k_fc1_bwd_fused waited for the tile it had just requested "an iteration ahead" in front of its
// first MFMA (phase stamps: 3,364 cycles for the first 80 input-gradient MFMAs, 1,676 for the second 80), k_fc2_fwd_bf waited
// for the acknowledgements of the output stores it had just issued (they share the counter).  An asm statement is invisible to
// that bookkeeping; the caller owns the counter waits that cover these requests (vector-memory results return in order:
// `s_waitcnt vmcnt(n)` with n = the number of vector-memory instructions issued since).
__device__ __forceinline__ void glds16(const void* gptr, void* lptr) {
  const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)(lptr));
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(la) : "memory", "m0");
}
__device__ __forceinline__ void glds4(const void* gptr, void* lptr) {
  const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)(lptr));
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(gptr), "s"(la) : "memory", "m0");
}

// Store a column-major ("D") fragment block held in registers (lane 16g+j: features 4g..4g+3 of row j) into the
// row-major ("R") image of the same 16x16 block (lane 16g'+c: rows 4g'..4g'+3 of feature c): pure addressing.
__device__ __forceinline__ void st_R(float* block, int lane, f32x4 v) {
  const int g = lane >> 4, j = lane & 15;
  float* p = block + 64 * (j >> 2) + 16 * g + (j & 3);
  p[0] = v[0];
  p[4] = v[1];
  p[8] = v[2];
  p[12] = v[3];
}

// sigma and its first three derivatives; conventions at kinks follow torch (relu'(0)=0, softplus threshold 20).
// Branch-free (selects only) so that the evaluation can be interleaved with MFMAs by the scheduler.
struct ActD {
  float s0, s1, s2, s3;
};

// The activation jets are straight-line fp32 polynomial arithmetic that runs on the SAME vector FMA lanes as the fp32
// MFMAs (tools/micro/mfma_valu_coissue.hip: on gfx950 every VALU instruction adds its ~5 issue cycles to a 33-cycle
// v_mfma_f32_16x16x4_f32, there is no co-execution), so their instruction count is paid in MFMA time.  Fused multiply-adds
// halve it; they are enabled HERE only (the library is built with -ffp-contract=off because the cell index / weights of the
// gather stage reproduce the reference's rounding bit by bit, which the jets do not need: they are compared to tolerance).
#define STPDE_JET_FMA _Pragma("clang fp contract(fast)")

// Hardware transcendentals only (v_exp_f32 / v_log_f32 / v_rcp_f32, 1 ulp each): no library call sequences, whose
// denormal handling costs ~10 extra instructions and makes the compiler wrap them in a divergent branch -- the activation
// jets must stay straight-line code inside the MFMA basic blocks so that the scheduler can interleave them with MFMAs.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }  // 1 ulp
// log(1 + e) for e in [0, 1]: series below 2^-10 (where log(1+e) would cancel), v_log_f32 above (argument in [1, 2])
__device__ __forceinline__ float log1p_unit(float e) {
  STPDE_JET_FMA
  const float u = 1.f + e;
  const float big = __builtin_amdgcn_logf(u) * 0.693147180559945309f;
  const float small = e * (1.f - e * (0.5f - e * 0.33333334f));
  return e < 9.765625e-4f ? small : big;
}


template <int ACT>
__device__ __forceinline__ ActD act_eval_t(float prm, float a) {
  STPDE_JET_FMA
  ActD r;
  if (ACT == STPDE_ACT_TANH) {
    const float e = fast_exp(-2.f * fabsf(a));
    const float t0 = (1.f - e) * fast_rcp(1.f + e);
    const float t = a < 0.f ? -t0 : t0;
    const float u = 1.f - t * t;
    r.s0 = t;
    r.s1 = u;
    r.s2 = -2.f * t * u;
    r.s3 = -2.f * u * (1.f - 3.f * t * t);
  } else if (ACT == STPDE_ACT_RELU) {
    const float m = a > 0.f ? 1.f : 0.f;
    r.s0 = a * m;
    r.s1 = m;
    r.s2 = 0.f;
    r.s3 = 0.f;
  } else if (ACT == STPDE_ACT_LEAKYRELU) {
    const float m = a > 0.f ? 1.f : 0.01f;
    r.s0 = a * m;
    r.s1 = m;
    r.s2 = 0.f;
    r.s3 = 0.f;
  } else if (ACT == STPDE_ACT_SOFTPLUS) {
    const float e = fast_exp(-fabsf(a));
    const float inv = fast_rcp(1.f + e);
    const float s = a >= 0.f ? inv : e * inv;   // sigmoid(a)
    const float q = e * inv * inv;              // s (1 - s)
    const bool big = a > 20.f;                  // torch threshold
    // above the threshold e < 2.1e-9, so max(a, 0) + log1p(e) rounds to a exactly: no select on the value (a select
    // with the log on one side makes the compiler emit a divergent branch, which splits the MFMA basic block)
    r.s0 = fmaxf(a, 0.f) + log1p_unit(e);
    r.s1 = big ? 1.f : s;
    r.s2 = big ? 0.f : q;
    r.s3 = big ? 0.f : q * (1.f - 2.f * s);
  } else if (ACT == STPDE_ACT_ELU) {
    const bool pos = a > 0.f;
    const float e = fast_exp(fminf(a, 0.f));
    r.s0 = pos ? a : expm1f(fminf(a, 0.f));
    r.s1 = pos ? 1.f : e;
    r.s2 = pos ? 0.f : e;
    r.s3 = pos ? 0.f : e;
  } else {  // STPDE_ACT_SWISH: x * sigmoid(beta x)
    const float ba = prm * a;
    const float e = fast_exp(-fabsf(ba));
    const float inv = fast_rcp(1.f + e);
    const float s = ba >= 0.f ? inv : e * inv;
    const float q = e * inv * inv;
    const float c = 1.f - 2.f * s;
    r.s0 = a * s;
    r.s1 = s + ba * q;
    r.s2 = prm * q * (2.f + ba * c);
    r.s3 = prm * prm * q * (3.f * c + ba * (c * c - 2.f * q));
  }
  return r;
}

// ACT >= 0: compile-time activation; ACT < 0: run-time (wave-uniform) switch
template <int ACT>
__device__ __forceinline__ ActD act_eval(int act, float prm, float a) {
  if (ACT >= 0) return act_eval_t<(ACT >= 0 ? ACT : 0)>(prm, a);
  switch (act) {
    case STPDE_ACT_TANH: return act_eval_t<STPDE_ACT_TANH>(prm, a);
    case STPDE_ACT_RELU: return act_eval_t<STPDE_ACT_RELU>(prm, a);
    case STPDE_ACT_SOFTPLUS: return act_eval_t<STPDE_ACT_SOFTPLUS>(prm, a);
    case STPDE_ACT_ELU: return act_eval_t<STPDE_ACT_ELU>(prm, a);
    case STPDE_ACT_LEAKYRELU: return act_eval_t<STPDE_ACT_LEAKYRELU>(prm, a);
    default: return act_eval_t<STPDE_ACT_SWISH>(prm, a);
  }
}

__device__ __forceinline__ float sel3(int d, float a0, float a1, float a2) { return d == 0 ? a0 : (d == 1 ? a1 : a2); }

// ---- packed fp32 (round 5) ------------------------------------------------------------------------------------------
// A wave64 VALU instruction occupies its SIMD for 4 cycles whether it is v_fma_f32 or v_pk_fma_f32 -- the packed forms
// (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two fp32 per lane) are how gfx950 reaches its fp32 vector peak -- and on the
// fp32 MFMA kernels every VALU issue cycle is an MFMA cycle (tools/micro/mfma_valu_coissue.hip).  The compiler's SLP pass
// packs almost nothing of the scalar jet code (phase stamps of the one-wave-per-SIMD kernel jet_fc1_bwd.hip: 4.7 cycles per
// instruction of a pure vector phase = issue-bound), so the jets are written on pairs of elements: the four values a lane
// holds of a fragment block go through as two f32x2.  Transcendentals, selects and max stay per component (no packed forms).
#ifndef STPDE_JET_PACKED
#define STPDE_JET_PACKED 1
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <class T>
struct ActDT {
  T s0, s1, s2, s3;
};
__device__ __forceinline__ f32x2 sel2(bool c0, bool c1, f32x2 a, f32x2 b) { return f32x2{c0 ? a.x : b.x, c1 ? a.y : b.y}; }
__device__ __forceinline__ f32x2 sel3(int d, f32x2 a0, f32x2 a1, f32x2 a2) { return d == 0 ? a0 : (d == 1 ? a1 : a2); }

template <int ACT>
__device__ __forceinline__ ActDT<f32x2> act_eval2_t(float prm, f32x2 a) {
  STPDE_JET_FMA
  ActDT<f32x2> r;
  if (ACT == STPDE_ACT_SOFTPLUS) {
    // the same expression sequence as act_eval_t<SOFTPLUS>, two elements at a time
    const f32x2 t = f32x2{-fabsf(a.x), -fabsf(a.y)} * 1.44269504088896341f;
    const f32x2 e = f32x2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    const f32x2 u = e + 1.f;
    const f32x2 inv = f32x2{fast_rcp(u.x), fast_rcp(u.y)};
    const f32x2 einv = e * inv;
    const f32x2 s = sel2(a.x >= 0.f, a.y >= 0.f, inv, einv);
    const f32x2 q = einv * inv;
    const bool b0 = a.x > 20.f, b1 = a.y > 20.f;
    const f32x2 lbig = f32x2{__builtin_amdgcn_logf(u.x), __builtin_amdgcn_logf(u.y)} * 0.693147180559945309f;
    const f32x2 lsmall = e * (1.f - e * (0.5f - e * 0.33333334f));
    const f32x2 l1p = sel2(e.x < 9.765625e-4f, e.y < 9.765625e-4f, lsmall, lbig);
    r.s0 = f32x2{fmaxf(a.x, 0.f), fmaxf(a.y, 0.f)} + l1p;
    const f32x2 one = f32x2{1.f, 1.f}, zero = f32x2{0.f, 0.f};
    r.s1 = sel2(b0, b1, one, s);
    r.s2 = sel2(b0, b1, zero, q);
    r.s3 = sel2(b0, b1, zero, q * (1.f - 2.f * s));
  } else {
    const ActD x = act_eval_t<ACT>(prm, a.x), y = act_eval_t<ACT>(prm, a.y);
    r.s0 = f32x2{x.s0, y.s0};
    r.s1 = f32x2{x.s1, y.s1};
    r.s2 = f32x2{x.s2, y.s2};
    r.s3 = f32x2{x.s3, y.s3};
  }
  return r;
}
template <int ACT>
__device__ __forceinline__ ActDT<f32x2> act_eval2(int act, float prm, f32x2 a) {
  if (ACT >= 0) return act_eval2_t<(ACT >= 0 ? ACT : 0)>(prm, a);
  const ActD x = act_eval<ACT>(act, prm, a.x), y = act_eval<ACT>(act, prm, a.y);
  ActDT<f32x2> r;
  r.s0 = f32x2{x.s0, y.s0};
  r.s1 = f32x2{x.s1, y.s1};
  r.s2 = f32x2{x.s2, y.s2};
  r.s3 = f32x2{x.s3, y.s3};
  return r;
}
// one evaluation interface for both widths
template <int ACT>
__device__ __forceinline__ ActDT<float> act_eval_w(int act, float prm, float a) {
  const ActD s = act_eval<ACT>(act, prm, a);
  return ActDT<float>{s.s0, s.s1, s.s2, s.s3};
}
template <int ACT>
__device__ __forceinline__ ActDT<f32x2> act_eval_w(int act, float prm, f32x2 a) {
  return act_eval2<ACT>(act, prm, a);
}

// Forward jet of the activation on one fragment block: pre[S] (a, adot_d, addot_p) -> h[S].
// S2 == 1 is the COMBINED second-order stream: cq[0..5] = per-row weights of adot_a*adot_b over the canonical pairs
// (0,0) (0,1) (0,2) (1,1) (1,2) (2,2); otherwise cq is unused.
// S1 == 0 && S2 > 0 is the VALUE-TILE mode of the forward-only (inference) kernels: the 1 + S2 "streams" are the value
// streams of 1 + S2 independent, consecutive row tiles that share one pass over the weights.
// (T = float: one element; T = f32x2: two elements of the lane's four, packed arithmetic)
template <int S1, int S2, int ACT, class T>
__device__ __forceinline__ void act_jet_fwd_w(const stpde_jet_cfg& cfg, const T* pre, T* h, const float* cq) {
  STPDE_JET_FMA
  const ActDT<T> s = act_eval_w<ACT>(cfg.act, cfg.act_param, pre[0]);
  h[0] = s.s0;
  if (S1 == 3) {
    const T a0 = pre[1], a1 = pre[2], a2 = pre[3];
    h[1] = s.s1 * a0;
    h[2] = s.s1 * a1;
    h[3] = s.s1 * a2;
    if (S2 == 1) {
      const T q = a0 * (cq[0] * a0 + cq[1] * a1 + cq[2] * a2) + a1 * (cq[3] * a1 + cq[4] * a2) + cq[5] * a2 * a2;
      h[4] = s.s2 * q + s.s1 * pre[4];
      return;
    }
#pragma unroll
    for (int p = 0; p < S2; ++p) {
      const T u = sel3(cfg.pair0[p], a0, a1, a2), v = sel3(cfg.pair1[p], a0, a1, a2);
      h[4 + p] = s.s2 * u * v + s.s1 * pre[4 + p];
    }
  }
}
template <int S1, int S2, int ACT>
__device__ __forceinline__ void act_jet_fwd(const stpde_jet_cfg& cfg, const f32x4* pre, f32x4* h,
                                            const float* cq = nullptr) {
  STPDE_JET_FMA
  constexpr int S = 1 + S1 + S2;
  if (S1 == 0 && S2 > 0) {
#pragma unroll
    for (int st = 0; st < 1 + S2; ++st) {
#if STPDE_JET_PACKED
      h[st].lo = act_eval_w<ACT>(cfg.act, cfg.act_param, (f32x2)pre[st].lo).s0;
      h[st].hi = act_eval_w<ACT>(cfg.act, cfg.act_param, (f32x2)pre[st].hi).s0;
#else
#pragma unroll
      for (int r = 0; r < 4; ++r) h[st][r] = act_eval<ACT>(cfg.act, cfg.act_param, pre[st][r]).s0;
#endif
    }
    return;
  }
#if STPDE_JET_PACKED
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    f32x2 p2[S], h2[S];
#pragma unroll
    for (int st = 0; st < S; ++st) p2[st] = half ? (f32x2)pre[st].hi : (f32x2)pre[st].lo;
    act_jet_fwd_w<S1, S2, ACT, f32x2>(cfg, p2, h2, cq);
#pragma unroll
    for (int st = 0; st < S; ++st) {
      if (half) h[st].hi = h2[st];
      else h[st].lo = h2[st];
    }
  }
#else
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float p1[S], h1[S];
#pragma unroll
    for (int st = 0; st < S; ++st) p1[st] = pre[st][r];
    act_jet_fwd_w<S1, S2, ACT, float>(cfg, p1, h1, cq);
#pragma unroll
    for (int st = 0; st < S; ++st) h[st][r] = h1[st];
  }
#endif
}

// Adjoint: given pre[S] and hbar[S], produce abar[S] (adjoint of the pre-activation streams).
template <int S1, int S2, int ACT, class T>
__device__ __forceinline__ void act_jet_adj_w(const stpde_jet_cfg& cfg, const T* pre, const T* hbar, T* abar, const float* cq) {
  STPDE_JET_FMA
  const ActDT<T> s = act_eval_w<ACT>(cfg.act, cfg.act_param, pre[0]);
  T ab = s.s1 * hbar[0];
  if (S1 == 3) {
    const T a0 = pre[1], a1 = pre[2], a2 = pre[3];
    T d0 = s.s1 * hbar[1], d1 = s.s1 * hbar[2], d2 = s.s1 * hbar[3];
    ab += s.s2 * (a0 * hbar[1] + a1 * hbar[2] + a2 * hbar[3]);
    if (S2 == 1) {
      const T hb = hbar[4];
      const T q = a0 * (cq[0] * a0 + cq[1] * a1 + cq[2] * a2) + a1 * (cq[3] * a1 + cq[4] * a2) + cq[5] * a2 * a2;
      ab += (s.s3 * q + s.s2 * pre[4]) * hb;
      const T t = s.s2 * hb;
      d0 += t * (2.f * cq[0] * a0 + cq[1] * a1 + cq[2] * a2);
      d1 += t * (cq[1] * a0 + 2.f * cq[3] * a1 + cq[4] * a2);
      d2 += t * (cq[2] * a0 + cq[4] * a1 + 2.f * cq[5] * a2);
      abar[4] = s.s1 * hb;
    }
#pragma unroll
    for (int p = 0; p < (S2 == 1 ? 0 : S2); ++p) {
      const int e0 = cfg.pair0[p], e1 = cfg.pair1[p];
      const T u = sel3(e0, a0, a1, a2), v = sel3(e1, a0, a1, a2);
      const T hb = hbar[4 + p];
      ab += (s.s3 * u * v + s.s2 * pre[4 + p]) * hb;
      const T t0 = s.s2 * v * hb;  // d hdd / d adot_{e0}
      const T t1 = s.s2 * u * hb;  // d hdd / d adot_{e1}
      const T z = t0 * 0.f;
      d0 += (e0 == 0 ? t0 : z) + (e1 == 0 ? t1 : z);
      d1 += (e0 == 1 ? t0 : z) + (e1 == 1 ? t1 : z);
      d2 += (e0 == 2 ? t0 : z) + (e1 == 2 ? t1 : z);
      abar[4 + p] = s.s1 * hb;
    }
    abar[1] = d0;
    abar[2] = d1;
    abar[3] = d2;
  }
  abar[0] = ab;
}
template <int S1, int S2, int ACT>
__device__ __forceinline__ void act_jet_adj(const stpde_jet_cfg& cfg, const f32x4* pre, const f32x4* hbar,
                                            f32x4* abar, const float* cq = nullptr) {
  constexpr int S = 1 + S1 + S2;
#if STPDE_JET_PACKED
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    f32x2 p2[S], hb2[S], ab2[S];
#pragma unroll
    for (int st = 0; st < S; ++st) {
      p2[st] = half ? (f32x2)pre[st].hi : (f32x2)pre[st].lo;
      hb2[st] = half ? (f32x2)hbar[st].hi : (f32x2)hbar[st].lo;
    }
    act_jet_adj_w<S1, S2, ACT, f32x2>(cfg, p2, hb2, ab2, cq);
#pragma unroll
    for (int st = 0; st < S; ++st) {
      if (half) abar[st].hi = ab2[st];
      else abar[st].lo = ab2[st];
    }
  }
#else
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float p1[S], hb1[S], ab1[S];
#pragma unroll
    for (int st = 0; st < S; ++st) {
      p1[st] = pre[st][r];
      hb1[st] = hbar[st][r];
    }
    act_jet_adj_w<S1, S2, ACT, float>(cfg, p1, hb1, ab1, cq);
#pragma unroll
    for (int st = 0; st < S; ++st) abar[st][r] = ab1[st];
  }
#endif
}

// Adjoint of the learnable swish beta (reference nonlinearities.py:5-12): sum over the block of
// hbar_s * d h_s / d beta, with z = beta a, s = sigmoid(z), q = s(1-s), c = 1-2s:
//   d s0/d beta = a^2 q,  d s1/d beta = a q (2 + z c),  d s2/d beta = q (2 + z c) + z q (3c + z (c^2 - 2q)).
template <int S1, int S2>
__device__ __forceinline__ float swish_beta_adj(const stpde_jet_cfg& cfg, const f32x4* pre, const f32x4* hbar,
                                                const float* cq) {
  STPDE_JET_FMA
  float sum = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float a = pre[0][r], z = cfg.act_param * a;
    const float e = fast_exp(-fabsf(z));
    const float inv = fast_rcp(1.f + e);
    const float s = z >= 0.f ? inv : e * inv;
    const float q = e * inv * inv, c = 1.f - 2.f * s;
    const float m = q * (2.f + z * c);
    const float b0 = a * a * q, b1 = a * m, b2 = m + z * q * (3.f * c + z * (c * c - 2.f * q));
    float t = b0 * hbar[0][r];
    if (S1 == 3) {
      const float a0 = pre[1][r], a1 = pre[2][r], a2 = pre[3][r];
      t += b1 * (a0 * hbar[1][r] + a1 * hbar[2][r] + a2 * hbar[3][r]);
      if (S2 == 1) {
        const float qq = a0 * (cq[0] * a0 + cq[1] * a1 + cq[2] * a2) + a1 * (cq[3] * a1 + cq[4] * a2) + cq[5] * a2 * a2;
        t += (b2 * qq + b1 * pre[4][r]) * hbar[4][r];
      }
#pragma unroll
      for (int p = 0; p < (S2 == 1 ? 0 : S2); ++p) {
        const float u = sel3(cfg.pair0[p], a0, a1, a2), v = sel3(cfg.pair1[p], a0, a1, a2);
        t += (b2 * u * v + b1 * pre[4 + p][r]) * hbar[4 + p][r];
      }
    }
    sum += t;
  }
  return sum;
}

// Sum over the 16 lanes of a DPP row (lanes 16g .. 16g+15 = the 16 rows of a fragment block for one feature group):
// four v_add_f32 with row_shr DPP modifiers on the first operand (lanes shifted in from outside the 16-lane row read 0);
// lane 16g+15 ends up with the full sum.  Written as inline assembly because the compiler otherwise emits v_mov_b32_dpp +
// v_add_f32 (two VALU issues, and on gfx950 VALU issue cycles are MFMA cycles).
// HAZARD: a DPP instruction must not read a VGPR that a VALU instruction wrote less than two wait states earlier, and the
// compiler's hazard recognizer does not look inside an asm statement.  Every DPP add of this file therefore sits in ONE
// asm volatile block that (a) opens with `s_nop 1` (covers whatever VALU wrote the operand in front of the block) and
// (b) keeps two wait states between consecutive shifts of the same register (s_nop 1 here; three other DPP adds in
// row_sum16x4).  tools/check_dpp_hazard.py scans the disassembly of the built library for violations (run by build()).
__device__ __forceinline__ float row_sum16(float v) {
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1"
      : "+v"(v));
  return v;
}

// The same sum for the four registers of a fragment block in ONE asm statement: every register's consecutive shifts are three
// instructions apart and the statement opens with the wait states for whatever wrote v in front of it.  (History: with two
// waves per SIMD the other wave's instructions happened to sit between back-to-back DPP adds; the one-wave-per-SIMD kernel of
// round 5, jet_fc1_bwd.hip, produced wrong row sums with four scalar row sums in a row.  Round 6: no unprotected form is left.)
__device__ __forceinline__ f32x4 row_sum16x4(f32x4 v) {
  float a = v[0], b = v[1], c = v[2], d = v[3];
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %2, %2, %2 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %3, %3, %3 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %3, %3, %3 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %2, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %3, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1"
      : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  return f32x4{a, b, c, d};
}

// ---- Order-independent ("long accumulator") sums: the deterministic mode of the U-Net kernels (round 6) -----------------------
// An fp32 / fp64 atomic sum depends on the order the hardware serves the atomics in.  Integer addition does not: a value is
// split exactly into 32-bit pieces at FIXED binary positions and each piece is added to its own signed 64-bit accumulator
// ("window") with an integer atomic -- whatever the order, the K windows end up holding the same integers, and det_value()
// turns them into the same floating-point number.  Window j weighs 2^(STPDE_DET_BASE + 32 j); a window takes 2^31 pieces before
// it can overflow.  Range: bit positions 2^-110 ... 2^+81 (values up to ~2^80; magnitudes below 2^-86 (float) / 2^-57 (double)
// lose their lowest bits -- truncated the same way every time).  A non-finite value poisons the top window (-> NaN).
// Layout: accumulator e of an array = windows [e * STPDE_DET_K ... + K); the caller zero-fills.
#define STPDE_DET_BASE (-110)      // (STPDE_DET_K = 6 windows: include/stpde_hip.h)
__device__ __forceinline__ void det_add_pieces(long long* acc, unsigned long long m, int shift, bool neg) {
  // |value| = m * 2^(BASE + shift), m < 2^53
  if (shift < 0) {
    m = shift > -64 ? (m >> (-shift)) : 0ull;
    shift = 0;
  }
  const int j = shift >> 5, r = shift & 31;
  const unsigned long long lo64 = m << r;                       // bits 0 .. 63 of m << r
  const unsigned long long hi = r ? (m >> (64 - r)) : 0ull;     // bits 64 .. of m << r (< 2^21)
  unsigned long long p0 = lo64 & 0xffffffffull, p1 = lo64 >> 32, p2 = hi;
  if (j >= STPDE_DET_K || (p1 && j + 1 >= STPDE_DET_K) || (p2 && j + 2 >= STPDE_DET_K)) {
    // too large for the windows (>= 2^70): poison the top window (-> NaN at read-out)
    atomicAdd(reinterpret_cast<unsigned long long*>(acc + STPDE_DET_K - 1), 1ull << 62);
    return;
  }
  if (neg) {
    p0 = 0ull - p0;
    p1 = 0ull - p1;
    p2 = 0ull - p2;
  }
  unsigned long long* a = reinterpret_cast<unsigned long long*>(acc) + j;
  if (p0) atomicAdd(a, p0);
  if (p1) atomicAdd(a + 1, p1);
  if (p2) atomicAdd(a + 2, p2);
}
__device__ __forceinline__ void det_add_f32(long long* acc, float v) {
  const unsigned b = __float_as_uint(v);
  const int ex = (b >> 23) & 0xff;
  if (ex == 0xff) {
    atomicAdd(reinterpret_cast<unsigned long long*>(acc + STPDE_DET_K - 1), 1ull << 62);
    return;
  }
  const unsigned long long m = ex ? ((b & 0x7fffffu) | 0x800000u) : (b & 0x7fffffu);
  if (!m) return;
  // value = m * 2^((ex ? ex : 1) - 150)
  det_add_pieces(acc, m, (ex ? ex : 1) - 150 - STPDE_DET_BASE, (b >> 31) != 0);
}
__device__ __forceinline__ void det_add_f64(long long* acc, double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const int ex = (int)((b >> 52) & 0x7ff);
  if (ex == 0x7ff) {
    atomicAdd(reinterpret_cast<unsigned long long*>(acc + STPDE_DET_K - 1), 1ull << 62);
    return;
  }
  const unsigned long long m = ex ? ((b & 0xfffffffffffffull) | 0x10000000000000ull) : (b & 0xfffffffffffffull);
  if (!m) return;
  det_add_pieces(acc, m, (ex ? ex : 1) - 1075 - STPDE_DET_BASE, (b >> 63) != 0);
}
// the sum as a double: the windows from the top down (each term exact or rounded the same way every time)
__device__ __forceinline__ double det_value(const long long* acc) {
  const long long top = acc[STPDE_DET_K - 1];
  if (top >= (1ll << 61) || top <= -(1ll << 61)) return __longlong_as_double(0x7ff8000000000000ll);
  double s = 0.;
#pragma unroll
  for (int j = STPDE_DET_K - 1; j >= 0; --j) s += ldexp((double)acc[j], STPDE_DET_BASE + 32 * j);
  return s;
}
// one destination element of a sum the kernels accumulate with atomics: plain fp32 atomic, or -- det -- long accumulator
__device__ __forceinline__ void acc_add_f32(float* dst, size_t idx, float v, int det) {
  if (det)
    det_add_f32(reinterpret_cast<long long*>(dst) + idx * STPDE_DET_K, v);
  else
    atomicAdd(dst + idx, v);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}


extern int stpde_trace_on;
void stpde_trace_note(const char* launcher, const char* kernel);

// Launch with a clean error slot: hipGetLastError() is sticky per thread, a stale error from an unrelated earlier
// runtime call must not be attributed to this launch.
// With tracing enabled (stpde_trace_enable, a test / debugging facility) every launch also notes WHICH template
// instantiation was dispatched: the enclosing launcher's __PRETTY_FUNCTION__ (carries the template argument values)
// plus the kernel expression as written (carries the literal flags).
#define STPDE_LAUNCH(kern, ...)                                              \
  do {                                                                       \
    (void)hipGetLastError();                                                 \
    if (stpde_trace_on) stpde_trace_note(__PRETTY_FUNCTION__, #kern);        \
    hipLaunchKernelGGL(kern, __VA_ARGS__);                                   \
  } while (0)

// per-row weights of the combined second-order stream: cw[P][8], point of row j of a tile = 2*tile + (j >> 3)
template <int S2>
__device__ __forceinline__ void load_cq(const float* cw, int point, float* cq) {
  if (S2 == 1) {
    // (16 + 8 bytes: where the second load stayed a 16-byte one, the allocator re-used its two dead result registers at once
    // and had to wait for the load in front of that write -- `s_waitcnt vmcnt(0)` right behind a prefetch, k_fc1_fwd_spec)
    const f32x4 v0 = ld4(cw + (size_t)point * 8);
    const float2 v1 = *reinterpret_cast<const float2*>(cw + (size_t)point * 8 + 4);
    cq[0] = v0[0];
    cq[1] = v0[1];
    cq[2] = v0[2];
    cq[3] = v0[3];
    cq[4] = v1.x;
    cq[5] = v1.y;
  }
}

// The same weights requested a row tile AHEAD of their use: the two loads' results stay whole vectors until unpack_cq() where
// they are used.  (As six loop-carried floats the allocator copied single elements into other registers right behind the
// loads -- `s_waitcnt vmcnt(1) ; v_mov_b32` in the listing of k_fc1_fwd_spec: the prefetch was waited for at once, round 6.)
struct CqRaw {
  f32x4 a;
  float2 b;
};
template <int S2>
__device__ __forceinline__ void load_cq_raw(const float* cw, int point, CqRaw& r) {
  if (S2 == 1) {
    r.a = ld4(cw + (size_t)point * 8);
    r.b = *reinterpret_cast<const float2*>(cw + (size_t)point * 8 + 4);
  }
}
template <int S2>
__device__ __forceinline__ void unpack_cq(const CqRaw& r, float* cq) {
  if (S2 == 1) {
    cq[0] = r.a[0];
    cq[1] = r.a[1];
    cq[2] = r.a[2];
    cq[3] = r.a[3];
    cq[4] = r.b.x;
    cq[5] = r.b.y;
  }
}

// Host-side helpers (api.cpp)
// mean / biased variance / rstd of one channel from the double-format BatchNorm sums ([STPDE_BN_REP][2][C]: sum x, sum x^2);
// every consumer (k_bn_apply, the on-load transform of k_conv_fused) calls this one function, so all of them see the same bits
// (det: the same sums as ONE set of long accumulators [2][C][STPDE_DET_K], stpde_bn_desc.det)
__device__ __forceinline__ void bn_stat_f64(const double* sums, int C, int c, long N, float eps, int det, float& mean, float& var,
                                            float& rstd) {
  double s1 = 0., s2 = 0.;
  if (det) {
    const long long* acc = reinterpret_cast<const long long*>(sums);
    s1 = det_value(acc + (size_t)c * STPDE_DET_K);
    s2 = det_value(acc + ((size_t)C + c) * STPDE_DET_K);
  } else {
    for (int r = 0; r < STPDE_BN_REP; ++r) {
      s1 += sums[(size_t)(2 * r) * C + c];
      s2 += sums[(size_t)(2 * r + 1) * C + c];
    }
  }
  const double m = s1 / (double)N;
  double v = s2 / (double)N - m * m;
  if (v < 0.) v = 0.;
  mean = (float)m;
  var = (float)v;
  rstd = 1.f / sqrtf(var + eps);
}
int stpde_bn_stats_f64(const float* x, long N, int C, double* sums, hipStream_t stream);
void stpde_set_error(const char* fmt, ...);
int stpde_check_launch(const char* what);
int stpde_tune_get(int key);      // test overrides of launch geometry (api.cpp: stpde_tune; 0 = the library's own choice)
