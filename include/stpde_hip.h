/*
 * stpde_hip.h -- C ABI of libstpde_hip.so: the MI355X (gfx950) device path behind the MeshfreeFlowNet
 * operator API (PDELayer / query_local_implicit_grid / ImNet / regular_nd_grid_interpolation).
 *
 * The reference (maxjiang93/space_time_pde) is pure Python on PyTorch and has NO native interface; each
 * entry point below states which reference code (file:line under /root/reference) it replaces.  The
 * binding a maintainer adds is the ctypes stub shown in INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into memory owned by the caller (PyTorch caching allocator);
 *     the library never allocates, frees or retains device memory;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued there, nothing synchronises;
 *   - return value: 0 = ok, otherwise one of STPDE_E_*; stpde_last_error() gives the text;
 *   - all arithmetic is IEEE fp32 (no fast-math, no FMA contraction outside MFMA), indices int32.
 *
 * Fragment ("stash") layout used by the jet kernels.  Query points are grouped in tiles of 2 points =
 * 16 corner rows (row j = 8*point_in_tile + corner, corner = 4*b0+2*b1+b2 as
 * src/regular_nd_grid_interpolation.py:55-56).  A block of 16 rows x 16 features is stored as 64 lanes x 4
 * floats (1 KiB): lane = 16*g + j holds features 4g..4g+3 of row j -- exactly the C/D register image of
 * v_mfma_f32_16x16x4_f32 for out^T = W * in^T, which is also the B-operand image of the next layer, so
 * layers chain without any re-layout.  A layer buffer is [tile][stream][feature_tile][64][4] floats.
 * Streams: 0 = value, 1..3 = d/dr_k (if S1 == 3), then S2 second-order pairs (pair0[k], pair1[k]).
 */
#ifndef STPDE_HIP_H
#define STPDE_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define STPDE_OK 0
#define STPDE_E_BADARG 1      /* shape / size / enum out of the supported range -> ValueError */
#define STPDE_E_UNSUPPORTED 2 /* stream configuration not compiled in            -> NotImplementedError */
#define STPDE_E_LAUNCH 3      /* HIP launch failure                              -> RuntimeError */

#define STPDE_ACT_TANH 0
#define STPDE_ACT_RELU 1
#define STPDE_ACT_SOFTPLUS 2
#define STPDE_ACT_ELU 3
#define STPDE_ACT_SWISH 4
#define STPDE_ACT_LEAKYRELU 5

#define STPDE_XT 3 /* the augmented raw input [r(3) ; latent(c) ; 1 ; 0-pad] occupies 3 feature tiles, the third one sparse (4 live slots): 3 + c + 1 <= 36, i.e. c <= 32 */
#define STPDE_PBAR_SLOTS 64 /* accumulation slots of the swish-beta adjoint (stpde_jet_layer_bwd), summed by the caller */

/* Derivative-stream configuration shared by the jet kernels. */
typedef struct {
  int S1;        /* 0 (value only) or 3 (value + d/dr_0..2)                         */
  int S2;        /* number of second-order streams: 0, 1 (combined, see below), 2 or 6 compiled in.
                  * S1 == 0 with S2 == 3 is the forward-only VALUE-TILE mode of stpde_jet_layer_fwd: the 4 "streams" are
                  * the value streams of 4 consecutive row tiles (desc.ntiles counts groups of 4 tiles), which share one
                  * pass over the weights; buffers are laid out exactly as for S1 = S2 = 0 with 4 * ntiles tiles. */
  int pair0[6];  /* second-order stream k is d2/dr_pair0[k] dr_pair1[k]             */
  int pair1[6];
  int act;       /* STPDE_ACT_*  (src/nonlinearities.py:15-22)                       */
  float act_param; /* swish beta; leaky-relu slope is fixed at 0.01 (torch default) */
  /* Combined second-order stream (S2 == 1): when every equation uses the second derivatives only through ONE linear
   * combination  L y = sum_k alpha[k] d2y/dq_a dq_b  over the canonical pairs (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
   * (e.g. the anisotropic Laplacian of the Rayleigh-Benard equations), the MLP carries that single stream instead of
   * one per pair: F' = sigma''(a) * sum_k c_k adot_a adot_b + sigma'(a) * (W F), with per-point weights
   * c_k = alpha[k] * kappa_a * kappa_b written by stpde_lig_gather into cw[P][8]. */
  int combo;     /* 1: S2 must be 1 and the stream is the combination above          */
  float alpha[6];
} stpde_jet_cfg;

int stpde_version(void);
int stpde_last_error(char* buf, unsigned long n);
/* Launch-geometry overrides for TESTS (round 6; nothing in the reference corresponds to it, and the library reads no
 * environment variable): the persistent-grid convolution kernels pick their grid and their minimum volume themselves; a
 * test forces other values to reach their multi-block and ragged-tail paths on volumes it can afford.  value 0 = the library's
 * own choice (the default of every key).  Keys: "conv3_lds_off" (1: the 3x3x3 LDS-tile kernel is not used),
 * "conv3_lds_minblk" (minimum number of blocks for it), "conv3_lds_gx", "conv_wgrad_lds_gx", "conv1_wgrad_lds_gx" (grid sizes).
 * Returns the previous value, -1 for an unknown key.  Process-wide, thread-safe. */
#define STPDE_TUNE_CONV3_LDS_OFF 0
#define STPDE_TUNE_CONV3_LDS_MINBLK 1
#define STPDE_TUNE_CONV3_LDS_GX 2
#define STPDE_TUNE_CONV_WGRAD_LDS_GX 3
#define STPDE_TUNE_CONV1_WGRAD_LDS_GX 4
#define STPDE_TUNE_COUNT 5
int stpde_tune(const char* name, int value);
/* Dispatch trace (test / debugging facility; nothing in the reference corresponds to it): while enabled, every kernel
 * launch records which template instantiation it dispatched ("<kernel expression> @ <launcher with template
 * arguments>", unique entries).  stpde_trace_enable clears the set; stpde_trace_read copies the newline-separated
 * entries (truncated to n - 1 bytes) and returns the number of bytes needed for all of them. */
int stpde_trace_enable(int on);
long stpde_trace_read(char* buf, unsigned long n);

/* ---- a1/a2: clip, cell index, corner gather, weights, relative coords ------------------------------
 * Replaces regular_nd_grid_interpolation_coefficients (src/regular_nd_grid_interpolation.py:14-78) +
 * the torch.cat of local_implicit_grid.py:49 for dim = 3.  Writes the augmented input X
 * [ntiles][3][64][4], per-point coefficients coef[P][16] = {omega[b][d] (6), domega[b][d] (6), kappa[d] (3), 0}
 * and cell[P] = linear index of corner 0 in the latent grid.  lo_c/hi_c = xmin+eps / xmax-eps and
 * cube = (xmax-xmin)/(size-1) are computed by the caller with the reference's own fp32 expressions
 * (:48-51) so that floor(q/cube) is bit-identical (:52).  xmin must be 0 (quirk: :52 ignores xmin).
 * P must be even; pts [P][3] (a chunk of the [B*N] point list starting at p_base); latent [B][n0][n1][n2][C]
 * contiguous. */
typedef struct {
  int P, N, B, n0, n1, n2, C;
  int p_base; /* global index of pts[0]: batch of point p is min((p_base + p) / N, B - 1) */
  float lo_c[3], hi_c[3], cube[3];
  float alpha[6]; /* combined second-order stream weights (only read when cw != NULL) */
} stpde_gather_desc;
/* XR (optional, normally NULL since round 5): the same augmented input in the row-major fragment image (lane 16g+c holds
 * rows 4g..4g+3 of feature c).  No kernel of the library reads it any more: the weight-gradient kernels take X and turn
 * the fragments they need into the row-major image inside LDS (1.5 KiB per point less written and re-read). */
int stpde_lig_gather(const stpde_gather_desc* d, const float* pts, const float* latent, float* X, float* XR,
                     float* coef, int* cell, float* cw /* [P][8] or NULL */, void* stream);

/* ---- a4/a5/a6/a7: one IM-NET layer on all derivative streams --------------------------------------
 * Replaces the addmm + activation + cat of src/implicit_net.py:48-54 evaluated on the
 * [b*p*2^d, d+c] matrix of src/local_implicit_grid.py:53, together with the torch.autograd.grad sweeps of
 * src/pde.py:8-9 (forward-mode streams instead of reverse sweeps).
 * out_pre[tile][S][MT] = W_h * act_jet(in_pre[tile][S][KT]) + W_s * X + tangent consts   (pre-activations)
 * first_hidden != 0: the input is layer 0's output, computed on the fly from X (in_pre ignored). */
typedef struct {
  int ntiles, KT, MT, first_hidden;
  stpde_jet_cfg cfg;
  /* BASELINE config 4 ("bf16 MFMA MLP path"): != 0 runs the hidden-to-hidden GEMMs of the wide layers (those served
   * by the workgroup-cooperative kernels: KT % 4 == 0, KT >= 8, MT % 8 == 0) on v_mfma_f32_16x16x32_bf16 -- operands
   * rounded to bf16 (round-to-nearest-even), fp32 accumulation; layer-0 regeneration, skip/bias GEMM, activation
   * jets, stash and epilogues stay fp32.  Narrow layers (HBM-bound) keep the fp32 kernels. */
  /* mfma_bf16 == 3 ("fp32x3", forward and input-gradient kernels only): fp32-ACCURATE products on the bf16 pipe -- every
   * fp32 operand is split exactly into three bf16 terms (hi + mid + lo = 8 + 8 + 8 mantissa bits; the weights on the
   * host: Wh_pack_bf16 is then [3][KT/2][MT][64] blocks, the activations in the kernel's produce stage) and the six
   * partial products of weight >= 2^-16 are accumulated in fp32; bf16 x bf16 is exact in fp32 and the dropped products are
   * below 2^-24 of the result, i.e. below fp32 rounding.  6 bf16 MFMAs (K = 32) replace 8 fp32 MFMAs at 1/16 of the
   * per-instruction time. */
  int mfma_bf16;
  /* PACKED layer buffers (round 3, bf16 mode only; every consumer of such a buffer is a bf16-operand kernel).  Two formats:
   *   packed STASH   (buffers of pre-activations): value stream fp32, derivative streams bf16 -- per row tile
   *                  [MT][64][4] fp32, then [S-1][MT][64][4] bf16: MT * (1024 + (S-1) * 512) bytes instead of MT * S * 1024;
   *   packed ADJOINT (buffers of adjoints): every stream bf16 -- per row tile [S][MT][64][4] bf16, MT * S * 512 bytes (all
   *                  consumers round every stream to a bf16 MFMA operand; the d-latent reduction reads stream 0).  An
   *                  adjoint buffer never aliases the stash it belongs to.
   * Bit mask: 1 = in_pre (hidden input of a forward call / the pre-activations a backward kernel takes its adjoint against)
   * is a packed stash, 2 = the buffer this call WRITES is packed (out_pre of a forward call: stash; the adjoint destination
   * of stpde_jet_layer_bwd / _bwd_to, including the layer-0 adjoint of a first-hidden-layer call: adjoint format),
   * 4 = abar_out (input of the backward / weight-gradient kernels) is a packed adjoint buffer. */
  int packed;
  /* Deterministic mode (round 6; stpde_jet_wgrad / stpde_jet_fc1_bwd only): != 0 -> dW_aug addresses long accumulators (six
   * zero-filled 64-bit integers per element instead of one float, see stpde_conv3d_desc.det), into which the workgroups'
   * partial sums are added with integer atomics -- order-independent, bit-identical from run to run; stpde_det_finalize turns
   * them into fp32.  (The adjoint of a learnable swish beta keeps its fp32 atomics.) */
  int det;
} stpde_layer_desc;
/* Wh_pack_bf16 (used when d->mfma_bf16 != 0, may be NULL otherwise): [KT/2][MT][64] blocks of 8 bf16 =
 * the two fp32 blocks (2q, mt) and (2q+1, mt) of Wh_pack, lane by lane, rounded to bf16.
 * z0 (first_hidden only, may be NULL = not kept; ignored in the value-tile mode): [tile][KT][256] receives the value
 * stream of the layer-0 pre-activations the kernel regenerates from X -- the stash stpde_jet_layer_bwd(first_hidden) reads. */
int stpde_jet_layer_fwd(const stpde_layer_desc* d, const float* in_pre, const float* X, const float* Wh_pack,
                        const float* Ws_pack, const float* tanc, const float* W0s_pack, const float* tanc0,
                        float* out_pre, const float* cw /* combined-stream weights or NULL */,
                        const void* Wh_pack_bf16, float* z0, void* stream);

/* Fused forward of the three narrowest layers fc3 -> fc4 -> fc5 (src/implicit_net.py:48-54) for nf = 16 * nf16, nf16 in
 * {1, 2}: equivalent to three stpde_jet_layer_fwd calls, but the inter-layer data stays in registers (the accumulator
 * tiles of one layer are the B operand of the next); the pre-activations of all three layers are written for the backward.
 * Wh_pack / Ws_pack / tanc / out_pre: HOST arrays of 3 device pointers (layers 3, 4, 5).  Stream sets (0,0) (3,0)
 * (3,1 combined) (3,2), and the forward-only value-tile set (0,3): ntiles = row tiles / 4, tanc may be NULL. */
int stpde_jet_tail_fwd(const stpde_jet_cfg* cfg, int ntiles, int nf16, const float* in_pre2, const float* X,
                       const float* const* Wh_pack, const float* const* Ws_pack, const float* const* tanc,
                       float* const* out_pre, const float* cw, void* stream);

/* bf16 mode (stpde_imnet_plan.mfma_bf16 == 1; nf = 32, S1 = 3): the same chain with the hidden-to-hidden products on the
 * bf16 MFMA and PACKED layer buffers (stpde_layer_desc.packed).  packed: 1 = in_pre2 is packed, 2 = out_pre[0] / [1] are
 * written packed (out_pre[2], the rows of the output layer, stays fp32 blocks); Wh16_pack: the bf16 A-operand packs of the
 * three layers in TWO terms, [2][KT/2][MT][64] x 8 bf16 (hi, then lo = bf16(w - hi); two k-tiles per block) -- these layers
 * are narrow, so both operands of a product are split into two bf16 terms and the product is three bf16 MFMAs (2^-16
 * relative).  Compiled for packed == 3 with all three packs; packed == 0 and Wh16_pack == NULL: stpde_jet_tail_fwd. */
int stpde_jet_tail_fwd_p(const stpde_jet_cfg* cfg, int ntiles, int nf16, const float* in_pre2, const float* X,
                         const float* const* Wh_pack, const float* const* Ws_pack, const float* const* tanc,
                         float* const* out_pre, const float* cw, int packed, const void* const* Wh16_pack, void* stream);

/* Fused input-gradient chain of the same three layers: abar5 (adjoint of fc5's output rows, from stpde_lig_reduce_bwd)
 * -> abar4 -> abar3 -> abar2; equivalent to stpde_jet_layer_bwd on layers 5, 4, 3, but the adjoints of layers 4 and 3 feed
 * the next GEMM from the registers.  pre[0..2]: stashed pre-activations of the outputs of fc2, fc3, fc4; abar_out[0..2]:
 * where their adjoints go -- abar_out[l] may be pre[l] (in place) when no later kernel needs those pre-activations: the
 * weight gradient of layer l+1 reads pre[l], so either it runs first or abar_out[l] is a separate buffer.
 * WhT_pack: HOST array of the 3 transposed packs of layers 3, 4, 5. */
int stpde_jet_tail_bwd(const stpde_jet_cfg* cfg, int ntiles, int nf16, const float* abar5, const float* const* WhT_pack,
                       const float* const* pre, float* const* abar_out, const float* cw, float* act_param_bar,
                       void* stream);

/* bf16 mode: the same chain on the bf16 MFMA with every pre[] a packed stash and every abar_out[] a packed ADJOINT buffer
 * (packed == 3; abar_out[l] must not alias pre[l]);
 * WhT16_pack: bf16 packs of the transposed weights of layers 3 and 4 ([MT/2][KT][64] x 8 bf16; entry [2] is not read,
 * the product through the output layer stays fp32).  packed == 0 and WhT16_pack == NULL: stpde_jet_tail_bwd. */
int stpde_jet_tail_bwd_p(const stpde_jet_cfg* cfg, int ntiles, int nf16, const float* abar5, const float* const* WhT_pack,
                         const float* const* pre, float* const* abar_out, const float* cw, float* act_param_bar, int packed,
                         const void* const* WhT16_pack, void* stream);

/* Backward of the same layer w.r.t. its hidden input (the autograd backward of the addmm/activation graph,
 * i.e. what loss.backward() at experiments/rb2d/train.py:77 does through src/implicit_net.py:48-54):
 *   hbar = W_h^T * abar_out ; abar_in = act_jet_adjoint(hbar, in_pre)   written over in_pre (in place) when
 * first_hidden == 0, or into abar0[tile][1+S1][KT] otherwise (layer 0; its pre-activations are read from z0
 * [tile][KT][256], the value-stream stash written by stpde_jet_layer_fwd(first_hidden) -- round 2: regenerating them from X
 * cost 3.5 % of this kernel's MFMAs; X and W0s_pack are no longer read and may be NULL).  z0 may be the same buffer as abar0
 * when abar0 holds the value stream only (abar0_tan given, or S1 == 0): every lane reads its element before it writes it.
 * abar0_tan (first_hidden, S1 == 3; may be NULL): layer 0's tangent streams are the constant columns W0[:, d], so their
 * adjoints are needed only summed over rows; when given, abar0 receives the VALUE stream only ([tile][KT][256]) and
 * abar0_tan [tile][KT][3][16] the per-tile row sums of the three tangent-stream adjoints (a quarter of the traffic);
 * stpde_jet_tan0_reduce adds them into d W0[:, d].
 * act_param_bar (swish only, may be NULL): STPDE_PBAR_SLOTS (64) floats that ACCUMULATE partial sums of the adjoint of
 * the learnable beta (src/nonlinearities.py:5-12); the caller zero-fills them and adds the slots up. */
int stpde_jet_layer_bwd(const stpde_layer_desc* d, const float* abar_out, const float* WhT_pack, float* in_pre,
                        const float* X, const float* W0s_pack, const float* tanc0, float* abar0, const float* cw,
                        float* act_param_bar, const void* WhT_pack_bf16 /* as Wh_pack_bf16, of WhT_pack */,
                        float* abar0_tan, const float* z0, void* stream);

/* The same input gradient of a hidden layer (first_hidden == 0) written to a SEPARATE buffer abar_in instead of over the
 * stashed pre-activations in_pre (which stay intact): lets the whole input-gradient chain run before the weight gradients
 * -- "dgrad-first" order, so that the partial d latent of a rank is complete, and its all-reduce in flight, while the weight
 * gradients are still being computed (the overlap DistributedDataParallel gets from bucketed all-reduces inside
 * loss.backward(), experiments/rb2d/train_ddp.py:401-406). */
int stpde_jet_layer_bwd_to(const stpde_layer_desc* d, const float* abar_out, const float* WhT_pack, const float* in_pre,
                           float* abar_in, const float* cw, float* act_param_bar, const void* WhT_pack_bf16, void* stream);

/* Weight gradient of one layer: dW_aug[16*MT][16*(KT+3)] += sum_rows abar_out (x) [act_jet(in_pre) ; X_aug]
 * (columns: hidden inputs, then r(3), latent(c), bias, pad).  abar_out [tile][SP][MT] (SP = S, or 1+S1 for layer 0)
 * and in_pre [tile][S][KT] are the ordinary (column-major) layer buffers -- call it BEFORE stpde_jet_layer_bwd of the
 * same layer overwrites in_pre.  first_hidden: in_pre is the z0 stash [tile][KT][256] written by
 * stpde_jet_layer_fwd(first_hidden) (value stream of layer 0's pre-activations; the tangent streams are the constant
 * columns tanc0 [3][KT][256] = W0[:, d] in the column-major image, the second-order streams are zero) -- call it before
 * stpde_jet_layer_bwd(first_hidden) writes the layer-0 adjoint over the stash.  X = the (column-major) augmented input from
 * stpde_lig_gather.  fp32 atomics; caller zero-fills dW_aug.  d->mfma_bf16: layers with MT >= 8 contract with
 * bf16-rounded operands (two derivative streams per v_mfma_f32_16x16x32_bf16), fp32 accumulation.  d->packed (bf16 mode,
 * bits 1 = in_pre is a packed stash, 4 = abar_out is a packed adjoint buffer): the narrow layers (MT < 8) and the raw-input
 * layer (KT = 0, mfma_bf16 = 1, packed = 4) then contract over the rows with bf16 MFMAs as well; the raw-input columns of
 * the hidden layers stay fp32. */
int stpde_jet_wgrad(const stpde_layer_desc* d, int SP, const float* abar_out, const float* in_pre, const float* X,
                    const float* tanc0, float* dW_aug, const float* cw, void* stream);

/* bf16 mode, first hidden layer of the reference width (round 5): stpde_jet_wgrad(first_hidden) + stpde_jet_layer_bwd(first_hidden)
 * in ONE call -- one kernel reads the packed adjoint tile of fc1's rows once (LDS) for both products, W1h^T abar1 and
 * abar1^T act_jet(z0), and evaluates the activation jet of every z0 element once for the layer-0 adjoint and for the weight
 * gradient's operand blocks (csrc/jet_fc1_bwd.hip; the backward of src/implicit_net.py:48-54 through fc1 that loss.backward(),
 * experiments/rb2d/train.py:77, performs).  d: the description stpde_jet_layer_bwd takes for this layer (first_hidden = 1,
 * KT = 32, MT = 16, mfma_bf16 = 1, packed bits 2 and 4, S1 = 3, S2 = 0 or the combined stream).  abar1: packed ADJOINT buffer
 * of fc1's rows; WhT_pack_bf16: bf16 pack of W1h^T; z0 / tanc0 / cw / X as for the two calls it replaces; abar0 (packed
 * ADJOINT blocks of the value stream, must not alias z0), abar0_tan ([tile][32][3][16] row sums) as stpde_jet_layer_bwd writes
 * them; dW_aug: fc1's block of the flat gradient buffer (accumulated with fp32 atomics).
 * stpde_jet_fc1_bwd_supported(d) != 0 iff this call serves d (a pure function of d). */
int stpde_jet_fc1_bwd_supported(const stpde_layer_desc* d);
int stpde_jet_fc1_bwd(const stpde_layer_desc* d, const float* abar1, const void* WhT_pack_bf16, const float* z0,
                      const float* tanc0, const float* cw, const float* X, float* abar0, float* abar0_tan, float* dW_aug,
                      float* act_param_bar, void* stream);

/* ---- a4: corner-weighted reduction (src/local_implicit_grid.py:59) on all streams ------------------
 * jets[(s*n_out + ch)*ldp + p] (ldp >= P lets a chunk of points write into a larger [S][n_out][Ptotal] array)
 * from out_pre[tile][S][1][64][4] (fc5 output) and coef; and its adjoint. */
/* S_mlp = number of streams held by the layer buffers: 1 + S1 + S2, or 1 + S1 for piecewise-linear activations
 * (relu / leaky-relu: sigma'' = 0, so every second-order MLP stream is identically zero and is not carried; the
 * second derivatives of y then come from the weight-derivative cross terms alone). */
int stpde_lig_reduce_fwd(const stpde_jet_cfg* cfg, int S_mlp, int P, int n_out, const float* out_pre,
                         const float* coef, float* jets, long ldp, void* stream);
int stpde_lig_reduce_bwd(const stpde_jet_cfg* cfg, int S_mlp, int P, int n_out, const float* jets_bar, long ldp,
                         const float* coef, float* abar_out, void* stream);

/* dW_aug[16*MT][ldw] (layer 0: ldw = 16 * XT) column d  +=  sum over tiles of abar0_tan[tile][mt][d][:]. */
int stpde_jet_tan0_reduce(int ntiles, int MT, const float* abar0_tan, float* dW_aug, int ldw, int det /* as stpde_layer_desc.det */,
                          void* stream);

/* ---- backward of the gather: d latent (index_put accumulate, backward of :65-66) --------------------
 * xbar = sum_l W_s,l^T * abar_l(value stream); latent channels are scatter-added into dlatent
 * [B][n0][n1][n2][C] at the 8 corner nodes of each point.  nlayers <= 8. */
typedef struct {
  int ntiles, nlayers, C, n1, n2;
  int MT[8];
  int SP[8];
  int packed[8];   /* != 0: abar[l] is a packed ADJOINT buffer (stpde_layer_desc.packed) of S[l] streams */
  int S[8];        /* streams of a packed buffer (only read where packed[l] != 0) */
} stpde_xbar_desc;
/* abar / WsL_pack: HOST arrays of nlayers device pointers.  WsL_pack[l] = [MT_l][XL][64 lanes][4]: A operand of W_s,l^T
 * restricted to the latent channels, XL = ceil(C / 16) tiles of 16 channels (lane (g, j), register r holds
 * W_s,l[16 mt + 4 g + r][dim + 16 xl + j], zero beyond channel C - 1); the coordinate / bias columns get no adjoint. */
int stpde_lig_xbar_scatter(const stpde_xbar_desc* d, const float* const* abar, const float* const* WsL_pack,
                           const int* cell, float* dlatent, void* stream);
/* Deterministic variant (default of the Python host): the same xbar, but the latent channels of every corner row are
 * written to xrows [16 * ntiles][CP] (CP = C rounded up to a multiple of 4; row = 8 * point + corner) instead of being
 * scatter-added with fp32 atomics; stpde_lig_dlatent_reduce then sums them per NODE in a fixed order (corner 0..7, points
 * of the owning cell in ascending index), like the reference's deterministic CPU index_put_(accumulate=True) (:65-66).
 * perm [P] = point indices in stable cell order, start [n_nodes + 1] = first position of each cell id (cell id = linear
 * index of the cell's corner-0 node incl. batch, as written by stpde_lig_gather); dlatent is accumulated into (+=). */
int stpde_lig_xbar_rows(const stpde_xbar_desc* d, const float* const* abar, const float* const* WsL_pack,
                        float* xrows, void* stream);
int stpde_lig_dlatent_reduce(int B, int n0, int n1, int n2, int C, const float* xrows, const int* perm,
                             const int* start, float* dlatent, void* stream);

/* ---- a4/a5/a7 in ONE call per direction (lig_imnet_jet_fwd / lig_imnet_jet_bwd) ---------------------
 * The whole query path of src/local_implicit_grid.py:47-59 (gather -> IM-NET on all derivative streams -> corner-weighted
 * sum) for one chunk of query points, and its backward (what loss.backward(), experiments/rb2d/train.py:77, does through
 * it: weight gradients of fc0..fc5, latent-grid gradient).  Launch sequencing only: the kernels are the ones behind the
 * per-layer entry points above, issued in the same order; the per-layer entry points stay exported (profiling, tests).
 * All buffers are the caller's (stpde_lig_workspace); sizes in floats for a chunk of P points, nt = P / 2 row tiles,
 * S = 1 + S1 + S2 streams of cfg_mlp, block = 256 floats:
 *   X [nt][3][block] (XR: unused since round 5, may be NULL)   coef [P][16]   cw [P][8] (combined stream only, else NULL)   cell [P] int
 *   pre[0] = z0 stash [nt][MT_0][block] (training only)      pre[l], l >= 1: [nt][S][MT_l][block]
 *   backward scratch: abar2x / abar3x = sizes of pre[2] / pre[3] (fresh adjoint buffers of the fused tail), tan0
 *   [nt][MT_0][48], abar0 [nt][1 + S1][MT_0][block] (only when tan0 is NULL and S1 == 3), xrows [16 nt][CP], perm [P],
 *   start [n_nodes + 1], sort_tmp (stpde_lig_sort_tmp_bytes). */
typedef struct {
  int nlayers;                 /* 6: fc0 .. fc5 of src/implicit_net.py:31-36 */
  int cin, cout, nf16;         /* latent channels, outputs, nf / 16 */
  int KT[8], MT[8];            /* hidden k-tiles / output tiles of each layer */
  const float* Wh[8];          /* A-operand packs (see stpde_jet_layer_fwd); Wh[0] / WhT[0] unused */
  const float* WhT[8];
  const float* Ws[8];
  const float* WsL[8];
  const float* tanc[8];
  const void* Wh16[8];         /* bf16 packs of the wide layers or NULL */
  const void* WhT16[8];
  int mfma_bf16;               /* 0 / 1 / 3 as stpde_layer_desc.mfma_bf16 */
  int packed_mask;             /* bit l >= 1: pre[l] is a packed stash and the adjoint of layer l's rows a packed adjoint buffer
                                  (stpde_layer_desc.packed); bit 0: the value-stream adjoint of layer 0 (workspace.abar0x) is
                                  a packed adjoint buffer.  Only with mfma_bf16 == 1; the library serves 0, 30 and 31. */
  long dw_off[8];              /* offset (floats) of layer l's dW_aug block [16 MT][16 (KT + 3)] in dW_flat */
} stpde_imnet_plan;
typedef struct {
  float* X;
  float* XR;
  float* coef;
  float* cw;
  int* cell;
  float* pre[8];
  float* abar2x;
  float* abar3x;
  float* tan0;
  float* abar0;
  float* abar1x;   /* dgrad-first order / packed buffers: fresh adjoint buffer of fc1's output rows */
  float* abar0x;   /* dgrad-first order / packed_mask bit 0: fresh layer-0 adjoint [nt][MT_0][block] (the z0 stash stays intact) */
  float* xrows;
  int* perm;
  int* start;
  void* sort_tmp;
  unsigned long sort_tmp_bytes;
  float* abar4x;   /* packed buffers: adjoint buffer of fc4's output rows (abar2x / abar3x / abar4x / abar1x then hold packed
                      ADJOINT buffers: nt * S * MT_l * 512 bytes) */
} stpde_lig_workspace;
#define STPDE_F_STASH 1          /* forward: keep what the backward needs (z0) */
#define STPDE_F_VALUE_TILES 2    /* forward-only value queries: four row tiles per pass over the weights */
#define STPDE_F_FUSED_TAIL 4     /* fc3 -> fc4 -> fc5 in one kernel each way (nf = 16 / 32) */
#define STPDE_F_TAN0_ROWSUM 8    /* backward: layer-0 tangent adjoints as per-tile row sums (needs workspace.tan0) */
#define STPDE_F_DETERMINISTIC 16 /* backward: d latent by per-node sums in a fixed order instead of fp32 atomics */
#define STPDE_F_WGRAD 32         /* backward: compute the weight gradients */
#define STPDE_F_WGRAD_FP32 64    /* fp32x3 mode: keep the wide layers' weight gradients on the exact-fp32 MFMA */
/* Backward in two calls, "dgrad-first" (needs workspace.abar1x / abar0x, the fused tail and the tangent row sums):
 * PHASE_A = corner-reduction adjoint, fc5 weight gradient, the whole input-gradient chain into fresh buffers, d latent;
 * PHASE_B = the remaining weight gradients (fc4 .. fc0).  Between the two the caller may start the all-reduce of d latent. */
#define STPDE_F_PHASE_A 128
#define STPDE_F_PHASE_B 256
/* bf16 mode: do NOT use the fused backward of the first hidden layer (stpde_jet_fc1_bwd); the input gradient and the weight
 * gradient of that layer then run as two kernels.  A/B switch of the tests.  Both phases of a dgrad-first backward must carry
 * the same value of this bit and of STPDE_F_WGRAD: phase B skips fc1's weight gradient exactly when phase A's fused kernel
 * produced it. */
#define STPDE_F_NO_FC1_FUSED 512
/* backward: dW_flat addresses long accumulators (stpde_layer_desc.det): [sum of the layers' dW_aug elements][6] int64, layer l at
 * element offset dw_off[l]; bit-reproducible IM-NET weight gradients */
#define STPDE_F_DET 1024
/* cfg_mlp = streams the layer kernels carry, cfg_out = streams of `jets` (they differ for piecewise-linear activations,
 * see stpde_lig_reduce_fwd); jets points at the first point of the chunk inside [S_out][n_out][ldp]. */
int stpde_lig_imnet_jet_fwd(const stpde_imnet_plan* plan, const stpde_jet_cfg* cfg_mlp, const stpde_jet_cfg* cfg_out,
                            const stpde_gather_desc* gd, const float* pts, const float* latent,
                            const stpde_lig_workspace* ws, float* jets, long ldp, int flags, void* stream);
/* cfg_val = cfg_mlp with S1 = S2 = 0 (layer-0 weight gradient of the value stream).  dW_flat (zero-filled by the caller,
 * accumulated into; may be NULL), dlatent (accumulated into; may be NULL), act_param_bar as stpde_jet_layer_bwd. */
int stpde_lig_imnet_jet_bwd(const stpde_imnet_plan* plan, const stpde_jet_cfg* cfg_mlp, const stpde_jet_cfg* cfg_out,
                            const stpde_jet_cfg* cfg_val, const stpde_gather_desc* gd, const stpde_lig_workspace* ws,
                            const float* jets_bar, long ldp, float* dW_flat, float* dlatent, float* act_param_bar,
                            int flags, void* stream);
/* Cell sort of the deterministic d latent (replaces the torch.sort + index_add_ + cumsum of the round-2 host): perm [P] =
 * point indices in stable cell order, start [n_nodes + 1] = number of points in cells < c; tmp = device scratch of at
 * least stpde_lig_sort_tmp_bytes(P, n_nodes) bytes. */
unsigned long stpde_lig_sort_tmp_bytes(int P, long n_nodes);
int stpde_lig_cell_sort(int P, long n_nodes, const int* cell, int* perm, int* start, void* tmp, unsigned long tmp_bytes,
                        void* stream);

/* ---- a2/a3 for any dim 1..4: plain multilinear interpolation ---------------------------------------
 * Replaces regular_nd_grid_interpolation (src/regular_nd_grid_interpolation.py:81-104) and the three outputs
 * of ..._coefficients (:14-78).  grid [B][n_0..n_{dim-1}][C], pts [B*N][dim].  Any output pointer may be NULL. */
typedef struct {
  int P, N, B, dim, C;
  int n[4];
  float lo_c[4], hi_c[4], cube[4];
} stpde_interp_desc;
int stpde_interp_fwd(const stpde_interp_desc* d, const float* grid, const float* pts, float* out /*[P][C]*/,
                     float* corner_values /*[P][2^dim][C]*/, float* weights /*[P][2^dim]*/,
                     float* x_relative /*[P][2^dim][dim]*/, void* stream);
/* dgrid[node] += w_j * out_bar[p] (out_bar [P][C]) and/or corner_bar[p][j] ([P][2^dim][C]); either may be NULL. */
int stpde_interp_bwd_grid(const stpde_interp_desc* d, const float* pts, const float* out_bar,
                          const float* corner_bar, float* dgrid, void* stream);

/* ---- a10: 3-D convolution of the U-Net encoder (src/unet3d.py:39-56: nn.Conv3d 1x1x1 and 3x3x3/pad 1, stride 1)
 * Activations are channels-last: x [B][T][Z][X][Ci], y [B][T][Z][X][Co], Ci and Co multiples of 16.
 * Implicit GEMM on v_mfma_f32_16x16x4_f32: y^T[Co x 16 voxels] = sum_tap W_tap[Co x Ci] * x_tap^T[Ci x 16 voxels].
 * w_pack [taps][Ci/16][Co/16][64][4] (A-operand image of W[:, :, tap]); bias [Co] or NULL.
 * The input-gradient is the same kernel called with the transposed, tap-flipped pack (no bias). */
typedef struct {
  int B, T, Z, X, Ci, Co, ksize; /* ksize 1 or 3 */
  /* Deterministic mode (round 6; 0 = off).  != 0: every sum the kernels of this call accumulate with atomics -- dW / dbias of
   * the weight-gradient calls, out_sums / m_bsum of stpde_conv3d_fused -- goes to ORDER-INDEPENDENT long accumulators instead
   * of fp32 / fp64 atomics: the destination pointer then addresses STPDE_DET_K (6) zero-filled signed 64-bit integers per
   * element (48 bytes instead of 4; element e at windows [6 e, 6 e + 6)), the result is bit-identical from run to run, and
   * stpde_det_finalize turns an accumulator array into fp32 (statistics / backward sums are read by the BatchNorm kernels
   * directly: a single replica [2][C][6] in the same scratch).  The forward does not split its taps over workgroups. */
  int det;
} stpde_conv3d_desc;
/* acc [n][STPDE_DET_K] long accumulators (a deterministic-mode destination, see stpde_conv3d_desc.det) -> out [n] fp32. */
#define STPDE_DET_K 6
int stpde_det_finalize(const void* acc, long n, float* out, void* stream);
int stpde_conv3d_fwd(const stpde_conv3d_desc* d, const float* x, const float* w_pack, const float* bias, float* y,
                     void* stream);
/* dW[tap][Co][Ci] += sum_voxels ybar[v][co] * x[v + offset(tap)][ci]  (fp32 atomics; caller zero-fills dW). */
int stpde_conv3d_wgrad(const stpde_conv3d_desc* d, const float* x, const float* ybar, float* dW, void* stream);
/* The same + the bias gradient: dbias[Co] += sum over the voxels of ybar (fp32 atomics; caller zero-fills), taken from the
 * ybar fragments the kernel loads anyway -- replaces a separate reduction pass over ybar (the reference computes it inside
 * its convolution backward, torch.nn.Conv3d). */
int stpde_conv3d_wgrad_bias(const stpde_conv3d_desc* d, const float* x, const float* ybar, float* dW, float* dbias,
                            void* stream);

/* ---- a10 (round 4): the convolutions of one ResBlock3D with its BatchNorm work folded in (src/unet3d.py:39-56:
 * conv1-bn1-relu-conv2-bn2-relu-conv3-bn3 + shortcut, relu).  One call = one convolution of the chain plus what the
 * BatchNorm next to it would otherwise do in passes of its own over the same tensor:
 *   out_sums  the per-channel sums of y and y^2 the following BatchNorm needs, from the accumulator tiles (doubles,
 *             [STPDE_BN_REP][2][Co], caller zero-fills; every wave sums around its own first voxel and converts in fp64)
 *   in_sums   ksize 1: x is the RAW output of the previous convolution; its training-mode BatchNorm + ReLU are applied to
 *             the operand as it is loaded: x' = max(0, (x - mean) * (rstd * gamma) + beta), mean / rstd from in_sums over
 *             the N = B*T*Z*X voxels; the call writes in_stat = [mean; rstd] and updates the running statistics
 *   y2        ksize 1: a second convolution of the same input (conv1 and the shortcut of a block: x is read once)
 *   x2        ksize 1: a second input, y = W x + W2 x2 (the input gradient of a block = conv1's + the shortcut's)
 *   m         input-gradient convolutions: y is the gradient of act = relu(bn(m)); the call stores dz = y * [act > 0] and adds
 *             sum(dz), sum(dz * xhat) to m_bsum ([STPDE_BN_REP][2][Co] floats, zero-filled) -- the reduction pass of
 *             stpde_bn_bwd.  Volumes small enough for the tap-split kernel cannot do this in their epilogue:
 *             *epilogue_done = 0 then (y holds the unmasked gradient) and the caller runs the full stpde_bn_bwd.
 * Unused features: NULL pointers / zero channel counts. */
typedef struct {
  stpde_conv3d_desc d;
  const float* x;
  const float* w_pack;
  const float* bias;
  float* y;
  const float* x2;       /* [voxels][Ci2] */
  const float* w2_pack;  /* [Ci2/16][Co/16][64][4] */
  float* y2;             /* [voxels][Co2] */
  const float* wo2_pack; /* [Ci/16][Co2/16][64][4] */
  const float* bias2;
  const double* in_sums;
  const float* in_gamma; /* NULL = 1 */
  const float* in_beta;  /* NULL = 0 */
  float* in_running_mean; /* NULL: no running statistics */
  float* in_running_var;
  float* in_stat;        /* [2][Ci] */
  double* out_sums;
  const float* m;        /* [voxels][Co] */
  const float* m_stat;   /* [2][Co]: mean, rstd of m */
  const float* m_gamma;
  const float* m_beta;
  float* m_bsum;
  int Ci2, Co2;
  float in_eps, in_momentum;
} stpde_conv3d_fused_args;
int stpde_conv3d_fused(const stpde_conv3d_fused_args* a, int* epilogue_done, void* stream);
/* Weight (+ bias) gradient of a 1x1x1 convolution whose input was x' = max(0, bn(x)) applied on load (in_stat = [mean; rstd]
 * of x as written by stpde_conv3d_fused, in_gamma / in_beta nullable): the same transform on the operand here. */
int stpde_conv3d_wgrad_onload(const stpde_conv3d_desc* d, const float* x, const float* ybar, float* dW, float* dbias,
                              const float* in_stat, const float* in_gamma, const float* in_beta, void* stream);

/* ---- a10: BatchNorm3d (+ residual add) (+ ReLU) of the ResBlock3D chain (src/unet3d.py:39-56) -------------
 * Channels-last x [N][C], N = B*T*Z*X, C a power of two in [16, 512].
 * forward:  y = act(bn(x) [+ residual]);  training != 0: batch statistics (biased variance for the normalisation,
 *           running_mean / running_var updated in place with torch's momentum rule and the unbiased variance);
 *           training == 0: running statistics.  sums: [STPDE_BN_REP][3][C] scratch (training), stat: [2][C] receives mean, rstd.
 * backward: dz = dy * [y > 0] (relu) ; dresidual = dz ; dx, dgamma, dbeta as torch's batch_norm backward
 *           (training: with the batch-statistics terms, evaluation: without).  bsum: [STPDE_BN_REP][2][C] scratch.
 * The reduction kernels spread their per-block atomics over STPDE_BN_REP replicas of the sums (one copy made ~1000 blocks
 * queue on 2 C addresses); the elementwise kernels add the replicas up in a fixed order.
 * Any of residual, gamma, beta, dx, dresidual, dgamma, dbeta may be NULL. */
#define STPDE_BN_REP 16
typedef struct {
  long N;
  int C, training, relu;
  float eps, momentum;
  int scratch_zeroed;   /* != 0: the caller hands over zero-filled sums / bsum scratch (e.g. slices of one buffer cleared once
                           per step), so no memset is queued in front of the reduction kernels */
  int stats_mode;       /* forward, training: 0 = sums is the float scratch above; 1 = sums is double [STPDE_BN_REP][2][C]
                           (sum of x, sum of x^2; the format stpde_conv3d_fused writes) and this call fills it; 2 = the same
                           format, already complete (no statistics pass) */
  int reduce_done;      /* backward: bsum is already complete and dy already carries the ReLU mask (stpde_conv3d_fused, m):
                           only the elementwise pass runs; pass relu = 0 */
  int det;              /* deterministic mode (stpde_conv3d_desc.det): sums / bsum hold long accumulators -- forward statistics
                           [2][C][6] int64 (sum x, sum x^2; stats_mode 1 / 2 only), backward sums [2][C][6] -- in the same
                           zero-filled scratch */
} stpde_bn_desc;
int stpde_bn_fwd(const stpde_bn_desc* d, const float* x, const float* residual, const float* gamma, const float* beta,
                 float* running_mean, float* running_var, float* sums, float* stat, float* y, void* stream);
int stpde_bn_bwd(const stpde_bn_desc* d, const float* x, const float* y, const float* dy, const float* gamma,
                 const float* stat, float* bsum, float* dx, float* dresidual, float* dgamma, float* dbeta, void* stream);

/* ---- a10: max pooling / nearest up-sampling between the U-Net levels (src/unet3d.py:163-176, 216-238) ----------
 * Channels-last [B][T][Z][X][C], C % 4 == 0.  The descriptor holds the dimensions of the SMALL tensor (pooled output /
 * up-sampling input) and the integer factors (1..4) per dimension; the big tensor is [B][T*ft][Z*fz][X*fx][C].
 * mode 0: max-pool forward (in = big, out = small)       mode 1: max-pool backward (in = d small, aux = forward
 * input, out = d big; ties go to the first maximum of the window like torch)
 * mode 2: nearest up-sampling forward (in = small, out = big)   mode 3: its backward (in = d big, out = d small). */
typedef struct {
  int B, T, Z, X, C, ft, fz, fx;
} stpde_resample_desc;
int stpde_resample3d(const stpde_resample_desc* d, int mode, const float* in, const float* aux, float* out,
                     void* stream);

/* ---- a7/a8/a9: PDE residuals from the jet streams ---------------------------------------------------
 * Replaces the elementwise algebra that src/pde.py:115-143 evaluates through the lambdified equation strings once the
 * derivatives are known (e.g. the four Rayleigh-Benard residuals of experiments/rb2d/physics.py:26-57), and its
 * autograd backward.  The host compiles every equation into ONE straight-line program in SSA form: instruction i
 * produces value i from earlier values a, b (indices < i):
 *   JET   value = jets[a * ld_stream + b * ld_channel + p]   (a = stream, b = output channel)
 *   X     value = x[p][a]            CONST value = c
 *   ADD SUB MUL DIV (a, b)   NEG (a)   POWI value = v[a]^b (integer b)   SIN COS EXP LOG SQRT TANH ABS (a)
 *   OUT   residual b of the point = v[a]
 * res [n_eq][P]; the adjoint adds d loss / d jets into jets_bar (same layout as jets, zero-filled by the caller).
 * The coordinates x are not differentiated here (the jets already are the coordinate derivatives). */
#define STPDE_RES_MAX_INS 192
enum {
  STPDE_RES_JET = 0, STPDE_RES_X, STPDE_RES_CONST, STPDE_RES_ADD, STPDE_RES_SUB, STPDE_RES_MUL, STPDE_RES_DIV,
  STPDE_RES_NEG, STPDE_RES_POWI, STPDE_RES_SIN, STPDE_RES_COS, STPDE_RES_EXP, STPDE_RES_LOG, STPDE_RES_SQRT,
  STPDE_RES_TANH, STPDE_RES_ABS, STPDE_RES_OUT
};
typedef struct {
  int op, a, b;
  float c;
} stpde_res_ins;
int stpde_residual_fwd(const stpde_res_ins* prog_dev, int nins, int n_eq, int n_out, int P, const float* jets,
                       long ld_stream, long ld_channel, const float* x /* [P][3] or NULL */, float* res, void* stream);
int stpde_residual_bwd(const stpde_res_ins* prog_dev, int nins, int n_eq, int n_out, int P, const float* jets,
                       long ld_stream, long ld_channel, const float* x, const float* res_bar, float* jets_bar,
                       void* stream);

/* ---- a11: loss reductions of the train step (experiments/rb2d/train.py:69-76) -------------------------------
 * out_sum += sum_i f(a[i] - b[i]) with f = |d| (L1), d^2 (L2) or smooth-l1 with beta 1 (b == NULL: against 0);
 * a SUM, not a mean: the caller divides by the GLOBAL element count (point-sharded multi-GPU step).  The caller
 * zero-fills out_sum.  stpde_loss_grad writes grad_a[i] = *grad_sum_dev * f'(a[i] - b[i]) (sign(0) = 0 like torch). */
#define STPDE_LOSS_L1 0
#define STPDE_LOSS_L2 1
#define STPDE_LOSS_HUBER 2
/* kind | STPDE_LOSS_DET (stpde_loss_sum only): out_sum addresses ONE long accumulator (six zero-filled 64-bit integers, see
 * stpde_conv3d_desc.det) instead of a float -- the block sums are added with integer atomics, bit-identical from run to run;
 * stpde_det_finalize(out_sum, 1, ...) gives the float. */
#define STPDE_LOSS_DET 16
int stpde_loss_sum(int kind, long n, const float* a, const float* b, float* out_sum, void* stream);
int stpde_loss_grad(int kind, long n, const float* a, const float* b, const float* grad_sum_dev, float* grad_a,
                    void* stream);

/* ---- N1: train-step tail -- gradient value clipping + Adam in one pass ---------------------------------
 * Replaces torch.nn.utils.clip_grad_value_ + optim.Adam.step of experiments/rb2d/train.py:79-83 for one
 * parameter tensor (fp32, 16-byte aligned).  g = clamp(grad, +-clip) (clip <= 0: off); m, v, p are updated in
 * place with torch.optim.Adam's formulas: step_size = lr / (1 - beta1^t), bias2_sqrt = sqrt(1 - beta2^t). */
typedef struct {
  long n;
  float clip, beta1, beta2, eps, weight_decay, step_size, bias2_sqrt;
} stpde_adam_desc;
int stpde_clip_adam(const stpde_adam_desc* d, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                    void* stream);
/* The same update for ALL parameter tensors in one launch (the reference's optimizer walks ~300 tensors).  Both
 * tables live in device memory and are built by the caller: one stpde_adam_tensor per parameter (16-byte aligned
 * pointers; step_size / bias2_sqrt per tensor, d->step_size / d->bias2_sqrt / d->n are ignored) and one
 * stpde_adam_chunk per block of work: elements [offset, offset + chunk_elems) of tensor `tensor`, offset % 4 == 0. */
typedef struct {
  float* p;
  const float* g;
  float* m;
  float* v;
  long n;
  float step_size, bias2_sqrt;
} stpde_adam_tensor;
typedef struct {
  int tensor, pad;
  long offset;
} stpde_adam_chunk;
int stpde_clip_adam_multi(const stpde_adam_desc* d, const stpde_adam_tensor* tensors_dev,
                          const stpde_adam_chunk* chunks_dev, int nchunks, int chunk_elems, void* stream);

#ifdef __cplusplus
}
#endif
#endif
