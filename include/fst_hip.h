/* libfst_hip.so — C ABI of the MI355X (gfx950) train-step kernels for the feature-level
 * style-transfer TSC hot path.
 *
 * The reference (BaeHann/feature_level_style_transfer_for_TSC) has no FFI: its hot path is stock
 * PyTorch ops called from nn.Modules.  Each entry point below names the reference call site it
 * replaces (file:line, relative to the reference root).  The Python host side
 * (feature_level_style_transfer_for_tsc_amd/_lib.py) binds exactly these symbols with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; the caller owns all memory;
 *     the library never allocates, frees or retains device memory;
 *   - activations are fp32, NCL, channel stride = L, explicit batch stride (so channel slices of a
 *     wider tensor can be passed without a copy);
 *   - launches are asynchronous on `stream` (a hipStream_t passed as void*); no implicit syncs;
 *   - return 0 on success, <0 for an argument error detected on the host before any launch,
 *     >0 = hipError_t; fst_last_error() returns a thread-local message.  No exceptions cross the ABI.
 */
#ifndef FST_HIP_H
#define FST_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FST_ABI_VERSION 12

int fst_version(void);
const char* fst_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Conv plan.  A "plan" is a small int32 device table describing how a (multi-tap, optionally
 * dilated, optionally two-input) 1-D convolution is cut into K-chunks and 32-row M-blocks for the
 * f32 MFMA implicit GEMM.  It is built on the host (plan.py) once per layer shape.
 *
 *   plan[0..15]  header: n_chunks, n_mgroups, MB (32-row blocks per M-group), ntaps, dil, pad_left,
 *                        chunk_cap (max padded channels per chunk, even), total_records,
 *                        n_items, items_per_wg, n_stages, 5 reserved (0)
 *   plan[16 + 4*q ...]                 chunk q:        src (0|1), c_begin, c_count, 0
 *   plan[16 + 4*n_chunks + 4*(g*n_chunks+q) ...]  (g,q): tap_lo, tap_hi, record_offset, stage_offset
 *   plan[16 + 4*n_chunks*(1+n_mgroups) + 4*i ...] item i (weight-gradient plans only):
 *                                                       g, q, row_block, 0      (q < 0: padding item)
 *
 * A "record" is MB*64 floats: for one (tap, channel pair) the MB A-operand registers of one
 * v_mfma_f32_32x32x2_f32 k-step (lane l ↔ row l&31 of the block, channel parity l>>5).
 * Every entry point takes the table twice: `plan_dev` (read by the kernels through scalar loads) and
 * `plan_host` + `plan_len` (the same ints in host memory, used to validate shapes and size the launch).
 * ------------------------------------------------------------------------------------------- */
#define FST_PLAN_HEADER 16

/* Weight source description for pack/unpack: element (m, c, tap) of input `s` lives at
 * w[s][off0 + m*sm + c*sc + tap*st].  Forward convs use (sm, sc, st) = (C*ntaps, ntaps, 1);
 * the data-gradient conv uses the transposed, tap-flipped view of the same tensor. */
typedef struct {
  const float* w;   /* device */
  int64_t off0, sm, sc, st;
} fst_wsrc;

/* Pack PyTorch-layout weights into the plan's record layout.  Replaces the implicit weight use in
 * F.conv1d at OS_CNN/OS_CNN.py:71, Simplified_NF_WaveGlow.py:36,41,103,107,112,116,123. */
int fst_pack_weights(const int32_t* plan_dev, const int32_t* plan_host, int plan_len,
                     const fst_wsrc* src0_host, const fst_wsrc* src1_host,
                     int M, int g_begin, int g_end /* M-group range, -1 = all */, int row_base /* view row = m - row_base */,
                     float* a_packed, void* stream);

/* Same as fst_pack_weights, into the split-bf16 image the FST_GEMM_BF16X3 path of fst_conv_gemm reads
 * (plan.n_stages * MB * 2048 bytes + a 16-byte zero block): every weight w is stored as two round-to-nearest bf16 parts,
 * hi = bf16(w) and lo = bf16(w - hi); stage s (= one live (M-group, chunk, tap) triple, 16 channels deep),
 * 32-row block mb: 64 lanes x 8 hi parts, then 64 lanes x 8 lo parts, lane l = row l&31, channels 8*(l>>5)+0..7
 * — the A-operand fragments of v_mfma_f32_32x32x16_bf16.  Only for plans made of single-tap stages of at
 * most 16 channels (the pipelined kernel's plans). */
int fst_pack_weights_bf16x3(const int32_t* plan_dev, const int32_t* plan_host, int plan_len,
                            const fst_wsrc* src0_host, const fst_wsrc* src1_host,
                            int M, int g_begin, int g_end, int row_base, void* a_image, void* stream);

/* Inverse of fst_pack_weights for weight gradients: scatter a packed gradient back to PyTorch
 * layout (dw0/dw1 must be zero-filled by the caller when the plan does not cover every tap). */
int fst_unpack_weights(const int32_t* plan_dev, const int32_t* plan_host, int plan_len,
                       const float* a_packed, int M,
                       float* dw0, int64_t off0_0, int64_t sm0, int64_t sc0, int64_t st0,
                       float* dw1, int64_t off0_1, int64_t sm1, int64_t sc1, int64_t st1,
                       int n_slabs /* a_packed = n_slabs partial sums of plan.packed_floats each, added here */,
                       void* stream);

/* Omni-scale re-masking W ← W ⊙ mask, mask given as per-output-channel live tap range.
 * Replaces OS_CNN/OS_CNN.py:68. */
int fst_mask_taps(float* w, const int32_t* live_lo, const int32_t* live_hi, int M, int C, int K, void* stream);

/* y[b,m,t] = bias[m] + Σ_k A[m,k]·xcol[b,k,t]  — the forward engine (f32 MFMA, LDS-staged input
 * window).  Rows m < msplit go to y (+res if given); rows ≥ m2_start go to y2[m − m2_start]
 * (accumulated if FST_EPI_ACC2).  Replaces F.conv1d/nn.Conv1d at OS_CNN/OS_CNN.py:71,164 and
 * Simplified_NF_WaveGlow.py:36,41,103,107,112,116,123; with transposed packing it is also the
 * data-gradient conv of every one of those.
 *   flags: FST_EPI_RELU, FST_EPI_ACC2, FST_EPI_ATOMIC (ksplit>1; y must be pre-initialised);
 *   FST_GEMM_BF16X3: a_packed is the image written by fst_pack_weights_bf16x3 and every product is formed as
 *   hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (both operands split into two
 *   round-to-nearest bf16 parts; relative error of a product <= 3*2^-18, measured 5e-6 of the output scale)
 *   — about 5x the f32 MFMA rate.  Needs a pipelined plan, L % 4 == 0 and 16-byte aligned activations. */
#define FST_EPI_RELU    1
#define FST_EPI_ACC2    2
#define FST_EPI_ATOMIC  4
#define FST_EPI_ACC1    8
#define FST_GEMM_BF16X3 16
#define FST_WGRAD_SLABS 32   /* fst_conv_wgrad: da_packed holds one partial-sum slab per K slice (no atomics, no zero fill) */
int fst_conv_gemm(const float* x0, int64_t x0_bs, const float* x1, int64_t x1_bs,
                  const float* a_packed, const int32_t* plan_dev, const int32_t* plan_host, int plan_len,
                  const float* bias,
                  float* y, int64_t y_bs, const float* res, int64_t res_bs,
                  float* y2, int64_t y2_bs, int msplit, int m2_start /* rows [msplit, m2_start) are padding */,
                  int B, int L, int M, int nb_cfg, int ksplit, int flags, void* stream);

/* dA_packed[(g,q) records] += Σ_{b,t} dy[b,m,t]·xcol[b,k,t]  — weight-gradient engine (f32 MFMA,
 * split over (b,t) with fp32 atomics; the caller zero-fills da_packed).  Replaces the weight
 * gradient autograd derives for every conv listed above (Q1: with a dense plan it also yields the
 * masked-tap gradients the reference's GradNorm consumes, train_and_test.py:685-690).
 * FST_GEMM_BF16X3: products on v_mfma_f32_32x32x16_bf16 with both operands split into two bf16 parts
 * (hi*hi + hi*lo + lo*hi, fp32 accumulate), same layout of da_packed.
 * FST_WGRAD_SLABS: instead of fp32 atomics into one zero-filled buffer (1.3 TB/s chip-wide: 50 MB of them per in_layer
 * launch), K slice s stores its partial sums plainly into da_packed + s·packed_floats (the buffer holds
 * min(ksplit, B·⌈L/32⌉) slabs, no zero fill); fst_unpack_weights(n_slabs) adds them.
 * x0_mul_off != 0 (single-input 1x1 plans only): the x operand is the elementwise product x0[i]·x0[i + x0_mul_off] formed
 * while staging — the res_skip weight gradient reads acts = t·s from the gate halves the fused forward saved
 * (x0 = the t rows of ts, offset n·L to the s rows), so acts itself is never written to HBM. */
int fst_conv_wgrad(const float* x0, int64_t x0_bs, const float* x1, int64_t x1_bs,
                   const float* dy, int64_t dy_bs, const float* dy2, int64_t dy2_bs, int msplit,
                   float* da_packed, const int32_t* plan_dev, const int32_t* plan_host, int plan_len,
                   int B, int L, int M, int ksplit, int flags /* 0 or FST_GEMM_BF16X3 */, int64_t x0_mul_off, void* stream);

/* out[m] = Σ_{b,t} x[b,m,t]   (bias gradients): one workgroup per row, written (no zero fill, no atomics, fixed order). */
int fst_row_sum(const float* x, int64_t x_bs, int B, int C, int L, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * BatchNorm1d over (B, L), train and eval — replaces nn.BatchNorm1d at OS_CNN/OS_CNN.py:72,165
 * and the ReLU / residual add at :74, :179.
 * stats layout (float[4*C]): mean | invstd | scale | shift   (scale = γ·invstd, shift = β − mean·scale)
 * ------------------------------------------------------------------------------------------- */
/* `numel` (here and below): the element count of the contiguous [B, C, L] tensors the launch walks, taken by the caller
 * from the tensors it holds — NOT recomputed from B, C, L.  The launcher refuses B*C*L != numel before launching, so a
 * batch argument that does not describe the buffers (the cause of round 1's GPU memory fault: the all-rank batch
 * passed as the launch batch of fst_bn_bwd_apply) is an error return, never an out-of-bounds walk. */
/* Batch moments as partials: part[c][slot] = (count, k, Σ(x − k), Σ(x − k)²) of the samples workgroup `slot` of channel c
 * walked, shifted by k = y[0][c][0] — FST_BN_SLOTS slots per channel, every one written (no zero fill by the caller, no
 * atomics).  Never Σx² − (Σx)²/N of the raw values: a channel of the univariate extractor's 1x1 shortcut has |mean|/std in
 * the hundreds, where that form has no correct digit of the variance in fp32. */
#define FST_BN_SLOTS 16
int fst_bn_stats(const float* y, int B, int C, int L, float* part /* [C][FST_BN_SLOTS][4], written */, int64_t numel, void* stream);
/* Turns every slot into (count, mean, M2) and merges the n_slots of a channel in slot order with Chan's formula (double
 * precision) into the batch mean / biased variance, updates
 * the running statistics (unbiased variance) and writes stats.  n_slots = FST_BN_SLOTS, or ranks·FST_BN_SLOTS in global-batch
 * data parallelism (SyncBN) after the caller gathered every rank's partials as part[c][rank·FST_BN_SLOTS + slot].
 * train = 0: part may be NULL; stats from the running statistics. */
int fst_bn_finalize(const float* part, int n_slots, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, int train, int C, float eps, float momentum,
                    float* stats, void* stream);
/* out = act(y*scale + shift (+ res*res_scale + res_shift | + res)) */
int fst_bn_apply(const float* y, const float* stats, const float* res, const float* res_stats,
                 float* out, int B, int C, int L, int relu, int64_t numel, void* stream);
/* reductions for backward, as FST_BN_SLOTS partial sums per channel (written, no zero fill, no atomics):
 * red[0][c][slot] = Σ dyʹ, red[1][c][slot] = Σ dyʹ·x̂, with dyʹ = dy·[out>0] when relu */
int fst_bn_bwd_reduce(const float* dy, const float* y, const float* out, const float* stats,
                      int B, int C, int L, int relu, float* red /* [2][C][FST_BN_SLOTS] */, int64_t numel, void* stream);
/* dx = scale·(dyʹ − r0/N − x̂·r1/N) in train mode, scale·dyʹ in eval mode; r = the n_slots partials of red added in slot order
 * (n_slots = FST_BN_SLOTS as fst_bn_bwd_reduce leaves them, or 1 when the caller has already added them, e.g. over ranks);
 * N = B_total·L.  B is the batch of the tensors (launch shape); B_total >= B the batch red was summed over (= B, or all
 * ranks' batches for SyncBN).  red_out (optional, [2C]): receives (r0 | r1) = (dβ | dγ).  row_sums (optional, [B][C]): receives
 * Σ_t dx[b][c][t] — summed over b, the bias gradient of the conv in front of the BatchNorm (OS_CNN.py:67-72: conv1d with bias →
 * BatchNorm1d), which then needs no pass of its own over dx. */
int fst_bn_bwd_apply(const float* dy, const float* y, const float* out, const float* stats, const float* red, int n_slots,
                     float* red_out, float* dx, float* row_sums, int B, int C, int L, int relu, int train, int B_total,
                     int64_t numel, void* stream);

/* ---------------------------------------------------------------------------------------------
 * WaveGlow pieces — Simplified_NF_WaveGlow.py
 * ------------------------------------------------------------------------------------------- */
/* gate (:44-54): g[B,2n,L] → t=tanh(g[:n]), s=sigmoid(g[n:]) written back over g; acts=t·s */
int fst_gate_fwd(float* g_ts, float* acts, int B, int n, int L, int64_t numel_acts, void* stream);
/* dg[:n] = dacts·s·(1−t²), dg[n:] = dacts·t·s·(1−s) */
int fst_gate_bwd(const float* ts, const float* dacts, float* dg, int B, int n, int L, int64_t numel_acts, void* stream);
/* affine coupling (:173-178): xn[:, :h]=u[:, :h]; xn[:, h:] = exp(o[:, h:])·u[:, h:] + o[:, :h].
 * sums (optional float[fst_coupling_sum_slots(B, h, L)][2], written): per-workgroup partials (Σ log_s, Σ xn²) — the
 * full-tensor reductions of WaveGlowLoss (Simplified_NF_WaveGlow.py:230-241) taken in the same pass; the caller adds the slots. */
int64_t fst_coupling_sum_slots(int B, int h, int L);
int fst_coupling_fwd(const float* u, const float* o, float* xn, int B, int h, int L, int64_t numel, float* sums, void* stream);
/* backward of the above given dxn (may be NULL = 0) and (added) d_logs; g_ls / g_sq (each optional, DEVICE scalars) = the
 * cotangents of the two sums: dxn_eff = dxn + 2·g_sq·xn, d log_s += g_ls; writes du (full 2h channels) and do */
int fst_coupling_bwd(const float* u, const float* o, const float* dxn, const float* dlogs, const float* g_ls, const float* g_sq,
                     float* du, float* d_o, int B, int h, int L, int64_t numel, void* stream);
/* inverse coupling (:193-196): xn[:, h:] = (x[:, h:] − o[:, :h]) / exp(o[:, h:]) */
int fst_coupling_inv_fwd(const float* x, const float* o, float* xn, int B, int h, int L, int64_t numel, void* stream);
int fst_coupling_inv_bwd(const float* xn, const float* o, const float* dxn,
                         float* dx, float* d_o, int B, int h, int L, int64_t numel, void* stream);

/* Weight-norm fold of every conv of one WN into the flat weight tensor the fused kernels read (and its backward), ONE launch per
 * direction (Simplified_NF_WaveGlow.py:69-99: old-style weight_norm, w = g·v/‖v‖ per output channel, 18 convs per WN).
 * table (DEVICE, n_rows x 6 int64): per row = one output channel of one conv, or one plain-copy segment (biases, the end conv):
 *   { address of the row of v, address of the row's g scalar (0 = plain copy), element offset into flat, row length,
 *     element offset of the row's v gradient in dpar, element offset of its g gradient in dpar }
 * forward: flat[dst + j] = v[j]·g/‖v‖ (norms[r] = ‖v‖ kept for the backward);  backward: dv = (g/‖v‖)(dw − v (dw·v)/‖v‖²),
 * dg = (dw·v)/‖v‖ written into the gradient buffer dpar (the caller hands out its segments as the parameters' gradients). */
int fst_wn_fold_fwd(const int64_t* table_dev, int n_rows, float* flat, float* norms /* [n_rows] */, void* stream);
int fst_wn_fold_bwd(const int64_t* table_dev, int n_rows, const float* d_flat, const float* norms, float* dpar, void* stream);

/* log|det W| of the invertible 1x1 conv's [n][n] weight (Simplified_NF_WaveGlow.py:40, torch.logdet) and W^{-T}, the gradient
 * d log|det W| / dW its backward needs, in ONE single-workgroup launch (in-place Gauss-Jordan with partial pivoting in double
 * precision in LDS) instead of an LU factorisation, two triangular solves and ~30 tiny launches.  out[0] = log|det W| with
 * torch.logdet's conventions (NaN for a negative determinant, -inf for a singular matrix), out[1] = sign(det);
 * inv_t [n][n] = (W^{-1})^T (NaN-filled when singular).  n <= 96. */
int fst_logdet_inv(const float* W, int n, float* out /* [2] */, float* inv_t /* [n][n] */, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused WN layer (Simplified_NF_WaveGlow.py:101-123 with the gate of :44-54): ONE launch per layer
 *   g = in_layer(a) + cond_layer(u0)[2n·i : 2n·(i+1)]  →  t = tanh(g[:n]), s = sigmoid(g[n:]), acts = t·s
 *   r = res_skip_layer(acts);  a_next = a + r[:n];  out (+)= r[n:]        (last layer: r has n rows, out += r)
 * on the split-bf16 matrix cores; the [B, 2n, L] pre-activation stays in registers.  Needs n < 128 (one spare K row
 * carries the biases through the GEMMs), L % 4 == 0 and 16-byte aligned tensors.
 *   fst_wn_image_bytes(n, h)   size of one layer's packed weight image
 *   fst_wn_pack                effective (weight-norm folded) weights of one layer → image.  in_w [2n][n][3],
 *                              cond_w [2n][h] (this layer's rows of cond_layer), in_b / cond_b [2n],
 *                              rs_w [2n][n] and rs_b [2n]  (last layer: [n][n], [n])
 *   fst_wn_layer_fwd           a [B][n][L] (batch stride a_bs), u0 [B][h][L] (batch stride u0_bs: a channel slice of a wider
 *                              tensor passes without a copy), ts [B][2n][L] ← (t | s) kept for backward, acts [B][n][L] or NULL,
 *                              a_next [B][n][L] (NULL iff last), out [B][n][L] (written if first, else accumulated);
 *                              numel_a = element count of the [B][n][L] tensors as the caller holds them.
 * ------------------------------------------------------------------------------------------- */
int64_t fst_wn_image_bytes(int n, int h);
int fst_wn_pack(const float* in_w, const float* cond_w, const float* in_b, const float* cond_b,
                const float* rs_w, const float* rs_b, int n, int h, int ntaps /* of in_w; the fused kernels are 3-tap: anything
                else is refused (Simplified_NF_WaveGlow.py:60 allows any kernel_size; the host side then takes the generic
                conv-engine path) */, int last, void* image, int64_t image_bytes, void* stream);
int fst_wn_layer_fwd(const float* a, int64_t a_bs, const float* u0, int64_t u0_bs, const void* image, int64_t image_bytes,
                     float* ts, float* acts, float* a_next, float* out, int first, int last,
                     int B, int L, int n, int h, int dil, int64_t numel_a, void* stream);

/* Backward of the same layer through res_skip and the gate, ONE launch:
 *   dacts = W_rsᵀ·[d_a_next ; d_out]   (last layer: W_rsᵀ·d_out),   dg[:n] = dacts·s·(1−t²),  dg[n:] = dacts·t·s·(1−s)
 * (the transposed 1x1 conv autograd derives for Simplified_NF_WaveGlow.py:116 and the gate backward of :44-54).
 * ts = the (t | s) tensor fst_wn_layer_fwd saved; dg [B][2n][L] feeds the in_layer data / weight gradients. */
int64_t fst_wn_bwd_image_bytes(int n, int last);
int fst_wn_pack_bwd(const float* rs_w, int n, int last,
                    int acc_order /* 0: the image fst_wn_layer_bwd reads; 1: the d_a stages in the k-order in which a 32x32 accumulator
                                     tile delivers its rows as a B operand — the image fst_wn_stack_bwd reads (ABI v10) */,
                    void* image, int64_t image_bytes, void* stream);
int fst_wn_layer_bwd(const float* d_a_next /* NULL iff last */, const float* d_out, const float* ts, const void* image,
                     int64_t image_bytes, float* dg,
                     float* row_sums /* optional [256][B·⌈L/128⌉]: per-workgroup Σ_t dg[row]; the caller adds each row's
                                        entries: the in_layer / cond_layer bias gradient without a pass over dg */,
                     int64_t row_sums_rows, int last, int B, int L, int n, int64_t numel_a, void* stream);

/* Data gradient of the same layer's in_layer + cond_layer in ONE launch (the transposed dilated conv and the transposed
 * 1x1 autograd derives for Simplified_NF_WaveGlow.py:107-112):
 *   d_a_new = d_a + Σ_τ W_in[:, :, τ]ᵀ · dg[t + (1 − τ)·dil]      (d_a NULL: no residual cotangent — the last layer)
 *   d_u0   += W_condᵀ · dg
 * in_w [2n][n][3], cond_w [2n][h] (this layer's rows); dg [B][2n][L]; d_a / d_a_new [B][n][L]; d_u0 [B][h][L] with batch stride
 * d_u0_bs (the first h channels of the coupling layer's full-width input gradient are accumulated into in place). */
int64_t fst_wn_dgrad_image_bytes(int n);
int fst_wn_pack_dgrad(const float* in_w, const float* cond_w, int n, int h, int ntaps /* must be 3 */, void* image,
                      int64_t image_bytes, void* stream);
/* 1 when fst_wn_layer_dgrad serves (n, h, dil) — n <= 128, h <= 32 and two window slots of 512 + 2·dil samples fit the 160 KiB
 * of LDS (dil <= 128: WN stacks of up to 8 layers) — else 0: the caller then uses fst_conv_gemm's data-gradient plan. */
int fst_wn_dgrad_fits(int n, int h, int dil);
int fst_wn_layer_dgrad(const float* dg, const void* image, int64_t image_bytes, const float* d_a, float* d_a_new, float* d_u0,
                       float* row_sums /* optional [128][B·⌈L/512⌉]: per-workgroup Σ_t d_a_new[row] */, int64_t row_sums_rows,
                       int B, int L, int n, int h, int dil, int64_t numel_a, int64_t d_u0_bs, void* stream);

/* The forward of a WHOLE WN stack (every layer of fst_wn_layer_fwd) as ONE persistent launch: one workgroup walks all 256-sample
 * tiles of its batch elements through all layers — a_next / out of one layer are read by the next through the same CU — so no launch
 * boundary synchronises the chip and every other workgroup starts half a tile late: one half of the CUs stores while the other
 * multiplies (the per-layer launches alternate between an idle and a saturated memory system).  Same arithmetic, same results as the
 * per-layer launches.  Tables of nl entries (host memory, copied into the kernel arguments): a_in[i] = input of layer i with batch
 * stride a_bs[i] (a_in[0] = the start conv's output, a_in[i+1] = a_next[i]), images[i] = fst_wn_pack image, ts[i] [B][2n][L] written,
 * a_next[i] [B][n][L] written (NULL on the last layer); out [B][n][L] the skip sum (written).  fst_wn_stack_fwd_ok: n < 128,
 * L % 256 == 0, nl <= 10.  Replaces the layer loop of WN.forward, /root/reference/Simplified_NF_WaveGlow.py:101-123. */
int fst_wn_stack_fwd_ok(int n, int h, int L, int nl);
int fst_wn_stack_fwd(const float* const* a_in, const int64_t* a_bs, const void* const* images, int64_t image_bytes, float* const* ts,
                     float* const* a_next, const float* u0, int64_t u0_bs, float* out, int nl, int B, int L, int n, int h,
                     int64_t numel_a, void* stream);

/* The whole backward of a WN stack — fst_wn_layer_bwd and fst_wn_layer_dgrad of every layer, top layer first — as ONE persistent
 * launch (the backward autograd derives for the loop of Simplified_NF_WaveGlow.py:104-121), for sequences of up to 512 samples:
 * a 512-sample tile is then a whole sequence, the dilated taps never leave it, and one workgroup walks all layers of its batch
 * element with the cotangent of the residual stream (d_a) kept in its accumulators from layer to layer; what a phase writes
 * (dg, d_u0) the next reads back from L2 / the Infinity Cache, and the CUs are not held in step by launch boundaries.
 * All tables are HOST arrays of nl entries (layer 0 = dilation 1 first; layer i has dilation 2^i):
 *   ts[i] the saved gate halves; img_b[i] the fst_wn_pack_bwd image with acc_order = 1; img_d[i] the fst_wn_pack_dgrad image;
 *   dg[i] [B][2n][L] written (entries may alias each other when nothing reads dg afterwards); da_out[i] [B][n][L] the cotangent
 *   of layer i's input, written when not NULL (da_out[0], the start conv's cotangent, is required; the others are the d_a
 *   operands of the res_skip weight gradients: da_out[i+1] is the residual cotangent entering layer i); rs_b[i] / rs_d[i]
 *   optional [256][B] / [128][B] per-sequence row sums of dg / da_out (both tables or neither).
 * d_u0 [B][h][L] (batch stride d_u0_bs) is accumulated into in place.  fst_wn_stack_bwd_ok: 1 when (n, h, L, nl) is served. */
int fst_wn_stack_bwd_ok(int n, int h, int L, int nl);
int fst_wn_stack_bwd(const float* const* ts, const void* const* img_b, const void* const* img_d, float* const* dg,
                     float* const* da_out, float* const* rs_b /* optional */, float* const* rs_d /* optional */,
                     const float* d_out, float* d_u0, int64_t d_u0_bs, int nl, int B, int L, int n, int h, int64_t numel_a,
                     void* stream);

/* Weight gradients of the same layer (the gradients autograd derives for Simplified_NF_WaveGlow.py:107-116), time as the MFMA
 * reduction index, split-bf16 products, per-workgroup partial slabs added in a fixed order (deterministic, no atomics):
 *   fst_wn_wgrad_in   dw_in[m][c][τ] = Σ_{b,t} dg[b,m,t]·a[b,c,t+(τ−1)·dil]   ([2n][n][3]),   dw_cond[m][c] = Σ dg[b,m,t]·u0[b,c,t]  ([2n][h])
 *   fst_wn_wgrad_rs   dw_rs[m][c] = Σ_{b,t} [d_a ; d_out][b,m,t]·(t·s)[b,c,t]  ([2n][n]; last layer: d_a NULL, [n][n]) — acts = t·s is
 *                     re-formed from the gate halves ts [B][2n][L] the fused forward saved
 * kind 0 = in_layer + cond_layer, 1 = res_skip.  fst_wn_wgrad_ok: 1 when the shape is served (L % 32 == 0, n < 128, h <= 32 and for
 * kind 0 dil % 4 == 0); 2 (kind 0, dil < 4) when it is served provided the caller guarantees 16 readable bytes in front of and
 * behind `a` (a_slack: the 16-byte pieces of a shifted tap start up to 3 samples outside a row); 0: the caller uses fst_conv_wgrad.
 * Operand SETS: the applications of one WN in a train step (forward on the target batch, forward on the source batch, inverse)
 * share their weights, so their gradients are summed; dg/a/u0 (d_a/d_out/ts) are HOST arrays of n_sets (1..3) device pointers to
 * same-shaped operands and one launch sums over all of them — the partial slabs and their reduction, which cost as much as the
 * products of one set, are paid once instead of n_sets times.
 * workspace: fst_wn_wgrad_workspace_floats(...) floats, written (independent of n_sets). */
int fst_wn_wgrad_ok(int kind, int B, int L, int n, int h, int dil);
int64_t fst_wn_wgrad_workspace_floats(int kind, int B, int L, int n, int h, int last);
int fst_wn_wgrad_in(const float* const* dg, const float* const* a, const float* const* u0, int n_sets, int64_t u0_bs, float* dw_in,
                    float* dw_cond, float* workspace, int64_t workspace_floats, int B, int L, int n, int h, int dil, int a_slack,
                    int64_t numel_a, void* stream);
int fst_wn_wgrad_rs(const float* const* d_a /* NULL iff last */, const float* const* d_out, const float* const* ts, int n_sets,
                    float* dw_rs, float* workspace, int64_t workspace_floats, int last, int B, int L, int n, int64_t numel_a,
                    void* stream);

/* C[m][n] = Σ_k A[m][k]·Bm[n][k]  (A [M][K], Bm [N][K] row-major, the reduction index contiguous in both; M <= 256, K % 32 == 0) on the
 * time-as-k kernel above: RandomLayer's feature-side product x·R₀ (/root/reference/C_DAN.py:21; Bm = R₀ᵀ kept contiguous) and its data
 * gradient dy·R₀ᵀ (Bm = R₀).  The K split leaves partial slabs in `workspace` (fst_nt_gemm_workspace_floats), added in a fixed order.
 * Optional epilogue = the rest of RandomLayer.forward (C_DAN.py:22-25): C[m][n] = acc·epi_scale·Σ_c epi_p[m][c]·epi_r1[c][n] with the
 * plain product acc stored in epi_raw (may be NULL); epi_p = epi_r1 = NULL: C = acc. */
int64_t fst_nt_gemm_workspace_floats(int M, int N, int K);
int fst_nt_gemm(const float* A, const float* Bm, float* C, float* workspace, int64_t workspace_floats, int M, int N, int K,
                const float* epi_p /* [M][ncls] */, const float* epi_r1 /* [ncls][N] */, int epi_ncls, float epi_scale,
                float* epi_raw /* [M][N] */, void* stream);

/* dW[m][c][τ] = Σ_{b,t} dy[b][m][t]·x[b][c][t + τ·dil − pad_left]: the dense weight gradient of a conv with at most four taps on the
 * time-as-k kernel (each tap = one k-row segment with its own shift) — the Q1 gradient of the shared omni-scale block's last layer
 * (/root/reference/OS_CNN/OS_CNN.py:67-71, 225 → 50 channels, two taps; train_and_test.py:685-690 reads it).  fst_tap_wgrad_ok: 1 served,
 * 2 served provided 16 bytes either side of x are readable (x_slack: a tap shift that is not a multiple of 4 samples, within ±3),
 * 0 not served (M > 256, L % 32 != 0, more than 4 taps, larger misaligned shifts): the caller uses fst_conv_wgrad. */
int fst_tap_wgrad_ok(int B, int L, int M, int C, int ntaps, int dil, int pad_left);
int64_t fst_tap_wgrad_workspace_floats(int B, int L, int M, int C, int ntaps);
int fst_tap_wgrad(const float* dy, const float* x, float* dw /* [M][C][ntaps], written */, float* workspace, int64_t workspace_floats,
                  int B, int L, int M, int C, int ntaps, int dil, int pad_left, int x_slack, int64_t numel_dy, int64_t numel_x,
                  void* stream);

/* The DENSE weight gradient of an omni-scale layer (every tap k < K <= 96 of every row, dilation 1; quirk Q1: the reference convolves the
 * dense Kmax kernel, /root/reference/OS_CNN/OS_CNN.py:67-71, and GradNorm's norms, train_and_test.py:685-690, run over the dense gradient):
 *   dw[m][c][k] = Σ_{b,t} dy[b][m][t]·x[b][c][t + k − pad_left]          (M <= 256, L % 32 == 0)
 * time as the MFMA reduction index; the K k-rows of a channel are read from eight pre-shifted bf16 copies of ONE staged window of the
 * channel (csrc/wn_wgrad.hip).  Partial slabs in `workspace` (fst_dense_tap_wgrad_workspace_floats), added in a fixed order: no atomics. */
int fst_dense_tap_wgrad_ok(int B, int L, int M, int C, int K, int pad_left);
int64_t fst_dense_tap_wgrad_workspace_floats(int B, int L, int M, int C, int K);
int fst_dense_tap_wgrad(const float* dy, const float* x, float* dw /* [M][C][K], written */, float* workspace, int64_t workspace_floats,
                        int B, int L, int M, int C, int K, int pad_left, int64_t numel_dy, int64_t numel_x, void* stream);

/* NoiseTransfer (/root/reference/widgets.py:150-167): new_t = avg_t + r_t·mean_b(z_t), new_s likewise, dist = new_t − new_s,
 * learned = selu(W·dist + bias) (unbatched 1x1 conv over the [C, L] map), out[b] = learned + z_s[b].
 *   fst_batch_sum            part[z][s][i] = Σ_{b in slice s} x_z[b][i] (z < 2 tensors, x1 may be NULL; S contiguous slices of the
 *                            batch, N = C·L, N % 4 == 0) — partials added in slice order by the consumers: deterministic
 *   fst_noise_transfer_fwd   slice sums [2][S][C·L] → avg_t, avg_s updated IN PLACE (the reference's detached state), dist, pre = W·dist +
 *                            bias and learned = selu(pre) written ([C][L] each).  r_*_dev: the call's accumulation ratios as device
 *                            scalars (a captured step refreshes them between replays) or NULL: the host values r_t, r_s
 *   fst_bcast_add            out[b][i] = x[b][i] + v[i]
 *   fst_noise_transfer_bwd   slice sums [S][C·L] of the cotangent of out → dpre = Σ_b g · selu'(pre), dd = Wᵀ·dpre (cotangent of dist)
 *   fst_noise_transfer_dw    dW[o][c] = Σ_l dpre[o][l]·dist[c][l], dbias[o] = Σ_l dpre[o][l]
 *   fst_noise_transfer_bwd_apply   dz_t[b] = (r_t/B)·dd,  dz_s[b] = g[b] − (r_s/B)·dd  (either output may be NULL) */
int fst_batch_sum(const float* x0, const float* x1, float* part, int B, int64_t N, int S, void* stream);
int fst_noise_transfer_fwd(const float* part, int S, int B, const float* r_t_dev, const float* r_s_dev, float r_t, float r_s,
                           float* avg_t, float* avg_s, const float* W, const float* bias, float* dist, float* pre, float* learned,
                           int C, int L, void* stream);
int fst_bcast_add(float* out, const float* x, const float* v, int B, int64_t N, void* stream);
int fst_noise_transfer_bwd(const float* part, int S, const float* pre, const float* W, float* dpre, float* dd, int C, int L,
                           void* stream);
int fst_noise_transfer_dw(const float* dpre, const float* dist, float* dW, float* dbias, int C, int L, void* stream);
int fst_noise_transfer_bwd_apply(const float* g, const float* dd, const float* r_t_dev, const float* r_s_dev, float r_t, float r_s,
                                 int B, float* dz_t, float* dz_s, int64_t N, void* stream);

/* generic fp32 elementwise helpers on contiguous buffers */
int fst_relu_bwd(const float* dy, const float* y, float* out, int64_t n, void* stream);   /* out = dy·[y > 0] (epilogue-fused ReLUs) */

/* Dense products of the small heads on the matrix cores in split-bf16 — replaces the aten addmm / mm calls behind
 * /root/reference/widgets.py:73-75 (DimensionUnification.length_unification: [B·C_s, L_s] × [L_t, L_s]ᵀ + ReLU), :113-131
 * (AdversarialNetworkforCDAN's Linear-ReLU chain), :32-42 (FeatureDiscriminatorforSource's Linear-LeakyReLU chain) and their gradients:
 *     C[m][n] = act( Σ_k A(m,k)·B(n,k) + bias[n] ),   m < M, n < N, k < K,   C row-major with row pitch ldc
 * ta = 0: A(m,k) = A[m·lda + k] (reduction index contiguous);  ta = 1: A(m,k) = A[k·lda + m] (reduction index major);  tb likewise for
 * B(n,k) — so   y = x·Wᵀ is (x, 0, W, 0),   dx = g·W is (g, 0, W, 1),   dW = gᵀ·x is (g, 1, x, 1),   no transposed copies.
 * Any M, N, K and pitches; 16-byte loads where rows allow them.  K is split across workgroups when the tiles alone do not fill the
 * chip: partial tiles go to `workspace` (fst_gemm_workspace_floats, 0 when no split is used) and are added in a fixed order — the
 * same bits on every run.  bias (may be NULL) is per output column; act = FST_ACT_*, slope = LeakyReLU's negative slope. */
#define FST_ACT_NONE  0
#define FST_ACT_RELU  1
#define FST_ACT_LEAKY 2
int64_t fst_gemm_workspace_floats(int M, int N, int K);
int fst_gemm(const float* A, int64_t lda, int ta, const float* B, int64_t ldb, int tb, float* C, int64_t ldc, int M, int N, int K,
             const float* bias, int act, float slope, float* workspace, int64_t workspace_floats, void* stream);
/* out = dy·act'(y) from the activation's output y: dy where y > 0, slope·dy elsewhere (slope = 0: ReLU) — nn.ReLU / nn.LeakyReLU backward
 * behind an activation that ran in fst_gemm's epilogue. */
int fst_act_bwd(const float* dy, const float* y, float* out, int64_t n, float slope, void* stream);
int fst_axpy(float* y, const float* x, float alpha, int64_t n, void* stream);          /* y += alpha*x */
int fst_add_slices(float* dst, int64_t dst_bs, const float* a, int64_t a_bs, const float* b, int64_t b_bs,
                   int B, int C, int L, void* stream);                                  /* dst = a + b (b may be null) */

/* ---------------------------------------------------------------------------------------------
 * CPC InfoNCE (Comparison/SLARDA/train.py:72-76): for every prediction step i,
 * total_i = enc_i·pred_iᵀ (B×B, K=C); nce_sum = Σ_i Σ_b log_softmax(total_i)[b,b]  (the caller
 * multiplies by −1/(B·T)).  enc element (i,b,c) is read in place at enc[i*s_i + b*s_b + c*s_c]
 * (a strided view of the [B,C,L] feature tensor); pred is [T,B,C] contiguous; lse [T,B] is kept
 * for backward.  gout is the DEVICE scalar d loss / d nce.  t0_dev (optional DEVICE int32 scalar) adds
 * t0_dev[0]·s_i elements to enc/denc, so the random start index can change between hipGraph replays.
 * Rows (B encodings of this rank) and columns (Bc predictions) may differ: "global batch" data parallelism scores
 * the local rows against the predictions gathered from every rank, the positives at columns col_off + b.
 * ------------------------------------------------------------------------------------------- */
/* Up to 256 columns and 64 channels (split-bf16 arithmetic): the forward transposes the T steps it reads into a workspace
 * [T][B][C] (whole lines in, whole lines out) and runs the cross-Gram on v_mfma_f32_32x32x16_bf16 with hi/lo operands (ABI v11).
 * More than 256 columns are processed as panels of 256: the forward then needs a workspace of
 * fst_cpc_workspace_floats(T, B, C, Bc) floats for the per-panel softmax statistics, and the backward
 * accumulates denc across panels with fp32 atomics (the caller zero-fills denc — it must anyway outside [t0, t0+T)). */
int64_t fst_cpc_workspace_floats(int T, int B, int C, int Bc);
/* nce_sum is an array of fst_cpc_nce_slots(T, B, Bc) partial sums, one per workgroup, every slot written (no zero fill): the
 * caller adds them in slot order, so the loss is bit-identical from run to run (ABI v9; a zeroed scalar fed by float atomics before). */
int64_t fst_cpc_nce_slots(int T, int B, int C, int Bc);
int fst_cpc_nce_fwd(const float* enc, int64_t s_i, int64_t s_b, int64_t s_c, const int32_t* t0_dev /* optional */,
                    const float* pred /* [T][Bc][C] */, int T, int B, int C,
                    int Bc /* columns = negatives; = B on one GPU */, int col_off /* column of row 0's positive */,
                    float* lse /* [T][B] */, float* nce_sum /* [fst_cpc_nce_slots] partial sums */, float* workspace /* NULL if Bc <= 256 */,
                    void* stream);
int fst_cpc_nce_bwd(const float* enc, int64_t s_i, int64_t s_b, int64_t s_c, const int32_t* t0_dev /* optional */,
                    const float* pred, const float* lse, int T, int B, int C, int Bc, int col_off, const float* gout,
                    float* denc /* same strides as enc */, float* dpred /* [T][Bc][C] */, void* stream);

/* ---------------------------------------------------------------------------------------------
 * GRU recurrence of the CPC context network (Comparison/SLARDA/train.py:65-67: nn.GRU(C, 64, batch_first=True), h0 = 0),
 * one persistent launch per direction.  xproj [B][S][3H] = x_t·W_ihᵀ + b_ih for every step (gate order r | z | n, torch's);
 * the kernel runs steps 0..t_last (t_last_dev: optional DEVICE scalar, read instead of t_last so a captured graph can vary
 * it), saves h_all [B][S][H] and gates [B][S][4H] = (r, z, n, W_hn·h + b_hn).  Only h at step t_last is consumed by CPC.
 * Backward: dout [B][H] = d loss / d h_{t_last}; writes dxproj [B][S][3H] (→ d input, dW_ih, db_ih by GEMMs outside) and
 * dgh [B][S][3H] (dW_hh = Σ dgh ⊗ h_{t−1}, db_hh = Σ dgh); both must be zero-filled by the caller (steps > t_last stay 0).
 * H = 64 only.  numel_h = element count of h_all as the caller holds it.
 * ------------------------------------------------------------------------------------------- */
int fst_gru_fwd(const float* xproj, const float* w_hh, const float* b_hh, float* h_all, float* gates,
                const int32_t* t_last_dev, int t_last, int B, int S, int H, int64_t numel_h, void* stream);
int fst_gru_bwd(const float* w_hh, const float* h_all, const float* gates, const float* dout,
                const int32_t* t_last_dev, int t_last, float* dxproj, float* dgh, int B, int S, int H,
                int64_t numel_h, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The 2-step LSTM of ProbTransfer (widgets.py:46-55: nn.LSTM(C, C, batch_first=True) over the pooled feature repeated
 * twice, h0 = c0 = 0; only h_n is consumed), one launch per direction.  xproj [B][4H] = x·W_ihᵀ + b_ih + b_hh (gate order
 * i | f | g | o, torch's; the same projection feeds both steps).  Forward: w_hh_t = W_hhᵀ [H][4H]; writes h2 [B][H] and
 * save [B][11H] = gates of step 1 | gates of step 2 | c1 | c2 | h1.  Backward: dh2 [B][H] = d loss / d h_n; writes
 * dxproj [B][4H] (→ d x, dW_ih, db_ih = db_hh by GEMMs outside) and dpre2 [B][4H] (dW_hh = Σ_b dpre2 ⊗ h1, h1 = save[..., 10H:]).
 * H <= 256.  numel_xproj = element count of the [B][4H] tensors as the caller holds them.
 * ------------------------------------------------------------------------------------------- */
int fst_lstm2_fwd(const float* xproj, const float* w_hh_t, float* h2, float* save, int B, int H, int64_t numel_xproj, void* stream);
int fst_lstm2_bwd(const float* w_hh, const float* save, const float* dh2, float* dxproj, float* dpre2, int B, int H,
                  int64_t numel_xproj, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Optimiser updates (train_and_test.py:97-106 torch.optim.RMSprop x10, Adam for CPC; stepped at :742-754) as single-pass
 * multi-tensor launches: the *_host arguments are HOST arrays of n_tensors device pointers / element counts / learning rates;
 * up to 64 tensors ride in one launch's kernel arguments (so a captured hipGraph replays them as they are).  Same update
 * formulas and operation order as torch (RMSprop: centered = False, momentum = 0; Adam: the capturable branch with ONE shared
 * device step counter, already incremented by the caller).
 * ------------------------------------------------------------------------------------------- */
int fst_rmsprop_multi(float* const* p_host, const float* const* g_host, float* const* v_host, const int64_t* numel_host,
                      const float* lr_host, int n_tensors, float alpha, float eps, void* stream);
int fst_adam_multi(float* const* p_host, const float* const* g_host, float* const* m_host, float* const* v_host,
                   const int64_t* numel_host, int n_tensors, const float* step_dev, float lr, float beta1, float beta2, float eps,
                   void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FST_HIP_H */
