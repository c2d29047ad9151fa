// CPC InfoNCE cross-Gram (Comparison/SLARDA/train.py:72-76), fused: for each prediction step i the
// B×B matrix total_i = enc_i · pred_iᵀ is never materialised.  Forward: the Gram product runs on the
// matrix cores (exact f32 MFMA) with the log-softmax and the diagonal taken on the accumulators.
// Backward: the same Gram tile re-formed on MFMA, turned into dt in the accumulators, which then feed
// dpred += dtᵀ·enc directly as an MFMA operand; denc = dt·pred goes through one LDS transpose.
// The whole loss is ≈1.7 GFLOP per call at B=256, T=256.
//
// enc is read in place from the feature tensor through (s_i, s_b, s_c) element strides, so no
// [T,B,C] gather copy is made; denc is written through the same strides.
//
// Rows (enc samples, B) and columns (pred samples = negatives, Bc) may differ: in "global batch" data
// parallelism a rank scores its B local rows against the Bc = world·B gathered predictions, its own
// positives sitting at columns col_off + b (SURVEY §8e mode B).  Single GPU: Bc = B, col_off = 0.
//
// More than 256 columns (8 ranks x 256 samples = 2048 negatives): the columns are cut into PANELS of 256 that fit LDS;
// a workgroup handles one panel (blockIdx.z / .y), the forward leaves per-(row, panel) partial (max, Σexp, diagonal) in a
// workspace that cpc_combine_kernel merges (online-softmax combine: lse = M + log Σ_p s_p·e^{m_p − M}), the backward —
// lse known — is independent per panel: dpred rows of the panel written, denc accumulated over panels with fp32 atomics.
#include "fst_common.h"
#include <string.h>

#define CPC_ROWS 16

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum_all(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

struct CpcParams {
  const float* enc;
  long long s_i, s_b, s_c;
  const float* pred;   // [T][B][C] contiguous
  float* lse;          // [T][B]
  float* nce_sum;      // [fst_cpc_nce_slots]: one partial sum per workgroup (per wave of the combine kernel), every slot written
  const float* gout;   // scalar upstream gradient of nce (device), bwd only
  float* denc;         // same strides as enc
  float* dpred;        // [T][B][C]
  const int* t0_dev;   // optional device scalar: extra time offset of enc/denc (elements of stride s_i)
  int T, B, C;
  int Bc, col_off;     // columns (pred rows) and the column of row 0's positive
  int n_panels;        // column panels of CPC_PANEL (1 when Bc <= CPC_PANEL)
  float* ws;           // [T][B][n_panels][3] partial (max, Σexp, diag) — forward, n_panels > 1 only
};

// Forward: the Gram product on the matrix cores.  A workgroup owns 32 rows (encodings) of step i against all Bc
// columns (predictions): total[32][Bc] = enc[32][C] · predᵀ[C][Bc] as v_mfma_f32_32x32x2_f32 (exact fp32 chain,
// K = C: 25 k-steps at C = 50), wave w taking column blocks w, w+4 (two 32x32 tiles).  Both operands come from LDS
// ([rows][C|1], odd stride: lane <-> row reads hit distinct banks).  The log-softmax over a row runs on the
// accumulators: C/D layout puts a tile's 32 columns on the 32 lanes of a half-wave and 16 rows in its registers, so a
// row's max / Σexp / diagonal is a 5-step butterfly inside the half-wave, then a 4-entry combine through LDS.
#define CPC_MAXTILES 2   // column tiles per wave: a panel is <= 4 * 32 * CPC_MAXTILES = 256 columns
#define CPC_PANEL (128 * CPC_MAXTILES)

__global__ __launch_bounds__(256) void cpc_fwd_kernel(CpcParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if (p.t0_dev) p.enc += (long long)p.t0_dev[0] * p.s_i;
  const int i = blockIdx.x;
  const int col0 = blockIdx.z * CPC_PANEL;     // this workgroup's column panel [col0, col0 + Bc)
  const int B = p.B, Bc = min(CPC_PANEL, p.Bc - col0), C = p.C, PS = C | 1;
  float* predl = lds;                          // [Bc][PS]   staged once per workgroup
  float* encl = predl + (size_t)Bc * PS;       // [32][PS]   one row block at a time
  float* red = encl + 32 * PS;                 // [3][4 waves][32 rows]: max, Σexp, diagonal
  const float* predg = p.pred + ((long long)i * p.Bc + col0) * C;
  for (int idx = threadIdx.x; idx < Bc * C; idx += 256) {
    const int j = idx / C, c = idx - j * C;
    predl[j * PS + c] = predg[idx];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
  const int ntiles = (Bc + 31) / 32;
  float local = 0.f;
  for (int r0 = blockIdx.y * 32; r0 < B; r0 += gridDim.y * 32) {
  __syncthreads();                             // previous row block's readers of encl / red are done
  for (int idx = threadIdx.x; idx < 32 * C; idx += 256) {
    const int r = idx / C, c = idx - r * C;
    const int b = r0 + r;
    encl[r * PS + c] = b < B ? p.enc[i * p.s_i + b * p.s_b + c * p.s_c] : 0.f;
  }
  __syncthreads();
  f32x16 acc[CPC_MAXTILES];
#pragma unroll
  for (int t = 0; t < CPC_MAXTILES; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const float* arow = encl + l31 * PS + half;
  for (int s = 0; 2 * s < C; ++s) {
    const bool k_ok = 2 * s + half < C;
    const float a = k_ok ? arow[2 * s] : 0.f;
#pragma unroll
    for (int t = 0; t < CPC_MAXTILES; ++t) {
      const int j = (wave + 4 * t) * 32 + l31;
      const float bv = (k_ok && j < Bc) ? predl[j * PS + 2 * s + half] : 0.f;
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[t], 0, 0, 0);
    }
  }
  // row statistics: register r of a lane is row (r&3) + 8*(r>>2) + 4*half of the block, column = tile*32 + l31
  float mx[16], se[16], dg[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
    float m = -INFINITY, d = 0.f;
#pragma unroll
    for (int t = 0; t < CPC_MAXTILES; ++t) {
      const int j = (wave + 4 * t) * 32 + l31;
      if (wave + 4 * t < ntiles && j < Bc) {
        m = fmaxf(m, acc[t][r]);
        if (col0 + j == r0 + row + p.col_off) d = acc[t][r];
      }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) { m = fmaxf(m, __shfl_xor(m, o, 64)); d += __shfl_xor(d, o, 64); }
    mx[r] = m; dg[r] = d;
    if (l31 == 0) { red[wave * 32 + row] = m; red[256 + wave * 32 + row] = d; }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
    const float m = fmaxf(fmaxf(red[row], red[32 + row]), fmaxf(red[64 + row], red[96 + row]));
    mx[r] = m;
    float sacc = 0.f;
#pragma unroll
    for (int t = 0; t < CPC_MAXTILES; ++t) {
      const int j = (wave + 4 * t) * 32 + l31;
      if (wave + 4 * t < ntiles && j < Bc) sacc += expf(acc[t][r] - m);
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o, 64);
    se[r] = sacc;
    if (l31 == 0) red[128 + wave * 32 + row] = sacc;
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      const int b = r0 + row;
      const float s4 = red[128 + row] + red[160 + row] + red[192 + row] + red[224 + row];
      const float d4 = red[256 + row] + red[288 + row] + red[320 + row] + red[352 + row];
      const float lse = mx[r] + logf(s4);
      if (l31 == 0 && b < B) {
        if (p.n_panels == 1) {
          p.lse[(long long)i * B + b] = lse;
          local += d4 - lse;
        } else {                                 // partial statistics of this panel; cpc_combine_kernel merges them
          float* w = p.ws + (((long long)i * B + b) * p.n_panels + blockIdx.z) * 3;
          w[0] = mx[r]; w[1] = s4; w[2] = d4;
        }
      }
    }
  }
  }   // row blocks
  if (wave == 0 && p.n_panels == 1) {
    local += __shfl_xor(local, 32, 64);                    // the two halves hold different rows
    // one slot per workgroup, summed by the caller in slot order: the loss is the same number in every run (a float atomic
    // per workgroup added them in arrival order)
    if (lane == 0) p.nce_sum[blockIdx.x * gridDim.y + blockIdx.y] = local;
  }
}

// ------------------------------------------------------------------------------------------------
// Forward, one panel (Bc <= 256), C <= 64: the cross-Gram on the bf16 matrix cores with split operands
// (hi·hi + hi·lo + lo·hi on v_mfma_f32_32x32x16_bf16, fp32 accumulation: ≈5e-6 of the logit scale), K = C padded to 64.
//
// cpc_fwd_kernel reads enc_i[b][c] = feat[b][c][t0 + i] in place: for a fixed step every element sits in a different 128-byte
// line, so a workgroup pulls 12.8 k lines for 51 KB of operand and the launch moves 130 MB for 39 MB of operands; and its
// exact-f32 MFMA (25 k-steps of 64 cycles per tile) runs at 12 % of even the f32 peak.  Here
//   cpc_enc_gather_kernel  transposes the T steps the loss reads into enc_t [T][B][C] through LDS — whole lines in, whole
//                          lines out (13 + 13 MB);
//   cpc_gram_bf3_kernel    one workgroup of 8 waves per step: pred_i and enc_i are staged ONCE as bf16 hi / lo images
//                          [256 rows][64 channels] (144-byte rows: a 16-lane group's ds_read_b128 covers all 64 banks), wave w
//                          keeps the B fragments of column tile w in registers for all eight row blocks (4 k-steps x 3 MFMAs
//                          per tile), the log-softmax runs on the accumulators as in cpc_fwd_kernel.
// ------------------------------------------------------------------------------------------------
typedef __bf16 cg_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 cg_bf16x2 __attribute__((ext_vector_type(2)));
typedef float cg_f32x2 __attribute__((ext_vector_type(2)));
#define CG_ROWB 144                                  // bytes per image row: 64 bf16 + 16 B so that rows start 36 banks apart
#define CG_IMG (256 * CG_ROWB)

__device__ __forceinline__ void cg_split_pair(float a, float b, unsigned& hi, unsigned& lo) {
  const cg_f32x2 v = {a, b};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, cg_bf16x2));
  const cg_f32x2 r = {a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u)};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, cg_bf16x2));
}

// enc_t[i][r] = enc[i·s_i + (b, c) strides], r = b·C + c: 64 rows x 32 steps per workgroup through an LDS tile
__global__ __launch_bounds__(256) void cpc_enc_gather_kernel(CpcParams p, float* __restrict__ enc_t) {
  __shared__ float tile[32][65];
  if (p.t0_dev) p.enc += (long long)p.t0_dev[0] * p.s_i;
  const int R = p.B * p.C, r0 = blockIdx.x * 64, i0 = blockIdx.y * 32;
  for (int idx = threadIdx.x; idx < 64 * 32; idx += 256) {
    const int rr = idx >> 5, st = idx & 31;          // 32 consecutive lanes = 32 consecutive steps of one (b, c) row
    const int r = r0 + rr, i = i0 + st;
    float v = 0.f;
    if (r < R && i < p.T) {
      const int b = r / p.C, c = r - b * p.C;
      v = p.enc[(long long)i * p.s_i + (long long)b * p.s_b + (long long)c * p.s_c];
    }
    tile[st][rr] = v;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 64 * 32; idx += 256) {
    const int st = idx >> 6, rr = idx & 63;          // 64 consecutive lanes = 64 consecutive (b, c) rows of one step
    const int r = r0 + rr, i = i0 + st;
    if (r < R && i < p.T) enc_t[(long long)i * R + r] = tile[st][rr];
  }
}

__global__ __launch_bounds__(512) void cpc_gram_bf3_kernel(CpcParams p, const float* __restrict__ enc_t) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* const ldsb = reinterpret_cast<char*>(lds);
  char* const ph = ldsb, * const pl = ph + CG_IMG, * const eh = pl + CG_IMG, * const el = eh + CG_IMG;
  float* const red = reinterpret_cast<float*>(el + CG_IMG);      // [3][8 waves][32 rows]: max, Σexp, diagonal
  const int i = blockIdx.x, B = p.B, Bc = p.Bc, C = p.C;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  // stage both operands: an item = 4 channels of one row -> 8 bytes of the hi image and of the lo image
  auto stage = [&](const float* src, int rows, char* hi_img, char* lo_img) {
    for (int idx = tid; idx < 256 * 16; idx += 512) {
      const int row = idx >> 4, g4 = idx & 15;
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = 4 * g4 + k;
        v[k] = (row < rows && c < C) ? src[(long long)row * C + c] : 0.f;
      }
      unsigned h0, h1, l0, l1;
      cg_split_pair(v[0], v[1], h0, l0);
      cg_split_pair(v[2], v[3], h1, l1);
      *reinterpret_cast<uint2*>(hi_img + row * CG_ROWB + g4 * 8) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(lo_img + row * CG_ROWB + g4 * 8) = make_uint2(l0, l1);
    }
  };
  stage(p.pred + (long long)i * Bc * C, Bc, ph, pl);
  stage(enc_t + (long long)i * B * C, B, eh, el);
  __syncthreads();
  // The tile is computed TRANSPOSED — rows = predictions j (this wave's 32: its A fragments stay in registers), columns = the
  // block's 32 encodings b — so the softmax of sample b runs over a lane's own 16 registers, one exchange between the lane
  // halves and a combine over the eight waves through LDS (with samples on the rows a row's maximum is a 5-step butterfly per
  // register: 240 dependent cross-lane exchanges per row block, which is what bounded the f32 kernel).
  cg_bf16x8 ah[4], al[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    ah[ks] = *reinterpret_cast<const cg_bf16x8*>(ph + (wave * 32 + l31) * CG_ROWB + ks * 32 + half * 16);
    al[ks] = *reinterpret_cast<const cg_bf16x8*>(pl + (wave * 32 + l31) * CG_ROWB + ks * 32 + half * 16);
  }
  float local = 0.f;
  for (int r0 = 0; r0 < B; r0 += 32) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const cg_bf16x8 bh = *reinterpret_cast<const cg_bf16x8*>(eh + (r0 + l31) * CG_ROWB + ks * 32 + half * 16);
      const cg_bf16x8 bl = *reinterpret_cast<const cg_bf16x8*>(el + (r0 + l31) * CG_ROWB + ks * 32 + half * 16);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks], bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks], bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks], bh, acc, 0, 0, 0);
    }
    // register r of a lane: prediction j = wave·32 + (r&3) + 8(r>>2) + 4·half, encoding b = r0 + l31
    const int jb = wave * 32 + 4 * half;
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = jb + (r & 3) + 8 * (r >> 2);
      if (j < Bc) m = fmaxf(m, acc[r]);
      if (j == r0 + l31 + p.col_off) red[512 + l31] = acc[r];        // the positive of sample b: exactly one lane of one wave
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    if (half == 0) red[wave * 32 + l31] = m;
    __syncthreads();
    float M = red[l31];
#pragma unroll
    for (int w = 1; w < 8; ++w) M = fmaxf(M, red[w * 32 + l31]);
    float se = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = jb + (r & 3) + 8 * (r >> 2);
      if (j < Bc) se += __expf(acc[r] - M);
    }
    se += __shfl_xor(se, 32, 64);
    if (half == 0) red[256 + wave * 32 + l31] = se;
    __syncthreads();
    if (wave == 0 && half == 0) {
      const int b = r0 + l31;
      float s8 = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) s8 += red[256 + w * 32 + l31];
      if (b < B) {
        const float lse = M + logf(s8);
        p.lse[(long long)i * B + b] = lse;
        local += red[512 + l31] - lse;
      }
    }
    __syncthreads();                                   // the statistics arrays are rewritten by the next row block
  }
  if (wave == 0) {
    local = wave_sum_all(local);                       // (lanes 32-63 hold zeros)
    if (lane == 0) p.nce_sum[i] = local;               // one slot per workgroup (= step), summed by the caller in slot order
  }
}

// merge the per-panel partials of one (step, row): lse = M + log Σ_p s_p·e^{m_p − M}, nce_sum += Σ_p d_p − lse
__global__ __launch_bounds__(256) void cpc_combine_kernel(CpcParams p) {
  const long long n = (long long)p.T * p.B;
  float local = 0.f;
  for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < n; r += (long long)gridDim.x * 256) {
    const float* w = p.ws + r * p.n_panels * 3;
    float M = -INFINITY, d = 0.f;
    for (int q = 0; q < p.n_panels; ++q) M = fmaxf(M, w[3 * q]);
    float S = 0.f;
    for (int q = 0; q < p.n_panels; ++q) { S += w[3 * q + 1] * expf(w[3 * q] - M); d += w[3 * q + 2]; }
    const float lse = M + logf(S);
    p.lse[r] = lse;
    local += d - lse;
  }
  local = wave_sum_all(local);
  if ((threadIdx.x & 63) == 0) p.nce_sum[blockIdx.x * 4 + (threadIdx.x >> 6)] = local;
}

// Backward on the matrix cores, one workgroup per step i (pred_i staged once), looping over 32-row blocks:
//   (1) total[32][Bc] as in the forward; dt = (exp(total − lse) − [j == b + col_off])·gout/(B·T) in place in the
//       accumulators (lane = column j, register s = row k(s,h) = (s&3) + 8(s>>2) + 4h).
//   (2) dpred[j][c] += Σ_b dt[b][j]·enc[b][c]:  the dt accumulators ARE the A operand (lane = output row j, and the
//       k index is a dummy: k-step s pairs row k(s,0) on the low half with row k(s,1) on the high half, the B operand
//       enc[k(s,h)][c] is read to match) — no lane movement, no LDS.  Accumulated over the row blocks in registers.
//   (3) denc[b][c] = Σ_j dt[b][j]·pred[j][c] sums over dt's COLUMN index: dt goes through LDS once ([32][Bc|1]) and
//       comes back as the A operand (lane = row b); the four waves split K = Bc and their partial tiles are summed
//       through LDS.
// Channels are processed in tiles of 32 (C = 50 → 2), any C; Bc <= 256.
#define CPC_CT_MAX 8     // channel tiles of 32: C <= 256

__global__ __launch_bounds__(256) void cpc_bwd_kernel(CpcParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if (p.t0_dev) {
    p.enc += (long long)p.t0_dev[0] * p.s_i;
    p.denc += (long long)p.t0_dev[0] * p.s_i;
  }
  const int i = blockIdx.x;
  const int col0 = blockIdx.y * CPC_PANEL;         // this workgroup's column panel
  const int B = p.B, Bc = min(CPC_PANEL, p.Bc - col0), C = p.C, PS = C | 1, DS = Bc | 1;
  const int nct = (C + 31) / 32, ntiles = (Bc + 31) / 32;
  float* predl = lds;                              // [Bc][PS]
  float* encl = predl + (size_t)Bc * PS;           // [32][PS]
  float* dtl = encl + 32 * PS;                     // [32][DS]
  float* part = dtl + 32 * DS;                     // [4 waves][32][32]  partial denc tiles of one channel tile
  float* lsel = part + 4 * 32 * 32;                // [32]
  const float* predg = p.pred + ((long long)i * p.Bc + col0) * C;
  for (int idx = threadIdx.x; idx < Bc * C; idx += 256) {
    const int j = idx / C, c = idx - j * C;
    predl[j * PS + c] = predg[idx];
  }
  const float gs = p.gout[0] / ((float)B * (float)p.T);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;

  for (int ct0 = 0; ct0 < nct; ct0 += 2) {         // dpred channel tiles kept in registers: two at a time
    f32x16 dP[CPC_MAXTILES][2];
#pragma unroll
    for (int t = 0; t < CPC_MAXTILES; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) dP[t][u][r] = 0.f;

    for (int r0 = 0; r0 < B; r0 += 32) {
      __syncthreads();                             // previous block's readers of encl / dtl / part / lsel are done
      for (int idx = threadIdx.x; idx < 32 * C; idx += 256) {
        const int r = idx / C, c = idx - r * C;
        const int b = r0 + r;
        encl[r * PS + c] = b < B ? p.enc[i * p.s_i + b * p.s_b + c * p.s_c] : 0.f;
      }
      if (threadIdx.x < 32) lsel[threadIdx.x] = r0 + threadIdx.x < B ? p.lse[(long long)i * B + r0 + threadIdx.x] : 0.f;
      __syncthreads();
      // (1) total tiles -> dt
      f32x16 acc[CPC_MAXTILES];
#pragma unroll
      for (int t = 0; t < CPC_MAXTILES; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
      const float* arow = encl + l31 * PS + half;
      for (int s = 0; 2 * s < C; ++s) {
        const bool k_ok = 2 * s + half < C;
        const float a = k_ok ? arow[2 * s] : 0.f;
#pragma unroll
        for (int t = 0; t < CPC_MAXTILES; ++t) {
          const int j = (wave + 4 * t) * 32 + l31;
          const float bv = (k_ok && j < Bc) ? predl[j * PS + 2 * s + half] : 0.f;
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[t], 0, 0, 0);
        }
      }
#pragma unroll
      for (int t = 0; t < CPC_MAXTILES; ++t) {
        const int j = (wave + 4 * t) * 32 + l31;
        const bool col_ok = wave + 4 * t < ntiles && j < Bc;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
          const int b = r0 + row;
          float v = 0.f;
          if (col_ok && b < B) v = (expf(acc[t][r] - lsel[row]) - (col0 + j == b + p.col_off ? 1.f : 0.f)) * gs;
          acc[t][r] = v;
          if (ct0 == 0 && col_ok) dtl[row * DS + j] = v;            // for (3), first channel pass only
        }
      }
      // (2) dpred tiles += dt^T · enc   (A = the dt accumulators, k-step s <-> rows k(s,0), k(s,1))
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int c = (ct0 + u) * 32 + l31;
        if (ct0 + u < nct) {
#pragma unroll
          for (int s = 0; s < 16; ++s) {
            const int k = (s & 3) + 8 * (s >> 2) + 4 * half;
            const float bv = c < C ? encl[k * PS + c] : 0.f;
#pragma unroll
            for (int t = 0; t < CPC_MAXTILES; ++t) dP[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[t][s], bv, dP[t][u], 0, 0, 0);
          }
        }
      }
      // (3) denc = dt · pred, once per row block (first channel pass), channel tile by channel tile
      if (ct0 == 0) {
        __syncthreads();                           // dtl complete
        for (int ct = 0; ct < nct; ++ct) {
          f32x16 d;
#pragma unroll
          for (int r = 0; r < 16; ++r) d[r] = 0.f;
          const int c = ct * 32 + l31;
          const int kq = (Bc + 3) / 4;             // this wave's quarter of K = Bc (rounded to even below)
          const int k_lo = (wave * kq) & ~1, k_hi = min(Bc, ((wave + 1) * kq) & ~1) ;
          const int k_end = wave == 3 ? Bc : k_hi;
          for (int k = k_lo; k < k_end; k += 2) {
            const int kk = k + half;
            const float a = kk < k_end ? dtl[l31 * DS + kk] : 0.f;
            const float bv = (kk < k_end && c < C) ? predl[kk * PS + c] : 0.f;
            d = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, d, 0, 0, 0);
          }
          __syncthreads();                         // previous channel tile's partials have been consumed
#pragma unroll
          for (int r = 0; r < 16; ++r) part[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = d[r];
          __syncthreads();
          for (int o = threadIdx.x; o < 32 * 32; o += 256) {
            const int row = o >> 5, cc = ct * 32 + (o & 31);
            const int b = r0 + row;
            if (b < B && cc < C) {
              const float v = (part[o] + part[1024 + o]) + (part[2048 + o] + part[3072 + o]);
              float* dst = p.denc + (i * p.s_i + b * p.s_b + cc * p.s_c);
              if (p.n_panels == 1) *dst = v; else atomicAdd(dst, v);      // panels: the caller zero-fills denc
            }
          }
        }
      }
    }
    // write this pass's dpred channel tiles: acc layout row = j (lane... transposed view): dP[t][u] register r holds
    // output row (r&3)+8(r>>2)+4h of tile t  =  column index j_local, and lane l31 is the channel
#pragma unroll
    for (int t = 0; t < CPC_MAXTILES; ++t) {
      if (wave + 4 * t >= ntiles) continue;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int c = (ct0 + u) * 32 + l31;
        if (ct0 + u >= nct || c >= C) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = (wave + 4 * t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (j < Bc) p.dpred[((long long)i * p.Bc + col0 + j) * C + c] = dP[t][u][r];
        }
      }
    }
  }
}

static int cpc_check(const CpcParams& p, size_t lds_bytes, const char* who) {
  FST_REQUIRE(p.enc && p.pred && p.lse && p.T > 0 && p.B > 0 && p.C > 0, "%s: bad arguments", who);
  FST_REQUIRE(p.Bc >= p.B && p.col_off >= 0 && p.col_off + p.B <= p.Bc, "%s: %d rows at column offset %d do not fit %d columns",
              who, p.B, p.col_off, p.Bc);
  FST_REQUIRE(lds_bytes <= 160 * 1024, "%s: pred tile %zu B exceeds LDS (B=%d C=%d)", who, lds_bytes, p.B, p.C);
  return 0;
}

// the split-bf16 single-panel forward (cpc_enc_gather_kernel + cpc_gram_bf3_kernel) serves these shapes
static inline bool cpc_gram_bf3_ok(int T, int B, int C, int Bc) {
  static const bool off = getenv("FST_CPC_GRAM") && atoi(getenv("FST_CPC_GRAM")) == 0;     // diagnostics / FST_MATH=f32: the exact-f32 kernel
  static const bool f32 = getenv("FST_MATH") && !strcmp(getenv("FST_MATH"), "f32");
  return !off && !f32 && T > 0 && B > 0 && Bc <= CPC_PANEL && B <= 256 && C > 0 && C <= 64;
}

extern "C" int64_t fst_cpc_workspace_floats(int T, int B, int C, int Bc) {
  const int np = (Bc + CPC_PANEL - 1) / CPC_PANEL;
  if (np > 1) return (int64_t)T * B * np * 3;          // per-panel softmax statistics
  return cpc_gram_bf3_ok(T, B, C, Bc) ? (int64_t)T * B * C : 0;   // enc_t [T][B][C]
}

// launch geometry of the forward, shared by fst_cpc_nce_slots: pred_i is staged once per workgroup; ~512 workgroups: the row blocks
// of a step are split only as far as that needs
static inline int cpc_fwd_ysplit(int T, int B) {
  const int row_blocks = (B + 31) / 32;
  int ysplit = (512 + T - 1) / T;
  if (ysplit > row_blocks) ysplit = row_blocks;
  return ysplit < 1 ? 1 : ysplit;
}
static inline long long cpc_combine_blocks(int T, int B) {
  long long blocks = ((long long)T * B + 255) / 256;
  return blocks > 1024 ? 1024 : blocks;
}

extern "C" int64_t fst_cpc_nce_slots(int T, int B, int C, int Bc) {
  if (T <= 0 || B <= 0 || Bc <= 0) return -1;
  const int np = (Bc + CPC_PANEL - 1) / CPC_PANEL;
  if (np == 1 && cpc_gram_bf3_ok(T, B, C, Bc)) return T;
  return np > 1 ? 4 * cpc_combine_blocks(T, B) : (int64_t)T * cpc_fwd_ysplit(T, B);
}

extern "C" int fst_cpc_nce_fwd(const float* enc, int64_t s_i, int64_t s_b, int64_t s_c, const int32_t* t0_dev,
                               const float* pred, int T, int B, int C, int Bc, int col_off, float* lse, float* nce_sum,
                               float* ws, void* stream) {
  CpcParams p = {};
  p.t0_dev = t0_dev;
  p.enc = enc; p.s_i = s_i; p.s_b = s_b; p.s_c = s_c; p.pred = pred; p.lse = lse; p.nce_sum = nce_sum;
  p.T = T; p.B = B; p.C = C; p.Bc = Bc; p.col_off = col_off;
  const int Bp = Bc < CPC_PANEL ? Bc : CPC_PANEL;
  p.n_panels = (Bc + CPC_PANEL - 1) / CPC_PANEL;
  p.ws = ws;
  const size_t lds_bytes = ((size_t)Bp * (C | 1) + 32 * (C | 1) + 3 * 4 * 32) * sizeof(float);
  if (int rc = cpc_check(p, lds_bytes, "fst_cpc_nce_fwd")) return rc;
  FST_REQUIRE(nce_sum, "fst_cpc_nce_fwd: nce_sum is null");
  FST_REQUIRE(p.n_panels == 1 || ws, "fst_cpc_nce_fwd: %d columns = %d panels need the workspace (fst_cpc_workspace_floats)", Bc,
              p.n_panels);
  if (p.n_panels == 1 && cpc_gram_bf3_ok(T, B, C, Bc)) {
    FST_REQUIRE(ws, "fst_cpc_nce_fwd: the split-bf16 forward needs the workspace (fst_cpc_workspace_floats: enc_t [T][B][C])");
    const size_t gram_lds = 4 * (size_t)CG_IMG + 3 * 8 * 32 * sizeof(float);
    if (int rc = fst_allow_full_lds((const void*)cpc_gram_bf3_kernel, "fst_cpc_nce_fwd")) return rc;
    hipLaunchKernelGGL(cpc_enc_gather_kernel, dim3((unsigned)((B * C + 63) / 64), (unsigned)((T + 31) / 32)), dim3(256), 0,
                       (hipStream_t)stream, p, ws);
    FST_LAUNCH_CHECK();
    hipLaunchKernelGGL(cpc_gram_bf3_kernel, dim3((unsigned)T), dim3(512), gram_lds, (hipStream_t)stream, p, (const float*)ws);
    FST_LAUNCH_CHECK();
    return 0;
  }
  if (lds_bytes > 48 * 1024)
    if (int rc = fst_allow_full_lds((const void*)cpc_fwd_kernel, "fst_cpc_nce_fwd")) return rc;
  const int ysplit = cpc_fwd_ysplit(T, B);
  hipLaunchKernelGGL(cpc_fwd_kernel, dim3(T, ysplit, p.n_panels), dim3(256), lds_bytes, (hipStream_t)stream, p);
  FST_LAUNCH_CHECK();
  if (p.n_panels > 1) {
    const long long blocks = cpc_combine_blocks(T, B);
    hipLaunchKernelGGL(cpc_combine_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    FST_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int fst_cpc_nce_bwd(const float* enc, int64_t s_i, int64_t s_b, int64_t s_c, const int32_t* t0_dev,
                               const float* pred, const float* lse, int T, int B, int C, int Bc, int col_off,
                               const float* gout, float* denc, float* dpred, void* stream) {
  CpcParams p = {};
  p.t0_dev = t0_dev;
  p.enc = enc; p.s_i = s_i; p.s_b = s_b; p.s_c = s_c; p.pred = pred; p.lse = const_cast<float*>(lse);
  p.gout = gout; p.denc = denc; p.dpred = dpred; p.T = T; p.B = B; p.C = C; p.Bc = Bc; p.col_off = col_off;
  const int Bp = Bc < CPC_PANEL ? Bc : CPC_PANEL;
  p.n_panels = (Bc + CPC_PANEL - 1) / CPC_PANEL;
  const size_t lds_bytes = ((size_t)Bp * (C | 1) + 32 * (C | 1) + 32 * (size_t)(Bp | 1) + 4 * 32 * 32 + 32) * sizeof(float);
  if (int rc = cpc_check(p, lds_bytes, "fst_cpc_nce_bwd")) return rc;
  FST_REQUIRE(gout && denc && dpred, "fst_cpc_nce_bwd: null gradient buffer");
  FST_REQUIRE(C <= 32 * CPC_CT_MAX, "fst_cpc_nce_bwd: C=%d > %d not supported yet", C, 32 * CPC_CT_MAX);
  if (lds_bytes > 48 * 1024)
    if (int rc = fst_allow_full_lds((const void*)cpc_bwd_kernel, "fst_cpc_nce_bwd")) return rc;
  hipLaunchKernelGGL(cpc_bwd_kernel, dim3(T, p.n_panels), dim3(256), lds_bytes, (hipStream_t)stream, p);
  FST_LAUNCH_CHECK();
  return 0;
}
