// CPC InfoNCE cross-Gram (Comparison/SLARDA/train.py:72-76), fused: for each prediction step i the
// B×B matrix total_i = enc_i · pred_iᵀ is never materialised.  Forward: the Gram product runs on the
// matrix cores (exact f32 MFMA) with the log-softmax and the diagonal taken on the accumulators.
// Backward: a workgroup keeps pred_i in LDS ([B][C+1], odd stride → conflict-free lane↔column reads),
// re-forms 16 rows of the product at a time on the VALU and contracts them with pred / enc.
// The whole loss is ≈1.7 GFLOP per call at B=256, T=256.
//
// enc is read in place from the feature tensor through (s_i, s_b, s_c) element strides, so no
// [T,B,C] gather copy is made; denc is written through the same strides.
//
// Rows (enc samples, B) and columns (pred samples = negatives, Bc) may differ: in "global batch" data
// parallelism a rank scores its B local rows against the Bc = world·B gathered predictions, its own
// positives sitting at columns col_off + b (SURVEY §8e mode B).  Single GPU: Bc = B, col_off = 0.
#include "fst_common.h"

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
  float* nce_sum;      // scalar
  const float* gout;   // scalar upstream gradient of nce (device), bwd only
  float* denc;         // same strides as enc
  float* dpred;        // [T][B][C]
  const int* t0_dev;   // optional device scalar: extra time offset of enc/denc (elements of stride s_i)
  int T, B, C;
  int Bc, col_off;     // columns (pred rows) and the column of row 0's positive
};

// Forward: the Gram product on the matrix cores.  A workgroup owns 32 rows (encodings) of step i against all Bc
// columns (predictions): total[32][Bc] = enc[32][C] · predᵀ[C][Bc] as v_mfma_f32_32x32x2_f32 (exact fp32 chain,
// K = C: 25 k-steps at C = 50), wave w taking column blocks w, w+4 (two 32x32 tiles).  Both operands come from LDS
// ([rows][C|1], odd stride: lane <-> row reads hit distinct banks).  The log-softmax over a row runs on the
// accumulators: C/D layout puts a tile's 32 columns on the 32 lanes of a half-wave and 16 rows in its registers, so a
// row's max / Σexp / diagonal is a 5-step butterfly inside the half-wave, then a 4-entry combine through LDS.
#define CPC_MAXTILES 2   // column tiles per wave: Bc <= 4 * 32 * CPC_MAXTILES = 256

__global__ __launch_bounds__(256) void cpc_fwd_kernel(CpcParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if (p.t0_dev) p.enc += (long long)p.t0_dev[0] * p.s_i;
  const int i = blockIdx.x;
  const int B = p.B, Bc = p.Bc, C = p.C, PS = C | 1;
  float* predl = lds;                          // [Bc][PS]   staged once per workgroup
  float* encl = predl + (size_t)Bc * PS;       // [32][PS]   one row block at a time
  float* red = encl + 32 * PS;                 // [3][4 waves][32 rows]: max, Σexp, diagonal
  const float* predg = p.pred + (long long)i * Bc * C;
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
        if (j == r0 + row + p.col_off) d = acc[t][r];
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
        p.lse[(long long)i * B + b] = lse;
        local += d4 - lse;
      }
    }
  }
  }   // row blocks
  if (wave == 0) {
    local += __shfl_xor(local, 32, 64);                    // the two halves hold different rows
    if (lane == 0) atomicAdd(p.nce_sum, local);
  }
}

template <int CMAX>
__global__ __launch_bounds__(256) void cpc_bwd_kernel(CpcParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if (p.t0_dev) {
    p.enc += (long long)p.t0_dev[0] * p.s_i;
    p.denc += (long long)p.t0_dev[0] * p.s_i;
  }
  const int i = blockIdx.x;
  const int B = p.B, Bc = p.Bc, C = p.C, PS = C | 1;
  float* predl = lds;                              // [Bc][PS]
  float* encl = predl + (size_t)Bc * PS;           // [CPC_ROWS][C]
  float* dtl = encl + CPC_ROWS * C;                // [CPC_ROWS][Bc]
  const float* predg = p.pred + (long long)i * Bc * C;
  for (int idx = threadIdx.x; idx < Bc * C; idx += 256) {
    const int j = idx / C, c = idx - j * C;
    predl[j * PS + c] = predg[idx];
  }
  const float gs = p.gout[0] / ((float)B * (float)p.T);   // d nce / d total = (softmax − I)/(B·T)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int jcol = threadIdx.x;                           // this thread's dpred column (Bc ≤ 256)
  // dpred is accumulated in registers, CMAX channels of one column per thread; wider features take several passes
  // over the row blocks (the dt tile is re-formed per pass, denc is written in the first one)
  for (int c0 = 0; c0 < C; c0 += CMAX) {
  float dp[CMAX];
#pragma unroll
  for (int c = 0; c < CMAX; ++c) dp[c] = 0.f;

  for (int r0 = 0; r0 < B; r0 += CPC_ROWS) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < CPC_ROWS * C; idx += 256) {
      const int r = idx / C, c = idx - r * C;
      const int b = r0 + r;
      encl[idx] = b < B ? p.enc[i * p.s_i + b * p.s_b + c * p.s_c] : 0.f;
    }
    __syncthreads();
    for (int r = wave; r < CPC_ROWS; r += 4) {
      const int b = r0 + r;
      const float* er = encl + r * C;
      const float lse = b < B ? p.lse[(long long)i * B + b] : 0.f;
      for (int j = lane; j < Bc; j += 64) {
        float v = 0.f;
        if (b < B) {
          const float* pj = predl + j * PS;
          float d = 0.f;
          for (int c = 0; c < C; ++c) d = fmaf(er[c], pj[c], d);
          v = (expf(d - lse) - (j == b + p.col_off ? 1.f : 0.f)) * gs;
        }
        dtl[r * Bc + j] = v;
      }
    }
    __syncthreads();
    // denc[b][c] = Σ_j dt[b][j]·pred[j][c]
    if (c0 == 0)
    for (int o = threadIdx.x; o < CPC_ROWS * C; o += 256) {
      const int r = o / C, c = o - r * C;
      const int b = r0 + r;
      if (b >= B) continue;
      const float* dr = dtl + r * Bc;
      float s = 0.f;
      for (int j = 0; j < Bc; ++j) s = fmaf(dr[j], predl[j * PS + c], s);
      p.denc[i * p.s_i + b * p.s_b + c * p.s_c] = s;
    }
    // dpred[j][c] += Σ_b dt[b][j]·enc[b][c]
    if (jcol < Bc) {
      for (int r = 0; r < CPC_ROWS; ++r) {
        const float d = dtl[r * Bc + jcol];
        const float* er = encl + r * C;
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
          if (c0 + c < C) dp[c] = fmaf(d, er[c0 + c], dp[c]);
      }
    }
  }
  if (jcol < Bc) {
    float* out = p.dpred + ((long long)i * Bc + jcol) * C + c0;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c0 + c < C) out[c] = dp[c];
  }
  }   // channel chunks
}

static int cpc_check(const CpcParams& p, size_t lds_bytes, const char* who) {
  FST_REQUIRE(p.enc && p.pred && p.lse && p.T > 0 && p.B > 0 && p.C > 0, "%s: bad arguments", who);
  FST_REQUIRE(p.Bc >= p.B && p.col_off >= 0 && p.col_off + p.B <= p.Bc, "%s: %d rows at column offset %d do not fit %d columns",
              who, p.B, p.col_off, p.Bc);
  FST_REQUIRE(lds_bytes <= 160 * 1024, "%s: pred tile %zu B exceeds LDS (B=%d C=%d)", who, lds_bytes, p.B, p.C);
  return 0;
}

extern "C" int fst_cpc_nce_fwd(const float* enc, int64_t s_i, int64_t s_b, int64_t s_c, const int32_t* t0_dev,
                               const float* pred, int T, int B, int C, int Bc, int col_off, float* lse, float* nce_sum,
                               void* stream) {
  CpcParams p = {};
  p.t0_dev = t0_dev;
  p.enc = enc; p.s_i = s_i; p.s_b = s_b; p.s_c = s_c; p.pred = pred; p.lse = lse; p.nce_sum = nce_sum;
  p.T = T; p.B = B; p.C = C; p.Bc = Bc; p.col_off = col_off;
  const size_t lds_bytes = ((size_t)Bc * (C | 1) + 32 * (C | 1) + 3 * 4 * 32) * sizeof(float);
  if (int rc = cpc_check(p, lds_bytes, "fst_cpc_nce_fwd")) return rc;
  FST_REQUIRE(nce_sum, "fst_cpc_nce_fwd: nce_sum is null");
  FST_REQUIRE(Bc <= 128 * CPC_MAXTILES, "fst_cpc_nce_fwd: %d > %d negatives (columns) not supported yet", Bc, 128 * CPC_MAXTILES);
  if (lds_bytes > 48 * 1024)
    if (int rc = fst_allow_full_lds((const void*)cpc_fwd_kernel, "fst_cpc_nce_fwd")) return rc;
  // pred_i is staged once per workgroup; ~512 workgroups: split the row blocks of a step only as far as that needs
  const int row_blocks = (B + 31) / 32;
  int ysplit = (512 + T - 1) / T;
  if (ysplit > row_blocks) ysplit = row_blocks;
  if (ysplit < 1) ysplit = 1;
  hipLaunchKernelGGL(cpc_fwd_kernel, dim3(T, ysplit), dim3(256), lds_bytes, (hipStream_t)stream, p);
  FST_LAUNCH_CHECK();
  return 0;
}

extern "C" int fst_cpc_nce_bwd(const float* enc, int64_t s_i, int64_t s_b, int64_t s_c, const int32_t* t0_dev,
                               const float* pred, const float* lse, int T, int B, int C, int Bc, int col_off,
                               const float* gout, float* denc, float* dpred, void* stream) {
  CpcParams p = {};
  p.t0_dev = t0_dev;
  p.enc = enc; p.s_i = s_i; p.s_b = s_b; p.s_c = s_c; p.pred = pred; p.lse = const_cast<float*>(lse);
  p.gout = gout; p.denc = denc; p.dpred = dpred; p.T = T; p.B = B; p.C = C; p.Bc = Bc; p.col_off = col_off;
  const size_t lds_bytes = ((size_t)Bc * (C | 1) + CPC_ROWS * C + (size_t)CPC_ROWS * Bc) * sizeof(float);
  if (int rc = cpc_check(p, lds_bytes, "fst_cpc_nce_bwd")) return rc;
  FST_REQUIRE(gout && denc && dpred, "fst_cpc_nce_bwd: null gradient buffer");
  FST_REQUIRE(Bc <= 256, "fst_cpc_nce_bwd: %d > 256 negatives (columns) not supported yet", Bc);
  void (*fn)(CpcParams) = C <= 32 ? cpc_bwd_kernel<32> : (C <= 64 ? cpc_bwd_kernel<64> : cpc_bwd_kernel<128>);
  if (lds_bytes > 48 * 1024)
    if (int rc = fst_allow_full_lds((const void*)fn, "fst_cpc_nce_bwd")) return rc;
  hipLaunchKernelGGL(fn, dim3(T), dim3(256), lds_bytes, (hipStream_t)stream, p);
  FST_LAUNCH_CHECK();
  return 0;
}
