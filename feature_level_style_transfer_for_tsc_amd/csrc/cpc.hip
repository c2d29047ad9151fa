// CPC InfoNCE cross-Gram (Comparison/SLARDA/train.py:72-76), fused: for each prediction step i the
// B×B matrix total_i = enc_i · pred_iᵀ is never materialised — a workgroup keeps pred_i in LDS
// ([B][C+1], odd stride → conflict-free lane↔column reads), forms 16 rows of the Gram product at a
// time, and reduces each row's log-softmax with wave64 shuffles.  K = C (50) is tiny, so this is a
// VALU/LDS kernel; the whole loss is ≈1.7 GFLOP per call at B=256, T=256.
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

__global__ __launch_bounds__(256) void cpc_fwd_kernel(CpcParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if (p.t0_dev) p.enc += (long long)p.t0_dev[0] * p.s_i;
  const int i = blockIdx.x, r0 = blockIdx.y * CPC_ROWS;
  const int B = p.B, Bc = p.Bc, C = p.C, PS = C | 1;
  float* predl = lds;                  // [Bc][PS]
  float* encl = lds + (size_t)Bc * PS; // [CPC_ROWS][C]
  const float* predg = p.pred + (long long)i * Bc * C;
  for (int idx = threadIdx.x; idx < Bc * C; idx += 256) {
    const int j = idx / C, c = idx - j * C;
    predl[j * PS + c] = predg[idx];
  }
  for (int idx = threadIdx.x; idx < CPC_ROWS * C; idx += 256) {
    const int r = idx / C, c = idx - r * C;
    const int b = r0 + r;
    encl[idx] = b < B ? p.enc[i * p.s_i + b * p.s_b + c * p.s_c] : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float local = 0.f;
  for (int r = wave; r < CPC_ROWS; r += 4) {
    const int b = r0 + r;
    if (b >= B) break;
    const float* er = encl + r * C;
    float mx = -INFINITY, diag = 0.f;
    // pass 1: max
    for (int j = lane; j < Bc; j += 64) {
      const float* pj = predl + j * PS;
      float d = 0.f;
      for (int c = 0; c < C; ++c) d = fmaf(er[c], pj[c], d);
      mx = fmaxf(mx, d);
      if (j == b + p.col_off) diag = d;
    }
    mx = wave_max(mx);
    float se = 0.f;
    for (int j = lane; j < Bc; j += 64) {
      const float* pj = predl + j * PS;
      float d = 0.f;
      for (int c = 0; c < C; ++c) d = fmaf(er[c], pj[c], d);
      se += expf(d - mx);
    }
    se = wave_sum_all(se);
    diag = wave_sum_all(diag);
    const float lse = mx + logf(se);
    if (lane == 0) {
      p.lse[(long long)i * B + b] = lse;
      local += diag - lse;
    }
  }
  if (lane == 0 && local != 0.f) atomicAdd(p.nce_sum, local);
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
          if (c < C) dp[c] = fmaf(d, er[c], dp[c]);
      }
    }
  }
  if (jcol < Bc) {
    float* out = p.dpred + ((long long)i * Bc + jcol) * C;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < C) out[c] = dp[c];
  }
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
  const size_t lds_bytes = ((size_t)Bc * (C | 1) + CPC_ROWS * C) * sizeof(float);
  if (int rc = cpc_check(p, lds_bytes, "fst_cpc_nce_fwd")) return rc;
  FST_REQUIRE(nce_sum, "fst_cpc_nce_fwd: nce_sum is null");
  if (lds_bytes > 48 * 1024)
    if (int rc = fst_allow_full_lds((const void*)cpc_fwd_kernel, "fst_cpc_nce_fwd")) return rc;
  hipLaunchKernelGGL(cpc_fwd_kernel, dim3(T, (B + CPC_ROWS - 1) / CPC_ROWS), dim3(256), lds_bytes, (hipStream_t)stream, p);
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
  FST_REQUIRE(C <= 128, "fst_cpc_nce_bwd: C=%d > 128 not supported yet", C);
  void (*fn)(CpcParams) = C <= 32 ? cpc_bwd_kernel<32> : (C <= 64 ? cpc_bwd_kernel<64> : cpc_bwd_kernel<128>);
  if (lds_bytes > 48 * 1024)
    if (int rc = fst_allow_full_lds((const void*)fn, "fst_cpc_nce_bwd")) return rc;
  hipLaunchKernelGGL(fn, dim3(T), dim3(256), lds_bytes, (hipStream_t)stream, p);
  FST_LAUNCH_CHECK();
  return 0;
}
