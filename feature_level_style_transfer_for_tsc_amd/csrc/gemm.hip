// Dense products of the small heads on the matrix cores (gfx950): DimensionUnification's length GEMM (widgets.py:73-75 of the
// reference: [B·C_s, L_s] × [L_t, L_s]ᵀ, 6.7 GFLOP at the metric shape, six times per step with its gradients), the Linear chains of
// the CDAN discriminator (widgets.py:113-131), of FeatureDiscriminatorforSource (widgets.py:32-42) and the classifier heads.
//
//     C[m][n] = act( Σ_k A(m,k)·B(n,k) + bias[n] )
//
// with each operand stored either reduction-index-contiguous ([rows][K]) or reduction-index-major ([K][rows]), which covers the three
// products of a Linear layer without a transposed copy:   y = x·Wᵀ (A = x [M][K], B = W [N][K]),   dx = g·W (A = g [M][N], B = W read
// [K'=N][rows=K]),   dW = gᵀ·x (A = g read [K'=M][rows=N], B = x read [K'=M][rows=K]).
//
// Arithmetic: split-bf16 (hi·hi + hi·lo + lo·hi on v_mfma_f32_32x32x16_bf16, fp32 accumulate) like every other product of the
// step.  A 256-thread workgroup (2 × 2 waves) owns a 128 × 128 or 64 × 64 tile of C; a stage = 32 k: every thread fetches its
// share of the two operand tiles into registers one stage ahead (16-byte loads along k where the layout allows, coalesced dword
// loads across rows for k-major operands), splits fp32 → bf16 hi / lo ONCE per element and writes rows of 32 bf16 (80-byte pitch:
// the 8-byte / 16-byte staging writes and the ds_read_b128 fragment reads are bank-conflict-free), so the multiply loop is
// 16 ds_read_b128 per 24 MFMAs and no conversion.  40 KiB of LDS per workgroup: three workgroups per CU hide each other's barriers.
// K is split over gridDim.z when the tiles alone do not fill the chip: partial tiles go to slabs that a second kernel adds in slab
// order (+ bias, activation) — deterministic, no atomics, no zero fill.
#include <stdlib.h>

#include "fst_common.h"

typedef __bf16 gm_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gm_bf16x2 __attribute__((ext_vector_type(2)));
typedef float gm_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned gm_u32x4 __attribute__((ext_vector_type(4)));

#define GM_BK 32
#define GM_PITCH 80

struct GemmParams {
  const float* A;
  const float* B;
  float* C;
  long long lda, ldb, ldc;
  int M, N, K;
  int ta, tb;              // operand stored [K][rows] (reduction index major) instead of [rows][K]
  int vec_a, vec_b;        // [rows][K] operands: 16-byte loads along k (16-byte aligned rows, K % 4 == 0)
  const float* bias;       // [N] or null
  int act;                 // FST_ACT_NONE / _RELU / _LEAKY
  float slope;
  int ksplit, k_per_split; // split z covers k in [z·k_per_split, min(K, (z+1)·k_per_split)), k_per_split % 32 == 0
  float* slab;             // [ksplit][M][N] when ksplit > 1
  int tiles_n;
};

__device__ __forceinline__ void gm_split_pair(float a, float b, unsigned& hi, unsigned& lo) {
  const gm_f32x2 v = {a, b};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, gm_bf16x2));
  const gm_f32x2 r = {a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u)};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, gm_bf16x2));
}

__device__ __forceinline__ float gm_act(float v, int act, float slope) {
  if (act == FST_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == FST_ACT_LEAKY) return v > 0.f ? v : slope * v;
  return v;
}

// One operand tile of ROWS rows × 32 k, ROWS/8 floats per thread.
//   [rows][K] storage: thread t holds k = 4(t&7) .. +3 of rows (t>>3) + 32j
//   [K][rows] storage: thread t holds row t % ROWS, k = KR·(t / ROWS) .. + KR − 1   (KR = ROWS/8 consecutive k)
template <int ROWS>
__device__ __forceinline__ void gm_fetch(float (&v)[ROWS / 8], const float* X, long long ld, int trans, int vec, int row0, int rows,
                                         int k0, int k_end, int tid) {
  constexpr int KR = ROWS / 8;
  if (!trans) {
    const int kk = k0 + 4 * (tid & 7), r0 = row0 + (tid >> 3);
#pragma unroll
    for (int j = 0; j < ROWS / 32; ++j) {
      const int r = r0 + 32 * j;
      const float* q = X + (long long)r * ld + kk;
      if (vec) {
        float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < rows && kk < k_end) w = *reinterpret_cast<const float4*>(q);
        v[4 * j] = w.x; v[4 * j + 1] = w.y; v[4 * j + 2] = w.z; v[4 * j + 3] = w.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * j + e] = (r < rows && kk + e < k_end) ? q[e] : 0.f;
      }
    }
  } else {
    const int i = row0 + tid % ROWS, kk = k0 + KR * (tid / ROWS);
    const float* q = X + (long long)kk * ld + i;
#pragma unroll
    for (int e = 0; e < KR; ++e) v[e] = (i < rows && kk + e < k_end) ? q[(long long)e * ld] : 0.f;
  }
}

template <int ROWS>
__device__ __forceinline__ void gm_stage(const float (&v)[ROWS / 8], char* hi_img, char* lo_img, int trans, int tid) {
  constexpr int KR = ROWS / 8;
  if (!trans) {
    const int kb = 8 * (tid & 7), r0 = tid >> 3;           // byte offset of k = 4(t&7) in a row of bf16
#pragma unroll
    for (int j = 0; j < ROWS / 32; ++j) {
      unsigned h0, l0, h1, l1;
      gm_split_pair(v[4 * j], v[4 * j + 1], h0, l0);
      gm_split_pair(v[4 * j + 2], v[4 * j + 3], h1, l1);
      const int off = (r0 + 32 * j) * GM_PITCH + kb;
      *reinterpret_cast<uint2*>(hi_img + off) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(lo_img + off) = make_uint2(l0, l1);
    }
  } else {
    const int off = (tid % ROWS) * GM_PITCH + 2 * KR * (tid / ROWS);
#pragma unroll
    for (int c = 0; c < KR / 8; ++c) {
      gm_u32x4 h, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        unsigned hh, ll;
        gm_split_pair(v[8 * c + 2 * e], v[8 * c + 2 * e + 1], hh, ll);
        h[e] = hh; l[e] = ll;
      }
      *reinterpret_cast<gm_u32x4*>(hi_img + off + 16 * c) = h;
      *reinterpret_cast<gm_u32x4*>(lo_img + off + 16 * c) = l;
    }
  }
}

template <int TM, int TN>
__global__ __launch_bounds__(256, TM * TN == 4 ? 2 : 3) void gemm_bf3_kernel(GemmParams p) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  __shared__ __attribute__((aligned(16))) char lds[2 * (BM + BN) * GM_PITCH];
  char* const a_hi = lds;
  char* const a_lo = a_hi + BM * GM_PITCH;
  char* const b_hi = a_lo + BM * GM_PITCH;
  char* const b_lo = b_hi + BN * GM_PITCH;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
  const int tile_m = blockIdx.x / p.tiles_n, tile_n = blockIdx.x - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int k_begin = blockIdx.z * p.k_per_split;
  const int k_end = min(p.K, k_begin + p.k_per_split);
  const int n_st = (k_end - k_begin + GM_BK - 1) / GM_BK;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float va[BM / 8], vb[BN / 8];
  gm_fetch<BM>(va, p.A, p.lda, p.ta, p.vec_a, m0, p.M, k_begin, k_end, tid);
  gm_fetch<BN>(vb, p.B, p.ldb, p.tb, p.vec_b, n0, p.N, k_begin, k_end, tid);
  const int a_off = (wm * 32 * TM + l31) * GM_PITCH + 16 * half;
  const int b_off = (wn * 32 * TN + l31) * GM_PITCH + 16 * half;
  for (int s = 0; s < n_st; ++s) {
    __syncthreads();                                       // every wave is past its fragment reads of the previous stage
    gm_stage<BM>(va, a_hi, a_lo, p.ta, tid);
    gm_stage<BN>(vb, b_hi, b_lo, p.tb, tid);
    __syncthreads();
    if (s + 1 < n_st) {                                    // the next stage's operands travel under this stage's MFMAs
      const int k0 = k_begin + (s + 1) * GM_BK;
      gm_fetch<BM>(va, p.A, p.lda, p.ta, p.vec_a, m0, p.M, k0, k_end, tid);
      gm_fetch<BN>(vb, p.B, p.ldb, p.tb, p.vec_b, n0, p.N, k0, k_end, tid);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      gm_bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = *reinterpret_cast<const gm_bf16x8*>(a_hi + a_off + i * 32 * GM_PITCH + 32 * ks);
        al[i] = *reinterpret_cast<const gm_bf16x8*>(a_lo + a_off + i * 32 * GM_PITCH + 32 * ks);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = *reinterpret_cast<const gm_bf16x8*>(b_hi + b_off + j * 32 * GM_PITCH + 32 * ks);
        bl[j] = *reinterpret_cast<const gm_bf16x8*>(b_lo + b_off + j * 32 * GM_PITCH + 32 * ks);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
  }

  // accumulator layout: lane = column n, register r = row (r&3) + 8(r>>2) + 4·half — one store instruction = 2 rows × 128 bytes
  const bool direct = p.ksplit == 1;
  float* const out = direct ? p.C : p.slab + (long long)blockIdx.z * p.M * p.N;
  const long long ldo = direct ? p.ldc : p.N;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * 32 * TN + 32 * j + l31;
    if (n >= p.N) continue;
    const float bv = (direct && p.bias) ? p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int mr = m0 + wm * 32 * TM + 32 * i + 4 * half;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mr + (r & 3) + 8 * (r >> 2);
        if (m < p.M) {
          const float v = acc[i][j][r] + bv;
          out[(long long)m * ldo + n] = direct ? gm_act(v, p.act, p.slope) : v;
        }
      }
    }
  }
}

// C[m][n] = act(Σ_z slab[z][m][n] + bias[n]), slabs added in z order (four independent chains, combined in a fixed tree)
__global__ __launch_bounds__(256) void gemm_reduce_kernel(GemmParams p) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long MN = (long long)p.M * p.N;
  if (idx >= MN) return;
  const int m = (int)(idx / p.N), n = (int)(idx - (long long)m * p.N);
  const float* q = p.slab + idx;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int z = 0;
  for (; z + 4 <= p.ksplit; z += 4) {
    s0 += q[(long long)z * MN]; s1 += q[(long long)(z + 1) * MN]; s2 += q[(long long)(z + 2) * MN]; s3 += q[(long long)(z + 3) * MN];
  }
  for (; z < p.ksplit; ++z) s0 += q[(long long)z * MN];
  const float v = ((s0 + s1) + (s2 + s3)) + (p.bias ? p.bias[n] : 0.f);
  p.C[(long long)m * p.ldc + n] = gm_act(v, p.act, p.slope);
}

// tile size and K split for a shape: the same answer in fst_gemm_workspace_floats and fst_gemm
static void gm_geometry(int M, int N, int K, int* big, int* ksplit, int* k_per_split) {
  const int cus = fst_cu_count() > 0 ? fst_cu_count() : 256;
  const long long t128 = (long long)((M + 127) / 128) * ((N + 127) / 128);
  const int kmax = K / 128 > 1 ? K / 128 : 1;              // at least four stages per split
  *big = M >= 96 && N >= 96 && t128 * kmax >= cus / 2;
  const long long tiles = *big ? t128 : (long long)((M + 63) / 64) * ((N + 63) / 64);
  long long want = cus / tiles;
  if (want < 1) want = 1;
  if (want > kmax) want = kmax;
  int kps = (int)((K + want - 1) / want);
  kps = (kps + GM_BK - 1) / GM_BK * GM_BK;
  *k_per_split = kps;
  *ksplit = (K + kps - 1) / kps;
}

extern "C" int64_t fst_gemm_workspace_floats(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return -1;
  int big, ks, kps;
  gm_geometry(M, N, K, &big, &ks, &kps);
  return ks > 1 ? (int64_t)ks * M * N : 0;
}

extern "C" int fst_gemm(const float* A, int64_t lda, int ta, const float* B, int64_t ldb, int tb, float* C, int64_t ldc, int M, int N,
                        int K, const float* bias, int act, float slope, float* workspace, int64_t workspace_floats, void* stream) {
  FST_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, "fst_gemm: bad arguments (M=%d N=%d K=%d)", M, N, K);
  FST_REQUIRE(lda >= (ta ? M : K) && ldb >= (tb ? N : K) && ldc >= N, "fst_gemm: leading dimensions lda=%lld ldb=%lld ldc=%lld are smaller "
              "than a row (M=%d N=%d K=%d ta=%d tb=%d)", (long long)lda, (long long)ldb, (long long)ldc, M, N, K, ta, tb);
  FST_REQUIRE(act == FST_ACT_NONE || act == FST_ACT_RELU || act == FST_ACT_LEAKY, "fst_gemm: unknown activation %d", act);
  GemmParams p = {};
  int big;
  gm_geometry(M, N, K, &big, &p.ksplit, &p.k_per_split);
  FST_REQUIRE(p.ksplit == 1 || (workspace && workspace_floats >= (int64_t)p.ksplit * M * N), "fst_gemm: workspace of %lld floats is too small "
              "(fst_gemm_workspace_floats(%d, %d, %d) = %lld)", (long long)workspace_floats, M, N, K, (long long)p.ksplit * M * N);
  auto vec = [](const float* q, int64_t ld, int K_) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0 && ld % 4 == 0 && K_ % 4 == 0; };
  p.A = A; p.B = B; p.C = C; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
  p.ta = ta ? 1 : 0; p.tb = tb ? 1 : 0;
  p.vec_a = !ta && vec(A, lda, K); p.vec_b = !tb && vec(B, ldb, K);
  p.bias = bias; p.act = act; p.slope = slope; p.slab = workspace;
  const int bt = big ? 128 : 64;
  p.tiles_n = (N + bt - 1) / bt;
  const unsigned tiles = (unsigned)(((M + bt - 1) / bt) * p.tiles_n);
  if (big) hipLaunchKernelGGL((gemm_bf3_kernel<2, 2>), dim3(tiles, 1, (unsigned)p.ksplit), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((gemm_bf3_kernel<1, 1>), dim3(tiles, 1, (unsigned)p.ksplit), dim3(256), 0, (hipStream_t)stream, p);
  FST_LAUNCH_CHECK();
  if (p.ksplit > 1) {
    hipLaunchKernelGGL(gemm_reduce_kernel, dim3((unsigned)(((long long)M * N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    FST_LAUNCH_CHECK();
  }
  return 0;
}

// out = dy·act'(y) from the activation's OUTPUT (y > 0 ⇔ pre-activation > 0 for ReLU and for LeakyReLU with a positive slope)
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* dy, const float* y, float* out, long long n, float slope) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    const float4 g = *reinterpret_cast<const float4*>(dy + i), v = *reinterpret_cast<const float4*>(y + i);
    float4 o;
    o.x = v.x > 0.f ? g.x : slope * g.x; o.y = v.y > 0.f ? g.y : slope * g.y;
    o.z = v.z > 0.f ? g.z : slope * g.z; o.w = v.w > 0.f ? g.w : slope * g.w;
    *reinterpret_cast<float4*>(out + i) = o;
  } else {
    for (long long j = i; j < n; ++j) out[j] = y[j] > 0.f ? dy[j] : slope * dy[j];
  }
}

extern "C" int fst_act_bwd(const float* dy, const float* y, float* out, int64_t n, float slope, void* stream) {
  FST_REQUIRE(dy && y && out && n > 0, "fst_act_bwd: bad arguments");
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  FST_REQUIRE(al16(dy) && al16(y) && al16(out), "fst_act_bwd: tensors must be 16-byte aligned");
  hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, dy, y, out, (long long)n, slope);
  FST_LAUNCH_CHECK();
  return 0;
}
