// HBM-bound pointwise / per-channel reduction kernels of the hot path (BatchNorm, gate, affine
// coupling, bias-gradient row sums).  One pass per tensor, coalesced along time, wave64 shuffles for
// the per-channel reductions, one atomic per block per channel.
#include "fst_common.h"

static thread_local char g_err[512] = "";
void fst_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* fst_last_error(void) { return g_err; }

int fst_allow_full_lds(const void* fn, const char* who) {
  static const void* done[64];
  static int n_done = 0;
  for (int i = 0; i < n_done; ++i)
    if (done[i] == fn) return 0;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) {
    fst_set_error("%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e));
    return (int)e;
  }
  if (n_done < 64) done[n_done++] = fn;
  return 0;
}
int fst_cu_count(void) {
  static int cached[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
    cached[dev] = n > 0 ? n : -1;
  }
  return cached[dev] > 0 ? cached[dev] : 0;
}
extern "C" int fst_version(void) { return FST_ABI_VERSION; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// block-wide sum of two values; result valid in thread 0
__device__ __forceinline__ void block_sum2(float& a, float& b) {
  __shared__ float sa[4], sb[4];
  a = wave_sum(a);
  b = wave_sum(b);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sa[wave] = a; sb[wave] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    a = sa[0] + sa[1] + sa[2] + sa[3];
    b = sb[0] + sb[1] + sb[2] + sb[3];
  }
  __syncthreads();
}

// Every launcher below sizes its grid from (B, C, L) and walks B*C*L contiguous elements per tensor.  The caller passes
// the element count of the tensors it actually holds (`numel`) separately, so a batch argument that does not describe
// them — e.g. a global (all-rank) batch handed over as the launch batch — is refused on the host instead of walking
// past the end of the allocation (the GPU memory fault of round 1, DESIGN.md "Fault log").
#define FST_REQUIRE_EXTENT(who, B, C, L, numel)                                                             \
  FST_REQUIRE((long long)(B) * (long long)(C) * (long long)(L) == (long long)(numel),                        \
              "%s: B*C*L = %d*%d*%d does not match the tensors' element count %lld", who, (int)(B), (int)(C), (int)(L), \
              (long long)(numel))


// 16-byte path of the per-row kernels (L a multiple of 4, 16-byte aligned bases): a row of L/4 float4 is covered by
// tpr = min(256, pow2ceil(L/4)) threads and a 256-thread block walks 256/tpr rows at a time, so every lane carries a
// 16-byte load whatever L is (the dword path keeps half the block idle at L = 512 and issues 4x the loads).
struct RowVec {
  int shift;   // log2(threads per row)
  int L4;      // float4 per row
};
static inline bool vec_ok(int L, std::initializer_list<const void*> ptrs) {
  if (L % 4 != 0) return false;
  for (const void* q : ptrs)
    if (q && (reinterpret_cast<uintptr_t>(q) & 15)) return false;
  return true;
}
static inline RowVec row_vec(int L) {
  RowVec v;
  v.L4 = L / 4;
  v.shift = 0;
  while ((1 << v.shift) < v.L4 && v.shift < 8) ++v.shift;
  return v;
}

// ---------------------------------------------------------------- row sums (bias gradients)
// out[c] = Σ_{b,t} x[b,c,t]: ONE 1024-thread workgroup per row c, every thread a strided run of 16-byte loads, a fixed
// shuffle / LDS tree — the result is stored (no zero fill in front, no atomics), bit-identical from run to run.
__device__ __forceinline__ float block_sum1024(float s) {
  __shared__ float sw[16];
  s = wave_sum(s);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sw[wave] = s;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 0; w < 16; ++w) r += sw[w];
  }
  return r;
}

__global__ __launch_bounds__(1024) void row_sum_kernel(const float* x, long long x_bs, int B, int C, int L, float* out) {
  const int c = blockIdx.x;
  float s = 0.f;
  for (int b = threadIdx.x >> 6; b < B; b += 16) {           // a wave per batch row
    const float* row = x + (long long)b * x_bs + (long long)c * L;
    for (int t = threadIdx.x & 63; t < L; t += 64) s += row[t];
  }
  s = block_sum1024(s);
  if (threadIdx.x == 0) out[c] = s;
}

__global__ __launch_bounds__(1024) void row_sum_vec_kernel(const float* x, long long x_bs, int B, int C, int L, float* out, RowVec rv) {
  // threads-per-row = min(256, pow2ceil(L/4)); 1024/tpr batch rows in flight per pass
  const int c = blockIdx.x, r = threadIdx.x >> rv.shift, t0 = threadIdx.x & ((1 << rv.shift) - 1), rpp = 1024 >> rv.shift;
  float s0 = 0.f, s1 = 0.f;
  for (int b = r; b < B; b += rpp) {
    const float4* row = reinterpret_cast<const float4*>(x + (long long)b * x_bs + (long long)c * L);
    for (int t = t0; t < rv.L4; t += 1 << rv.shift) {
      const float4 v = row[t];
      s0 += v.x + v.y;
      s1 += v.z + v.w;
    }
  }
  const float s = block_sum1024(s0 + s1);
  if (threadIdx.x == 0) out[c] = s;
}

extern "C" int fst_row_sum(const float* x, int64_t x_bs, int B, int C, int L, float* out, void* stream) {
  FST_REQUIRE(x && out && B > 0 && C > 0 && L > 0, "fst_row_sum: bad arguments");
  FST_REQUIRE(B == 1 || x_bs >= (int64_t)C * L, "fst_row_sum: batch stride %lld < C*L = %lld", (long long)x_bs, (long long)C * L);
  if (vec_ok(L, {x}) && x_bs % 4 == 0)
    hipLaunchKernelGGL(row_sum_vec_kernel, dim3(C), dim3(1024), 0, (hipStream_t)stream, x, (long long)x_bs, B, C, L, out, row_vec(L));
  else
    hipLaunchKernelGGL(row_sum_kernel, dim3(C), dim3(1024), 0, (hipStream_t)stream, x, (long long)x_bs, B, C, L, out);
  FST_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- BatchNorm
// Batch moments from SHIFTED sums, never Σx² − (Σx)²/N of the raw values: the 1x1 shortcut of a univariate extractor is
// y = w·x + b per channel, and a channel whose |w| happens to be small has |mean| / std in the hundreds (903 measured at the
// metric configuration), where the textbook form loses every digit of the variance in fp32 (found by
// tests/test_gpu_full_size.py: 8.7e-4 on the feature at B = 256, invisible at B = 4).  Every workgroup of a channel shifts by
// the same sample k = y[0, c, 0], so |x − k| is of the order of the standard deviation and the fp32 sums of (x − k) and
// (x − k)² carry ~1e-6; the slot stores (count, k, Σ(x−k), Σ(x−k)²) and fst_bn_finalize turns the slots into (count, mean, M2)
// and merges them with Chan's formula IN DOUBLE, in slot order: no atomics, no zero fill, the same bits on every run, and
// slots of other ranks (their own shifts) merge the same way.
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* y, int B, int C, int L, float* part) {
  const int c = blockIdx.x;
  const float k = y[(long long)c * L];
  float s1 = 0.f, s2 = 0.f;
  int n = 0;
  for (int b = blockIdx.y; b < B; b += gridDim.y) {
    const float* row = y + ((long long)b * C + c) * L;
    for (int t = threadIdx.x; t < L; t += 256) {
      const float d = row[t] - k;
      s1 += d; s2 += d * d;
    }
    n += L;
  }
  block_sum2(s1, s2);
  if (threadIdx.x == 0) {
    float* o = part + ((long long)c * FST_BN_SLOTS + blockIdx.y) * 4;
    o[0] = (float)n; o[1] = k; o[2] = s1; o[3] = s2;
  }
}

__global__ __launch_bounds__(256) void bn_stats_vec_kernel(const float* y, int B, int C, int L, float* part, RowVec rv) {
  const int c = blockIdx.x, r = threadIdx.x >> rv.shift, t0 = threadIdx.x & ((1 << rv.shift) - 1), rpp = 256 >> rv.shift;
  const float k = y[(long long)c * L];
  float s1 = 0.f, s2 = 0.f;
  for (int b = blockIdx.y + r * gridDim.y; b < B; b += rpp * gridDim.y) {
    const float4* row = reinterpret_cast<const float4*>(y + ((long long)b * C + c) * L);
    for (int t = t0; t < rv.L4; t += 1 << rv.shift) {
      const float4 v = row[t];
      const float dx = v.x - k, dy = v.y - k, dz = v.z - k, dw = v.w - k;
      s1 += (dx + dy) + (dz + dw);
      s2 += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
  }
  block_sum2(s1, s2);
  if (threadIdx.x == 0) {
    // batch rows of this slot: b = y + j·gridDim.y < B
    const int rows = blockIdx.y < B ? (B - 1 - blockIdx.y) / gridDim.y + 1 : 0;
    float* o = part + ((long long)c * FST_BN_SLOTS + blockIdx.y) * 4;
    o[0] = (float)rows * (float)L; o[1] = k; o[2] = s1; o[3] = s2;
  }
}

extern "C" int fst_bn_stats(const float* y, int B, int C, int L, float* part, int64_t numel, void* stream) {
  FST_REQUIRE(y && part && B > 0 && C > 0 && L > 0, "fst_bn_stats: bad arguments");
  FST_REQUIRE_EXTENT("fst_bn_stats", B, C, L, numel);
  // always FST_BN_SLOTS workgroups per channel: a slot without samples stores a zero count, so no slot is left unwritten
  if (vec_ok(L, {y}))
    hipLaunchKernelGGL(bn_stats_vec_kernel, dim3(C, FST_BN_SLOTS), dim3(256), 0, (hipStream_t)stream, y, B, C, L, part, row_vec(L));
  else
    hipLaunchKernelGGL(bn_stats_kernel, dim3(C, FST_BN_SLOTS), dim3(256), 0, (hipStream_t)stream, y, B, C, L, part);
  FST_LAUNCH_CHECK();
  return 0;
}

__global__ void bn_finalize_kernel(const float* part, int n_slots, const float* gamma, const float* beta, float* rmean,
                                   float* rvar, int train, int C, float eps, float momentum, float* stats) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float mean, var;
  if (train) {
    // slots merged in index order, in double: deterministic, and exact enough that the order would not matter anyway
    double n = 0.0, m = 0.0, m2 = 0.0;
    const float* q = part + (long long)c * n_slots * 4;
    for (int s = 0; s < n_slots; ++s) {
      const double nb = q[4 * s];
      if (nb <= 0.0) continue;
      const double s1 = q[4 * s + 2], s2 = q[4 * s + 3];
      const double mb = (double)q[4 * s + 1] + s1 / nb;                  // the slot's mean and M2 from its shifted sums
      double Mb = s2 - s1 * s1 / nb;
      if (Mb < 0.0) Mb = 0.0;
      const double nn = n + nb, d = mb - m;
      m += d * (nb / nn);
      m2 += Mb + d * d * (n * nb / nn);
      n = nn;
    }
    const double v = n > 0.0 ? m2 / n : 0.0;
    mean = (float)m;
    var = (float)v;
    const float unbiased = n > 1.0 ? (float)(m2 / (n - 1.0)) : var;
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * unbiased;
  } else {
    mean = rmean[c];
    var = rvar[c];
  }
  const float invstd = 1.0f / sqrtf(var + eps);
  const float scale = gamma[c] * invstd;
  stats[c] = mean;
  stats[C + c] = invstd;
  stats[2 * C + c] = scale;
  stats[3 * C + c] = beta[c] - mean * scale;
}

extern "C" int fst_bn_finalize(const float* part, int n_slots, const float* gamma, const float* beta, float* running_mean,
                               float* running_var, int train, int C, float eps, float momentum, float* stats, void* stream) {
  FST_REQUIRE(gamma && beta && running_mean && running_var && stats && C > 0, "fst_bn_finalize: bad arguments");
  FST_REQUIRE(!train || (part && n_slots > 0), "fst_bn_finalize: train mode needs the moment partials (n_slots=%d)", n_slots);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, part, n_slots, gamma, beta,
                     running_mean, running_var, train, C, eps, momentum, stats);
  FST_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* y, const float* stats, const float* res,
                                                       const float* res_stats, float* out, int C, int L, int relu) {
  const int bc = blockIdx.x;           // b*C + c
  const int c = bc % C;
  const float sc = stats[2 * C + c], sh = stats[3 * C + c];
  float rsc = 1.f, rsh = 0.f;
  if (res && res_stats) { rsc = res_stats[2 * C + c]; rsh = res_stats[3 * C + c]; }
  const long long base = (long long)bc * L;
  for (int t = threadIdx.x; t < L; t += 256) {
    float v = y[base + t] * sc + sh;
    if (res) v += res[base + t] * rsc + rsh;
    if (relu) v = fmaxf(v, 0.f);
    out[base + t] = v;
  }
}

__global__ __launch_bounds__(256) void bn_apply_vec_kernel(const float* y, const float* stats, const float* res,
                                                           const float* res_stats, float* out, int rows, int C, int L,
                                                           int relu, RowVec rv) {
  const int r = threadIdx.x >> rv.shift, t0 = threadIdx.x & ((1 << rv.shift) - 1), rpp = 256 >> rv.shift;
  const int bc = blockIdx.x * rpp + r;           // b*C + c
  if (bc >= rows) return;
  const int c = bc % C;
  const float sc = stats[2 * C + c], sh = stats[3 * C + c];
  float rsc = 1.f, rsh = 0.f;
  if (res && res_stats) { rsc = res_stats[2 * C + c]; rsh = res_stats[3 * C + c]; }
  const long long base = (long long)bc * rv.L4;
  const float4* y4 = reinterpret_cast<const float4*>(y) + base;
  const float4* r4 = res ? reinterpret_cast<const float4*>(res) + base : nullptr;
  float4* o4 = reinterpret_cast<float4*>(out) + base;
  for (int t = t0; t < rv.L4; t += 1 << rv.shift) {
    float4 v = y4[t];
    v.x = v.x * sc + sh; v.y = v.y * sc + sh; v.z = v.z * sc + sh; v.w = v.w * sc + sh;
    if (r4) {
      const float4 q = r4[t];
      v.x += q.x * rsc + rsh; v.y += q.y * rsc + rsh; v.z += q.z * rsc + rsh; v.w += q.w * rsc + rsh;
    }
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    o4[t] = v;
  }
}

extern "C" int fst_bn_apply(const float* y, const float* stats, const float* res, const float* res_stats, float* out,
                            int B, int C, int L, int relu, int64_t numel, void* stream) {
  FST_REQUIRE(y && stats && out && B > 0 && C > 0 && L > 0, "fst_bn_apply: bad arguments");
  FST_REQUIRE_EXTENT("fst_bn_apply", B, C, L, numel);
  if (vec_ok(L, {y, res, out})) {
    const RowVec rv = row_vec(L);
    const int rpp = 256 >> rv.shift;
    hipLaunchKernelGGL(bn_apply_vec_kernel, dim3((B * C + rpp - 1) / rpp), dim3(256), 0, (hipStream_t)stream, y, stats, res,
                       res_stats, out, B * C, C, L, relu, rv);
  } else {
    hipLaunchKernelGGL(bn_apply_kernel, dim3(B * C), dim3(256), 0, (hipStream_t)stream, y, stats, res, res_stats, out, C, L, relu);
  }
  FST_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* dy, const float* y, const float* out,
                                                            const float* stats, int B, int C, int L, int relu, float* red) {
  const int c = blockIdx.x;
  const float mean = stats[c], invstd = stats[C + c];
  float s1 = 0.f, s2 = 0.f;
  for (int b = blockIdx.y; b < B; b += gridDim.y) {
    const long long base = ((long long)b * C + c) * L;
    for (int t = threadIdx.x; t < L; t += 256) {
      float g = dy[base + t];
      if (relu && !(out[base + t] > 0.f)) g = 0.f;
      s1 += g;
      s2 += g * (y[base + t] - mean) * invstd;
    }
  }
  block_sum2(s1, s2);
  if (threadIdx.x == 0) {
    red[(long long)c * FST_BN_SLOTS + blockIdx.y] = s1;
    red[((long long)C + c) * FST_BN_SLOTS + blockIdx.y] = s2;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_vec_kernel(const float* dy, const float* y, const float* out,
                                                                const float* stats, int B, int C, int L, int relu,
                                                                float* red, RowVec rv) {
  const int c = blockIdx.x, r = threadIdx.x >> rv.shift, t0 = threadIdx.x & ((1 << rv.shift) - 1), rpp = 256 >> rv.shift;
  const float mean = stats[c], invstd = stats[C + c];
  float s1 = 0.f, s2 = 0.f;
  for (int b = blockIdx.y + r * gridDim.y; b < B; b += rpp * gridDim.y) {
    const long long base = ((long long)b * C + c) * rv.L4;
    const float4* dy4 = reinterpret_cast<const float4*>(dy) + base;
    const float4* y4 = reinterpret_cast<const float4*>(y) + base;
    const float4* o4 = reinterpret_cast<const float4*>(out) + base;
    for (int t = t0; t < rv.L4; t += 1 << rv.shift) {
      float4 g = dy4[t];
      const float4 v = y4[t];
      if (relu) {
        const float4 o = o4[t];
        if (!(o.x > 0.f)) g.x = 0.f;
        if (!(o.y > 0.f)) g.y = 0.f;
        if (!(o.z > 0.f)) g.z = 0.f;
        if (!(o.w > 0.f)) g.w = 0.f;
      }
      s1 += (g.x + g.y) + (g.z + g.w);
      s2 += (g.x * (v.x - mean) + g.y * (v.y - mean) + g.z * (v.z - mean) + g.w * (v.w - mean)) * invstd;
    }
  }
  block_sum2(s1, s2);
  if (threadIdx.x == 0) {
    red[(long long)c * FST_BN_SLOTS + blockIdx.y] = s1;
    red[((long long)C + c) * FST_BN_SLOTS + blockIdx.y] = s2;
  }
}

extern "C" int fst_bn_bwd_reduce(const float* dy, const float* y, const float* out, const float* stats, int B, int C,
                                 int L, int relu, float* red, int64_t numel, void* stream) {
  FST_REQUIRE(dy && y && stats && red && (!relu || out) && B > 0 && C > 0 && L > 0, "fst_bn_bwd_reduce: bad arguments");
  FST_REQUIRE_EXTENT("fst_bn_bwd_reduce", B, C, L, numel);
  if (vec_ok(L, {dy, y, out}))
    hipLaunchKernelGGL(bn_bwd_reduce_vec_kernel, dim3(C, FST_BN_SLOTS), dim3(256), 0, (hipStream_t)stream, dy,
                       y, out, stats, B, C, L, relu, red, row_vec(L));
  else
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(C, FST_BN_SLOTS), dim3(256), 0, (hipStream_t)stream, dy, y,
                       out, stats, B, C, L, relu, red);
  FST_LAUNCH_CHECK();
  return 0;
}

// red: [2][C][n_slots] partial sums (n_slots = FST_BN_SLOTS straight from fst_bn_bwd_reduce, or 1 once the caller has added
// them, e.g. across ranks), added here in slot order.  red_out (optional, [2C]): the slot sums = (dβ, dγ), written by the
// threads of batch row 0 — the parameter gradients need no reduction launch of their own.
__device__ __forceinline__ void bn_red_sums(const float* red, int n_slots, int C, int c, float& r1, float& r2) {
  r1 = 0.f; r2 = 0.f;
  for (int s = 0; s < n_slots; ++s) {
    r1 += red[(long long)c * n_slots + s];
    r2 += red[((long long)C + c) * n_slots + s];
  }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* dy, const float* y, const float* out,
                                                           const float* stats, const float* red, int n_slots, float* red_out,
                                                           float* dx, float* row_sums, int C, int L, int relu, int train, float invN) {
  const int bc = blockIdx.x;
  const int c = bc % C;
  const float mean = stats[c], invstd = stats[C + c], scale = stats[2 * C + c];
  float r1 = 0.f, r2 = 0.f;
  if (red) bn_red_sums(red, n_slots, C, c, r1, r2);
  if (red_out && bc < C && threadIdx.x == 0) { red_out[c] = r1; red_out[C + c] = r2; }
  const float m1 = train ? r1 * invN : 0.f, m2 = train ? r2 * invN : 0.f;
  const long long base = (long long)bc * L;
  float rs = 0.f, dummy = 0.f;
  for (int t = threadIdx.x; t < L; t += 256) {
    float g = dy[base + t];
    if (relu && !(out[base + t] > 0.f)) g = 0.f;
    const float xh = (y[base + t] - mean) * invstd;
    const float d = scale * (g - m1 - xh * m2);
    dx[base + t] = d;
    rs += d;
  }
  if (row_sums) {                                          // uniform
    block_sum2(rs, dummy);
    if (threadIdx.x == 0) row_sums[bc] = rs;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_vec_kernel(const float* dy, const float* y, const float* out,
                                                               const float* stats, const float* red, int n_slots,
                                                               float* red_out, float* dx, float* row_sums, int rows, int C,
                                                               int L, int relu, int train, float invN, RowVec rv) {
  __shared__ float wsum[4];
  const int r = threadIdx.x >> rv.shift, t0 = threadIdx.x & ((1 << rv.shift) - 1), rpp = 256 >> rv.shift;
  const int bc_raw = blockIdx.x * rpp + r;
  const bool live = bc_raw < rows;
  const int bc = live ? bc_raw : rows - 1;                 // (idle row groups of the last block walk the last row and store nothing)
  const int c = bc % C;
  const float mean = stats[c], invstd = stats[C + c], scale = stats[2 * C + c];
  float r1 = 0.f, r2 = 0.f;
  if (red) bn_red_sums(red, n_slots, C, c, r1, r2);
  if (red_out && live && bc < C && t0 == 0) { red_out[c] = r1; red_out[C + c] = r2; }
  const float m1 = train ? r1 * invN : 0.f, m2 = train ? r2 * invN : 0.f;
  const long long base = (long long)bc * rv.L4;
  const float4* dy4 = reinterpret_cast<const float4*>(dy) + base;
  const float4* y4 = reinterpret_cast<const float4*>(y) + base;
  const float4* o4 = reinterpret_cast<const float4*>(out) + base;
  float4* dx4 = reinterpret_cast<float4*>(dx) + base;
  float rs = 0.f;
  for (int t = t0; t < rv.L4; t += 1 << rv.shift) {
    float4 g = dy4[t];
    const float4 v = y4[t];
    if (relu) {
      const float4 o = o4[t];
      if (!(o.x > 0.f)) g.x = 0.f;
      if (!(o.y > 0.f)) g.y = 0.f;
      if (!(o.z > 0.f)) g.z = 0.f;
      if (!(o.w > 0.f)) g.w = 0.f;
    }
    float4 d;
    d.x = scale * (g.x - m1 - (v.x - mean) * invstd * m2);
    d.y = scale * (g.y - m1 - (v.y - mean) * invstd * m2);
    d.z = scale * (g.z - m1 - (v.z - mean) * invstd * m2);
    d.w = scale * (g.w - m1 - (v.w - mean) * invstd * m2);
    if (live) dx4[t] = d;
    rs += (d.x + d.y) + (d.z + d.w);
  }
  // Σ_t dx of this (sample, channel) row — the bias gradient of the conv in front of the BatchNorm is Σ_b of these: the conv's
  // backward then needs no pass of its own over dx.  A row's 2^shift threads are consecutive: a xor butterfly inside the wave,
  // then (rows wider than a wave) the waves' totals through LDS in wave order.
  if (row_sums) {                                          // uniform
    const int w = rv.shift < 6 ? rv.shift : 6;
    for (int o = 1 << (w - 1); o > 0; o >>= 1) rs += __shfl_xor(rs, o, 64);
    if (rv.shift <= 6) {
      if (live && t0 == 0) row_sums[bc] = rs;
    } else {
      if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = rs;
      __syncthreads();
      if (live && t0 == 0) {
        const int w0 = threadIdx.x >> 6, nw = 1 << (rv.shift - 6);
        float tot = 0.f;
        for (int i = 0; i < nw; ++i) tot += wsum[w0 + i];
        row_sums[bc] = tot;
      }
    }
  }
}

extern "C" int fst_bn_bwd_apply(const float* dy, const float* y, const float* out, const float* stats, const float* red,
                                int n_slots, float* red_out, float* dx, float* row_sums, int B, int C, int L, int relu, int train,
                                int B_total, int64_t numel, void* stream) {
  FST_REQUIRE(dy && y && stats && dx && (!relu || out) && (!train || red), "fst_bn_bwd_apply: bad arguments");
  FST_REQUIRE(B > 0 && C > 0 && L > 0 && B_total >= B, "fst_bn_bwd_apply: B=%d C=%d L=%d B_total=%d", B, C, L, B_total);
  FST_REQUIRE(!red || n_slots > 0, "fst_bn_bwd_apply: n_slots=%d", n_slots);
  FST_REQUIRE(!red_out || red, "fst_bn_bwd_apply: red_out needs red");
  FST_REQUIRE_EXTENT("fst_bn_bwd_apply", B, C, L, numel);     // the launch walks B (not B_total) samples
  if (vec_ok(L, {dy, y, out, dx})) {
    const RowVec rv = row_vec(L);
    const int rpp = 256 >> rv.shift;
    hipLaunchKernelGGL(bn_bwd_apply_vec_kernel, dim3((B * C + rpp - 1) / rpp), dim3(256), 0, (hipStream_t)stream, dy, y, out,
                       stats, red, n_slots, red_out, dx, row_sums, B * C, C, L, relu, train, 1.0f / ((float)B_total * (float)L), rv);
  } else {
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(B * C), dim3(256), 0, (hipStream_t)stream, dy, y, out, stats, red, n_slots,
                       red_out, dx, row_sums, C, L, relu, train, 1.0f / ((float)B_total * (float)L));
  }
  FST_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- WaveGlow gate
__global__ __launch_bounds__(256) void gate_fwd_kernel(float* g, float* acts, int n, int L) {
  const int b = blockIdx.y;
  const long long total = (long long)n * L;
  float* gt = g + (long long)b * 2 * total;
  float* gs = gt + total;
  float* ab = acts + (long long)b * total;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const float t = tanhf(gt[i]);
    const float s = 1.0f / (1.0f + expf(-gs[i]));
    gt[i] = t;
    gs[i] = s;
    ab[i] = t * s;
  }
}

extern "C" int fst_gate_fwd(float* g_ts, float* acts, int B, int n, int L, int64_t numel_acts, void* stream) {
  FST_REQUIRE(g_ts && acts && B > 0 && n > 0 && L > 0, "fst_gate_fwd: bad arguments");
  FST_REQUIRE_EXTENT("fst_gate_fwd", B, n, L, numel_acts);
  long long blocks = ((long long)n * L + 255) / 256;
  if (blocks > 64) blocks = 64;
  hipLaunchKernelGGL(gate_fwd_kernel, dim3((unsigned)blocks, B), dim3(256), 0, (hipStream_t)stream, g_ts, acts, n, L);
  FST_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* ts, const float* dacts, float* dg, int n, int L) {
  const int b = blockIdx.y;
  const long long total = (long long)n * L;
  const float* tt = ts + (long long)b * 2 * total;
  const float* ss = tt + total;
  const float* da = dacts + (long long)b * total;
  float* dgt = dg + (long long)b * 2 * total;
  float* dgs = dgt + total;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const float t = tt[i], s = ss[i], d = da[i];
    dgt[i] = d * s * (1.f - t * t);
    dgs[i] = d * t * s * (1.f - s);
  }
}

extern "C" int fst_gate_bwd(const float* ts, const float* dacts, float* dg, int B, int n, int L, int64_t numel_acts,
                            void* stream) {
  FST_REQUIRE(ts && dacts && dg && B > 0 && n > 0 && L > 0, "fst_gate_bwd: bad arguments");
  FST_REQUIRE_EXTENT("fst_gate_bwd", B, n, L, numel_acts);
  long long blocks = ((long long)n * L + 255) / 256;
  if (blocks > 64) blocks = 64;
  hipLaunchKernelGGL(gate_bwd_kernel, dim3((unsigned)blocks, B), dim3(256), 0, (hipStream_t)stream, ts, dacts, dg, n, L);
  FST_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- affine coupling
// mode 0: forward   xn1 = exp(s)*u1 + b          mode 1: inverse   xn1 = (x1 - b) / exp(s)
// sums (optional, forward mode): per-workgroup partials of Σ log_s and Σ xn² — the two full-tensor reductions of
// WaveGlowLoss (Simplified_NF_WaveGlow.py:230-241) taken while the tensors pass through anyway: slot (b·gridDim.x +
// blockIdx.x) gets (Σ log_s, Σ xn²) of that workgroup's elements; the caller adds the slots (16 k atomics on two
// addresses would serialise: 160 µs measured).
__global__ __launch_bounds__(256) void coupling_fwd_kernel(const float* u, const float* o, float* xn, int h, int L, int mode,
                                                           float* sums) {
  const int b = blockIdx.y;
  const long long half = (long long)h * L;
  const float* ub = u + (long long)b * 2 * half;
  const float* ob = o + (long long)b * 2 * half;
  float* xb = xn + (long long)b * 2 * half;
  float acc_ls = 0.f, acc_sq = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < half; i += (long long)gridDim.x * 256) {
    const float u0 = ub[i];
    xb[i] = u0;
    const float bb = ob[i], s = ob[half + i], u1 = ub[half + i];
    const float x1 = mode == 0 ? expf(s) * u1 + bb : (u1 - bb) / expf(s);
    xb[half + i] = x1;
    acc_ls += s;
    acc_sq += u0 * u0 + x1 * x1;
  }
  if (sums) {                                            // kernel argument: uniform
    block_sum2(acc_ls, acc_sq);
    if (threadIdx.x == 0) {
      float* slot = sums + 2 * ((long long)b * gridDim.x + blockIdx.x);
      slot[0] = acc_ls;
      slot[1] = acc_sq;
    }
  }
}

static long long coupling_blocks(int h, int L) {
  long long blocks = ((long long)h * L + 1023) / 1024;       // four elements per thread before the block reduction
  return blocks > 64 ? 64 : (blocks < 1 ? 1 : blocks);
}
extern "C" int64_t fst_coupling_sum_slots(int B, int h, int L) { return (int64_t)B * coupling_blocks(h, L); }

static int launch_coupling_fwd(const float* u, const float* o, float* xn, int B, int h, int L, int mode, int64_t numel,
                               float* sums, void* stream) {
  FST_REQUIRE(u && o && xn && B > 0 && h > 0 && L > 0, "fst_coupling: bad arguments");
  FST_REQUIRE_EXTENT("fst_coupling", B, 2 * h, L, numel);
  const long long blocks = coupling_blocks(h, L);
  hipLaunchKernelGGL(coupling_fwd_kernel, dim3((unsigned)blocks, B), dim3(256), 0, (hipStream_t)stream, u, o, xn, h, L, mode, sums);
  FST_LAUNCH_CHECK();
  return 0;
}
extern "C" int fst_coupling_fwd(const float* u, const float* o, float* xn, int B, int h, int L, int64_t numel, float* sums,
                                void* stream) {
  return launch_coupling_fwd(u, o, xn, B, h, L, 0, numel, sums, stream);
}
extern "C" int fst_coupling_inv_fwd(const float* x, const float* o, float* xn, int B, int h, int L, int64_t numel,
                                    void* stream) {
  return launch_coupling_fwd(x, o, xn, B, h, L, 1, numel, nullptr, stream);
}

// forward-coupling backward.  dxn: grad of xn (2h ch, may be null = 0); dlogs: extra grad flowing into log_s (may be null);
// g_ls, g_sq (each may be null): DEVICE scalars d/dΣlog_s and d/dΣxn² of the fused loss reductions:  dxn_eff = dxn + 2·g_sq·xn
//   du0 = dxn0 ; du1 = dxn1*exp(s) ; db = dxn1 ; ds = dxn1*u1*exp(s) + dlogs + g_ls
__global__ __launch_bounds__(256) void coupling_bwd_kernel(const float* u, const float* o, const float* dxn,
                                                           const float* dlogs, const float* g_ls_p, const float* g_sq_p,
                                                           float* du, float* d_o, int h, int L) {
  const int b = blockIdx.y;
  const long long half = (long long)h * L;
  const long long off = (long long)b * 2 * half;
  const float g_ls = g_ls_p ? g_ls_p[0] : 0.f, g_sq2 = g_sq_p ? 2.f * g_sq_p[0] : 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < half; i += (long long)gridDim.x * 256) {
    const float es = expf(o[off + half + i]);
    const float u0 = u[off + i], u1 = u[off + half + i];
    const float x1 = es * u1 + o[off + i];
    const float g0 = (dxn ? dxn[off + i] : 0.f) + g_sq2 * u0;
    const float g1 = (dxn ? dxn[off + half + i] : 0.f) + g_sq2 * x1;
    du[off + i] = g0;
    du[off + half + i] = g1 * es;
    d_o[off + i] = g1;
    float ds = g1 * u1 * es + g_ls;
    if (dlogs) ds += dlogs[(long long)b * half + i];
    d_o[off + half + i] = ds;
  }
}

extern "C" int fst_coupling_bwd(const float* u, const float* o, const float* dxn, const float* dlogs, const float* g_ls,
                                const float* g_sq, float* du, float* d_o, int B, int h, int L, int64_t numel, void* stream) {
  FST_REQUIRE(u && o && (dxn || g_ls || g_sq) && du && d_o && B > 0 && h > 0 && L > 0, "fst_coupling_bwd: bad arguments");
  FST_REQUIRE_EXTENT("fst_coupling_bwd", B, 2 * h, L, numel);
  long long blocks = ((long long)h * L + 255) / 256;
  if (blocks > 64) blocks = 64;
  hipLaunchKernelGGL(coupling_bwd_kernel, dim3((unsigned)blocks, B), dim3(256), 0, (hipStream_t)stream, u, o, dxn, dlogs, g_ls, g_sq,
                     du, d_o, h, L);
  FST_LAUNCH_CHECK();
  return 0;
}

// inverse-coupling backward: xn1 = (x1 - b)*exp(-s)
//   dx0 = dxn0 ; dx1 = dxn1*exp(-s) ; db = -dxn1*exp(-s) ; ds = -dxn1*xn1
__global__ __launch_bounds__(256) void coupling_inv_bwd_kernel(const float* xn, const float* o, const float* dxn,
                                                               float* dx, float* d_o, int h, int L) {
  const int b = blockIdx.y;
  const long long half = (long long)h * L;
  const long long off = (long long)b * 2 * half;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < half; i += (long long)gridDim.x * 256) {
    const float ies = 1.0f / expf(o[off + half + i]);
    const float g1 = dxn[off + half + i];
    dx[off + i] = dxn[off + i];
    dx[off + half + i] = g1 * ies;
    d_o[off + i] = -g1 * ies;
    d_o[off + half + i] = -g1 * xn[off + half + i];
  }
}

extern "C" int fst_coupling_inv_bwd(const float* xn, const float* o, const float* dxn, float* dx, float* d_o, int B,
                                    int h, int L, int64_t numel, void* stream) {
  FST_REQUIRE(xn && o && dxn && dx && d_o && B > 0 && h > 0 && L > 0, "fst_coupling_inv_bwd: bad arguments");
  FST_REQUIRE_EXTENT("fst_coupling_inv_bwd", B, 2 * h, L, numel);
  long long blocks = ((long long)h * L + 255) / 256;
  if (blocks > 64) blocks = 64;
  hipLaunchKernelGGL(coupling_inv_bwd_kernel, dim3((unsigned)blocks, B), dim3(256), 0, (hipStream_t)stream, xn, o, dxn, dx, d_o, h, L);
  FST_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- small helpers
__global__ void axpy_kernel(float* y, const float* x, float alpha, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    y[i] += alpha * x[i];
}
extern "C" int fst_axpy(float* y, const float* x, float alpha, int64_t n, void* stream) {
  FST_REQUIRE(y && x && n > 0, "fst_axpy: bad arguments");
  long long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y, x, alpha, (long long)n);
  FST_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(256) void add_slices_kernel(float* dst, long long dst_bs, const float* a, long long a_bs,
                                                         const float* bsrc, long long b_bs, int C, int L) {
  const int b = blockIdx.y;
  const long long total = (long long)C * L;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    float v = a[(long long)b * a_bs + i];
    if (bsrc) v += bsrc[(long long)b * b_bs + i];
    dst[(long long)b * dst_bs + i] = v;
  }
}
extern "C" int fst_add_slices(float* dst, int64_t dst_bs, const float* a, int64_t a_bs, const float* b, int64_t b_bs,
                              int B, int C, int L, void* stream) {
  FST_REQUIRE(dst && a && B > 0 && C > 0 && L > 0, "fst_add_slices: bad arguments");
  FST_REQUIRE(B == 1 || (dst_bs >= (int64_t)C * L && a_bs >= (int64_t)C * L && (!b || b_bs >= (int64_t)C * L)),
              "fst_add_slices: a batch stride is smaller than the C*L = %lld slice it strides over", (long long)C * L);
  long long blocks = ((long long)C * L + 255) / 256;
  if (blocks > 64) blocks = 64;
  hipLaunchKernelGGL(add_slices_kernel, dim3((unsigned)blocks, B), dim3(256), 0, (hipStream_t)stream, dst, (long long)dst_bs, a,
                     (long long)a_bs, b, (long long)b_bs, C, L);
  FST_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- log|det W| and W^{-T} of the invertible 1x1 conv
// torch.logdet(W) (Simplified_NF_WaveGlow.py:40) is an LU factorisation plus ~17 tiny launches forward and two triangular
// solves plus ~10 launches backward — for a 50x50 matrix, three times per step, all on the critical path of the captured step.
// One workgroup does both here: in-place Gauss-Jordan inversion with partial pivoting, in double precision in LDS; the pivots
// give log|det| and its sign, the inverse (transposed) is the gradient d log|det W| / dW the backward needs.
//   out[0] = log|det W| with torch.logdet's conventions: NaN for a negative determinant, -inf for a singular matrix
//   out[1] = sign of the determinant (+1, -1, 0)
__global__ __launch_bounds__(256) void logdet_inv_kernel(const float* W, int n, float* out, float* inv_t) {
  extern __shared__ __attribute__((aligned(16))) double sa[];       // [n][n]
  __shared__ int piv_row[256];
  __shared__ double red_v[256];
  __shared__ int red_i[256];
  __shared__ double s_logabs;
  __shared__ int s_sign, s_p;
  const int tid = threadIdx.x;
  for (int i = tid; i < n * n; i += 256) sa[i] = (double)W[i];
  if (tid == 0) { s_logabs = 0.0; s_sign = 1; }
  __syncthreads();
  for (int k = 0; k < n; ++k) {
    // pivot: largest |a[i][k]|, i >= k (lowest index on ties: deterministic)
    double best = -1.0; int bi = k;
    for (int i = k + tid; i < n; i += 256) {
      const double v = fabs(sa[i * n + k]);
      if (v > best) { best = v; bi = i; }
    }
    red_v[tid] = best; red_i[tid] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (tid < s) {
        const double v2 = red_v[tid + s]; const int i2 = red_i[tid + s];
        if (v2 > red_v[tid] || (v2 == red_v[tid] && i2 < red_i[tid])) { red_v[tid] = v2; red_i[tid] = i2; }
      }
      __syncthreads();
    }
    if (tid == 0) {
      const int p = red_i[0];
      s_p = p; piv_row[k] = p;
      const double pv = sa[p * n + k];
      if (pv == 0.0) s_sign = 0;
      else {
        s_logabs += log(fabs(pv));
        if ((pv < 0.0) != (p != k)) s_sign = -s_sign;          // a negative pivot and a row swap each flip the sign
      }
    }
    __syncthreads();
    if (s_sign == 0) break;                                       // singular (uniform: read from LDS after the barrier)
    const int p = s_p;
    if (p != k) {
      for (int j = tid; j < n; j += 256) { const double t = sa[k * n + j]; sa[k * n + j] = sa[p * n + j]; sa[p * n + j] = t; }
      __syncthreads();
    }
    const double rp = 1.0 / sa[k * n + k];
    __syncthreads();
    for (int j = tid; j < n; j += 256) sa[k * n + j] = (j == k ? 1.0 : sa[k * n + j]) * rp;
    __syncthreads();
    // eliminate column k from every other row; the column itself becomes -f * (1/pivot) (in-place inverse bookkeeping)
    for (int e = tid; e < n * n; e += 256) {
      const int i = e / n, j = e - i * n;
      if (i == k || j == k) continue;
      sa[e] -= sa[i * n + k] * sa[k * n + j];
    }
    __syncthreads();
    for (int i = tid; i < n; i += 256)
      if (i != k) sa[i * n + k] = -sa[i * n + k] * rp;
    __syncthreads();
  }
  const int sign = s_sign;
  if (sign != 0) {
    // undo the row swaps as column swaps, last first: A^{-1} = (P·A)^{-1}·P
    for (int k = n - 1; k >= 0; --k) {
      const int p = piv_row[k];
      if (p != k) {
        for (int i = tid; i < n; i += 256) { const double t = sa[i * n + k]; sa[i * n + k] = sa[i * n + p]; sa[i * n + p] = t; }
        __syncthreads();
      }
    }
  }
  for (int e = tid; e < n * n; e += 256) {
    const int i = e / n, j = e - i * n;
    inv_t[e] = sign != 0 ? (float)sa[j * n + i] : __builtin_nanf("");
  }
  if (tid == 0) {
    out[0] = sign > 0 ? (float)s_logabs : (sign < 0 ? __builtin_nanf("") : -__builtin_inff());
    out[1] = (float)sign;
  }
}

extern "C" int fst_logdet_inv(const float* W, int n, float* out, float* inv_t, void* stream) {
  FST_REQUIRE(W && out && inv_t && n > 0 && n <= 96, "fst_logdet_inv: needs 0 < n <= 96 (one workgroup holds the matrix in LDS as doubles); n=%d", n);
  const size_t lds = (size_t)n * n * sizeof(double);
  if (lds > 48 * 1024) {
    // (not fst_allow_full_lds: the kernel also holds a few KiB of static LDS, so "all 160 KiB dynamic" is refused)
    static bool raised = false;
    if (!raised) {
      hipError_t e = hipFuncSetAttribute((const void*)logdet_inv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 96 * (int)sizeof(double));
      FST_REQUIRE(e == hipSuccess, "fst_logdet_inv: hipFuncSetAttribute: %s", hipGetErrorString(e));
      raised = true;
    }
  }
  hipLaunchKernelGGL(logdet_inv_kernel, dim3(1), dim3(256), lds, (hipStream_t)stream, W, n, out, inv_t);
  FST_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- weight-norm fold of a whole WN into its flat weight tensor
// The reference wraps 18 convs of every WN in old-style weight_norm (Simplified_NF_WaveGlow.py:69-99): w = g·v/‖v‖ per output
// channel, recomputed every forward — 18 launches forward and 18 backward per WN as stock ops, plus the concatenation into the
// flat tensor the fused kernels read.  Here ONE launch per WN and direction walks a row table: row r = one output channel of one
// conv (or one plain-copy segment: biases, the un-normed end conv), a wave per row.
//   table[r] = { v_row (pointer), g (pointer to the row's scalar, 0 = plain copy), dst (element offset into flat), len,
//                dv (element offset into the gradient buffer), dg (element offset of the row's g gradient) }   (6 x int64)
struct WnFoldRow { long long v, g, dst, len, dv, dg; };

__global__ __launch_bounds__(256) void wn_fold_fwd_kernel(const WnFoldRow* __restrict__ table, int n_rows, float* flat, float* norms) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= n_rows) return;
  const WnFoldRow row = table[r];
  const float* v = reinterpret_cast<const float*>(row.v);
  const int len = (int)row.len;
  float* dst = flat + row.dst;
  if (row.g == 0) {
    for (int j = lane; j < len; j += 64) dst[j] = v[j];
    return;
  }
  float ss = 0.f;
  for (int j = lane; j < len; j += 64) ss += v[j] * v[j];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  const float norm = sqrtf(ss);
  const float scale = *reinterpret_cast<const float*>(row.g) / norm;
  for (int j = lane; j < len; j += 64) dst[j] = v[j] * scale;
  if (lane == 0) norms[r] = norm;
}

// dv = (g/‖v‖)·(dw − v·(dw·v)/‖v‖²),  dg = (dw·v)/‖v‖   (torch's weight_norm backward); copy rows: dv = dw
__global__ __launch_bounds__(256) void wn_fold_bwd_kernel(const WnFoldRow* __restrict__ table, int n_rows, const float* d_flat,
                                                          const float* norms, float* dpar) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= n_rows) return;
  const WnFoldRow row = table[r];
  const float* v = reinterpret_cast<const float*>(row.v);
  const int len = (int)row.len;
  const float* dw = d_flat + row.dst;
  float* dv = dpar + row.dv;
  if (row.g == 0) {
    for (int j = lane; j < len; j += 64) dv[j] = dw[j];
    return;
  }
  float dot = 0.f;
  for (int j = lane; j < len; j += 64) dot += dw[j] * v[j];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
  const float norm = norms[r], g = *reinterpret_cast<const float*>(row.g);
  const float a = g / norm, b = dot / (norm * norm);
  for (int j = lane; j < len; j += 64) dv[j] = a * (dw[j] - v[j] * b);
  if (lane == 0) dpar[row.dg] = dot / norm;
}

extern "C" int fst_wn_fold_fwd(const int64_t* table_dev, int n_rows, float* flat, float* norms, void* stream) {
  FST_REQUIRE(table_dev && flat && norms && n_rows > 0, "fst_wn_fold_fwd: bad arguments");
  hipLaunchKernelGGL(wn_fold_fwd_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const WnFoldRow*>(table_dev), n_rows, flat, norms);
  FST_LAUNCH_CHECK();
  return 0;
}
extern "C" int fst_wn_fold_bwd(const int64_t* table_dev, int n_rows, const float* d_flat, const float* norms, float* dpar,
                               void* stream) {
  FST_REQUIRE(table_dev && d_flat && norms && dpar && n_rows > 0, "fst_wn_fold_bwd: bad arguments");
  hipLaunchKernelGGL(wn_fold_bwd_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const WnFoldRow*>(table_dev), n_rows, d_flat, norms, dpar);
  FST_LAUNCH_CHECK();
  return 0;
}
