// NoiseTransfer (widgets.py:150-167 of the reference) for gfx950: the latent mean shift
//     new_t = avg_t + r_t·mean_b(z_t),  new_s = avg_s + r_s·mean_b(z_s),  dist = new_t − new_s,
//     learned = selu(W·dist + bias)   (an unbatched 1x1 conv over the [C, L] map),   out[b] = learned + z_s[b]
// as three launches forward and four backward (the ATen composition is ~12 and ~15): both batch means in one pass over the two
// [B, C, L] tensors (per-slice partial sums, added in a fixed order: deterministic), the state update + 1x1 conv + SELU in one
// small kernel, the broadcast add; backward: the batch sum of the cotangent, SELU' and Wᵀ in one small kernel, the two small
// parameter gradients, and ONE pass writing both input gradients.  All HBM-bound elementwise / reduction work — no MFMA.
#include "fst_common.h"

#define NT_SELU_ALPHA 1.6732632423543772848170429916717f
#define NT_SELU_SCALE 1.0507009873554804934193349852946f

// part[z][s][i] = Σ_{b in slice s} x_z[b][i],  i < N (N % 4 == 0), slices = contiguous runs of samples
__global__ __launch_bounds__(256) void batch_sum_kernel(const float* x0, const float* x1, float* part, int B, long long N, int S) {
  const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= N) return;
  const int s = blockIdx.y;
  const float* x = blockIdx.z ? x1 : x0;
  const int b0 = (int)((long long)s * B / S), b1 = (int)((long long)(s + 1) * B / S);
  float4 acc = {0.f, 0.f, 0.f, 0.f};
  const float* q = x + (long long)b0 * N + i4;
#pragma unroll 8
  for (int b = b0; b < b1; ++b, q += N) {
    const float4 v = *reinterpret_cast<const float4*>(q);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  *reinterpret_cast<float4*>(part + ((long long)blockIdx.z * S + s) * N + i4) = acc;
}

extern "C" int fst_batch_sum(const float* x0, const float* x1, float* part, int B, int64_t N, int S, void* stream) {
  FST_REQUIRE(x0 && part && B > 0 && N > 0 && N % 4 == 0 && S > 0 && S <= B, "fst_batch_sum: bad arguments (B=%d N=%lld S=%d; N %% 4 == 0, "
              "1 <= S <= B)", B, (long long)N, S);
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  FST_REQUIRE(al16(x0) && al16(x1) && al16(part), "fst_batch_sum: tensors must be 16-byte aligned");
  hipLaunchKernelGGL(batch_sum_kernel, dim3((unsigned)((N / 4 + 255) / 256), (unsigned)S, x1 ? 2 : 1), dim3(256), 0, (hipStream_t)stream,
                     x0, x1, part, B, (long long)N, S);
  FST_LAUNCH_CHECK();
  return 0;
}

struct NtFwdParams {
  const float* part;          // [2][S][C·L] slice sums of z_t, z_s
  const float* r_t_dev;       // this call's accumulation ratios: device scalars (a captured step refreshes them) or, when null,
  const float* r_s_dev;       //   the host values below
  float r_t, r_s;
  float* avg_t;               // [C][L] running sums, updated in place (widgets.py:155-158: detached state)
  float* avg_s;
  const float* W;             // [C][C] 1x1 conv
  const float* bias;          // [C]
  float* dist;                // [C][L] out: new_t − new_s     (kept for the backward)
  float* pre;                 // [C][L] out: W·dist + bias
  float* learned;             // [C][L] out: selu(pre)
  int S, B, C, L;
};

// a workgroup = 64 positions × 4 channel groups; dist of all channels at those positions goes through LDS for the C×C product
__global__ __launch_bounds__(256) void noise_transfer_fwd_kernel(NtFwdParams p) {
  extern __shared__ float sd[];                            // [C][64]
  const int lx = threadIdx.x & 63, cg = threadIdx.x >> 6, l = blockIdx.x * 64 + lx;
  const int C = p.C, L = p.L, S = p.S;
  const long long N = (long long)C * L;
  const float rt = p.r_t_dev ? *p.r_t_dev : p.r_t, rs = p.r_s_dev ? *p.r_s_dev : p.r_s;
  const float fB = (float)p.B;
  if (l < L)
    for (int c = cg; c < C; c += 4) {
      const long long i = (long long)c * L + l;
      float st = 0.f, ss = 0.f;
      for (int s = 0; s < S; ++s) {
        st += p.part[(long long)s * N + i];
        ss += p.part[((long long)S + s) * N + i];
      }
      const float nt = p.avg_t[i] + rt * (st / fB), ns = p.avg_s[i] + rs * (ss / fB);
      p.avg_t[i] = nt;
      p.avg_s[i] = ns;
      const float d = nt - ns;
      p.dist[i] = d;
      sd[c * 64 + lx] = d;
    }
  __syncthreads();
  if (l < L)
    for (int o = cg; o < C; o += 4) {
      float acc = p.bias[o];
      const float* w = p.W + (long long)o * C;
      for (int c = 0; c < C; ++c) acc += w[c] * sd[c * 64 + lx];
      const long long i = (long long)o * L + l;
      p.pre[i] = acc;
      p.learned[i] = NT_SELU_SCALE * (acc > 0.f ? acc : NT_SELU_ALPHA * expm1f(acc));
    }
}

extern "C" int fst_noise_transfer_fwd(const float* part, int S, int B, const float* r_t_dev, const float* r_s_dev, float r_t, float r_s,
                                      float* avg_t, float* avg_s, const float* W, const float* bias, float* dist, float* pre,
                                      float* learned, int C, int L, void* stream) {
  FST_REQUIRE(part && avg_t && avg_s && W && bias && dist && pre && learned, "fst_noise_transfer_fwd: null operand");
  FST_REQUIRE(S > 0 && B > 0 && C > 0 && C <= 512 && L > 0, "fst_noise_transfer_fwd: bad shape S=%d B=%d C=%d L=%d (C <= 512)", S, B, C, L);
  NtFwdParams p = {part, r_t_dev, r_s_dev, r_t, r_s, avg_t, avg_s, W, bias, dist, pre, learned, S, B, C, L};
  const size_t lds = (size_t)C * 64 * sizeof(float);
  if (lds > 48 * 1024)
    if (int rc = fst_allow_full_lds((const void*)noise_transfer_fwd_kernel, "fst_noise_transfer_fwd")) return rc;
  hipLaunchKernelGGL(noise_transfer_fwd_kernel, dim3((unsigned)((L + 63) / 64)), dim3(256), lds, (hipStream_t)stream, p);
  FST_LAUNCH_CHECK();
  return 0;
}

// out[b][i] = x[b][i] + v[i]
__global__ __launch_bounds__(256) void bcast_add_kernel(float* out, const float* x, const float* v, long long N4, long long total4) {
  for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < total4; j += (long long)gridDim.x * 256) {
    const float4 a = reinterpret_cast<const float4*>(x)[j], b = reinterpret_cast<const float4*>(v)[j % N4];
    reinterpret_cast<float4*>(out)[j] = {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w};
  }
}

extern "C" int fst_bcast_add(float* out, const float* x, const float* v, int B, int64_t N, void* stream) {
  FST_REQUIRE(out && x && v && B > 0 && N > 0 && N % 4 == 0, "fst_bcast_add: bad arguments (B=%d N=%lld, N %% 4 == 0)", B, (long long)N);
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  FST_REQUIRE(al16(out) && al16(x) && al16(v), "fst_bcast_add: tensors must be 16-byte aligned");
  const long long total4 = (long long)B * N / 4;
  long long blocks = (total4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(bcast_add_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, x, v, (long long)(N / 4), total4);
  FST_LAUNCH_CHECK();
  return 0;
}

struct NtBwdParams {
  const float* part;          // [S][C·L] slice sums of the cotangent of out
  const float* pre;           // [C][L]
  const float* W;             // [C][C]
  float* dpre;                // [C][L] out: Σ_b g · selu'(pre)
  float* dd;                  // [C][L] out: Wᵀ·dpre  = the cotangent of dist
  int S, C, L;
};

__global__ __launch_bounds__(256) void noise_transfer_bwd_kernel(NtBwdParams p) {
  extern __shared__ float sd[];                            // [C][64] dpre at this workgroup's positions
  const int lx = threadIdx.x & 63, cg = threadIdx.x >> 6, l = blockIdx.x * 64 + lx;
  const int C = p.C, L = p.L, S = p.S;
  const long long N = (long long)C * L;
  if (l < L)
    for (int o = cg; o < C; o += 4) {
      const long long i = (long long)o * L + l;
      float G = 0.f;
      for (int s = 0; s < S; ++s) G += p.part[(long long)s * N + i];
      const float x = p.pre[i];
      const float dp = G * NT_SELU_SCALE * (x > 0.f ? 1.0f : NT_SELU_ALPHA * expf(x));
      p.dpre[i] = dp;
      sd[o * 64 + lx] = dp;
    }
  __syncthreads();
  if (l < L)
    for (int c = cg; c < C; c += 4) {
      float acc = 0.f;
      for (int o = 0; o < C; ++o) acc += p.W[(long long)o * C + c] * sd[o * 64 + lx];
      p.dd[(long long)c * L + l] = acc;
    }
}

extern "C" int fst_noise_transfer_bwd(const float* part, int S, const float* pre, const float* W, float* dpre, float* dd, int C, int L,
                                      void* stream) {
  FST_REQUIRE(part && pre && W && dpre && dd, "fst_noise_transfer_bwd: null operand");
  FST_REQUIRE(S > 0 && C > 0 && C <= 512 && L > 0, "fst_noise_transfer_bwd: bad shape S=%d C=%d L=%d (C <= 512)", S, C, L);
  NtBwdParams p = {part, pre, W, dpre, dd, S, C, L};
  const size_t lds = (size_t)C * 64 * sizeof(float);
  if (lds > 48 * 1024)
    if (int rc = fst_allow_full_lds((const void*)noise_transfer_bwd_kernel, "fst_noise_transfer_bwd")) return rc;
  hipLaunchKernelGGL(noise_transfer_bwd_kernel, dim3((unsigned)((L + 63) / 64)), dim3(256), lds, (hipStream_t)stream, p);
  FST_LAUNCH_CHECK();
  return 0;
}

// dW[o][c] = Σ_l dpre[o][l]·dist[c][l],  dbias[o] = Σ_l dpre[o][l]: one workgroup per output channel o, wave w takes the input
// channels c ≡ w (mod 4), lanes run over time (coalesced), a butterfly per dot product
__global__ __launch_bounds__(256) void noise_transfer_dw_kernel(const float* dpre, const float* dist, float* dW, float* dbias, int C, int L) {
  const int o = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float* dp = dpre + (long long)o * L;
  for (int c = w; c <= C; c += 4) {                        // c == C: the bias row (a row of ones)
    float acc = 0.f;
    if (c < C) {
      const float* x = dist + (long long)c * L;
      for (int l = lane; l < L; l += 64) acc += dp[l] * x[l];
    } else {
      for (int l = lane; l < L; l += 64) acc += dp[l];
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if (lane == 0) {
      if (c < C) dW[(long long)o * C + c] = acc;
      else dbias[o] = acc;
    }
  }
}

extern "C" int fst_noise_transfer_dw(const float* dpre, const float* dist, float* dW, float* dbias, int C, int L, void* stream) {
  FST_REQUIRE(dpre && dist && dW && dbias && C > 0 && L > 0, "fst_noise_transfer_dw: bad arguments");
  hipLaunchKernelGGL(noise_transfer_dw_kernel, dim3((unsigned)C), dim3(256), 0, (hipStream_t)stream, dpre, dist, dW, dbias, C, L);
  FST_LAUNCH_CHECK();
  return 0;
}

// dz_t[b][i] = (r_t/B)·dd[i];   dz_s[b][i] = g[b][i] − (r_s/B)·dd[i]      (either output may be null)
__global__ __launch_bounds__(256) void noise_transfer_bwd_apply_kernel(const float* g, const float* dd, const float* r_t_dev,
                                                                       const float* r_s_dev, float r_t, float r_s, int B, float* dz_t,
                                                                       float* dz_s, long long N4, long long total4) {
  const float fB = (float)B;
  const float kt = (r_t_dev ? *r_t_dev : r_t) / fB, ks = (r_s_dev ? *r_s_dev : r_s) / fB;
  for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < total4; j += (long long)gridDim.x * 256) {
    const float4 d = reinterpret_cast<const float4*>(dd)[j % N4];
    if (dz_t) reinterpret_cast<float4*>(dz_t)[j] = {kt * d.x, kt * d.y, kt * d.z, kt * d.w};
    if (dz_s) {
      const float4 a = reinterpret_cast<const float4*>(g)[j];
      reinterpret_cast<float4*>(dz_s)[j] = {a.x - ks * d.x, a.y - ks * d.y, a.z - ks * d.z, a.w - ks * d.w};
    }
  }
}

extern "C" int fst_noise_transfer_bwd_apply(const float* g, const float* dd, const float* r_t_dev, const float* r_s_dev, float r_t,
                                            float r_s, int B, float* dz_t, float* dz_s, int64_t N, void* stream) {
  FST_REQUIRE(g && dd && (dz_t || dz_s) && B > 0 && N > 0 && N % 4 == 0, "fst_noise_transfer_bwd_apply: bad arguments (B=%d N=%lld)", B,
              (long long)N);
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  FST_REQUIRE(al16(g) && al16(dd) && al16(dz_t) && al16(dz_s), "fst_noise_transfer_bwd_apply: tensors must be 16-byte aligned");
  const long long total4 = (long long)B * N / 4;
  long long blocks = (total4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(noise_transfer_bwd_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, dd, r_t_dev, r_s_dev,
                     r_t, r_s, B, dz_t, dz_s, (long long)(N / 4), total4);
  FST_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- ReLU backward for epilogue-fused ReLUs
// DimensionUnification (widgets.py:73-78 of the reference) runs its two ReLUs in the epilogues of the length GEMM and of the 1x1 conv;
// their backward needs dy·[y > 0] once, in front of the data / weight gradient GEMMs:  out[i] = y[i] > 0 ? dy[i] : 0
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* dy, const float* y, float* out, long long n4, long long n) {
  for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < n4; j += (long long)gridDim.x * 256) {
    const float4 g = reinterpret_cast<const float4*>(dy)[j], v = reinterpret_cast<const float4*>(y)[j];
    reinterpret_cast<float4*>(out)[j] = {v.x > 0.f ? g.x : 0.f, v.y > 0.f ? g.y : 0.f, v.z > 0.f ? g.z : 0.f, v.w > 0.f ? g.w : 0.f};
  }
  if (blockIdx.x == 0)
    for (long long i = 4 * n4 + threadIdx.x; i < n; i += 256) out[i] = y[i] > 0.f ? dy[i] : 0.f;
}

extern "C" int fst_relu_bwd(const float* dy, const float* y, float* out, int64_t n, void* stream) {
  FST_REQUIRE(dy && y && out && n > 0, "fst_relu_bwd: bad arguments");
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  FST_REQUIRE(al16(dy) && al16(y) && al16(out), "fst_relu_bwd: tensors must be 16-byte aligned");
  long long blocks = (n / 4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, y, out, (long long)(n / 4), (long long)n);
  FST_LAUNCH_CHECK();
  return 0;
}
