// The 2-step LSTM of ProbTransfer (widgets.py:46-55: nn.LSTM(C, C, batch_first=True) over the pooled feature repeated
// twice, h0 = c0 = 0, only h_n consumed) as ONE launch per direction instead of MIOpen's step-by-step RNN (≈30 small
// launches forward + backward per call, three calls per train step).
//
// Both steps see the same input, so the input projection xproj = x·W_ihᵀ + b_ih + b_hh is one GEMM outside and is used
// twice (torch adds b_hh at every step, also at step 1 where h0 = 0).  Gate order i | f | g | o (torch's).  One workgroup
// per batch row, thread j = one of the 4H gate pre-activations:
//   step 1:  gates1 = act(xproj);            c1 = i1·g1;            h1 = o1·tanh(c1)
//   step 2:  gates2 = act(xproj + W_hh·h1);  c2 = f2·c1 + i2·g2;    h2 = o2·tanh(c2)
// forward saves (gates1, gates2, c1, c2, h1); backward returns dxproj = dpre1 + dpre2 (→ dx, dW_ih, db_ih = db_hh by GEMMs
// outside) and dpre2 (dW_hh = Σ_b dpre2 ⊗ h1).  fp32 FMA chains throughout.
#include "fst_common.h"

struct Lstm2Params {
  const float* xproj;   // [B][4H]
  const float* w_hh_t;  // forward: W_hhᵀ [H][4H] (thread j reads column j: coalesced)
  const float* w_hh;    // backward: W_hh [4H][H]
  float* h2;            // [B][H]
  float* save;          // [B][2·4H + 3H]: gates1 | gates2 | c1 | c2 | h1
  const float* dh2;     // [B][H]
  float* dxproj;        // [B][4H]
  float* dpre2;         // [B][4H]
  int B, H;
};

__device__ __forceinline__ float lstm_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ void lstm2_fwd_kernel(Lstm2Params p) {
  extern __shared__ float sm[];                          // gates [4H] | h1 [H]
  const int H = p.H, H4 = 4 * H, b = blockIdx.x, j = threadIdx.x;
  float* gates = sm;
  float* h1 = sm + H4;
  float* sv = p.save + (long long)b * (2 * H4 + 3 * H);
  const bool live = j < H4;
  const float xp = live ? p.xproj[(long long)b * H4 + j] : 0.f;
  const bool is_g = j >= 2 * H && j < 3 * H;
  // step 1
  if (live) {
    const float a = is_g ? tanhf(xp) : lstm_sigmoid(xp);
    gates[j] = a;
    sv[j] = a;
  }
  __syncthreads();
  float c1 = 0.f;
  if (j < H) {
    c1 = gates[j] * gates[2 * H + j];
    const float h = gates[3 * H + j] * tanhf(c1);
    h1[j] = h;
    sv[2 * H4 + j] = c1;
    sv[2 * H4 + 2 * H + j] = h;
  }
  __syncthreads();
  // step 2
  float pre = xp;
  if (live)
    for (int k = 0; k < H; ++k) pre = fmaf(p.w_hh_t[(long long)k * H4 + j], h1[k], pre);
  __syncthreads();                                       // every thread has read gates (step 1) and h1
  if (live) {
    const float a = is_g ? tanhf(pre) : lstm_sigmoid(pre);
    gates[j] = a;
    sv[H4 + j] = a;
  }
  __syncthreads();
  if (j < H) {
    const float c2 = gates[H + j] * c1 + gates[j] * gates[2 * H + j];
    sv[2 * H4 + H + j] = c2;
    p.h2[(long long)b * H + j] = gates[3 * H + j] * tanhf(c2);
  }
}

__global__ void lstm2_bwd_kernel(Lstm2Params p) {
  extern __shared__ float sm[];                          // dpre2 [4H] | dh1 [H]
  const int H = p.H, H4 = 4 * H, b = blockIdx.x, j = threadIdx.x;
  float* dpre2 = sm;
  float* dh1s = sm + H4;
  const float* sv = p.save + (long long)b * (2 * H4 + 3 * H);
  const float* g1 = sv;
  const float* g2 = sv + H4;
  float dc1 = 0.f;
  if (j < H) {
    const float i2 = g2[j], f2 = g2[H + j], gg2 = g2[2 * H + j], o2 = g2[3 * H + j];
    const float c1 = sv[2 * H4 + j], c2 = sv[2 * H4 + H + j];
    const float d = p.dh2[(long long)b * H + j];
    const float tc = tanhf(c2);
    const float dc2 = d * o2 * (1.f - tc * tc);
    dpre2[j] = dc2 * gg2 * i2 * (1.f - i2);
    dpre2[H + j] = dc2 * c1 * f2 * (1.f - f2);
    dpre2[2 * H + j] = dc2 * i2 * (1.f - gg2 * gg2);
    dpre2[3 * H + j] = d * tc * o2 * (1.f - o2);
    dc1 = dc2 * f2;
  }
  __syncthreads();
  if (j < H) {                                           // dh1 = W_hhᵀ · dpre2 (thread k reads w_hh[m][k]: coalesced across k)
    float a = 0.f;
    for (int m = 0; m < H4; ++m) a = fmaf(p.w_hh[(long long)m * H + j], dpre2[m], a);
    dh1s[j] = a;
  }
  __syncthreads();
  float d1[4] = {0.f, 0.f, 0.f, 0.f};
  if (j < H) {
    const float i1 = g1[j], gg1 = g1[2 * H + j], o1 = g1[3 * H + j];
    const float c1 = sv[2 * H4 + j];
    const float tc = tanhf(c1);
    const float dh = dh1s[j];
    dc1 += dh * o1 * (1.f - tc * tc);
    d1[0] = dc1 * gg1 * i1 * (1.f - i1);                 // c0 = 0: the forget gate of step 1 has no gradient
    d1[2] = dc1 * i1 * (1.f - gg1 * gg1);
    d1[3] = dh * tc * o1 * (1.f - o1);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long long o = (long long)b * H4 + q * H + j;
      p.dxproj[o] = d1[q] + dpre2[q * H + j];
      p.dpre2[o] = dpre2[q * H + j];
    }
  }
}

static int lstm2_check(int B, int H, const char* who) {
  FST_REQUIRE(B > 0 && H > 0 && H <= 256, "%s: B=%d H=%d (needs 1 <= H <= 256: one thread per gate row)", who, B, H);
  return 0;
}
static unsigned lstm2_threads(int H) { return (unsigned)((4 * H + 63) / 64 * 64); }

extern "C" int fst_lstm2_fwd(const float* xproj, const float* w_hh_t, float* h2, float* save, int B, int H, int64_t numel_xproj,
                             void* stream) {
  if (int rc = lstm2_check(B, H, "fst_lstm2_fwd")) return rc;
  FST_REQUIRE(xproj && w_hh_t && h2 && save, "fst_lstm2_fwd: null operand");
  FST_REQUIRE((long long)B * 4 * H == (long long)numel_xproj, "fst_lstm2_fwd: B*4H = %d*%d does not match xproj's element count %lld",
              B, 4 * H, (long long)numel_xproj);
  Lstm2Params p = {};
  p.xproj = xproj; p.w_hh_t = w_hh_t; p.h2 = h2; p.save = save; p.B = B; p.H = H;
  hipLaunchKernelGGL(lstm2_fwd_kernel, dim3((unsigned)B), dim3(lstm2_threads(H)), (size_t)5 * H * sizeof(float), (hipStream_t)stream, p);
  FST_LAUNCH_CHECK();
  return 0;
}

extern "C" int fst_lstm2_bwd(const float* w_hh, const float* save, const float* dh2, float* dxproj, float* dpre2, int B, int H,
                             int64_t numel_xproj, void* stream) {
  if (int rc = lstm2_check(B, H, "fst_lstm2_bwd")) return rc;
  FST_REQUIRE(w_hh && save && dh2 && dxproj && dpre2, "fst_lstm2_bwd: null operand");
  FST_REQUIRE((long long)B * 4 * H == (long long)numel_xproj, "fst_lstm2_bwd: B*4H = %d*%d does not match dxproj's element count %lld",
              B, 4 * H, (long long)numel_xproj);
  Lstm2Params p = {};
  p.w_hh = w_hh; p.save = const_cast<float*>(save); p.dh2 = dh2; p.dxproj = dxproj; p.dpre2 = dpre2; p.B = B; p.H = H;
  hipLaunchKernelGGL(lstm2_bwd_kernel, dim3((unsigned)B), dim3(lstm2_threads(H)), (size_t)5 * H * sizeof(float), (hipStream_t)stream, p);
  FST_LAUNCH_CHECK();
  return 0;
}
