// GRU recurrence of the CPC context network (Comparison/SLARDA/train.py:65-67: nn.GRU(C, 64, batch_first), h0 = 0) as
// ONE persistent launch per direction instead of MIOpen's step-by-step RNN (≈10 tiny launches per time step: 2.5 k
// launches per CPC call at T/2 = 128 steps, 5 k per train step).
//
// The input projection x_t·W_ihᵀ + b_ih of every step is one GEMM outside (xproj [B][S][3H], gate order r | z | n as in
// torch); what is sequential is   h_t = GRUCell(xproj_t, h_{t-1})   with its 3H×H matrix–vector product per step.
// A workgroup of 3H threads owns GRU_ROWS batch rows for all S steps: thread j keeps row j of W_hh (H floats) in
// registers for the whole kernel, h lives in LDS (broadcast reads), two barriers per step.  Everything is fp32 FMA chains.
//   forward   saves, per step, h_t and the four quantities the backward needs (r, z, n, W_hn·h + b_hn);
//   backward  walks the steps down from t_last (the only step whose h is consumed), thread (g, i) keeping COLUMN i of
//             gate block g of W_hh in registers for the transposed product; emits dxproj (→ d input, dW_ih, db_ih by GEMMs
//             outside) and dgh (→ dW_hh = Σ dgh ⊗ h_{t-1}, db_hh, outside).
// t_last may be given as a DEVICE scalar so that a captured hipGraph can change it between replays; steps beyond it are
// skipped (forward: not computed; backward: their gradients stay at the caller's zeros).
#include "fst_common.h"

#define GRU_H 64
#define GRU_ROWS 2

struct GruParams {
  const float* xproj;   // [B][S][3H]
  const float* w_hh;    // [3H][H]
  const float* b_hh;    // [3H]
  float* h_all;         // [B][S][H]   h after each step
  float* gates;         // [B][S][4H]  r | z | n | hn
  const int* t_last_dev;
  int t_last, B, S;
  // backward
  const float* dout;    // [B][H]  gradient of h at step t_last
  float* dxproj;        // [B][S][3H]  zero-filled by the caller
  float* dgh;           // [B][S][3H]  zero-filled by the caller
};

__device__ __forceinline__ float gru_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(3 * GRU_H) void gru_fwd_kernel(GruParams p) {
  __shared__ float hs[GRU_ROWS][GRU_H];
  __shared__ float dots[GRU_ROWS][3 * GRU_H];
  const int j = threadIdx.x, g = j / GRU_H, i = j - g * GRU_H;
  const int b0 = blockIdx.x * GRU_ROWS;
  const int t_last = p.t_last_dev ? p.t_last_dev[0] : p.t_last;
  float w[GRU_H];
#pragma unroll
  for (int k = 0; k < GRU_H; ++k) w[k] = p.w_hh[(long long)j * GRU_H + k];
  const float bj = p.b_hh[j];
  if (g < GRU_ROWS) hs[g][i] = 0.f;
  __syncthreads();
  // gate phase: thread group g handles batch row b0 + g
  const int b = b0 + g;
  const bool gate_thread = g < GRU_ROWS && b < p.B;
  const float* xp = p.xproj + ((long long)b * p.S) * (3 * GRU_H);
  float xr = 0.f, xz = 0.f, xn = 0.f;
  if (gate_thread && t_last >= 0) { xr = xp[i]; xz = xp[GRU_H + i]; xn = xp[2 * GRU_H + i]; }
  for (int s = 0; s <= t_last && s < p.S; ++s) {
#pragma unroll
    for (int r = 0; r < GRU_ROWS; ++r) {
      float d = bj;
#pragma unroll
      for (int k = 0; k < GRU_H; ++k) d = fmaf(w[k], hs[r][k], d);
      dots[r][j] = d;
    }
    __syncthreads();
    float hnew = 0.f;
    if (gate_thread) {
      const float rr = gru_sigmoid(xr + dots[g][i]);
      const float zz = gru_sigmoid(xz + dots[g][GRU_H + i]);
      const float hn = dots[g][2 * GRU_H + i];
      const float nn = tanhf(xn + rr * hn);
      hnew = (1.f - zz) * nn + zz * hs[g][i];
      float* gt = p.gates + ((long long)b * p.S + s) * (4 * GRU_H);
      gt[i] = rr; gt[GRU_H + i] = zz; gt[2 * GRU_H + i] = nn; gt[3 * GRU_H + i] = hn;
      p.h_all[((long long)b * p.S + s) * GRU_H + i] = hnew;
      if (s + 1 <= t_last && s + 1 < p.S) {              // next step's input projection, in flight during the product
        const float* xq = xp + (long long)(s + 1) * (3 * GRU_H);
        xr = xq[i]; xz = xq[GRU_H + i]; xn = xq[2 * GRU_H + i];
      }
    }
    // every thread left the product phase at the barrier above, and element (g, i) of hs is read by this thread only in
    // the gate phase: it can be replaced right away; the barrier below orders it (and the reads of dots) before the
    // next product
    if (gate_thread) hs[g][i] = hnew;
    __syncthreads();
  }
}

__global__ __launch_bounds__(3 * GRU_H) void gru_bwd_kernel(GruParams p) {
  __shared__ float dh[GRU_ROWS][GRU_H];
  __shared__ float dg[GRU_ROWS][3 * GRU_H];
  __shared__ float part[GRU_ROWS][3 * GRU_H];
  const int j = threadIdx.x, g = j / GRU_H, i = j - g * GRU_H;
  const int b0 = blockIdx.x * GRU_ROWS;
  const int t_last = min(p.S - 1, p.t_last_dev ? p.t_last_dev[0] : p.t_last);
  float wt[GRU_H];                                       // column i of gate block g: W_hh[g*H + k][i]
#pragma unroll
  for (int k = 0; k < GRU_H; ++k) wt[k] = p.w_hh[(long long)(g * GRU_H + k) * GRU_H + i];
  const int b = b0 + g;
  const bool gate_thread = g < GRU_ROWS && b < p.B;
  if (g < GRU_ROWS) dh[g][i] = gate_thread && t_last >= 0 ? p.dout[(long long)b * GRU_H + i] : 0.f;
  __syncthreads();
  for (int s = t_last; s >= 0; --s) {
    float dhp = 0.f;
    if (g < GRU_ROWS) {
      float drp = 0.f, dzp = 0.f, dnp = 0.f, dhn = 0.f;
      if (gate_thread) {
        const float* gt = p.gates + ((long long)b * p.S + s) * (4 * GRU_H);
        const float rr = gt[i], zz = gt[GRU_H + i], nn = gt[2 * GRU_H + i], hn = gt[3 * GRU_H + i];
        const float hprev = s > 0 ? p.h_all[((long long)b * p.S + s - 1) * GRU_H + i] : 0.f;
        const float d = dh[g][i];
        dhp = d * zz;
        dnp = d * (1.f - zz) * (1.f - nn * nn);
        dzp = d * (hprev - nn) * zz * (1.f - zz);
        dhn = dnp * rr;
        drp = dnp * hn * rr * (1.f - rr);
        const long long o = ((long long)b * p.S + s) * (3 * GRU_H);
        p.dxproj[o + i] = drp; p.dxproj[o + GRU_H + i] = dzp; p.dxproj[o + 2 * GRU_H + i] = dnp;
        p.dgh[o + i] = drp; p.dgh[o + GRU_H + i] = dzp; p.dgh[o + 2 * GRU_H + i] = dhn;
      }
      dg[g][i] = drp; dg[g][GRU_H + i] = dzp; dg[g][2 * GRU_H + i] = dhn;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < GRU_ROWS; ++r) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < GRU_H; ++k) a = fmaf(wt[k], dg[r][g * GRU_H + k], a);
      part[r][j] = a;
    }
    __syncthreads();
    if (g < GRU_ROWS) dh[g][i] = dhp + part[g][i] + part[g][GRU_H + i] + part[g][2 * GRU_H + i];
    __syncthreads();
  }
}

static int gru_check(int B, int S, int H, const char* who) {
  FST_REQUIRE(H == GRU_H, "%s: hidden size %d not supported (this kernel is built for %d)", who, H, GRU_H);
  FST_REQUIRE(B > 0 && S > 0, "%s: B=%d S=%d", who, B, S);
  return 0;
}

extern "C" int fst_gru_fwd(const float* xproj, const float* w_hh, const float* b_hh, float* h_all, float* gates,
                           const int32_t* t_last_dev, int t_last, int B, int S, int H, int64_t numel_h, void* stream) {
  if (int rc = gru_check(B, S, H, "fst_gru_fwd")) return rc;
  FST_REQUIRE(xproj && w_hh && b_hh && h_all && gates, "fst_gru_fwd: null operand");
  FST_REQUIRE((long long)B * S * H == (long long)numel_h, "fst_gru_fwd: B*S*H = %d*%d*%d does not match h_all's element count %lld",
              B, S, H, (long long)numel_h);
  FST_REQUIRE(t_last_dev || (t_last >= 0 && t_last < S), "fst_gru_fwd: t_last=%d outside [0,%d)", t_last, S);
  GruParams p = {};
  p.xproj = xproj; p.w_hh = w_hh; p.b_hh = b_hh; p.h_all = h_all; p.gates = gates; p.t_last_dev = t_last_dev;
  p.t_last = t_last; p.B = B; p.S = S;
  hipLaunchKernelGGL(gru_fwd_kernel, dim3((unsigned)((B + GRU_ROWS - 1) / GRU_ROWS)), dim3(3 * GRU_H), 0, (hipStream_t)stream, p);
  FST_LAUNCH_CHECK();
  return 0;
}

extern "C" int fst_gru_bwd(const float* w_hh, const float* h_all, const float* gates, const float* dout,
                           const int32_t* t_last_dev, int t_last, float* dxproj, float* dgh, int B, int S, int H,
                           int64_t numel_h, void* stream) {
  if (int rc = gru_check(B, S, H, "fst_gru_bwd")) return rc;
  FST_REQUIRE(w_hh && h_all && gates && dout && dxproj && dgh, "fst_gru_bwd: null operand");
  FST_REQUIRE((long long)B * S * H == (long long)numel_h, "fst_gru_bwd: B*S*H = %d*%d*%d does not match h_all's element count %lld",
              B, S, H, (long long)numel_h);
  FST_REQUIRE(t_last_dev || (t_last >= 0 && t_last < S), "fst_gru_bwd: t_last=%d outside [0,%d)", t_last, S);
  GruParams p = {};
  p.w_hh = w_hh; p.h_all = const_cast<float*>(h_all); p.gates = const_cast<float*>(gates); p.dout = dout; p.t_last_dev = t_last_dev; p.t_last = t_last;
  p.dxproj = dxproj; p.dgh = dgh; p.B = B; p.S = S;
  hipLaunchKernelGGL(gru_bwd_kernel, dim3((unsigned)((B + GRU_ROWS - 1) / GRU_ROWS)), dim3(3 * GRU_H), 0, (hipStream_t)stream, p);
  FST_LAUNCH_CHECK();
  return 0;
}
