// Conv engine for gfx950: every convolution on the train-step hot path (omni-scale prime-kernel
// layers, 1x1 shortcuts, WaveGlow dilated/1x1 convs, their data gradients and weight gradients) is
// one of the kernels in this file, driven by a small "plan" table (plan.py):
//
//   y[b,m,t]  = Σ_k A[m,k] · xcol[b,k,t]         forward and data gradient
//     conv_gemm_bf3_kernel   single-tap 16-channel stages through a 3-slot LDS-DMA ring          (split-bf16 MFMA)
//     conv_win_bf3_kernel    omni-scale layers: one [time][16 ch] window per chunk, tap = row     (split-bf16 MFMA)
//     conv_gemm_pipe_kernel  the f32 form of the stages (L % 4 != 0, FST_MATH=f32)                (f32 MFMA)
//     conv_gemm_kernel       the f32 form of the window (C_in < 8, FST_MATH=f32)                  (f32 MFMA)
//   dA[m,k] += Σ_{b,t} dy[b,m,t] · xcol[b,k,t]   weight gradient, split over (b,t), fp32 atomics
//     conv_wgrad_kernel<..., BF3>                                                        (split-bf16 or f32 MFMA)
//
// Two arithmetics.  f32: v_mfma_f32_32x32x2_f32 (exact fp32 chain, 64 FLOP/clk/SIMD).  Split-bf16 ("bf16x3"):
// every fp32 operand v = hi + lo with hi = bf16(v), lo = bf16(v - hi) (round to nearest), a product formed as
// hi·hi + hi·lo + lo·hi by three v_mfma_f32_32x32x16_bf16 with fp32 accumulation — relative error of a product
// <= 3·2^-18 (measured 5e-6 of the output scale) at a third of the 2.5 PFLOP/s bf16 peak instead of 157 TFLOP/s.
// Weights are split when packed (fst_pack_weights_bf16x3), activations in registers on their way to the MFMA.
//
// Shared ideas: lane <-> output row / time sample so LDS reads are conflict-free; a tap is an LDS offset; per
// 32-row block only live taps are multiplied; the plan table is a separate __restrict__ kernel argument so it is
// read with scalar loads; one branch-free epilogue (bias / residual / accumulate / split / atomics) with 16-byte
// stores through a wave-private LDS transpose.
#include "fst_common.h"
#include <type_traits>

// ------------------------------------------------------------------------------------------------
// forward / data-gradient
// ------------------------------------------------------------------------------------------------
struct ConvGemmParams {
  const float* x[2];
  long long x_bs[2];
  const float* a;
  const int32_t* plan;
  const float* bias;
  float* y;
  long long y_bs;
  const float* res;
  long long res_bs;
  float* y2;
  long long y2_bs;
  int msplit, m2_start;
  int B, L, M;
  int tiles_per_seq, ksplit, flags, mg_per_wg, ldw;
  int stage_vec;     // 1: inputs are 16-B aligned (L, strides, bases): windows may be staged with float4 loads
  int epi_vec;       // 1: 16-byte epilogue through a wave-private LDS transpose (all outputs 16-B aligned)
  int epi_lds_off;   // float offset of the 4 x [32][36] transpose tiles in dynamic LDS
};

// Epilogue shared by both forward kernels.  C/D layout of the 32x32 MFMA: col = lane&31 (time),
// row = (r&3) + 8*(r>>2) + 4*(lane>>5).  All loads of one 32-row block (residual / accumulate operands) are
// issued before any store of that block: the output may alias them, so interleaving would serialise one
// load→add→store chain per element on memory latency.
// Epilogue of one 32-row block.  C/D layout of the 32x32 MFMA: col = lane&31 (time), row = (r&3) + 8*(r>>2) +
// 4*(lane>>5).  MODE 0: plain store; 1: store with residual / accumulate operands; 2: fp32 atomics (K split).
// MODE 1 is written branch-free: every operand load is unconditional — a lane that needs no operand reads a
// fixed safe address instead — so hipcc issues the 8×NB×2 loads of a batch back to back and waits once.  (With
// predicated loads it emits one exec-mask branch and one s_waitcnt vmcnt(0) PER LOAD: the res_skip forward then
// runs at a quarter of the speed.)  Addresses are a wave-uniform row base plus ONE per-lane offset.
template <int MB, int NB, int MODE>
__device__ __forceinline__ void conv_epilogue_block(const ConvGemmParams& p, f32x16 (&accb)[NB], int mb, int g, int b,
                                                    int t0, int wave_n0, int half, int l31, bool add_bias) {
  const int L = p.L;
  const int vo = half * 4 * L + wave_n0 + l31;
  const float* safe = p.y2 ? p.y2 : p.y;
  const bool has_res = p.res != nullptr, acc1 = (p.flags & FST_EPI_ACC1) != 0, acc2 = (p.flags & FST_EPI_ACC2) != 0;
#pragma unroll
  for (int rh = 0; rh < 16; rh += 8) {                 // batches of 8 accumulator registers
    float ea[8][NB], eb[8][NB];
    if (MODE == 1) {
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int r = rh + rr;
        const int m_lo = (g * MB + mb) * 32 + (r & 3) + 8 * (r >> 2);   // wave-uniform; this lane's row is m_lo + 4*half
        const int m = m_lo + 4 * half;
        const bool first = m < p.msplit, second = m >= p.m2_start && m < p.M;
        const float* res_row = p.res + ((long long)b * p.res_bs + (long long)m_lo * L + t0);
        const float* y_row = p.y + ((long long)b * p.y_bs + (long long)m_lo * L + t0);
        const float* y2_row = p.y2 + ((long long)b * p.y2_bs + (long long)(m_lo - p.m2_start) * L + t0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const bool t_ok = t0 + wave_n0 + nb * 32 + l31 < L;
          const bool use_a = t_ok && first && has_res;
          const bool use_b = t_ok && ((first && acc1) || (second && acc2));
          const float* pa = use_a ? res_row + vo + nb * 32 : safe;
          const float* pb = use_b ? (first ? y_row : y2_row) + vo + nb * 32 : safe;
          const float va = *pa, vb = *pb;
          ea[rr][nb] = use_a ? va : 0.f;
          eb[rr][nb] = use_b ? vb : 0.f;
        }
      }
    }
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int r = rh + rr;
      const int m_lo = (g * MB + mb) * 32 + (r & 3) + 8 * (r >> 2);
      const int m = m_lo + 4 * half;
      const bool first = m < p.msplit, second = m >= p.m2_start && m < p.M;
      if (first || second) {
        const float bias_v = add_bias ? p.bias[m] : 0.f;
        float* dst = first ? p.y + ((long long)b * p.y_bs + (long long)m_lo * L + t0)
                           : p.y2 + ((long long)b * p.y2_bs + (long long)(m_lo - p.m2_start) * L + t0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          if (t0 + wave_n0 + nb * 32 + l31 < L) {
            float v = accb[nb][r] + bias_v;
            if (MODE == 1) v += ea[rr][nb] + eb[rr][nb];
            if (MODE == 2) {
              atomicAdd(dst + vo + nb * 32, v);
            } else {
              if (p.flags & FST_EPI_RELU) v = fmaxf(v, 0.f);
              dst[vo + nb * 32] = v;
            }
          }
        }
      }
    }
  }
}

// 16-byte epilogue of one 32-row block: the accumulator tile (lane = time, registers = rows) is transposed through a
// wave-private LDS tile [32][36] so that each lane owns 4 consecutive time samples of a row: 4 float4 stores (and
// float4 loads of the residual / accumulate operands) per 32x32 tile instead of 16 dword ones, each wave-instruction
// covering 8 rows x 128 contiguous bytes.  Loads of a tile are all issued before its stores.
template <int MB, int NB, int MODE>
__device__ __forceinline__ void conv_epilogue_block_vec(const ConvGemmParams& p, f32x16 (&accb)[NB], int mb, int g, int b,
                                                        int t0, int wave_n0, int half, int l31, bool add_bias,
                                                        float* tile) {
  const int L = p.L, lane = half * 32 + l31;
  const bool has_res = p.res != nullptr, acc1 = (p.flags & FST_EPI_ACC1) != 0, acc2 = (p.flags & FST_EPI_ACC2) != 0;
  const int rrow = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[((r & 3) + 8 * (r >> 2) + 4 * half) * 36 + l31] = accb[nb][r];
    const int t = t0 + wave_n0 + nb * 32 + c4;
    const bool t_ok = t < L;
    float4 v[4], ea[4], eb[4];
    float* dst[4];
    bool live[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = (g * MB + mb) * 32 + rrow + 8 * j;
      const bool first = m < p.msplit, second = m >= p.m2_start && m < p.M;
      live[j] = t_ok && (first || second);
      dst[j] = first ? p.y + ((long long)b * p.y_bs + (long long)m * L + t)
                     : p.y2 + ((long long)b * p.y2_bs + (long long)(m - p.m2_start) * L + t);
      ea[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      eb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (MODE == 1 && live[j]) {
        if (first && has_res) ea[j] = *reinterpret_cast<const float4*>(p.res + ((long long)b * p.res_bs + (long long)m * L + t));
        if ((first && acc1) || (second && acc2)) eb[j] = *reinterpret_cast<const float4*>(dst[j]);
      }
      const float bias_v = (add_bias && (first || second)) ? p.bias[m] : 0.f;
      v[j] = *reinterpret_cast<const float4*>(tile + (rrow + 8 * j) * 36 + c4);
      v[j].x += bias_v; v[j].y += bias_v; v[j].z += bias_v; v[j].w += bias_v;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (!live[j]) continue;
      float4 o = v[j];
      if (MODE == 1) {
        o.x += ea[j].x + eb[j].x; o.y += ea[j].y + eb[j].y; o.z += ea[j].z + eb[j].z; o.w += ea[j].w + eb[j].w;
      }
      if (p.flags & FST_EPI_RELU) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
      *reinterpret_cast<float4*>(dst[j]) = o;
    }
  }
}

// Blocks are expanded by hand so every accumulator index is a compile-time constant (a runtime index would send
// the accumulators to scratch).
template <int MB, int NB, int MODE, bool VECE>
__device__ __forceinline__ void conv_epilogue_mode(const ConvGemmParams& p, f32x16 (&acc)[MB][NB], int g, int b, int t0,
                                                   int wave_n0, int half, int l31, bool add_bias, float* tile) {
#define FST_EPI_BLOCK(I)                                                                                         \
  if constexpr (MB > I) {                                                                                        \
    if constexpr (VECE) conv_epilogue_block_vec<MB, NB, MODE>(p, acc[I], I, g, b, t0, wave_n0, half, l31, add_bias, tile); \
    else conv_epilogue_block<MB, NB, MODE>(p, acc[I], I, g, b, t0, wave_n0, half, l31, add_bias);               \
  }
  FST_EPI_BLOCK(0) FST_EPI_BLOCK(1) FST_EPI_BLOCK(2) FST_EPI_BLOCK(3)
  FST_EPI_BLOCK(4) FST_EPI_BLOCK(5) FST_EPI_BLOCK(6) FST_EPI_BLOCK(7)
#undef FST_EPI_BLOCK
}

// ``lds`` = base of the kernel's dynamic LDS (the transpose tiles live at p.epi_lds_off; the caller guarantees no
// wave still reads anything stored there).
template <int MB, int NB>
__device__ __forceinline__ void conv_epilogue(const ConvGemmParams& p, f32x16 (&acc)[MB][NB], int g, int b, int t0,
                                              int wave_n0, int half, int l31, bool add_bias, float* lds, int tile_idx = -1) {
  float* tile = lds + p.epi_lds_off + (tile_idx >= 0 ? tile_idx : wave_n0 / (NB * 32)) * (32 * 36);
  const bool add = p.res != nullptr || (p.flags & (FST_EPI_ACC1 | FST_EPI_ACC2));
  if (p.flags & FST_EPI_ATOMIC)
    conv_epilogue_mode<MB, NB, 2, false>(p, acc, g, b, t0, wave_n0, half, l31, add_bias, tile);
  else if (p.epi_vec && add)
    conv_epilogue_mode<MB, NB, 1, true>(p, acc, g, b, t0, wave_n0, half, l31, add_bias, tile);
  else if (p.epi_vec)
    conv_epilogue_mode<MB, NB, 0, true>(p, acc, g, b, t0, wave_n0, half, l31, add_bias, tile);
  else if (add)
    conv_epilogue_mode<MB, NB, 1, false>(p, acc, g, b, t0, wave_n0, half, l31, add_bias, tile);
  else
    conv_epilogue_mode<MB, NB, 0, false>(p, acc, g, b, t0, wave_n0, half, l31, add_bias, tile);
}

template <int MB, int NB>
__global__ __launch_bounds__(256) void conv_gemm_kernel(ConvGemmParams p, const int32_t* __restrict__ plan) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int TILE_N = 128 * NB;
  const PlanView pv = plan_view(plan);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.x / p.tiles_per_seq;
  const int t0 = (blockIdx.x - b * p.tiles_per_seq) * TILE_N;
  const int g_begin = blockIdx.y * p.mg_per_wg;
  const int g_end = min(pv.n_mgroups, g_begin + p.mg_per_wg);
  const int q_begin = (int)(((long long)blockIdx.z * pv.n_chunks) / p.ksplit);
  const int q_end = (int)(((long long)(blockIdx.z + 1) * pv.n_chunks) / p.ksplit);
  const int wave_n0 = wave * NB * 32;
  const int ldw = p.ldw, L = p.L, dil = pv.dil;

  // staged window of chunk q, in units of LDS columns relative to (t0 - pad_left): [jlo, jlo+width)
  auto window = [&](int q, int g0, int g1, int& jlo, int& width) {
    int lo = 1 << 30, hi = -1;
    for (int g = g0; g < g1; ++g) {
      const int32_t* e = pv.mg + 4 * (g * pv.n_chunks + q);
      if (e[1] > e[0]) { lo = min(lo, e[0]); hi = max(hi, e[1]); }
    }
    if (hi < 0) { jlo = 0; width = 0; return; }
    jlo = lo * dil;
    width = (hi - 1 - lo) * dil + TILE_N;
  };
  auto stage = [&](int q, int jlo, int width) {
    const int32_t* c = pv.chunk + 4 * q;
    const int src = c[0], c_begin = c[1], c_count = c[2];
    const int c_pad = (c_count + 1) & ~1;
    const float* xb = p.x[src] + (long long)b * p.x_bs[src] + (long long)c_begin * L;
    const int tbase = t0 - pv.pad_left + jlo;
    if (p.stage_vec && (tbase & 3) == 0 && (width & 3) == 0 && (ldw & 3) == 0) {
      // 16-byte staging: L, strides and bases are multiples of 4 floats (host-checked), so a float4 is entirely
      // inside or outside [0, L)
      const int w4 = width >> 2;
      for (int cc = wave; cc < c_pad; cc += 4) {
        const float* row = xb + (long long)cc * L;
        float4* dst = reinterpret_cast<float4*>(lds + cc * ldw);
        const bool live = cc < c_count;
        for (int j4 = lane; j4 < w4; j4 += 64) {
          const int t = tbase + 4 * j4;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (live && t >= 0 && t < L) v = *reinterpret_cast<const float4*>(row + t);
          dst[j4] = v;
        }
      }
      return;
    }
    for (int cc = wave; cc < c_pad; cc += 4) {
      const float* row = xb + (long long)cc * L;
      float* dst = lds + cc * ldw;
      const bool live = cc < c_count;
      for (int j = lane; j < width; j += 64) {
        const int t = tbase + j;
        dst[j] = (live && t >= 0 && t < L) ? row[t] : 0.f;
      }
    }
  };

  const bool stage_once = (q_end - q_begin == 1);
  int jlo = 0, width = 0;
  if (stage_once) {
    window(q_begin, g_begin, g_end, jlo, width);
    stage(q_begin, jlo, width);
    __syncthreads();
  }

  for (int g = g_begin; g < g_end; ++g) {
    f32x16 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;

    for (int q = q_begin; q < q_end; ++q) {
      const int32_t* e = pv.mg + 4 * (g * pv.n_chunks + q);
      const int lo = e[0], hi = e[1];
      if (!stage_once) {
        window(q, g, g + 1, jlo, width);
        __syncthreads();   // previous chunk's readers are done
        stage(q, jlo, width);
        __syncthreads();
      }
      if (hi <= lo) continue;
      const int c_pad = (pv.chunk[4 * q + 2] + 1) & ~1;
      const int half_c = c_pad >> 1;
      const int nk = (hi - lo) * half_c;                     // k-steps of this (M-group, chunk): tap-major
      const float* ap = p.a + (long long)e[2] * (MB * 64) + lane;
      // The A records stream from L2 (≈500+ cycles) while a k-step is only MB*NB*64 cycles of MFMA: keep RING
      // k-steps of A in flight in a register ring (statically indexed through the unrolled inner loop).
      constexpr int RING = (MB * NB >= 8) ? 2 : (MB * NB >= 4 ? 4 : 8);
      float ring[RING][MB];
#pragma unroll
      for (int j = 0; j < RING; ++j) {
        const int kk = j < nk ? j : nk - 1;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) ring[j][mb] = ap[((long long)kk * MB + mb) * 64];
      }
      int tap = lo, cp = 0;
      const float* bbase = lds + half * ldw - jlo + wave_n0 + l31;
      auto kstep = [&](const float (&av)[MB]) {
        float bv[NB];
        const float* bp = bbase + tap * dil + 2 * cp * ldw;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) bv[nb] = bp[nb * 32];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mb], bv[nb], acc[mb][nb], 0, 0, 0);
        if (++cp == half_c) { cp = 0; ++tap; }
      };
      int kb = 0;
      for (; kb + RING <= nk; kb += RING) {                   // full blocks: no branches around the ring loads
#pragma unroll
        for (int j = 0; j < RING; ++j) {
          float av[MB];
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) av[mb] = ring[j][mb];
          const int kn = kb + j + RING < nk ? kb + j + RING : nk - 1;   // refill this slot (clamped at the tail)
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) ring[j][mb] = ap[((long long)kn * MB + mb) * 64];
          kstep(av);
        }
      }
#pragma unroll
      for (int j = 0; j < RING; ++j) {                        // tail: operands are already in the ring
        if (kb + j < nk) {
          float av[MB];
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) av[mb] = ring[j][mb];
          kstep(av);
        }
      }
    }

    conv_epilogue<MB, NB>(p, acc, g, b, t0, wave_n0, half, l31, p.bias != nullptr && blockIdx.z == 0, lds);
  }
}

#ifdef FST_STAMPS
// Diagnostic build only (tools/build_stamps.sh): per-phase cycle sums of the pipelined kernel, lane 0 of every wave.
__device__ unsigned long long fst_stamps[8];
__device__ __forceinline__ unsigned long long fst_now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
extern "C" int fst_debug_stamps(unsigned long long* out_host, int reset) {
  if (out_host) hipMemcpyFromSymbol(out_host, HIP_SYMBOL(fst_stamps), sizeof(unsigned long long) * 8);
  if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(fst_stamps), z, sizeof(z)); }
  return 0;
}
// phase sums live in registers and are flushed ONCE per wave (per-stage atomics would serialise on 8 words and
// sit in vmcnt, i.e. measure themselves)
#define FST_T(var) const unsigned long long var = fst_now()
#define FST_ACC(slot, a, b) fst_sum[slot] += (b) - (a)
#define FST_SUMS unsigned long long fst_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define FST_FLUSH \
  if (lane == 0) for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&fst_stamps[i_], fst_sum[i_])
#else
#define FST_T(var)
#define FST_ACC(slot, a, b)
#define FST_SUMS
#define FST_FLUSH
#endif

// ------------------------------------------------------------------------------------------------
// forward / data-gradient, software-pipelined variant.
//
// Requirements (checked on the host; plan.py builds such plans with chunk_c=16, split_taps=True):
// every live (M-group, chunk) entry is a SINGLE tap of at most PIPE_C channels, i.e. one "stage" =
// at most PIPE_C/2 k-steps.  Both operands of a stage go through LDS: the A records (packed weights,
// MB·256 B per k-step, copied with 16-B loads) and the B tile [PIPE_C][TILE_N] of shifted input rows.
// Stage s+1 is fetched global→registers while stage s is multiplied, then written to the other LDS
// buffer: one barrier per stage, global latency hidden behind 64 MFMAs per wave, and at ≤ 48 KiB LDS
// and < 256 VGPRs two workgroups share a CU so one's barrier is the other's MFMA time.
// ------------------------------------------------------------------------------------------------
#define PIPE_C 16

template <int MB, int NB, bool VEC>
__global__ __launch_bounds__(256, 2) void conv_gemm_pipe_kernel(ConvGemmParams p, const int32_t* __restrict__ plan) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int TILE_N = 128 * NB;
  constexpr int NREC = PIPE_C / 2;
  constexpr int A_FLOATS = NREC * MB * 64;
  constexpr int B_FLOATS = PIPE_C * TILE_N;
  constexpr int AV = (A_FLOATS / 4 + 255) / 256;     // float4 per thread per stage
  constexpr int BV = B_FLOATS / 256;                 // floats per thread per stage
  constexpr int LOG_N = NB == 1 ? 7 : (NB == 2 ? 8 : 9);
  const PlanView pv = plan_view(plan);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.x / p.tiles_per_seq;
  const int t0 = (blockIdx.x - b * p.tiles_per_seq) * TILE_N;
  const int g = blockIdx.y;
  const int q_begin = (int)(((long long)blockIdx.z * pv.n_chunks) / p.ksplit);
  const int q_end = (int)(((long long)(blockIdx.z + 1) * pv.n_chunks) / p.ksplit);
  const int wave_n0 = wave * NB * 32;
  const int L = p.L;
  constexpr int STAGE_FLOATS = A_FLOATS + B_FLOATS;      // buffer b: A at lds + b*STAGE_FLOATS, B right after it
  const int32_t* ent = pv.mg + 4 * (g * pv.n_chunks);

  auto next_live = [&](int q) {
    while (q < q_end && ent[4 * q + 1] <= ent[4 * q]) ++q;
    return q;
  };

  float4 a_st[AV];
  float b_st[BV];
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int jcol = tid & (TILE_N - 1);
  auto fetch = [&](int q) {
    const int32_t* c = pv.chunk + 4 * q;
    const int32_t* e = ent + 4 * q;
    const int c_count = c[2];
    const int nvec = (((c_count + 1) & ~1) / 2) * MB * 16;
    const float4* asrc = reinterpret_cast<const float4*>(p.a + (long long)e[2] * (MB * 64));
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      const int idx = tid + 256 * i;
      a_st[i] = idx < nvec ? asrc[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float* xb = p.x[c[0]] + (long long)b * p.x_bs[c[0]] + (long long)c[1] * L;
    const int tbase = t0 - pv.pad_left + e[0] * pv.dil;
    if (VEC) {
      // 16-byte loads: the host guarantees every tap shift, L, the strides and the base pointers are multiples
      // of 4 floats, so a float4 is entirely inside or entirely outside [0, L).  Element ev = tid + 256*i of the
      // [PIPE_C][TILE_N/4] float4 tile: row = (tid >> (LOG_N-2)) + (1024 >> LOG_N)*i (wave-uniform), column 4*(tid & (TILE_N/4-1)).
      const int row0 = wave_s >> (LOG_N - 8 >= 0 ? LOG_N - 8 : 0);
      const int row_lane = LOG_N - 8 >= 0 ? 0 : ((tid & 63) >> (LOG_N - 2));      // TILE_N=128: a wave spans 2 rows
      const int col = (tid & (TILE_N / 4 - 1)) * 4;
      const int t = tbase + col;
      const bool t_ok = t >= 0 && t < L;
#pragma unroll
      for (int i = 0; i < BV / 4; ++i) {
        const int cc_u = (LOG_N - 8 >= 0 ? row0 : wave_s * 2) + (1024 >> LOG_N) * i;   // wave-uniform part of the row
        const int cc = cc_u + row_lane;
        const float* rp = xb + ((long long)cc_u * L + tbase);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t_ok && cc < c_count) v = *reinterpret_cast<const float4*>(rp + row_lane * L + col);
        b_st[4 * i + 0] = v.x; b_st[4 * i + 1] = v.y; b_st[4 * i + 2] = v.z; b_st[4 * i + 3] = v.w;
      }
      return;
    }
    // element el = tid + 256*i of the [PIPE_C][TILE_N] tile: row = (tid >> LOG_N) + (256 >> LOG_N)*i is
    // wave-uniform, column j = tid & (TILE_N-1) is the only per-lane address part
    const int row0 = wave_s >> (LOG_N - 6);
    const bool t_ok = tbase + jcol >= 0 && tbase + jcol < L;
#pragma unroll
    for (int i = 0; i < BV; ++i) {
      const int cc = row0 + (256 >> LOG_N) * i;
      const float* rp = xb + ((long long)cc * L + tbase);
      b_st[i] = (t_ok && cc < c_count) ? rp[jcol] : 0.f;
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      const int idx = tid + 256 * i;
      if (idx < A_FLOATS / 4) reinterpret_cast<float4*>(lds + buf * STAGE_FLOATS)[idx] = a_st[i];
    }
    if (VEC) {
#pragma unroll
      for (int i = 0; i < BV / 4; ++i)
        reinterpret_cast<float4*>(lds + buf * STAGE_FLOATS + A_FLOATS)[tid + 256 * i] =
            make_float4(b_st[4 * i], b_st[4 * i + 1], b_st[4 * i + 2], b_st[4 * i + 3]);
    } else {
#pragma unroll
      for (int i = 0; i < BV; ++i) lds[buf * STAGE_FLOATS + A_FLOATS + tid + 256 * i] = b_st[i];
    }
  };

  f32x16 acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;

  FST_SUMS;
  FST_T(ts0);
  int q = next_live(q_begin);
  if (q < q_end) {
    fetch(q);
    commit(0);
  }
  __syncthreads();
  FST_T(ts1);
  FST_ACC(0, ts0, ts1);                                   // prologue
  int buf = 0;
  while (q < q_end) {
    FST_T(ta);
    const int nrec = ((pv.chunk[4 * q + 2] + 1) & ~1) / 2;   // read before the prefetch is in flight
    const int qn = next_live(q + 1);
    if (qn < q_end) fetch(qn);
    FST_T(tb);
    FST_ACC(1, ta, tb);                                   // fetch issue
    const float* ap = lds + buf * STAGE_FLOATS + lane;
    const float* bp = lds + buf * STAGE_FLOATS + A_FLOATS + half * TILE_N + wave_n0 + l31;
    for (int r = 0; r < nrec; ++r) {
      float av[MB], bv[NB];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) av[mb] = ap[(r * MB + mb) * 64];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) bv[nb] = bp[2 * r * TILE_N + nb * 32];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mb], bv[nb], acc[mb][nb], 0, 0, 0);
    }
    FST_T(tc);
    FST_ACC(2, tb, tc);                                   // k-steps
#ifdef FST_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    FST_T(td);
    FST_ACC(3, tc, td);                                   // waiting for the prefetched stage
    if (qn < q_end) commit(buf ^ 1);
    FST_T(te);
    FST_ACC(4, td, te);                                   // LDS commit
    __syncthreads();
    FST_T(tf);
    FST_ACC(5, te, tf);                                   // barrier
    q = qn;
    buf ^= 1;
  }
  FST_T(tg);

  conv_epilogue<MB, NB>(p, acc, g, b, t0, wave_n0, half, l31, p.bias != nullptr && blockIdx.z == 0, lds);
  FST_T(th);
  FST_ACC(6, tg, th);                                     // epilogue
  FST_ACC(7, ts0, th);                                    // whole wave
  FST_FLUSH;
}

// ------------------------------------------------------------------------------------------------
// forward / data-gradient on the bf16 matrix cores with split operands ("bf16x3").
//
// Same plans, stages and epilogue as conv_gemm_pipe_kernel; the product of a stage is formed as
// hi(A)·hi(B) + hi(A)·lo(B) + lo(A)·hi(B) by three v_mfma_f32_32x32x16_bf16 (fp32 accumulate), where
// v = hi + lo + O(2^-18 |v|), hi = bf16_rne(v), lo = bf16_rne(v − hi).  One stage (≤ 16 channels of one tap) is
// exactly one 16-deep k-step: 3·MB·NB MFMAs of 32 cycles — far shorter than a global round trip, so operands
// arrive through a 3-slot LDS ring filled by LDS-DMA (global_load_lds_dwordx4: no staging registers, two stages
// in flight while one is multiplied; one raw s_barrier per stage behind a counted vmcnt).
//   A: the image written by fst_pack_weights_bf16x3 (per stage and 32-row block 1 KiB of hi fragments, 1 KiB of
//      lo fragments), copied verbatim, read back with one ds_read_b128 per fragment.
//   B: raw fp32 input samples of the 16-byte aligned window [t4, t4 + TILE_N + 4) of the stage's 16 channels,
//      t4 = tap-shifted tile start rounded down to 4 samples, stored as 1-KiB sub-tiles of 8 channels × 32
//      samples (one wave-instruction each; out-of-range pieces are fetched from a 16-byte zero block behind
//      the A image).  A wave reads its MFMA B fragment (8 channels of one time sample per lane) with eight
//      conflict-free ds_read_b32 at column n + (shift mod 4) — the tap's sub-shift costs nothing — and splits it
//      into hi / lo in registers.
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define FST_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define FST_LDS_VOID(p) ((__attribute__((address_space(3))) void*)(p))

// two floats -> (hi pair, lo pair), each a dword of two round-to-nearest bf16 (first element in the low half)
__device__ __forceinline__ void split_bf16_pair(float a, float b, unsigned& hi, unsigned& lo) {
  const f32x2 v = {a, b};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  const f32x2 r = {a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u)};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Diagnostic builds only (tools/build_exp.sh): FST_EXP is a bit mask that removes one cost at a time from the
// bf3 kernel (wrong results, timing only): 1 no MFMAs, 2 no B loads, 8 no hi/lo split, 16 no A loads.
#ifndef FST_EXP
#define FST_EXP 0
#endif

template <int MB, int NB>
__global__ __launch_bounds__(256, 2) void conv_gemm_bf3_kernel(ConvGemmParams p, const int32_t* __restrict__ plan) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int TILE_N = 128 * NB;
  constexpr int NBLK = TILE_N / 32 + 1;               // 32-sample column blocks per row group (the last takes the sub-shift spill)
  constexpr int GS = NBLK * 1024 + 128;               // bytes per 8-channel row group (+128: the two lane halves hit different banks)
  constexpr int A_BYTES = MB * 2048;
  constexpr int SLOT = A_BYTES + 2 * GS;
  constexpr int NA = A_BYTES / 1024;                  // 1-KiB pieces of A per stage
  constexpr int NI = NA + 2 * NBLK;                   // LDS-DMA wave-instructions per stage
  constexpr int NPW = (NI + 3) / 4;                   // per wave (waves >= NI % 4 issue one fewer when NI % 4 != 0)
  char* const ldsb = reinterpret_cast<char*>(lds);
  const PlanView pv = plan_view(plan);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.x / p.tiles_per_seq;
  const int t0 = (blockIdx.x - b * p.tiles_per_seq) * TILE_N;
  const int g = blockIdx.y;
  const int q_begin = (int)(((long long)blockIdx.z * pv.n_chunks) / p.ksplit);
  const int q_end = (int)(((long long)(blockIdx.z + 1) * pv.n_chunks) / p.ksplit);
  const int wave_n0 = wave * NB * 32;
  const int L = p.L;
  const int32_t* ent = pv.mg + 4 * (g * pv.n_chunks);
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* const zero16 = reinterpret_cast<const char*>(p.a) + (long long)pv.n_stages * A_BYTES;

  auto next_live = [&](int q) {
    while (q < q_end && ent[4 * q + 1] <= ent[4 * q]) ++q;
    return q;
  };

  // LDS-DMA of stage q into ring slot `slot`: piece idx = wave + 4*i; pieces [0, NA) are A, the rest B sub-tiles
  auto issue = [&](int q, int slot) {
    const int32_t* c = pv.chunk + 4 * q;
    const int32_t* e = ent + 4 * q;
    const int c_count = c[2];
    const char* asrc = reinterpret_cast<const char*>(p.a) + (long long)e[3] * A_BYTES;
    const float* xb = p.x[c[0]] + (long long)b * p.x_bs[c[0]] + (long long)c[1] * L;
    const int tbase = t0 - pv.pad_left + e[0] * pv.dil;
    const int t4 = tbase & ~3;
    const bool spill = (tbase & 3) != 0;
    char* const sl = ldsb + slot * SLOT;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int idx = wave_s + 4 * i;
      if (idx >= NI) break;                            // wave-uniform
      if (idx < NA) {
        const char* src = (FST_EXP & 16) ? zero16 : asrc + idx * 1024 + lane * 16;
        __builtin_amdgcn_global_load_lds(FST_GLOBAL_PTR(src), FST_LDS_VOID(sl + idx * 1024), 16, 0, 0);
      } else {
        const int bi = idx - NA;
        const int gq = bi >= NBLK ? 1 : 0, m = bi - gq * NBLK;
        const int row = 8 * gq + (lane >> 3);
        const int t = t4 + 32 * m + 4 * (lane & 7);
        bool ok = row < c_count && t >= 0 && t < L;
        if (m == NBLK - 1) ok = ok && spill && (lane & 7) == 0;
        if (FST_EXP & 2) ok = false;
        const char* src = ok ? reinterpret_cast<const char*>(xb + ((long long)row * L + t)) : zero16;
        __builtin_amdgcn_global_load_lds(FST_GLOBAL_PTR(src), FST_LDS_VOID(sl + A_BYTES + gq * GS + m * 1024), 16, 0, 0);
      }
    }
  };
  // wait until at most `newer` stages issued after the one about to be read are still in flight
  auto wait_stage = [&](int newer) {
    constexpr int R = NI % 4;
    const bool full = R == 0 || wave_s < R;            // this wave issues NPW pieces per stage, else NPW-1
    if (newer == 0) wait_vmcnt<0>();
    else if (newer == 1) { if (full) wait_vmcnt<NPW>(); else wait_vmcnt<NPW - 1>(); }
    else { if (full) wait_vmcnt<2 * NPW>(); else wait_vmcnt<2 * NPW - 2>(); }
  };

  f32x16 acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;

  FST_SUMS;
  FST_T(ts0);
  int q0 = next_live(q_begin);
  int q1 = q0 < q_end ? next_live(q0 + 1) : q_end;
  if (q0 < q_end) issue(q0, 0);
  if (q1 < q_end) issue(q1, 1);
  int slot = 0;
  FST_T(ts1);
  FST_ACC(0, ts0, ts1);                                // prologue
  while (q0 < q_end) {
    FST_T(ta);
    const int q2 = q1 < q_end ? next_live(q1 + 1) : q_end;
    wait_stage(q1 < q_end ? 1 : 0);                    // stage q0 has landed (this wave's pieces) ...
    FST_T(tb);
    FST_ACC(1, ta, tb);                                // next_live + vmcnt wait
    __builtin_amdgcn_s_barrier();                      // ... and everyone's; every wave is done reading the slot refilled next
    FST_T(tc);
    FST_ACC(2, tb, tc);                                // barrier
    if (q2 < q_end) issue(q2, slot >= 1 ? slot - 1 : 2);
    FST_T(td);
    FST_ACC(3, tc, td);                                // LDS-DMA issue
    const int sub = (t0 - pv.pad_left + ent[4 * q0] * pv.dil) & 3;
    const char* base = ldsb + slot * SLOT;
    bf16x8 bh[NB], bl[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int colx = wave_n0 + nb * 32 + l31 + sub;
      const char* bp = base + A_BYTES + half * GS + (colx >> 5) * 1024 + (colx & 31) * 4;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float*>(bp + j * 128);
      u32x4 h, l;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned hh, ll;
        if (FST_EXP & 8) { hh = __float_as_uint(v[2 * j]); ll = __float_as_uint(v[2 * j + 1]); }
        else split_bf16_pair(v[2 * j], v[2 * j + 1], hh, ll);
        h[j] = hh; l[j] = ll;
      }
      bh[nb] = __builtin_bit_cast(bf16x8, h);
      bl[nb] = __builtin_bit_cast(bf16x8, l);
    }
#ifdef FST_STAMPS
    asm volatile("" ::"v"(bh[0]), "v"(bl[0]), "v"(bh[NB - 1]), "v"(bl[NB - 1]));
#endif
    FST_T(te);
    FST_ACC(4, td, te);                                // B fragments: LDS reads + split
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(base + mb * 2048 + lane * 16);
      const bf16x8 al = *reinterpret_cast<const bf16x8*>(base + mb * 2048 + 1024 + lane * 16);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        if (FST_EXP & 1) { asm volatile("" ::"v"(al), "v"(ah), "v"(bh[nb]), "v"(bl[nb])); continue; }
        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[nb], acc[mb][nb], 0, 0, 0);
        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[nb], acc[mb][nb], 0, 0, 0);
        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[nb], acc[mb][nb], 0, 0, 0);
      }
    }
    FST_T(tf);
    FST_ACC(5, te, tf);                                // A fragments + MFMAs (issue)
    q0 = q1;
    q1 = q2;
    slot = slot == 2 ? 0 : slot + 1;
  }
  FST_T(tg);
  // all LDS-DMA has landed (the last wait was vmcnt(0)); the epilogue reuses the ring as transpose tiles once every
  // wave is past its last fragment read
  __syncthreads();
  conv_epilogue<MB, NB>(p, acc, g, b, t0, wave_n0, half, l31, p.bias != nullptr && blockIdx.z == 0, lds);
  FST_T(th);
  FST_ACC(6, tg, th);                                  // epilogue
  FST_ACC(7, ts0, th);                                 // whole wave
  FST_FLUSH;
}

// ------------------------------------------------------------------------------------------------
// omni-scale layers on the bf16 matrix cores (split operands, see conv_gemm_bf3_kernel).
//
// The window idea of conv_gemm_kernel with 16-channel chunks: a chunk's input window is staged ONCE per
// workgroup, already split, as two bf16 images laid out [time][16 channels] (32-byte rows).  A tap is then a ROW
// offset: the MFMA B fragment of (tap, 32 time samples) — 8 channels of one sample per lane — is one 16-byte
// aligned ds_read_b128 per image whatever the tap (a wave's 64 lanes read 2 KiB contiguous), with no VALU work
// in the tap loop.  A fragments (the hi/lo weight image of fst_pack_weights_bf16x3, one 16-deep stage per
// (M-group, chunk, tap)) stream from L2 through a register ring.  Per 32-row block only the block's live taps are
// multiplied; all M-groups of a workgroup share the staged windows when every chunk fits in LDS at once.
// ------------------------------------------------------------------------------------------------
template <int MB, int NB>
__global__ __launch_bounds__(256, 2) void conv_win_bf3_kernel(ConvGemmParams p, const int32_t* __restrict__ plan) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int TILE_N = 128 * NB;
  char* const ldsb = reinterpret_cast<char*>(lds);
  const PlanView pv = plan_view(plan);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.x / p.tiles_per_seq;
  const int t0 = (blockIdx.x - b * p.tiles_per_seq) * TILE_N;
  const int g_begin = blockIdx.y * p.mg_per_wg;
  const int g_end = min(pv.n_mgroups, g_begin + p.mg_per_wg);
  const int q_begin = (int)(((long long)blockIdx.z * pv.n_chunks) / p.ksplit);
  const int q_end = (int)(((long long)(blockIdx.z + 1) * pv.n_chunks) / p.ksplit);
  const int wave_n0 = wave * NB * 32;
  const int L = p.L, dil = pv.dil;
  const int slot_bytes = p.ldw * 64;                    // one chunk: hi image [ldw rows][32 B], then lo image

  auto window = [&](int q, int g0, int g1, int& jlo, int& width) {
    int lo = 1 << 30, hi = -1;
    for (int g = g0; g < g1; ++g) {
      const int32_t* e = pv.mg + 4 * (g * pv.n_chunks + q);
      if (e[1] > e[0]) { lo = min(lo, e[0]); hi = max(hi, e[1]); }
    }
    if (hi < 0) { jlo = 0; width = 0; return; }
    jlo = lo * dil;
    width = (hi - 1 - lo) * dil + TILE_N;
  };
  // stage chunk q: thread = (time j, channel quad cq): 4 channels of one sample -> 8 bytes of each image row
  auto stage = [&](int q, int slot, int jlo, int width) {
    const int32_t* c = pv.chunk + 4 * q;
    const int src = c[0], c_begin = c[1], c_count = c[2];
    const float* xb = p.x[src] + (long long)b * p.x_bs[src] + (long long)c_begin * L;
    const int tbase = t0 - pv.pad_left + jlo;
    char* const hi_img = ldsb + slot * slot_bytes;
    char* const lo_img = hi_img + p.ldw * 32;
    const int cq = tid & 3;
    for (int j = tid >> 2; j < width; j += 64) {
      const int t = tbase + j;
      const bool t_ok = t >= 0 && t < L;
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int cc = 4 * cq + k;
        v[k] = (t_ok && cc < c_count) ? xb[(long long)cc * L + t] : 0.f;
      }
      unsigned h0, h1, l0, l1;
      split_bf16_pair(v[0], v[1], h0, l0);
      split_bf16_pair(v[2], v[3], h1, l1);
      *reinterpret_cast<uint2*>(hi_img + j * 32 + cq * 8) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(lo_img + j * 32 + cq * 8) = make_uint2(l0, l1);
    }
  };

  // every chunk of this K range resident at once (host-decided): stage them all, then every M-group reuses them
  const bool resident = p.stage_vec != 0;               // (field reused: 1 = all chunks fit)
  int jlo_all = 0, width_all = 0;
  if (resident) {
    for (int q = q_begin; q < q_end; ++q) {
      int jl, w;
      window(q, g_begin, g_end, jl, w);
      // one common origin for all chunks: the union window
      if (q == q_begin) { jlo_all = jl; width_all = w; }
      else if (w > 0) {
        const int lo2 = min(jlo_all, jl), hi2 = max(jlo_all + width_all, jl + w);
        if (width_all == 0) { jlo_all = jl; width_all = w; } else { jlo_all = lo2; width_all = hi2 - lo2; }
      }
    }
    for (int q = q_begin; q < q_end; ++q) stage(q, q - q_begin, jlo_all, width_all);
    __syncthreads();
  }

  for (int g = g_begin; g < g_end; ++g) {
    f32x16 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;

    for (int q = q_begin; q < q_end; ++q) {
      const int32_t* e = pv.mg + 4 * (g * pv.n_chunks + q);
      const int lo = e[0], hi = e[1];
      int jlo = jlo_all, slot = q - q_begin;
      if (!resident) {
        int width;
        window(q, g, g + 1, jlo, width);
        slot = 0;
        __syncthreads();   // previous chunk's readers are done
        stage(q, 0, jlo, width);
        __syncthreads();
      }
      if (hi <= lo) continue;
      const int nk = hi - lo;                                // one 16-deep k-step per live tap
      const uint4* ap = reinterpret_cast<const uint4*>(p.a) + (long long)e[3] * (MB * 128) + lane;
      constexpr int RING = MB == 1 ? 4 : 2;
      uint4 rh[RING][MB], rl[RING][MB];
#pragma unroll
      for (int j = 0; j < RING; ++j) {
        const int kk = j < nk ? j : nk - 1;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          rh[j][mb] = ap[(kk * MB + mb) * 128];
          rl[j][mb] = ap[(kk * MB + mb) * 128 + 64];
        }
      }
      const char* hi_img = ldsb + slot * slot_bytes + (wave_n0 + l31 - jlo) * 32 + half * 16;
      const char* lo_img = hi_img + p.ldw * 32;
      int tap = lo;
      auto kstep = [&](const uint4 (&ah)[MB], const uint4 (&al)[MB]) {
        const int roff = tap * dil * 32;
        bf16x8 bh[NB], bl[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          bh[nb] = *reinterpret_cast<const bf16x8*>(hi_img + roff + nb * 1024);
          bl[nb] = *reinterpret_cast<const bf16x8*>(lo_img + roff + nb * 1024);
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const bf16x8 a_h = __builtin_bit_cast(bf16x8, ah[mb]), a_l = __builtin_bit_cast(bf16x8, al[mb]);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_l, bh[nb], acc[mb][nb], 0, 0, 0);
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, bl[nb], acc[mb][nb], 0, 0, 0);
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, bh[nb], acc[mb][nb], 0, 0, 0);
          }
        }
        ++tap;
      };
      int kb = 0;
      for (; kb + RING <= nk; kb += RING) {
#pragma unroll
        for (int j = 0; j < RING; ++j) {
          uint4 ah[MB], al[MB];
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) { ah[mb] = rh[j][mb]; al[mb] = rl[j][mb]; }
          const int kn = kb + j + RING < nk ? kb + j + RING : nk - 1;
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) {
            rh[j][mb] = ap[(kn * MB + mb) * 128];
            rl[j][mb] = ap[(kn * MB + mb) * 128 + 64];
          }
          kstep(ah, al);
        }
      }
#pragma unroll
      for (int j = 0; j < RING; ++j) {
        if (kb + j < nk) {
          uint4 ah[MB], al[MB];
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) { ah[mb] = rh[j][mb]; al[mb] = rl[j][mb]; }
          kstep(ah, al);
        }
      }
    }

    conv_epilogue<MB, NB>(p, acc, g, b, t0, wave_n0, half, l31, p.bias != nullptr && blockIdx.z == 0, lds);
  }
}

// ------------------------------------------------------------------------------------------------
// The same window kernel with the waves split over OUTPUT ROWS instead of time (32-row M-groups, resident windows):
// wave w takes the M-groups g ≡ w (mod 4), paired small-with-large so the four waves carry similar tap counts, against ALL
// eight 32-sample column blocks of the 256-sample tile.  In conv_win_bf3_kernel<1, 2> every wave streams the weight
// fragments of every M-group for its own 64 samples: four copies of the stream per workgroup through the vector-memory
// pipe — 68 of the layer-1 kernel's 216 us (cost removal: profiles/r04_win_cost_removal.txt).  Here a fragment pair is
// loaded by exactly one wave and feeds 24 MFMAs instead of 6; the waves share nothing but the read-only windows, so
// after the staging barrier there is no synchronisation at all.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void conv_win_rows_kernel(ConvGemmParams p, const int32_t* __restrict__ plan) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NB = 8, TILE_N = 256;
  char* const ldsb = reinterpret_cast<char*>(lds);
  const PlanView pv = plan_view(plan);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.x / p.tiles_per_seq;
  const int t0 = (blockIdx.x - b * p.tiles_per_seq) * TILE_N;
  const int L = p.L, dil = pv.dil, G = pv.n_mgroups, Q = pv.n_chunks;
  const int slot_bytes = p.ldw * 64;
  // the union window of every (M-group, chunk): one common origin
  int lo_all = 1 << 30, hi_all = -1;
  for (int g = 0; g < G; ++g)
    for (int q = 0; q < Q; ++q) {
      const int32_t* e = pv.mg + 4 * (g * Q + q);
      if (e[1] > e[0]) { lo_all = min(lo_all, e[0]); hi_all = max(hi_all, e[1]); }
    }
  const int jlo = lo_all * dil, width = (hi_all - 1 - lo_all) * dil + TILE_N;
  for (int q = 0; q < Q; ++q) {                          // stage chunk q: thread = (time j, channel quad cq)
    const int32_t* c = pv.chunk + 4 * q;
    const int src = c[0], c_begin = c[1], c_count = c[2];
    const float* xb = p.x[src] + (long long)b * p.x_bs[src] + (long long)c_begin * L;
    const int tbase = t0 - pv.pad_left + jlo;
    char* const hi_img = ldsb + q * slot_bytes;
    char* const lo_img = hi_img + p.ldw * 32;
    const int cq = tid & 3;
    for (int j = tid >> 2; j < width; j += 64) {
      const int t = tbase + j;
      const bool t_ok = t >= 0 && t < L;
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int cc = 4 * cq + k;
        v[k] = (t_ok && cc < c_count) ? xb[(long long)cc * L + t] : 0.f;
      }
      unsigned h0, h1, l0, l1;
      split_bf16_pair(v[0], v[1], h0, l0);
      split_bf16_pair(v[2], v[3], h1, l1);
      *reinterpret_cast<uint2*>(hi_img + j * 32 + cq * 8) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(lo_img + j * 32 + cq * 8) = make_uint2(l0, l1);
    }
  }
  __syncthreads();
  // M-groups of this wave: position i of its list is group 4i + w on even i, 4i + 3 - w on odd i (the tap count grows with
  // the group index: alternating the direction pairs a wave's small groups with large ones)
  for (int i = 0; 4 * i < G; ++i) {
    const int g = 4 * i + ((i & 1) ? 3 - wave : wave);
    if (g >= G) continue;
    f32x16 acc[1][NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][nb][r] = 0.f;
    for (int q = 0; q < Q; ++q) {
      const int32_t* e = pv.mg + 4 * (g * Q + q);
      const int lo = e[0], nk = e[1] - e[0];
      if (nk <= 0) continue;
      const uint4* ap = reinterpret_cast<const uint4*>(p.a) + (long long)e[3] * 128 + lane;
      constexpr int RING = 4;
      uint4 rh[RING], rl[RING];
#pragma unroll
      for (int j = 0; j < RING; ++j) {
        const int kk = j < nk ? j : nk - 1;
        rh[j] = ap[kk * 128];
        rl[j] = ap[kk * 128 + 64];
      }
      const char* hi_img = ldsb + q * slot_bytes + (l31 - jlo) * 32 + half * 16;
      const char* lo_img = hi_img + p.ldw * 32;
      int tap = lo;
      auto kstep = [&](const uint4 ah, const uint4 al) {
        const int roff = tap * dil * 32;
        const bf16x8 a_h = __builtin_bit_cast(bf16x8, ah), a_l = __builtin_bit_cast(bf16x8, al);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const bf16x8 bh = *reinterpret_cast<const bf16x8*>(hi_img + roff + nb * 1024);
          const bf16x8 bl = *reinterpret_cast<const bf16x8*>(lo_img + roff + nb * 1024);
          acc[0][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_l, bh, acc[0][nb], 0, 0, 0);
          acc[0][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, bl, acc[0][nb], 0, 0, 0);
          acc[0][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, bh, acc[0][nb], 0, 0, 0);
        }
        ++tap;
      };
      int kb = 0;
      for (; kb + RING <= nk; kb += RING) {
#pragma unroll
        for (int j = 0; j < RING; ++j) {
          const uint4 ah = rh[j], al = rl[j];
          const int kn = kb + j + RING < nk ? kb + j + RING : nk - 1;
          rh[j] = ap[kn * 128];
          rl[j] = ap[kn * 128 + 64];
          kstep(ah, al);
        }
      }
#pragma unroll
      for (int j = 0; j < RING; ++j)
        if (kb + j < nk) kstep(rh[j], rl[j]);
    }
    // plain stores only (the launcher sends residual / accumulate / atomic epilogues to conv_win_bf3_kernel): the other modes'
    // operand arrays for eight column blocks would cost this kernel its registers
    float* tile = lds + p.epi_lds_off + wave * (32 * 36);
    if (p.epi_vec) conv_epilogue_mode<1, NB, 0, true>(p, acc, g, b, t0, 0, half, l31, p.bias != nullptr, tile);
    else conv_epilogue_mode<1, NB, 0, false>(p, acc, g, b, t0, 0, half, l31, p.bias != nullptr, tile);
  }
}

static bool plan_is_pipeable(const PlanView& pv) {
  for (int q = 0; q < pv.n_chunks; ++q) {
    if (((pv.chunk[4 * q + 2] + 1) & ~1) > PIPE_C) return false;
    for (int g = 0; g < pv.n_mgroups; ++g) {
      const int32_t* e = pv.mg + 4 * (g * pv.n_chunks + q);
      if (e[1] > e[0] + 1) return false;
    }
  }
  return true;
}

typedef void (*conv_gemm_fn)(ConvGemmParams, const int32_t*);

static conv_gemm_fn pick_conv_gemm(int MB, int NB) {
  if (NB == 1) {
    switch (MB) {
      case 1: return conv_gemm_kernel<1, 1>;
      case 2: return conv_gemm_kernel<2, 1>;
      case 4: return conv_gemm_kernel<4, 1>;
      case 8: return conv_gemm_kernel<8, 1>;
    }
  } else if (NB == 2) {
    switch (MB) {
      case 1: return conv_gemm_kernel<1, 2>;
      case 2: return conv_gemm_kernel<2, 2>;
      case 4: return conv_gemm_kernel<4, 2>;
    }
  } else if (NB == 4) {
    switch (MB) {
      case 1: return conv_gemm_kernel<1, 4>;
      case 2: return conv_gemm_kernel<2, 4>;
    }
  }
  return nullptr;
}

template <bool VEC>
static conv_gemm_fn pick_conv_gemm_pipe_v(int MB, int NB) {
  if (NB == 1) {
    switch (MB) {
      case 1: return conv_gemm_pipe_kernel<1, 1, VEC>;
      case 2: return conv_gemm_pipe_kernel<2, 1, VEC>;
      case 4: return conv_gemm_pipe_kernel<4, 1, VEC>;
      case 8: return conv_gemm_pipe_kernel<8, 1, VEC>;
    }
  } else if (NB == 2) {
    switch (MB) {
      case 1: return conv_gemm_pipe_kernel<1, 2, VEC>;
      case 2: return conv_gemm_pipe_kernel<2, 2, VEC>;
      case 4: return conv_gemm_pipe_kernel<4, 2, VEC>;
    }
  }
  return nullptr;
}
static conv_gemm_fn pick_conv_gemm_pipe(int MB, int NB, bool vec = false) {
  return vec ? pick_conv_gemm_pipe_v<true>(MB, NB) : pick_conv_gemm_pipe_v<false>(MB, NB);
}

static conv_gemm_fn pick_conv_win_bf3(int MB, int NB) {
  if (NB == 1) {
    switch (MB) {
      case 1: return conv_win_bf3_kernel<1, 1>;
      case 2: return conv_win_bf3_kernel<2, 1>;
      case 4: return conv_win_bf3_kernel<4, 1>;
    }
  } else if (NB == 2) {
    switch (MB) {
      case 1: return conv_win_bf3_kernel<1, 2>;
      case 2: return conv_win_bf3_kernel<2, 2>;
    }
  } else if (NB == 4) {
    switch (MB) {
      case 1: return conv_win_bf3_kernel<1, 4>;
    }
  }
  return nullptr;                                      // <4, 2> and <2, 4> (8 accumulator tiles + the fragment ring) spilled: not built; the host side never asks for them
}

static conv_gemm_fn pick_conv_gemm_bf3(int MB, int NB) {
  if (NB == 1) {
    switch (MB) {
      case 1: return conv_gemm_bf3_kernel<1, 1>;
      case 2: return conv_gemm_bf3_kernel<2, 1>;
      case 4: return conv_gemm_bf3_kernel<4, 1>;
      case 8: return conv_gemm_bf3_kernel<8, 1>;
    }
  } else if (NB == 2) {
    switch (MB) {
      case 1: return conv_gemm_bf3_kernel<1, 2>;
      case 2: return conv_gemm_bf3_kernel<2, 2>;
      case 4: return conv_gemm_bf3_kernel<4, 2>;
    }
  }
  return nullptr;
}

int fst_check_plan(const int32_t* ph, int plan_len, int M, const char* who) {
  FST_REQUIRE(ph != nullptr && plan_len >= FST_PLAN_HDR, "%s: plan missing or shorter than its header", who);
  FST_REQUIRE(plan_expected_len(ph) == plan_len, "%s: plan length %d != expected %d", who, plan_len,
              plan_expected_len(ph));
  const PlanView pv = plan_view(ph);
  FST_REQUIRE(pv.n_chunks > 0 && pv.n_mgroups > 0 && pv.ntaps > 0 && pv.dil > 0, "%s: bad plan header", who);
  FST_REQUIRE(pv.MB == 1 || pv.MB == 2 || pv.MB == 4 || pv.MB == 8, "%s: MB=%d unsupported", who, pv.MB);
  FST_REQUIRE(M > 0 && M <= pv.n_mgroups * pv.MB * 32, "%s: M=%d exceeds plan rows %d", who, M,
              pv.n_mgroups * pv.MB * 32);
  FST_REQUIRE((pv.chunk_cap & 1) == 0 && pv.chunk_cap > 0, "%s: chunk_cap must be even", who);
  for (int q = 0; q < pv.n_chunks; ++q) {
    const int32_t* c = pv.chunk + 4 * q;
    FST_REQUIRE((c[0] == 0 || c[0] == 1) && c[1] >= 0 && c[2] > 0 && ((c[2] + 1) & ~1) <= pv.chunk_cap,
                "%s: bad chunk %d", who, q);
    for (int g = 0; g < pv.n_mgroups; ++g) {
      const int32_t* e = pv.mg + 4 * (g * pv.n_chunks + q);
      FST_REQUIRE(e[0] >= 0 && e[1] <= pv.ntaps && e[2] >= 0, "%s: bad tap range at (g=%d,q=%d)", who, g, q);
      const int nrec = e[1] > e[0] ? (e[1] - e[0]) * (((c[2] + 1) & ~1) / 2) : 0;
      FST_REQUIRE(e[2] + nrec <= pv.total_records, "%s: record overflow at (g=%d,q=%d)", who, g, q);
      FST_REQUIRE(e[3] >= 0 && e[3] + (e[1] > e[0] ? e[1] - e[0] : 0) <= pv.n_stages, "%s: stage overflow at (g=%d,q=%d)",
                  who, g, q);
    }
  }
  return 0;
}

extern "C" int fst_conv_gemm(const float* x0, int64_t x0_bs, const float* x1, int64_t x1_bs, const float* a_packed,
                             const int32_t* plan_dev, const int32_t* plan_host, int plan_len, const float* bias,
                             float* y, int64_t y_bs, const float* res, int64_t res_bs, float* y2, int64_t y2_bs,
                             int msplit, int m2_start, int B, int L, int M, int nb_cfg, int ksplit, int flags, void* stream) {
  if (int rc = fst_check_plan(plan_host, plan_len, M, "fst_conv_gemm")) return rc;
  const PlanView pv = plan_view(plan_host);
  FST_REQUIRE(x0 && a_packed && plan_dev, "fst_conv_gemm: null operand");
  FST_REQUIRE(B > 0 && L > 0, "fst_conv_gemm: B=%d L=%d", B, L);
  FST_REQUIRE(msplit >= 0 && msplit <= M, "fst_conv_gemm: msplit=%d outside [0,%d]", msplit, M);
  FST_REQUIRE(msplit == 0 || y != nullptr, "fst_conv_gemm: y is null but msplit=%d", msplit);
  FST_REQUIRE(msplit == M || y2 != nullptr, "fst_conv_gemm: y2 is null but msplit=%d < M=%d", msplit, M);
  FST_REQUIRE(m2_start >= msplit && m2_start <= M, "fst_conv_gemm: m2_start=%d outside [msplit=%d, M=%d]", m2_start, msplit, M);
  FST_REQUIRE(ksplit >= 1 && ksplit <= pv.n_chunks, "fst_conv_gemm: ksplit=%d vs %d chunks", ksplit, pv.n_chunks);
  FST_REQUIRE(ksplit == 1 || (flags & FST_EPI_ATOMIC), "fst_conv_gemm: ksplit>1 needs FST_EPI_ATOMIC");
  bool needs_x1 = false;
  for (int q = 0; q < pv.n_chunks; ++q) needs_x1 |= pv.chunk[4 * q] == 1;
  FST_REQUIRE(!needs_x1 || x1 != nullptr, "fst_conv_gemm: plan reads input 1 but x1 is null");
  // a batch stride shorter than the rows a sample owns would make the B samples of the launch overlap / overrun
  {
    int rows_in[2] = {0, 0};
    for (int q = 0; q < pv.n_chunks; ++q) {
      const int32_t* c = pv.chunk + 4 * q;
      rows_in[c[0]] = rows_in[c[0]] > c[1] + c[2] ? rows_in[c[0]] : c[1] + c[2];
    }
    FST_REQUIRE(B == 1 || (x0_bs >= (int64_t)rows_in[0] * L && (!needs_x1 || x1_bs >= (int64_t)rows_in[1] * L)),
                "fst_conv_gemm: an input batch stride (%lld, %lld) is smaller than its %d / %d channels x L=%d",
                (long long)x0_bs, (long long)x1_bs, rows_in[0], rows_in[1], L);
    FST_REQUIRE(B == 1 || ((msplit == 0 || y_bs >= (int64_t)msplit * L) && (res == nullptr || res_bs >= (int64_t)msplit * L) &&
                           (msplit == M || y2_bs >= (int64_t)(M - m2_start) * L)),
                "fst_conv_gemm: an output batch stride (y %lld, res %lld, y2 %lld) is smaller than the rows it holds",
                (long long)y_bs, (long long)res_bs, (long long)y2_bs);
  }
  const bool pipe = plan_is_pipeable(pv) && pick_conv_gemm_pipe(pv.MB, nb_cfg) != nullptr;
  // 16-byte B-tile loads need every address of a stage row to be 16-B aligned
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const bool shifts4 = pv.pad_left % 4 == 0 && (pv.ntaps == 1 || pv.dil % 4 == 0);   // every tap shift ≡ 0 (mod 4)
  const bool vec = pipe && L % 4 == 0 && shifts4 && x0_bs % 4 == 0 && al16(x0) &&
                   (x1 == nullptr || (x1_bs % 4 == 0 && al16(x1)));
  const bool bf3 = (flags & FST_GEMM_BF16X3) != 0;
  const bool win3 = bf3 && !pipe;                        // windowed plan of <= 16-channel chunks: the bf16 window kernel
  if (bf3 && pipe) {
    FST_REQUIRE(L % 4 == 0 && x0_bs % 4 == 0 && al16(x0) && (x1 == nullptr || (x1_bs % 4 == 0 && al16(x1))),
                "fst_conv_gemm: FST_GEMM_BF16X3 needs L %% 4 == 0 and 16-byte aligned activations (L=%d)", L);
  }
  if (win3)
    FST_REQUIRE(pv.chunk_cap <= 16 && pick_conv_win_bf3(pv.MB, nb_cfg) != nullptr,
                "fst_conv_gemm: FST_GEMM_BF16X3 on a windowed plan needs chunks of <= 16 channels (got %d) and a supported "
                "MB x NB (%d x %d)", pv.chunk_cap, pv.MB, nb_cfg);
  conv_gemm_fn fn = win3 ? pick_conv_win_bf3(pv.MB, nb_cfg)
                         : (bf3 ? pick_conv_gemm_bf3(pv.MB, nb_cfg)
                                : (pipe ? pick_conv_gemm_pipe(pv.MB, nb_cfg, vec) : pick_conv_gemm(pv.MB, nb_cfg)));
  FST_REQUIRE(fn != nullptr, "fst_conv_gemm: no kernel for MB=%d NB=%d", pv.MB, nb_cfg);
  const int TILE_N = 128 * nb_cfg;

  ConvGemmParams p;
  p.x[0] = x0; p.x[1] = x1; p.x_bs[0] = x0_bs; p.x_bs[1] = x1_bs;
  p.a = a_packed; p.plan = plan_dev; p.bias = bias;
  p.y = y; p.y_bs = y_bs; p.res = res; p.res_bs = res_bs; p.y2 = y2; p.y2_bs = y2_bs; p.msplit = msplit;
  p.m2_start = m2_start;
  p.B = B; p.L = L; p.M = M;
  p.tiles_per_seq = (L + TILE_N - 1) / TILE_N;
  p.ksplit = ksplit; p.flags = flags;
  // one staged window feeds every M-group when the whole K range is a single chunk (omni-scale layers)
  p.mg_per_wg = (pv.n_chunks == 1) ? pv.n_mgroups : 1;
  int max_w = 0;
  for (int q = 0; q < pv.n_chunks; ++q) {
    if (p.mg_per_wg > 1) {
      int lo = 1 << 30, hi = -1;
      for (int g = 0; g < pv.n_mgroups; ++g) {
        const int32_t* e = pv.mg + 4 * (g * pv.n_chunks + q);
        if (e[1] > e[0]) { lo = lo < e[0] ? lo : e[0]; hi = hi > e[1] ? hi : e[1]; }
      }
      if (hi >= 0) { int w = (hi - 1 - lo) * pv.dil + TILE_N; max_w = max_w > w ? max_w : w; }
    } else {
      for (int g = 0; g < pv.n_mgroups; ++g) {
        const int32_t* e = pv.mg + 4 * (g * pv.n_chunks + q);
        if (e[1] > e[0]) { int w = (e[1] - 1 - e[0]) * pv.dil + TILE_N; max_w = max_w > w ? max_w : w; }
      }
    }
  }
  int max_w_single = 0;
  for (int q = 0; q < pv.n_chunks; ++q)
    for (int g = 0; g < pv.n_mgroups; ++g) {
      const int32_t* e = pv.mg + 4 * (g * pv.n_chunks + q);
      if (e[1] > e[0]) { int w = (e[1] - 1 - e[0]) * pv.dil + TILE_N; max_w_single = max_w_single > w ? max_w_single : w; }
    }
  FST_REQUIRE(max_w > 0, "fst_conv_gemm: plan has no live taps");
  p.ldw = (max_w + 3) & ~3;                               // 16-B aligned LDS rows (float4 staging)
  size_t lds_bytes = (size_t)pv.chunk_cap * p.ldw * sizeof(float);
  auto al16o = [](const void* q) { return q == nullptr || (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  p.stage_vec = L % 4 == 0 && x0_bs % 4 == 0 && al16o(x0) && (x1 == nullptr || (x1_bs % 4 == 0 && al16o(x1)));
  p.epi_vec = !(flags & FST_EPI_ATOMIC) && L % 4 == 0 && y_bs % 4 == 0 && y2_bs % 4 == 0 && res_bs % 4 == 0 &&
              al16o(y) && al16o(y2) && al16o(res);
  const size_t epi_bytes = 4 * 32 * 36 * sizeof(float);     // one transpose tile per wave
  if (pipe) {
    p.mg_per_wg = 1;
    lds_bytes = bf3 ? 3 * ((size_t)pv.MB * 2048 + 2 * ((size_t)(TILE_N / 32 + 1) * 1024 + 128))
                    : 2 * ((size_t)(PIPE_C / 2) * pv.MB * 64 + (size_t)PIPE_C * TILE_N) * sizeof(float);
    p.epi_lds_off = 0;                                       // the staging buffers are dead after the last barrier
    if (lds_bytes < epi_bytes) lds_bytes = epi_bytes;
  } else if (win3) {
    // bf16 window images: per chunk [ldw rows][16 ch] hi + lo = ldw*64 bytes.  All chunks resident (every M-group of
    // the workgroup reuses them) when they fit next to a second workgroup; otherwise one slot, one M-group per
    // workgroup.  max_w above was computed for mg_per_wg == 1 or a single chunk; recompute for the union window.
    int lo_all = 1 << 30, hi_all = -1;
    for (int q = 0; q < pv.n_chunks; ++q)
      for (int g = 0; g < pv.n_mgroups; ++g) {
        const int32_t* e = pv.mg + 4 * (g * pv.n_chunks + q);
        if (e[1] > e[0]) { lo_all = lo_all < e[0] ? lo_all : e[0]; hi_all = hi_all > e[1] ? hi_all : e[1]; }
      }
    const int w_union = (hi_all - 1 - lo_all) * pv.dil + TILE_N;
    const bool resident = ksplit == 1 && (size_t)pv.n_chunks * w_union * 64 + epi_bytes <= 76 * 1024;
    p.stage_vec = resident ? 1 : 0;
    p.mg_per_wg = resident ? pv.n_mgroups : 1;
    p.ldw = resident ? w_union : max_w_single;
    lds_bytes = (size_t)(resident ? pv.n_chunks : 1) * p.ldw * 64;
    p.epi_lds_off = (int)((lds_bytes / sizeof(float) + 3) / 4 * 4);
    lds_bytes = (size_t)p.epi_lds_off * sizeof(float) + epi_bytes;       // the epilogue's transpose tiles (vec or not)
    // 32-row M-groups on 256-sample tiles with resident windows: the waves split over the M-groups instead of over time
    // (conv_win_rows_kernel: a weight fragment is fetched by one wave, not by all four)
    static const bool rows_off = getenv("FST_WIN_ROWS") && atoi(getenv("FST_WIN_ROWS")) == 0;     // diagnostics
    if (resident && pv.MB == 1 && nb_cfg == 2 && pv.n_mgroups >= 4 && res == nullptr &&
        !(flags & (FST_EPI_ATOMIC | FST_EPI_ACC1 | FST_EPI_ACC2)) && !rows_off)
      fn = conv_win_rows_kernel;
  } else {
    p.epi_lds_off = (int)((lds_bytes / sizeof(float) + 3) / 4 * 4);   // after the staged window (reused across M-groups)
    if (p.epi_vec) lds_bytes = (size_t)p.epi_lds_off * sizeof(float) + epi_bytes;
  }
  FST_REQUIRE(lds_bytes <= 160 * 1024, "fst_conv_gemm: LDS window %zu B exceeds 160 KiB (chunk_cap=%d ldw=%d)",
              lds_bytes, pv.chunk_cap, p.ldw);
  if (lds_bytes > 48 * 1024)
    if (int rc = fst_allow_full_lds((const void*)fn, "fst_conv_gemm")) return rc;
  dim3 grid((unsigned)(B * p.tiles_per_seq), (unsigned)((pv.n_mgroups + p.mg_per_wg - 1) / p.mg_per_wg), (unsigned)ksplit);
  hipLaunchKernelGGL(fn, grid, dim3(256), lds_bytes, (hipStream_t)stream, p, plan_dev);
  FST_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------------
struct WgradParams {
  const float* x[2];
  long long x_bs[2];
  const float* dy;
  long long dy_bs;
  const float* dy2;
  long long dy2_bs;
  int msplit;
  float* da;
  const int32_t* plan;
  int B, L, M, ksplit, tiles_per_seq, ldw, region_floats, n_regions;
  long long x0_mul_off;   // != 0: the x operand is x0[i]·x0[i + x0_mul_off] (acts = t·s read from the saved gate halves)
  long long slab_floats;  // != 0: K slice s stores its partial sums plainly into da + s·slab_floats (summed by the unpack)
};

#define WG_ITEMS 4   // row-blocks (32 packed K-rows each) per workgroup
// Diagnostic builds only (tools/build_wg_exp.sh): WG_EXP removes one cost at a time from the split-bf16 weight-gradient
// kernel (wrong results, timing only): 1 no atomics, 2 no MFMAs, 4 no dy fetch, 8 no x fetch, 16 no LDS commit.
#ifndef WG_EXP
#define WG_EXP 0
#endif

// CB: output-channel blocks per wave (M tile = 4*CB*32).  WIDE=false: every chunk is a single tap of ≤ 32
// channels (window = TW columns, up to WG_ITEMS windows per workgroup); WIDE=true: one windowed chunk of
// ≤ 64 channels and ≤ 128 columns (omni-scale layers, all items of a workgroup share it).
// The next (b,t) tile is fetched global→registers while the current one is multiplied.
//
// BF3: the products run on the bf16 matrix cores with split operands (see conv_gemm_bf3_kernel): time is cut into
// 16-deep k-steps; wave w owns item w (its 32 packed K-rows are the MFMA rows) against ALL 4·CB output-channel
// blocks, so the A fragment — 8 consecutive raw fp32 samples of the staged x window at any tap offset, eight
// ds_read_b32 — is split into hi / lo once, by the only wave that uses it; dy is split when it is staged, into two
// bf16 images [M rows][32 samples] (80-byte rows: a 16-lane group's ds_read_b128 covers all 64 banks) shared by the
// four waves.
template <int CB, int TW, bool WIDE, int VEC, bool BF3>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(WgradParams p, const int32_t* __restrict__ plan) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  static_assert(TW == 32, "staging maps one half-wave to one 32-sample row");
  constexpr int MBW = 4 * CB;
  constexpr int DYS = TW + 1;   // odd row stride: lanes ↔ rows never share a bank
  constexpr int DYB = 2 * TW + 16;                       // BF3: bytes per row of a dy image
  constexpr int DYV = MBW * 32 / 8;                      // dy floats per thread per tile
  constexpr int NREG = WIDE ? 1 : WG_ITEMS;              // staged windows per workgroup
  constexpr int XV = WIDE ? 32 : (VEC == 2 ? 5 : 4);     // x floats per thread per window (VEC 2: one float4 + the sub-shift tail)
  const PlanView pv = plan_view(plan);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int L = p.L, dil = pv.dil, ldw = p.ldw;

  float* xreg = lds;                                     // n_regions windows of region_floats
  float* zrow = lds + p.n_regions * p.region_floats;     // TW+2 zeros
  float* dyt = zrow + (TW + 2 + 3) / 4 * 4;              // [MBW*32][DYS]   (BF3: hi image, then lo image, [MBW*32][DYB bytes])
  char* const dyh = reinterpret_cast<char*>(dyt);
  char* const dyl = dyh + MBW * 32 * DYB;

  // decode this workgroup's items (all share one M-group)
  int it_q[WG_ITEMS], it_rb[WG_ITEMS], it_region[WG_ITEMS];
  int g = -1, nit = 0;
#pragma unroll
  for (int i = 0; i < WG_ITEMS; ++i) {
    const int32_t* it = pv.item + 4 * (blockIdx.y * WG_ITEMS + i);
    it_q[i] = it[1]; it_rb[i] = it[2];
    if (it[1] >= 0) { g = it[0]; nit = i + 1; }
    // consecutive items of one chunk share a staged window
    it_region[i] = i == 0 ? 0 : (it_q[i] == it_q[i - 1] ? it_region[i - 1] : it_region[i - 1] + 1);
  }
  if (g < 0) return;
  const int m0 = g * MBW * 32;

  // per staged window: source rows, channel count, first time sample relative to the tile
  const float* reg_src[NREG];
  int reg_cnt[NREG], reg_shift[NREG], reg_width[NREG], reg_sub[NREG];
  long long reg_bs[NREG];
#pragma unroll
  for (int r = 0; r < NREG; ++r) { reg_src[r] = nullptr; reg_cnt[r] = 0; reg_shift[r] = 0; reg_width[r] = 0; reg_bs[r] = 0; reg_sub[r] = 0; }
#pragma unroll
  for (int i = 0; i < WG_ITEMS; ++i) {
    if (i >= nit || (i > 0 && it_region[i] == it_region[i - 1])) continue;
    const int32_t* c = pv.chunk + 4 * it_q[i];
    const int32_t* e = pv.mg + 4 * (g * pv.n_chunks + it_q[i]);
#pragma unroll
    for (int r = 0; r < NREG; ++r) {
      if (r == it_region[i]) {
        reg_src[r] = p.x[c[0]] + (long long)c[1] * L;
        reg_bs[r] = p.x_bs[c[0]];
        reg_cnt[r] = c[2];
        reg_shift[r] = e[0] * dil - pv.pad_left;
        reg_sub[r] = VEC == 2 ? (reg_shift[r] & 3) : 0;  // samples by which the window starts past a 16-byte boundary
        reg_width[r] = (e[1] - 1 - e[0]) * dil + TW;
      }
    }
  }

  // per-lane LDS row offset of each item's A-operand row (a packed K-row = (tap, channel))
  int rowoff[WG_ITEMS];
#pragma unroll
  for (int i = 0; i < WG_ITEMS; ++i) {
    rowoff[i] = (int)(zrow - lds);
    if (i < nit) {
      const int q = it_q[i];
      const int32_t* e = pv.mg + 4 * (g * pv.n_chunks + q);
      const int c_pad = (pv.chunk[4 * q + 2] + 1) & ~1;
      const int nrows = (e[1] - e[0]) * c_pad;
      const int r = it_rb[i] * 32 + l31;
      if (r < nrows) {
        const int tapi = r / c_pad, c = r - tapi * c_pad;
        rowoff[i] = it_region[i] * p.region_floats + c * ldw + tapi * dil;
      }
    }
  }
  for (int j = tid; j < TW + 2; j += 256) zrow[j] = 0.f;

  f32x16 acc[WG_ITEMS][CB];
#pragma unroll
  for (int i = 0; i < WG_ITEMS; ++i)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][cb][r] = 0.f;

  const int n_tiles = p.B * p.tiles_per_seq;
  const int tile_begin = (int)(((long long)blockIdx.x * n_tiles) / p.ksplit);
  const int tile_end = (int)(((long long)(blockIdx.x + 1) * n_tiles) / p.ksplit);

  float dy_st[DYV];
  float x_st[NREG][XV];
  // Addressing: the row part of every address is wave-uniform (SGPR arithmetic); one per-lane offset
  // (half·L + l31, or lane) is the only address VGPR, so nothing large is hoisted out of the tile loop.
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int vlane = half * L + l31;
  // VEC (narrow windows, every address 16-B aligned — checked on the host): 16-byte loads, 8 lanes per 32-sample
  // row.  dy: row = (tid>>3) + 32*i; windows: row = tid>>3.  12 load instructions per thread per tile instead of 48.
  const int vrow = tid >> 3, vcol = (tid & 7) * 4;
  auto fetch = [&](int tile) {
    const int b = tile / p.tiles_per_seq;
    const int tt0 = (tile - b * p.tiles_per_seq) * TW;
    if (VEC) {
      const bool tv_ok = tt0 + vcol < L;
#pragma unroll
      for (int i = 0; i < DYV / 4; ++i) {
        const int m = m0 + vrow + 32 * i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tv_ok && m < p.M) {
          const float* rp = (m < p.msplit) ? p.dy + ((long long)b * p.dy_bs + (long long)m * L)
                                           : p.dy2 + ((long long)b * p.dy2_bs + (long long)(m - p.msplit) * L);
          v = *reinterpret_cast<const float4*>(rp + tt0 + vcol);
        }
        dy_st[4 * i] = v.x; dy_st[4 * i + 1] = v.y; dy_st[4 * i + 2] = v.z; dy_st[4 * i + 3] = v.w;
      }
#pragma unroll
      for (int r = 0; r < NREG; ++r) {
        if (reg_src[r] == nullptr) continue;
        if (WIDE) {
          // window [<=64 rows][<=128 columns] = 32 float4 per row: row = (tid>>5) + 8*i, column 4*(tid&31)
          const int wc = (tid & 31) * 4;
          const int t = tt0 + reg_shift[r] + wc;
          const bool ok = wc < reg_width[r] && t >= 0 && t < L;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int cc = (tid >> 5) + 8 * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok && cc < reg_cnt[r])
              v = *reinterpret_cast<const float4*>(reg_src[r] + ((long long)b * reg_bs[r] + (long long)cc * L + t));
            x_st[r][4 * i] = v.x; x_st[r][4 * i + 1] = v.y; x_st[r][4 * i + 2] = v.z; x_st[r][4 * i + 3] = v.w;
          }
        } else {
          // a tap shift that is not a multiple of 4 samples: the row's eight float4 start `sub` samples early (window
          // columns vcol-sub .. vcol-sub+3) and lanes 0..sub-1 of the row fetch the tail columns 32-sub .. 31 one by one
          const int sub = reg_sub[r];
          const int t = tt0 + reg_shift[r] - sub + vcol;
          const float* rowp = reg_src[r] + ((long long)b * reg_bs[r] + (long long)vrow * L);
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (t >= 0 && t < L && vrow < reg_cnt[r]) {
            v = *reinterpret_cast<const float4*>(rowp + t);
            if (p.x0_mul_off) {
              const float4 w = *reinterpret_cast<const float4*>(rowp + t + p.x0_mul_off);
              v.x *= w.x; v.y *= w.y; v.z *= w.z; v.w *= w.w;
            }
          }
          x_st[r][0] = v.x; x_st[r][1] = v.y; x_st[r][2] = v.z; x_st[r][3] = v.w;
          if constexpr (VEC == 2) {
            float xe = 0.f;
            const int te = tt0 + reg_shift[r] + TW - sub + (tid & 7);
            if ((tid & 7) < sub && te >= 0 && te < L && vrow < reg_cnt[r]) {
              xe = rowp[te];
              if (p.x0_mul_off) xe *= rowp[te + p.x0_mul_off];
            }
            x_st[r][4] = xe;
          }
        }
      }
      return;
    }
    const bool t_ok = tt0 + l31 < L;
    // dy tile: one half-wave per 32-sample row; rows m_even + half (msplit is even: a wave's two rows share a side)
#pragma unroll
    for (int i = 0; i < DYV; ++i) {
      const int m_even = m0 + wave_s * 2 + 8 * i;
      const float* rp = (m_even < p.msplit)
                            ? p.dy + ((long long)b * p.dy_bs + (long long)m_even * L + tt0)
                            : p.dy2 + ((long long)b * p.dy2_bs + (long long)(m_even - p.msplit) * L + tt0);
      dy_st[i] = (t_ok && m_even + half < p.M) ? rp[vlane] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < NREG; ++r) {
      if (reg_src[r] == nullptr) continue;
      const int tbase = tt0 + reg_shift[r];
      if (WIDE) {
        // rows cc = wave + 4*i (i < 16), columns lane and lane+64
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int cc = wave_s + 4 * i;
          const float* rp = reg_src[r] + ((long long)b * reg_bs[r] + (long long)cc * L + tbase);
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int j = lane + 64 * k, t = tbase + j;
            x_st[r][2 * i + k] = (cc < reg_cnt[r] && j < reg_width[r] && t >= 0 && t < L) ? rp[j] : 0.f;
          }
        }
      } else {
        // rows cc = wave*2 + half + 8*i (i < 4), column l31
        const int t = tbase + l31;
        const bool ok = t >= 0 && t < L;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int cc_even = wave_s * 2 + 8 * i;
          const float* rp = reg_src[r] + ((long long)b * reg_bs[r] + (long long)cc_even * L + tbase);
          float xv = 0.f;
          if (ok && cc_even + half < reg_cnt[r]) {
            xv = rp[vlane];
            if (p.x0_mul_off) xv *= rp[vlane + p.x0_mul_off];
          }
          x_st[r][i] = xv;
        }
      }
    }
  };
  auto commit = [&]() {
    if (VEC) {
      if (BF3) {
#pragma unroll
        for (int i = 0; i < DYV / 4; ++i) {
          unsigned h0, h1, l0, l1;
          split_bf16_pair(dy_st[4 * i], dy_st[4 * i + 1], h0, l0);
          split_bf16_pair(dy_st[4 * i + 2], dy_st[4 * i + 3], h1, l1);
          const int off = (vrow + 32 * i) * DYB + vcol * 2;
          *reinterpret_cast<uint2*>(dyh + off) = make_uint2(h0, h1);
          *reinterpret_cast<uint2*>(dyl + off) = make_uint2(l0, l1);
        }
      } else {
#pragma unroll
      for (int i = 0; i < DYV / 4; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) dyt[(vrow + 32 * i) * DYS + vcol + k] = dy_st[4 * i + k];
      }
#pragma unroll
      for (int r = 0; r < NREG; ++r) {
        if (reg_src[r] == nullptr) continue;
        float* reg = xreg + r * p.region_floats;
        if (WIDE) {
          const int wc = (tid & 31) * 4;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int cc = (tid >> 5) + 8 * i;
            if (cc < pv.chunk_cap && wc < reg_width[r]) {
#pragma unroll
              for (int k = 0; k < 4; ++k) reg[cc * ldw + wc + k] = x_st[r][4 * i + k];
            }
          }
        } else if (vrow < pv.chunk_cap) {
          const int sub = reg_sub[r], c0 = vcol - sub;
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (c0 + k >= 0) reg[vrow * ldw + c0 + k] = x_st[r][k];
          if constexpr (VEC == 2)
            if ((tid & 7) < sub) reg[vrow * ldw + TW - sub + (tid & 7)] = x_st[r][4];
        }
      }
      return;
    }
    if (BF3) {
#pragma unroll
      for (int i = 0; i < DYV; ++i) {
        unsigned hh, ll;
        split_bf16_pair(dy_st[i], 0.f, hh, ll);
        const int off = (wave * 2 + half + 8 * i) * DYB + l31 * 2;
        *reinterpret_cast<unsigned short*>(dyh + off) = (unsigned short)hh;
        *reinterpret_cast<unsigned short*>(dyl + off) = (unsigned short)ll;
      }
    } else {
#pragma unroll
    for (int i = 0; i < DYV; ++i) dyt[(wave * 2 + half + 8 * i) * DYS + l31] = dy_st[i];
    }
#pragma unroll
    for (int r = 0; r < NREG; ++r) {
      if (reg_src[r] == nullptr) continue;
      float* reg = xreg + r * p.region_floats;
      if (WIDE) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int cc = wave + 4 * i, j = lane + 64 * k;
            if (cc < pv.chunk_cap && j < reg_width[r]) reg[cc * ldw + j] = x_st[r][2 * i + k];
          }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int cc = wave * 2 + half + 8 * i;
          if (cc < pv.chunk_cap) reg[cc * ldw + l31] = x_st[r][i];
        }
      }
    }
  };

  if (tile_begin < tile_end && !(WG_EXP & 12)) fetch(tile_begin);
  for (int tile = tile_begin; tile < tile_end; ++tile) {
    __syncthreads();   // previous tile's readers are done
    if (!(WG_EXP & 16)) commit();
    __syncthreads();
    if (tile + 1 < tile_end && !(WG_EXP & 12)) fetch(tile + 1);

    if (BF3) {
      // wave w multiplies item w: rows = its 32 packed K-rows (lane's row: rowoff_w), columns = every LIVE output block
      const int nblk_live = min(MBW, (p.M - m0 + 31) / 32);
      int roff = rowoff[0];
#pragma unroll
      for (int i = 1; i < WG_ITEMS; ++i) roff = wave == i ? rowoff[i] : roff;
#pragma unroll
      for (int ks = 0; ks < TW / 16; ++ks) {
        const float* ap = lds + roff + ks * 16 + 8 * half;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ap[j];
        u32x4 h, l;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          unsigned hh, ll;
          split_bf16_pair(v[2 * j], v[2 * j + 1], hh, ll);
          h[j] = hh; l[j] = ll;
        }
        const bf16x8 ah = __builtin_bit_cast(bf16x8, h), al = __builtin_bit_cast(bf16x8, l);
        const int boff = l31 * DYB + (ks * 16 + 8 * half) * 2;
#pragma unroll
        for (int i = 0; i < WG_ITEMS; ++i)
#pragma unroll
          for (int cb = 0; cb < CB; ++cb) {
            const int blk = i * CB + cb;
            if constexpr (CB == 1)                        // (the 8-block form measured 10-50 % slower with this test in its loop)
              if (blk >= nblk_live) continue;            // output-channel blocks beyond M (M = 25: three of four) hold zeros
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(dyh + blk * 32 * DYB + boff);
            const bf16x8 bl = *reinterpret_cast<const bf16x8*>(dyl + blk * 32 * DYB + boff);
            if (WG_EXP & 2) { asm volatile("" ::"v"(al), "v"(ah), "v"(bh), "v"(bl)); continue; }
            acc[i][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[i][cb], 0, 0, 0);
            acc[i][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i][cb], 0, 0, 0);
            acc[i][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i][cb], 0, 0, 0);
          }
      }
      continue;
    }
    const float* bbase = dyt + (wave * CB * 32 + l31) * DYS + half;
    // straight-line k-steps: padding items multiply the zero row instead of branching around their MFMAs (all
    // workgroups of a launch are co-resident, so the launch lasts as long as a full workgroup either way)
#pragma unroll 2
    for (int tau = 0; tau < TW; tau += 2) {
      float av[WG_ITEMS], bv[CB];
#pragma unroll
      for (int i = 0; i < WG_ITEMS; ++i) av[i] = lds[rowoff[i] + tau + half];
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) bv[cb] = bbase[cb * 32 * DYS + tau];
#pragma unroll
      for (int i = 0; i < WG_ITEMS; ++i)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
          acc[i][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[cb], acc[i][cb], 0, 0, 0);
    }
  }

  if (BF3) {
    // this wave's item: acc[i][cb] is output-channel block i*CB+cb of item `wave`
    int my_q = it_q[0], my_rb = it_rb[0];
#pragma unroll
    for (int i = 1; i < WG_ITEMS; ++i) { my_q = wave == i ? it_q[i] : my_q; my_rb = wave == i ? it_rb[i] : my_rb; }
    if (wave >= nit || my_q < 0) return;
    const int32_t* e = pv.mg + 4 * (g * pv.n_chunks + my_q);
    const int c_pad = (pv.chunk[4 * my_q + 2] + 1) & ~1;
    const int nrows = (e[1] - e[0]) * c_pad;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = my_rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (row >= nrows) continue;
      const int tapi = row / c_pad, c = row - tapi * c_pad;
      const long long rec = (long long)e[2] + tapi * (c_pad / 2) + (c >> 1);
#pragma unroll
      for (int i = 0; i < WG_ITEMS; ++i)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
          if (WG_EXP & 1) { if (acc[i][cb][r] == 12345.678f) p.da[0] = 1.f; continue; }
          float* dst = p.da + (long long)blockIdx.x * p.slab_floats + (rec * MBW + i * CB + cb) * 64 + (c & 1) * 32 + l31;
          if (p.slab_floats) *dst = acc[i][cb][r]; else atomicAdd(dst, acc[i][cb][r]);
        }
    }
    return;
  }
  // epilogue: acc rows = packed K-rows (tap, channel), cols = output channel m (lane) → record layout
#pragma unroll
  for (int i = 0; i < WG_ITEMS; ++i) {
    if (i >= nit) continue;
    const int q = it_q[i];
    const int32_t* e = pv.mg + 4 * (g * pv.n_chunks + q);
    const int c_pad = (pv.chunk[4 * q + 2] + 1) & ~1;
    const int nrows = (e[1] - e[0]) * c_pad;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = it_rb[i] * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (row >= nrows) continue;
      const int tapi = row / c_pad, c = row - tapi * c_pad;
      const long long rec = (long long)e[2] + tapi * (c_pad / 2) + (c >> 1);
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        const int mbw = wave * CB + cb;
        float* dst = p.da + (long long)blockIdx.x * p.slab_floats + (rec * MBW + mbw) * 64 + (c & 1) * 32 + l31;
        if (p.slab_floats) *dst = acc[i][cb][r]; else atomicAdd(dst, acc[i][cb][r]);
      }
    }
  }
}

extern "C" int fst_conv_wgrad(const float* x0, int64_t x0_bs, const float* x1, int64_t x1_bs, const float* dy,
                              int64_t dy_bs, const float* dy2, int64_t dy2_bs, int msplit, float* da_packed,
                              const int32_t* plan_dev, const int32_t* plan_host, int plan_len, int B, int L, int M,
                              int ksplit, int flags, int64_t x0_mul_off, void* stream) {
  if (int rc = fst_check_plan(plan_host, plan_len, M, "fst_conv_wgrad")) return rc;
  const PlanView pv = plan_view(plan_host);
  FST_REQUIRE(x0 && dy && da_packed && plan_dev, "fst_conv_wgrad: null operand");
  FST_REQUIRE(pv.MB == 4 || pv.MB == 8, "fst_conv_wgrad: plan MB must be 4 or 8 (got %d)", pv.MB);
  FST_REQUIRE(pv.n_items > 0 && pv.items_per_wg == WG_ITEMS && pv.n_items % WG_ITEMS == 0,
              "fst_conv_wgrad: plan has no item table for %d items/workgroup", WG_ITEMS);
  FST_REQUIRE(msplit >= 0 && msplit <= M && (msplit == M || dy2 != nullptr), "fst_conv_wgrad: bad msplit=%d", msplit);
  FST_REQUIRE(msplit == M || msplit % 2 == 0, "fst_conv_wgrad: a split dy needs an even msplit (got %d)", msplit);
  FST_REQUIRE(B > 0 && L > 0 && ksplit >= 1, "fst_conv_wgrad: bad sizes");
  bool needs_x1 = false;
  for (int q = 0; q < pv.n_chunks; ++q) needs_x1 |= pv.chunk[4 * q] == 1;
  FST_REQUIRE(!needs_x1 || x1 != nullptr, "fst_conv_wgrad: plan reads input 1 but x1 is null");
  {
    int rows_in[2] = {0, 0};
    for (int q = 0; q < pv.n_chunks; ++q) {
      const int32_t* c = pv.chunk + 4 * q;
      rows_in[c[0]] = rows_in[c[0]] > c[1] + c[2] ? rows_in[c[0]] : c[1] + c[2];
    }
    FST_REQUIRE(B == 1 || (x0_bs >= (int64_t)rows_in[0] * L && (!needs_x1 || x1_bs >= (int64_t)rows_in[1] * L) &&
                           dy_bs >= (int64_t)msplit * L && (msplit == M || dy2_bs >= (int64_t)(M - msplit) * L)),
                "fst_conv_wgrad: a batch stride (x0 %lld, x1 %lld, dy %lld, dy2 %lld) is smaller than the rows it strides over",
                (long long)x0_bs, (long long)x1_bs, (long long)dy_bs, (long long)dy2_bs);
  }
  for (int i = 0; i < pv.n_items; ++i) {
    const int32_t* it = pv.item + 4 * i;
    FST_REQUIRE(it[1] < pv.n_chunks && (it[1] < 0 || (it[0] >= 0 && it[0] < pv.n_mgroups && it[2] >= 0)),
                "fst_conv_wgrad: bad item %d", i);
  }
  constexpr int TW = 32;
  WgradParams p;
  p.x[0] = x0; p.x[1] = x1; p.x_bs[0] = x0_bs; p.x_bs[1] = x1_bs;
  p.dy = dy; p.dy_bs = dy_bs; p.dy2 = dy2; p.dy2_bs = dy2_bs; p.msplit = msplit;
  p.da = da_packed; p.plan = plan_dev; p.B = B; p.L = L; p.M = M;
  p.x0_mul_off = x0_mul_off;
  p.slab_floats = (flags & FST_WGRAD_SLABS) ? (long long)pv.total_records * pv.MB * 64 : 0;
  p.tiles_per_seq = (L + TW - 1) / TW;
  const int n_tiles = B * p.tiles_per_seq;
  p.ksplit = ksplit < n_tiles ? ksplit : n_tiles;
  int max_w = 0;
  bool wide = false;
  for (int q = 0; q < pv.n_chunks; ++q)
    for (int g = 0; g < pv.n_mgroups; ++g) {
      const int32_t* e = pv.mg + 4 * (g * pv.n_chunks + q);
      if (e[1] > e[0]) {
        int w = (e[1] - 1 - e[0]) * pv.dil + TW;
        max_w = max_w > w ? max_w : w;
        if (e[1] > e[0] + 1) wide = true;
      }
    }
  FST_REQUIRE(max_w > 0, "fst_conv_wgrad: plan has no live taps");
  p.ldw = max_w | 1;                                     // odd stride: A-operand lanes walk channels
  p.region_floats = (pv.chunk_cap * p.ldw + 3) / 4 * 4;
  p.n_regions = 1;
  for (int w = 0; w < pv.n_items / WG_ITEMS; ++w) {
    int distinct = 0, prev = -2;
    for (int i = 0; i < WG_ITEMS; ++i) {
      const int q = pv.item[4 * (w * WG_ITEMS + i) + 1];
      if (q >= 0 && q != prev) ++distinct;
      prev = q;
    }
    p.n_regions = p.n_regions > distinct ? p.n_regions : distinct;
  }
  FST_REQUIRE(x0_mul_off == 0 || (!wide && !needs_x1 && x0_mul_off % 4 == 0),
              "fst_conv_wgrad: a product operand (x0_mul_off=%lld) needs a single-input, single-tap plan and a multiple-of-4 offset",
              (long long)x0_mul_off);
  if (wide) {
    FST_REQUIRE(p.n_regions == 1 && pv.chunk_cap <= 64 && max_w <= 128,
                "fst_conv_wgrad: windowed plan needs one chunk per workgroup, <= 64 channels, <= 128 columns "
                "(regions=%d chunk_cap=%d width=%d)", p.n_regions, pv.chunk_cap, max_w);
  } else {
    FST_REQUIRE(pv.chunk_cap <= 32, "fst_conv_wgrad: single-tap plan needs chunks of <= 32 channels (got %d)", pv.chunk_cap);
  }
  const bool bf3 = (flags & FST_GEMM_BF16X3) != 0;
  const size_t dy_bytes = bf3 ? 2 * (size_t)pv.MB * 32 * (2 * TW + 16) : (size_t)pv.MB * 32 * (TW + 1) * sizeof(float);
  const size_t lds_bytes = ((size_t)p.n_regions * p.region_floats + (TW + 2 + 3) / 4 * 4) * sizeof(float) + dy_bytes;
  FST_REQUIRE(lds_bytes <= 160 * 1024, "fst_conv_wgrad: LDS %zu B exceeds 160 KiB", lds_bytes);
  // 16-byte loads for narrow windows when every tile row address is 16-B aligned
  auto al16 = [](const void* q) { return q == nullptr || (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const bool shifts4 = pv.pad_left % 4 == 0 && (pv.ntaps == 1 || pv.dil % 4 == 0);
  // windowed plans: the window of every (M-group, chunk) must also START on a multiple of 4 samples and be a
  // multiple of 4 wide
  bool starts4 = true;
  for (int q = 0; q < pv.n_chunks; ++q)
    for (int g = 0; g < pv.n_mgroups; ++g) {
      const int32_t* e = pv.mg + 4 * (g * pv.n_chunks + q);
      if (e[1] > e[0] && ((e[0] * pv.dil - pv.pad_left) % 4 != 0 || ((e[1] - 1 - e[0]) * pv.dil) % 4 != 0)) starts4 = false;
    }
  // (single-tap windows take any shift: the kernel starts the row's 16-byte loads `shift mod 4` samples early)
  const bool vec = L % 4 == 0 && (wide ? starts4 : true) && x0_bs % 4 == 0 && x1_bs % 4 == 0 && dy_bs % 4 == 0 &&
                   dy2_bs % 4 == 0 && al16(x0) && al16(x1) && al16(dy) && al16(dy2);
  void (*fn)(WgradParams, const int32_t*);
  // VEC: 0 dword staging, 1 16-byte staging, 2 16-byte staging of single-tap windows whose shift is not a multiple of 4
  const int vmode = !vec ? 0 : (wide || shifts4 ? 1 : 2);
#define FST_WGRAD_PICK(CBV, BF) \
  (wide ? (vmode ? conv_wgrad_kernel<CBV, TW, true, 1, BF> : conv_wgrad_kernel<CBV, TW, true, 0, BF>) \
        : (vmode == 2 ? conv_wgrad_kernel<CBV, TW, false, 2, BF> \
                      : (vmode ? conv_wgrad_kernel<CBV, TW, false, 1, BF> : conv_wgrad_kernel<CBV, TW, false, 0, BF>)))
  if (bf3) fn = pv.MB == 8 ? FST_WGRAD_PICK(2, true) : FST_WGRAD_PICK(1, true);
  else fn = pv.MB == 8 ? FST_WGRAD_PICK(2, false) : FST_WGRAD_PICK(1, false);
#undef FST_WGRAD_PICK
  if (lds_bytes > 48 * 1024)
    if (int rc = fst_allow_full_lds((const void*)fn, "fst_conv_wgrad")) return rc;
  dim3 grid((unsigned)p.ksplit, (unsigned)(pv.n_items / WG_ITEMS), 1);
  hipLaunchKernelGGL(fn, grid, dim3(256), lds_bytes, (hipStream_t)stream, p, plan_dev);
  FST_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// pack / unpack between PyTorch weight layout and the plan's record layout
// ------------------------------------------------------------------------------------------------
struct PackParams {
  const int32_t* plan;
  fst_wsrc src[2];
  float* dst[2];       // unpack targets
  float* a;            // packed buffer (written by pack, read by unpack)
  int M;               // rows [row_base, M) are valid; the weight view is indexed with m - row_base
  int unpack;
  int g_begin, g_end, row_base;
  int n_slabs;         // unpack: number of partial-sum slabs to add (1 = a plain packed gradient)
  long long slab_floats;
};

__global__ __launch_bounds__(256) void pack_kernel(PackParams p, const int32_t* __restrict__ plan) {
  const PlanView pv = plan_view(plan);
  const int gq = blockIdx.y + p.g_begin * pv.n_chunks;
  const int g = gq / pv.n_chunks, q = gq - g * pv.n_chunks;
  const int32_t* e = pv.mg + 4 * gq;
  const int32_t* c = pv.chunk + 4 * q;
  const int lo = e[0], hi = e[1];
  if (hi <= lo) return;
  const int MB = pv.MB, c_pad = (c[2] + 1) & ~1, half_c = c_pad / 2;
  const long long elems = (long long)(hi - lo) * half_c * MB * 64;
  const int s = c[0];
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < elems; idx += (long long)gridDim.x * 256) {
    const int lane = (int)(idx & 63);
    const long long rm = idx >> 6;
    const int mb = (int)(rm % MB);
    const int rec = (int)(rm / MB);
    const int tapi = rec / half_c, cp = rec - tapi * half_c;
    const int m = (g * MB + mb) * 32 + (lane & 31);
    const int cl = 2 * cp + (lane >> 5);
    const bool valid = m >= p.row_base && m < p.M && cl < c[2];
    const long long woff = p.src[s].off0 + (long long)(m - p.row_base) * p.src[s].sm + (long long)(c[1] + cl) * p.src[s].sc +
                           (long long)(lo + tapi) * p.src[s].st;
    float* ap = p.a + ((long long)e[2] + rec) * (MB * 64) + mb * 64 + lane;
    if (p.unpack) {
      if (valid) {
        // partial-sum slabs: eight independent loads in flight per thread (a serial chain of n_slabs loads is latency-bound)
        float v = 0.f;
        int sl = 0;
        for (; sl + 8 <= p.n_slabs; sl += 8) {
          float t[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) t[j] = ap[(long long)(sl + j) * p.slab_floats];
          v += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
        }
        for (; sl < p.n_slabs; ++sl) v += ap[(long long)sl * p.slab_floats];
        p.dst[s][woff] = v;
      }
    } else {
      *ap = valid ? p.src[s].w[woff] : 0.f;
    }
  }
}

// Split-bf16 image for conv_gemm_bf3_kernel: one thread per (tap of the entry, 32-row block, lane) writes the
// lane's 8 hi parts and 8 lo parts (2 x 16 bytes).
__global__ __launch_bounds__(256) void pack_bf3_kernel(PackParams p, const int32_t* __restrict__ plan) {
  const PlanView pv = plan_view(plan);
  const int gq = blockIdx.y + p.g_begin * pv.n_chunks;
  const int g = gq / pv.n_chunks, q = gq - g * pv.n_chunks;
  const int32_t* e = pv.mg + 4 * gq;
  const int32_t* c = pv.chunk + 4 * q;
  const int lo = e[0], hi = e[1];
  const int MB = pv.MB, s = c[0];
  if (blockIdx.y == 0 && blockIdx.x == 0 && threadIdx.x < 4)            // the 16-byte zero block behind the image (masked LDS-DMA pieces read it)
    p.a[(long long)pv.n_stages * MB * 512 + threadIdx.x] = 0.f;
  if (hi <= lo) return;
  const int total = (hi - lo) * MB * 64;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int lane = idx & 63;
    const int mb = (idx >> 6) % MB;
    const int tapi = (idx >> 6) / MB;
    const int m = (g * MB + mb) * 32 + (lane & 31);
    const bool row_ok = m >= p.row_base && m < p.M;
    unsigned hh[4], ll[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int cl = 8 * (lane >> 5) + 2 * j + k;
        const long long woff = p.src[s].off0 + (long long)(m - p.row_base) * p.src[s].sm +
                               (long long)(c[1] + cl) * p.src[s].sc + (long long)(lo + tapi) * p.src[s].st;
        v[k] = (row_ok && cl < c[2]) ? p.src[s].w[woff] : 0.f;
      }
      split_bf16_pair(v[0], v[1], hh[j], ll[j]);
    }
    uint4* dst = reinterpret_cast<uint4*>(p.a) + ((long long)(e[3] + tapi) * MB + mb) * 128 + lane;
    dst[0] = make_uint4(hh[0], hh[1], hh[2], hh[3]);
    dst[64] = make_uint4(ll[0], ll[1], ll[2], ll[3]);
  }
}

static int launch_pack(const int32_t* plan_dev, const int32_t* plan_host, int plan_len, const PackParams& p0,
                       const char* who, void* stream, bool bf3 = false) {
  if (int rc = fst_check_plan(plan_host, plan_len, p0.M, who)) return rc;
  const PlanView pv = plan_view(plan_host);
  FST_REQUIRE(p0.g_begin >= 0 && p0.g_begin < p0.g_end && p0.g_end <= pv.n_mgroups && p0.row_base >= 0,
              "%s: bad M-group range [%d,%d) of %d", who, p0.g_begin, p0.g_end, pv.n_mgroups);
  long long max_elems = 0;
  for (int gq = p0.g_begin * pv.n_chunks; gq < p0.g_end * pv.n_chunks; ++gq) {
    const int32_t* e = pv.mg + 4 * gq;
    const int32_t* c = pv.chunk + 4 * (gq % pv.n_chunks);
    if (e[1] > e[0]) {
      long long el = (long long)(e[1] - e[0]) * (((c[2] + 1) & ~1) / 2) * pv.MB * 64;
      max_elems = max_elems > el ? max_elems : el;
    }
  }
  FST_REQUIRE(max_elems > 0, "%s: plan has no live taps", who);
  long long bx = (max_elems + 255) / 256;
  if (bx > 1024) bx = 1024;
  PackParams p = p0;
  p.plan = plan_dev;
  if (bf3) {
    FST_REQUIRE(pv.chunk_cap <= PIPE_C, "%s: the split-bf16 image needs chunks of <= %d channels (got %d)", who, PIPE_C,
                pv.chunk_cap);
    // one thread per (tap, 32-row block, lane) of the entry with the most taps; entries with fewer leave some workgroups idle
    long long bxb = 0;
    for (int gq = p0.g_begin * pv.n_chunks; gq < p0.g_end * pv.n_chunks; ++gq) {
      const int32_t* e = pv.mg + 4 * gq;
      const long long t = (long long)(e[1] > e[0] ? e[1] - e[0] : 0) * pv.MB * 64;
      bxb = bxb > t ? bxb : t;
    }
    bxb = (bxb + 255) / 256;
    bxb = bxb < 1 ? 1 : (bxb > 64 ? 64 : bxb);
    hipLaunchKernelGGL(pack_bf3_kernel, dim3((unsigned)bxb, (unsigned)(pv.n_chunks * (p0.g_end - p0.g_begin))), dim3(256), 0,
                       (hipStream_t)stream, p, plan_dev);
  } else {
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)bx, (unsigned)(pv.n_chunks * (p0.g_end - p0.g_begin))), dim3(256), 0,
                       (hipStream_t)stream, p, plan_dev);
  }
  FST_LAUNCH_CHECK();
  return 0;
}

extern "C" int fst_pack_weights(const int32_t* plan_dev, const int32_t* plan_host, int plan_len,
                                const fst_wsrc* src0, const fst_wsrc* src1, int M, int g_begin, int g_end, int row_base,
                                float* a_packed, void* stream) {
  FST_REQUIRE(plan_dev && plan_host && src0 && src0->w && a_packed, "fst_pack_weights: null operand");
  PackParams p = {};
  p.src[0] = *src0;
  if (src1) p.src[1] = *src1;
  p.a = a_packed; p.M = M; p.unpack = 0;
  p.g_begin = g_begin; p.g_end = g_end < 0 ? (plan_len >= FST_PLAN_HDR ? plan_host[1] : 0) : g_end; p.row_base = row_base;
  const PlanView pv = plan_view(plan_host);
  if (plan_len >= FST_PLAN_HDR)
    for (int q = 0; q < pv.n_chunks; ++q)
      FST_REQUIRE(pv.chunk[4 * q] == 0 || (src1 && src1->w), "fst_pack_weights: plan reads input 1 but src1 is null");
  return launch_pack(plan_dev, plan_host, plan_len, p, "fst_pack_weights", stream);
}

extern "C" int fst_pack_weights_bf16x3(const int32_t* plan_dev, const int32_t* plan_host, int plan_len,
                                       const fst_wsrc* src0, const fst_wsrc* src1, int M, int g_begin, int g_end,
                                       int row_base, void* a_image, void* stream) {
  FST_REQUIRE(plan_dev && plan_host && src0 && src0->w && a_image, "fst_pack_weights_bf16x3: null operand");
  PackParams p = {};
  p.src[0] = *src0;
  if (src1) p.src[1] = *src1;
  p.a = static_cast<float*>(a_image); p.M = M; p.unpack = 0;
  p.g_begin = g_begin; p.g_end = g_end < 0 ? (plan_len >= FST_PLAN_HDR ? plan_host[1] : 0) : g_end; p.row_base = row_base;
  const PlanView pv = plan_view(plan_host);
  if (plan_len >= FST_PLAN_HDR)
    for (int q = 0; q < pv.n_chunks; ++q)
      FST_REQUIRE(pv.chunk[4 * q] == 0 || (src1 && src1->w), "fst_pack_weights_bf16x3: plan reads input 1 but src1 is null");
  return launch_pack(plan_dev, plan_host, plan_len, p, "fst_pack_weights_bf16x3", stream, true);
}

extern "C" int fst_unpack_weights(const int32_t* plan_dev, const int32_t* plan_host, int plan_len, const float* a_packed,
                                  int M, float* dw0, int64_t off0_0, int64_t sm0, int64_t sc0, int64_t st0, float* dw1,
                                  int64_t off0_1, int64_t sm1, int64_t sc1, int64_t st1, int n_slabs, void* stream) {
  FST_REQUIRE(plan_dev && dw0 && a_packed && n_slabs >= 1, "fst_unpack_weights: null operand / n_slabs=%d", n_slabs);
  PackParams p = {};
  p.n_slabs = n_slabs;
  p.slab_floats = (plan_host && plan_len >= FST_PLAN_HDR) ? (long long)plan_host[7] * plan_host[2] * 64 : 0;
  p.src[0] = {nullptr, off0_0, sm0, sc0, st0};
  p.src[1] = {nullptr, off0_1, sm1, sc1, st1};
  p.dst[0] = dw0; p.dst[1] = dw1;
  p.a = const_cast<float*>(a_packed); p.M = M; p.unpack = 1;
  p.g_begin = 0; p.g_end = (plan_host && plan_len >= FST_PLAN_HDR) ? plan_host[1] : 0; p.row_base = 0;
  const PlanView pv = plan_view(plan_host);
  if (plan_len >= FST_PLAN_HDR)
    for (int q = 0; q < pv.n_chunks; ++q)
      FST_REQUIRE(pv.chunk[4 * q] == 0 || dw1, "fst_unpack_weights: plan reads input 1 but dw1 is null");
  return launch_pack(plan_dev, plan_host, plan_len, p, "fst_unpack_weights", stream);
}

// W ← W ⊙ mask for an omni-scale layer: taps outside [lo[m], hi[m]) of output channel m are zeroed.
__global__ void mask_taps_kernel(float* w, const int32_t* lo, const int32_t* hi, int M, int C, int K) {
  const long long n = (long long)M * C * K;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % K);
    const int m = (int)(i / ((long long)C * K));
    if (k < lo[m] || k >= hi[m]) w[i] = 0.f;
  }
}

extern "C" int fst_mask_taps(float* w, const int32_t* live_lo, const int32_t* live_hi, int M, int C, int K, void* stream) {
  FST_REQUIRE(w && live_lo && live_hi && M > 0 && C > 0 && K > 0, "fst_mask_taps: bad arguments");
  const long long n = (long long)M * C * K;
  long long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(mask_taps_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, live_lo, live_hi, M, C, K);
  FST_LAUNCH_CHECK();
  return 0;
}
