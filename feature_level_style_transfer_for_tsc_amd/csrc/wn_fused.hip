// Fused WaveGlow WN layer for gfx950 (Simplified_NF_WaveGlow.py:101-123, gate :44-54).
//
// One launch per layer and direction instead of three:
//
//   forward   g = W_in (*) a  (3 dilated taps, n channels each) + W_cond · u0 + (b_in + b_cond)        GEMM 1
//             t = tanh(g[:n]),  s = sigmoid(g[n:]),  acts = t·s                       in registers
//             r = W_rs · acts + b_rs;   a_next = a + r[:n];   out (+)= r[n:]                          GEMM 2
//             (last layer: W_rs has n rows, all of them skip rows: out += r)
//   backward  dacts = W_rsᵀ · [d_a_next ; d_out]                                                     GEMM 3
//             dg[:n] = dacts·s·(1−t²),  dg[n:] = dacts·t·s·(1−s)                      in registers
//
// so the [B, 2n, L] pre-activation never reaches HBM and acts is (optionally) never written: per layer forward the
// traffic is  a + u0 + out (read)  and  a_next + out + t,s (written) — 391 MB at B=256, L=512, n=120 against 880 MB for the
// three-launch form.
//
// Structure (shared with conv_gemm_bf3_kernel): split-bf16 products — hi·hi + hi·lo + lo·hi on
// v_mfma_f32_32x32x16_bf16, fp32 accumulation — operands through a 3-slot LDS ring filled by LDS-DMA, one raw
// s_barrier per 16-deep stage behind a counted vmcnt.  A workgroup = 4 waves = one batch element × 128 time samples;
// every wave owns ALL 256 (padded) output rows of its 32 samples, so that
//   * row m (tanh half) and row m+128 (sigmoid half) of GEMM 1 sit in the same lane and register index (blocks b and b+4):
//     the gate is pure per-lane VALU work on the accumulators;
//   * the 32×32 accumulator tiles of acts are, register for register, the B operand of GEMM 2 (a following MFMA that
//     sums over the accumulator's ROW index takes it with no lane movement: cdna_hip_programming.md §3) — the weights
//     of GEMM 2 are packed in the k-order that layout implies (wn_pack_kernel).
// Biases ride in the GEMMs: GEMM 1 has a constant-one input row (the first padding channel of the conditioning
// chunk, fed from a 16-byte block of ones), GEMM 2 sees acts[n] = 1 — so n < 128 is required (one spare K row).
// Two workgroups share a CU (79 KB of LDS, ≤ 256 registers): one's t,s store burst is the other's MFMA time.
#include "fst_common.h"
#include <type_traits>

typedef __bf16 wn_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wn_bf16x2 __attribute__((ext_vector_type(2)));
typedef float wn_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned wn_u32x4 __attribute__((ext_vector_type(4)));
#define WN_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define WN_LDS_VOID(p) ((__attribute__((address_space(3))) void*)(p))

#define WS_MAXL 10                                  // layers of a WN stack the one-launch kernels take (tables in the kernel arguments)
#define WN_TN 128                                  // time samples per workgroup
// forward kernel, per NW waves: WN_NBLK = NW + 1 32-sample column blocks per 8-channel row group (+1: sub-shift spill), WN_GS =
// WN_NBLK·1024 + 128 bytes per row group (+128: the lane halves hit different banks), ring slot = WN_A_BYTES + 2·WN_GS
#define WN_A_BYTES (8 * 2048)                      // 8 row blocks × (1 KiB hi + 1 KiB lo fragments)
#define WN_TILE_BYTES (32 * 36 * 4)                // one wave's [32][36] fp32 transpose tile
#define WN_FWD_LDS(NW) (3 * (WN_A_BYTES + 2 * (((NW) + 1) * 1024 + 128)))

__device__ __forceinline__ void wn_split_pair(float a, float b, unsigned& hi, unsigned& lo) {
  const wn_f32x2 v = {a, b};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, wn_bf16x2));
  const wn_f32x2 r = {a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u)};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, wn_bf16x2));
}

template <int N>
__device__ __forceinline__ void wn_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// sigmoid and tanh from v_exp_f32 / v_rcp_f32 (≈1 ulp each): absolute error ≈ 2e-7, saturating correctly at ±inf
__device__ __forceinline__ float wn_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * x));
}
__device__ __forceinline__ float wn_tanh(float x) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008f * x));
}

// GEMM-1 stage k < 3·CH → (chunk, tap): chunks in groups of WN_GRP, tap-major inside a group — the three windows of a chunk
// (the same lines but for 2·dil samples) are fetched WN_GRP stages apart: late enough for the first to have landed (adjacent
// stages pulling the same lines while in flight measured 35 % slower), early enough to still be in L2.
#ifndef WN_GRP
#define WN_GRP 2
#endif
__host__ __device__ __forceinline__ void wn_stage_chunk_tap(int k, int CH, int& c, int& tap) {
  const int g = k / (3 * WN_GRP), rem = k - g * 3 * WN_GRP;
  const int gs = CH - g * WN_GRP < WN_GRP ? CH - g * WN_GRP : WN_GRP;
  tap = rem / gs;
  c = g * WN_GRP + rem - gs * tap;
}

// ------------------------------------------------------------------------------------------------
// weight image
//   [S1 stages of GEMM 1][8 k-steps of GEMM 2][16 B of zeros][16 B of ones]
// one stage / k-step = 8 row blocks × (64 lanes × 8 bf16 hi, 64 lanes × 8 bf16 lo) = 16 KiB, the A fragments of
// v_mfma_f32_32x32x16_bf16 (lane l: row l&31 of the block, k = 8·(l>>5) + j).
//   GEMM 1  stage k < 3·CH = (chunk c < CH = ⌈n/16⌉, tap) in wn_stage_chunk_tap order: channels 16c.. of `a` at tap `tap`;
//           then CH2 = ⌈(h+1)/16⌉ stages
//           of the conditioning input (zero shift) whose channel h is the constant-one row carrying b_in + b_cond.
//           rows: blocks 0-3 = tanh rows 0..n-1, blocks 4-7 = sigmoid rows n..2n-1.
//   GEMM 2  k-step ks = (blk, s): element j of lane half hh multiplies acts row blk·32 + 16s + 8(j>>2) + 4hh + (j&3)
//           (the order in which an accumulator tile delivers its rows as a B operand); k = n carries b_rs.
//           rows: blocks 0-3 = residual rows (→ a_next), blocks 4-7 = skip rows (→ out); last layer: only skip rows.
// ------------------------------------------------------------------------------------------------
struct WnPackParams {
  const float* in_w;    // [2n][n][3]
  const float* cond_w;  // [2n][h]
  const float* in_b;    // [2n]
  const float* cond_b;  // [2n]
  const float* rs_w;    // [2n][n]  (last: [n][n])
  const float* rs_b;    // [2n]     (last: [n])
  int n, h, last, CH, CH2;
  uint4* img;
};

__global__ __launch_bounds__(64) void wn_pack_kernel(WnPackParams p) {
  const int lane = threadIdx.x, blk = blockIdx.x & 7, st = blockIdx.x >> 3;
  const int S1 = 3 * p.CH + p.CH2, n = p.n, h = p.h;
  const int hh = lane >> 5, rowb = (blk & 3) * 32 + (lane & 31);
  const bool row_ok = rowb < n;
  float v[8];
  if (st < S1) {
    const int row = (blk < 4 ? 0 : n) + rowb;                       // tanh | sigmoid half of the in_layer / cond_layer
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float w = 0.f;
      if (row_ok) {
        if (st < 3 * p.CH) {
          int cc, tap;
          wn_stage_chunk_tap(st, p.CH, cc, tap);
          const int c = cc * 16 + 8 * hh + j;
          if (c < n) w = p.in_w[((long long)row * n + c) * 3 + tap];
        } else {
          const int c = (st - 3 * p.CH) * 16 + 8 * hh + j;
          if (c < h) w = p.cond_w[(long long)row * h + c];
          else if (c == h) w = p.in_b[row] + p.cond_b[row];
        }
      }
      v[j] = w;
    }
  } else {
    const int ks = st - S1, kb = ks >> 1, s = ks & 1;
    const bool skip = blk >= 4;
    const bool live = row_ok && (!p.last || skip);
    const int row = (skip && !p.last ? n : 0) + rowb;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = kb * 32 + 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3);
      float w = 0.f;
      if (live) {
        if (k < n) w = p.rs_w[(long long)row * n + k];
        else if (k == n) w = p.rs_b[row];
      }
      v[j] = w;
    }
  }
  unsigned hi[4], lo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) wn_split_pair(v[2 * j], v[2 * j + 1], hi[j], lo[j]);
  uint4* dst = p.img + ((long long)st * 8 + blk) * 128 + lane;
  dst[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
  dst[64] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
  if (blockIdx.x == 0 && lane < 2) {
    const unsigned one = lane ? 0x3f800000u : 0u;
    p.img[(long long)(S1 + 8) * 8 * 128 + lane] = make_uint4(one, one, one, one);
  }
}

static inline int wn_ch(int n) { return (n + 15) / 16; }
static inline int wn_ch2(int h) { return (h + 1 + 15) / 16; }

extern "C" int64_t fst_wn_image_bytes(int n, int h) {
  if (n <= 0 || h <= 0) return -1;
  return (int64_t)(3 * wn_ch(n) + wn_ch2(h) + 8) * WN_A_BYTES + 32;
}

extern "C" int fst_wn_pack(const float* in_w, const float* cond_w, const float* in_b, const float* cond_b,
                           const float* rs_w, const float* rs_b, int n, int h, int ntaps, int last, void* image,
                           int64_t image_bytes, void* stream) {
  FST_REQUIRE(in_w && cond_w && in_b && cond_b && rs_w && rs_b && image, "fst_wn_pack: null operand");
  FST_REQUIRE(ntaps == 3, "fst_wn_pack: the fused WN kernels are 3-tap (in_w [2n][n][3]); got %d taps", ntaps);
  FST_REQUIRE(n > 0 && n < 128 && h > 0, "fst_wn_pack: needs 0 < n < 128 (one spare K row carries the bias), h > 0; n=%d h=%d", n, h);
  FST_REQUIRE(image_bytes == fst_wn_image_bytes(n, h), "fst_wn_pack: image is %lld bytes, expected %lld",
              (long long)image_bytes, (long long)fst_wn_image_bytes(n, h));
  FST_REQUIRE((reinterpret_cast<uintptr_t>(image) & 15) == 0, "fst_wn_pack: image must be 16-byte aligned");
  WnPackParams p = {in_w, cond_w, in_b, cond_b, rs_w, rs_b, n, h, last ? 1 : 0, wn_ch(n), wn_ch2(h), static_cast<uint4*>(image)};
  const int stages = 3 * p.CH + p.CH2 + 8;
  hipLaunchKernelGGL(wn_pack_kernel, dim3((unsigned)(stages * 8)), dim3(64), 0, (hipStream_t)stream, p);
  FST_LAUNCH_CHECK();
  return 0;
}

// Diagnostic builds only (tools/build_wn_exp.sh): WN_EXP is a bit mask that removes one cost at a time from the fused
// forward kernel (wrong results, timing only): 1 no MFMAs, 2 B pieces from the zero block (no activation fetch), 4 A pieces
// from the zero block (no weight fetch), 8 no t,s / acts stores, 16 no final epilogue, 32 no B fragment reads / split, 64 no skip half
// (no load / store of the running skip sum, no skip-row MFMAs in GEMM 2: what a deferred skip GEMM would leave in this kernel).
#ifndef WN_EXP
#define WN_EXP 0
#endif
#ifndef WN_INTERLEAVE
#define WN_INTERLEAVE 1      // LDS-DMA pieces of stage k+2 issued between the MFMA triples of stage k (0: all up front)
#endif

#ifdef FST_STAMPS
// Diagnostic build only (tools/build_stamps.sh): per-phase s_memtime sums of the fused forward kernel, lane 0 of every wave.
__device__ unsigned long long wn_stamps[12];
__device__ __forceinline__ unsigned long long wn_now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
extern "C" int fst_debug_wn_stamps(unsigned long long* out_host, int reset) {
  if (out_host) hipMemcpyFromSymbol(out_host, HIP_SYMBOL(wn_stamps), sizeof(unsigned long long) * 12);
  if (reset) { unsigned long long z[12] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(wn_stamps), z, sizeof(z)); }
  return 0;
}
#define WN_T(var) const unsigned long long var = wn_now()
#define WN_ACC(slot, a, b) wn_sum[slot] += (b) - (a)
#define WN_SUMS unsigned long long wn_sum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define WN_FLUSH \
  if (lane == 0) for (int i_ = 0; i_ < 12; ++i_) atomicAdd(&wn_stamps[i_], wn_sum[i_])
#else
#define WN_T(var)
#define WN_ACC(slot, a, b)
#define WN_SUMS
#define WN_FLUSH
#endif

// ------------------------------------------------------------------------------------------------
// forward layer
// ------------------------------------------------------------------------------------------------
struct WnFwdParams {
  const float* a;
  long long a_bs;
  const float* u0;
  long long u0_bs;
  const char* img;
  float* ts;       // [B][2n][L]
  float* acts;     // [B][n][L] or null
  float* a_next;   // [B][n][L] (null on the last layer)
  float* out;      // [B][n][L]
  int B, L, n, h, dil, first, last;
  int CH, CH2, tiles_per_seq, n_wg;
};

// One accumulator tile (lane = time sample, registers = rows) → wave-private LDS tile [32][36] → each lane owns 4
// consecutive samples of rows rrow + 8j: four 16-byte global accesses per 32×32 tile, 8 rows × 128 contiguous bytes
// per wave-instruction.  MODE 0: store; 1: dst = v + res; 2: dst += v.
template <int MODE>
__device__ __forceinline__ void wn_store_tile(const float (&v)[16], float* tile, float* dst_rows, const float* res_rows,
                                              int rows_valid, int L, int t, int lane) {
  const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int r = 0; r < 16; ++r) tile[((r & 3) + 8 * (r >> 2) + 4 * half) * 36 + l31] = v[r];
  const int rrow = lane >> 3, c4 = (lane & 7) * 4;
  const bool t_ok = t + c4 < L;
  float4 o[4], e[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = rrow + 8 * j;
    o[j] = *reinterpret_cast<const float4*>(tile + row * 36 + c4);
    e[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE != 0 && t_ok && row < rows_valid) {
      const float* src = (MODE == 1 ? res_rows : dst_rows) + (long long)row * L + t + c4;
      e[j] = *reinterpret_cast<const float4*>(src);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = rrow + 8 * j;
    if (t_ok && row < rows_valid) {
      float4 w = o[j];
      w.x += e[j].x; w.y += e[j].y; w.z += e[j].z; w.w += e[j].w;
      *reinterpret_cast<float4*>(dst_rows + (long long)row * L + t + c4) = w;
    }
  }
}

// the four 16-byte pieces (rows rrow + 8j, samples c4..c4+3) a lane contributes to a 32×32 tile
__device__ __forceinline__ void wn_fetch_tile(float4 (&q)[4], const float* src_rows, int rows_valid, int L, int t, int lane) {
  const int rrow = lane >> 3, c4 = (lane & 7) * 4;
  const bool t_ok = t + c4 < L;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = rrow + 8 * j;
    q[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t_ok && row < rows_valid) q[j] = *reinterpret_cast<const float4*>(src_rows + (long long)row * L + t + c4);
  }
}
// Accumulator layout ↔ global memory directly (lane = time sample, register r = row (r&3) + 8(r>>2) + 4·half): one
// wave-instruction moves two rows × 32 consecutive samples (two full 128-byte segments).  Sixteen dword accesses per tile
// instead of four 16-byte ones, but no LDS transpose and — for the loads — no wait inside an epilogue: operand tiles are
// read INTO the accumulators before the GEMM that adds to them (out = operand + A·B), so their latency hides under the
// ring's first stages, and being older than every LDS-DMA piece they are covered by the counted waits as they stand.
__device__ __forceinline__ void wn_acc_load(f32x16& v, const float* src_rows, int rows_valid, int L, int t, int lane) {
  const int half = lane >> 5, l31 = lane & 31;
  const bool t_ok = t + l31 < L && src_rows != nullptr;
  // one per-lane address; the row term of every access is wave-uniform (scalar offset)
  const float* lp = src_rows + ((long long)(4 * half) * L + t + l31);
  const int rv = rows_valid - 4 * half;                  // rows this lane half may touch: (r&3) + 8(r>>2) < rv
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2);
    v[r] = (t_ok && row < rv) ? lp[(long long)row * L] : 0.f;
  }
}
__device__ __forceinline__ void wn_acc_store(const f32x16& v, float* dst_rows, int rows_valid, int L, int t, int lane) {
  const int half = lane >> 5, l31 = lane & 31;
  const bool t_ok = t + l31 < L;
  float* lp = dst_rows + ((long long)(4 * half) * L + t + l31);
  const int rv = rows_valid - 4 * half;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2);
    if (t_ok && row < rv) lp[(long long)row * L] = v[r];
  }
}
__device__ __forceinline__ void wn_acc_store(const float (&v)[16], float* dst_rows, int rows_valid, int L, int t, int lane) {
  const int half = lane >> 5, l31 = lane & 31;
  const bool t_ok = t + l31 < L;
  float* lp = dst_rows + ((long long)(4 * half) * L + t + l31);
  const int rv = rows_valid - 4 * half;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2);
    if (t_ok && row < rv) lp[(long long)row * L] = v[r];
  }
}
// ... through the wave-private tile into the accumulator layout (lane = sample, registers = rows)
__device__ __forceinline__ void wn_tile_to_acc(float (&v)[16], const float4 (&q)[4], float* tile, int lane) {
  const int rrow = lane >> 3, c4 = (lane & 7) * 4, half = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(tile + (rrow + 8 * j) * 36 + c4) = q[j];
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = tile[((r & 3) + 8 * (r >> 2) + 4 * half) * 36 + l31];
}

// Accumulator layout <-> global memory through raw buffer instructions: one descriptor per [rows][L] matrix of a batch element,
// ONE per-lane byte offset for the whole kernel (vlane = (4·half·L + l31)·4) and the row / column term of every access in the
// instruction's scalar offset — an access is one buffer_load/store_dword with no VALU work, no 64-bit address and no exec mask
// (wn_acc_load / _store form sixteen per-lane addresses per tile; the persistent kernel below has no registers for that).
// Columns beyond the sequence: the lane's offset is pushed out of the descriptor's range (loads return 0, stores are dropped by
// the hardware's range check).  Rows beyond the matrix: wave-uniform tests per register (the two lane halves of a register
// are 4 rows apart: a register whose upper half only is out of range goes through the half-masked offset).
#define WS_OOB 0x80000000u
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ws_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x40000000, 0x00020000);
}
struct WsLane { unsigned vo, vo_lo; };     // offsets of a lane whose column exists: both halves / lower half only
__device__ __forceinline__ WsLane ws_lane(unsigned vlane, int tcol, int L, int lane) {
  WsLane w;
  w.vo = tcol + (lane & 31) < L ? vlane : WS_OOB;
  w.vo_lo = lane < 32 ? w.vo : WS_OOB;
  return w;
}
// tile = rows row0 .. row0+31 (rv of them exist, rv may be <= 0 or >= 32), columns tcol .. tcol+31 of the matrix behind `rs`
// (The scalar offsets are formed from an opaque copy of L: loop-invariant otherwise, the compiler computes those of every tile of
// the kernel ahead of the layer loop and keeps — i.e. spills — hundreds of them.)
__device__ __forceinline__ int ws_opaque(int x) {
  asm volatile("" : "+s"(x));
  return x;
}
template <int AUX = 0>
__device__ __forceinline__ void ws_acc_load(f32x16& v, __amdgpu_buffer_rsrc_t rs, const WsLane w, int row0, int tcol, int rv, int L) {
  L = ws_opaque(L);
  rv = ws_opaque(rv);
  const int sbase = (row0 * L + tcol) * 4;
  if (rv >= 32) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      v[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, w.vo, sbase + ((r & 3) + 8 * (r >> 2)) * L * 4, AUX));
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2);
      float x = 0.f;
      if (row < rv) x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, row + 4 < rv ? w.vo : w.vo_lo, sbase + row * L * 4, AUX));
      v[r] = x;
    }
  }
}
template <int AUX = 0, class V>
__device__ __forceinline__ void ws_acc_store(const V& v, __amdgpu_buffer_rsrc_t rs, const WsLane w, int row0, int tcol, int rv, int L) {
  L = ws_opaque(L);
  rv = ws_opaque(rv);
  const int sbase = (row0 * L + tcol) * 4;
  if (rv >= 32) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)v[r]), rs, w.vo, sbase + ((r & 3) + 8 * (r >> 2)) * L * 4, AUX);
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2);
      if (row < rv)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)v[r]), rs, row + 4 < rv ? w.vo : w.vo_lo, sbase + row * L * 4, AUX);
    }
  }
}

// Row sums of an accumulator tile on its way out (bias gradients: Σ_{b,t} of every row): after the transpose each lane
// holds four consecutive samples of rows rrow + 8j, so a row's 32 samples are 8 lanes × float4 — three butterfly steps,
// then one LDS float add per row and wave.  `rows` = this WAVE's LDS array indexed by the tile's absolute row: one lane owns
// an entry, so the add is a plain read-modify-write in program order and the workgroup's sum over its waves (taken by the
// caller in wave order) is the same in every run — LDS float atomics from several waves were not.
__device__ __forceinline__ void wn_tile_row_sums(const float* tile, float* rows, int row0, int rows_valid, int L, int t, int lane) {
  const int rrow = lane >> 3, c4 = (lane & 7) * 4;
  const bool t_ok = t + c4 < L;                          // samples beyond the sequence are not part of the sum (L % 4 == 0)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 o = *reinterpret_cast<const float4*>(tile + (rrow + 8 * j) * 36 + c4);
    float s4 = t_ok ? (o.x + o.y) + (o.z + o.w) : 0.f;
    s4 += __shfl_xor(s4, 1, 64);
    s4 += __shfl_xor(s4, 2, 64);
    s4 += __shfl_xor(s4, 4, 64);
    if ((lane & 7) == 0 && rrow + 8 * j < rows_valid) rows[row0 + rrow + 8 * j] += s4;
  }
}

// NW waves per workgroup = 32·NW time samples: every wave streams the whole weight image through LDS whatever the tile width, so
// the L2→LDS fill per sample — the limiter of the 4-wave form (830 MB of pieces per launch) — halves at NW = 8
// (one workgroup of 8 waves per CU instead of two of 4: the same 2 waves per SIMD).
// one tile (32·NW samples from t0 of batch element b) of one layer: the whole body of the forward kernel
// (the arguments stay in the kernel-argument segment — constant address space: scalar loads at the point of use, rematerialised
// rather than kept in registers — for the per-layer kernel and for the persistent one alike)
typedef const __attribute__((address_space(4))) WnFwdParams WnFwdArgs;
template <int NW>
__device__ __forceinline__ void wn_fwd_tile(WnFwdArgs& p, const int b, const int t0, char* const ldsb) {
  constexpr int F_TN = 32 * NW, F_NBLK = F_TN / 32 + 1, F_GS = F_NBLK * 1024 + 128, F_SLOT = WN_A_BYTES + 2 * F_GS;
  constexpr int NA = WN_A_BYTES / 1024;              // 16 one-KiB pieces of A per stage
  constexpr int NI1 = NA + 2 * F_NBLK;              // 26 (NW = 4) / 34 (NW = 8) LDS-DMA wave-instructions per GEMM-1 stage
  int tid_o = threadIdx.x;
  asm volatile("" : "+v"(tid_o));                    // opaque per call: in the persistent kernel nothing derived from the lane id is
  const int tid = tid_o, lane = tid & 63;            //   hoisted out of the tile loops and kept alive across the whole body
  const int half = lane >> 5, l31 = lane & 31;
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_n0 = wave_s * 32;
  const int L = p.L, n = p.n, h = p.h, CH = p.CH;
  const int S1 = 3 * CH + p.CH2;
  const char* const zero16 = p.img + (long long)(S1 + 8) * WN_A_BYTES;
  const char* const ones16 = zero16 + 16;
  const float* const ab = p.a + (long long)b * p.a_bs;
  const float* const ub = p.u0 + (long long)b * p.u0_bs;

  // LDS-DMA of stage k into ring slot `slot`.  GEMM-1 stages: pieces [0, NA) are A, the rest the B sub-tiles
  // (8 channels × 32 samples of the 16-byte aligned, tap-shifted window); GEMM-2 k-steps: A only.
  // One stage's LDS-DMA, described by wave-uniform scalars (stage_src) and issued piece by piece (issue_piece: piece i of
  // this wave is ring piece wave + 4i).  GEMM-1 stages: pieces [0, NA) are A, the rest the B sub-tiles (8 channels × 32
  // samples of the 16-byte aligned, tap-shifted window); GEMM-2 k-steps: A only.
  struct StageSrc {
    char* sl;
    const char* asrc;
    const float* xb;
    int c_count, ones_row, t4, a0;
    bool spill, gemm2;
  };
  auto stage_src = [&](int k, int slot) {
    StageSrc ss;
    ss.sl = ldsb + slot * F_SLOT;
    ss.asrc = p.img + (long long)k * WN_A_BYTES;
    ss.gemm2 = k >= S1;
    ss.a0 = p.last ? 8 : 0;                            // last layer: only the skip-row blocks of GEMM 2 exist
    int shift = 0;
    if (k < 3 * CH) {
      int c, tap;
      wn_stage_chunk_tap(k, CH, c, tap);
      ss.xb = ab + (long long)(16 * c) * L;
      ss.c_count = min(16, n - 16 * c);
      ss.ones_row = -1;
      shift = (tap - 1) * p.dil;
    } else {
      const int c = k - 3 * CH;
      ss.xb = ub + (long long)(16 * c) * L;
      ss.c_count = min(16, h - 16 * c);
      ss.ones_row = h - 16 * c;                        // in [0, 16) on exactly one conditioning stage
    }
    const int tbase = t0 + shift;
    ss.t4 = tbase & ~3;
    ss.spill = (tbase & 3) != 0;
    return ss;
  };
  auto issue_piece = [&](const StageSrc& ss, int i) {
    if (ss.gemm2) {
      const int idx = ss.a0 + wave_s + NW * i;
      if (i < 16 / NW && idx < NA)
        __builtin_amdgcn_global_load_lds(WN_GLOBAL_PTR((WN_EXP & 4) ? zero16 : ss.asrc + idx * 1024 + lane * 16),
                                         WN_LDS_VOID(ss.sl + idx * 1024), 16, 0, 0);
      return;
    }
    const int idx = wave_s + NW * i;
    if (idx >= NI1) return;                            // wave-uniform
    if (idx < NA) {
      __builtin_amdgcn_global_load_lds(WN_GLOBAL_PTR((WN_EXP & 4) ? zero16 : ss.asrc + idx * 1024 + lane * 16),
                                       WN_LDS_VOID(ss.sl + idx * 1024), 16, 0, 0);
    } else {
      const int bi = idx - NA;
      const int gq = bi >= F_NBLK ? 1 : 0, m = bi - gq * F_NBLK;
      const int row = 8 * gq + (lane >> 3);
      const int t = ss.t4 + 32 * m + 4 * (lane & 7);
      bool ok = row < ss.c_count && t >= 0 && t < L;
      if (m == F_NBLK - 1) ok = ok && ss.spill && (lane & 7) == 0;
      if (WN_EXP & 2) ok = false;
      const char* src = ok ? reinterpret_cast<const char*>(ss.xb + ((long long)row * L + t)) : (row == ss.ones_row ? ones16 : zero16);
      __builtin_amdgcn_global_load_lds(WN_GLOBAL_PTR(src), WN_LDS_VOID(ss.sl + WN_A_BYTES + gq * F_GS + m * 1024), 16, 0, 0);
    }
  };
  constexpr int NPW1 = (NI1 + NW - 1) / NW;            // pieces per wave per GEMM-1 stage (waves >= NI1 % NW issue one fewer)
  static_assert(NPW1 <= 8, "the pieces of a stage are issued between the eight MFMA triples");
  auto issue = [&](int k, int slot) {
    const StageSrc ss = stage_src(k, slot);
#pragma unroll
    for (int i = 0; i < NPW1; ++i) issue_piece(ss, i);
  };
  // wait until this wave's pieces of the stage about to be read have landed; `next` = the one stage issued after it
  auto wait_for = [&](int next, int S) {
    if (next >= S) wn_wait_vmcnt<0>();
    else if (next < S1) { if (wave_s < (NI1 % NW)) wn_wait_vmcnt<(NI1 + NW - 1) / NW>(); else wn_wait_vmcnt<NI1 / NW>(); }
    else if (p.last) wn_wait_vmcnt<8 / NW>();
    else wn_wait_vmcnt<16 / NW>();
  };

  f32x16 acc[8];
#pragma unroll
  for (int mb = 0; mb < 8; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;

  const int S = S1 + 8;
  WN_SUMS;
  WN_T(ts0);
  issue(0, 0);
  issue(1, 1);
  int slot = 0;
  WN_T(ts1);
  WN_ACC(0, ts0, ts1);                                 // prologue
  // ---------------------------------------------------------------- GEMM 1: g = [W_in | W_cond | b] · [a taps ; u0 ; 1]
  for (int k = 0; k < S1; ++k) {
    WN_T(ta);
    wait_for(k + 1, S);
    WN_T(tb);
    WN_ACC(1, ta, tb);                                 // vmcnt wait
    __builtin_amdgcn_s_barrier();                      // stage k is in LDS for everyone; the slot refilled next is drained
    WN_T(tc);
    WN_ACC(2, tb, tc);                                 // barrier
    // stage k+2 goes into the slot every wave has just finished reading (k + 2 < S always holds here: 8 k-steps of
    // GEMM 2 follow).  Its pieces are issued one by one BETWEEN the MFMA triples below: an LDS-DMA issue costs the wave
    // 60-180 cycles, an MFMA holds the issue port for 8 of its 32 — interleaved, the address arithmetic and the issue
    // ride in the matrix pipe's shadow instead of in front of it.
    const StageSrc nxt = stage_src(k + 2, slot >= 1 ? slot - 1 : 2);
#if !WN_INTERLEAVE
#pragma unroll
    for (int i = 0; i < NPW1; ++i) issue_piece(nxt, i);
#endif
    WN_T(td);
    WN_ACC(3, tc, td);                                 // LDS-DMA issue
    int shift = 0;
    if (k < 3 * CH) {
      int c_, tap_;
      wn_stage_chunk_tap(k, CH, c_, tap_);
      shift = (tap_ - 1) * p.dil;
    }
    const int sub = (t0 + shift) & 3;
    const char* base = ldsb + slot * F_SLOT;
    const int colx = wave_n0 + l31 + sub;
    const char* bp = base + WN_A_BYTES + half * F_GS + (colx >> 5) * 1024 + (colx & 31) * 4;
    // A fragments run ONE row block ahead of the MFMAs that consume them (two register pairs in rotation): read right in front
    // of their MFMAs — as the first form of this loop did, one `ds_read_b128; s_waitcnt lgkmcnt(0)` per fragment — every one of the
    // 16 reads of a stage exposed the LDS latency to a wave with nothing else to issue (115 such waits in the kernel's code)
    wn_bf16x8 fah[2], fal[2];
    auto a_frag = [&](int mb) {
      fah[mb % 2] = *reinterpret_cast<const wn_bf16x8*>(base + mb * 2048 + lane * 16);
      fal[mb % 2] = *reinterpret_cast<const wn_bf16x8*>(base + mb * 2048 + 1024 + lane * 16);
    };
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (WN_EXP & 32) ? (float)(k + j) : *reinterpret_cast<const float*>(bp + j * 128);
    a_frag(0);                                             // lands under the split of the B fragment below
    wn_u32x4 bh4, bl4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned hh, ll;
      if (WN_EXP & 32) { hh = __float_as_uint(v[2 * j]); ll = __float_as_uint(v[2 * j + 1]); }
      else wn_split_pair(v[2 * j], v[2 * j + 1], hh, ll);
      bh4[j] = hh; bl4[j] = ll;
    }
    const wn_bf16x8 bh = __builtin_bit_cast(wn_bf16x8, bh4), bl = __builtin_bit_cast(wn_bf16x8, bl4);
#ifdef FST_STAMPS
    asm volatile("" ::"v"(bh), "v"(bl));
#endif
    WN_T(te);
    WN_ACC(4, td, te);                                 // B fragment: LDS reads + split
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
      if (mb + 1 < 8) a_frag(mb + 1);
      const wn_bf16x8 ah = fah[mb % 2], al = fal[mb % 2];
      if (WN_EXP & 1) { asm volatile("" ::"v"(al), "v"(ah), "v"(bh), "v"(bl)); }
      else {
        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[mb], 0, 0, 0);
        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[mb], 0, 0, 0);
        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[mb], 0, 0, 0);
      }
#if WN_INTERLEAVE
      if (mb < NPW1) {
        __builtin_amdgcn_sched_barrier(0);
        issue_piece(nxt, mb);
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
    }
    WN_T(tf);
    WN_ACC(5, te, tf);                                 // A fragments + MFMA issue
    slot = slot == 2 ? 0 : slot + 1;
  }
  WN_T(tg0);

  // ---------------------------------------------------------------- gate (between stage S1-1 and the first k-step of GEMM 2)
  // Drain: the two k-steps in flight (S1, S1+1: A only, L2 hits issued one and two stages ago) land, so from here on
  // the t,s stores below are OLDER than every LDS-DMA a counted vmcnt will wait for — a counted wait retires
  // everything older than what it waits for, i.e. the stores get two k-steps of MFMA time to complete before the wait
  // in front of k-step S1+2 can see them.
  wn_wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();                        // every wave is past its last B read: the B areas are free
  const int tcol = t0 + wave_n0;
  // accumulator-layout global accesses as raw buffer instructions: one descriptor per matrix of this batch element, ONE per-lane
  // offset, row / column terms in the scalar offset (no 64-bit per-lane addresses, no exec masks: see ws_acc_load)
  const WsLane wl = ws_lane((unsigned)(4 * half * L + l31) * 4u, tcol, L, lane);
  const __amdgpu_buffer_rsrc_t ts_rs = ws_rsrc(p.ts + (long long)b * (2 * n) * L);
  const __amdgpu_buffer_rsrc_t a_rs = ws_rsrc(ab), out_rs = ws_rsrc(p.out + (long long)b * n * L);
  const __amdgpu_buffer_rsrc_t an_rs = ws_rsrc(p.a_next ? p.a_next + (long long)b * n * L : p.out);
  wn_bf16x8 bh2[8], bl2[8];
#pragma unroll
  for (int blk = 0; blk < 4; ++blk) {
    float tv[16], sv[16], av[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float gt = acc[blk][r], gs = acc[blk + 4][r];
      tv[r] = wn_tanh(gt);
      sv[r] = wn_sigmoid(gs);
      const int row = blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      av[r] = row == n ? 1.0f : tv[r] * sv[r];         // acts[n] = 1 carries b_rs through GEMM 2 (rows > n: tanh(0)·σ(0) = 0)
    }
    const int rows_valid = (WN_EXP & 8) ? 0 : n - blk * 32;               // may be <= 0: nothing stored
    ws_acc_store(tv, ts_rs, wl, blk * 32, tcol, rows_valid, L);
    ws_acc_store(sv, ts_rs, wl, n + blk * 32, tcol, rows_valid, L);
    if (p.acts) {
      float aw[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) aw[r] = tv[r] * sv[r];
      wn_acc_store(aw, p.acts + ((long long)b * n + blk * 32) * L, rows_valid, L, tcol, lane);
    }
    // GEMM 2 accumulates onto what its rows are added to — the layer input for the residual rows (a_next = a + r), the
    // running skip sum for the skip rows (out += r) — read here, in accumulator layout, into the two accumulators the
    // gate has just consumed: the round trip hides under the remaining gate blocks and the first k-steps
    {
      const int rows_e = (WN_EXP & 16) ? 0 : n - blk * 32;
      ws_acc_load(acc[blk], a_rs, wl, blk * 32, tcol, p.last ? 0 : rows_e, L);
      ws_acc_load(acc[blk + 4], out_rs, wl, blk * 32, tcol, ((p.first != 0) | ((WN_EXP & 64) != 0)) ? 0 : rows_e, L);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      wn_u32x4 h4, l4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned hh, ll;
        wn_split_pair(av[8 * s + 2 * j], av[8 * s + 2 * j + 1], hh, ll);
        h4[j] = hh; l4[j] = ll;
      }
      bh2[2 * blk + s] = __builtin_bit_cast(wn_bf16x8, h4);
      bl2[2 * blk + s] = __builtin_bit_cast(wn_bf16x8, l4);
    }
  }
  asm volatile("" ::: "memory");                       // the stores stay in front of the LDS-DMA issued below
  WN_T(tg1);
  WN_ACC(6, tg0, tg1);                                 // drain + gate + t,s stores (issue)

  // ---------------------------------------------------------------- GEMM 2: [a ; out] += [W_rs | b] · [acts ; 1]
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    const int k = S1 + ks;
    if (ks >= 2) wait_for(k + 1, S);                   // k-steps S1 and S1+1 landed at the drain above
    if (ks >= 1) __builtin_amdgcn_s_barrier();
    const bool more = k + 2 < S;
    const StageSrc nxt = stage_src(more ? k + 2 : k, slot >= 1 ? slot - 1 : 2);
#if !WN_INTERLEAVE
    if (more) {
#pragma unroll
      for (int i = 0; i < 16 / NW; ++i) issue_piece(nxt, i);
    }
#endif
    const char* base = ldsb + slot * F_SLOT;
    // (fragments one row block ahead, as in GEMM 1; on the last layer the residual-row blocks 0-3 are neither read nor multiplied)
    wn_bf16x8 fah[2], fal[2];
    auto a_frag = [&](int mb) {
      fah[mb % 2] = *reinterpret_cast<const wn_bf16x8*>(base + mb * 2048 + lane * 16);
      fal[mb % 2] = *reinterpret_cast<const wn_bf16x8*>(base + mb * 2048 + 1024 + lane * 16);
    };
    if (!p.last) a_frag(0);
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
      if (mb + 1 < 8 && !(p.last && mb + 1 < 4)) a_frag(mb + 1);
      if (!(p.last && mb < 4) && !((WN_EXP & 64) && mb >= 4)) {                       // wave-uniform
        const wn_bf16x8 ah = fah[mb % 2], al = fal[mb % 2];
        if (WN_EXP & 1) { asm volatile("" ::"v"(al), "v"(ah), "v"(bh2[ks]), "v"(bl2[ks])); }
        else {
          acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh2[ks], acc[mb], 0, 0, 0);
          acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl2[ks], acc[mb], 0, 0, 0);
          acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh2[ks], acc[mb], 0, 0, 0);
        }
      }
#if WN_INTERLEAVE
      if (more && mb >= 4) {                           // four pieces per wave, behind the skip-row blocks (live on every layer)
        __builtin_amdgcn_sched_barrier(0);
        issue_piece(nxt, mb - 4);
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
    }
    slot = slot == 2 ? 0 : slot + 1;
  }

  WN_T(tg2);
  WN_ACC(7, tg1, tg2);                                 // GEMM 2
  // ---------------------------------------------------------------- a_next = a + r[:n];  out (+)= r[n:]
  // straight from the accumulators (the operands were read into them before GEMM 2): stores only, nothing to wait for
#pragma unroll
  for (int blk = 0; blk < ((WN_EXP & 16) ? 0 : 4); ++blk) {
    const int rows_e = n - blk * 32;
    if (!p.last) ws_acc_store(acc[blk], an_rs, wl, blk * 32, tcol, rows_e, L);
    if (!(WN_EXP & 64)) ws_acc_store(acc[blk + 4], out_rs, wl, blk * 32, tcol, rows_e, L);
  }
  WN_T(tg3);
  WN_ACC(8, tg2, tg3);                                 // final epilogue (issue; includes waiting for the operand loads)
  WN_ACC(9, ts0, tg3);                                 // whole wave
  WN_FLUSH;
}

template <int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void wn_layer_fwd_kernel(WnFwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // Workgroup → tile: ids that differ by a multiple of 8 share an XCD (round-robin dispatch), so give each XCD a
  // contiguous run of tiles: the tiles of one sequence, which re-read each other's halo for the dilated taps, then
  // meet in one L2.  Speed only.
  int wg = blockIdx.x;
  if ((p.n_wg & 7) == 0) wg = (wg & 7) * (p.n_wg >> 3) + (wg >> 3);
  const int b = wg / p.tiles_per_seq;
  const int t0 = (wg - b * p.tiles_per_seq) * (32 * NW);
  wn_fwd_tile<NW>(*(WnFwdArgs*)__builtin_amdgcn_kernarg_segment_ptr(), b, t0, reinterpret_cast<char*>(lds));
}

// ------------------------------------------------------------------------------------------------
// The forward of a whole WN stack in ONE persistent launch
//
// Launched layer by layer, every workgroup of the forward kernel walks the same phases at the same time — operand fetch, GEMM 1,
// the gate with its t,s stores, GEMM 2, the a_next / out stores — and the memory system alternates between idle (the GEMMs) and
// saturated (245 MB of stores + 195 MB of loads per layer in two bursts per tile round: the epilogue alone is 126 MB that no
// CU can multiply under; stamps: a third of a wave's life, profiles/r03_wn_fwd_stamps.txt).  Here ONE workgroup walks all tiles
// of its batch elements through all layers (the same tile body, the layer's arguments from tables in the kernel-argument
// segment; a_next / out of one layer are read by the next through the SAME CU: s_waitcnt vmcnt(0) + barrier between tiles, as in
// wn_stack_bwd_kernel), nothing synchronises the workgroups with each other, and `stagger` starts every other workgroup half a
// tile late: one half of the chip stores while the other multiplies.
// ------------------------------------------------------------------------------------------------
struct WnFwdStackParams {
  WnFwdParams layer[WS_MAXL];      // every layer's arguments as the per-layer launch would get them (a = the previous layer's a_next)
  int nl, stagger;
};

__global__ __launch_bounds__(512, 1) void wn_stack_fwd_kernel(WnFwdStackParams ps_by_value) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // a layer's arguments are read where they are used, with scalar loads from the kernel-argument segment at a run-time offset
  // (invariant memory: nothing has to stay in registers across the tile body)
  typedef const __attribute__((address_space(4))) WnFwdStackParams StackArgs;
  StackArgs& ps = *(StackArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  (void)ps_by_value;
  char* const ldsb = reinterpret_cast<char*>(lds);
  if ((blockIdx.x >> 3) & 1)                                // workgroup ids 8 apart share an XCD: alternate WITHIN each XCD
    for (int i = 0; i < ps.stagger; ++i) __builtin_amdgcn_s_sleep(127);
  const int B = ps.layer[0].B, passes = ps.layer[0].tiles_per_seq;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    for (int layer = 0; layer < ps.nl; ++layer) {
      for (int pass = 0; pass < passes; ++pass) {
        WnFwdArgs* qp = &ps.layer[layer];
        asm volatile("" : "+s"(qp));                       // (opaque: the layer's arguments are loaded inside the tile body, not hoisted out of the loops)
        wn_fwd_tile<8>(*qp, b, pass * 256, ldsb);
        // this tile's stores (a_next, out, t,s) are complete and every wave is past its LDS reads before the next tile's
        // LDS-DMA refills the ring and reads what was stored
        wn_wait_vmcnt<0>();
        __syncthreads();
      }
    }
  }
}

extern "C" int fst_wn_stack_fwd_ok(int n, int h, int L, int nl) {
  return n > 0 && n < 128 && h > 0 && L > 0 && L % 256 == 0 && nl >= 1 && nl <= WS_MAXL;
}

extern "C" int fst_wn_stack_fwd(const float* const* a_in, const int64_t* a_bs, const void* const* images, int64_t image_bytes,
                                float* const* ts, float* const* a_next, const float* u0, int64_t u0_bs, float* out, int nl, int B,
                                int L, int n, int h, int64_t numel_a, void* stream) {
  FST_REQUIRE(a_in && a_bs && images && ts && a_next && u0 && out, "fst_wn_stack_fwd: null operand");
  FST_REQUIRE(fst_wn_stack_fwd_ok(n, h, L, nl) && B > 0, "fst_wn_stack_fwd: B=%d L=%d n=%d h=%d nl=%d (needs n < 128, L %% 256 == 0, "
              "1 <= nl <= %d)", B, L, n, h, nl, WS_MAXL);
  FST_REQUIRE((long long)B * n * L == (long long)numel_a, "fst_wn_stack_fwd: B*n*L = %d*%d*%d does not match the element count %lld "
              "of the [B, n, L] tensors", B, n, L, (long long)numel_a);
  FST_REQUIRE(image_bytes == fst_wn_image_bytes(n, h), "fst_wn_stack_fwd: images are %lld bytes, expected %lld",
              (long long)image_bytes, (long long)fst_wn_image_bytes(n, h));
  FST_REQUIRE(B == 1 || u0_bs >= (int64_t)h * L, "fst_wn_stack_fwd: batch stride of u0 smaller than a sample");
  auto al16 = [](const void* q) { return q == nullptr || (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  FST_REQUIRE(u0_bs % 4 == 0 && al16(u0) && al16(out), "fst_wn_stack_fwd: needs 16-byte aligned tensors");
  WnFwdStackParams p = {};
  for (int i = 0; i < nl; ++i) {
    FST_REQUIRE(a_in[i] && images[i] && ts[i] && (a_next[i] || i == nl - 1), "fst_wn_stack_fwd: null operand of layer %d", i);
    FST_REQUIRE(a_bs[i] >= (int64_t)n * L && a_bs[i] % 4 == 0 && al16(a_in[i]) && al16(images[i]) && al16(ts[i]) && al16(a_next[i]),
                "fst_wn_stack_fwd: layer %d: batch stride %lld smaller than a sample, or an operand that is not 16-byte aligned", i,
                (long long)a_bs[i]);
    WnFwdParams& q = p.layer[i];
    q.a = a_in[i]; q.a_bs = a_bs[i]; q.u0 = u0; q.u0_bs = u0_bs; q.img = static_cast<const char*>(images[i]);
    q.ts = ts[i]; q.acts = nullptr; q.a_next = i == nl - 1 ? nullptr : a_next[i]; q.out = out;
    q.B = B; q.L = L; q.n = n; q.h = h; q.dil = 1 << i; q.first = i == 0; q.last = i == nl - 1;
    q.CH = wn_ch(n); q.CH2 = wn_ch2(h); q.tiles_per_seq = L / 256; q.n_wg = 0;
  }
  p.nl = nl;
  // diagnostics: every other workgroup of an XCD starts `stagger` x 3.4 us late (measured: no gain at any value — the phases of a tile
  // are bound inside the CU, not by the memory system the CUs share; profiles/r04_wn_fwd_stack_timing.txt)
  static const int stagger_env = getenv("FST_WN_FWD_STAGGER") ? atoi(getenv("FST_WN_FWD_STAGGER")) : -1;
  p.stagger = stagger_env >= 0 ? stagger_env : 0;
  const int cus = fst_cu_count() > 0 ? fst_cu_count() : 256;
  if (int rc = fst_allow_full_lds((const void*)wn_stack_fwd_kernel, "fst_wn_stack_fwd")) return rc;
  hipLaunchKernelGGL(wn_stack_fwd_kernel, dim3((unsigned)(B < cus ? B : cus)), dim3(512), WN_FWD_LDS(8), (hipStream_t)stream, p);
  FST_LAUNCH_CHECK();
  return 0;
}

extern "C" int fst_wn_layer_fwd(const float* a, int64_t a_bs, const float* u0, int64_t u0_bs, const void* image,
                                int64_t image_bytes, float* ts, float* acts, float* a_next, float* out, int first, int last,
                                int B, int L, int n, int h, int dil, int64_t numel_a, void* stream) {
  FST_REQUIRE(a && u0 && image && ts && out && (last || a_next), "fst_wn_layer_fwd: null operand");
  FST_REQUIRE(B > 0 && L > 0 && n > 0 && n < 128 && h > 0 && dil > 0, "fst_wn_layer_fwd: B=%d L=%d n=%d h=%d dil=%d (needs n < 128)",
              B, L, n, h, dil);
  FST_REQUIRE((long long)B * n * L == (long long)numel_a, "fst_wn_layer_fwd: B*n*L = %d*%d*%d does not match the element count %lld "
              "of the [B, n, L] tensors", B, n, L, (long long)numel_a);
  FST_REQUIRE(image_bytes == fst_wn_image_bytes(n, h), "fst_wn_layer_fwd: image is %lld bytes, expected %lld",
              (long long)image_bytes, (long long)fst_wn_image_bytes(n, h));
  FST_REQUIRE(B == 1 || (a_bs >= (int64_t)n * L && u0_bs >= (int64_t)h * L), "fst_wn_layer_fwd: batch stride smaller than a sample");
  auto al16 = [](const void* q) { return q == nullptr || (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  FST_REQUIRE(L % 4 == 0 && a_bs % 4 == 0 && u0_bs % 4 == 0 && al16(a) && al16(u0) && al16(image) && al16(ts) && al16(acts) &&
              al16(a_next) && al16(out), "fst_wn_layer_fwd: needs L %% 4 == 0 and 16-byte aligned tensors (L=%d)", L);
  WnFwdParams p;
  p.a = a; p.a_bs = a_bs; p.u0 = u0; p.u0_bs = u0_bs; p.img = static_cast<const char*>(image);
  p.ts = ts; p.acts = acts; p.a_next = a_next; p.out = out;
  p.B = B; p.L = L; p.n = n; p.h = h; p.dil = dil; p.first = first ? 1 : 0; p.last = last ? 1 : 0;
  p.CH = wn_ch(n); p.CH2 = wn_ch2(h);
  // 256-sample tiles (8 waves) when they divide the sequence and still fill the chip, else 128-sample tiles (4 waves, two per CU)
  static const int nw_env = getenv("FST_WN_FWD_NW") ? atoi(getenv("FST_WN_FWD_NW")) : 0;        // diagnostics: force 4 or 8
  const int cus = fst_cu_count() > 0 ? fst_cu_count() : 256;
  int nw = (L % 256 == 0 && (long long)B * (L / 256) >= cus) ? 8 : 4;
  if (nw_env == 4 || nw_env == 8) nw = nw_env;
  const int tn = 32 * nw;
  p.tiles_per_seq = (L + tn - 1) / tn;
  p.n_wg = B * p.tiles_per_seq;
  if (nw == 8) {
    if (int rc = fst_allow_full_lds((const void*)wn_layer_fwd_kernel<8>, "fst_wn_layer_fwd")) return rc;
    hipLaunchKernelGGL(wn_layer_fwd_kernel<8>, dim3((unsigned)p.n_wg), dim3(512), WN_FWD_LDS(8), (hipStream_t)stream, p);
  } else {
    if (int rc = fst_allow_full_lds((const void*)wn_layer_fwd_kernel<4>, "fst_wn_layer_fwd")) return rc;
    hipLaunchKernelGGL(wn_layer_fwd_kernel<4>, dim3((unsigned)p.n_wg), dim3(256), WN_FWD_LDS(4), (hipStream_t)stream, p);
  }
  FST_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// backward through res_skip and the gate:  dacts = W_rsᵀ·[d_a_next ; d_out],  dg = gate'(t, s)·dacts
//
// GEMM 3 has M = n (4 row blocks), K = 2n (n on the last layer): 12 MFMAs per 16-deep stage against 16 KiB of operand
// fill, and its inputs and outputs (d_a_next, d_out, t, s read once; dg written once) are 378 MB per layer — HBM-bound by a
// factor seven, so the kernel is shaped for streaming: small workgroups (one batch element × 128 samples, 50 KB of LDS,
// two per CU — WN_BWD_OCC), the t,s tiles of the first row block prefetched before the GEMM, the next block's while one is gated.
// Image: S3 = 2·CH (CH on the last layer) stages × 4 row blocks × (1 KiB hi + 1 KiB lo), then 16 B of zeros; stage (src, c):
// lane l holds W_rs[src·n + 16c + 8(l>>5) + j][blk·32 + (l&31)] — W_rs transposed, rows = acts channels.
// ------------------------------------------------------------------------------------------------
#ifndef WN_BWD_OCC
#define WN_BWD_OCC 2                                 // workgroups per CU the register budget is set for: 1024 tiles of the metric shape = two full
                                                     // rounds of 512 slots (three per CU: 1.33 rounds, 77 -> 70 us; and no scratch)
#endif
#define WN_BW_NB 4                                   // column blocks of a B row group (no tap shift: no spill block)
#define WN_BW_GS (WN_BW_NB * 1024 + 128)
#define WN_BW_A (4 * 2048)
#define WN_BW_SLOT (WN_BW_A + 2 * WN_BW_GS)
#define WN_BW_LDS (3 * WN_BW_SLOT)

struct WnPackBwdParams {
  const float* rs_w;   // [2n][n] (last: [n][n])
  int n, last, CH;
  uint4* img;
  int acc_order;       // the d_a stages in the k-order in which a 32x32 accumulator tile delivers its rows as a B operand
};

__global__ __launch_bounds__(64) void wn_pack_bwd_kernel(WnPackBwdParams p) {
  const int lane = threadIdx.x, blk = blockIdx.x & 3, st = blockIdx.x >> 2;
  const int n = p.n, hh = lane >> 5, m = blk * 32 + (lane & 31);
  const int src = p.last ? 0 : st / p.CH, c = st - src * p.CH;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    // acc_order (fst_wn_stack_bwd keeps d_a in its accumulators and multiplies straight out of them): element j of lane half
    // hh multiplies row 16c + 8(j>>2) + 4hh + (j&3) — registers 8s..8s+7 of the tile of row block c>>1, s = c&1
    const bool perm = p.acc_order && !p.last && src == 0;
    const int ch = 16 * c + (perm ? 8 * (j >> 2) + 4 * hh + (j & 3) : 8 * hh + j);
    v[j] = (m < n && ch < n) ? p.rs_w[(long long)(src * n + ch) * n + m] : 0.f;
  }
  unsigned hi[4], lo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) wn_split_pair(v[2 * j], v[2 * j + 1], hi[j], lo[j]);
  uint4* dst = p.img + ((long long)st * 4 + blk) * 128 + lane;
  dst[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
  dst[64] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
  if (blockIdx.x == 0 && lane == 0) p.img[(long long)gridDim.x * 128] = make_uint4(0u, 0u, 0u, 0u);
}

extern "C" int64_t fst_wn_bwd_image_bytes(int n, int last) {
  if (n <= 0) return -1;
  return (int64_t)(last ? 1 : 2) * wn_ch(n) * WN_BW_A + 16;
}

extern "C" int fst_wn_pack_bwd(const float* rs_w, int n, int last, int acc_order, void* image, int64_t image_bytes, void* stream) {
  FST_REQUIRE(rs_w && image && n > 0 && n <= 128, "fst_wn_pack_bwd: bad arguments (n=%d, needs n <= 128)", n);
  FST_REQUIRE(image_bytes == fst_wn_bwd_image_bytes(n, last), "fst_wn_pack_bwd: image is %lld bytes, expected %lld",
              (long long)image_bytes, (long long)fst_wn_bwd_image_bytes(n, last));
  FST_REQUIRE((reinterpret_cast<uintptr_t>(image) & 15) == 0, "fst_wn_pack_bwd: image must be 16-byte aligned");
  WnPackBwdParams p = {rs_w, n, last ? 1 : 0, wn_ch(n), static_cast<uint4*>(image), acc_order ? 1 : 0};
  const int stages = (last ? 1 : 2) * p.CH;
  hipLaunchKernelGGL(wn_pack_bwd_kernel, dim3((unsigned)(stages * 4)), dim3(64), 0, (hipStream_t)stream, p);
  FST_LAUNCH_CHECK();
  return 0;
}

struct WnBwdParams {
  const float* d_a;    // [B][n][L], null on the last layer
  const float* d_out;  // [B][n][L]
  const float* ts;     // [B][2n][L]
  const char* img;
  float* dg;           // [B][2n][L]
  float* row_sums;     // optional [256][n_wg]: per-workgroup Σ_t dg[row] (rows [0, 2n)) — the in_layer / cond_layer bias gradient
  int B, L, n, last, CH, tiles_per_seq, n_wg;
};

__global__ __launch_bounds__(256, WN_BWD_OCC) void wn_layer_bwd_kernel(WnBwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* const ldsb = reinterpret_cast<char*>(lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int half = lane >> 5, l31 = lane & 31;
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
  int wg = blockIdx.x;
  if ((p.n_wg & 7) == 0) wg = (wg & 7) * (p.n_wg >> 3) + (wg >> 3);
  const int b = wg / p.tiles_per_seq;
  const int t0 = (wg - b * p.tiles_per_seq) * WN_TN;
  const int wave_n0 = wave_s * 32;
  const int L = p.L, n = p.n, CH = p.CH;
  const int S3 = (p.last ? 1 : 2) * CH;
  const char* const zero16 = p.img + (long long)S3 * WN_BW_A;
  const float* const ts_b = p.ts + (long long)b * (2 * n) * L;
  const int tcol = t0 + wave_n0;

  // pieces per stage: 8 of A, 2 x 4 of B = 16 -> exactly four per wave
  auto issue = [&](int k, int slot) {
    char* const sl = ldsb + slot * WN_BW_SLOT;
    const char* asrc = p.img + (long long)k * WN_BW_A;
    const bool from_da = !p.last && k < CH;
    const int c = from_da ? k : k - (p.last ? 0 : CH);
    const float* xb = (from_da ? p.d_a : p.d_out) + ((long long)b * n + 16 * c) * L;
    const int c_count = min(16, n - 16 * c);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = wave_s + 4 * i;
      if (idx < 8) {
        __builtin_amdgcn_global_load_lds(WN_GLOBAL_PTR(asrc + idx * 1024 + lane * 16), WN_LDS_VOID(sl + idx * 1024), 16, 0, 0);
      } else {
        const int bi = idx - 8;
        const int gq = bi >> 2, m = bi & 3;
        const int row = 8 * gq + (lane >> 3);
        const int t = t0 + 32 * m + 4 * (lane & 7);
        const bool ok = row < c_count && t < L;
        const char* src = ok ? reinterpret_cast<const char*>(xb + ((long long)row * L + t)) : zero16;
        __builtin_amdgcn_global_load_lds(WN_GLOBAL_PTR(src), WN_LDS_VOID(sl + WN_BW_A + gq * WN_BW_GS + m * 1024), 16, 0, 0);
      }
    }
  };

  f32x16 acc[4];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;

  // The gate halves of the first row block, issued BEFORE any LDS-DMA: they are then the oldest vector-memory operations
  // of the wave, so every counted wait below (which retires everything older than what it leaves in flight) covers them
  // and the counts stay "pieces of the next stage".  First use is after the GEMM.
  float4 qt[4], qs[4];
  wn_fetch_tile(qt, ts_b, n, L, tcol, lane);
  wn_fetch_tile(qs, ts_b + (long long)n * L, n, L, tcol, lane);
  asm volatile("" ::: "memory");
  issue(0, 0);
  if (S3 > 1) issue(1, 1);
  int slot = 0;
  for (int k = 0; k < S3; ++k) {
    if (k + 1 < S3) wn_wait_vmcnt<4>(); else wn_wait_vmcnt<0>();     // stage k landed; stage k+1's 4 pieces may fly on
    __builtin_amdgcn_s_barrier();
    if (k + 2 < S3) issue(k + 2, slot >= 1 ? slot - 1 : 2);
    const char* base = ldsb + slot * WN_BW_SLOT;
    const int colx = wave_n0 + l31;
    const char* bp = base + WN_BW_A + half * WN_BW_GS + (colx >> 5) * 1024 + (colx & 31) * 4;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float*>(bp + j * 128);
    wn_u32x4 bh4, bl4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned hh, ll;
      wn_split_pair(v[2 * j], v[2 * j + 1], hh, ll);
      bh4[j] = hh; bl4[j] = ll;
    }
    const wn_bf16x8 bh = __builtin_bit_cast(wn_bf16x8, bh4), bl = __builtin_bit_cast(wn_bf16x8, bl4);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      const wn_bf16x8 ah = *reinterpret_cast<const wn_bf16x8*>(base + mb * 2048 + lane * 16);
      const wn_bf16x8 al = *reinterpret_cast<const wn_bf16x8*>(base + mb * 2048 + 1024 + lane * 16);
      acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[mb], 0, 0, 0);
      acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[mb], 0, 0, 0);
      acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[mb], 0, 0, 0);
    }
    slot = slot == 2 ? 0 : slot + 1;
  }
  __syncthreads();                                     // every wave is past its last fragment read: the ring becomes tiles
  float* const tile = reinterpret_cast<float*>(ldsb + wave_s * WN_TILE_BYTES);
  float* const rsum = reinterpret_cast<float*>(ldsb + 4 * WN_TILE_BYTES);      // [4 waves][256] row sums of this workgroup's dg tile
  float* const rsum_w = rsum + wave_s * 256;
  if (p.row_sums) {
#pragma unroll
    for (int w = 0; w < 4; ++w) rsum[w * 256 + tid] = 0.f;
    __syncthreads();
  }
  float* const dg_b = p.dg + (long long)b * (2 * n) * L;
#pragma unroll
  for (int blk = 0; blk < 4; ++blk) {
    const int rows_valid = n - blk * 32;
    float tv[16], sv[16], gt[16], gs[16];
    wn_tile_to_acc(tv, qt, tile, lane);
    wn_tile_to_acc(sv, qs, tile, lane);
    if (blk < 3) {                                     // next row block's gate halves, in flight while this one is gated
      wn_fetch_tile(qt, ts_b + (long long)((blk + 1) * 32) * L, n - (blk + 1) * 32, L, tcol, lane);
      wn_fetch_tile(qs, ts_b + (long long)(n + (blk + 1) * 32) * L, n - (blk + 1) * 32, L, tcol, lane);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float d = acc[blk][r], t = tv[r], s = sv[r];
      gt[r] = d * s * (1.f - t * t);
      gs[r] = d * t * s * (1.f - s);
    }
    wn_store_tile<0>(gt, tile, dg_b + (long long)(blk * 32) * L, nullptr, rows_valid, L, tcol, lane);
    if (p.row_sums) wn_tile_row_sums(tile, rsum_w, blk * 32, rows_valid, L, tcol, lane);
    wn_store_tile<0>(gs, tile, dg_b + (long long)(n + blk * 32) * L, nullptr, rows_valid, L, tcol, lane);
    if (p.row_sums) wn_tile_row_sums(tile, rsum_w, n + blk * 32, rows_valid, L, tcol, lane);
  }
  if (p.row_sums) {
    __syncthreads();
    // [256][n_wg]: the caller's sum runs over the contiguous axis; waves added in a fixed order
    p.row_sums[(long long)tid * p.n_wg + wg] = (rsum[tid] + rsum[256 + tid]) + (rsum[512 + tid] + rsum[768 + tid]);
  }
}

extern "C" int fst_wn_layer_bwd(const float* d_a, const float* d_out, const float* ts, const void* image, int64_t image_bytes,
                                float* dg, float* row_sums, int64_t row_sums_rows, int last, int B, int L, int n,
                                int64_t numel_a, void* stream) {
  FST_REQUIRE(d_out && ts && image && dg && (last || d_a), "fst_wn_layer_bwd: null operand");
  FST_REQUIRE(B > 0 && L > 0 && n > 0 && n <= 128, "fst_wn_layer_bwd: B=%d L=%d n=%d (needs n <= 128)", B, L, n);
  FST_REQUIRE((long long)B * n * L == (long long)numel_a, "fst_wn_layer_bwd: B*n*L = %d*%d*%d does not match the element count %lld "
              "of the [B, n, L] tensors", B, n, L, (long long)numel_a);
  FST_REQUIRE(image_bytes == fst_wn_bwd_image_bytes(n, last), "fst_wn_layer_bwd: image is %lld bytes, expected %lld",
              (long long)image_bytes, (long long)fst_wn_bwd_image_bytes(n, last));
  auto al16 = [](const void* q) { return q == nullptr || (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  FST_REQUIRE(L % 4 == 0 && al16(d_a) && al16(d_out) && al16(ts) && al16(image) && al16(dg),
              "fst_wn_layer_bwd: needs L %% 4 == 0 and 16-byte aligned tensors (L=%d)", L);
  WnBwdParams p;
  p.d_a = last ? nullptr : d_a; p.d_out = d_out; p.ts = ts; p.img = static_cast<const char*>(image); p.dg = dg;
  p.row_sums = row_sums;
  p.B = B; p.L = L; p.n = n; p.last = last ? 1 : 0; p.CH = wn_ch(n);
  p.tiles_per_seq = (L + WN_TN - 1) / WN_TN;
  p.n_wg = B * p.tiles_per_seq;
  FST_REQUIRE(row_sums == nullptr || row_sums_rows == p.n_wg, "fst_wn_layer_bwd: row_sums has %lld rows, the launch has %d "
              "workgroups (B x ceil(L/128))", (long long)row_sums_rows, p.n_wg);
  if (int rc = fst_allow_full_lds((const void*)wn_layer_bwd_kernel, "fst_wn_layer_bwd")) return rc;
  hipLaunchKernelGGL(wn_layer_bwd_kernel, dim3((unsigned)p.n_wg), dim3(256), WN_BW_LDS, (hipStream_t)stream, p);
  FST_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// data gradient of the in_layer + cond_layer:   d_a = d_a_next + W_inᵀ (*) dg  (3 dilated taps),   d_u0 += W_condᵀ · dg
//
// The LDS-DMA path, not the matrix pipe, bounds these GEMMs (tools/dma_ring_probe.hip: ≈34 GB/s per CU, 8.7 TB/s chip-wide,
// whatever the prefetch depth): so the kernel is shaped to move few bytes per MFMA.
//   * one stage = 16 channels of dg with ALL THREE taps: the B window [16 ch][256 + 2·dil (+3)] is fetched once and each tap
//     reads it at its own column offset (a third of the B bytes for small dilations; the per-tap form fetched three windows);
//   * the 25 conditioning rows ride along as a fifth row block of the centre tap instead of a second pass over dg;
//   * a workgroup = 8 waves = one batch element × 256 time samples (each wave 32 samples × 5 row blocks): the weight bytes
//     per output sample are half those of a 128-sample tile.
// Image: per 16-channel chunk 13 row blocks × (1 KiB hi + 1 KiB lo): [tap 0: 4 blocks of d_a rows][tap 1: 4 + the d_u0 block]
// [tap 2: 4]; tap τ multiplies dg at t + (1 − τ)·dil.  Ring of 3 slots (2 when the window of a large dilation needs the room).
// ------------------------------------------------------------------------------------------------
#define DG_NCB 2                                     // 32-sample column blocks per wave
#define DG_TN (8 * 32 * DG_NCB)                       // time samples per workgroup (8 waves)
#define DG_A_BLOCKS 13
#define DG_A_BYTES (DG_A_BLOCKS * 2048)
#ifndef DG_EXP
#define DG_EXP 0   // diagnostic builds only (timing, wrong results): 1 no MFMA, 2 B pieces from the zero block, 4 A pieces from the
#endif             // zero block, 8 no epilogue stores, 16 no epilogue tiles at all, 32 no operand loads, 64 no LDS-DMA issued at all

struct WnPackDgradParams {
  const float* in_w;    // [2n][n][3]
  const float* cond_w;  // [2n][h]
  int n, h, CHK;
  uint4* img;
};

__global__ __launch_bounds__(64) void wn_pack_dgrad_kernel(WnPackDgradParams p) {
  const int lane = threadIdx.x, ab = blockIdx.x % DG_A_BLOCKS, c = blockIdx.x / DG_A_BLOCKS;
  const int n = p.n, h = p.h, hh = lane >> 5, l31 = lane & 31;
  // block ab of the chunk -> (tap, row block)
  int tap, blk;
  if (ab < 4) { tap = 0; blk = ab; } else if (ab < 9) { tap = 1; blk = ab - 4; } else { tap = 2; blk = ab - 9; }
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 16 * c + 8 * hh + j;                   // dg channel = output row of the forward convs
    float w = 0.f;
    if (k < 2 * n) {
      if (blk < 4) {
        const int m = blk * 32 + l31;
        if (m < n) w = p.in_w[((long long)k * n + m) * 3 + tap];
      } else {
        if (l31 < h) w = p.cond_w[(long long)k * h + l31];
      }
    }
    v[j] = w;
  }
  unsigned hi[4], lo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) wn_split_pair(v[2 * j], v[2 * j + 1], hi[j], lo[j]);
  uint4* dst = p.img + (long long)blockIdx.x * 128 + lane;
  dst[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
  dst[64] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
  if (blockIdx.x == 0 && lane == 0) p.img[(long long)gridDim.x * 128] = make_uint4(0u, 0u, 0u, 0u);
}

extern "C" int64_t fst_wn_dgrad_image_bytes(int n) {
  if (n <= 0) return -1;
  return (int64_t)((2 * n + 15) / 16) * DG_A_BYTES + 16;
}

extern "C" int fst_wn_pack_dgrad(const float* in_w, const float* cond_w, int n, int h, int ntaps, void* image, int64_t image_bytes,
                                 void* stream) {
  FST_REQUIRE(in_w && cond_w && image && n > 0 && n <= 128 && h > 0 && h <= 32, "fst_wn_pack_dgrad: needs n <= 128, h <= 32 (n=%d h=%d)", n, h);
  FST_REQUIRE(ntaps == 3, "fst_wn_pack_dgrad: the fused WN kernels are 3-tap (in_w [2n][n][3]); got %d taps", ntaps);
  FST_REQUIRE(image_bytes == fst_wn_dgrad_image_bytes(n), "fst_wn_pack_dgrad: image is %lld bytes, expected %lld",
              (long long)image_bytes, (long long)fst_wn_dgrad_image_bytes(n));
  FST_REQUIRE((reinterpret_cast<uintptr_t>(image) & 15) == 0, "fst_wn_pack_dgrad: image must be 16-byte aligned");
  WnPackDgradParams p = {in_w, cond_w, n, h, (2 * n + 15) / 16, static_cast<uint4*>(image)};
  hipLaunchKernelGGL(wn_pack_dgrad_kernel, dim3((unsigned)(p.CHK * DG_A_BLOCKS)), dim3(64), 0, (hipStream_t)stream, p);
  FST_LAUNCH_CHECK();
  return 0;
}

struct WnDgradParams {
  const float* dg;      // [B][2n][L]
  const char* img;
  const float* d_a;     // [B][n][L] residual cotangent, or null
  float* d_a_new;       // [B][n][L]
  float* d_u0;          // [B][h][L] with batch stride d_u0_bs (a channel slice of a wider tensor passes as it is), accumulated
  long long d_u0_bs;
  float* row_sums;      // optional [128][n_wg]: per-workgroup Σ_t d_a_new[row] — the res rows of the next res_skip bias gradient
  int B, L, n, h, dil, CHK, tiles_per_seq, n_wg;
  int nblkw;            // 32-sample column blocks of the window
  int gsw;              // bytes per 8-channel row group of the window
  int slot;             // bytes per ring slot
  int ns;               // ring slots (3, or 2 for wide windows)
};

template <int N>
__device__ __forceinline__ void wn_wait_sw(int n) {
  // counted wait for a wave-uniform run-time count (the immediates are instantiated below)
  if (n == N) { wn_wait_vmcnt<N>(); return; }
  if constexpr (N > 0) wn_wait_sw<N - 1>(n);
}

__global__ __launch_bounds__(512, 2) void wn_layer_dgrad_kernel(WnDgradParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* const ldsb = reinterpret_cast<char*>(lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int half = lane >> 5, l31 = lane & 31;
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_n0 = wave_s * (32 * DG_NCB);
  const int L = p.L, n = p.n, CHK = p.CHK, dil = p.dil;
  const char* const zero16 = p.img + (long long)CHK * DG_A_BYTES;
  const int NI = DG_A_BLOCKS * 2 + 2 * p.nblkw;        // 1-KiB pieces per stage
  int my_pieces = (NI - wave_s + 7) >> 3;              // pieces idx = wave + 8 i < NI
  if (DG_EXP & 128) {                                  // diagnostics: no A pieces at all
    my_pieces = 0;
    for (int idx = wave_s; idx < NI; idx += 8) my_pieces += idx >= 2 * DG_A_BLOCKS;
  }
  const int depth = p.ns - 1;                          // stages in flight

  // Workgroups are persistent: the launch has one per CU (the ring leaves room for one) and workgroup j walks the virtual
  // ids j, j + gridDim, ...  Ids that differ by a multiple of 8 share an XCD (round-robin dispatch), so each XCD gets a
  // contiguous run of tiles — the tiles of one sequence, which re-read each other's halo, meet in one L2.
  auto tile_of = [&](int v, int& b, int& t0) {
    int wg = v;
    if ((p.n_wg & 7) == 0) wg = (wg & 7) * (p.n_wg >> 3) + (wg >> 3);
    b = wg / p.tiles_per_seq;
    t0 = (wg - b * p.tiles_per_seq) * DG_TN;
    return wg;
  };
  auto issue = [&](int b, int t0, int c, int slot) {
    char* const sl = ldsb + slot * p.slot;
    const char* asrc = p.img + (long long)c * DG_A_BYTES;
    const float* xb = p.dg + ((long long)b * (2 * n) + 16 * c) * L;
    const int c_count = min(16, 2 * n - 16 * c);
    const int w4 = (t0 - dil) & ~3;                      // 16-byte aligned start of the window (first column any tap needs)
    if (DG_EXP & 64) return;
    for (int idx = wave_s; idx < NI; idx += 8) {         // wave-uniform trip count
      if (idx < 2 * DG_A_BLOCKS) {
        if (DG_EXP & 128) continue;
        __builtin_amdgcn_global_load_lds(WN_GLOBAL_PTR((DG_EXP & 4) ? zero16 : asrc + idx * 1024 + lane * 16), WN_LDS_VOID(sl + idx * 1024), 16, 0, 0);
      } else {
        const int bi = idx - 2 * DG_A_BLOCKS;
        const int gq = bi >= p.nblkw ? 1 : 0, m = bi - gq * p.nblkw;
        const int row = 8 * gq + (lane >> 3);
        const int t = w4 + 32 * m + 4 * (lane & 7);
        const bool ok = row < c_count && t >= 0 && t < L && !(DG_EXP & 2);
        const char* src = ok ? reinterpret_cast<const char*>(xb + ((long long)row * L + t)) : zero16;
        __builtin_amdgcn_global_load_lds(WN_GLOBAL_PTR(src), WN_LDS_VOID(sl + DG_A_BYTES + gq * p.gsw + m * 1024), 16, 0, 0);
      }
    }
  };

  // The epilogue's transpose tiles and row-sum array live in ring slot 0; a tile whose predecessor's epilogue is still
  // running starts its ring at slot 1, so its first stages stream in underneath that epilogue.
  int first_slot = 0;
  for (int v = blockIdx.x; v < p.n_wg; v += gridDim.x) {
    int b, t0;
    const int wg = tile_of(v, b, t0);
    const int sub = (t0 - dil) & 3;
    // accumulators start as the tensors the products are added to: d_a rows of block i (if a residual cotangent comes
    // in), then the d_u0 rows (accumulated in place)
    f32x16 acc[5][DG_NCB];
#pragma unroll
    for (int mb = 0; mb < 5; ++mb)
#pragma unroll
      for (int cb = 0; cb < DG_NCB; ++cb) {
        const float* src = mb < 4 ? (p.d_a ? p.d_a + ((long long)b * n + mb * 32) * L : nullptr) : p.d_u0 + (long long)b * p.d_u0_bs;
        const int rows = (DG_EXP & 32) ? 0 : (mb < 4 ? (p.d_a ? n - mb * 32 : 0) : p.h);
        wn_acc_load(acc[mb][cb], src, rows, L, t0 + wave_n0 + 32 * cb, lane);
      }
    // first tile: the ring is primed AFTER the operand loads, which are then older than every LDS-DMA piece and covered by
    // the counted waits as they stand (a later tile's ring was primed under its predecessor's epilogue: its first wait
    // also waits for these loads — vmcnt retires in order — which is merely conservative)
    if (first_slot == 0)
      for (int d = 0; d < depth && d < CHK; ++d) issue(b, t0, d, d);

    int slot = first_slot;
    for (int c = 0; c < CHK; ++c) {
      // stages still allowed in flight behind stage c: those already issued, i.e. min(depth - 1, CHK - 1 - c).  (After a
      // predecessor tile the epilogue's own stores may still be in flight and count too: they are NEWER than the stage
      // waited for, so the count is merely conservative — vmcnt retires in issue order.)
      const int newer = min(depth - 1, CHK - 1 - c);
      wn_wait_sw<16>(newer * my_pieces);
      __builtin_amdgcn_s_barrier();
      if (c + depth < CHK) issue(b, t0, c + depth, (slot + depth) % p.ns);
      const char* base = ldsb + slot * p.slot;
#pragma unroll
      for (int tap = 0; tap < 3; ++tap) {
        // this wave's DG_NCB column blocks of the tap's window: the fragment of a weight block is read from LDS once and
        // multiplied against all of them (LDS reads, 128 B/clk/CU, bound the one-column-block form of this loop)
        wn_bf16x8 bh[DG_NCB], bl[DG_NCB];
#pragma unroll
        for (int cb = 0; cb < DG_NCB; ++cb) {
          const int colx = wave_n0 + 32 * cb + l31 + (2 - tap) * dil + sub;
          const char* bp = base + DG_A_BYTES + half * p.gsw + (colx >> 5) * 1024 + (colx & 31) * 4;
          float v8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v8[j] = *reinterpret_cast<const float*>(bp + j * 128);
          wn_u32x4 bh4, bl4;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            unsigned hh, ll;
            wn_split_pair(v8[2 * j], v8[2 * j + 1], hh, ll);
            bh4[j] = hh; bl4[j] = ll;
          }
          bh[cb] = __builtin_bit_cast(wn_bf16x8, bh4); bl[cb] = __builtin_bit_cast(wn_bf16x8, bl4);
        }
        const char* ab = base + (tap == 0 ? 0 : (tap == 1 ? 4 : 9)) * 2048;
#pragma unroll
        for (int mb = 0; mb < 5; ++mb) {
          if (mb == 4 && tap != 1) continue;               // the d_u0 block exists on the centre tap only
          const wn_bf16x8 ah = *reinterpret_cast<const wn_bf16x8*>(ab + mb * 2048 + lane * 16);
          const wn_bf16x8 al = *reinterpret_cast<const wn_bf16x8*>(ab + mb * 2048 + 1024 + lane * 16);
#pragma unroll
          for (int cb = 0; cb < DG_NCB; ++cb) {
            if (DG_EXP & 1) { asm volatile("" ::"v"(al), "v"(ah), "v"(bh[cb]), "v"(bl[cb])); continue; }
            acc[mb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[cb], acc[mb][cb], 0, 0, 0);
            acc[mb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[cb], acc[mb][cb], 0, 0, 0);
            acc[mb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[cb], acc[mb][cb], 0, 0, 0);
          }
        }
      }
      slot = slot + 1 == p.ns ? 0 : slot + 1;
    }
    __syncthreads();                                     // every wave is past its last fragment read: the ring is free
    // the next tile's first stages go into slots 1.. while this tile's epilogue works in slot 0
    if (v + (int)gridDim.x < p.n_wg) {
      int bn, tn;
      tile_of(v + gridDim.x, bn, tn);
      for (int d = 0; d < depth && d < CHK; ++d) issue(bn, tn, d, 1 + d);
      first_slot = 1;
    }
    float* const tile = reinterpret_cast<float*>(ldsb + wave_s * WN_TILE_BYTES);
    float* const rsum = reinterpret_cast<float*>(ldsb + 8 * WN_TILE_BYTES);      // [8 waves][128]
    float* const rsum_w = rsum + wave_s * 128;
    if (p.row_sums) {
      rsum[tid] = 0.f; rsum[512 + tid] = 0.f;
      __syncthreads();
    }
    // 5 × DG_NCB tiles straight from the accumulators; the bias-gradient row sums (when asked for) go through the
    // wave's transpose tile
#pragma unroll
    for (int k = 0; k < 5 * DG_NCB; ++k) {
      const int i = k / DG_NCB, cb = k % DG_NCB, tcol = t0 + wave_n0 + 32 * cb;
      float* dst = i < 4 ? p.d_a_new + ((long long)b * n + i * 32) * L : p.d_u0 + (long long)b * p.d_u0_bs;
      const int rows = i < 4 ? n - i * 32 : p.h;
      if (DG_EXP & 16) { if (acc[i][cb][0] == 12345.678f) dst[0] = 1.f; continue; }
      wn_acc_store(acc[i][cb], dst, (DG_EXP & 8) ? 0 : rows, L, tcol, lane);
      if (p.row_sums && i < 4) {
#pragma unroll
        for (int r = 0; r < 16; ++r) tile[((r & 3) + 8 * (r >> 2) + 4 * half) * 36 + l31] = acc[i][cb][r];
        wn_tile_row_sums(tile, rsum_w, i * 32, rows, L, tcol, lane);
      }
    }
    if (p.row_sums) {
      __syncthreads();
      if (tid < 128) {                                                        // [128][n_wg]; waves added in a fixed order
        float s8 = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) s8 += rsum[w * 128 + tid];
        p.row_sums[(long long)tid * p.n_wg + wg] = s8;
      }
    }
  }
}

// window geometry of one fst_wn_layer_dgrad stage at dilation dil: (32-sample blocks per window, bytes per window, bytes per ring slot)
static inline void wn_dgrad_geometry(int dil, int* nblkw, int* gsw, int* slot) {
  *nblkw = (DG_TN + 2 * dil + 3 + 31) / 32;
  *gsw = *nblkw * 1024 + 128;
  *slot = DG_A_BYTES + 2 * *gsw;
}

// 1 when fst_wn_layer_dgrad serves (n, h, dil): two ring slots fit the 160 KiB of LDS and the counted-wait table covers a stage
// (with 512-sample tiles: up to dilation 128, i.e. WN stacks of up to 8 layers); the host side falls back to the generic
// data-gradient launch otherwise.
extern "C" int fst_wn_dgrad_fits(int n, int h, int dil) {
  if (!(n > 0 && n <= 128 && h > 0 && h <= 32 && dil > 0)) return 0;
  int nblkw, gsw, slot;
  wn_dgrad_geometry(dil, &nblkw, &gsw, &slot);
  if (2 * slot > 160 * 1024) return 0;
  const int ns = 3 * slot <= 160 * 1024 ? 3 : 2;
  const int NI = DG_A_BLOCKS * 2 + 2 * nblkw;
  return ((NI + 7) / 8) * (ns - 2) <= 16 ? 1 : 0;
}

extern "C" int fst_wn_layer_dgrad(const float* dg, const void* image, int64_t image_bytes, const float* d_a, float* d_a_new,
                                  float* d_u0, float* row_sums, int64_t row_sums_rows, int B, int L, int n, int h, int dil,
                                  int64_t numel_a, int64_t d_u0_bs, void* stream) {
  FST_REQUIRE(dg && image && d_a_new && d_u0, "fst_wn_layer_dgrad: null operand");
  FST_REQUIRE(B > 0 && L > 0 && n > 0 && n <= 128 && h > 0 && h <= 32 && dil > 0, "fst_wn_layer_dgrad: B=%d L=%d n=%d h=%d dil=%d", B, L,
              n, h, dil);
  FST_REQUIRE((long long)B * n * L == (long long)numel_a,
              "fst_wn_layer_dgrad: B*n*L does not match the element count %lld of the [B][n][L] tensors", (long long)numel_a);
  FST_REQUIRE(d_u0_bs >= (int64_t)h * L && d_u0_bs % 4 == 0,
              "fst_wn_layer_dgrad: d_u0 batch stride %lld (needs >= h*L = %lld and a multiple of 4)", (long long)d_u0_bs, (long long)h * L);
  FST_REQUIRE(image_bytes == fst_wn_dgrad_image_bytes(n), "fst_wn_layer_dgrad: image is %lld bytes, expected %lld",
              (long long)image_bytes, (long long)fst_wn_dgrad_image_bytes(n));
  auto al16 = [](const void* q) { return q == nullptr || (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  FST_REQUIRE(L % 4 == 0 && al16(dg) && al16(image) && al16(d_a) && al16(d_a_new) && al16(d_u0),
              "fst_wn_layer_dgrad: needs L %% 4 == 0 and 16-byte aligned tensors (L=%d)", L);
  WnDgradParams p;
  p.dg = dg; p.img = static_cast<const char*>(image); p.d_a = d_a; p.d_a_new = d_a_new; p.d_u0 = d_u0; p.d_u0_bs = d_u0_bs;
  p.row_sums = row_sums;
  p.B = B; p.L = L; p.n = n; p.h = h; p.dil = dil; p.CHK = (2 * n + 15) / 16;
  p.tiles_per_seq = (L + DG_TN - 1) / DG_TN;
  p.n_wg = B * p.tiles_per_seq;
  FST_REQUIRE(row_sums == nullptr || row_sums_rows == p.n_wg, "fst_wn_layer_dgrad: row_sums has %lld rows, the launch has %d "
              "workgroups (B x ceil(L/512))", (long long)row_sums_rows, p.n_wg);
  wn_dgrad_geometry(dil, &p.nblkw, &p.gsw, &p.slot);
  p.ns = 3 * p.slot <= 160 * 1024 ? 3 : 2;
  FST_REQUIRE(2 * p.slot <= 160 * 1024, "fst_wn_layer_dgrad: dilation %d needs a %d-byte window slot: too large for LDS", dil, p.slot);
  const int NI = DG_A_BLOCKS * 2 + 2 * p.nblkw;
  FST_REQUIRE(((NI + 7) / 8) * (p.ns - 2) <= 16, "fst_wn_layer_dgrad: %d pieces per stage exceed the counted-wait table", NI);
  size_t lds_bytes = (size_t)p.ns * p.slot;
  if (lds_bytes < 8 * WN_TILE_BYTES + 4096) lds_bytes = 8 * WN_TILE_BYTES + 4096;   // tiles + the per-wave row-sum arrays
  if (int rc = fst_allow_full_lds((const void*)wn_layer_dgrad_kernel, "fst_wn_layer_dgrad")) return rc;
  // one persistent workgroup per CU; with a ring of one slot only (never: ns >= 2) the next tile could not start early
  int grid = p.n_wg;
  const int cus = fst_cu_count();
  if (cus > 0 && grid > cus && (size_t)2 * lds_bytes > 160 * 1024) grid = cus;
  hipLaunchKernelGGL(wn_layer_dgrad_kernel, dim3((unsigned)grid), dim3(512), lds_bytes, (hipStream_t)stream, p);
  FST_LAUNCH_CHECK();
  return 0;
}

// Row sums of an accumulator tile WITHOUT an LDS transpose (the persistent kernel's LDS holds a layer's weights): a lane holds
// 16 rows x 1 column; a halving butterfly over the 32 lanes of a half — at distance 16 a lane keeps registers 0-7 (bit 4 clear)
// or 8-15 and adds the partner's copy of the same registers, at distance 8 four of those eight, ... — leaves every lane with ONE
// row's total after 8 + 4 + 2 + 1 exchanges, one more at distance 1 completes it: 16 exchanges per tile instead of 80.
// Lane l31 (even) of half hh ends with row (j&3) + 8(j>>2) + 4hh, j = 8·bit4 + 4·bit3 + 2·bit2 + bit1 of l31; it adds the total
// to ITS entry of the wave's row array (plain read-modify-write in program order: the same sum in every run).
__device__ __forceinline__ void ws_row_sums(const float (&v)[16], float* rows_w, int row0, int rows_valid, bool col_ok, int lane) {
  const int l31 = lane & 31, half = lane >> 5;
  float a8[8], a4[4], a2[2];
  const bool b4 = (l31 & 16) != 0, b3 = (l31 & 8) != 0, b2 = (l31 & 4) != 0, b1 = (l31 & 2) != 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float lo = col_ok ? v[j] : 0.f, hi = col_ok ? v[8 + j] : 0.f;
    a8[j] = (b4 ? hi : lo) + __shfl_xor(b4 ? lo : hi, 16, 64);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) a4[j] = (b3 ? a8[4 + j] : a8[j]) + __shfl_xor(b3 ? a8[j] : a8[4 + j], 8, 64);
#pragma unroll
  for (int j = 0; j < 2; ++j) a2[j] = (b2 ? a4[2 + j] : a4[j]) + __shfl_xor(b2 ? a4[j] : a4[2 + j], 4, 64);
  float a1 = (b1 ? a2[1] : a2[0]) + __shfl_xor(b1 ? a2[0] : a2[1], 2, 64);
  a1 += __shfl_xor(a1, 1, 64);
  const int j = (b4 ? 8 : 0) + (b3 ? 4 : 0) + (b2 ? 2 : 0) + (b1 ? 1 : 0);
  const int row = (j & 3) + 8 * (j >> 2) + 4 * half;
  if ((l31 & 1) == 0 && row < rows_valid) rows_w[row0 + row] += a1;
}

// ------------------------------------------------------------------------------------------------
// The whole backward of a WN stack in ONE persistent launch (sequences of up to 512 samples)
//
// fst_wn_layer_bwd + fst_wn_layer_dgrad, per layer, are two launches whose every workgroup walks the same phases at the same
// time — operand loads, the GEMM, stores — and d_a / dg / d_u0 make a round trip through the memory fabric at every launch
// boundary: 660 MB per layer at B=256, L=512 (counters: profiles/r04_stack_*), which is what bounds them.  With L <= 512 a
// 512-sample tile IS a whole sequence: the dilated taps never look outside the tile, so the chain
//     dacts = W_rsᵀ·[d_a ; d_out]  →  dg = gate'(t, s)·dacts  →  d_a += W_inᵀ (*) dg,  d_u0 += W_condᵀ·dg        (layer nl−1 … 0)
// of one batch element depends on nothing another workgroup produces.  One workgroup (8 waves × 64 samples) therefore walks
// all layers of its batch element and d_a NEVER LEAVES ITS ACCUMULATORS: phase B (the data gradient) accumulates into them,
// phase A (res_skip / gate backward) multiplies straight out of them — an accumulator tile is, register for register, the B
// operand of an MFMA that sums over its rows; the weight image holds the d_a stages in that k-order (fst_wn_pack_bwd,
// acc_order) — and d_a is stored only where something outside reads it (layer 0: the start conv; every layer in the full
// pass: the res_skip weight gradients).  That takes the three d_a passes (189 of the 660 MB) off the fabric, the launch
// boundaries (2 per layer) are gone, and — nothing synchronising the workgroups with each other — the CUs drift apart, so one
// CU's load / store phases run under other CUs' GEMM phases.  In GradNorm's partial passes (no weight gradients) dg lives in a
// scratch tensor that is rewritten layer after layer.
//
// Phase A runs as TWO column passes (the wave's two 32-sample column blocks one after the other): d_a occupies 128 registers,
// so the dacts accumulators get 64 (4 row blocks × 1 column block).  Per pass: the d_out stages (operand through the LDS-DMA
// ring; they depend on nothing the previous phase stored, so the first two are issued under its last stores), then the d_a
// stages (weights through the ring, the operand from the accumulators), then the gate on four tiles.  Phase B = the body of
// wn_layer_dgrad_kernel.  Global stores of one phase (dg, d_u0) are read by the next through LDS-DMA / loads of the SAME
// workgroup: s_waitcnt vmcnt(0) by every wave + a workgroup barrier orders them (one CU, one L1).
// ------------------------------------------------------------------------------------------------
#ifndef WS_NT
#define WS_NT 0     // cache policy of the streamed tensors (t,s, dg, d_out): 0 default, 2 non-temporal
#endif
#define WS_TN 512
#define WS_NB 8                                    // 32-sample column blocks of a phase-A window row group: one per wave and column pass
#define WS_GS (WS_NB * 1024 + 128)
#define WS_SLOT_A (WN_BW_A + 2 * WS_GS)
#define WS_LDS_A (16 * WN_BW_A)                     // a layer's resident W_rsᵀ image: up to 16 stages (n <= 128) of 8 KiB

struct WnStackParams {
  const float* ts[WS_MAXL];      // saved gate halves [B][2n][L] per layer
  const char* img_b[WS_MAXL];    // fst_wn_pack_bwd images (acc_order = 1)
  const char* img_d[WS_MAXL];    // fst_wn_pack_dgrad images
  float* dg[WS_MAXL];            // [B][2n][L] per layer (partial passes: one scratch tensor for all)
  float* da_out[WS_MAXL];        // cotangent of the layer's input [B][n][L]: written when not null (layer 0: always)
  float* rs_b[WS_MAXL];          // optional [256][B]: per-sequence Σ_t dg[row]
  float* rs_d[WS_MAXL];          // optional [128][B]: per-sequence Σ_t da_out[row]
  const float* d_out;            // [B][n][L]
  float* d_u0;                   // [B][h][L] with batch stride d_u0_bs, accumulated in place
  long long d_u0_bs;
  int dil[WS_MAXL], nblkw[WS_MAXL], gsw[WS_MAXL], slot_d[WS_MAXL];
  int nl, B, L, n, h, CH, CHK;
};

__global__ __launch_bounds__(512, 2) void wn_stack_bwd_kernel(WnStackParams p_by_value) {
  // The per-layer tables are indexed with a run-time layer number: read through the kernarg segment they are scalar loads at a
  // computed offset; indexing the by-value struct would make the compiler copy it to scratch (1 KiB per lane) or keep all of
  // it in SGPRs (639 spilled).
  const auto& p = *(const __attribute__((address_space(4))) WnStackParams*)__builtin_amdgcn_kernarg_segment_ptr();
  (void)p_by_value;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* const ldsb = reinterpret_cast<char*>(lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int half = lane >> 5, l31 = lane & 31;
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_n0 = wave_s * 64;
  const int L = p.L, n = p.n, CH = p.CH, CHK = p.CHK;
  // phase A's LDS: [the layer's whole W_rsᵀ image, resident for both column passes: up to 16 x 8 KiB][the row-sum arrays] — the
  // next layer's image streams in while a phase-B epilogue is still adding row sums
  float* const rsum = reinterpret_cast<float*>(ldsb + WS_LDS_A);        // [8 waves][256]
  float* const rsum_w = rsum + wave_s * 256;
  const unsigned vlane = (unsigned)(4 * half * L + l31) * 4u;
  const WsLane wl0 = ws_lane(vlane, wave_n0, L, lane), wl1 = ws_lane(vlane, wave_n0 + 32, L, lane);

  // layer i's W_rsᵀ image ([d_a stages][d_out stages]; the top layer: d_out stages only): one 1-KiB piece per wave and stage
  auto issue_res = [&](int i) {
    const int S3 = (i == p.nl - 1 ? 1 : 2) * CH;
    const char* const img = p.img_b[i];
    for (int c = 0; c < S3; ++c)
      __builtin_amdgcn_global_load_lds(WN_GLOBAL_PTR(img + (long long)c * WN_BW_A + wave_s * 1024 + lane * 16),
                                       WN_LDS_VOID(ldsb + c * WN_BW_A + wave_s * 1024), 16, 0, 0);
  };
  // per-lane byte offset of a B-fragment element of a [channel][time] matrix read straight from memory: lane half hh reads
  // channels 8hh + j of a 16-channel chunk at column l31 of its block
  const unsigned vfrag = (unsigned)(8 * half * L + l31) * 4u;

  WN_SUMS;
  WN_T(tw0);
  bool primed = false;                                     // the weight image of the layer at hand is already in flight
  for (int b = blockIdx.x; b < p.B; b += gridDim.x) {
    // acc[0..3] = the cotangent of the residual stream (d_a), carried from layer to layer; acc[4] = the conditioning rows (d_u0)
    f32x16 acc[5][2];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][cb][r] = 0.f;     // the top layer has no residual output
    for (int i = p.nl - 1; i >= 0; --i) {
      const bool last = i == p.nl - 1;
      // ============================================================ phase A: dg = gate'(t, s) · W_rsᵀ·[d_a ; d_out]
      {
        // the tanh | sigmoid halves are two [n][L] matrices
        const __amdgpu_buffer_rsrc_t ts_t = ws_rsrc(p.ts[i] + (long long)b * (2 * n) * L);
        const __amdgpu_buffer_rsrc_t ts_s = ws_rsrc(p.ts[i] + ((long long)b * 2 + 1) * n * L);
        const __amdgpu_buffer_rsrc_t dg_t = ws_rsrc(p.dg[i] + (long long)b * (2 * n) * L);
        const __amdgpu_buffer_rsrc_t dg_s = ws_rsrc(p.dg[i] + ((long long)b * 2 + 1) * n * L);
        const __amdgpu_buffer_rsrc_t dout_r = ws_rsrc(p.d_out + (long long)b * n * L);
        float* const rs_out = p.rs_b[i];
        WN_T(ta0);
        if (!primed) issue_res(i);
        primed = false;
        if (rs_out) {
#pragma unroll
          for (int w = 0; w < 4; ++w) rsum[w * 512 + tid] = 0.f;
        }
        wn_wait_vmcnt<0>();                                  // this wave's pieces of the image have landed ...
        __syncthreads();                                     // ... and everyone's; the row-sum arrays are zeroed
        WN_T(ta1);
        WN_ACC(0, ta0, ta1);                             // phase A: waiting for the weight image
        const char* const w_dout = ldsb + (long long)(last ? 0 : CH) * WN_BW_A;
        auto run_pass = [&](auto pc) {
          constexpr int cb = decltype(pc)::value;
          const WsLane wl = cb ? wl1 : wl0;
          const int tcol = wave_n0 + 32 * cb;
          f32x16 da[4];                                    // dacts rows of this column block
#pragma unroll
          for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) da[mb][r] = 0.f;
          auto multiply = [&](const char* base, const wn_bf16x8 bh, const wn_bf16x8 bl) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
              const wn_bf16x8 ah = *reinterpret_cast<const wn_bf16x8*>(base + mb * 2048 + lane * 16);
              const wn_bf16x8 al = *reinterpret_cast<const wn_bf16x8*>(base + mb * 2048 + 1024 + lane * 16);
              da[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, da[mb], 0, 0, 0);
              da[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, da[mb], 0, 0, 0);
              da[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, da[mb], 0, 0, 0);
            }
          };
          WN_T(tp0);
          // ---- the d_out stages: the operand comes straight from memory in fragment layout (eight dword loads per stage: lane
          // half hh reads channels 8hh + j at its column), the next stage's in flight while one is multiplied — no ring, no barrier
          const unsigned vf = tcol + l31 < L ? vfrag : WS_OOB, vf_lo = half == 0 ? vf : WS_OOB;
          auto load_dout = [&](float (&v)[8], int c) {
            const int Lq = ws_opaque(L), nq = ws_opaque(n);
            const int sbase = (16 * c * Lq + tcol) * 4;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              float x = 0.f;
              if (16 * c + j < nq)
                x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dout_r, 16 * c + 8 + j < nq ? vf : vf_lo, sbase + j * Lq * 4, WS_NT));
              v[j] = x;
            }
          };
          float vn[8];
          load_dout(vn, 0);
          for (int k = 0; k < CH; ++k) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = vn[j];
            if (k + 1 < CH) load_dout(vn, k + 1);
            wn_u32x4 bh4, bl4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              unsigned hh, ll;
              wn_split_pair(v[2 * j], v[2 * j + 1], hh, ll);
              bh4[j] = hh; bl4[j] = ll;
            }
            multiply(w_dout + k * WN_BW_A, __builtin_bit_cast(wn_bf16x8, bh4), __builtin_bit_cast(wn_bf16x8, bl4));
          }
          // ---- the d_a stages: the operand IS the accumulator tile (registers 8s..8s+7 of row block c>>1 are k-step c&1)
          if (!last) {
            auto da_stage = [&](auto cc) {
              constexpr int c = decltype(cc)::value;
              if (c < CH) {                                // wave-uniform
                wn_u32x4 bh4, bl4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  unsigned hh, ll;
                  wn_split_pair(acc[c >> 1][cb][8 * (c & 1) + 2 * j], acc[c >> 1][cb][8 * (c & 1) + 2 * j + 1], hh, ll);
                  bh4[j] = hh; bl4[j] = ll;
                }
                multiply(ldsb + c * WN_BW_A, __builtin_bit_cast(wn_bf16x8, bh4), __builtin_bit_cast(wn_bf16x8, bl4));
                __builtin_amdgcn_sched_barrier(0);        // stage by stage: hoisted fragment reads of later stages cost registers
              }
            };
            da_stage(std::integral_constant<int, 0>{}); da_stage(std::integral_constant<int, 1>{});
            da_stage(std::integral_constant<int, 2>{}); da_stage(std::integral_constant<int, 3>{});
            da_stage(std::integral_constant<int, 4>{}); da_stage(std::integral_constant<int, 5>{});
            da_stage(std::integral_constant<int, 6>{}); da_stage(std::integral_constant<int, 7>{});
          }
          WN_T(tp1);
          WN_ACC(1, tp0, tp1);                           // phase A: GEMM stages of a pass
          // ---- gate: four tiles (row blocks) of this column block
#pragma unroll
          for (int blk = 0; blk < 4; ++blk) {
            const int rows_valid = n - blk * 32;
            f32x16 tv, sv;
            ws_acc_load<WS_NT>(tv, ts_t, wl, blk * 32, tcol, rows_valid, L);
            ws_acc_load<WS_NT>(sv, ts_s, wl, blk * 32, tcol, rows_valid, L);
            float gt[16], gs[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float d = da[blk][r], t = tv[r], s2 = sv[r];
              gt[r] = d * s2 * (1.f - t * t);
              gs[r] = d * t * s2 * (1.f - s2);
            }
            ws_acc_store<WS_NT>(gt, dg_t, wl, blk * 32, tcol, rows_valid, L);
            ws_acc_store<WS_NT>(gs, dg_s, wl, blk * 32, tcol, rows_valid, L);
            if (rs_out) {
              ws_row_sums(gt, rsum_w, blk * 32, rows_valid, tcol + l31 < L, lane);
              ws_row_sums(gs, rsum_w, n + blk * 32, rows_valid, tcol + l31 < L, lane);
            }
            __builtin_amdgcn_sched_barrier(0);            // tile by tile (a scheduler that hoists the loads of later tiles runs out of registers)
          }
          WN_T(tp2);
          WN_ACC(2, tp1, tp2);                           // phase A: gate epilogue of a pass
        };
        run_pass(std::integral_constant<int, 0>{});
        run_pass(std::integral_constant<int, 1>{});
        if (rs_out) {
          __syncthreads();
          if (tid < 256) {
            float s8 = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) s8 += rsum[w * 256 + tid];
            rs_out[(long long)tid * p.B + b] = s8;            // [256][B]
          }
        }
        WN_T(ta3);
        wn_wait_vmcnt<0>();                                  // this wave's dg stores have reached L2 ...
        __syncthreads();                                     // ... and so have everyone's: phase B may fetch them; the ring is free
        WN_T(ta4);
        WN_ACC(3, ta3, ta4);                             // phase A -> B: store drain + barrier
      }
      // ============================================================ phase B: d_a += W_inᵀ (*) dg,  d_u0 += W_condᵀ·dg
      {
        const int dil = p.dil[i], nblkw = p.nblkw[i], gsw = p.gsw[i], slotb = p.slot_d[i];
        const char* const img = p.img_d[i];
        const char* const zero16 = img + (long long)CHK * DG_A_BYTES;
        const float* const dgr = p.dg[i] + (long long)b * (2 * n) * L;
        float* const da_store = p.da_out[i];
        const __amdgpu_buffer_rsrc_t dan_r = ws_rsrc(da_store ? da_store + (long long)b * n * L : nullptr);
        const __amdgpu_buffer_rsrc_t du_r = ws_rsrc(p.d_u0 + (long long)b * p.d_u0_bs);
        float* const rs_out = p.rs_d[i];
        const int NI = DG_A_BLOCKS * 2 + 2 * nblkw;
        const int sub = (0 - dil) & 3;
        const int w4 = (0 - dil) & ~3;
        auto issue = [&](int c, int slot) {
          char* const sl = ldsb + slot * slotb;
          const char* asrc = img + (long long)c * DG_A_BYTES;
          const float* xb = dgr + (long long)(16 * c) * L;
          const int c_count = min(16, 2 * n - 16 * c);
          for (int idx = wave_s; idx < NI; idx += 8) {
            if (idx < 2 * DG_A_BLOCKS) {
              __builtin_amdgcn_global_load_lds(WN_GLOBAL_PTR(asrc + idx * 1024 + lane * 16), WN_LDS_VOID(sl + idx * 1024), 16, 0, 0);
            } else {
              const int bi = idx - 2 * DG_A_BLOCKS;
              const int gq = bi >= nblkw ? 1 : 0, m = bi - gq * nblkw;
              const int row = 8 * gq + (lane >> 3);
              const int t = w4 + 32 * m + 4 * (lane & 7);
              const bool ok = row < c_count && t >= 0 && t < L;
              const char* src = ok ? reinterpret_cast<const char*>(xb + ((long long)row * L + t)) : zero16;
              __builtin_amdgcn_global_load_lds(WN_GLOBAL_PTR(src), WN_LDS_VOID(sl + DG_A_BYTES + gq * gsw + m * 1024), 16, 0, WS_NT);
            }
          }
        };
        WN_T(tb0);
        // the d_a tiles are where the previous layer left them; the conditioning rows come from memory
        ws_acc_load(acc[4][0], du_r, wl0, 0, wave_n0, p.h, L);
        ws_acc_load(acc[4][1], du_r, wl1, 0, wave_n0 + 32, p.h, L);
        asm volatile("" ::: "memory");
        issue(0, 0);                                         // two ring slots: one stage in flight while one is multiplied
        int slot = 0;
        WN_T(tb1);
        WN_ACC(4, tb0, tb1);                             // phase B: conditioning-row loads + first stage (issue)
        for (int c = 0; c < CHK; ++c) {
          wn_wait_vmcnt<0>();
          __builtin_amdgcn_s_barrier();
          if (c + 1 < CHK) issue(c + 1, slot ^ 1);
          const char* base = ldsb + slot * slotb;
#pragma unroll
          for (int tap = 0; tap < 3; ++tap) {
            wn_bf16x8 bh[DG_NCB], bl[DG_NCB];
#pragma unroll
            for (int cb = 0; cb < DG_NCB; ++cb) {
              const int colx = wave_n0 + 32 * cb + l31 + (2 - tap) * dil + sub;
              const char* bp = base + DG_A_BYTES + half * gsw + (colx >> 5) * 1024 + (colx & 31) * 4;
              float v8[8];
#pragma unroll
              for (int j = 0; j < 8; ++j) v8[j] = *reinterpret_cast<const float*>(bp + j * 128);
              wn_u32x4 bh4, bl4;
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                unsigned hh, ll;
                wn_split_pair(v8[2 * j], v8[2 * j + 1], hh, ll);
                bh4[j] = hh; bl4[j] = ll;
              }
              bh[cb] = __builtin_bit_cast(wn_bf16x8, bh4); bl[cb] = __builtin_bit_cast(wn_bf16x8, bl4);
            }
            const char* ab = base + (tap == 0 ? 0 : (tap == 1 ? 4 : 9)) * 2048;
#pragma unroll
            for (int mb = 0; mb < 5; ++mb) {
              if (mb == 4 && tap != 1) continue;
              const wn_bf16x8 ah = *reinterpret_cast<const wn_bf16x8*>(ab + mb * 2048 + lane * 16);
              const wn_bf16x8 al = *reinterpret_cast<const wn_bf16x8*>(ab + mb * 2048 + 1024 + lane * 16);
#pragma unroll
              for (int cb = 0; cb < DG_NCB; ++cb) {
                acc[mb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[cb], acc[mb][cb], 0, 0, 0);
                acc[mb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[cb], acc[mb][cb], 0, 0, 0);
                acc[mb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[cb], acc[mb][cb], 0, 0, 0);
              }
            }
          }
          slot ^= 1;
        }
        WN_T(tb2);
        WN_ACC(5, tb1, tb2);                             // phase B: GEMM loop
        __syncthreads();                                     // every wave is past its last fragment read: the ring is free
        // the weight image of the next phase A (the layer below, or the top layer of this workgroup's next sequence) depends on
        // nothing this phase stores: it streams into LDS now, under the stores below
        {
          const int ni = i > 0 ? i - 1 : p.nl - 1, nb = i > 0 ? b : b + (int)gridDim.x;
          if (nb < p.B) {
            issue_res(ni);
            primed = true;
          }
        }
        if (rs_out) {
          rsum[tid] = 0.f; rsum[512 + tid] = 0.f;
          __syncthreads();
        }
#pragma unroll
        for (int k = 0; k < 5 * DG_NCB; ++k) {
          const int ib = k / DG_NCB, cb = k % DG_NCB, tcol = wave_n0 + 32 * cb;
          const int rows = ib < 4 ? n - ib * 32 : p.h;
          if (ib < 4) {
            if (da_store) ws_acc_store(acc[ib][cb], dan_r, cb ? wl1 : wl0, ib * 32, tcol, rows, L);
          } else {
            ws_acc_store(acc[ib][cb], du_r, cb ? wl1 : wl0, 0, tcol, rows, L);
          }
          if (rs_out && ib < 4) {
            float av[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) av[r] = acc[ib][cb][r];
            ws_row_sums(av, rsum + wave_s * 128, ib * 32, rows, tcol + l31 < L, lane);
          }
        }
        if (rs_out) {
          __syncthreads();
          if (tid < 128) {
            float s8 = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) s8 += rsum[w * 128 + tid];
            rs_out[(long long)tid * p.B + b] = s8;            // [128][B]
          }
        }
        WN_T(tb3);
        WN_ACC(6, tb2, tb3);                             // phase B: epilogue stores (issue)
        // d_u0 of this layer must be in L2 before the next layer's phase B reads it back (the wait also retires the primed stages)
        wn_wait_vmcnt<0>();
        __syncthreads();
        WN_T(tb4);
        WN_ACC(7, tb3, tb4);                             // phase B -> A: store drain + barrier
      }
    }
  }
  WN_T(tw1);
  WN_ACC(9, tw0, tw1);
  WN_FLUSH;
}

// 1 when fst_wn_stack_bwd serves a WN stack of nl layers on sequences of L samples (dilations 1, 2, 4, ... as the reference's WN)
extern "C" int fst_wn_stack_bwd_ok(int n, int h, int L, int nl) {
  if (!(n > 0 && n <= 128 && h > 0 && h <= 32 && L > 0 && L <= WS_TN && L % 4 == 0 && nl > 0 && nl <= WS_MAXL)) return 0;
  for (int i = 0; i < nl; ++i)
    if (i >= 20 || !fst_wn_dgrad_fits(n, h, 1 << i)) return 0;
  return 1;
}

extern "C" int fst_wn_stack_bwd(const float* const* ts, const void* const* img_b, const void* const* img_d, float* const* dg,
                                float* const* da_out, float* const* rs_b, float* const* rs_d, const float* d_out, float* d_u0,
                                int64_t d_u0_bs, int nl, int B, int L, int n, int h, int64_t numel_a, void* stream) {
  FST_REQUIRE(ts && img_b && img_d && dg && da_out && d_out && d_u0, "fst_wn_stack_bwd: null table");
  FST_REQUIRE(fst_wn_stack_bwd_ok(n, h, L, nl), "fst_wn_stack_bwd: not served: n=%d h=%d L=%d nl=%d (needs n <= 128, h <= 32, "
              "L <= 512, L %% 4 == 0, nl <= %d)", n, h, L, nl, WS_MAXL);
  FST_REQUIRE(B > 0 && (long long)B * n * L == (long long)numel_a,
              "fst_wn_stack_bwd: B*n*L does not match the element count %lld of the [B][n][L] tensors", (long long)numel_a);
  FST_REQUIRE(d_u0_bs >= (int64_t)h * L && d_u0_bs % 4 == 0,
              "fst_wn_stack_bwd: d_u0 batch stride %lld (needs >= h*L = %lld and a multiple of 4)", (long long)d_u0_bs, (long long)h * L);
  FST_REQUIRE((rs_b == nullptr) == (rs_d == nullptr), "fst_wn_stack_bwd: row sums of both kinds or of neither");
  auto al16 = [](const void* q) { return q == nullptr || (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  WnStackParams p;
  size_t lds_bytes = WS_LDS_A + 8192;                                // phase A: the weight image and the row-sum arrays
  for (int i = 0; i < nl; ++i) {
    FST_REQUIRE(ts[i] && img_b[i] && img_d[i] && dg[i] && (i > 0 || da_out[i]), "fst_wn_stack_bwd: null operand of layer %d", i);
    FST_REQUIRE(al16(ts[i]) && al16(img_b[i]) && al16(img_d[i]) && al16(dg[i]) && al16(da_out[i]),
                "fst_wn_stack_bwd: operands of layer %d must be 16-byte aligned", i);
    p.ts[i] = ts[i]; p.img_b[i] = static_cast<const char*>(img_b[i]); p.img_d[i] = static_cast<const char*>(img_d[i]);
    p.dg[i] = dg[i]; p.da_out[i] = da_out[i];
    p.rs_b[i] = rs_b ? rs_b[i] : nullptr; p.rs_d[i] = rs_d ? rs_d[i] : nullptr;
    p.dil[i] = 1 << i;
    wn_dgrad_geometry(p.dil[i], &p.nblkw[i], &p.gsw[i], &p.slot_d[i]);
    if ((size_t)2 * p.slot_d[i] > lds_bytes) lds_bytes = (size_t)2 * p.slot_d[i];
  }
  FST_REQUIRE(al16(d_out) && al16(d_u0), "fst_wn_stack_bwd: d_out / d_u0 must be 16-byte aligned");
  FST_REQUIRE(lds_bytes <= 160 * 1024, "fst_wn_stack_bwd: %zu bytes of LDS", lds_bytes);
  p.d_out = d_out; p.d_u0 = d_u0; p.d_u0_bs = d_u0_bs;
  p.nl = nl; p.B = B; p.L = L; p.n = n; p.h = h; p.CH = wn_ch(n); p.CHK = (2 * n + 15) / 16;
  if (int rc = fst_allow_full_lds((const void*)wn_stack_bwd_kernel, "fst_wn_stack_bwd")) return rc;
  const int cus = fst_cu_count() > 0 ? fst_cu_count() : 256;
  const int grid = B < cus ? B : cus;                    // one resident workgroup per CU walks its batch elements
  hipLaunchKernelGGL(wn_stack_bwd_kernel, dim3((unsigned)grid), dim3(512), lds_bytes, (hipStream_t)stream, p);
  FST_LAUNCH_CHECK();
  return 0;
}
