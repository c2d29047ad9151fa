// Weight gradients of a WaveGlow WN layer for gfx950 (the weight gradients autograd derives for
// Simplified_NF_WaveGlow.py:107-116):
//
//   in_layer + cond_layer   dW_in[m][c][τ] = Σ_{b,t} dg[b,m,t]·a[b,c,t+(τ−1)·dil],   dW_cond[m][c] = Σ_{b,t} dg[b,m,t]·u0[b,c,t]
//   res_skip                dW_rs[m][c]    = Σ_{b,t} [d_a ; d_out][b,m,t]·(t·s)[b,c,t]          (acts = t·s re-formed from the saved halves)
//
// A GEMM whose reduction index is TIME: D[m][k] = Σ_t dy[m][t]·x[k][t] with M = 2n output rows, K = 3n + h (or n) "k-rows"
// and B·L = 131 072 reduction steps.  On v_mfma_f32_32x32x16_bf16 both operands are "row, 8 consecutive time samples" fragments,
// i.e. 32 contiguous bytes of an fp32 activation row — no transposition anywhere.  Design, against the generic
// conv_wgrad_kernel it replaces for these shapes (99 µs, MFMA-busy 0.24, 706 MB through L2 → LDS for 202 MB of operands):
//   * one 8-wave workgroup per CU owns ALL 2n output rows × 192 k-rows (6 blocks; in_layer: two such groups, so dy passes the
//     L2 → LDS path twice instead of four times) for a contiguous range of 32-sample time tiles;
//   * operands reach LDS by LDS-DMA as raw fp32 rows of 16 samples (64 B = one MFMA k-step) through a ring of 4-5 slots with
//     3-4 stages in flight behind counted vmcnt waits, ONE barrier per stage (a two-slot ring of 32-sample tiles, which issues
//     a tile only when its predecessor has landed, ran at a third of this rate: no overlap of one tile's latency with the
//     next one's transfer); a row's four 16-byte pieces are XOR-swizzled by a function of the row, so the fragment reads
//     (two ds_read_b128 per fragment) are bank-conflict-free although rows are 16 banks apart;
//   * each wave multiplies a 2 × 3 (res_skip: 2 × 2) tile of 32×32 blocks: five raw fragments are split into bf16 hi/lo in
//     registers per k-step for 18 triple-MFMAs (hi·hi + hi·lo + lo·hi, fp32 accumulate);
//   * a k-row count one past a multiple of 32 (3·120 + 25 = 385) does not cost a thirteenth block: the leftover row is
//     accumulated on the VALU from the dy fragments the wave holds anyway;
//   * every workgroup stores its partial D tile into its own slab (plain stores, two full 128-B segments per instruction);
//     wn_wgrad_reduce_kernel adds the slabs in a fixed order and writes PyTorch layout: deterministic, no atomics.
// Served: L % 16 == 0, n < 128 (M ≤ 256), h ≤ 32, tap shifts that are multiples of 4 samples (dil % 4 == 0), 16-byte aligned
// tensors; everything else stays on conv_wgrad_kernel (fst_wn_wgrad_ok tells).
#include "fst_common.h"

typedef __bf16 ww_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 ww_bf16x2 __attribute__((ext_vector_type(2)));
typedef float ww_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned ww_u32x4 __attribute__((ext_vector_type(4)));
#define WW_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define WW_LDS_VOID(p) ((__attribute__((address_space(3))) void*)(p))

#define WW_TT 16          // time samples per stage = one MFMA k-step = one 64-byte LDS row
#define WW_MROWS 256      // staged dy rows (8 blocks of 32)
#define WW_MAX_NI 5       // LDS-DMA instructions (16 rows each) per wave and stage

__device__ __attribute__((aligned(16))) float ww_zero16[4] = {0.f, 0.f, 0.f, 0.f};

struct WwSeg {
  const float* ptr;     // rows of a [B][.][L] tensor: row c of sample b at ptr + b·bs + c·L
  long long bs;         // batch stride (floats)
  long long mul_off;    // product operand: the partner row lives mul_off floats further (t·s halves); 0 = none
  int rows, shift;      // rows of this segment; time shift of the row's samples (x[k][t] = row[t + shift])
  int out, out_off, out_sc, out_sm;   // reduce: k-row c of this segment, output row m → w[out][m·out_sm + c·out_sc + out_off]
};

struct WwParams {
  WwSeg dy[2];          // output-row segments (M rows in all)
  WwSeg x[4];           // k-row segments (K rows in all, the first K_main of them on the matrix cores)
  int n_dy, n_x, M, K, K_main, n_extra;
  int xr;               // staged k-rows per group = 32·2·KT
  int mul;              // 1: every k-row is staged twice (row and partner) and multiplied when its fragment is read
  int n_groups, ksplit, B, L, tiles_per_seq, n_tiles;
  int R;                // staged rows per stage (multiple of 16)
  int ns;               // ring slots
  int misaligned;       // some tap shift is not a multiple of 4 samples (|shift| < 4): straddling pieces are patched in LDS
  int Kcols;            // slab row length = n_groups·xr
  float* slab;          // [ksplit][256][Kcols]
  float* slab_extra;    // [ksplit][256][2]
  float* w[2];          // reduce outputs
};

__device__ __forceinline__ void ww_split_pair(float a, float b, unsigned& hi, unsigned& lo) {
  const ww_f32x2 v = {a, b};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, ww_bf16x2));
  const ww_f32x2 r = {a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u)};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, ww_bf16x2));
}

__device__ __forceinline__ void ww_split8(const float4& v0, const float4& v1, ww_bf16x8& hi, ww_bf16x8& lo) {
  ww_u32x4 h, l;
  unsigned hh, ll;
  ww_split_pair(v0.x, v0.y, hh, ll); h[0] = hh; l[0] = ll;
  ww_split_pair(v0.z, v0.w, hh, ll); h[1] = hh; l[1] = ll;
  ww_split_pair(v1.x, v1.y, hh, ll); h[2] = hh; l[2] = ll;
  ww_split_pair(v1.z, v1.w, hh, ll); h[3] = hh; l[3] = ll;
  hi = __builtin_bit_cast(ww_bf16x8, h);
  lo = __builtin_bit_cast(ww_bf16x8, l);
}

template <int N>
__device__ __forceinline__ void ww_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void ww_wait_sw(int n) {      // counted wait for a wave-uniform run-time count
  if (n >= N) { ww_wait_vmcnt<N>(); return; }
  if constexpr (N > 0) ww_wait_sw<N - 1>(n);
}

// A staged row is 16 samples = four 16-byte pieces (64 B); row r sits at r·64 and its piece q at slot q ^ ww_g((r & 31) >> 2):
// the sixteen lanes a ds_read_b128 services together (rows {0-3, 12-15, 20-27} or {4-11, 16-19, 28-31} of a block) then
// cover all 64 banks (checked exhaustively on the host side of the tests' development; rows alone are 16 banks apart).
__device__ __forceinline__ int ww_g(int x) { return x < 4 ? (x >> 1) : 2 + (x & 1); }

// FULL: every output-row block and every k-row block of every group is live (M = 256-ish, K_main a multiple of the group size):
// no block tests in the inner loop.
template <int MT, int KT, bool FULL>
__global__ __launch_bounds__(512, 2) void wn_wgrad_kernel(WwParams p) {
  extern __shared__ __attribute__((aligned(16))) char ww_lds[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave_s >> 1, wk = wave_s & 1;          // 4 (output-row pairs) × 2 (k-row halves of the group)
  const int g = blockIdx.y, L = p.L;
  const int slot_bytes = p.R * 64;
  const char* const zero16 = reinterpret_cast<const char*>(ww_zero16);

  // ---- per (LDS-DMA instruction of this wave, lane): which 16 bytes of which tensor row it fetches.  Instruction i fills the
  // LDS rows 16i..16i+15 (1 KiB, lane-linear): lane → row 16i + (lane >> 2), piece slot lane & 3.  Staged rows: [0, 256) dy
  // rows, [256, 256 + xr) this group's k-rows, [.., + xr) their partners (product operand), then 16 rows for the leftover
  // k-rows (group 0 only).
  const int NI = p.R >> 4;
  const int my_ni = (NI - wave_s + 7) >> 3;               // instructions i = wave, wave + 8, ...  (wave-uniform count)
  const float* src0[WW_MAX_NI];                           // address of the piece in batch element 0 at t0 = 0 (null: zero fill)
  int src_bs[WW_MAX_NI], src_t[WW_MAX_NI];                // batch stride (floats); first sample of the piece relative to t0
#pragma unroll
  for (int k = 0; k < WW_MAX_NI; ++k) {
    src0[k] = nullptr; src_bs[k] = 0; src_t[k] = 0;
    const int i = wave_s + 8 * k;
    if (k >= my_ni) continue;
    const int r = 16 * i + (lane >> 2);
    const int q = (lane ^ ww_g((r & 31) >> 2)) & 3;
    int which = -1, c = 0;                                 // 0, 1: dy segments; 2..5: x segments
    bool partner = false;
    if (r < WW_MROWS) {
      if (r < p.M) {
        if (p.n_dy > 1 && r >= p.dy[0].rows) { which = 1; c = r - p.dy[0].rows; } else { which = 0; c = r; }
      }
    } else {
      const int rr = r - WW_MROWS;
      int kk = -1;
      if (rr < p.xr) kk = g * p.xr + rr;
      else if (p.mul && rr < 2 * p.xr) { kk = g * p.xr + rr - p.xr; partner = true; }
      else {
        const int e = rr - p.xr * (1 + p.mul);            // leftover rows: the k-rows K_main .. K-1, staged by group 0
        if (g == 0 && e < p.n_extra) kk = p.K_main + e;
      }
      if (kk >= 0 && kk < p.K && (kk < p.K_main || rr >= p.xr * (1 + p.mul))) {
        int s = 0, c0 = kk;
#pragma unroll
        for (int j = 0; j < 3; ++j)
          if (s + 1 < p.n_x && c0 >= p.x[s].rows) { c0 -= p.x[s].rows; ++s; }
        which = 2 + s; c = c0;
      }
    }
    // the segment's fields by a chain of compares (kernel arguments are read with scalar loads; no pointer into the argument block)
    const float* sp = nullptr;
    long long sbs = 0, smo = 0;
    int ssh = 0;
#define WW_PICK(W, SEG) if (which == (W)) { sp = SEG.ptr; sbs = SEG.bs; smo = SEG.mul_off; ssh = SEG.shift; }
    WW_PICK(0, p.dy[0]) WW_PICK(1, p.dy[1]) WW_PICK(2, p.x[0]) WW_PICK(3, p.x[1]) WW_PICK(4, p.x[2]) WW_PICK(5, p.x[3])
#undef WW_PICK
    if (which >= 0) {
      src0[k] = sp + ((long long)c * L + ssh + 4 * q) + (partner ? smo : 0);
      src_bs[k] = (int)sbs;
      src_t[k] = ssh + 4 * q;
    }
  }

  auto issue = [&](int tile, int slot) {
    const int b = tile / p.tiles_per_seq;
    const int t0 = (tile - b * p.tiles_per_seq) * WW_TT;
    char* const sl = ww_lds + slot * slot_bytes;
#pragma unroll
    for (int k = 0; k < WW_MAX_NI; ++k) {
      if (k >= my_ni) break;                              // wave-uniform
      const int i = wave_s + 8 * k;
      const int t = t0 + src_t[k];
      const bool ok = src0[k] != nullptr && t > -4 && t < L;   // the piece overlaps the sequence (a straddling one is patched below)
      const char* src = ok ? reinterpret_cast<const char*>(src0[k] + ((long long)b * src_bs[k] + t0)) : zero16;
      __builtin_amdgcn_global_load_lds(WW_GLOBAL_PTR(src), WW_LDS_VOID(sl + i * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[MT][KT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < KT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float ev[MT][2];
#pragma unroll
  for (int i = 0; i < MT; ++i) ev[i][0] = ev[i][1] = 0.f;

  const int tile_begin = (int)(((long long)blockIdx.x * p.n_tiles) / p.ksplit);
  const int tile_end = (int)(((long long)(blockIdx.x + 1) * p.n_tiles) / p.ksplit);
  const int m_blocks = (p.M + 31) >> 5;
  const int k_blocks_here = min(2 * KT, ((p.K_main + 31) >> 5) - g * 2 * KT);   // live k-row blocks of this group
  const bool do_extra = g == 0 && wk == 0 && p.n_extra > 0;

  // fragment addressing: row block·32 + l31, pieces 2·half and 2·half + 1 (the 8 samples of this lane half)
  const int gsw = ww_g(l31 >> 2);
  const int row_l = l31 << 6;
  const int pc0 = (((2 * half) ^ gsw) & 3) << 4, pc1 = (((2 * half + 1) ^ gsw) & 3) << 4;
  const int a_base = ((wm * MT * 32) << 6) + row_l;
  const int x_base = ((WW_MROWS + wk * KT * 32) << 6) + row_l, xp_base = x_base + (p.xr << 6);
  const int e_base = (WW_MROWS + p.xr * (1 + p.mul)) << 6;      // rows e < 16 of this octet: g(e >> 2) = e >> 3 for e < 8: 0

  // ring: `depth` stages in flight; one barrier per stage
  const int ns = p.ns, depth = ns - 1;
  const int n_st = tile_end - tile_begin;
  for (int d = 0; d < depth && d < n_st; ++d) issue(tile_begin + d, d);
  int slot = 0;
  for (int c = 0; c < n_st; ++c) {
    // stages still allowed in flight behind stage c: those already issued, i.e. min(depth - 1, n_st - 1 - c)
    const int newer = min(depth - 1, n_st - 1 - c);
    ww_wait_sw<16>(newer * my_ni);
    __builtin_amdgcn_s_barrier();                          // everyone's pieces of stage c have landed; stage c - 1 has been read by all
    if (c + depth < n_st) {
      int sl_i = slot + depth;
      if (sl_i >= ns) sl_i -= ns;
      issue(tile_begin + c + depth, sl_i);
    }
    const char* const sl = ww_lds + slot * slot_bytes;
    if (p.misaligned) {                                    // kernel argument: uniform
      // tap shifts that are not multiples of 4 samples (dilation 1, 2): the LDS-DMA source is only 4-byte aligned (the hardware
      // takes it), and in the first / last stage of a sequence one piece per shifted row straddles the sequence's end: its
      // out-of-range samples are a neighbouring row's data — zero them here, behind one more barrier (2 stages in 32)
      const int tile = tile_begin + c, bq = tile / p.tiles_per_seq, t0 = (tile - bq * p.tiles_per_seq) * WW_TT;
      const bool at_start = t0 == 0, at_end = t0 + WW_TT >= L;
      if (at_start || at_end) {
        if (tid < p.xr) {
          const int kk = g * p.xr + tid;
          int sh = 0, c0 = kk, si = 0;
#pragma unroll
          for (int j = 0; j < 3; ++j)
            if (si + 1 < p.n_x && c0 >= p.x[si].rows) { c0 -= p.x[si].rows; ++si; }
          sh = si == 0 ? p.x[0].shift : (si == 1 ? p.x[1].shift : (si == 2 ? p.x[2].shift : p.x[3].shift));
          if (kk < p.K_main && (sh & 3) != 0) {
            float* row = reinterpret_cast<float*>(ww_lds + slot * slot_bytes + ((WW_MROWS + tid) << 6));
            const int gs = ww_g((tid & 31) >> 2);
            if (at_start && sh < 0 && sh > -4)             // samples t0 + sh + j < 0 of piece 0
              for (int j = 0; j < -sh; ++j) row[(((0 ^ gs) & 3) << 2) + j] = 0.f;
            if (at_end && sh > 0 && sh < 4)                // samples t0 + sh + 12 + j >= L of piece 3
              for (int j = 4 - sh; j < 4; ++j) row[(((3 ^ gs) & 3) << 2) + j] = 0.f;
          }
        }
        __syncthreads();
      }
    }
    ww_bf16x8 ah[MT], al[MT];
    float4 araw[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const char* ap = sl + a_base + ((i * 32) << 6);
      araw[i][0] = *reinterpret_cast<const float4*>(ap + pc0);
      araw[i][1] = *reinterpret_cast<const float4*>(ap + pc1);
      ww_split8(araw[i][0], araw[i][1], ah[i], al[i]);
    }
    if (do_extra) {
      // leftover k-rows on the VALU: every lane of a half reads the same 8 samples of the row (a broadcast), lane = output row
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (e >= p.n_extra) break;
        const char* ep = sl + e_base + (e << 6);
        const float4 x0 = *reinterpret_cast<const float4*>(ep + ((2 * half) << 4));
        const float4 x1 = *reinterpret_cast<const float4*>(ep + ((2 * half + 1) << 4));
#pragma unroll
        for (int i = 0; i < MT; ++i)
          ev[i][e] += (araw[i][0].x * x0.x + araw[i][0].y * x0.y) + (araw[i][0].z * x0.z + araw[i][0].w * x0.w) +
                      (araw[i][1].x * x1.x + araw[i][1].y * x1.y) + (araw[i][1].z * x1.z + araw[i][1].w * x1.w);
      }
    }
#pragma unroll
    for (int j = 0; j < KT; ++j) {
      if (!FULL && wk * KT + j >= k_blocks_here) break;    // wave-uniform: blocks beyond K hold zeros
      const char* bp = sl + x_base + ((j * 32) << 6);
      float4 b0 = *reinterpret_cast<const float4*>(bp + pc0);
      float4 b1 = *reinterpret_cast<const float4*>(bp + pc1);
      if (p.mul) {                                         // kernel argument: uniform
        const char* qp = sl + xp_base + ((j * 32) << 6);
        const float4 c0 = *reinterpret_cast<const float4*>(qp + pc0);
        const float4 c1 = *reinterpret_cast<const float4*>(qp + pc1);
        b0.x *= c0.x; b0.y *= c0.y; b0.z *= c0.z; b0.w *= c0.w;
        b1.x *= c1.x; b1.y *= c1.y; b1.z *= c1.z; b1.w *= c1.w;
      }
      ww_bf16x8 bh, bl;
      ww_split8(b0, b1, bh, bl);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        if (!FULL && wm * MT + i >= m_blocks) break;       // wave-uniform
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh, acc[i][j], 0, 0, 0);
      }
    }
    slot = slot + 1 == ns ? 0 : slot + 1;
  }

  // ---- partial D tile → this workgroup's slab: register r of a tile = output row (r&3) + 8(r>>2) + 4·half, lane & 31 = k-row:
  // one wave-instruction stores two rows × 32 consecutive floats
  float* const slab = p.slab + (long long)blockIdx.x * WW_MROWS * p.Kcols;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    if (wm * MT + i >= m_blocks) break;
#pragma unroll
    for (int j = 0; j < KT; ++j) {
      if (wk * KT + j >= k_blocks_here) break;
      float* dst = slab + (long long)((wm * MT + i) * 32 + 4 * half) * p.Kcols + g * p.xr + (wk * KT + j) * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[(long long)((r & 3) + 8 * (r >> 2)) * p.Kcols] = acc[i][j][r];
    }
  }
  if (do_extra) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      if (wm * MT + i >= m_blocks) break;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float s = ev[i][e] + __shfl_xor(ev[i][e], 32, 64);
        if (half == 0 && e < p.n_extra) p.slab_extra[((long long)blockIdx.x * WW_MROWS + (wm * MT + i) * 32 + l31) * 2 + e] = s;
      }
    }
  }
}

// out[m][k-row] = Σ_slabs, written in PyTorch layout: k-row kk of x segment s, channel c → w[seg.out][m·out_sm + c·out_sc + out_off].
// One workgroup per (output row, 256 k-rows): thread (column quad cx, slab group sg) adds every fourth slab of its 16 bytes with
// eight loads in flight, the four groups meet in LDS — fixed order, so the result is the same bits on every run.  (One thread
// per element walking all slabs on its own had ~6 KB in flight per CU: 60 µs for 50 MB.)
__device__ __forceinline__ void ww_scatter(const WwParams& p, int m, int kk, float v) {
  int si = 0, c = kk;
#pragma unroll
  for (int j = 0; j < 3; ++j)
    if (si + 1 < p.n_x && c >= p.x[si].rows) { c -= p.x[si].rows; ++si; }
  int out = p.x[0].out, off = p.x[0].out_off, sc = p.x[0].out_sc, sm = p.x[0].out_sm;
  if (si == 1) { out = p.x[1].out; off = p.x[1].out_off; sc = p.x[1].out_sc; sm = p.x[1].out_sm; }
  if (si == 2) { out = p.x[2].out; off = p.x[2].out_off; sc = p.x[2].out_sc; sm = p.x[2].out_sm; }
  if (si == 3) { out = p.x[3].out; off = p.x[3].out_off; sc = p.x[3].out_sc; sm = p.x[3].out_sm; }
  float* w = out == 0 ? p.w[0] : p.w[1];
  w[(long long)m * sm + (long long)c * sc + off] = v;
}

__global__ __launch_bounds__(256) void wn_wgrad_reduce_kernel(WwParams p) {
  __shared__ float4 part[4][64];
  const int m = blockIdx.y, cx = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const int kk0 = blockIdx.x * 256 + cx * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (kk0 < p.Kcols) {
    const long long st = (long long)WW_MROWS * p.Kcols;
    const float* q = p.slab + (long long)m * p.Kcols + kk0;
    int sl = sg;
    for (; sl + 28 < p.ksplit; sl += 32) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(q + (sl + 4 * u) * st);
#pragma unroll
      for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; sl < p.ksplit; sl += 4) {
      const float4 v = *reinterpret_cast<const float4*>(q + sl * st);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  part[sg][cx] = s;
  __syncthreads();
  if (sg == 0 && kk0 < p.Kcols) {
    float4 t = part[0][cx];
#pragma unroll
    for (int g2 = 1; g2 < 4; ++g2) { const float4 o = part[g2][cx]; t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
    const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (kk0 + j < p.K_main) ww_scatter(p, m, kk0 + j, tv[j]);
  }
  // the leftover k-rows (VALU sums): the last column block's idle slab-group-1 threads take them
  if (blockIdx.x == gridDim.x - 1 && sg == 1 && cx < p.n_extra && p.K_main < p.K) {
    const float* q = p.slab_extra + (long long)m * 2 + cx;
    float e = 0.f;
    for (int sl = 0; sl < p.ksplit; ++sl) e += q[(long long)sl * WW_MROWS * 2];
    ww_scatter(p, m, p.K_main + cx, e);
  }
}

// ------------------------------------------------------------------------------------------------ host side
static inline bool ww_al16(const void* q) { return q == nullptr || (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

// geometry shared by the size query and the launchers; kind 0 = in_layer + cond_layer, 1 = res_skip
static int ww_geometry(int kind, int B, int L, int n, int h, int last, WwParams* p) {
  const int KT = kind == 0 ? 3 : 2;
  p->M = kind == 0 ? 2 * n : (last ? n : 2 * n);
  p->K = kind == 0 ? 3 * n + h : n;
  const int rem = p->K & 31;
  p->n_extra = (kind == 0 && rem >= 1 && rem <= 2) ? rem : 0;      // one or two k-rows past a multiple of 32: VALU rows
  p->K_main = p->n_extra ? p->K - rem : p->K;
  p->xr = 32 * 2 * KT;
  const int kb = (p->K_main + 31) / 32;
  p->n_groups = (kb + 2 * KT - 1) / (2 * KT);
  p->mul = kind == 1;
  p->R = WW_MROWS + p->xr * (1 + p->mul) + (p->n_extra ? 16 : 0);
  p->ns = (int)((160 * 1024) / ((size_t)p->R * 64));
  if (p->ns > 5) p->ns = 5;
  p->Kcols = p->n_groups * p->xr;
  p->B = B; p->L = L;
  p->tiles_per_seq = L / WW_TT;
  p->n_tiles = B * p->tiles_per_seq;
  const int cus = fst_cu_count() > 0 ? fst_cu_count() : 256;
  int ks = cus / p->n_groups;
  if (ks < 1) ks = 1;
  if (ks > p->n_tiles) ks = p->n_tiles;
  p->ksplit = ks;
  return KT;
}

// 1: served; 2 (kind 0 only): served if 16 bytes in front of and behind `a` are readable (dilation 1-3: a tap's 16-byte pieces
// start up to 3 samples outside a row); 0: not served
extern "C" int fst_wn_wgrad_ok(int kind, int B, int L, int n, int h, int dil) {
  if (!(B > 0 && L > 0 && L % WW_TT == 0 && n > 0 && n < 128)) return 0;      // (16-sample stages)
  if (kind == 0) {
    if (!(h > 0 && h <= 32 && dil > 0 && (3 * n + h + 31) / 32 <= 18)) return 0;
    return dil % 4 == 0 ? 1 : (dil < 4 ? 2 : 0);
  }
  return kind == 1;
}

extern "C" int64_t fst_wn_wgrad_workspace_floats(int kind, int B, int L, int n, int h, int last) {
  if (!fst_wn_wgrad_ok(kind, B, L, n, h, 4)) return -1;                        // (the workspace does not depend on the dilation)
  WwParams p;
  ww_geometry(kind, B, L, n, h, last, &p);
  return (int64_t)p.ksplit * WW_MROWS * (p.Kcols + 2);
}

static int ww_launch(WwParams& p, int KT, void* stream) {
  const size_t lds = (size_t)p.ns * p.R * 64;
  FST_REQUIRE(p.ns >= 2 && lds <= 160 * 1024 && (p.R >> 4) <= 8 * WW_MAX_NI && ((p.R >> 4) + 7) / 8 * (p.ns - 2) <= 16,
              "fst_wn_wgrad: %d staged rows per stage do not fit (ring of %d slots, LDS %zu B)", p.R, p.ns, lds);
  const bool full = (p.M + 31) / 32 == 8 && ((p.K_main + 31) / 32) % (2 * KT) == 0;
  void (*fn)(WwParams) = KT == 3 ? (full ? wn_wgrad_kernel<2, 3, true> : wn_wgrad_kernel<2, 3, false>)
                                 : (full ? wn_wgrad_kernel<2, 2, true> : wn_wgrad_kernel<2, 2, false>);
  if (int rc = fst_allow_full_lds((const void*)fn, "fst_wn_wgrad")) return rc;
  hipLaunchKernelGGL(fn, dim3((unsigned)p.ksplit, (unsigned)p.n_groups), dim3(512), lds, (hipStream_t)stream, p);
  FST_LAUNCH_CHECK();
  hipLaunchKernelGGL(wn_wgrad_reduce_kernel, dim3((unsigned)((p.Kcols + 255) / 256), (unsigned)p.M), dim3(256), 0, (hipStream_t)stream, p);
  FST_LAUNCH_CHECK();
  return 0;
}

extern "C" int fst_wn_wgrad_in(const float* dg, const float* a, const float* u0, int64_t u0_bs, float* dw_in, float* dw_cond,
                               float* workspace, int64_t workspace_floats, int B, int L, int n, int h, int dil, int a_slack,
                               int64_t numel_a, void* stream) {
  FST_REQUIRE(dg && a && u0 && dw_in && dw_cond && workspace, "fst_wn_wgrad_in: null operand");
  const int served = fst_wn_wgrad_ok(0, B, L, n, h, dil);
  FST_REQUIRE(served == 1 || (served == 2 && a_slack), "fst_wn_wgrad_in: unsupported shape B=%d L=%d n=%d h=%d dil=%d (needs L %% 16 == 0, "
              "n < 128, h <= 32, and dil %% 4 == 0 or — with 16 readable bytes either side of a — dil < 4)", B, L, n, h, dil);
  FST_REQUIRE((long long)B * n * L == (long long)numel_a, "fst_wn_wgrad_in: B*n*L does not match the element count %lld of a", (long long)numel_a);
  FST_REQUIRE(B == 1 || u0_bs >= (int64_t)h * L, "fst_wn_wgrad_in: u0 batch stride %lld < h*L", (long long)u0_bs);
  FST_REQUIRE(u0_bs % 4 == 0 && ww_al16(dg) && ww_al16(a) && ww_al16(u0) && ww_al16(workspace), "fst_wn_wgrad_in: operands must be 16-byte aligned");
  WwParams p = {};
  const int KT = ww_geometry(0, B, L, n, h, 0, &p);
  FST_REQUIRE(workspace_floats >= (int64_t)p.ksplit * WW_MROWS * (p.Kcols + 2), "fst_wn_wgrad_in: workspace of %lld floats is too small",
              (long long)workspace_floats);
  p.n_dy = 1;
  p.dy[0] = {dg, (long long)2 * n * L, 0, 2 * n, 0, 0, 0, 0, 0};
  p.n_x = 4;
  for (int tap = 0; tap < 3; ++tap) p.x[tap] = {a, (long long)n * L, 0, n, (tap - 1) * dil, 0, tap, 3, 3 * n};
  p.x[3] = {u0, (long long)u0_bs, 0, h, 0, 1, 0, 1, h};
  p.misaligned = dil % 4 != 0;
  p.slab = workspace;
  p.slab_extra = workspace + (long long)p.ksplit * WW_MROWS * p.Kcols;
  p.w[0] = dw_in; p.w[1] = dw_cond;
  return ww_launch(p, KT, stream);
}

extern "C" int fst_wn_wgrad_rs(const float* d_a, const float* d_out, const float* ts, float* dw_rs, float* workspace,
                               int64_t workspace_floats, int last, int B, int L, int n, int64_t numel_a, void* stream) {
  FST_REQUIRE(d_out && ts && dw_rs && workspace && (last || d_a), "fst_wn_wgrad_rs: null operand");
  FST_REQUIRE(fst_wn_wgrad_ok(1, B, L, n, 0, 4), "fst_wn_wgrad_rs: unsupported shape B=%d L=%d n=%d (needs L %% 16 == 0, n < 128)", B, L, n);
  FST_REQUIRE((long long)B * n * L == (long long)numel_a, "fst_wn_wgrad_rs: B*n*L does not match the element count %lld", (long long)numel_a);
  FST_REQUIRE(ww_al16(d_a) && ww_al16(d_out) && ww_al16(ts) && ww_al16(workspace), "fst_wn_wgrad_rs: operands must be 16-byte aligned");
  WwParams p = {};
  const int KT = ww_geometry(1, B, L, n, 0, last, &p);
  FST_REQUIRE(workspace_floats >= (int64_t)p.ksplit * WW_MROWS * (p.Kcols + 2), "fst_wn_wgrad_rs: workspace of %lld floats is too small",
              (long long)workspace_floats);
  if (last) {
    p.n_dy = 1;
    p.dy[0] = {d_out, (long long)n * L, 0, n, 0, 0, 0, 0, 0};
  } else {
    p.n_dy = 2;
    p.dy[0] = {d_a, (long long)n * L, 0, n, 0, 0, 0, 0, 0};
    p.dy[1] = {d_out, (long long)n * L, 0, n, 0, 0, 0, 0, 0};
  }
  p.n_x = 1;
  p.x[0] = {ts, (long long)2 * n * L, (long long)n * L, n, 0, 0, 0, 1, n};   // t rows; the s rows n·L floats further
  p.slab = workspace;
  p.slab_extra = workspace + (long long)p.ksplit * WW_MROWS * p.Kcols;
  p.w[0] = dw_rs; p.w[1] = nullptr;
  return ww_launch(p, KT, stream);
}
