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
//   * operands reach LDS by LDS-DMA as raw fp32 rows of 32 samples (128 B: whole cache lines — 64-byte rows fetched every line
//     twice, 152 µs) through two rings behind counted vmcnt waits, ONE barrier per stage: three slots for the dy rows (two
//     stages ahead), two for the k-rows (all that fits in 160 KiB), so the wait of a stage exposes the latency and transfer of
//     the k-rows only; a row's eight 16-byte pieces are XOR-swizzled by (row & 7) and every second octet of rows swaps row
//     parity, so the fragment reads (two ds_read_b128 per fragment) are bank-conflict-free although rows are 128 B apart;
//   * each wave multiplies a 2 × 3 (res_skip: 2 × 2) tile of 32×32 blocks: five raw fragments are split into bf16 hi/lo in
//     registers per k-step for 18 triple-MFMAs (hi·hi + hi·lo + lo·hi, fp32 accumulate);
//   * a k-row count one past a multiple of 32 (3·120 + 25 = 385) does not cost a thirteenth block: the leftover row is
//     accumulated on the VALU from the dy fragments the wave holds anyway;
//   * every workgroup stores its partial D tile into its own slab (plain stores, two full 128-B segments per instruction);
//     wn_wgrad_reduce_kernel adds the slabs in a fixed order and writes PyTorch layout: deterministic, no atomics.
// Served: L % 32 == 0, n < 128 (M ≤ 256), h ≤ 32, tap shifts that are multiples of 4 samples (dil % 4 == 0), 16-byte aligned
// tensors; everything else stays on conv_wgrad_kernel (fst_wn_wgrad_ok tells).
#include <stdlib.h>
#include <type_traits>

#include "fst_common.h"

typedef __bf16 ww_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 ww_bf16x2 __attribute__((ext_vector_type(2)));
typedef float ww_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned ww_u32x4 __attribute__((ext_vector_type(4)));
#define WW_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define WW_LDS_VOID(p) ((__attribute__((address_space(3))) void*)(p))

#define WW_TT 32          // time samples per stage = two MFMA k-steps = one 128-byte LDS row (a full cache line per row)
#define WW_MROWS 256      // staged dy rows (8 blocks of 32)
#define WW_ND 3           // ring slots of the dy rows
#define WW_NX 2           // ring slots of the k-rows
#define WW_MAX_NX 5       // LDS-DMA instructions (8 rows each) per wave and stage for the k-rows

__device__ __attribute__((aligned(16))) float ww_zero16[4] = {0.f, 0.f, 0.f, 0.f};

#define WW_MAX_SETS 3
struct WwSeg {
  const float* ptr[WW_MAX_SETS];   // one tensor per operand set (application of the WN); rows of a [B][.][L] tensor: row c of sample
                                   // b of set s at ptr[s] + b·bs + c·L
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
  int n_groups, ksplit, B, L, tiles_per_seq, n_tiles;   // n_tiles: per operand set
  // fst_nt_gemm only — RandomLayer's epilogue (C_DAN.py:20-25): C[m][n] = acc·epi_scale·Σ_c epi_p[m][c]·epi_r1[c][n], the plain product
  // acc kept in epi_raw (the backward needs it); all null / 0 for the weight gradients
  const float* epi_p;
  const float* epi_r1;
  float* epi_raw;
  int epi_ncls;
  float epi_scale;
  int n_sets;           // operand sets whose gradients are SUMMED (the applications of one WN in a train step share their weights:
                        // one launch, one set of slabs and one reduction for all of them); workgroup x works on set x % n_sets
  int RX;               // staged rows of a k-row slot: xr·(1 + mul) (+ 8 for the leftover rows), a multiple of 8
  int misaligned;       // some tap shift is not a multiple of 4 samples (|shift| < 4): straddling pieces are patched in LDS
  int exp;              // diagnostics (FST_WW_EXP, timing only, wrong results): 1 every LDS-DMA piece from the zero block, 2 no k-step
                        // arithmetic (no LDS reads, splits, MFMAs), 4 no LDS-DMA at all
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

template <class V4>
__device__ __forceinline__ void ww_split8(const V4& v0, const V4& v1, ww_bf16x8& hi, ww_bf16x8& lo) {
  ww_u32x4 h, l;
  unsigned hh, ll;
  ww_split_pair(v0.x, v0.y, hh, ll); h[0] = hh; l[0] = ll;
  ww_split_pair(v0.z, v0.w, hh, ll); h[1] = hh; l[1] = ll;
  ww_split_pair(v1.x, v1.y, hh, ll); h[2] = hh; l[2] = ll;
  ww_split_pair(v1.z, v1.w, hh, ll); h[3] = hh; l[3] = ll;
  hi = __builtin_bit_cast(ww_bf16x8, h);
  lo = __builtin_bit_cast(ww_bf16x8, l);
}

// LDS fragment reads as inline asm.  hipcc's wait-count pass cannot tell ring slots apart: in front of the first compiler-visible
// LDS read that follows an LDS-DMA issue it places s_waitcnt vmcnt(0) — which drained both rings every stage (the DMA of the
// next stages, issued a few instructions earlier, had to land before the current stage could be multiplied: 139 µs).  The pass
// does not look inside asm; ordering is by the counted vmcnt + barrier at the top of the stage, and the reads are retired by
// the explicit lgkmcnt wait + scheduling barrier below.
typedef float ww_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ ww_f32x4 ww_lds_read16(const char* p) {
  ww_f32x4 v;
  const unsigned addr = (unsigned)(uintptr_t)((__attribute__((address_space(3))) const char*)p);
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void ww_lds_write16(char* p, const ww_u32x4& v) {
  const unsigned addr = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)p);
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void ww_lds_zero4(char* p) {
  const unsigned addr = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)p);
  asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(0.0f) : "memory");
}
// One LDS-DMA piece as inline asm too: with the builtin the same pass put s_waitcnt vmcnt(0) in front of the address arithmetic of
// every piece but the first once the issues were spread between the MFMAs (each issue then waited for all earlier pieces to land).
// M0 carries the wave-uniform LDS byte address; the 64 lanes' 16 bytes land at M0 + 16·lane.  M0 is a register the compiler
// reserves for itself (it may not appear in a clobber list: "may not be preserved"), so the block saves it and puts it back:
// the instruction reads M0 when it issues, the restore right behind it is safe.
__device__ __forceinline__ void ww_dma16(const char* gsrc, char* lds_dst) {
  const unsigned addr = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)lds_dst);
  unsigned saved_m0;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(saved_m0) : "v"(gsrc), "s"(addr) : "memory");
}
__device__ __forceinline__ void ww_lds_wait() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);                       // nothing that consumes a read may be hoisted above the wait
}

// 8 consecutive samples (two raw 16-byte pieces) → their bf16 hi parts and lo parts, 16 bytes each
template <class V4>
__device__ __forceinline__ void ww_split8u(const V4& v0, const V4& v1, ww_u32x4& h, ww_u32x4& l) {
  unsigned hh, ll;
  ww_split_pair(v0.x, v0.y, hh, ll); h[0] = hh; l[0] = ll;
  ww_split_pair(v0.z, v0.w, hh, ll); h[1] = hh; l[1] = ll;
  ww_split_pair(v1.x, v1.y, hh, ll); h[2] = hh; l[2] = ll;
  ww_split_pair(v1.z, v1.w, hh, ll); h[3] = hh; l[3] = ll;
}

template <int N>
__device__ __forceinline__ void ww_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// A staged row is 32 samples = eight 16-byte pieces (128 B, one cache line of the tensor row).  Logical row r of a region sits at
// LDS row r ^ ((r >> 3) & 1) (odd octets swap the parity of their rows) and its piece q at slot (q ^ r) & 7: the sixteen lanes a
// ds_read_b128 services together (rows {0-3, 12-15, 20-27} or {4-11, 16-19, 28-31} of a block) then cover all 64 banks although
// rows are a multiple of 128 B apart.  An LDS-DMA instruction fills 8 LDS rows (1 KiB, lane-linear): lane → LDS row 8i + (lane >> 3),
// piece slot lane & 7.
__device__ __forceinline__ int ww_dma_row(int i, int lane) { return 8 * i + ((lane >> 3) ^ (i & 1)); }
// byte offset of piece q of logical row r inside a slot
__device__ __forceinline__ int ww_lds_off(int r, int q) { return ((r ^ ((r >> 3) & 1)) << 7) + (((q ^ r) & 7) << 4); }

// which tensor row / piece a lane of a k-row DMA instruction fetches → (address for batch 0, t0 = 0; batch stride; first sample)
struct WwSrc { const float* p; int bs, t; };

// FULL: every output-row block and every k-row block of every group is live: no block tests in the inner loop.  MUL: product
// operand (every k-row fragment is the product of two staged rows).  NE: leftover k-rows accumulated on the VALU (0 or 1).
// The k-step body is straight-line code (no run-time branches: the compiler then issues all of a k-step's fragment reads up
// front and waits for them one by one as the splits consume them; with uniform branches around the optional parts it read,
// waited, split and multiplied fragment by fragment — 139 µs instead of the ring's rate).
template <int MT, int KT, bool FULL, bool MUL, int NE>
__global__ __launch_bounds__(512, 2) void wn_wgrad_kernel(WwParams p) {
  extern __shared__ __attribute__((aligned(16))) char ww_lds[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave_s >> 1, wk = wave_s & 1;          // 4 (output-row pairs) × 2 (k-row halves of the group)
  const int g = blockIdx.y, L = p.L;
  const int set = blockIdx.x % p.n_sets;                  // this workgroup's operand set (wave-uniform)
  const int dslot_bytes = WW_MROWS * 128, xslot_bytes = p.RX * 128;
  char* const xring = ww_lds + WW_ND * dslot_bytes;
  const char* const zero16 = reinterpret_cast<const char*>(ww_zero16);

  // ---- dy rows: 32 LDS-DMA instructions per stage, 4 per wave (i = wave + 8k)
  WwSrc dsrc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = wave_s + 8 * k;
    const int r = ww_dma_row(i, lane), q = (lane ^ r) & 7;
    dsrc[k].p = nullptr; dsrc[k].bs = 0; dsrc[k].t = 4 * q;
    if (r < p.M) {
      if (p.n_dy > 1 && r >= p.dy[0].rows) {
        dsrc[k].p = p.dy[1].ptr[set] + ((long long)(r - p.dy[0].rows) * L + 4 * q); dsrc[k].bs = (int)p.dy[1].bs;
      } else {
        dsrc[k].p = p.dy[0].ptr[set] + ((long long)r * L + 4 * q); dsrc[k].bs = (int)p.dy[0].bs;
      }
    }
  }
  // ---- k-rows of this group: [0, xr) the rows, [xr, 2·xr) their partners (product operand), then 8 rows for the leftover
  // k-rows (group 0 only); RX/8 instructions per stage
  const int NXI = p.RX >> 3;
  const int my_nx = (NXI - wave_s + 7) >> 3;
  WwSrc xsrc[WW_MAX_NX];
#pragma unroll
  for (int k = 0; k < WW_MAX_NX; ++k) {
    xsrc[k].p = nullptr; xsrc[k].bs = 0; xsrc[k].t = 0;
    const int i = wave_s + 8 * k;
    if (k >= my_nx) continue;
    const int rr = ww_dma_row(i, lane), q = (lane ^ rr) & 7;
    int kk = -1;
    bool partner = false;
    if (rr < p.xr) kk = g * p.xr + rr;
    else if (p.mul && rr < 2 * p.xr) { kk = g * p.xr + rr - p.xr; partner = true; }
    else {
      const int e = rr - p.xr * (1 + p.mul);              // leftover rows: the k-rows K_main .. K-1, staged by group 0
      if (g == 0 && e < p.n_extra) kk = p.K_main + e;
    }
    if (kk >= 0 && kk < p.K && (kk < p.K_main || rr >= p.xr * (1 + p.mul))) {
      int si = 0, c0 = kk;
#pragma unroll
      for (int j = 0; j < 3; ++j)
        if (si + 1 < p.n_x && c0 >= p.x[si].rows) { c0 -= p.x[si].rows; ++si; }
      // the segment's fields by a chain of compares (kernel arguments are read with scalar loads; no pointer into the argument block)
      const float* sp = p.x[0].ptr[set];
      long long sbs = p.x[0].bs, smo = p.x[0].mul_off;
      int ssh = p.x[0].shift;
      if (si == 1) { sp = p.x[1].ptr[set]; sbs = p.x[1].bs; smo = p.x[1].mul_off; ssh = p.x[1].shift; }
      if (si == 2) { sp = p.x[2].ptr[set]; sbs = p.x[2].bs; smo = p.x[2].mul_off; ssh = p.x[2].shift; }
      if (si == 3) { sp = p.x[3].ptr[set]; sbs = p.x[3].bs; smo = p.x[3].mul_off; ssh = p.x[3].shift; }
      xsrc[k].p = sp + ((long long)c0 * L + ssh + 4 * q) + (partner ? smo : 0);
      xsrc[k].bs = (int)sbs;
      xsrc[k].t = ssh + 4 * q;
    }
  }

  // (misaligned taps) the tap shift of the staged k-row `tid` of this group, for the boundary patch below; 0 = nothing to patch
  int patch_sh = 0;
  if (p.misaligned && tid < p.xr) {
    const int kk = g * p.xr + tid;
    int c0 = kk, si = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j)
      if (si + 1 < p.n_x && c0 >= p.x[si].rows) { c0 -= p.x[si].rows; ++si; }
    const int sh = si == 0 ? p.x[0].shift : (si == 1 ? p.x[1].shift : (si == 2 ? p.x[2].shift : p.x[3].shift));
    if (kk < p.K_main && (sh & 3) != 0 && sh > -4 && sh < 4) patch_sh = sh;
  }
  // The tables above are per-lane selections of kernel arguments: the compiler reads some of them with vector loads, and — not
  // seeing the inline-asm waits below — would make every first use inside the stage loop wait for vmcnt(0), i.e. for the whole
  // LDS-DMA ring.  A wait it does see, here, retires them once.
  __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0), expcnt / lgkmcnt untouched

  auto tile_bt = [&](int tile, int& b, int& t0) {
    b = tile / p.tiles_per_seq;
    t0 = (tile - b * p.tiles_per_seq) * WW_TT;
  };
  // one LDS-DMA piece: k-th dy instruction / k-th k-row instruction of this wave for the stage (b, t0) into the given slot
  auto issue_dy1 = [&](int k, int b, int t0, int slot) {
    if (p.exp & 4) return;
    const bool ok = dsrc[k].p != nullptr && t0 + dsrc[k].t < L && !(p.exp & 1);
    const char* src = ok ? reinterpret_cast<const char*>(dsrc[k].p + ((long long)b * dsrc[k].bs + t0)) : zero16;
    ww_dma16(src, ww_lds + slot * dslot_bytes + (wave_s + 8 * k) * 1024);
  };
  auto issue_x1 = [&](int k, int b, int t0, int slot) {
    if (k >= my_nx || (p.exp & 4)) return;                 // wave-uniform
    const int t = t0 + xsrc[k].t;
    const bool ok = xsrc[k].p != nullptr && t > -4 && t < L && !(p.exp & 1);   // the piece overlaps the sequence (a straddling one is patched)
    const char* src = ok ? reinterpret_cast<const char*>(xsrc[k].p + ((long long)b * xsrc[k].bs + t0)) : zero16;
    ww_dma16(src, xring + slot * xslot_bytes + (wave_s + 8 * k) * 1024);
  };
  auto issue_dy = [&](int tile, int slot) {
    int b, t0;
    tile_bt(tile, b, t0);
#pragma unroll
    for (int k = 0; k < 4; ++k) issue_dy1(k, b, t0, slot);
  };
  auto issue_x = [&](int tile, int slot) {
    int b, t0;
    tile_bt(tile, b, t0);
#pragma unroll
    for (int k = 0; k < WW_MAX_NX; ++k) issue_x1(k, b, t0, slot);
  };

  f32x16 acc[MT][KT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < KT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float ev[2] = {0.f, 0.f};                                // leftover k-row: partial dot products of this thread's two dy units

  // the workgroups x ≡ set (mod n_sets) share the set's tiles
  const int wg_in_set = blockIdx.x / p.n_sets, wgs_of_set = (p.ksplit - set + p.n_sets - 1) / p.n_sets;
  const int tile_begin = (int)(((long long)wg_in_set * p.n_tiles) / wgs_of_set);
  const int tile_end = (int)(((long long)(wg_in_set + 1) * p.n_tiles) / wgs_of_set);
  const int m_blocks = (p.M + 31) >> 5;
  const int k_blocks_here = min(2 * KT, ((p.K_main + 31) >> 5) - g * 2 * KT);   // live k-row blocks of this group
  const bool do_extra = NE > 0 && g == 0;

  // fragment addressing: row block·32 + l31 (LDS row l31 ^ ((l31 >> 3) & 1) of the block), pieces 4·ks + 2·half and the next
  const int row_l = (l31 ^ ((l31 >> 3) & 1)) << 7;
  const int sw = l31 & 7;
  int pc[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    pc[ks][0] = (((4 * ks + 2 * half) ^ sw) & 7) << 4;
    pc[ks][1] = (((4 * ks + 2 * half + 1) ^ sw) & 7) << 4;
  }
  const int a_off = ((wm * MT * 32) << 7) + row_l;
  const int x_off = ((wk * KT * 32) << 7) + row_l, xp_off = x_off + (p.xr << 7);
  const int e_off = (p.xr * (1 + p.mul)) << 7;           // the leftover rows' octet: LDS row = e, piece slot = (q ^ e) & 7

  // Two rings, one barrier per stage: the dy rows run two stages ahead (three slots), the k-rows one (two slots — all that fits
  // beside them in 160 KiB).  Issue order per stage c: x(c+1), then dy(c+2); so the wait for stage c leaves exactly the pieces of
  // dy(c+1) in flight, and what is exposed per stage is the latency plus the transfer of the k-rows alone.
  const int n_st = tile_end - tile_begin;
  if (n_st > 0) { issue_dy(tile_begin, 0); issue_x(tile_begin, 0); }
  if (n_st > 1) issue_dy(tile_begin + 1, 1);
  int dslot = 0, xslot = 0;
  for (int c = 0; c < n_st; ++c) {
    if (c + 1 < n_st) ww_wait_vmcnt<4>(); else ww_wait_vmcnt<0>();   // (a wave issues 4 dy pieces per stage)
    __builtin_amdgcn_s_barrier();                          // everyone's pieces of stage c have landed; stage c - 1 has been read by all
    // The pieces of x(c+1), then of dy(c+2), are issued ONE AT A TIME between the MFMA groups below (issue order unchanged, so
    // the counted wait stands): an LDS-DMA issue blocks its wave for 40-150 cycles while the CU's one address path takes the
    // instruction — all eight waves issuing their 7-8 pieces back to back after the barrier cost the whole CU ~1 µs per stage in
    // which nothing multiplied (cost removal: arithmetic 64 µs + LDS-DMA 36 µs = the 97 µs measured); spread out, a wave's blocked
    // issue sits beside its SIMD partner's MFMAs.
    const bool have_x = c + 1 < n_st, have_dy = c + 2 < n_st;
    int nb = 0, nt0 = 0, db = 0, dt0 = 0;
    if (have_x) tile_bt(tile_begin + c + 1, nb, nt0);
    if (have_dy) tile_bt(tile_begin + c + 2, db, dt0);
    const int x_next = xslot ^ 1, d_next = dslot >= 1 ? dslot - 1 : 2;
    auto issue_pos = [&](int pos) {                        // pos 0..4: k-row pieces, 5..8: dy pieces
      if (pos < WW_MAX_NX) { if (have_x) issue_x1(pos, nb, nt0, x_next); }
      else if (pos < WW_MAX_NX + 4) { if (have_dy) issue_dy1(pos - WW_MAX_NX, db, dt0, d_next); }
    };
    const char* const dsl = ww_lds + dslot * dslot_bytes;
    const char* const xsl = xring + xslot * xslot_bytes;
    if (p.misaligned) {                                    // kernel argument: uniform
      // tap shifts that are not multiples of 4 samples (dilation 1-3): the LDS-DMA source is only 4-byte aligned (the hardware
      // takes it), and in the first / last stage of a sequence one piece per shifted row straddles the sequence's end: its
      // out-of-range samples are a neighbouring row's data — zero them here, behind one more barrier (2 stages in L/32)
      int bq, t0;
      tile_bt(tile_begin + c, bq, t0);
      const bool at_start = t0 == 0, at_end = t0 + WW_TT >= L;
      if (at_start || at_end) {
        if (patch_sh != 0) {
          char* rowp = xring + xslot * xslot_bytes + ((tid ^ ((tid >> 3) & 1)) << 7);
          if (at_start && patch_sh < 0)                      // samples t0 + sh + j < 0 of piece 0
            for (int j = 0; j < -patch_sh; ++j) ww_lds_zero4(rowp + (((0 ^ tid) & 7) << 4) + 4 * j);
          if (at_end && patch_sh > 0)                        // samples t0 + sh + 28 + j >= L of piece 7
            for (int j = 4 - patch_sh; j < 4; ++j) ww_lds_zero4(rowp + (((7 ^ tid) & 7) << 4) + 4 * j);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the zeroing stores; NOT __syncthreads(): its fence would drain the ring
        __builtin_amdgcn_s_barrier();
      }
    }
    // ---- split pass: every staged 8-sample unit (two raw 16-byte pieces of a row) is split ONCE per workgroup, in place, into
    // its bf16 hi parts (written over the first piece) and lo parts (over the second) — not once per consuming wave (each dy
    // fragment is consumed by 2 waves, each k-row fragment by 4: 280 VALU instructions per wave and stage against 36 MFMAs).
    // Thread t owns the dy units t and t + 512 (rows t>>2 and 128 + (t>>2), unit t&3) and the k-row units t (and t + 512).
    if (!(p.exp & 2)) {
      char* const dw = ww_lds + dslot * dslot_bytes;
      char* const xw = xring + xslot * xslot_bytes;
      const int u = tid & 3, r0 = tid >> 2;
      ww_f32x4 e0, e1;
      if constexpr (NE > 0) {                              // the leftover k-row's 8 samples of this unit (raw, never split)
        e0 = ww_lds_read16(xw + e_off + ((2 * u) << 4));
        e1 = ww_lds_read16(xw + e_off + ((2 * u + 1) << 4));
      }
      ww_f32x4 d[2][2], xq[2][2], xm[2][2];
      char* da[2][2]; char* xa[2][2];
      const int nxu = (p.xr * 4 > 512 + tid) ? 2 : 1;      // k-row units of this thread: rows r0 and (if it exists) 128 + r0
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = r0 + 128 * i;
        da[i][0] = dw + ww_lds_off(r, 2 * u); da[i][1] = dw + ww_lds_off(r, 2 * u + 1);
        d[i][0] = ww_lds_read16(da[i][0]); d[i][1] = ww_lds_read16(da[i][1]);
        xa[i][0] = xw + ww_lds_off(r, 2 * u); xa[i][1] = xw + ww_lds_off(r, 2 * u + 1);
        if (i < nxu && r < p.xr) {
          xq[i][0] = ww_lds_read16(xa[i][0]); xq[i][1] = ww_lds_read16(xa[i][1]);
          if constexpr (MUL) {
            xm[i][0] = ww_lds_read16(xa[i][0] + (p.xr << 7)); xm[i][1] = ww_lds_read16(xa[i][1] + (p.xr << 7));
          }
        }
      }
      ww_lds_wait();
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = r0 + 128 * i;
        if constexpr (NE > 0)
          ev[i] += (d[i][0].x * e0.x + d[i][0].y * e0.y) + (d[i][0].z * e0.z + d[i][0].w * e0.w) +
                   (d[i][1].x * e1.x + d[i][1].y * e1.y) + (d[i][1].z * e1.z + d[i][1].w * e1.w);
        ww_u32x4 h4, l4;
        ww_split8u(d[i][0], d[i][1], h4, l4);
        ww_lds_write16(da[i][0], h4); ww_lds_write16(da[i][1], l4);
        if (i < nxu && r < p.xr) {
          if constexpr (MUL) { xq[i][0] *= xm[i][0]; xq[i][1] *= xm[i][1]; }
          ww_split8u(xq[i][0], xq[i][1], h4, l4);
          ww_lds_write16(xa[i][0], h4); ww_lds_write16(xa[i][1], l4);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    // ---- multiply: fragments are read as they stand (hi = first piece, lo = second piece of the lane's unit)
    if (!(p.exp & 2))
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      ww_f32x4 araw[MT][2], braw[KT][2];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const char* ap = dsl + a_off + ((i * 32) << 7);
        araw[i][0] = ww_lds_read16(ap + pc[ks][0]);
        araw[i][1] = ww_lds_read16(ap + pc[ks][1]);
      }
#pragma unroll
      for (int j = 0; j < KT; ++j) {
        const char* bp = xsl + x_off + ((j * 32) << 7);
        braw[j][0] = ww_lds_read16(bp + pc[ks][0]);
        braw[j][1] = ww_lds_read16(bp + pc[ks][1]);
      }
      ww_lds_wait();
      issue_pos(5 * ks);
#pragma unroll
      for (int j = 0; j < KT; ++j) {
        // (blocks beyond K hold zeros: their products are skipped — wave-uniform — but NOT the LDS-DMA piece issued behind them:
        // a `break` here once left the later stages of a workgroup without part of their operands whenever a wave had a dead
        // k-row block, i.e. for every n <= 96 once a workgroup had more than one stage; found by the many-tile small-n cases of
        // test_time_as_k_weight_gradient_kernels)
        if (FULL || wk * KT + j < k_blocks_here) {
          const ww_bf16x8 bh = __builtin_bit_cast(ww_bf16x8, braw[j][0]), bl = __builtin_bit_cast(ww_bf16x8, braw[j][1]);
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            if (!FULL && wm * MT + i >= m_blocks) break;   // wave-uniform
            const ww_bf16x8 ah = __builtin_bit_cast(ww_bf16x8, araw[i][0]), al = __builtin_bit_cast(ww_bf16x8, araw[i][1]);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i][j], 0, 0, 0);
          }
        }
        issue_pos(5 * ks + 1 + j);
      }
      if constexpr (KT < 3) issue_pos(5 * ks + 3);
      issue_pos(5 * ks + 4);
    }
    dslot = dslot == WW_ND - 1 ? 0 : dslot + 1;
    xslot ^= 1;
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
    // thread t holds the sums of unit t&3 of the dy rows t>>2 and 128 + (t>>2): the four units of a row sit in adjacent lanes
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float sv = ev[i];
      sv += __shfl_xor(sv, 1, 64);
      sv += __shfl_xor(sv, 2, 64);
      if ((tid & 3) == 0) p.slab_extra[((long long)blockIdx.x * WW_MROWS + (tid >> 2) + 128 * i) * 2] = sv;
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
  const long long at = (long long)m * sm + (long long)c * sc + off;
  if (p.epi_p) {                                          // (uniform) RandomLayer: Hadamard product with the class-side map, scaled
    float sd = 0.f;
    for (int j = 0; j < p.epi_ncls; ++j) sd += p.epi_p[m * p.epi_ncls + j] * p.epi_r1[(long long)j * sm + c];
    if (p.epi_raw) p.epi_raw[at] = v;
    v = v * p.epi_scale * sd;
  }
  w[at] = v;
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
static int ww_geometry(int kind, int B, int L, int n, int h, int last, int n_sets, WwParams* p) {
  const int KT = kind == 0 ? 3 : 2;
  p->M = kind == 0 ? 2 * n : (last ? n : 2 * n);
  p->K = kind == 0 ? 3 * n + h : n;
  const int rem = p->K & 31;
  p->n_extra = (kind == 0 && rem == 1) ? 1 : 0;                    // one k-row past a multiple of 32: a VALU row
  p->K_main = p->n_extra ? p->K - rem : p->K;
  p->xr = 32 * 2 * KT;
  const int kb = (p->K_main + 31) / 32;
  p->n_groups = (kb + 2 * KT - 1) / (2 * KT);
  p->mul = kind == 1;
  p->RX = p->xr * (1 + p->mul) + (p->n_extra ? 8 : 0);
  p->Kcols = p->n_groups * p->xr;
  p->B = B; p->L = L;
  p->tiles_per_seq = L / WW_TT;
  p->n_tiles = B * p->tiles_per_seq;
  const int cus = fst_cu_count() > 0 ? fst_cu_count() : 256;
  int ks = cus / p->n_groups;
  if (ks > p->n_tiles * n_sets) ks = p->n_tiles * n_sets;
  if (ks < n_sets) ks = n_sets;                                    // every operand set has a workgroup of its own
  p->ksplit = ks;
  p->n_sets = n_sets;
  return KT;
}

// 1: served; 2 (kind 0 only): served if 16 bytes in front of and behind `a` are readable (dilation 1-3: a tap's 16-byte pieces
// start up to 3 samples outside a row); 0: not served
extern "C" int fst_wn_wgrad_ok(int kind, int B, int L, int n, int h, int dil) {
  if (!(B > 0 && L > 0 && L % WW_TT == 0 && n > 0 && n < 128)) return 0;      // (32-sample stages)
  if (kind == 0) {
    if (!(h > 0 && h <= 32 && dil > 0 && (3 * n + h + 31) / 32 <= 18)) return 0;
    return dil % 4 == 0 ? 1 : (dil < 4 ? 2 : 0);
  }
  return kind == 1;
}

extern "C" int64_t fst_wn_wgrad_workspace_floats(int kind, int B, int L, int n, int h, int last) {
  if (!fst_wn_wgrad_ok(kind, B, L, n, h, 4)) return -1;                        // (the workspace does not depend on the dilation)
  WwParams p;
  ww_geometry(kind, B, L, n, h, last, WW_MAX_SETS, &p);                        // (nor — above n_sets workgroups — on the set count)
  return (int64_t)p.ksplit * WW_MROWS * (p.Kcols + 2);
}

static int ww_launch(WwParams& p, int KT, void* stream, bool reduce = true) {
  static const int exp_env = getenv("FST_WW_EXP") ? atoi(getenv("FST_WW_EXP")) : 0;
  p.exp = exp_env;
  const size_t lds = (size_t)WW_ND * WW_MROWS * 128 + (size_t)WW_NX * p.RX * 128;
  FST_REQUIRE(lds <= 160 * 1024 && (p.RX >> 3) <= 8 * WW_MAX_NX, "fst_wn_wgrad: %d staged k-rows per stage do not fit (LDS %zu B)", p.RX, lds);
  const bool full = (p.M + 31) / 32 == 8 && ((p.K_main + 31) / 32) % (2 * KT) == 0;
  void (*fn)(WwParams);
  if (KT == 3) {
    if (p.n_extra) fn = full ? wn_wgrad_kernel<2, 3, true, false, 1> : wn_wgrad_kernel<2, 3, false, false, 1>;
    else fn = full ? wn_wgrad_kernel<2, 3, true, false, 0> : wn_wgrad_kernel<2, 3, false, false, 0>;
  } else if (p.mul) {
    fn = full ? wn_wgrad_kernel<2, 2, true, true, 0> : wn_wgrad_kernel<2, 2, false, true, 0>;
  } else {
    fn = full ? wn_wgrad_kernel<2, 2, true, false, 0> : wn_wgrad_kernel<2, 2, false, false, 0>;
  }
  if (int rc = fst_allow_full_lds((const void*)fn, "fst_wn_wgrad")) return rc;
  hipLaunchKernelGGL(fn, dim3((unsigned)p.ksplit, (unsigned)p.n_groups), dim3(512), lds, (hipStream_t)stream, p);
  FST_LAUNCH_CHECK();
  if (!reduce) return 0;
  hipLaunchKernelGGL(wn_wgrad_reduce_kernel, dim3((unsigned)((p.Kcols + 255) / 256), (unsigned)p.M), dim3(256), 0, (hipStream_t)stream, p);
  FST_LAUNCH_CHECK();
  return 0;
}

extern "C" int fst_wn_wgrad_in(const float* const* dg, const float* const* a, const float* const* u0, int n_sets, int64_t u0_bs,
                               float* dw_in, float* dw_cond, float* workspace, int64_t workspace_floats, int B, int L, int n, int h,
                               int dil, int a_slack, int64_t numel_a, void* stream) {
  FST_REQUIRE(dg && a && u0 && dw_in && dw_cond && workspace, "fst_wn_wgrad_in: null operand");
  FST_REQUIRE(n_sets >= 1 && n_sets <= WW_MAX_SETS, "fst_wn_wgrad_in: %d operand sets (1..%d)", n_sets, WW_MAX_SETS);
  const int served = fst_wn_wgrad_ok(0, B, L, n, h, dil);
  FST_REQUIRE(served == 1 || (served == 2 && a_slack), "fst_wn_wgrad_in: unsupported shape B=%d L=%d n=%d h=%d dil=%d (needs L %% 32 == 0, "
              "n < 128, h <= 32, and dil %% 4 == 0 or — with 16 readable bytes either side of a — dil < 4)", B, L, n, h, dil);
  FST_REQUIRE((long long)B * n * L == (long long)numel_a, "fst_wn_wgrad_in: B*n*L does not match the element count %lld of a", (long long)numel_a);
  FST_REQUIRE(B == 1 || u0_bs >= (int64_t)h * L, "fst_wn_wgrad_in: u0 batch stride %lld < h*L", (long long)u0_bs);
  FST_REQUIRE(u0_bs % 4 == 0 && ww_al16(workspace), "fst_wn_wgrad_in: operands must be 16-byte aligned");
  WwParams p = {};
  const int KT = ww_geometry(0, B, L, n, h, 0, n_sets, &p);
  FST_REQUIRE(workspace_floats >= (int64_t)p.ksplit * WW_MROWS * (p.Kcols + 2), "fst_wn_wgrad_in: workspace of %lld floats is too small",
              (long long)workspace_floats);
  p.n_dy = 1;
  p.dy[0] = {{}, (long long)2 * n * L, 0, 2 * n, 0, 0, 0, 0, 0};
  p.n_x = 4;
  for (int tap = 0; tap < 3; ++tap) p.x[tap] = {{}, (long long)n * L, 0, n, (tap - 1) * dil, 0, tap, 3, 3 * n};
  p.x[3] = {{}, (long long)u0_bs, 0, h, 0, 1, 0, 1, h};
  for (int s = 0; s < n_sets; ++s) {
    FST_REQUIRE(dg[s] && a[s] && u0[s], "fst_wn_wgrad_in: null operand in set %d", s);
    FST_REQUIRE(ww_al16(dg[s]) && ww_al16(a[s]) && ww_al16(u0[s]), "fst_wn_wgrad_in: operands must be 16-byte aligned");
    p.dy[0].ptr[s] = dg[s];
    for (int tap = 0; tap < 3; ++tap) p.x[tap].ptr[s] = a[s];
    p.x[3].ptr[s] = u0[s];
  }
  p.misaligned = dil % 4 != 0;
  p.slab = workspace;
  p.slab_extra = workspace + (long long)p.ksplit * WW_MROWS * p.Kcols;
  p.w[0] = dw_in; p.w[1] = dw_cond;
  return ww_launch(p, KT, stream);
}

extern "C" int fst_wn_wgrad_rs(const float* const* d_a, const float* const* d_out, const float* const* ts, int n_sets, float* dw_rs,
                               float* workspace, int64_t workspace_floats, int last, int B, int L, int n, int64_t numel_a,
                               void* stream) {
  FST_REQUIRE(d_out && ts && dw_rs && workspace && (last || d_a), "fst_wn_wgrad_rs: null operand");
  FST_REQUIRE(n_sets >= 1 && n_sets <= WW_MAX_SETS, "fst_wn_wgrad_rs: %d operand sets (1..%d)", n_sets, WW_MAX_SETS);
  FST_REQUIRE(fst_wn_wgrad_ok(1, B, L, n, 0, 4), "fst_wn_wgrad_rs: unsupported shape B=%d L=%d n=%d (needs L %% 32 == 0, n < 128)", B, L, n);
  FST_REQUIRE((long long)B * n * L == (long long)numel_a, "fst_wn_wgrad_rs: B*n*L does not match the element count %lld", (long long)numel_a);
  FST_REQUIRE(ww_al16(workspace), "fst_wn_wgrad_rs: operands must be 16-byte aligned");
  WwParams p = {};
  const int KT = ww_geometry(1, B, L, n, 0, last, n_sets, &p);
  FST_REQUIRE(workspace_floats >= (int64_t)p.ksplit * WW_MROWS * (p.Kcols + 2), "fst_wn_wgrad_rs: workspace of %lld floats is too small",
              (long long)workspace_floats);
  p.n_dy = last ? 1 : 2;
  p.dy[0] = {{}, (long long)n * L, 0, n, 0, 0, 0, 0, 0};
  p.dy[1] = {{}, (long long)n * L, 0, n, 0, 0, 0, 0, 0};
  p.n_x = 1;
  p.x[0] = {{}, (long long)2 * n * L, (long long)n * L, n, 0, 0, 0, 1, n};   // t rows; the s rows n·L floats further
  for (int s = 0; s < n_sets; ++s) {
    FST_REQUIRE(d_out[s] && ts[s] && (last || d_a[s]), "fst_wn_wgrad_rs: null operand in set %d", s);
    FST_REQUIRE(ww_al16(d_out[s]) && ww_al16(ts[s]) && (last || ww_al16(d_a[s])), "fst_wn_wgrad_rs: operands must be 16-byte aligned");
    if (last) p.dy[0].ptr[s] = d_out[s];
    else { p.dy[0].ptr[s] = d_a[s]; p.dy[1].ptr[s] = d_out[s]; }
    p.x[0].ptr[s] = ts[s];
  }
  p.slab = workspace;
  p.slab_extra = workspace + (long long)p.ksplit * WW_MROWS * p.Kcols;
  p.w[0] = dw_rs; p.w[1] = nullptr;
  return ww_launch(p, KT, stream);
}

// ------------------------------------------------------------------------------------------------ C = A·Bᵀ on the same kernel
// C[m][n] = Σ_k A[m][k]·Bm[n][k]: both operands row-major with the reduction index contiguous — exactly the weight-gradient
// product above with A's rows as the output rows, Bm's rows as the k-rows and k as "time" (one sequence of K samples).
// RandomLayer's feature-side GEMM (C_DAN.py:21: [B, 25 600] × the fixed [25 600, 1024] Gaussian, kept transposed) and its data
// gradient ([B, 1024] × the matrix as it is) are both of this form; the 105 MB matrix is read once from HBM by the LDS-DMA ring,
// the K split leaves partial slabs that the reduce kernel adds in a fixed order (no atomics) while applying the layer's epilogue.
extern "C" int64_t fst_nt_gemm_workspace_floats(int M, int N, int K) {
  if (!(M > 0 && M <= WW_MROWS && N > 0 && K > 0 && K % WW_TT == 0)) return -1;
  const int n_groups = ((N + 31) / 32 + 3) / 4;
  const int cus = fst_cu_count() > 0 ? fst_cu_count() : 256;
  int ks = cus / n_groups;
  if (ks > K / WW_TT) ks = K / WW_TT;
  if (ks < 1) ks = 1;
  return (int64_t)ks * WW_MROWS * (n_groups * 128 + 2);
}

extern "C" int fst_nt_gemm(const float* A, const float* Bm, float* C, float* workspace, int64_t workspace_floats, int M, int N, int K,
                           const float* epi_p, const float* epi_r1, int epi_ncls, float epi_scale, float* epi_raw, void* stream) {
  FST_REQUIRE(A && Bm && C && workspace, "fst_nt_gemm: null operand");
  FST_REQUIRE(M > 0 && M <= WW_MROWS && N > 0 && K > 0 && K % WW_TT == 0, "fst_nt_gemm: M=%d N=%d K=%d (needs M <= 256, K %% 32 == 0)", M, N, K);
  FST_REQUIRE((long long)N * K < (1LL << 31) * 4 && (long long)M * N < (1LL << 31), "fst_nt_gemm: operand too large");
  FST_REQUIRE(ww_al16(A) && ww_al16(Bm) && ww_al16(workspace), "fst_nt_gemm: operands must be 16-byte aligned");
  FST_REQUIRE((epi_p == nullptr) == (epi_r1 == nullptr) && (epi_p == nullptr || epi_ncls > 0) && (epi_p != nullptr || epi_raw == nullptr),
              "fst_nt_gemm: the epilogue takes p [M][ncls] and r1 [ncls][N] together (and the raw product only with them)");
  WwParams p = {};
  p.M = M; p.K = N; p.K_main = N; p.n_extra = 0;
  p.xr = 128;
  p.n_groups = ((N + 31) / 32 + 3) / 4;
  p.mul = 0;
  p.RX = p.xr;
  p.Kcols = p.n_groups * p.xr;
  p.B = 1; p.L = K;
  p.tiles_per_seq = K / WW_TT;
  p.n_tiles = p.tiles_per_seq;
  const int cus = fst_cu_count() > 0 ? fst_cu_count() : 256;
  int ks = cus / p.n_groups;
  if (ks > p.n_tiles) ks = p.n_tiles;
  if (ks < 1) ks = 1;
  p.ksplit = ks;
  p.n_sets = 1;
  FST_REQUIRE(workspace_floats >= (int64_t)p.ksplit * WW_MROWS * (p.Kcols + 2), "fst_nt_gemm: workspace of %lld floats is too small",
              (long long)workspace_floats);
  p.n_dy = 1;
  p.dy[0] = {{A, nullptr, nullptr}, 0, 0, M, 0, 0, 0, 0, 0};
  p.n_x = 1;
  p.x[0] = {{Bm, nullptr, nullptr}, 0, 0, N, 0, 0, 0, 1, N};              // C[m][n] at w[0][m·N + n]
  p.slab = workspace;
  p.slab_extra = workspace + (long long)p.ksplit * WW_MROWS * p.Kcols;
  p.w[0] = C; p.w[1] = nullptr;
  p.epi_p = epi_p; p.epi_r1 = epi_r1; p.epi_ncls = epi_ncls; p.epi_scale = epi_scale; p.epi_raw = epi_raw;
  // one K slice of all 256 rows whose slab [256][n_groups·128] IS the output [M][N] (the data gradient of the random layer):
  // the workgroups store straight into C, no second pass
  const bool direct = p.ksplit == 1 && M == WW_MROWS && p.Kcols == N && !epi_p;
  if (direct) p.slab = C;
  return ww_launch(p, 2, stream, !direct);
}

// ------------------------------------------------------------------------------------------------ few-tap conv weight gradient
// dW[m][c][τ] = Σ_{b,t} dy[b][m][t]·x[b][c][t + τ·dil − pad_left] for a conv with at most four taps (a tap = one k-row segment with
// its own shift) — the dense (Q1) gradient of the shared omni-scale block's last layer (225 → 50 channels, two taps), which the
// generic item-table kernel ran at 29 TFLOP/s.  Tap shifts that are not multiples of 4 samples must be within ±3 and need 16
// readable bytes either side of x (x_slack), as fst_wn_wgrad_in.
extern "C" int fst_tap_wgrad_ok(int B, int L, int M, int C, int ntaps, int dil, int pad_left) {
  if (!(B > 0 && L > 0 && L % WW_TT == 0 && M > 0 && M <= WW_MROWS && C > 0 && ntaps >= 1 && ntaps <= 4 && dil > 0)) return 0;
  if (((long long)ntaps * C + 31) / 32 > 18) return 0;                       // (k-row blocks of all groups: LDS geometry as above)
  int need_slack = 0;
  for (int tap = 0; tap < ntaps; ++tap) {
    const int sh = tap * dil - pad_left;
    if (sh % 4 != 0) {
      if (sh <= -4 || sh >= 4) return 0;
      need_slack = 1;
    }
  }
  return need_slack ? 2 : 1;
}

static void tap_geometry(int B, int L, int M, int C, int ntaps, WwParams* p) {
  p->M = M; p->K = ntaps * C; p->K_main = p->K; p->n_extra = 0;
  p->xr = 192;
  p->n_groups = ((p->K + 31) / 32 + 5) / 6;
  p->mul = 0;
  p->RX = p->xr;
  p->Kcols = p->n_groups * p->xr;
  p->B = B; p->L = L;
  p->tiles_per_seq = L / WW_TT;
  p->n_tiles = B * p->tiles_per_seq;
  const int cus = fst_cu_count() > 0 ? fst_cu_count() : 256;
  int ks = cus / p->n_groups;
  if (ks > p->n_tiles) ks = p->n_tiles;
  if (ks < 1) ks = 1;
  p->ksplit = ks;
  p->n_sets = 1;
}

extern "C" int64_t fst_tap_wgrad_workspace_floats(int B, int L, int M, int C, int ntaps) {
  if (!fst_tap_wgrad_ok(B, L, M, C, ntaps, 4, 0)) return -1;
  WwParams p = {};
  tap_geometry(B, L, M, C, ntaps, &p);
  return (int64_t)p.ksplit * WW_MROWS * (p.Kcols + 2);
}

extern "C" int fst_tap_wgrad(const float* dy, const float* x, float* dw, float* workspace, int64_t workspace_floats, int B, int L, int M,
                             int C, int ntaps, int dil, int pad_left, int x_slack, int64_t numel_dy, int64_t numel_x, void* stream) {
  FST_REQUIRE(dy && x && dw && workspace, "fst_tap_wgrad: null operand");
  const int served = fst_tap_wgrad_ok(B, L, M, C, ntaps, dil, pad_left);
  FST_REQUIRE(served == 1 || (served == 2 && x_slack), "fst_tap_wgrad: unsupported shape B=%d L=%d M=%d C=%d ntaps=%d dil=%d pad_left=%d "
              "(needs L %% 32 == 0, M <= 256, at most 4 taps, tap shifts multiples of 4 or — with 16 readable bytes either side of x — "
              "within 3 samples)", B, L, M, C, ntaps, dil, pad_left);
  FST_REQUIRE((long long)B * M * L == (long long)numel_dy && (long long)B * C * L == (long long)numel_x,
              "fst_tap_wgrad: B*M*L / B*C*L do not match the element counts %lld / %lld", (long long)numel_dy, (long long)numel_x);
  FST_REQUIRE(ww_al16(dy) && ww_al16(x) && ww_al16(workspace), "fst_tap_wgrad: operands must be 16-byte aligned");
  WwParams p = {};
  tap_geometry(B, L, M, C, ntaps, &p);
  FST_REQUIRE(workspace_floats >= (int64_t)p.ksplit * WW_MROWS * (p.Kcols + 2), "fst_tap_wgrad: workspace of %lld floats is too small",
              (long long)workspace_floats);
  p.n_dy = 1;
  p.dy[0] = {{dy, nullptr, nullptr}, (long long)M * L, 0, M, 0, 0, 0, 0, 0};
  p.n_x = ntaps;
  p.misaligned = 0;
  for (int tap = 0; tap < ntaps; ++tap) {
    const int sh = tap * dil - pad_left;
    p.x[tap] = {{x, nullptr, nullptr}, (long long)C * L, 0, C, sh, 0, tap, ntaps, C * ntaps};    // dW[m][c][tap] at m·C·ntaps + c·ntaps + tap
    if (sh % 4 != 0) p.misaligned = 1;
  }
  p.slab = workspace;
  p.slab_extra = workspace + (long long)p.ksplit * WW_MROWS * p.Kcols;
  p.w[0] = dw; p.w[1] = nullptr;
  return ww_launch(p, 3, stream);
}

// ================================================================================================ dense many-tap weight gradient
// dW[m][c][k] = Σ_{b,t} dy[b][m][t]·x[b][c][t + k − pad]  for EVERY tap k < K <= 96 of an omni-scale layer (the Q1 gradient of the
// shared OS_block, /root/reference/OS_CNN/OS_CNN.py:67-71 + train_and_test.py:685-690: the reference convolves the dense Kmax kernel,
// so autograd's weight gradient is dense and GradNorm's norms run over it).  Time is the MFMA reduction index as above, but the
// K k-rows of a channel are the SAME row of x at K consecutive shifts: one 160-sample window of the channel is staged per stage
// (one LDS-DMA instruction, always 16-byte aligned) and the split pass writes it out EIGHT times, as bf16 hi / lo copies shifted
// by 0..7 samples.  The B fragment of k-row "shift s" at samples t.. t+7 is then the 16-byte aligned unit ⌊o/8⌋ of copy o mod 8
// (o = s + t − window start): one ds_read_b128 per part, no per-shift staging — the item-table kernel stages every (tap, channel)
// row on its own and reaches 0.23 of the matrix peak on the 225 × 25 × 89 layer.
// A workgroup = all M rows × CW channels × 3 blocks of 32 shifts; its 8 waves = MP row pairs × CW = 8/MP channels, 2 × 3 tiles of
// 32 × 32 each: MP = 2 (128 rows, 4 channels) or 1 (up to 64 rows, 8 channels) — few-row layers (the first omni-scale layer: 25 rows)
// put their waves on channels instead of on row blocks that do not exist; more than 128 rows are two row halves (blockIdx.z), each
// streaming only its own dy rows: dy is then staged by ⌈C/4⌉ channel groups instead of ⌈C/2⌉ (225 × 25 × 89: 551 → … µs).
#define TZ_CPB 288                                         // bytes per shifted copy: 16 units of 16 B + 32 (16-lane groups then hit all banks)
#define TZ_XRAW 1024                                       // bytes per staged raw window (40 of 64 pieces used)
#define TZ_ND 4                                            // ring slots (dy rows and raw windows): multiply c, split c + 1, c + 2 and c + 3 in flight

struct TzParams {
  const float* dy;      // [B][M][L]
  const float* x;       // [B][C][L]
  float* slab;          // [ksplit][256][Kcols]
  float* dw;            // [M][C][K]
  int B, L, M, C, K, P4;        // P4: the window of stage t0 starts at sample t0 − P4 (pad rounded up to 4, plus 4)
  int off0;                     // P4 − pad: window offset of (shift 0, sample t0)
  int n_groups, ksplit, tiles_per_seq, n_tiles, Kcols;
  int m_halves;                 // 1, or 2: workgroup z takes the rows [128·z, min(M, 128·z + 128))
};

template <int N>
__device__ __forceinline__ void tz_wait_at_most() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

#ifndef TZ_EXP
#define TZ_EXP 0   // diagnostic builds (tools/build_tz_exp.sh; timing only, wrong results): 1 no MFMAs, 2 no split pass, 4 no LDS-DMA,
#endif             // 8 no fragment reads, 16 no slab stores, 32 no stages

#ifdef TZ_STAMPS
// Diagnostic build only (tools/build_tz_exp.sh stamps): per-phase s_memtime sums of tz_wgrad_kernel, lane 0 of every wave.
__device__ unsigned long long tz_stamps[12];
__device__ __forceinline__ unsigned long long tz_now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
extern "C" int fst_debug_tz_stamps(unsigned long long* out_host, int reset) {
  if (out_host) hipMemcpyFromSymbol(out_host, HIP_SYMBOL(tz_stamps), sizeof(unsigned long long) * 12);
  if (reset) { unsigned long long z[12] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(tz_stamps), z, sizeof(z)); }
  return 0;
}
#define TZ_T(var) const unsigned long long var = tz_now()
#define TZ_ACC(slot, a, b) tz_sum[slot] += (b) - (a)
#define TZ_SUMS unsigned long long tz_sum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define TZ_FLUSH \
  if (lane == 0) for (int i_ = 0; i_ < 12; ++i_) atomicAdd(&tz_stamps[i_], tz_sum[i_])
#else
#define TZ_T(var)
#define TZ_ACC(slot, a, b)
#define TZ_SUMS
#define TZ_FLUSH
#endif

// LDS accesses by byte address with the constant part of the address as the instruction's immediate offset: the fragment and
// split-pass reads of a stage share a handful of per-lane address registers (see the instruction budget below)
template <int OFF>
__device__ __forceinline__ ww_f32x4 tz_read16(unsigned addr) {
  ww_f32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
template <int OFF>
__device__ __forceinline__ float tz_read4(unsigned addr) {
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
template <int OFF>
__device__ __forceinline__ void tz_write16(unsigned addr, const ww_u32x4& v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
// one LDS-DMA piece: the 64 lanes' 16 bytes land at LDS byte address m0_addr + 16·lane (M0 saved and restored: see ww_dma16)
__device__ __forceinline__ void tz_dma16(const char* gsrc, unsigned m0_addr) {
  unsigned saved_m0;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(saved_m0) : "v"(gsrc), "s"(m0_addr) : "memory");
}

// INSTRUCTION BUDGET.  With two waves per SIMD a wave issues at most one instruction every four cycles and a taken branch costs
// tens of cycles: a stage of 36 MFMAs (2 × 1152 cycles of matrix pipe per SIMD) leaves room for ≈300 other instructions per wave.
// The first form of this loop carried 483 (49 branches, 281 scalar) and ran at 5.9 k cycles per stage — an EMPTY loop with its
// waits, barrier, index divisions and uniform tests alone took 1.4 k (cost removal: profiles/r04_tz_*).  Hence: diagnostics are
// compile-time (TZ_EXP), the stage position advances by additions (no divisions), every wave issues the same number of pieces
// per stage (rows beyond the row half re-fetch row 0 — finite values in LDS rows nobody stores — so one counted wait fits all),
// stages past the end are "virtual" (they re-fetch and re-split the last stage instead of being tested away), FULL drops the
// block tests around the MFMAs, and LDS addresses are a few per-lane registers plus immediates.
template <int MP, bool FULL>
__global__ __launch_bounds__(512, 2) void tz_wgrad_kernel(TzParams p) {
  constexpr int CW = 8 / MP;                               // channels per group = waves per row pair
  constexpr int NDW = MP == 2 ? 2 : 1;                     // dy LDS-DMA instructions per wave and stage (8 rows each: 128 / 64 rows)
  constexpr int DSLOT = (MP == 2 ? 128 : 64) * 128;        // a stage's dy rows of ONE row half
  constexpr int XSLOT = CW * TZ_XRAW;                      // a stage's raw windows
  constexpr int LO = CW * 8 * TZ_CPB, COPIES = 2 * LO;     // one copies buffer: [hi | lo][CW][8 copies][TZ_CPB]
  constexpr int X0 = TZ_ND * DSLOT, C0 = X0 + TZ_ND * XSLOT;
  constexpr int XU = (128 * CW) / 512;                     // x copy units per thread (1 or 2): every thread has work
  extern __shared__ __attribute__((aligned(16))) char tz_lds[];
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)tz_lds);
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave_s / CW, wk = wave_s % CW;            // row pair, channel of the group
  const int g = blockIdx.y, L = p.L;
  const char* const zero16 = reinterpret_cast<const char*>(ww_zero16);

  // ---- LDS-DMA sources.  dy: instruction i = wave + 8k fills LDS rows 8i .. 8i+7 of the slot (lane → row, piece: ww_dma_row)
  const int m_base = blockIdx.z * 128;
  const int M_here = p.m_halves > 1 ? min(128, p.M - m_base) : p.M;
  const float* dsrc[NDW];
#pragma unroll
  for (int k = 0; k < NDW; ++k) {
    const int r = ww_dma_row(wave_s + 8 * k, lane), q = (lane ^ r) & 7;
    dsrc[k] = p.dy + ((long long)(m_base + (r < M_here ? r : 0)) * L + 4 * q);
  }
  // the window of channel CW·g + wave (waves < CW): piece `lane` (< 40) of the 160 samples from t0 − P4
  const bool has_x = wave_s < CW;                          // wave-uniform
  const int my_ch = CW * g + wave_s;
  const bool x_lane = has_x && my_ch < p.C && lane < 40;
  const float* const xrow = p.x + ((long long)(x_lane ? my_ch : 0) * L + 4 * lane - p.P4);
  const int xt = 4 * lane - p.P4;                          // the piece's first sample relative to t0
  const long long dy_bs = (long long)p.M * L, x_bs = (long long)p.C * L;
  __builtin_amdgcn_s_waitcnt(0x0F70);                      // (see wn_wgrad_kernel: per-lane selections of kernel arguments)

  // stage position: batch element and first sample as running offsets into dy / x (advanced by additions)
  struct Pos { int t0; long long doff, xoff; };
  auto advance = [&](Pos& q) {
    q.t0 += WW_TT; q.doff += WW_TT; q.xoff += WW_TT;
    if (q.t0 == L) { q.t0 = 0; q.doff += dy_bs - L; q.xoff += x_bs - L; }
  };
  auto issue_stage = [&](const Pos& q, int slot) {         // NDW dy pieces, then (waves < CW) the window piece
    if (TZ_EXP & 4) return;
#pragma unroll
    for (int k = 0; k < NDW; ++k)
      tz_dma16(reinterpret_cast<const char*>(dsrc[k] + q.doff), lds0 + slot * DSLOT + (wave_s + 8 * k) * 1024);
    if (has_x) {
      const int t = q.t0 + xt;                             // (multiples of 4: a piece is inside the sequence or outside it)
      const char* src = (x_lane && t >= 0 && t < L) ? reinterpret_cast<const char*>(xrow + q.xoff) : zero16;
      tz_dma16(src, lds0 + X0 + slot * XSLOT + wave_s * TZ_XRAW);
    }
  };
  auto wait_keep_one = [&]() {                             // all but the youngest stage's pieces of this wave have landed
    if (TZ_EXP & 4) return;
    if (has_x) tz_wait_at_most<NDW + 1>(); else tz_wait_at_most<NDW>();
  };

  f32x16 acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int tile_begin = (int)(((long long)blockIdx.x * p.n_tiles) / p.ksplit);
  const int tile_end = (int)(((long long)(blockIdx.x + 1) * p.n_tiles) / p.ksplit);
  const int m_blocks = (M_here + 31) >> 5;
  const int k_blocks = (p.K + 31) >> 5;                    // live 32-shift blocks (<= 3)
  const bool ch_live = CW * g + wk < p.C;

  // ---- per-lane LDS offsets.  A fragments as in wn_wgrad_kernel: row block·32 + l31 (LDS row l31 ^ ((l31 >> 3) & 1)), unit
  // 2·ks + half: hi = first piece, lo = second; the block index and the slot are immediates / one scalar
  const int row_l = (l31 ^ ((l31 >> 3) & 1)) << 7, sw = l31 & 7;
  unsigned a_lane[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int j = 0; j < 2; ++j) a_lane[ks][j] = ((wm * 2 * 32) << 7) + row_l + ((((4 * ks + 2 * half + j) ^ sw) & 7) << 4);
  // B fragments: shift s = 32·sb + l31 at samples 16·ks + 8·half + 0..7 → window offset o = off0 + s + 16·ks + 8·half: copy o & 7
  // (the same for every ks, sb), unit o >> 3
  const int o0 = p.off0 + l31 + 8 * half;
  const unsigned b_lane = wk * (8 * TZ_CPB) + (o0 & 7) * TZ_CPB + ((o0 >> 3) << 4);
  // split pass.  dy: thread t owns unit t & 3 (8 samples) of row t >> 2: pieces 2u, 2u + 1 (MP = 1: the slot has 64 rows, waves
  // 0-3 only).  x: unit id = t + 512j → channel id >> 7, copy (id >> 4) & 7, 16-byte unit id & 15: window samples 8q + r .. + 7
  const int su = tid & 3, sr = tid >> 2;
  const bool dy_thread = MP == 2 || wave_s < 4;            // wave-uniform
  const unsigned sd_lane0 = ww_lds_off(sr, 2 * su), sd_lane1 = ww_lds_off(sr, 2 * su + 1);
  unsigned sx_src[XU], sx_dst[XU];
#pragma unroll
  for (int j = 0; j < XU; ++j) {
    const int id = tid + 512 * j, xc = id >> 7, xr = (id >> 4) & 7, xq = id & 15;
    sx_src[j] = xc * TZ_XRAW + (8 * xq + xr) * 4;
    sx_dst[j] = xc * (8 * TZ_CPB) + xr * TZ_CPB + (xq << 4);
  }

  // Software pipeline over stages: while stage c multiplies, stage c + 1 — landed in its ring slot during stage c − 1 — goes through
  // the split pass (dy rows in place, the x windows into the OTHER copies buffer) and the pieces of stage c + 3 are issued, in chunks
  // between the MFMA groups.  One barrier per stage.
  const int n_st = (TZ_EXP & 32) ? 0 : tile_end - tile_begin;
  ww_f32x4 sd0, sd1;
  float sxv[XU][8];
  auto split_load = [&](unsigned d_slot, unsigned x_slot) {
    if (TZ_EXP & 2) return;
    if (dy_thread) { sd0 = tz_read16<0>(d_slot + sd_lane0); sd1 = tz_read16<0>(d_slot + sd_lane1); }
#pragma unroll
    for (int j = 0; j < XU; ++j) {
      const unsigned a = x_slot + sx_src[j];
      sxv[j][0] = tz_read4<0>(a); sxv[j][1] = tz_read4<4>(a); sxv[j][2] = tz_read4<8>(a); sxv[j][3] = tz_read4<12>(a);
      sxv[j][4] = tz_read4<16>(a); sxv[j][5] = tz_read4<20>(a); sxv[j][6] = tz_read4<24>(a); sxv[j][7] = tz_read4<28>(a);
    }
  };
  auto split_dy_store = [&](unsigned d_slot) {
    if ((TZ_EXP & 2) || !dy_thread) return;
    ww_u32x4 h4, l4;
    ww_split8u(sd0, sd1, h4, l4);
    tz_write16<0>(d_slot + sd_lane0, h4); tz_write16<0>(d_slot + sd_lane1, l4);
  };
  auto split_x_store = [&](unsigned cbuf) {
    if (TZ_EXP & 2) return;
#pragma unroll
    for (int j = 0; j < XU; ++j) {
      ww_u32x4 h4, l4;
      unsigned hh, ll;
#pragma unroll
      for (int e = 0; e < 4; ++e) { ww_split_pair(sxv[j][2 * e], sxv[j][2 * e + 1], hh, ll); h4[e] = hh; l4[e] = ll; }
      tz_write16<0>(cbuf + sx_dst[j], h4);
      tz_write16<LO>(cbuf + sx_dst[j], l4);
    }
  };

  Pos pos;                                                 // position of the stage issued next
  {
    const int b = tile_begin / p.tiles_per_seq;
    pos.t0 = (tile_begin - b * p.tiles_per_seq) * WW_TT;
    pos.doff = (long long)b * dy_bs + pos.t0;
    pos.xoff = (long long)b * x_bs + pos.t0;
  }
  int issued = 0;                                          // stages issued so far; past n_st the last one is issued again ("virtual")
  auto issue_next = [&](int slot) {
    issue_stage(pos, slot);
    ++issued;
    if (issued < n_st) advance(pos);
  };
  // ONE barrier per stage, in its middle, and no burst of LDS reads behind it: the fragments of k-step 0 of stage c + 1 are read
  // right after that barrier, under the MFMAs of k-step 1 of stage c; those of k-step 1 at the top of the stage, under the MFMAs of
  // k-step 0.  (Stamps of the form with the barrier at the top: the issue of the 20 LDS reads behind it took 20 % of a wave's stage —
  // all eight waves read at once and nobody could multiply yet — the barrier 13 %, profiles/r04_tz_stamps.txt.)
  ww_f32x4 araw[2][2][2], braw[2][3][2];
  auto frag_reads = [&](auto ks_c, unsigned d_slot, unsigned cbuf) {
    constexpr int ks = decltype(ks_c)::value;
    if (TZ_EXP & 8) return;
    araw[ks][0][0] = tz_read16<0>(d_slot + a_lane[ks][0]);    araw[ks][0][1] = tz_read16<0>(d_slot + a_lane[ks][1]);
    araw[ks][1][0] = tz_read16<4096>(d_slot + a_lane[ks][0]); araw[ks][1][1] = tz_read16<4096>(d_slot + a_lane[ks][1]);
    const unsigned bb = cbuf + b_lane;
    braw[ks][0][0] = tz_read16<((2 * ks) << 4)>(bb);      braw[ks][0][1] = tz_read16<((2 * ks) << 4) + LO>(bb);
    braw[ks][1][0] = tz_read16<((2 * ks + 4) << 4)>(bb);  braw[ks][1][1] = tz_read16<((2 * ks + 4) << 4) + LO>(bb);
    braw[ks][2][0] = tz_read16<((2 * ks + 8) << 4)>(bb);  braw[ks][2][1] = tz_read16<((2 * ks + 8) << 4) + LO>(bb);
  };
  auto mfma_group = [&](int ks, int sb) {
    if (!(TZ_EXP & 1) && (FULL || (sb < k_blocks && ch_live))) {       // wave-uniform
      const ww_bf16x8 bh = __builtin_bit_cast(ww_bf16x8, braw[ks][sb][0]), bl = __builtin_bit_cast(ww_bf16x8, braw[ks][sb][1]);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (!FULL && wm * 2 + i >= m_blocks) break;        // wave-uniform
        const ww_bf16x8 ah = __builtin_bit_cast(ww_bf16x8, araw[ks][i][0]), al = __builtin_bit_cast(ww_bf16x8, araw[ks][i][1]);
        acc[i][sb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[i][sb], 0, 0, 0);
        acc[i][sb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i][sb], 0, 0, 0);
        acc[i][sb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i][sb], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  TZ_SUMS;
  TZ_T(tz_begin);
  if (n_st > 0) {
    issue_next(0);
    issue_next(1);
    wait_keep_one();                                       // stage 0 has landed
    __builtin_amdgcn_s_barrier();
    split_load(lds0, lds0 + X0);                           // stage 0 has nothing to hide under
    ww_lds_wait();
    split_dy_store(lds0); split_x_store(lds0 + C0);
    issue_next(2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wait_keep_one();                                       // stage 1 has landed
    __builtin_amdgcn_s_barrier();                          // split(0) of every wave is visible
    frag_reads(std::integral_constant<int, 0>{}, lds0, lds0 + C0);
  }
  int slot = 0, cb = 0;                                    // ring slot of stage c
  for (int c = 0; c < n_st; ++c) {
    const int s_next = (slot + 1) & (TZ_ND - 1), s_dma = (slot + 3) & (TZ_ND - 1);   // c + 3 goes where multiply(c − 1) read
    const unsigned d_cur = lds0 + slot * DSLOT, d_nxt = lds0 + s_next * DSLOT, x_nxt = lds0 + X0 + s_next * XSLOT;
    const unsigned c_cur = lds0 + C0 + cb * COPIES, c_nxt = lds0 + C0 + (cb ^ 1) * COPIES;
    TZ_T(ta);
    ww_lds_wait();                                         // the k-step-0 fragments (read under the previous stage's last MFMAs)
    TZ_T(tb);
    TZ_ACC(0, ta, tb);                                     // wait for the k-step-0 fragments
    frag_reads(std::integral_constant<int, 1>{}, d_cur, c_cur);
    split_load(d_nxt, x_nxt);                              // stage c + 1 landed before the previous barrier (virtual past the end)
    TZ_T(tc);
    TZ_ACC(2, tb, tc);                                     // issue of the k-step-1 fragment reads + split reads
    mfma_group(0, 0);
    TZ_T(td);
    TZ_ACC(4, tc, td);                                     // MFMA group (0, 0)
    ww_lds_wait();
    TZ_T(te);
    TZ_ACC(3, td, te);                                     // wait for those reads
    split_dy_store(d_nxt);
    __builtin_amdgcn_sched_barrier(0);
    TZ_T(tf);
    TZ_ACC(5, te, tf);                                     // dy split
    mfma_group(0, 1);
    TZ_T(tg);
    TZ_ACC(4, tf, tg);
    split_x_store(c_nxt);                                  // (its stores complete under the group below)
    __builtin_amdgcn_sched_barrier(0);
    TZ_T(th);
    TZ_ACC(7, tg, th);                                     // x split
    mfma_group(0, 2);
    TZ_T(ti);
    TZ_ACC(4, th, ti);
    issue_next(s_dma);                                     // this wave's pieces of stage c + 3
    TZ_T(tj);
    TZ_ACC(6, ti, tj);                                     // LDS-DMA issue
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's split stores of stage c + 1 are done ...
    wait_keep_one();                                       // ... and its pieces of stage c + 2 have landed (c + 3 may still fly)
    TZ_T(tk);
    TZ_ACC(8, tj, tk);                                     // waits in front of the barrier
    __builtin_amdgcn_s_barrier();                          // split(c + 1) is visible; everyone is past its reads of stage c
    TZ_T(tl);
    TZ_ACC(1, tk, tl);                                     // barrier
    frag_reads(std::integral_constant<int, 0>{}, d_nxt, c_nxt);     // k-step 0 of stage c + 1, under the MFMAs below
    mfma_group(1, 0);
    mfma_group(1, 1);
    mfma_group(1, 2);
    TZ_T(tm);
    TZ_ACC(9, tl, tm);                                     // fragment reads of the next stage + MFMA groups of k-step 1
    slot = s_next;
    cb ^= 1;
  }
  TZ_T(tz_end);
  TZ_ACC(10, tz_begin, tz_end);                            // whole loop
  TZ_FLUSH;
  // (virtual pieces may still be in flight: they target LDS only, and the wave's end waits for them)

  float* const slab = p.slab + (long long)blockIdx.x * WW_MROWS * p.Kcols;
  if (TZ_EXP & 16) return;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (wm * 2 + i >= m_blocks) break;
#pragma unroll
    for (int sb = 0; sb < 3; ++sb) {
      float* dst = slab + (long long)(m_base + (wm * 2 + i) * 32 + 4 * half) * p.Kcols + g * (CW * 96) + wk * 96 + sb * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[(long long)((r & 3) + 8 * (r >> 2)) * p.Kcols] = acc[i][sb][r];
    }
  }
}

// dw[m][c][k] = Σ_slabs slab[m][c·96 + k]: one thread per (m, c, k), four independent loads in flight
__global__ __launch_bounds__(256) void tz_reduce_kernel(TzParams p) {
  const int m = blockIdx.y;
  const int col = blockIdx.x * 256 + threadIdx.x;
  const int c = col / 96, k = col - c * 96;
  if (c >= p.C || k >= p.K) return;
  const float* q = p.slab + (long long)m * p.Kcols + col;
  const long long st = (long long)WW_MROWS * p.Kcols;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int sl = 0;
  for (; sl + 4 <= p.ksplit; sl += 4) {
    s0 += q[(long long)sl * st]; s1 += q[(long long)(sl + 1) * st]; s2 += q[(long long)(sl + 2) * st]; s3 += q[(long long)(sl + 3) * st];
  }
  for (; sl < p.ksplit; ++sl) s0 += q[(long long)sl * st];
  p.dw[((long long)m * p.C + c) * p.K + k] = (s0 + s1) + (s2 + s3);
}

static int tz_geometry(int B, int L, int M, int C, int K, int pad, TzParams* p) {
  const int MP = M <= 64 ? 1 : 2, TZ_CW = 8 / MP;
  p->m_halves = M > 128 ? 2 : 1;
  p->B = B; p->L = L; p->M = M; p->C = C; p->K = K;
  p->P4 = ((pad + 3) & ~3) + 4;
  p->off0 = p->P4 - pad;
  p->n_groups = (C + TZ_CW - 1) / TZ_CW;
  p->Kcols = p->n_groups * TZ_CW * 96;
  p->tiles_per_seq = L / WW_TT;
  p->n_tiles = B * p->tiles_per_seq;
  const int cus = fst_cu_count() > 0 ? fst_cu_count() : 256;
  int ks = cus / (p->n_groups * p->m_halves);
  if (ks > p->n_tiles) ks = p->n_tiles;
  if (ks < 1) ks = 1;
  p->ksplit = ks;
  return MP;
}

extern "C" int fst_dense_tap_wgrad_ok(int B, int L, int M, int C, int K, int pad_left) {
  return B > 0 && L > 0 && L % WW_TT == 0 && M > 0 && M <= WW_MROWS && C > 0 && K > 4 && K <= 96 && pad_left >= 0 && pad_left < K;
}

extern "C" int64_t fst_dense_tap_wgrad_workspace_floats(int B, int L, int M, int C, int K) {
  if (!fst_dense_tap_wgrad_ok(B, L, M, C, K, 0)) return -1;
  TzParams p = {};
  (void)tz_geometry(B, L, M, C, K, 0, &p);
  return (int64_t)p.ksplit * WW_MROWS * p.Kcols;
}

extern "C" int fst_dense_tap_wgrad(const float* dy, const float* x, float* dw, float* workspace, int64_t workspace_floats, int B, int L,
                                   int M, int C, int K, int pad_left, int64_t numel_dy, int64_t numel_x, void* stream) {
  FST_REQUIRE(dy && x && dw && workspace, "fst_dense_tap_wgrad: null operand");
  FST_REQUIRE(fst_dense_tap_wgrad_ok(B, L, M, C, K, pad_left), "fst_dense_tap_wgrad: unsupported shape B=%d L=%d M=%d C=%d K=%d pad_left=%d "
              "(needs L %% 32 == 0, M <= 256, 4 < K <= 96, 0 <= pad_left < K)", B, L, M, C, K, pad_left);
  FST_REQUIRE((long long)B * M * L == (long long)numel_dy && (long long)B * C * L == (long long)numel_x,
              "fst_dense_tap_wgrad: B*M*L / B*C*L do not match the element counts %lld / %lld", (long long)numel_dy, (long long)numel_x);
  FST_REQUIRE(ww_al16(dy) && ww_al16(x) && ww_al16(workspace), "fst_dense_tap_wgrad: operands must be 16-byte aligned");
  TzParams p = {};
  const int MP = tz_geometry(B, L, M, C, K, pad_left, &p);
  FST_REQUIRE(workspace_floats >= (int64_t)p.ksplit * WW_MROWS * p.Kcols, "fst_dense_tap_wgrad: workspace of %lld floats is too small",
              (long long)workspace_floats);
  p.dy = dy; p.x = x; p.dw = dw; p.slab = workspace;
  const int cw = 8 / MP;
  const size_t lds = (size_t)TZ_ND * (MP == 1 ? 64 : 128) * 128 + TZ_ND * cw * TZ_XRAW + 2 * (size_t)cw * 2 * 8 * TZ_CPB;   // dy ring, window ring, two copies buffers
  // FULL: every 32-row block of every row half and all three 32-shift blocks are live: no block tests around the MFMAs
  const int rows_block = MP == 2 ? 128 : 64, last_half = M - (p.m_halves - 1) * 128;
  const bool full = K > 64 && (p.m_halves > 1 ? last_half > 96 : M > rows_block - 32);
  void (*fn)(TzParams) = MP == 1 ? (full ? tz_wgrad_kernel<1, true> : tz_wgrad_kernel<1, false>)
                                 : (full ? tz_wgrad_kernel<2, true> : tz_wgrad_kernel<2, false>);
  if (int rc = fst_allow_full_lds((const void*)fn, "fst_dense_tap_wgrad")) return rc;
  hipLaunchKernelGGL(fn, dim3((unsigned)p.ksplit, (unsigned)p.n_groups, (unsigned)p.m_halves), dim3(512), lds, (hipStream_t)stream, p);
  FST_LAUNCH_CHECK();
  hipLaunchKernelGGL(tz_reduce_kernel, dim3((unsigned)((p.C * 96 + 255) / 256), (unsigned)M), dim3(256), 0, (hipStream_t)stream, p);
  FST_LAUNCH_CHECK();
  return 0;
}
