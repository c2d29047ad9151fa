// Shared host/device helpers for libfst_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/fst_hip.h"

#define FST_PLAN_HDR FST_PLAN_HEADER   // 16 ints; see plan.py (fields 0..10 used)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

void fst_set_error(const char* fmt, ...);

#define FST_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      fst_set_error(__VA_ARGS__);         \
      return -1;                          \
    }                                     \
  } while (0)

#define FST_LAUNCH_CHECK()                                                      \
  do {                                                                          \
    hipError_t e__ = hipGetLastError();                                         \
    if (e__ != hipSuccess) {                                                    \
      fst_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
      return (int)e__;                                                          \
    }                                                                           \
  } while (0)

struct PlanView {
  int n_chunks, n_mgroups, MB, ntaps, dil, pad_left, chunk_cap, total_records, n_items, items_per_wg, n_stages;
  const int32_t* chunk;  // [n_chunks][4]  src, c_begin, c_count, 0
  const int32_t* mg;     // [n_mgroups][n_chunks][4]  tap_lo, tap_hi, rec_off, stage_off
  const int32_t* item;   // [n_items][4]  g, q, row_block, 0   (q < 0: padding)
};

static inline __host__ __device__ PlanView plan_view(const int32_t* p) {
  PlanView v;
  v.n_chunks = p[0]; v.n_mgroups = p[1]; v.MB = p[2]; v.ntaps = p[3]; v.dil = p[4]; v.pad_left = p[5];
  v.chunk_cap = p[6]; v.total_records = p[7]; v.n_items = p[8]; v.items_per_wg = p[9]; v.n_stages = p[10];
  v.chunk = p + FST_PLAN_HDR;
  v.mg = v.chunk + 4 * v.n_chunks;
  v.item = v.mg + 4 * v.n_chunks * v.n_mgroups;
  return v;
}

static inline int plan_expected_len(const int32_t* p) {
  return FST_PLAN_HDR + 4 * p[0] + 4 * p[0] * p[1] + 4 * p[8];
}

// Raise a kernel's dynamic-LDS limit to the full 160 KiB once (idempotent; never called again for that kernel,
// so nothing but launches happens while a stream is being captured into a hipGraph).
int fst_allow_full_lds(const void* fn, const char* who);
int fst_cu_count(void);   // compute units of the current device (0 if the query fails)

// Host-side sanity check of a plan against the tensor shapes a launch will touch.
int fst_check_plan(const int32_t* plan_host, int plan_len, int M, const char* who);
