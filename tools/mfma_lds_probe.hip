// Micro-benchmark: v_mfma_f32_32x32x16_bf16 fed like the fused WN kernels' inner loop — per group of 6 MFMAs two
// 16-byte LDS fragment reads (hi, lo of a weight block) used against two resident B fragment pairs.
//   MODE 0: register operands only (pipe ceiling)
//   MODE 1: read the group's fragments, wait, multiply            (what the compiler emits for the straightforward loop)
//   MODE 2: fragments of group g+1 requested before the MFMAs of group g (register double buffer)
// 8 waves per workgroup (2 per SIMD), one workgroup per CU, optional barrier every 13 groups (a ring stage).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, bool BARRIER>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  for (int i = threadIdx.x; i < 13 * 2048 / 4; i += 512) reinterpret_cast<float*>(lds)[i] = 1e-3f * (i & 15);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x16 acc[5][2];
  for (int m = 0; m < 5; ++m) for (int c = 0; c < 2; ++c) for (int r = 0; r < 16; ++r) acc[m][c][r] = 0.f;
  bf16x8 bh[2], bl[2];
  for (int c = 0; c < 2; ++c) for (int j = 0; j < 8; ++j) { bh[c][j] = (__bf16)(0.01f * (lane + j + c)); bl[c][j] = (__bf16)(1e-4f * j); }
  auto frag = [&](int g, int lo) { return *reinterpret_cast<const bf16x8*>(lds + (g % 13) * 2048 + lo * 1024 + lane * 16); };
  auto group = [&](int g, bf16x8 ah, bf16x8 al) {
    const int m = g % 5;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      acc[m][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[c], acc[m][c], 0, 0, 0);
      acc[m][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[c], acc[m][c], 0, 0, 0);
      acc[m][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[c], acc[m][c], 0, 0, 0);
    }
  };
  for (int it = 0; it < iters; ++it) {
    if (BARRIER) __builtin_amdgcn_s_barrier();
    if (MODE == 0) {
#pragma unroll
      for (int g = 0; g < 13; ++g) group(g, bh[0], bl[1]);
    } else if (MODE == 1) {
#pragma unroll
      for (int g = 0; g < 13; ++g) {
        const bf16x8 ah = frag(g, 0), al = frag(g, 1);
        __builtin_amdgcn_sched_barrier(0);
        group(g, ah, al);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      bf16x8 ah = frag(0, 0), al = frag(0, 1);
#pragma unroll
      for (int g = 0; g < 13; ++g) {
        bf16x8 nh = ah, nl = al;
        if (g + 1 < 13) { nh = frag(g + 1, 0); nl = frag(g + 1, 1); }
        __builtin_amdgcn_sched_barrier(0);
        group(g, ah, al);
        __builtin_amdgcn_sched_barrier(0);
        ah = nh; al = nl;
      }
    }
  }
  float s = 0;
  for (int m = 0; m < 5; ++m) for (int c = 0; c < 2; ++c) for (int r = 0; r < 16; ++r) s += acc[m][c][r];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE, bool BARRIER>
void run(const char* name) {
  float* out; hipMalloc(&out, sizeof(float) * 256 * 512);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)k<MODE, BARRIER>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((k<MODE, BARRIER>), dim3(256), dim3(512), 100 * 1024, 0, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, BARRIER>), dim3(256), dim3(512), 100 * 1024, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma = 256.0 * 8 * iters * 13 * 6;
  printf("%-52s %.3f ms  %.1f TFLOP/s bf16 (%.2f of 2516)  %.1f ns per 13-group stage\n", name, ms, mfma * 32768 / (ms * 1e-3) / 1e12,
         mfma * 32768 / (ms * 1e-3) / 2516.6e12, ms * 1e6 / iters);
  hipFree(out);
}

int main() {
  run<0, false>("registers only");
  run<1, false>("read fragments, wait, multiply");
  run<2, false>("fragments one group ahead");
  run<1, true>("read, wait, multiply + barrier per stage");
  run<2, true>("one group ahead + barrier per stage");
  return 0;
}
