// Micro-benchmark: sustained rate of v_mfma_f32_32x32x2_f32 per SIMD, (a) register operands only,
// (b) with the k-step pattern of conv_gemm_pipe_kernel<8,1> (8 A + 1 B ds_read_b32 per 8 MFMAs).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, int iters) {
  __shared__ float lds[8 * 8 * 64 + 16 * 128];
  for (int i = threadIdx.x; i < 8 * 8 * 64 + 16 * 128; i += blockDim.x) lds[i] = 1e-3f * (i & 7);
  __syncthreads();
  f32x16 acc[8];
  for (int m = 0; m < 8; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  const int lane = threadIdx.x & 63;
  float a0 = lane * 1e-3f, b0 = 1.0f;
  for (int it = 0; it < iters; ++it) {
    for (int r = 0; r < 8; ++r) {
      float av[8], bv;
      if (MODE == 0) {
#pragma unroll
        for (int m = 0; m < 8; ++m) av[m] = a0 + m;
        bv = b0;
      } else {
#pragma unroll
        for (int m = 0; m < 8; ++m) av[m] = lds[(r * 8 + m) * 64 + lane];
        bv = lds[8 * 8 * 64 + (2 * r + (lane >> 5)) * 128 + (lane & 31)];
      }
#pragma unroll
      for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv, acc[m], 0, 0, 0);
    }
  }
  float s = 0;
  for (int m = 0; m < 8; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int WAVES>
void run(const char* name, int blocks) {
  float* out; hipMalloc(&out, sizeof(float) * blocks * 64 * WAVES);
  int iters = 400;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double mfma_per_wave = (double)iters * 64;
  double waves_per_simd = (double)blocks * WAVES / (256.0 * 4);
  double cyc_per_mfma = ms * 1e-3 * 2.4e9 / (mfma_per_wave * waves_per_simd);
  double tflops = (double)blocks * WAVES * mfma_per_wave * 4096 / (ms * 1e-3) / 1e12;
  printf("%-34s blocks=%4d waves/blk=%d  %.3f ms  %.1f cycles/MFMA/SIMD (@2.4GHz)  %.1f TFLOP/s\n", name, blocks, WAVES, ms, cyc_per_mfma, tflops);
  hipFree(out);
}

int main() {
  run<0, 4>("regs, 1 wave/SIMD", 256);
  run<0, 4>("regs, 2 waves/SIMD", 512);
  run<1, 4>("lds k-step, 1 wave/SIMD", 256);
  run<1, 4>("lds k-step, 2 waves/SIMD", 512);
  run<1, 4>("lds k-step, 4 waves/SIMD", 1024);
  return 0;
}
