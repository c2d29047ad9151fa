// Probe: how fast does a CU get accumulator-layout tiles out to memory, by row pitch?
//   hipcc -O3 --offload-arch=gfx950 tools/store_pattern_probe.hip -o tools/store_pattern_probe && ./tools/store_pattern_probe
// One 512-thread workgroup per CU (as the fused WN kernels run); every wave stores 24 tiles of 32 rows x 32 samples in the
// accumulator layout (a store instruction = 2 rows x 128 contiguous bytes) to rows `pitch` bytes apart — the t,s / a_next / out
// stores of one forward tile — `reps` times over different batch elements.  Variants: dword stores vs 16-byte stores (a lane
// owns 4 consecutive samples of a row), and every CU storing at once vs one CU in `stride` storing.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int WIDE>
__global__ __launch_bounds__(512) void store_kernel(float* out, long long pitch_f, long long seq_f, int rows, int reps, int active_stride) {
  if (blockIdx.x % active_stride) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  float v = (float)tid;
  for (int rep = 0; rep < reps; ++rep) {
    float* base = out + ((long long)rep * gridDim.x + blockIdx.x) * seq_f + wave * 32;      // this wave's 32 columns of the tile
    for (int row0 = 0; row0 < rows; row0 += 32) {
      if (!WIDE) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * half;
          base[(long long)row * pitch_f + l31] = v;
        }
      } else {
        const int rrow = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = row0 + rrow + 8 * j;
          *reinterpret_cast<float4*>(base + (long long)row * pitch_f + c4) = make_float4(v, v, v, v);
        }
      }
    }
  }
}

int main() {
  int dev = 0; hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, dev));
  const int cus = prop.multiProcessorCount, rows = 768, reps = 8;        // 768 rows = t,s (240) + a_next (120) + out (120) ... of a tile, rounded up: 24 tiles of 32 rows per wave
  printf("%s, %d CUs; per launch every active CU stores %d x %d rows x 1 KB\n", prop.name, cus, reps, rows);
  for (int pitch_b : {2048, 2048 + 128, 2048 + 256, 4096, 4096 + 128, 1024}) {
    const long long pitch_f = pitch_b / 4, seq_f = (long long)rows * pitch_f;
    const size_t bytes = (size_t)reps * cus * seq_f * 4;
    float* out; CHECK(hipMalloc(&out, bytes));
    CHECK(hipMemset(out, 0, bytes));
    for (int wide = 0; wide < 2; ++wide)
      for (int stride : {1, 2, 8}) {
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        auto launch = [&]() {
          if (wide) hipLaunchKernelGGL(store_kernel<1>, dim3(cus), dim3(512), 0, 0, out, pitch_f, seq_f, rows, reps, stride);
          else hipLaunchKernelGGL(store_kernel<0>, dim3(cus), dim3(512), 0, 0, out, pitch_f, seq_f, rows, reps, stride);
        };
        launch(); CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) launch();
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double active = (cus + stride - 1) / stride;
        const double gb = active * reps * rows * 1024.0 * 10 / 1e9;
        printf("pitch %5d B  %s  1 CU in %d active: %8.1f us per launch, %7.1f GB/s total, %6.1f GB/s per active CU\n", pitch_b,
               wide ? "16-byte stores" : "dword stores  ", stride, ms * 100, gb / (ms * 1e-3), gb / (ms * 1e-3) / active);
      }
    CHECK(hipFree(out));
  }
  return 0;
}
