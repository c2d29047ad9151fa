// Micro-benchmark: what LDS-DMA fill rate does a CU sustain through the stage ring the GEMM kernels use?
// A workgroup of 4 waves streams `stages` stages of PIECES 1-KiB LDS-DMA pieces (global_load_lds_dwordx4) into a ring of
// NS slots, DEPTH stages in flight, one counted vmcnt + one raw s_barrier per stage, nothing else (no LDS reads, no MFMA).
// Source A: a small image every workgroup re-reads (L2-resident weights); source B: a large buffer streamed once (HBM).
//   build: hipcc -O3 --offload-arch=gfx950 tools/dma_ring_probe.hip -o tools/dma_ring_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define GP(p) ((const __attribute__((address_space(1))) void*)(p))
#define LP(p) ((__attribute__((address_space(3))) void*)(p))
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// PPW = pieces per wave per stage (4*PPW pieces of 1 KiB per stage); FRAC_A = how many of the PPW pieces per wave are "A"
template <int PPW, int DEPTH, int NS>
__global__ __launch_bounds__(256) void ring(const char* a_img, long long a_bytes, const char* b_buf, long long b_bytes, int stages,
                                            int a_ppw, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int SLOT = 4 * PPW * 1024;
  const long long wg = blockIdx.x;
  auto issue = [&](int k, int slot) {
    char* sl = lds + slot * SLOT;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int idx = wave + 4 * i;
      const char* src;
      if (i < a_ppw) src = a_img + (((long long)k * 4 * PPW + idx) * 1024) % a_bytes + lane * 16;
      else src = b_buf + ((wg * stages + k) * (4LL * PPW) + idx) * 1024 % b_bytes + lane * 16;
      __builtin_amdgcn_global_load_lds(GP(src), LP(sl + idx * 1024), 16, 0, 0);
    }
  };
  for (int d = 0; d < DEPTH && d < stages; ++d) issue(d, d % NS);
  for (int k = 0; k < stages; ++k) {
    if (k + DEPTH - 1 < stages) wait_vm<PPW*(DEPTH - 1)>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (k + DEPTH < stages) issue(k + DEPTH, (k + DEPTH) % NS);
  }
  __syncthreads();
  if (threadIdx.x == 0 && lds[blockIdx.x & 1023] == 123) sink[0] = 1;
}

template <int PPW, int DEPTH, int NS>
void run(const char* a, long long ab, const char* b, long long bb, int wgs, int stages, int a_ppw, int* sink) {
  const size_t ldsb = (size_t)NS * 4 * PPW * 1024;
  auto fn = ring<PPW, DEPTH, NS>;
  hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(fn, dim3(wgs), dim3(256), ldsb, 0, a, ab, b, bb, stages, a_ppw, sink);
  hipEventRecord(e0);
  const int reps = 5;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(fn, dim3(wgs), dim3(256), ldsb, 0, a, ab, b, bb, stages, a_ppw, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  const double bytes = (double)wgs * stages * 4 * PPW * 1024;
  const int per_cu = (int)(160 * 1024 / ldsb); 
  printf("PPW=%d (%2d KiB/stage, A share %d/%d) depth=%d slots=%d LDS=%3zu KiB (<=%d WG/CU) wgs=%5d: %7.1f us  %6.2f TB/s  %5.1f GB/s/CU\n",
         PPW, 4 * PPW, a_ppw, PPW, DEPTH, NS, ldsb / 1024, per_cu, wgs, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
}

int main() {
  const long long ab = 544 * 1024, bb = 1LL << 30;
  char *a, *b; int* sink;
  hipMalloc(&a, ab + 4096); hipMalloc(&b, bb + (1 << 20)); hipMalloc(&sink, 4);
  hipMemset(a, 0, ab); hipMemset(b, 0, bb);
  const int stages = 34;
  // the fused forward's shape: 26 pieces/stage -> PPW 7 (28 KiB), 16 of 26 from the image
  for (int wgs : {1024, 2048}) {
    run<7, 2, 3>(a, ab, b, bb, wgs, stages, 4, sink);
    run<7, 3, 4>(a, ab, b, bb, wgs, stages, 4, sink);
    run<7, 4, 5>(a, ab, b, bb, wgs, stages, 4, sink);
    run<7, 2, 3>(a, ab, b, bb, wgs, stages, 7, sink);   // everything from the L2-resident image
    run<7, 2, 3>(a, ab, b, bb, wgs, stages, 0, sink);   // everything streamed from HBM
    run<4, 2, 3>(a, ab, b, bb, wgs, stages, 2, sink);
    run<4, 4, 5>(a, ab, b, bb, wgs, stages, 2, sink);
    run<4, 6, 8>(a, ab, b, bb, wgs, stages, 2, sink);
    run<2, 8, 10>(a, ab, b, bb, wgs, stages * 2, 1, sink);
    run<9, 2, 3>(a, ab, b, bb, wgs / 2, stages, 4, sink); // TN=256-like: 34.7 KiB/stage, 1 WG/CU
    run<9, 3, 4>(a, ab, b, bb, wgs / 2, stages, 4, sink);
  }
  return 0;
}
