// Probe: does global_load_lds_dwordx4 (LDS-DMA, 16 bytes per lane) accept a source address that is only 4-byte aligned?
//   hipcc -O3 --offload-arch=gfx950 tools/dma_align_probe.hip -o tools/dma_align_probe && ./tools/dma_align_probe
// Prints, per source offset 0..3 floats, whether the 1 KiB a wave moved equals src[off .. off + 256).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define GP(p) ((const __attribute__((address_space(1))) void*)(p))
#define LP(p) ((__attribute__((address_space(3))) void*)(p))

__global__ void probe(const float* src, int off, float* out) {
  __shared__ __attribute__((aligned(16))) float lds[256];
  const int lane = threadIdx.x;
  __builtin_amdgcn_global_load_lds(GP(src + off + 4 * lane), LP(lds), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 256; i += 64) out[i] = lds[i];
}

int main() {
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = (float)i;
  float *d, *o;
  hipMalloc(&d, 4096); hipMalloc(&o, 1024);
  hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
  for (int off = 0; off < 4; ++off) {
    hipMemset(o, 0, 1024);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d + 64, off, o);
    std::vector<float> r(256);
    hipError_t e = hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += r[i] != (float)(64 + off + i);
    printf("offset %d floats: %s (%d of 256 wrong; first values %g %g %g %g) %s\n", off, bad ? "MISMATCH" : "ok", bad, r[0], r[1], r[2], r[3],
           e == hipSuccess ? "" : hipGetErrorString(e));
  }
  return 0;
}
