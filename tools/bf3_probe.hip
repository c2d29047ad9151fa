// Probe (diagnostic, not part of the library): one 32x32x16 bf16 MFMA with the A fragment read row-wise
// (ds_read_b128) and the B fragment read through ds_read_b64_tr_b16 from a [k][n] image (n contiguous),
// using hi/lo split operands (3 MFMAs) -- checks the lane maps and the split accuracy against fp64.
//   hipcc -O3 --offload-arch=gfx950 tools/bf3_probe.hip -o tools/bf3_probe && tools/bf3_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define LDS3(T, p) ((__attribute__((address_space(3))) T*)(p))

#define RS 576   // bytes per image row: 256 columns of bf16 + 64 B pad

__device__ inline unsigned pack_hi(float a, float b) {     // two RNE bf16 in one dword (a low, b high)
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 v = {a, b};
  bf16x2 h = __builtin_convertvector(v, bf16x2);
  return __builtin_bit_cast(unsigned, h);
}

__global__ void probe(const float* A /*[32][16]*/, const float* B /*[16][256]*/, float* C /*[32][32]*/, int n0) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* bhi = lds;
  char* blo = lds + 16 * RS;
  char* ahi = lds + 32 * RS;      // [lane][8 bf16]
  char* alo = ahi + 1024;
  const int tid = threadIdx.x;
  // stage B: thread -> 4 consecutive columns of one row
  for (int e = tid; e < 16 * 64; e += 64) {
    const int row = e >> 6, col = (e & 63) * 4;
    float v[4], r[4];
    for (int i = 0; i < 4; ++i) v[i] = B[row * 256 + col + i];
    unsigned h0 = pack_hi(v[0], v[1]), h1 = pack_hi(v[2], v[3]);
    r[0] = v[0] - __uint_as_float(h0 << 16); r[1] = v[1] - __uint_as_float(h0 & 0xffff0000u);
    r[2] = v[2] - __uint_as_float(h1 << 16); r[3] = v[3] - __uint_as_float(h1 & 0xffff0000u);
    unsigned l0 = pack_hi(r[0], r[1]), l1 = pack_hi(r[2], r[3]);
    *reinterpret_cast<uint2*>(bhi + row * RS + col * 2) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(blo + row * RS + col * 2) = make_uint2(l0, l1);
  }
  // stage A in fragment order: lane l holds A[l&31][8*(l>>5) + j]
  {
    const int r = tid & 31, h = tid >> 5;
    unsigned hh[4], ll[4];
    for (int j = 0; j < 4; ++j) {
      float a = A[r * 16 + 8 * h + 2 * j], b = A[r * 16 + 8 * h + 2 * j + 1];
      hh[j] = pack_hi(a, b);
      ll[j] = pack_hi(a - __uint_as_float(hh[j] << 16), b - __uint_as_float(hh[j] & 0xffff0000u));
    }
    *reinterpret_cast<uint4*>(ahi + tid * 16) = make_uint4(hh[0], hh[1], hh[2], hh[3]);
    *reinterpret_cast<uint4*>(alo + tid * 16) = make_uint4(ll[0], ll[1], ll[2], ll[3]);
  }
  __syncthreads();
  const int lane = tid, h = lane >> 5, g = lane >> 4, u = lane & 15, q = u >> 2, pp = u & 3;
  const int off = (8 * h + q) * RS + 2 * (n0 + 16 * (g & 1) + 4 * pp);
  s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS3(s16x4, bhi + off));
  s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS3(s16x4, bhi + off + 4 * RS));
  s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS3(s16x4, blo + off));
  s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS3(s16x4, blo + off + 4 * RS));
  bf16x8 bh = __builtin_bit_cast(bf16x8, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
  bf16x8 bl = __builtin_bit_cast(bf16x8, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
  bf16x8 ah = *reinterpret_cast<bf16x8*>(ahi + lane * 16);
  bf16x8 al = *reinterpret_cast<bf16x8*>(alo + lane * 16);
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + (lane & 31)] = acc[r];
}

int main() {
  float hA[32 * 16], hB[16 * 256], hC[32 * 32];
  srand(1);
  for (auto& v : hA) v = (float)rand() / RAND_MAX * 2 - 1;
  for (auto& v : hB) v = ((float)rand() / RAND_MAX * 2 - 1) * 3.f;
  float *dA, *dB, *dC;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  int bad = 0;
  for (int n0 = 0; n0 < 256; n0 += 32) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 32 * RS + 2048, 0, dA, dB, dC, n0);
    hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
    double maxerr = 0, maxerr32 = 0, scale = 0;
    for (int m = 0; m < 32; ++m)
      for (int n = 0; n < 32; ++n) {
        double s = 0; float s32 = 0;
        for (int k = 0; k < 16; ++k) { s += (double)hA[m * 16 + k] * hB[k * 256 + n0 + n]; s32 = fmaf(hA[m * 16 + k], hB[k * 256 + n0 + n], s32); }
        maxerr = fmax(maxerr, fabs(hC[m * 32 + n] - s)); maxerr32 = fmax(maxerr32, fabs(s32 - s)); scale = fmax(scale, fabs(s));
      }
    printf("n0=%3d  max err %.3e (fp32 chain %.3e)  scale %.3f  rel %.2e\n", n0, maxerr, maxerr32, scale, maxerr / scale);
    if (maxerr > 1e-4 * scale) bad = 1;
  }
  printf(bad ? "PROBE FAILED\n" : "PROBE OK\n");
  return bad;
}
