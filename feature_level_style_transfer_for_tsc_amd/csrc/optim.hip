// Optimiser updates of the train step as single-pass multi-tensor kernels (train_and_test.py:97-106, 742-754: ten
// torch.optim.RMSprop and one Adam over 770 small tensors).  torch's foreach implementation makes five (RMSprop) to eleven
// (Adam) passes over every tensor, each a separate multi_tensor_apply launch of at most 36-110 tensors: 150 launches and 2 ms
// per step for 36 MB of parameters.  Here a launch takes up to 64 (parameter, gradient, state...) pointer tuples BY VALUE in its
// kernel arguments (no device-side table to build or keep valid under hipGraph replay: the captured launch carries them) and
// makes ONE pass: read p, g and the moments, write p and the moments.
#include "fst_common.h"

#define OPT_MAX_T 64

struct RmspropArgs {
  float* p[OPT_MAX_T];
  const float* g[OPT_MAX_T];
  float* v[OPT_MAX_T];
  int numel[OPT_MAX_T];
  float lr[OPT_MAX_T];
  int n;
  float alpha, eps;
};

// torch.optim.RMSprop (centered = False, momentum = 0, weight_decay = 0), in torch's operation order:
//   v ← v·α;  v ← v + (1−α)·g·g;  avg = √v + ε;  p ← p + (−lr)·(g / avg)
__global__ __launch_bounds__(256) void rmsprop_multi_kernel(RmspropArgs a) {
  const int t = blockIdx.y;
  if (t >= a.n) return;
  float* __restrict__ p = a.p[t];
  const float* __restrict__ g = a.g[t];
  float* __restrict__ v = a.v[t];
  const int n = a.numel[t];
  const float lr = a.lr[t], alpha = a.alpha, eps = a.eps, oma = 1.0f - a.alpha;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float gi = g[i];
    float vi = v[i] * alpha;
    vi = vi + oma * gi * gi;
    v[i] = vi;
    p[i] = p[i] + (-lr) * (gi / (sqrtf(vi) + eps));
  }
}

extern "C" int fst_rmsprop_multi(float* const* p_host, const float* const* g_host, float* const* v_host, const int64_t* numel_host,
                                 const float* lr_host, int n_tensors, float alpha, float eps, void* stream) {
  FST_REQUIRE(p_host && g_host && v_host && numel_host && lr_host && n_tensors >= 0, "fst_rmsprop_multi: bad arguments");
  for (int base = 0; base < n_tensors; base += OPT_MAX_T) {
    RmspropArgs a;
    a.n = n_tensors - base < OPT_MAX_T ? n_tensors - base : OPT_MAX_T;
    a.alpha = alpha; a.eps = eps;
    long long most = 0;
    for (int i = 0; i < a.n; ++i) {
      FST_REQUIRE(p_host[base + i] && g_host[base + i] && v_host[base + i] && numel_host[base + i] > 0 && numel_host[base + i] < (1LL << 31),
                  "fst_rmsprop_multi: tensor %d: null pointer or bad element count", base + i);
      a.p[i] = p_host[base + i]; a.g[i] = g_host[base + i]; a.v[i] = v_host[base + i];
      a.numel[i] = (int)numel_host[base + i]; a.lr[i] = lr_host[base + i];
      most = most > numel_host[base + i] ? most : numel_host[base + i];
    }
    int bx = (int)((most + 1023) / 1024);                  // four elements per thread for the largest tensor ...
    bx = bx < 1 ? 1 : (bx > 64 ? 64 : bx);                 // ... within 64 workgroups per tensor (grid-stride beyond)
    hipLaunchKernelGGL(rmsprop_multi_kernel, dim3((unsigned)bx, (unsigned)a.n), dim3(256), 0, (hipStream_t)stream, a);
    FST_LAUNCH_CHECK();
  }
  return 0;
}

struct AdamArgs {
  float* p[OPT_MAX_T];
  const float* g[OPT_MAX_T];
  float* m[OPT_MAX_T];
  float* v[OPT_MAX_T];
  int numel[OPT_MAX_T];
  int n;
  const float* step;    // DEVICE scalar: the step count t (already incremented), shared by every tensor
  float lr, beta1, beta2, eps;
};

// torch.optim.Adam (capturable branch, amsgrad = False, weight_decay = 0):
//   m ← β₁m + (1−β₁)g;  v ← β₂v + (1−β₂)g²;  p ← p − (lr / (1−β₁ᵗ)) · m / (√v / √(1−β₂ᵗ) + ε)
__global__ __launch_bounds__(256) void adam_multi_kernel(AdamArgs a) {
  const int t = blockIdx.y;
  if (t >= a.n) return;
  float* __restrict__ p = a.p[t];
  const float* __restrict__ g = a.g[t];
  float* __restrict__ m = a.m[t];
  float* __restrict__ v = a.v[t];
  const int n = a.numel[t];
  const float step = a.step[0];
  const float bc1 = 1.0f - powf(a.beta1, step), bc2s = sqrtf(1.0f - powf(a.beta2, step));
  const float step_size = a.lr / bc1;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float gi = g[i];
    const float mi = m[i] * a.beta1 + (1.0f - a.beta1) * gi;
    const float vi = v[i] * a.beta2 + (1.0f - a.beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] = p[i] - step_size * (mi / (sqrtf(vi) / bc2s + a.eps));
  }
}

extern "C" int fst_adam_multi(float* const* p_host, const float* const* g_host, float* const* m_host, float* const* v_host,
                              const int64_t* numel_host, int n_tensors, const float* step_dev, float lr, float beta1, float beta2,
                              float eps, void* stream) {
  FST_REQUIRE(p_host && g_host && m_host && v_host && numel_host && step_dev && n_tensors >= 0, "fst_adam_multi: bad arguments");
  for (int base = 0; base < n_tensors; base += OPT_MAX_T) {
    AdamArgs a;
    a.n = n_tensors - base < OPT_MAX_T ? n_tensors - base : OPT_MAX_T;
    a.step = step_dev; a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
    long long most = 0;
    for (int i = 0; i < a.n; ++i) {
      FST_REQUIRE(p_host[base + i] && g_host[base + i] && m_host[base + i] && v_host[base + i] && numel_host[base + i] > 0 &&
                  numel_host[base + i] < (1LL << 31), "fst_adam_multi: tensor %d: null pointer or bad element count", base + i);
      a.p[i] = p_host[base + i]; a.g[i] = g_host[base + i]; a.m[i] = m_host[base + i]; a.v[i] = v_host[base + i];
      a.numel[i] = (int)numel_host[base + i];
      most = most > numel_host[base + i] ? most : numel_host[base + i];
    }
    int bx = (int)((most + 1023) / 1024);
    bx = bx < 1 ? 1 : (bx > 64 ? 64 : bx);
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)bx, (unsigned)a.n), dim3(256), 0, (hipStream_t)stream, a);
    FST_LAUNCH_CHECK();
  }
  return 0;
}
