// Shared device helpers for the NeuralUDF hot-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NUDF_WAVE 64

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

extern "C" void nudf_set_error(const char* where, hipError_t e);

#define NUDF_CHECK_LAUNCH(where)                         \
  do {                                                   \
    hipError_t _e = hipGetLastError();                   \
    if (_e != hipSuccess) {                              \
      nudf_set_error(where, _e);                         \
      return (int)_e;                                    \
    }                                                    \
  } while (0)

// ---- wave-level scans (64 lanes) -------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ float wave_incl_scan_mul(float v) {
  const int l = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    float o = __shfl_up(v, d, 64);
    if (l >= d) v *= o;
  }
  return v;
}
__device__ __forceinline__ float wave_incl_scan_add(float v) {
  const int l = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    float o = __shfl_up(v, d, 64);
    if (l >= d) v += o;
  }
  return v;
}
// inclusive suffix sum: out[l] = sum_{j>=l} v[j]
__device__ __forceinline__ float wave_incl_rscan_add(float v) {
  const int l = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    float o = __shfl_down(v, d, 64);
    if (l + d < 64) v += o;
  }
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}
__device__ __forceinline__ float wave_bcast(float v, int src) { return __shfl(v, src, 64); }

// ---- activations with torch semantics ----------------------------------------------
// nn.Softplus(beta=100, threshold=20): x if 100x>20 else log1p(exp(100x))/100
__device__ __forceinline__ float softplus100(float a) {
  float t = 100.0f * a;
  return (t > 20.0f) ? a : log1pf(expf(t)) * 0.01f;
}
// its derivative as autograd computes it: 1 above the threshold, z/(z+1) below
__device__ __forceinline__ float softplus100_grad(float a) {
  float t = 100.0f * a;
  if (t > 20.0f) return 1.0f;
  float z = expf(t);
  return z / (z + 1.0f);
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
