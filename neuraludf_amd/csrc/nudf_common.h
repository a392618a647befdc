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

// ---- wave-level scans (64 lanes) on DPP -------------------------------------------------------------
// Cross-lane steps are DPP modifiers of the VALU op itself (row_shr within the 16-lane rows, then row_bcast:15 /
// row_bcast:31 to carry across rows -- GFX9/CDNA encodings), i.e. 6 VALU instructions per 64-lane scan and no LDS
// traffic; the shuffle-based (ds_bpermute) version cost ~4 instructions + an LDS round trip per step.
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

#define NUDF_DPP_ROW_SHR(n) (0x110 + (n))
#define NUDF_DPP_ROW_BCAST15 0x142
#define NUDF_DPP_ROW_BCAST31 0x143
#define NUDF_DPP_WAVE_SHL1 0x130
#define NUDF_DPP_WAVE_SHR1 0x138

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float identity, float v) {
  // lanes whose DPP source is out of range (or masked off) read `identity`
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity),
                                                                __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}

__device__ __forceinline__ float wave_incl_scan_mul(float v) {
  v *= dpp_f<NUDF_DPP_ROW_SHR(1), 0xf>(1.0f, v);
  v *= dpp_f<NUDF_DPP_ROW_SHR(2), 0xf>(1.0f, v);
  v *= dpp_f<NUDF_DPP_ROW_SHR(4), 0xf>(1.0f, v);
  v *= dpp_f<NUDF_DPP_ROW_SHR(8), 0xf>(1.0f, v);
  v *= dpp_f<NUDF_DPP_ROW_BCAST15, 0xa>(1.0f, v);
  v *= dpp_f<NUDF_DPP_ROW_BCAST31, 0xc>(1.0f, v);
  return v;
}
__device__ __forceinline__ float wave_incl_scan_add(float v) {
  v += dpp_f<NUDF_DPP_ROW_SHR(1), 0xf>(0.0f, v);
  v += dpp_f<NUDF_DPP_ROW_SHR(2), 0xf>(0.0f, v);
  v += dpp_f<NUDF_DPP_ROW_SHR(4), 0xf>(0.0f, v);
  v += dpp_f<NUDF_DPP_ROW_SHR(8), 0xf>(0.0f, v);
  v += dpp_f<NUDF_DPP_ROW_BCAST15, 0xa>(0.0f, v);
  v += dpp_f<NUDF_DPP_ROW_BCAST31, 0xc>(0.0f, v);
  return v;
}
// lane l <- lane l-1 (lane 0 <- `first`) / lane l <- lane l+1 (lane 63 <- `last`)
__device__ __forceinline__ float wave_shift_up1(float v, float first) { return dpp_f<NUDF_DPP_WAVE_SHR1, 0xf>(first, v); }
__device__ __forceinline__ float wave_shift_down1(float v, float last) { return dpp_f<NUDF_DPP_WAVE_SHL1, 0xf>(last, v); }
__device__ __forceinline__ float wave_bcast(float v, int src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}
// sum over the 64 lanes, returned to every lane
__device__ __forceinline__ float wave_sum(float v) { return wave_bcast(wave_incl_scan_add(v), 63); }
// inclusive suffix sum: out[l] = sum_{j>=l} v[j].  Shuffle based: DPP has no backward row broadcast, and
// total - prefix would cancel catastrophically for the small suffixes behind a surface.
__device__ __forceinline__ float wave_incl_rscan_add(float v) {
  const int l = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    float o = __shfl_down(v, d, 64);
    if (l + d < 64) v += o;
  }
  return v;
}

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
