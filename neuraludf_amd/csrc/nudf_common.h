// Shared device helpers for the NeuralUDF hot-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NUDF_WAVE 64

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

extern "C" void nudf_set_error(const char* where, hipError_t e);
extern "C" int32_t* nudf_status_flag(void);     // nudf_api.hip: the caller's non-finite status word (or NULL)

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
  // v_mul_f32 with the DPP modifier on its first source: lanes whose DPP source does not exist (or whose row is
  // masked off) are simply not written, so no identity operand is needed -- 6 VALU instructions.  hipcc does not
  // fold update_dpp(1.0, v) + fmul (it does fold the add scan), which cost a v_mov identity + a v_mov_dpp per step.
  // The s_nop 1 before each step is the 2-wait-state VALU-write -> DPP-read hazard (inline asm is not scanned).
  asm volatile(
      "s_nop 1\n\t"
      "v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_mul_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_mul_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  return v;
}
__device__ __forceinline__ float wave_incl_scan_add(float v) {
  // same form as the product scan (hipcc folds only the four row_shr steps of the builtin version; the two masked
  // row_bcast steps became v_mov identity + v_mov_dpp + v_add each)
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  return v;
}
// ---- several independent 64-lane scans at once: the steps of 2-4 scans are interleaved so that the two wait states
// a DPP read needs after the VALU write of its source are filled with the other scans' instructions instead of
// s_nop (a lone scan spends as many issue cycles on s_nop as on arithmetic).  OP = mul | add.
__device__ __forceinline__ void wave_incl_scan_mul2(float& v0, float& v1) {
  asm volatile(
      "s_nop 1\n\t"
      "s_nop 0\n\t"
      "v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_mul_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_mul_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_mul_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_mul_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v0), "+v"(v1));
}
__device__ __forceinline__ void wave_incl_scan_mul3(float& v0, float& v1, float& v2) {
  asm volatile(
      "s_nop 1\n\t"
      "v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %2, %2, %2 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %2, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_mul_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_mul_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_mul_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_mul_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_mul_f32_dpp %2, %2, %2 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v0), "+v"(v1), "+v"(v2));
}
__device__ __forceinline__ void wave_incl_scan_mul4(float& v0, float& v1, float& v2, float& v3) {
  asm volatile(
      "s_nop 1\n\t"
      "v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %2, %2, %2 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %3, %3, %3 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %3, %3, %3 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %2, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %3, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_mul_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_mul_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_mul_f32_dpp %3, %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_mul_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_mul_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_mul_f32_dpp %2, %2, %2 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_mul_f32_dpp %3, %3, %3 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
}
__device__ __forceinline__ void wave_incl_scan_add2(float& v0, float& v1) {
  asm volatile(
      "s_nop 1\n\t"
      "s_nop 0\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v0), "+v"(v1));
}
__device__ __forceinline__ void wave_incl_scan_add3(float& v0, float& v1, float& v2) {
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v0), "+v"(v1), "+v"(v2));
}
__device__ __forceinline__ void wave_incl_scan_add4(float& v0, float& v1, float& v2, float& v3) {
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
}
template <int N>
__device__ __forceinline__ void wave_incl_scan_mul_n(float (&v)[N]) {
  int i = 0;
  if (N == 3 || N == 6) {
#pragma unroll
    for (int b = 0; b + 3 <= N; b += 3) wave_incl_scan_mul3(v[b], v[b + 1], v[b + 2]);
    return;
  }
#pragma unroll
  for (; i + 4 <= N; i += 4) wave_incl_scan_mul4(v[i], v[i + 1], v[i + 2], v[i + 3]);
  if (N - i == 3) wave_incl_scan_mul3(v[i], v[i + 1], v[i + 2]);
  else if (N - i == 2) wave_incl_scan_mul2(v[i], v[i + 1]);
  else if (N - i == 1) v[i] = wave_incl_scan_mul(v[i]);
}
// totals over the 64 lanes of N values, returned to every lane
template <int N>
__device__ __forceinline__ void wave_sum_n(float (&v)[N]) {
  int i = 0;
#pragma unroll
  for (; i + 4 <= N; i += 4) wave_incl_scan_add4(v[i], v[i + 1], v[i + 2], v[i + 3]);
  if (N - i == 3) wave_incl_scan_add3(v[i], v[i + 1], v[i + 2]);
  else if (N - i == 2) wave_incl_scan_add2(v[i], v[i + 1]);
  else if (N - i == 1) v[i] = wave_incl_scan_add(v[i]);
#pragma unroll
  for (int k = 0; k < N; ++k)
    v[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[k]), 63));
}

// lane l <- lane l-1 (lane 0 <- `first`) / lane l <- lane l+1 (lane 63 <- `last`)
__device__ __forceinline__ float wave_shift_up1(float v, float first) { return dpp_f<NUDF_DPP_WAVE_SHR1, 0xf>(first, v); }
__device__ __forceinline__ float wave_shift_down1(float v, float last) { return dpp_f<NUDF_DPP_WAVE_SHL1, 0xf>(last, v); }
__device__ __forceinline__ float wave_bcast(float v, int src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}
// sum over the 64 lanes, returned to every lane
__device__ __forceinline__ float wave_sum(float v) { return wave_bcast(wave_incl_scan_add(v), 63); }
// inclusive suffix sum: out[l] = sum_{j>=l} v[j]: reverse the lane order (one ds_bpermute, crossbar only), run the
// DPP prefix scan, reverse back -- 8 instructions.  (DPP has no backward row broadcast, and total - prefix would
// cancel catastrophically for the small suffixes behind a surface; the earlier version walked 6 shuffle steps of
// bpermute + select + add.)
__device__ __forceinline__ float wave_reverse(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((63 - lane_id()) << 2, __builtin_bit_cast(int, v)));
}
__device__ __forceinline__ float wave_incl_rscan_add(float v) { return wave_reverse(wave_incl_scan_add(wave_reverse(v))); }

template <int N>
__device__ __forceinline__ void wave_incl_rscan_add_n(float (&v)[N]) {
  // N independent suffix scans: reverse, interleaved prefix scans, reverse back
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = wave_reverse(v[k]);
  int i = 0;
#pragma unroll
  for (; i + 4 <= N; i += 4) wave_incl_scan_add4(v[i], v[i + 1], v[i + 2], v[i + 3]);
  if (N - i == 3) wave_incl_scan_add3(v[i], v[i + 1], v[i + 2]);
  else if (N - i == 2) wave_incl_scan_add2(v[i], v[i + 1]);
  else if (N - i == 1) v[i] = wave_incl_scan_add(v[i]);
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = wave_reverse(v[k]);
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
