// What can ONE wave per SIMD sustain on v_mfma_f32_32x32x2_f32 (gfx950)?  Ceiling measurements for the wave-private
// chain kernel (mlp_chain_rows.hip):
//   A  pure MFMA stream, 8 independent accumulators, operands fixed in registers
//   B  A + the weight stream of the chain K loop: 8 coalesced global_load_dwordx4 per 32 MFMAs from a 256 KB
//      (L2-resident) buffer, prefetch distance 1 group (two register sets)
//   C  B with prefetch distance 2 groups (three register sets)
//   D  B + one ds_read_b128 per group (the activation operand)
//   V  VALU-only epilogue-like stream: cycles per plain / transcendental instruction for a lone wave
// Every variant runs with 1 wave per SIMD (150 KB of LDS per workgroup) and with 2 (no LDS).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_1wave mfma_1wave.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int LDS_BYTES>
__global__ __launch_bounds__(256, (LDS_BYTES > 1024 ? 1 : 2)) void k(const f32x4* __restrict__ W, float* out, unsigned long long* ticks, int layers) {
  __shared__ __attribute__((aligned(16))) float lds[LDS_BYTES / 4];
  const int lane = threadIdx.x & 63;
  f32x16 acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
  if (LDS_BYTES > 1024) lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  float x = threadIdx.x * 1e-3f;
  const f32x4* bptr = W + lane;
  const float* arow = lds + (lane & 31) * 292 + 4 * (lane >> 5);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int l = 0; l < layers; ++l) {
    if (MODE == 0) {
#pragma unroll 1
      for (int g = 0; g < 32; ++g) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, acc[j], 0, 0, 0);
      }
    } else if (MODE == 1 || MODE == 3) {
      f32x4 b0[8], b1[8], a0 = {x, x, x, x}, a1 = {x, x, x, x};
#pragma unroll
      for (int j = 0; j < 8; ++j) b0[j] = bptr[j * 64];
#pragma unroll 1
      for (int g = 0; g < 32; g += 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) b1[j] = bptr[(g + 1) * 512 + j * 64];
        if (MODE == 3) a1 = *reinterpret_cast<const f32x4*>(arow + (g + 1) * 8);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[j][jj], a0[jj], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        const int gn = (g + 2) & 31;
#pragma unroll
        for (int j = 0; j < 8; ++j) b0[j] = bptr[gn * 512 + j * 64];
        if (MODE == 3) a0 = *reinterpret_cast<const f32x4*>(arow + gn * 8);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[j][jj], a1[jj], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (MODE == 2) {
      f32x4 b0[8], b1[8], b2[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        b0[j] = bptr[j * 64];
        b1[j] = bptr[512 + j * 64];
      }
#pragma unroll 1
      for (int g = 0; g < 30; g += 3) {
#define STEP(BC, BN, GN)                                                                                     \
  _Pragma("unroll") for (int j = 0; j < 8; ++j) BN[j] = bptr[((GN)&31) * 512 + j * 64];                      \
  __builtin_amdgcn_sched_barrier(0);                                                                         \
  _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) _Pragma("unroll") for (int j = 0; j < 8; ++j) acc[j] =    \
      __builtin_amdgcn_mfma_f32_32x32x2f32(BC[j][jj], x, acc[j], 0, 0, 0);                                   \
  __builtin_amdgcn_sched_barrier(0);
        STEP(b0, b2, g + 2)
        STEP(b1, b0, g + 3)
        STEP(b2, b1, g + 4)
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[j][jj], x, acc[j], 0, 0, 0);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[j][jj], x, acc[j], 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

// VALU stream of a lone wave: KIND 0 plain fma chain x16 independent, 1 v_exp, 2 v_log, 3 softplus element
template <int KIND, int LDS_BYTES>
__global__ __launch_bounds__(256, 1) void kv(float* out, unsigned long long* ticks, int iters) {
  __shared__ float lds[LDS_BYTES / 4];
  if (LDS_BYTES > 1024) lds[threadIdx.x] = 1.0f;
  __syncthreads();
  float v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = threadIdx.x * 1e-3f + r;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (KIND == 0) v[r] = __builtin_fmaf(v[r], 1.0001f, 0.5f);
      if (KIND == 1) v[r] = __builtin_amdgcn_exp2f(v[r]);
      if (KIND == 2) v[r] = __builtin_amdgcn_logf(v[r]);
      if (KIND == 3) {
        const float tt = 100.0f * v[r];
        const float z = __builtin_amdgcn_exp2f(fabsf(tt) * -1.44269504f);
        v[r] = (fmaxf(tt, 0.0f) + __builtin_amdgcn_logf(1.0f + z) * 0.69314718f) * 0.01f;
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += v[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE, int LDS_BYTES>
void run(const char* name, const f32x4* W, float* out, unsigned long long* ticks, int nblk) {
  const int layers = 40;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, LDS_BYTES>), dim3(nblk), dim3(256), 0, 0, W, out, ticks, layers);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2048];
    hipMemcpy(h, ticks, nblk * 8, hipMemcpyDeviceToHost);
    double mt = 0, mean = 0;
    for (int i = 0; i < nblk; ++i) {
      mt = h[i] > mt ? h[i] : mt;
      mean += h[i];
    }
    mean /= nblk;
    const double nm = (double)layers * 1024.0;   // MFMAs per wave
    if (rep == 1)
      printf("%-34s blocks %4d: %.3f ms %.1f TFLOP/s  %.2f ticks per MFMA (mean over blocks), max/mean %.3f\n", name, nblk, ms,
             (double)nblk * 4 * nm * 4096.0 / ms / 1e9, mean / nm, mt / mean);
  }
}

template <int KIND, int LDS_BYTES>
void runv(const char* name, float* out, unsigned long long* ticks, int nblk) {
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((kv<KIND, LDS_BYTES>), dim3(nblk), dim3(256), 0, 0, out, ticks, iters);
    hipDeviceSynchronize();
    unsigned long long h[2048];
    hipMemcpy(h, ticks, nblk * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < nblk; ++i) mean += h[i];
    mean /= nblk;
    if (rep == 1) printf("%-34s blocks %4d: %.2f ticks per 16-element group\n", name, nblk, mean / iters);
  }
}

int main() {
  f32x4* W;
  float* out;
  unsigned long long* ticks;
  hipMalloc(&W, 32 * 512 * 16 + 4096 * 16);
  hipMemset(W, 0, 32 * 512 * 16 + 4096 * 16);
  hipMalloc(&out, 2048 * 256 * 4);
  hipMalloc(&ticks, 2048 * 8);
  constexpr int BIG = 150 * 1024;
  run<0, BIG>("A pure MFMA, 1 wave/SIMD", W, out, ticks, 256);
  run<0, 1024>("A pure MFMA, 2 waves/SIMD", W, out, ticks, 512);
  run<1, BIG>("B + weight stream pf1, 1 wave/SIMD", W, out, ticks, 256);
  run<2, BIG>("C + weight stream pf2, 1 wave/SIMD", W, out, ticks, 256);
  run<3, BIG>("D + weights pf1 + ds_read, 1 wave", W, out, ticks, 256);
  runv<0, BIG>("V fma x16, 1 wave/SIMD", out, ticks, 256);
  runv<0, 1024>("V fma x16, 2 waves/SIMD", out, ticks, 512);
  runv<1, BIG>("V exp2 x16, 1 wave/SIMD", out, ticks, 256);
  runv<2, BIG>("V log2 x16, 1 wave/SIMD", out, ticks, 256);
  runv<3, BIG>("V softplus x16, 1 wave/SIMD", out, ticks, 256);
  runv<3, 1024>("V softplus x16, 2 waves/SIMD", out, ticks, 512);
  return 0;
}
