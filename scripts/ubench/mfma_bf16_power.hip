// What does this GPU sustain on NOTHING but v_mfma_f32_32x32x16_bf16 -- no LDS, no memory -- and how does that depend on the
// operand bits?  The chip clocks to its 1400 W package limit: the dense bf16 "peak" a kernel can be priced against is the rate
// of this loop on data of the kernel's bit activity, not 2.5 PFLOP/s at 2.4 GHz.  Two waves per SIMD (512 workgroups x 4 waves),
// four independent accumulators per wave, eight operand register sets rotated so consecutive MFMAs see different bits.
// Prints TFLOP/s, the shader clock s_memtime saw, and (via rocm-smi, if present) nothing -- run scripts/tn3_power.py for power.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256, 2) void k(const u32x4* __restrict__ src, float* out, unsigned long long* ticks, int iters) {
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  u32x4 q[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) q[i] = src[(blockIdx.x % 64) * 256 * 8 + i * 256 + threadIdx.x];
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      const bf16x8 x = __builtin_bit_cast(bf16x8, q[j]), y = __builtin_bit_cast(bf16x8, q[j + 1]);
      const bf16x8 x2 = __builtin_bit_cast(bf16x8, q[(j + 3) & 7]), y2 = __builtin_bit_cast(bf16x8, q[(j + 4) & 7]);
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x2, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y2, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x2, y2, a3, 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
static unsigned short bf16_of(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
int main() {
  const int nblk = 512, iters = 6000;
  const size_t nsrc = (size_t)64 * 256 * 8 * 8;       // bf16 values
  unsigned short* h = (unsigned short*)malloc(nsrc * 2);
  u32x4* src; float* out; unsigned long long* ticks;
  hipMalloc(&src, nsrc * 2); hipMalloc(&out, nblk * 256 * 4); hipMalloc(&ticks, nblk * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[4] = {"zeros", "one constant (1.0)", "uniform [0, 1) (activation-like: one sign, narrow exponents)", "normal-like (sum of 4 uniforms, both signs)"};
  for (int kind = 0; kind < 4; ++kind) {
    srand(1);
    for (size_t i = 0; i < nsrc; ++i) {
      float v = 0.0f;
      if (kind == 1) v = 1.0f;
      if (kind == 2) v = (float)rand() / RAND_MAX;
      if (kind == 3) v = ((float)rand() + rand() + rand() + rand()) / RAND_MAX - 2.0f;
      h[i] = bf16_of(v);
    }
    hipMemcpy(src, h, nsrc * 2, hipMemcpyHostToDevice);
    float best = 0, best_ms = 0; double clk = 0;
    for (int rep = 0; rep < 6; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(nblk), dim3(256), 0, 0, src, out, ticks, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long hs[512]; hipMemcpy(hs, ticks, sizeof(hs), hipMemcpyDeviceToHost);
      double tk = 0; for (int b = 0; b < nblk; ++b) tk += (double)hs[b]; tk /= nblk;
      const double tf = (double)nblk * 4 * iters * 16.0 * 32768.0 / (ms * 1e-3) / 1e12;
      if (rep >= 2 && tf > best) { best = (float)tf; best_ms = ms; clk = tk / (ms * 1e-3) / 1e6; }
    }
    printf("%-70s %8.1f TFLOP/s  (%.3f ms per launch, shader clock >= %.0f MHz: ticks of a workgroup / launch time)\n", names[kind], best, best_ms, clk);
  }
  return 0;
}
