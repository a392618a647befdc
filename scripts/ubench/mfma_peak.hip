// Pure v_mfma_f32_32x32x2_f32 issue-rate micro-benchmark: what does this GPU sustain with nothing but MFMAs
// (no LDS, no memory)?  Prints TFLOP/s and the shader clock derived from s_memtime.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256, 2) void k(float* out, unsigned long long* ticks, int iters) {
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
int main() {
  const int nblk = 512, iters = 20000;
  float* out; unsigned long long* ticks;
  hipMalloc(&out, nblk * 256 * 4); hipMalloc(&ticks, nblk * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(nblk), dim3(256), 0, 0, out, ticks, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[nblk]; hipMemcpy(h, ticks, nblk * 8, hipMemcpyDeviceToHost);
    double mt = 0; for (int i = 0; i < nblk; ++i) mt = h[i] > mt ? h[i] : mt;
    double flops = (double)nblk * 4 * iters * 4.0 * 4096.0;
    printf("rep %d: %.3f ms  %.1f TFLOP/s  max ticks %.0f -> %.3f GHz (ticks/us)\n", rep, ms, flops / ms / 1e9, mt, mt / (ms * 1e3) / 1e3);
  }
  return 0;
}
