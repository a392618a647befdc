// Do VALU instructions of ONE wave take v_mfma_f32_32x32x2_f32 issue slots away from ANOTHER wave on the same SIMD
// (gfx950)?  512-thread workgroups, one per CU (150 KB of LDS): waves 0-3 run an MFMA stream (8 independent
// accumulators), waves 4-7 -- same SIMDs -- run a VALU / transcendental / LDS-write / global-store stream.  Each
// role is timed alone and together; if the pipes were independent both would keep their solo time.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_share mfma_valu_share.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(512, 1) void k(float* out, unsigned long long* ticks, int n_mfma, int n_valu, float* sink) {
  __shared__ float lds[150 * 256];
  const int wave = threadIdx.x >> 6;
  lds[threadIdx.x] = 1.0f;
  __syncthreads();
  unsigned long long t0 = 0, t1 = 0;
  float s = 0.0f;
  if (wave < 4) {
    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    const float a = threadIdx.x * 1e-3f, b = 1.0f;
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < n_mfma; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[j][0];
  } else {
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = threadIdx.x * 1e-3f + r;
    float* lp = lds + 512 + (threadIdx.x - 256) * 4;
    float* gp = sink + ((size_t)blockIdx.x * 256 + (threadIdx.x - 256)) * 4;
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < n_valu; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (KIND == 0) v[r] = __builtin_fmaf(v[r], 1.0001f, 0.5f);
        if (KIND == 1) v[r] = __builtin_amdgcn_exp2f(v[r]);
        if (KIND == 2) {
          const float tt = 100.0f * v[r];
          const float z = __builtin_amdgcn_exp2f(fabsf(tt) * -1.44269504f);
          v[r] = (fmaxf(tt, 0.0f) + __builtin_amdgcn_logf(1.0f + z) * 0.69314718f) * 0.01f;
        }
      }
      if (KIND == 3) {   // LDS writes: 4 x ds_write_b128 per iteration
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(lp + q * 1024) = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
      }
      if (KIND == 4) {   // global stores: 4 x dwordx4 per iteration
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(gp + q * 1024) = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
      }
    }
    t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int r = 0; r < 16; ++r) s += v[r];
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int KIND>
void run(const char* name, float* out, unsigned long long* ticks, float* sink, int n_mfma, int n_valu) {
  double res[3][2];
  for (int cfg = 0; cfg < 3; ++cfg) {   // 0 both, 1 MFMA only, 2 VALU only
    const int nm = cfg == 2 ? 0 : n_mfma, nv = cfg == 1 ? 0 : n_valu;
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(512), 0, 0, out, ticks, nm, nv, sink);
      hipDeviceSynchronize();
    }
    unsigned long long h[2048];
    hipMemcpy(h, ticks, 2048 * 8, hipMemcpyDeviceToHost);
    double m = 0, v = 0;
    for (int b = 0; b < 256; ++b)
      for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += (double)h[b * 8 + w];
    res[cfg][0] = m / 1024.0;
    res[cfg][1] = v / 1024.0;
  }
  // s_memtime ticks at 100 MHz; report ns per unit
  printf("%-28s MFMA: alone %7.1f ns/MFMA, with VALU wave %7.1f | VALU: alone %7.1f ns/16-elem group, with MFMA wave %7.1f\n", name,
         res[1][0] * 10.0 / (n_mfma * 8.0), res[0][0] * 10.0 / (n_mfma * 8.0), res[2][1] * 10.0 / n_valu, res[0][1] * 10.0 / n_valu);
}

int main() {
  float *out, *sink;
  unsigned long long* ticks;
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&sink, (size_t)256 * 256 * 4 * 4 * 8);
  hipMalloc(&ticks, 2048 * 8);
  // sized so that both roles run for about the same time (~1.3 ms)
  run<0>("plain fma x16", out, ticks, sink, 5000, 40000);
  run<1>("exp2 x16", out, ticks, sink, 5000, 16000);
  run<2>("softplus x16", out, ticks, sink, 5000, 4000);
  run<3>("fma x16 + 4 ds_write_b128", out, ticks, sink, 5000, 30000);
  run<4>("fma x16 + 4 global stores", out, ticks, sink, 5000, 20000);
  return 0;
}
