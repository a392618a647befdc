// Which companion work lowers the shader clock under an fp32 MFMA stream (gfx950)?  512-thread workgroups, one per CU:
// waves 0-3 run v_mfma_f32_32x32x2_f32 back to back, waves 4-7 (same SIMDs) run a companion stream:
//   0 nothing, 1 plain fma, 2 exp2, 3 LDS reads (ds_read_b128), 4 HBM streaming loads (global_load_dwordx4 over 1 GB),
//   5 L2-resident streaming loads (a 256 KB buffer re-read by every CU, like the chains' weight fragments), 6 = 3 + 5 + 1
// The clock is (s_memtime ticks) / (wall_clock64 ticks at 100 MHz) of the MFMA waves.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_clock mfma_clock.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(512, 1) void k(float* out, unsigned long long* stamps, int n_mfma, int n_comp, const f32x4* big,
                                             size_t big_n) {
  __shared__ f32x4 lds[150 * 64];
  const int wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 150 * 64; i += 512) lds[i] = f32x4{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  float s = 0.0f;
  if (wave < 4) {
    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    const float a = threadIdx.x * 1e-3f, b = 1.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
#pragma unroll 1
    for (int i = 0; i < n_mfma; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[j][0];
    if ((threadIdx.x & 63) == 0) {
      stamps[(blockIdx.x * 4 + wave) * 2] = t1 - t0;
      stamps[(blockIdx.x * 4 + wave) * 2 + 1] = w1 - w0;
    }
  } else if (KIND != 0) {
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = threadIdx.x * 1e-3f + r;
    f32x4 accv = {0.f, 0.f, 0.f, 0.f};
    const int l = threadIdx.x - 256;
    size_t gi = ((size_t)blockIdx.x * 256 + l);
#pragma unroll 1
    for (int i = 0; i < n_comp; ++i) {
      if (KIND == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = __builtin_fmaf(v[r], 1.0001f, 0.5f);
      } else if (KIND == 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = __builtin_amdgcn_exp2f(v[r]);
      } else if (KIND == 3) {
#pragma unroll
        for (int r = 0; r < 8; ++r) accv += lds[(l + 37 * r + i) % (150 * 64)];
      } else if (KIND == 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          accv += big[gi % big_n];
          gi += (size_t)256 * 256;
        }
      } else {   // 5, 6: 256 KB = 16384 float4, every CU reads the same buffer
#pragma unroll
        for (int r = 0; r < 4; ++r) accv += big[(size_t)((l + 256 * (4 * i + r)) & 16383)];
        if (KIND == 6) {
#pragma unroll
          for (int r = 0; r < 4; ++r) accv += lds[(l + 37 * r + i) % (150 * 64)];
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = __builtin_fmaf(v[r], 1.0001f, 0.5f);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) s += v[r];
    s += accv[0] + accv[1] + accv[2] + accv[3];
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, float* out, unsigned long long* stamps, const f32x4* big, size_t big_n, int n_comp) {
  const int n_mfma = 20000;   // ~5 ms at 64 cycles per MFMA
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(512), 0, 0, out, stamps, n_mfma, n_comp, big, big_n);
    hipDeviceSynchronize();
  }
  unsigned long long h[2048];
  hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost);
  double t = 0, w = 0;
  for (int i = 0; i < 1024; ++i) { t += (double)h[2 * i]; w += (double)h[2 * i + 1]; }
  printf("%-34s %6.1f ticks per MFMA, shader clock %6.0f MHz, %6.1f TFLOP/s\n", name, t / 1024 / (n_mfma * 8.0), t / w * 100.0,
         1024.0 * n_mfma * 8 * 4096.0 / (w / 1024 / 100e6) / 1e12);
}

int main() {
  float* out;
  unsigned long long* stamps;
  f32x4* big;
  const size_t big_n = (size_t)64 << 20;   // 1 GB
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&stamps, 2048 * 8);
  hipMalloc(&big, big_n * 16);
  hipMemset(big, 0, big_n * 16);
  run<0>("MFMA alone", out, stamps, big, big_n, 0);
  run<1>("MFMA + plain fma wave", out, stamps, big, big_n, 160000);
  run<2>("MFMA + exp2 wave", out, stamps, big, big_n, 60000);
  run<3>("MFMA + LDS read wave", out, stamps, big, big_n, 100000);
  run<4>("MFMA + HBM streaming wave", out, stamps, big, big_n, 40000);
  run<5>("MFMA + L2 streaming wave", out, stamps, big, big_n, 120000);
  run<6>("MFMA + L2 + LDS + fma wave", out, stamps, big, big_n, 80000);
  return 0;
}
