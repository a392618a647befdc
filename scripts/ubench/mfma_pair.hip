// What does a k-step of the weight-gradient GEMM lose beside its 64 MFMAs?  (gfx950)  A skeleton of that loop -- 256-thread
// workgroups, wave w on SIMD w, per k-step 64 x v_mfma_f32_32x32x2_f32 on four rotating accumulators -- with the other
// parts added one at a time:
//   0 MFMAs only           1 + the 64 ds_read_b32 operand reads (software-pipelined one group of 16 ahead)
//   2 + a barrier per k-step       3 + the 8 ds_write_b128 of the next k-step's operands
//   4 + their 8 global_load_dwordx4 (a 1 GB stream, issued at the top of the k-step, waited for at the LDS writes)
// each run SOLO (one workgroup per CU) and PAIRED (two per CU = two waves per SIMD, as the GEMM runs).
// Output: shader ticks per MFMA per SIMD (64 = the matrix pipe never idles).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_pair mfma_pair.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <type_traits>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LDT 132
#define TILE (32 * LDT)

template <int KIND>
__global__ __launch_bounds__(256, 2) void k(float* out, unsigned long long* stamps, int nk, const f32x4* big, size_t big_n) {
  __shared__ __attribute__((aligned(16))) float smem[4 * TILE];   // 67.6 KB: two workgroups per CU at most
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 4 * TILE; i += 256) smem[i] = 1e-3f * (i & 255);
  __syncthreads();
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
  const float* as0 = smem + (lane >> 5) * LDT + (wave >> 1) * 64 + (lane & 31);
  const float* bs0 = smem + 2 * TILE + (lane >> 5) * LDT + (wave & 1) * 64 + (lane & 31);
  float av[2][4][2], bv[2][4][2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) { av[s][q][i] = tid * 1e-3f + i; bv[s][q][i] = 1.0f + q; }
  f32x4 ra[4], rb[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) { ra[p] = f32x4{1.f, 2.f, 3.f, 4.f}; rb[p] = ra[p]; }
  size_t gi = (size_t)blockIdx.x * 256 + tid;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (KIND >= 4) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        ra[p] = big[gi % big_n]; gi += (size_t)512 * 256;
        rb[p] = big[gi % big_n]; gi += (size_t)512 * 256;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    const float* as = as0 + cur * TILE;
    const float* bs = bs0 + cur * TILE;
    auto rd = [&](int set, int c) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int kk = c * 4 + q;
#pragma unroll
        for (int i = 0; i < 2; ++i) av[set][q][i] = as[(2 * kk) * LDT + 32 * i];
#pragma unroll
        for (int j = 0; j < 2; ++j) bv[set][q][j] = bs[(2 * kk) * LDT + 32 * j];
      }
    };
    if (KIND >= 1) rd(0, 0);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (KIND >= 1 && c + 1 < 4) rd((c + 1) & 1, c + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c & 1][q][i], bv[c & 1][q][j], acc[i * 2 + j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (KIND >= 3) {
      float* da = smem + (cur ^ 1) * TILE + (tid >> 5) * LDT + (tid & 31) * 4;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        *reinterpret_cast<f32x4*>(da + p * 8 * LDT) = ra[p];
        *reinterpret_cast<f32x4*>(da + 2 * TILE + p * 8 * LDT) = rb[p];
      }
    }
    if (KIND >= 2) __syncthreads();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.0f;
#pragma unroll
  for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][7];
  out[blockIdx.x * 256 + tid] = s;
  if (lane == 0) stamps[blockIdx.x * 4 + wave] = t1 - t0;
}


// 5: the same instruction counts as 4, re-ordered so that no wave ever issues a long run of non-MFMA instructions: one
// LDS read behind every MFMA, the LDS writes spread over the third group of 16 MFMAs, the barrier BEFORE the fourth
// group (whose operands are in registers), the next k-step's first reads and the global loads (for the k-step after the
// next) spread over that fourth group.
__global__ __launch_bounds__(256, 2) void k5(float* out, unsigned long long* stamps, int nk, const f32x4* big, size_t big_n) {
  __shared__ __attribute__((aligned(16))) float smem[4 * TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 4 * TILE; i += 256) smem[i] = 1e-3f * (i & 255);
  __syncthreads();
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
  const float* as0 = smem + (lane >> 5) * LDT + (wave >> 1) * 64 + (lane & 31);
  const float* bs0 = smem + 2 * TILE + (lane >> 5) * LDT + (wave & 1) * 64 + (lane & 31);
  float av[2][4][2], bv[2][4][2];
  f32x4 ra[4], rb[4];
  size_t gi = (size_t)blockIdx.x * 256 + tid;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    ra[p] = big[gi % big_n]; gi += (size_t)512 * 256;
    rb[p] = big[gi % big_n]; gi += (size_t)512 * 256;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int i = 0; i < 2; ++i) { av[0][q][i] = as0[(2 * q) * LDT + 32 * i]; bv[0][q][i] = bs0[(2 * q) * LDT + 32 * i]; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    // group c: 16 MFMAs on operand set c & 1, each followed by one LDS read of group c + 1 (of the next k-step for c = 3)
    auto group = [&](auto C) {
      constexpr int c = decltype(C)::value;
      constexpr int ns = (c + 1) & 1, nc = (c + 1) & 3;
      const float* as = as0 + (c == 3 ? cur ^ 1 : cur) * TILE;
      const float* bs = bs0 + (c == 3 ? cur ^ 1 : cur) * TILE;
      float* da = smem + (cur ^ 1) * TILE + (tid >> 5) * LDT + (tid & 31) * 4;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c & 1][q][i], bv[c & 1][q][j], acc[i * 2 + j], 0, 0, 0);
            const int r = j * 2 + i, kk = nc * 4 + q;
            if (r < 2) av[ns][q][r] = as[(2 * kk) * LDT + 32 * r];
            else bv[ns][q][r - 2] = bs[(2 * kk) * LDT + 32 * (r - 2)];
            const int m = q * 4 + r;   // MFMA index within the group
            if (c == 2 && (m & 1)) {
              const int p = m >> 1;
              if (p < 4) *reinterpret_cast<f32x4*>(da + p * 8 * LDT) = ra[p];
              else *reinterpret_cast<f32x4*>(da + 2 * TILE + (p - 4) * 8 * LDT) = rb[p - 4];
            }
            if (c == 3 && (m & 1)) {
              const int p = m >> 1;
              if (p < 4) ra[p] = big[gi % big_n]; else rb[p - 4] = big[gi % big_n];
              gi += (size_t)512 * 256;
            }
          }
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (c == 2 && (m & 1)) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        if (c == 3 && (m & 1)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    __builtin_amdgcn_sched_barrier(0);
    group(std::integral_constant<int, 0>{});
    group(std::integral_constant<int, 1>{});
    group(std::integral_constant<int, 2>{});
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    group(std::integral_constant<int, 3>{});
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.0f;
#pragma unroll
  for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][7];
  out[blockIdx.x * 256 + tid] = s + ra[0][0] + rb[3][1];
  if (lane == 0) stamps[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int KIND>
void run(const char* name, float* out, unsigned long long* stamps, const f32x4* big, size_t big_n) {
  const int nk = 2000;
  double r[2];
  for (int paired = 0; paired < 2; ++paired) {
    const int wgs = paired ? 512 : 256;
    for (int rep = 0; rep < 3; ++rep) {
      if (KIND == 5) hipLaunchKernelGGL(k5, dim3(wgs), dim3(256), 0, 0, out, stamps, nk, big, big_n);
      else hipLaunchKernelGGL((k<(KIND == 5 ? 4 : KIND)>), dim3(wgs), dim3(256), 0, 0, out, stamps, nk, big, big_n);
      hipDeviceSynchronize();
    }
    static unsigned long long h[2048];
    hipMemcpy(h, stamps, wgs * 4 * 8, hipMemcpyDeviceToHost);
    double t = 0;
    for (int i = 0; i < wgs * 4; ++i) t += (double)h[i];
    r[paired] = t / (wgs * 4) / (nk * 64.0) / (paired ? 2.0 : 1.0);   // ticks per MFMA of the SIMD
  }
  printf("%-58s solo %6.2f  paired %6.2f ticks per MFMA per SIMD  (pipe busy %4.1f %% / %4.1f %%)\n", name, r[0], r[1],
         6400.0 / r[0], 6400.0 / r[1]);
}

int main() {
  float* out;
  unsigned long long* stamps;
  f32x4* big;
  const size_t big_n = (size_t)64 << 20;   // 1 GB
  hipMalloc(&out, 512 * 256 * 4);
  hipMalloc(&stamps, 2048 * 8);
  hipMalloc(&big, big_n * 16);
  hipMemset(big, 0, big_n * 16);
  run<0>("64 MFMAs per k-step", out, stamps, big, big_n);
  run<1>("+ 64 ds_read_b32 (one group of 16 ahead)", out, stamps, big, big_n);
  run<2>("+ barrier per k-step", out, stamps, big, big_n);
  run<3>("+ 8 ds_write_b128 before the barrier", out, stamps, big, big_n);
  run<4>("+ 8 global_load_dwordx4 (HBM stream) at the top", out, stamps, big, big_n);
  run<5>("the same as the last line, interleaved (see k5)", out, stamps, big, big_n);
  return 0;
}
