// What does the composite kernel's ACCESS PATTERN sustain with no arithmetic?  One wave per ray reads z, udf [S] and
// grad, color, color_base [S,3] (12 B per lane, as the kernel does), writes weights [S] and 11 floats per ray.
// Variants: (0) all loads of the ray up-front (the kernel's form); (1) grid-stride persistent waves; (2) plain
// flat float4 read+write of the same byte counts (an ideal streaming kernel).  Prints TB/s of algorithmic bytes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
struct P { const float *z, *u, *g, *c, *cb; float* w; float* o; int N, S; };
template <int NC, bool PERSIST>
__global__ __launch_bounds__(256) void probe(P p) {
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
  int ray = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
  const int stride = gridDim.x * 4;
  for (; ray < p.N; ray += PERSIST ? stride : p.N) {
    const size_t b = (size_t)ray * p.S;
    const float* z = p.z + b; const float* u = p.u + b;
    const float* g = p.g + b * 3; const float* c = p.c + b * 3; const float* cb = p.cb + b * 3;
    float acc[NC];
    float v[NC][11];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      const unsigned i = k * 64 + l;
      v[k][0] = z[i]; v[k][1] = u[i];
      v[k][2] = g[i * 3]; v[k][3] = g[i * 3 + 1]; v[k][4] = g[i * 3 + 2];
      v[k][5] = c[i * 3]; v[k][6] = c[i * 3 + 1]; v[k][7] = c[i * 3 + 2];
      v[k][8] = cb[i * 3]; v[k][9] = cb[i * 3 + 1]; v[k][10] = cb[i * 3 + 2];
    }
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      acc[k] = 0.f;
#pragma unroll
      for (int q = 0; q < 11; ++q) acc[k] += v[k][q];
      p.w[b + k * 64 + l] = acc[k];
      tot += acc[k];
    }
    if (l < 11) p.o[(size_t)ray * 11 + l] = tot;
  }
}
__global__ __launch_bounds__(256) void flat(const float4* __restrict__ in, float4* __restrict__ out, size_t n_in, size_t n_out) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  float4 s = {0, 0, 0, 0};
  for (size_t j = i; j < n_in; j += stride) { float4 t = in[j]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
  for (size_t j = i; j < n_out; j += stride) out[j] = s;
}
int main() {
  const int N = 32768, S = 256;
  P p; p.N = N; p.S = S;
  float *z, *u, *g, *c, *cb, *w, *o;
  size_t ns = (size_t)N * S;
  hipMalloc(&z, ns * 4); hipMalloc(&u, ns * 4); hipMalloc(&g, ns * 12); hipMalloc(&c, ns * 12); hipMalloc(&cb, ns * 12);
  hipMalloc(&w, ns * 4); hipMalloc(&o, (size_t)N * 11 * 4);
  hipMemset(z, 0, ns * 4); hipMemset(u, 0, ns * 4); hipMemset(g, 0, ns * 12); hipMemset(c, 0, ns * 12); hipMemset(cb, 0, ns * 12);
  p.z = z; p.u = u; p.g = g; p.c = c; p.cb = cb; p.w = w; p.o = o;
  const double bytes = 48.0 * ns + 44.0 * N;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float* big; hipMalloc(&big, ns * 44); hipMemset(big, 0, ns * 44);
  for (int var = 0; var < 4; ++var) {
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      hipEventRecord(e0);
      if (var == 0) hipLaunchKernelGGL((probe<4, false>), dim3(N / 4), dim3(256), 0, 0, p);
      else if (var == 1) hipLaunchKernelGGL((probe<4, true>), dim3(256 * 7), dim3(256), 0, 0, p);
      else if (var == 2) hipLaunchKernelGGL((probe<4, true>), dim3(256 * 4), dim3(256), 0, 0, p);
      else hipLaunchKernelGGL(flat, dim3(256 * 8), dim3(256), 0, 0, (const float4*)big, (float4*)w, ns * 11 / 4, ns / 4);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best) best = ms;
    }
    const char* names[] = {"wave per ray, one block per 4 rays", "persistent, 7 blocks/CU", "persistent, 4 blocks/CU", "flat float4 stream (44 B in, 4 B out per sample)"};
    printf("%-52s %.1f us  %.2f TB/s (%.1f %% of 8)\n", names[var], best * 1e3, bytes / best / 1e9, bytes / best / 8e7);
  }
  return 0;
}
