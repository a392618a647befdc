// Hierarchical importance re-sampling: one wavefront per ray, no autograd (the reference runs
// this under torch.no_grad()).
//
//   nudf_upsample : up_sample_unbias / up_sample_no_occ_aware + sample_pdf(det=True)
//                   models/udf_renderer_blending.py:197-272, 834-866, 66-104
//   nudf_merge    : cat + sort + gather of cat_z_vals (:278-288) as a rank-based merge of two
//                   sorted lists that carries the UDF values along
//
// Three wave scans (two exclusive products, one prefix sum) with carries across 64-sample
// chunks; the CDF lives in LDS and the K quantiles are binary-searched lane-parallel.
#include "nudf_common.h"
#include "../../include/nudf.h"

#define UP_MAX_M 512
#define UP_NC (UP_MAX_M / 64)

namespace up_fast {
#include "upsample_body.inc"
}
#pragma clang fp contract(off)
namespace up_exact {
#include "upsample_body.inc"
}
#pragma clang fp contract(fast)

extern "C" int nudf_upsample(const NudfUpsample* args, void* stream) {
  const NudfUpsample& p = *args;
  if (p.N <= 0) return 0;
  if (p.M < 2 || p.M > UP_MAX_M || p.K < 1) {
    nudf_set_error("nudf_upsample: 2 <= M <= 512 and K >= 1 required", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  if (p.merge_K < 0 || p.merge_K >= p.M ||
      (p.merge_K > 0 && (!p.prev_z || !p.prev_udf || !p.add_z || !p.add_udf || !p.z_merged || !p.udf_merged))) {
    nudf_set_error("nudf_upsample: merge_K in [0, M) with all six merge arrays required", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  dim3 grid((p.N + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)stream;
  const int nc = (p.M + 63) / 64;
  int32_t* status = nudf_status_flag();
  if (p.mode & NUDF_UP_NOCONTRACT) {
    switch (nc) {
      case 1: hipLaunchKernelGGL(up_exact::upsample_kernel<1>, grid, block, 0, st, p, status); break;
      case 2: hipLaunchKernelGGL(up_exact::upsample_kernel<2>, grid, block, 0, st, p, status); break;
      case 3: hipLaunchKernelGGL(up_exact::upsample_kernel<3>, grid, block, 0, st, p, status); break;
      case 4: hipLaunchKernelGGL(up_exact::upsample_kernel<4>, grid, block, 0, st, p, status); break;
      default: hipLaunchKernelGGL(up_exact::upsample_kernel<8>, grid, block, 0, st, p, status); break;
    }
  } else {
    switch (nc) {
      case 1: hipLaunchKernelGGL(up_fast::upsample_kernel<1>, grid, block, 0, st, p, status); break;
      case 2: hipLaunchKernelGGL(up_fast::upsample_kernel<2>, grid, block, 0, st, p, status); break;
      case 3: hipLaunchKernelGGL(up_fast::upsample_kernel<3>, grid, block, 0, st, p, status); break;
      case 4: hipLaunchKernelGGL(up_fast::upsample_kernel<4>, grid, block, 0, st, p, status); break;
      default: hipLaunchKernelGGL(up_fast::upsample_kernel<8>, grid, block, 0, st, p, status); break;
    }
  }
  NUDF_CHECK_LAUNCH("nudf_upsample");
  return 0;
}

// ------------------------------------------------------------------------------------------
// merge two ascending lists (a: M entries, b: K entries) per ray; ties keep a before b.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void merge_kernel(const float* __restrict__ za, const float* __restrict__ ua,
                                                    const float* __restrict__ zb, const float* __restrict__ ub, int N,
                                                    int M, int K, float* __restrict__ zo, float* __restrict__ uo) {
  extern __shared__ float sm[];
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + wave;
  if (ray >= N) return;
  float* a = sm + wave * (M + K);
  float* b = a + M;
  for (int i = l; i < M; i += 64) a[i] = za[(size_t)ray * M + i];
  for (int j = l; j < K; j += 64) b[j] = zb[(size_t)ray * K + j];
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  const size_t ob = (size_t)ray * (M + K);
  for (int i = l; i < M; i += 64) {
    const float v = a[i];
    int lo = 0, hi = K;  // count of b strictly less than v
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (b[mid] < v) lo = mid + 1; else hi = mid;
    }
    zo[ob + i + lo] = v;
    if (uo) uo[ob + i + lo] = ua[(size_t)ray * M + i];
  }
  for (int j = l; j < K; j += 64) {
    const float v = b[j];
    int lo = 0, hi = M;  // count of a less than or equal to v
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (a[mid] <= v) lo = mid + 1; else hi = mid;
    }
    zo[ob + j + lo] = v;
    if (uo) uo[ob + j + lo] = ub[(size_t)ray * K + j];
  }
}

extern "C" int nudf_merge(const float* z, const float* udf, const float* z_new, const float* udf_new, int N, int M,
                          int K, float* z_out, float* udf_out, void* stream) {
  if (N <= 0) return 0;
  const size_t lds = (size_t)4 * (M + K) * sizeof(float);
  if (lds > 64 * 1024) {
    nudf_set_error("nudf_merge: M + K too large", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  hipLaunchKernelGGL(merge_kernel, dim3((N + 3) / 4), dim3(256), lds, (hipStream_t)stream, z, udf, z_new, udf_new, N, M,
                     K, z_out, udf_out);
  NUDF_CHECK_LAUNCH("nudf_merge");
  return 0;
}

// ------------------------------------------------------------------------------------------
// the schedule's LAST merge (z only) + the interval mid points o + d (z + dist / 2) of the merged samples
// (ray_points_kernel mode 1) + their mirror into the [.. | pts | 0] columns of the colour network's input rows
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void merge_points_kernel(const float* __restrict__ za, const float* __restrict__ zb,
                                                           int N, int M, int K, float* __restrict__ zo,
                                                           const float* __restrict__ ro, const float* __restrict__ rd,
                                                           const float* __restrict__ sample_dist,
                                                           float* __restrict__ pts, float* __restrict__ xrows, int ldx,
                                                           int xcols) {
  extern __shared__ float sm[];
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + wave;
  if (ray >= N) return;
  const int S = M + K;
  float* a = sm + wave * (5 * S);
  float* b = a + M;
  float* out = a + S;          // merged z
  float* pl = out + S;         // the S points (3 S floats)
  for (int i = l; i < M; i += 64) a[i] = za[(size_t)ray * M + i];
  for (int j = l; j < K; j += 64) b[j] = zb[(size_t)ray * K + j];
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  for (int i = l; i < M; i += 64) {
    const float v = a[i];
    int lo = 0, hi = K;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (b[mid] < v) lo = mid + 1; else hi = mid;
    }
    out[i + lo] = v;
  }
  for (int j = l; j < K; j += 64) {
    const float v = b[j];
    int lo = 0, hi = M;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (a[mid] <= v) lo = mid + 1; else hi = mid;
    }
    out[j + lo] = v;
  }
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  const float ox = ro[ray * 3 + 0], oy = ro[ray * 3 + 1], oz = ro[ray * 3 + 2];
  const float dx = rd[ray * 3 + 0], dy = rd[ray * 3 + 1], dz = rd[ray * 3 + 2];
  const float sd = sample_dist[0];
  const size_t ob = (size_t)ray * S;
  for (int i = l; i < S; i += 64) {
    float t = out[i];
    zo[ob + i] = t;
    const float dist = (i < S - 1) ? (out[i + 1] - t) : sd;
    t = t + dist * 0.5f;
    const float px = ox + dx * t, py = oy + dy * t, pz = oz + dz * t;
    pts[(ob + i) * 3 + 0] = px;
    pts[(ob + i) * 3 + 1] = py;
    pts[(ob + i) * 3 + 2] = pz;
    pl[3 * i + 0] = px; pl[3 * i + 1] = py; pl[3 * i + 2] = pz;
  }
  if (xrows) {
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    for (int e = l; e < S * xcols; e += 64) {      // consecutive lanes = consecutive columns of a row
      const int r = e / xcols, c = e - r * xcols;
      xrows[(ob + r) * (size_t)ldx + c] = (c < 3) ? pl[3 * r + c] : 0.0f;
    }
  }
}

extern "C" int nudf_merge_points(const float* z, const float* z_new, int N, int M, int K, float* z_out, const float* rays_o,
                                 const float* rays_d, const float* sample_dist, float* pts, float* xrows, int ldx, int xcols,
                                 void* stream) {
  if (N <= 0) return 0;
  const size_t lds = (size_t)4 * 5 * (M + K) * sizeof(float);
  if (lds > 64 * 1024 || M < 1 || K < 0 || (xrows && (xcols < 3 || xcols > ldx))) {
    nudf_set_error("nudf_merge_points: M + K <= 819, 3 <= xcols <= ldx required", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  hipLaunchKernelGGL(merge_points_kernel, dim3((N + 3) / 4), dim3(256), lds, (hipStream_t)stream, z, z_new, N, M, K, z_out,
                     rays_o, rays_d, sample_dist, pts, xrows, ldx, xcols);
  NUDF_CHECK_LAUNCH("nudf_merge_points");
  return 0;
}
