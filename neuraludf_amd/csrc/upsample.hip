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

__device__ __forceinline__ float clip01u(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

__device__ __forceinline__ float sdf2alpha_plain(float sdf, float cosv, float dist, float inv_s, int theorical) {
  // sdf2alpha with cos_anneal_ratio=None: iter_cos = true_cos (:298-320)
  if (theorical) {   // sdf2alpha_type == 'theorical' (:321-323): 1 - exp(-relu(|cos| inv_s (1 - sigmoid(sdf inv_s))) dist)
    const float raw = fabsf(cosv) * inv_s * (1.0f - sigmoidf_(sdf * inv_s));
    return 1.0f - expf(-fmaxf(raw, 0.0f) * dist);
  }
  const float en = sdf + cosv * dist * 0.5f;
  const float ep = sdf - cosv * dist * 0.5f;
  const float P = sigmoidf_(ep * inv_s);
  const float Nx = sigmoidf_(en * inv_s);
  return clip01u((P - Nx + 1e-5f) / (P + 1e-5f));
}

template <int NC>
__global__ __launch_bounds__(256) void upsample_kernel(NudfUpsample p) {
  __shared__ float s_z[4][NC * 64];
  __shared__ float s_cdf[4][NC * 64];
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + wave;
  if (ray >= p.N) return;  // no block-level barriers below
  const int M = p.M;
  float* zs = s_z[wave];
  float* cdf = s_cdf[wave];

  const float ox = p.rays_o[ray * 3 + 0], oy = p.rays_o[ray * 3 + 1], oz = p.rays_o[ray * 3 + 2];
  const float dx = p.rays_d[ray * 3 + 0], dy = p.rays_d[ray * 3 + 1], dz = p.rays_d[ray * 3 + 2];
  const float sdist = p.sample_dist[0];
  const float gamma = p.gamma_dev ? p.gamma_dev[0] : p.gamma;

  float z[NC], u[NC], rad[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int i = c * 64 + l;
    z[c] = 0.f; u[c] = 0.f; rad[c] = 1e30f;
    if (i < M) {
      z[c] = p.z[(size_t)ray * M + i];
      u[c] = p.udf[(size_t)ray * M + i];
      const float px = ox + dx * z[c], py = oy + dy * z[c], pz = oz + dz * z[c];
      rad[c] = sqrtf(px * px + py * py + pz * pz);
      zs[i] = z[c];
    }
  }

  // weights w_i for the M-1 sections
  float w[NC];
  const int theorical = (p.mode >> 8) & 1;   // NUDF_UP_THEORICAL
  if ((p.mode & 0xff) == 1) {
    // up_sample_no_occ_aware (:846-858): w = alpha_occ[:, :-1]
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int i = c * 64 + l;
      const float zo = (c + 1 < NC) ? wave_bcast(z[(c + 1 < NC) ? c + 1 : c], 0) : 0.f;
      const float zn = wave_shift_down1(z[c], zo);
      const float dist = (i < M - 1) ? (zn - z[c]) : sdist;
      const float e = expf(-p.beta * u[c]);
      const float raw = p.beta * e / ((1.0f + e) * (1.0f + e)) * gamma;  // udf2logistic(udf, beta, gamma, 1)
      w[c] = (i < M - 1) ? (1.0f - expf(-fmaxf(raw, 0.0f) * dist)) : 0.0f;
    }
  } else {
    // up_sample_unbias (:205-262)
    float tc[NC], cosv[NC], aocc[NC], dists[NC], midu[NC];
    // section slopes
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int i = c * 64 + l;
      const int cn = (c + 1 < NC) ? c + 1 : c;
      const float zo = wave_bcast(z[cn], 0), uo = wave_bcast(u[cn], 0), ro = wave_bcast(rad[cn], 0);
      const float zn = wave_shift_down1(z[c], zo), un = wave_shift_down1(u[c], uo), rn = wave_shift_down1(rad[c], ro);
      const bool sec = i < M - 1;
      dists[c] = sec ? (zn - z[c]) : 0.f;
      midu[c] = (u[c] + un) * 0.5f;
      tc[c] = sec ? (un - u[c]) / (zn - z[c] + 1e-5f) : 0.f;
      const bool inside = (rad[c] < 1.0f) || (rn < 1.0f);
      cosv[c] = inside ? 1.0f : 0.0f;  // holds the inside flag for now
      const float draw = sec ? (zn - z[c]) : sdist;
      const float e = expf(-p.beta * u[c]);
      const float raw = p.beta * e / ((1.0f + e) * (1.0f + e));  // udf2logistic(udf, beta, 1, 1)
      aocc[c] = 1.0f - expf(-fmaxf(raw, 0.0f) * gamma * draw);
    }
    // cos_val = min(prev, cur).clip(-1e3, 0) * inside ; vis_mask_i = (tc_{i-1} < 0.05), first = 1
    float carry = 1.0f;
    float vis[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int i = c * 64 + l;
      const float cur = -fabsf(tc[c]);
      const int cp = (c > 0) ? c - 1 : c;
      const float prev_o = wave_bcast(-fabsf(tc[cp]), 63), tprev_o = wave_bcast(tc[cp], 63);
      const float prev = wave_shift_up1(cur, (c > 0) ? prev_o : 0.0f);
      const float tprev = wave_shift_up1(tc[c], tprev_o);
      const float inside = cosv[c];
      float cv = fminf(prev, cur);
      cv = fminf(fmaxf(cv, -1e3f), 0.0f) * inside;
      cosv[c] = cv;
      const float vm = (i == 0) ? 1.0f : ((tprev < 0.05f) ? 1.0f : 0.0f);
      const float q = (i < M) ? (clip01u(1.0f - aocc[c] + vm) + 1e-7f) : 1.0f;
      float inc = wave_incl_scan_mul(q) * carry;
      const float exc = wave_shift_up1(inc, carry);
      carry = wave_bcast(inc, 63);
      vis[c] = exc;
    }
    carry = 1.0f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int i = c * 64 + l;
      float alpha = 0.f;
      if (i < M - 1) {
        const float ap = sdf2alpha_plain(midu[c], cosv[c], dists[c], p.inv_s, theorical);
        const float am = sdf2alpha_plain(-midu[c], cosv[c], dists[c], p.inv_s, theorical);
        alpha = ap * vis[c] + am * (1.0f - vis[c]);
      }
      const float f = (i < M - 1) ? (1.0f - alpha + 1e-7f) : 1.0f;
      float inc = wave_incl_scan_mul(f) * carry;
      const float exc = wave_shift_up1(inc, carry);
      carry = wave_bcast(inc, 63);
      w[c] = (i < M - 1) ? alpha * exc : 0.0f;
    }
  }

  // sample_pdf (:66-95): w += 1e-5 ; pdf = w / sum ; cdf = [0, cumsum(pdf)]
  float tot = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int i = c * 64 + l;
    w[c] = (i < M - 1) ? (w[c] + 1e-5f) : 0.0f;
    tot += w[c];
  }
  tot = wave_sum(tot);
  float scarry = 0.f;
  if (l == 0) cdf[0] = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int i = c * 64 + l;
    const float pdf = w[c] / tot;
    float inc = wave_incl_scan_add(pdf) + scarry;
    scarry = wave_bcast(inc, 63);
    if (i < M - 1) cdf[i + 1] = inc;
  }
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();

  // inverse CDF at the deterministic quantiles u_k
  for (int k = l; k < p.K; k += 64) {
    const float uq = p.u[k];
    // searchsorted(cdf, u, right=True): number of entries <= u
    int lo = 0, hi = M;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf[mid] <= uq) lo = mid + 1; else hi = mid;
    }
    const int below = max(lo - 1, 0);
    const int above = min(lo, M - 1);
    const float c0 = cdf[below], c1 = cdf[above];
    const float b0 = zs[below], b1 = zs[above];
    float den = c1 - c0;
    if (den < 1e-5f) den = 1.0f;
    const float t = (uq - c0) / den;
    const float zn = b0 + t * (b1 - b0);
    p.z_new[(size_t)ray * p.K + k] = zn;
    if (p.pts_new) {
      const size_t o = ((size_t)ray * p.K + k) * 3;
      p.pts_new[o + 0] = ox + dx * zn;
      p.pts_new[o + 1] = oy + dy * zn;
      p.pts_new[o + 2] = oz + dz * zn;
    }
  }
}

extern "C" int nudf_upsample(const NudfUpsample* args, void* stream) {
  const NudfUpsample& p = *args;
  if (p.N <= 0) return 0;
  if (p.M < 2 || p.M > UP_MAX_M || p.K < 1) {
    nudf_set_error("nudf_upsample: 2 <= M <= 512 and K >= 1 required", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  dim3 grid((p.N + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)stream;
  const int nc = (p.M + 63) / 64;
  switch (nc) {
    case 1: hipLaunchKernelGGL(upsample_kernel<1>, grid, block, 0, st, p); break;
    case 2: hipLaunchKernelGGL(upsample_kernel<2>, grid, block, 0, st, p); break;
    case 3: hipLaunchKernelGGL(upsample_kernel<3>, grid, block, 0, st, p); break;
    case 4: hipLaunchKernelGGL(upsample_kernel<4>, grid, block, 0, st, p); break;
    default: hipLaunchKernelGGL(upsample_kernel<8>, grid, block, 0, st, p); break;
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
