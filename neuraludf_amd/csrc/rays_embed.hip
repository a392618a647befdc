// Ray points, positional encodings (forward / JVP / VJP) and the small per-point head
// kernels around the MLP GEMM chains.  All HBM-bound elementwise work, one thread per
// output element so that stores are coalesced.
#include "nudf_common.h"
#include "../../include/nudf.h"

static inline int nblocks(long long n, int bs) { return (int)((n + bs - 1) / bs); }

// ---------------------------------------------------------------------------------------
// coarse samples: z = near + (far-near)*linspace(0,1,S) + t_rand*2/S
// (models/udf_renderer_blending.py:606-608, 617-619); sample_dist = mean((far-near)/S) (:605)
// ---------------------------------------------------------------------------------------
__global__ void coarse_z_kernel(const float* near, const float* far, int nf_stride, const float* t_rand,
                                int N, int S, float* z) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * S) return;
  int r = idx / S, s = idx - r * S;
  float nr = near[r * nf_stride], fr = far[r * nf_stride];
  // torch.linspace(0,1,S): i*step for the first half, 1-(S-1-i)*step for the second
  float step = 1.0f / (float)(S - 1);
  float t = (s < S / 2) ? s * step : 1.0f - (S - 1 - s) * step;
  float v = nr + (fr - nr) * t;
  if (t_rand) v = v + t_rand[r] * 2.0f / (float)S;
  z[idx] = v;
}

__global__ void sample_dist_kernel(const float* near, const float* far, int n, int S, float* out) {
  // single block: mean over n of (far-near)/S
  __shared__ float red[256];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) acc += (far[i] - near[i]) / (float)S;
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0] / (float)n;
}

extern "C" int nudf_coarse_z(const float* near, const float* far, int nf_stride, const float* t_rand, int N, int S,
                             float* z, float* sample_dist, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(coarse_z_kernel, dim3(nblocks((long long)N * S, 256)), dim3(256), 0, st, near, far, nf_stride,
                     t_rand, N, S, z);
  if (sample_dist)
    hipLaunchKernelGGL(sample_dist_kernel, dim3(1), dim3(256), 0, st, near, far, nf_stride ? N : 1, S, sample_dist);
  NUDF_CHECK_LAUNCH("nudf_coarse_z");
  return 0;
}

// coarse_z + sample_dist + ray_points(mode 0) as one launch (the three were ~5 us each on the critical path of a step, a
// fourth launch subtracted the 0.5 from the jitter): same expressions, block 0 also reduces the mean spacing.
__global__ __launch_bounds__(256) void coarse_start_kernel(const float* near, const float* far, int nf_stride,
                                                           const float* t_rand, int center, int N, int S, float* z,
                                                           float* sample_dist, const float* __restrict__ o,
                                                           const float* __restrict__ d, float* __restrict__ pts) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < N * S) {
    const int r = idx / S, s = idx - r * S;
    const float nr = near[r * nf_stride], fr = far[r * nf_stride];
    const float step = 1.0f / (float)(S - 1);
    const float t = (s < S / 2) ? s * step : 1.0f - (S - 1 - s) * step;
    float v = nr + (fr - nr) * t;
    if (t_rand) {
      const float tr = center ? __fsub_rn(t_rand[r], 0.5f) : t_rand[r];
      v = v + tr * 2.0f / (float)S;
    }
    z[idx] = v;
    if (pts) {
      pts[(size_t)idx * 3 + 0] = o[r * 3 + 0] + d[r * 3 + 0] * v;
      pts[(size_t)idx * 3 + 1] = o[r * 3 + 1] + d[r * 3 + 1] * v;
      pts[(size_t)idx * 3 + 2] = o[r * 3 + 2] + d[r * 3 + 2] * v;
    }
  }
  if (blockIdx.x == 0 && sample_dist) {      // sample_dist_kernel's order: 256 strided partial sums, LDS tree
    __shared__ float red[256];
    const int n = nf_stride ? N : 1;
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) acc += (far[i] - near[i]) / (float)S;
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
      __syncthreads();
    }
    if (threadIdx.x == 0) sample_dist[0] = red[0] / (float)n;
  }
}
extern "C" int nudf_coarse_start(const float* near, const float* far, int nf_stride, const float* t_rand, int center, int N,
                                 int S, float* z, float* sample_dist, const float* rays_o, const float* rays_d, float* pts,
                                 void* stream) {
  if (N <= 0 || S < 2) {
    nudf_set_error("nudf_coarse_start: N >= 1 and S >= 2 required", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  hipLaunchKernelGGL(coarse_start_kernel, dim3(nblocks((long long)N * S, 256)), dim3(256), 0, (hipStream_t)stream, near, far,
                     nf_stride, t_rand, center, N, S, z, sample_dist, rays_o, rays_d, pts);
  NUDF_CHECK_LAUNCH("nudf_coarse_start");
  return 0;
}

// outside samples (models/udf_renderer_blending.py:611, 621-630):
//   lin = linspace(1e-3, 1-1/(n_out+1), n_out) [optionally stratified-jittered, done by caller]
//   z_out = far / flip(lin) + 1/n_samples
__global__ void outside_z_kernel(const float* far, int f_stride, const float* lin, int N, int n_out, float inv_ns,
                                 float* z_out) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * n_out) return;
  int r = idx / n_out, j = idx - r * n_out;
  z_out[idx] = far[r * f_stride] / lin[n_out - 1 - j] + inv_ns;
}
extern "C" int nudf_outside_z(const float* far, int f_stride, const float* lin, int N, int n_out, int n_samples,
                              float* z_out, void* stream) {
  hipLaunchKernelGGL(outside_z_kernel, dim3(nblocks((long long)N * n_out, 256)), dim3(256), 0, (hipStream_t)stream,
                     far, f_stride, lin, N, n_out, 1.0f / (float)n_samples, z_out);
  NUDF_CHECK_LAUNCH("nudf_outside_z");
  return 0;
}

// ---------------------------------------------------------------------------------------
// points along rays.  mode 0: end points o + d*z (up-sampling code, :205, 277, 729)
//                     mode 1: interval mid points o + d*(z + dist/2) (:352-357, 164-169),
//                             dist = z[i+1]-z[i], last = sample_dist
//                     mode 2: mode 1 + NeRF++ inverted sphere (x/r, 1/r), r = clip(|x|,1,1e10) (:171-173)
// ---------------------------------------------------------------------------------------
__global__ void ray_points_kernel(const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ z,
                                  const float* sample_dist, int N, int S, int mode, float* __restrict__ pts) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * S) return;
  int r = idx / S, s = idx - r * S;
  float t = z[idx];
  if (mode >= 1) {
    float dist = (s < S - 1) ? (z[idx + 1] - t) : sample_dist[0];
    t = t + dist * 0.5f;
  }
  float px = o[r * 3 + 0] + d[r * 3 + 0] * t;
  float py = o[r * 3 + 1] + d[r * 3 + 1] * t;
  float pz = o[r * 3 + 2] + d[r * 3 + 2] * t;
  if (mode == 2) {
    float rr = sqrtf(px * px + py * py + pz * pz);
    rr = fminf(fmaxf(rr, 1.0f), 1e10f);
    pts[(size_t)idx * 4 + 0] = px / rr;
    pts[(size_t)idx * 4 + 1] = py / rr;
    pts[(size_t)idx * 4 + 2] = pz / rr;
    pts[(size_t)idx * 4 + 3] = 1.0f / rr;
  } else {
    pts[(size_t)idx * 3 + 0] = px;
    pts[(size_t)idx * 3 + 1] = py;
    pts[(size_t)idx * 3 + 2] = pz;
  }
}
extern "C" int nudf_ray_points(const float* rays_o, const float* rays_d, const float* z, const float* sample_dist, int N,
                               int S, int mode, float* pts, void* stream) {
  if (N * S == 0) return 0;
  hipLaunchKernelGGL(ray_points_kernel, dim3(nblocks((long long)N * S, 256)), dim3(256), 0, (hipStream_t)stream, rays_o,
                     rays_d, z, sample_dist, N, S, mode, pts);
  NUDF_CHECK_LAUNCH("nudf_ray_points");
  return 0;
}

// ---------------------------------------------------------------------------------------
// positional encoding [x, sin(2^k x), cos(2^k x)]_k  (models/embedder.py:15-36), written into
// up to two destinations (column offset + scale each): e.g. the UDF net's layer-0 input and the
// skip-concat tail of its layer-4 input (/sqrt(2), fields.py:202-203).
// mode 0: value ; mode 1: JVP with tangent v (d/dx applied to v).
// x row for point p is x[(p / xdiv) * xld .. +D]  (xdiv = S broadcasts per-ray directions)
// ---------------------------------------------------------------------------------------
__global__ void posenc_kernel(const float* __restrict__ x, int xld, int xdiv, const float* __restrict__ v, int D, int L,
                              float in_scale, int P, float* d1, int ld1, float s1, float* d2, int ld2, float s2,
                              int mode) {
  const int E = D * (2 * L + 1);
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)P * E) return;
  int p = (int)(idx / E), e = (int)(idx - (long long)p * E);
  int blk = e / D, j = e - blk * D;
  float xv = x[(size_t)(p / xdiv) * xld + j] * in_scale;
  float val;
  if (blk == 0) {
    val = (mode == 0) ? xv : v[(size_t)p * D + j] * in_scale;
  } else {
    int k = (blk - 1) >> 1;
    float f = (float)(1 << k);
    float a = xv * f;
    bool is_sin = ((blk - 1) & 1) == 0;
    if (mode == 0) val = is_sin ? sinf(a) : cosf(a);
    else val = (is_sin ? cosf(a) : -sinf(a)) * f * v[(size_t)p * D + j] * in_scale;
  }
  if (d1) d1[(size_t)p * ld1 + e] = val * s1;
  if (d2) d2[(size_t)p * ld2 + e] = val * s2;
}
extern "C" int nudf_posenc(const float* x, int xld, int xdiv, const float* tangent, int D, int L, float in_scale, int P,
                           float* dst1, int ld1, float scale1, float* dst2, int ld2, float scale2, void* stream) {
  if (P == 0) return 0;
  const int E = D * (2 * L + 1);
  hipLaunchKernelGGL(posenc_kernel, dim3(nblocks((long long)P * E, 256)), dim3(256), 0, (hipStream_t)stream, x, xld,
                     xdiv, tangent, D, L, in_scale, P, dst1, ld1, scale1, dst2, ld2, scale2, tangent ? 1 : 0);
  NUDF_CHECK_LAUNCH("nudf_posenc");
  return 0;
}

// VJP of the encoding: g[p, j] = in_scale * sum_e dE[p,e] * dE_e/dx_j, dE gathered from up to two
// sources (scaled).  One block per 64 points: the dE rows are staged in LDS with coalesced reads (a thread
// walking its own row touches 64 different cache lines per load), then one thread per (point, dimension).
#define PV_PTS 64
#define PV_MAXE 96
__global__ __launch_bounds__(256) void posenc_vjp_kernel(const float* __restrict__ x, int xld, int D, int L,
                                                         float in_scale, int P, const float* __restrict__ s1, int ld1,
                                                         float c1, const float* __restrict__ s2, int ld2, float c2,
                                                         float* __restrict__ g) {
  __shared__ float de[PV_PTS][PV_MAXE + 1];
  const int E = D * (2 * L + 1);
  const int p0 = blockIdx.x * PV_PTS;
  for (int idx = threadIdx.x; idx < PV_PTS * E; idx += 256) {
    const int r = idx / E, e = idx - r * E;
    const int p = min(p0 + r, P - 1);
    float t = 0.f;
    if (s1) t += s1[(size_t)p * ld1 + e] * c1;
    if (s2) t += s2[(size_t)p * ld2 + e] * c2;
    de[r][e] = t;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < PV_PTS * D; idx += 256) {
    const int r = idx / D, j = idx - r * D;
    const int p = p0 + r;
    if (p >= P) continue;
    const float xv = x[(size_t)p * xld + j] * in_scale;
    float acc = de[r][j];
    for (int k = 0; k < L; ++k) {
      const float f = (float)(1 << k);
      float sn, cs;
      sincosf(xv * f, &sn, &cs);
      acc += f * (de[r][D * (1 + 2 * k) + j] * cs - de[r][D * (2 + 2 * k) + j] * sn);
    }
    g[(size_t)p * D + j] = acc * in_scale;
  }
}
extern "C" int nudf_posenc_vjp(const float* x, int xld, int D, int L, float in_scale, int P, const float* src1, int ld1,
                               float scale1, const float* src2, int ld2, float scale2, float* g, void* stream) {
  if (P == 0) return 0;
  if (D * (2 * L + 1) > PV_MAXE) {
    nudf_set_error("nudf_posenc_vjp: encoding wider than 96", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  hipLaunchKernelGGL(posenc_vjp_kernel, dim3(nblocks(P, PV_PTS)), dim3(256), 0, (hipStream_t)stream, x, xld, D, L,
                     in_scale, P, src1, ld1, scale1, src2, ld2, scale2, g);
  NUDF_CHECK_LAUNCH("nudf_posenc_vjp");
  return 0;
}

// ---------------------------------------------------------------------------------------
// column block copy: dst[p, 0:ncols] = src[(p/sdiv), 0:ncols] * scale   (pts -> colour-net input etc.)
// ---------------------------------------------------------------------------------------
__global__ void copy_cols_kernel(const float* __restrict__ src, int lds, int sdiv, float* __restrict__ dst, int ldd,
                                 int ncols, int P, float scale) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)P * ncols) return;
  int p = (int)(idx / ncols), c = (int)(idx - (long long)p * ncols);
  dst[(size_t)p * ldd + c] = src[(size_t)(p / sdiv) * lds + c] * scale;
}
extern "C" int nudf_copy_cols(const float* src, int lds, int sdiv, float* dst, int ldd, int ncols, int P, float scale,
                              void* stream) {
  if (P == 0) return 0;
  hipLaunchKernelGGL(copy_cols_kernel, dim3(nblocks((long long)P * ncols, 256)), dim3(256), 0, (hipStream_t)stream, src,
                     lds, sdiv, dst, ldd, ncols, P, scale);
  NUDF_CHECK_LAUNCH("nudf_copy_cols");
  return 0;
}

// ---------------------------------------------------------------------------------------
// seed of the reverse sweep for d udf / d x (fields.py:219-231 via autograd):
//   da_last_hidden[p, c] = sign[p] * W_last[0, c] * inv_scale * softplus'(a)[p, c]
// ---------------------------------------------------------------------------------------
__global__ void udf_grad_seed_kernel(const float* __restrict__ sign, const float* __restrict__ wrow,
                                     const float* __restrict__ h, int ldh, float hscale, int P, int C,
                                     float inv_scale, float* __restrict__ out, int ldo) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)P * C) return;
  int p = (int)(idx / C), c = (int)(idx - (long long)p * C);
  // softplus' recovered from the stored activation (see sp_derivs_from_h in gemm_f32_mfma.hip)
  const float x = 100.0f * hscale * h[(size_t)p * ldh + c];
  float sg = 1.0f;
  if (x <= 20.0f) sg = (x < 0.01f) ? x * (1.0f - x * (0.5f - x * 0.16666667f)) : 1.0f - __expf(-x);
  out[(size_t)p * ldo + c] = sign[p] * wrow[c] * inv_scale * sg;
}
extern "C" int nudf_udf_grad_seed(const float* sign, const float* w_row0, const float* h, int ldh, float hscale, int P,
                                  int C, float inv_scale, float* out, int ldo, void* stream) {
  if (P == 0) return 0;
  hipLaunchKernelGGL(udf_grad_seed_kernel, dim3(nblocks((long long)P * C, 256)), dim3(256), 0, (hipStream_t)stream, sign,
                     w_row0, h, ldh, hscale, P, C, inv_scale, out, ldo);
  NUDF_CHECK_LAUNCH("nudf_udf_grad_seed");
  return 0;
}

// adjoint of the UDF output layer: abar[p,0] = sign[p]*scale*dudf[p] ; abar[p,1+c] = dfeat[p,c]
__global__ void udf_head_bwd_kernel(const float* __restrict__ sign, const float* __restrict__ dudf,
                                    const float* __restrict__ dfeat, int ldf, int P, int F, float scale,
                                    float* __restrict__ out, int ldo) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int C = F + 1;
  if (idx >= (long long)P * C) return;
  int p = (int)(idx / C), c = (int)(idx - (long long)p * C);
  float v;
  if (c == 0) v = dudf ? sign[p] * scale * dudf[p] : 0.f;
  else v = dfeat ? dfeat[(size_t)p * ldf + (c - 1)] : 0.f;
  out[(size_t)p * ldo + c] = v;
}
extern "C" int nudf_udf_head_bwd(const float* sign, const float* dudf, const float* dfeat, int ldf, int P, int F,
                                 float scale, float* out, int ldo, void* stream) {
  if (P == 0) return 0;
  hipLaunchKernelGGL(udf_head_bwd_kernel, dim3(nblocks((long long)P * (F + 1), 256)), dim3(256), 0, (hipStream_t)stream,
                     sign, dudf, dfeat, ldf, P, F, scale, out, ldo);
  NUDF_CHECK_LAUNCH("nudf_udf_head_bwd");
  return 0;
}

// sign-weighted column sums: out[c] += sum_p sign[p] * R[p, c]  (gradient of row 0 of the last
// UDF layer through the d udf/dx path).  Block = 64 rows x C columns, 4 row groups of 16 rows with the
// loads of a group independent (the old 256-row serial walk per thread was latency-bound at 100 us).
__global__ __launch_bounds__(256) void signed_colsum_kernel(const float* __restrict__ sign, const float* __restrict__ R,
                                                            int ldr, int P, int C, float scale,
                                                            float* __restrict__ out) {
  __shared__ float part[4][64];
  const int c0 = blockIdx.y * 64;
  const int c = c0 + (threadIdx.x & 63);
  const int grp = threadIdx.x >> 6;
  const int p0 = blockIdx.x * 256 + grp * 64;
  float acc = 0.f;
  if (c < C) {
#pragma unroll 8
    for (int i = 0; i < 64; ++i) {
      const int p = p0 + i;
      if (p < P) acc += sign[p] * R[(size_t)p * ldr + c];
    }
  }
  part[grp][threadIdx.x & 63] = acc;
  __syncthreads();
  if (threadIdx.x < 64 && c < C)
    atomicAdd(out + c, (part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]) * scale);
}
extern "C" int nudf_signed_colsum(const float* sign, const float* R, int ldr, int P, int C, float scale, float* out,
                                  void* stream) {
  if (P == 0) return 0;
  hipLaunchKernelGGL(signed_colsum_kernel, dim3(nblocks(P, 256), (C + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                     sign, R, ldr, P, C, scale, out);
  NUDF_CHECK_LAUNCH("nudf_signed_colsum");
  return 0;
}

// backward of a sigmoid head: out[p,c] = (c < nsig) ? dy*y*(1-y) (+ extra) : draw ; columns [nsig+nraw, ldo)
// are written as zeros (out is the A operand of a GEMM whose K is padded to ldo)
__global__ void sigmoid_head_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                        const float* __restrict__ dy_extra, int ldx, int nsig,
                                        const float* __restrict__ draw, int ldr, int nraw, int P,
                                        float* __restrict__ out, int ldo) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)P * ldo) return;
  int p = (int)(idx / ldo), c = (int)(idx - (long long)p * ldo);
  float v = 0.f;
  if (c < nsig) {
    float s = y[(size_t)p * nsig + c];
    float d = dy ? dy[(size_t)p * nsig + c] : 0.f;
    if (dy_extra) d += dy_extra[(size_t)p * ldx + c];
    v = d * s * (1.0f - s);
  } else if (c < nsig + nraw) {
    v = draw ? draw[(size_t)p * ldr + (c - nsig)] : 0.f;
  }
  out[idx] = v;
}
extern "C" int nudf_sigmoid_head_bwd(const float* y, const float* dy, const float* dy_extra, int ldx, int nsig,
                                     const float* draw, int ldr, int nraw, int P, float* out, int ldo, void* stream) {
  if (P == 0) return 0;
  hipLaunchKernelGGL(sigmoid_head_bwd_kernel, dim3(nblocks((long long)P * ldo, 256)), dim3(256), 0,
                     (hipStream_t)stream, y, dy, dy_extra, ldx, nsig, draw, ldr, nraw, P, out, ldo);
  NUDF_CHECK_LAUNCH("nudf_sigmoid_head_bwd");
  return 0;
}

// out[p,c] = a[p,c] + b[p,c]  (adjoint joins, e.g. the hidden tap of the colour net)
__global__ void add_cols_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
                                float* __restrict__ out, int ldo, int P, int C) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)P * C) return;
  int p = (int)(idx / C), c = (int)(idx - (long long)p * C);
  out[(size_t)p * ldo + c] = a[(size_t)p * lda + c] + b[(size_t)p * ldb + c];
}
extern "C" int nudf_add_cols(const float* a, int lda, const float* b, int ldb, float* out, int ldo, int P, int C,
                             void* stream) {
  if (P == 0) return 0;
  hipLaunchKernelGGL(add_cols_kernel, dim3(nblocks((long long)P * C, 256)), dim3(256), 0, (hipStream_t)stream, a, lda, b,
                     ldb, out, ldo, P, C);
  NUDF_CHECK_LAUNCH("nudf_add_cols");
  return 0;
}

// ---------------------------------------------------------------------------------------
// weight_norm packing: W = g * v / ||v||_row  (torch.nn.utils.weight_norm, applied at
// fields.py:175-176, 433-446), written zero-padded into the two layouts the GEMMs read:
//   W  [out_pad, in_pad]  (backward-data GEMMs)   and   Wt [in_pad, out_pad]  (forward GEMMs),
// with an optional input-column permutation (activation buffers use their own column order).
// One wave per output row.  g == NULL: plain nn.Linear (NeRF, fields.py:577-594).
// ---------------------------------------------------------------------------------------
__global__ void wn_pack_kernel(const float* __restrict__ v, const float* __restrict__ g, int out, int in,
                               const int* __restrict__ perm, float* __restrict__ W, int ldw, float* __restrict__ Wt,
                               int ldwt, float* __restrict__ inv_norm) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= out) return;
  const int l = threadIdx.x & 63;
  float sc = 1.0f;
  if (g) {
    float ss = 0.f;
    for (int i = l; i < in; i += 64) {
      float t = v[(size_t)row * in + i];
      ss += t * t;
    }
    ss = wave_sum(ss);
    float inv = 1.0f / sqrtf(ss);
    if (l == 0 && inv_norm) inv_norm[row] = inv;
    sc = g[row] * inv;
  }
  for (int i = l; i < in; i += 64) {
    int c = perm ? perm[i] : i;
    float w = v[(size_t)row * in + i] * sc;
    if (W) W[(size_t)row * ldw + c] = w;
    if (Wt) Wt[(size_t)c * ldwt + row] = w;
  }
}
extern "C" int nudf_weightnorm_pack(const float* v, const float* g, int out, int in, const int* perm, float* W, int ldw,
                                    float* Wt, int ldwt, float* inv_norm, void* stream) {
  hipLaunchKernelGGL(wn_pack_kernel, dim3((out + 3) / 4), dim3(256), 0, (hipStream_t)stream, v, g, out, in, perm, W, ldw,
                     Wt, ldwt, inv_norm);
  NUDF_CHECK_LAUNCH("nudf_weightnorm_pack");
  return 0;
}

// backward of the packing: dW [out, ldw] (packed column order) -> dv [out,in], dg [out]
//   dg = <dW, v> * inv ;  dv = g*inv * (dW - v*inv * dg*inv ... ) = g*inv*dW - v * (g*inv^3 * <dW,v>)
__global__ void wn_unpack_grad_kernel(const float* __restrict__ dW, int ldw, const float* __restrict__ v,
                                      const float* __restrict__ g, const float* __restrict__ inv_norm, int out, int in,
                                      const int* __restrict__ perm, float* __restrict__ dv, float* __restrict__ dg) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= out) return;
  const int l = threadIdx.x & 63;
  if (!g) {
    for (int i = l; i < in; i += 64) dv[(size_t)row * in + i] = dW[(size_t)row * ldw + (perm ? perm[i] : i)];
    return;
  }
  float dot = 0.f;
  for (int i = l; i < in; i += 64) dot += dW[(size_t)row * ldw + (perm ? perm[i] : i)] * v[(size_t)row * in + i];
  dot = wave_sum(dot);
  const float inv = inv_norm[row];
  const float gg = g[row];
  if (l == 0) dg[row] = dot * inv;
  const float c1 = gg * inv, c2 = gg * inv * inv * inv * dot;
  for (int i = l; i < in; i += 64)
    dv[(size_t)row * in + i] = c1 * dW[(size_t)row * ldw + (perm ? perm[i] : i)] - c2 * v[(size_t)row * in + i];
}
extern "C" int nudf_weightnorm_unpack_grad(const float* dW, int ldw, const float* v, const float* g,
                                           const float* inv_norm, int out, int in, const int* perm, float* dv, float* dg,
                                           void* stream) {
  hipLaunchKernelGGL(wn_unpack_grad_kernel, dim3((out + 3) / 4), dim3(256), 0, (hipStream_t)stream, dW, ldw, v, g,
                     inv_norm, out, in, perm, dv, dg);
  NUDF_CHECK_LAUNCH("nudf_weightnorm_unpack_grad");
  return 0;
}

// ---------------------------------------------------------------------------------------
// multi-layer versions (one launch per network): table by value, one wave per (layer, output row)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void frag_store(const NudfPackFrag& f, int o, int c, float w) {
  // element (o, c) of the packed [out, in] matrix -> its slot in the fragment-ordered operand
  const int k = f.transpose ? (c - f.i0) : (o - f.o0);
  const int n = f.transpose ? (o - f.o0) : (c - f.i0);
  if (k < 0 || n < 0 || k >= f.K || n >= f.N) return;
  const int NT = (f.N + 31) >> 5;
  if (f.dtype == 0) {
    const int g = k >> 3, r = k & 7;
    const int lane = 32 * (r >> 2) + (n & 31);
    f.dst[((size_t)(g * NT + (n >> 5)) * 64 + lane) * 4 + (r & 3)] = w;
  } else if (f.dtype == 3) {
    // bf16x3 split fragments (fp32 emulated on the bf16 matrix pipe): w = hi + mid + lo EXACTLY, each part the bf16
    // nearest to what the previous parts left (w has 24 significant bits, a round-to-nearest part takes 8 and leaves a
    // remainder of at most 16, then 8 bits).  The three planes of a (k group, column tile) are stored back to back:
    // dst16[(((g*NT + T)*3 + plane)*64 + lane)*8 + j].
    const int g = k >> 4, r = k & 15;
    const int lane = 32 * (r >> 3) + (n & 31);
    const __bf16 hi = (__bf16)w;
    const float r1 = w - (float)hi;
    const __bf16 mid = (__bf16)r1;
    const __bf16 lo = (__bf16)(r1 - (float)mid);
    unsigned short* d16 = reinterpret_cast<unsigned short*>(f.dst) + ((size_t)(g * NT + (n >> 5)) * 3 * 64 + lane) * 8 + (r & 7);
    d16[0] = __builtin_bit_cast(unsigned short, hi);
    d16[512] = __builtin_bit_cast(unsigned short, mid);
    d16[1024] = __builtin_bit_cast(unsigned short, lo);
  } else if (f.dtype == 4) {
    // f16x2 split fragments (NudfChainStep.prec 4): hi = fp16(w), lo = fp16((w - hi) 2^11); two planes back to back
    const int g = k >> 4, r = k & 15;
    const int lane = 32 * (r >> 3) + (n & 31);
    const _Float16 hi = (_Float16)w;
    const _Float16 lo = (_Float16)((w - (float)hi) * 2048.0f);
    unsigned short* d16 = reinterpret_cast<unsigned short*>(f.dst) + ((size_t)(g * NT + (n >> 5)) * 2 * 64 + lane) * 8 + (r & 7);
    d16[0] = __builtin_bit_cast(unsigned short, hi);
    d16[512] = __builtin_bit_cast(unsigned short, lo);
  } else {   // 16-bit fragments of v_mfma_f32_32x32x16_{f16,bf16}
    const int g = k >> 4, r = k & 15;
    const int lane = 32 * (r >> 3) + (n & 31);
    unsigned short bits;
    if (f.dtype == 1) {
      const _Float16 hv = (_Float16)w;
      bits = __builtin_bit_cast(unsigned short, hv);
    } else {
      const __bf16 bv = (__bf16)w;
      bits = __builtin_bit_cast(unsigned short, bv);
    }
    reinterpret_cast<unsigned short*>(f.dst)[((size_t)(g * NT + (n >> 5)) * 64 + lane) * 8 + (r & 7)] = bits;
  }
}

// One workgroup per (layer, panel of 32 output rows).  Phase 1: a wave per row forms the weight_norm scale (same sums,
// same order as wn_pack_kernel) and stages the scaled row in LDS (packed column order) while writing W.  Phase 2 writes
// W^T and the fragment copies FROM the panel: a bf16x3 fragment slot holds 8 consecutive k of one column = 16 bytes per
// plane, so a thread builds whole slots (three 16-byte stores, consecutive threads = consecutive slots) instead of 24
// two-byte stores scattered over as many cache lines.  Slots whose 8 rows straddle two panels (row offsets that are not
// a multiple of 8: the abs head's feature rows) and the fp32 / 16-bit fragment kinds keep the per-element stores.
// Measured (scripts/pack_bench.py under rocprofv3, the nine UDF layers alone): 18.4 us with every fragment kind of a train
// step, 9.7 us with none (W, W^T, norms; an empty launch is 4.5), 1.5-2 us per slot-built kind, 5.7 us for the kind that
// still goes element by element; inside the step 25.0 + 16 us (the wave-per-row form it replaces: 25.5 + 18.1) -- the
// launch is not where the step's time is.  All 197 chain outputs bit-identical to the previous form.
#define WNP_ROWS 32
#define WNP_THREADS 1024
__device__ __forceinline__ void wnp_split8(const float (&w)[8], uint4& hi, uint4& mid, uint4& lo) {
  unsigned short h[8], m[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {          // the same three roundings as frag_store's dtype 3
    const __bf16 a = (__bf16)w[j];
    const float r1 = w[j] - (float)a;
    const __bf16 b = (__bf16)r1;
    const __bf16 c = (__bf16)(r1 - (float)b);
    h[j] = __builtin_bit_cast(unsigned short, a);
    m[j] = __builtin_bit_cast(unsigned short, b);
    q[j] = __builtin_bit_cast(unsigned short, c);
  }
  hi = uint4{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16),
             (unsigned)h[4] | ((unsigned)h[5] << 16), (unsigned)h[6] | ((unsigned)h[7] << 16)};
  mid = uint4{(unsigned)m[0] | ((unsigned)m[1] << 16), (unsigned)m[2] | ((unsigned)m[3] << 16),
              (unsigned)m[4] | ((unsigned)m[5] << 16), (unsigned)m[6] | ((unsigned)m[7] << 16)};
  lo = uint4{(unsigned)q[0] | ((unsigned)q[1] << 16), (unsigned)q[2] | ((unsigned)q[3] << 16),
             (unsigned)q[4] | ((unsigned)q[5] << 16), (unsigned)q[6] | ((unsigned)q[7] << 16)};
}

__device__ __forceinline__ void wnp_split8_f16(const float (&w)[8], uint4& hi, uint4& lo) {
  unsigned short h[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {          // the same two roundings as frag_store's dtype 4
    const _Float16 a = (_Float16)w[j];
    const _Float16 b = (_Float16)((w[j] - (float)a) * 2048.0f);
    h[j] = __builtin_bit_cast(unsigned short, a);
    q[j] = __builtin_bit_cast(unsigned short, b);
  }
  hi = uint4{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16),
             (unsigned)h[4] | ((unsigned)h[5] << 16), (unsigned)h[6] | ((unsigned)h[7] << 16)};
  lo = uint4{(unsigned)q[0] | ((unsigned)q[1] << 16), (unsigned)q[2] | ((unsigned)q[3] << 16),
             (unsigned)q[4] | ((unsigned)q[5] << 16), (unsigned)q[6] | ((unsigned)q[7] << 16)};
}

__global__ __launch_bounds__(WNP_THREADS) void wn_pack_multi_kernel(NudfPackMulti a, int pw) {
  extern __shared__ float panel[];             // [WNP_ROWS][pw], pw = max in rounded up to 32, + 4
  int li = 0, b = blockIdx.x;
  for (; li < a.n_layers; ++li) {
    const int nb = (a.layer[li].out + WNP_ROWS - 1) / WNP_ROWS;
    if (b < nb) break;
    b -= nb;
  }
  if (li >= a.n_layers) return;
  const NudfPackLayer& L = a.layer[li];
  const int row0 = b * WNP_ROWS;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, tid = threadIdx.x;
  for (int e = tid; e < WNP_ROWS * pw; e += WNP_THREADS) panel[e] = 0.0f;      // rows >= out, columns >= in: zeros
  __syncthreads();
  for (int r = wave; r < WNP_ROWS; r += WNP_THREADS / 64) {   // two rows per wave: the row loads are latency, not bandwidth
    const int row = row0 + r;
    if (row >= L.out) break;
    float sc = 1.0f;
    if (L.g) {
      float ss = 0.f;
      for (int i = l; i < L.in; i += 64) {
        const float t = L.v[(size_t)row * L.in + i];
        ss += t * t;
      }
      ss = wave_sum(ss);
      const float inv = 1.0f / sqrtf(ss);
      if (l == 0 && L.inv_norm) L.inv_norm[row] = inv;
      sc = L.g[row] * inv;
    }
    for (int i = l; i < L.in; i += 64) {
      const int c = L.perm ? L.perm[i] : i;
      const float w = L.v[(size_t)row * L.in + i] * sc;
      if (L.W) L.W[(size_t)row * L.ldw + c] = w;
      panel[r * pw + c] = w;
    }
  }
  __syncthreads();
  const int nrows = min(WNP_ROWS, L.out - row0);
  if (L.Wt) {
    for (int e = tid; e < L.in * WNP_ROWS; e += WNP_THREADS) {      // lanes = consecutive rows of one column: 128-byte runs
      const int c = e >> 5, r = e & 31;
      if (r < nrows) L.Wt[(size_t)c * L.ldwt + row0 + r] = panel[r * pw + c];
    }
  }
  for (int fi = 0; fi < L.nfrag; ++fi) {
    const NudfPackFrag& f = L.frag[fi];
    const int NT = (f.N + 31) >> 5;
    uint4* d4 = reinterpret_cast<uint4*>(f.dst);
    const bool split16 = f.dtype == 3 || f.dtype == 4;      // slot-built split kinds: np planes of 64 uint4 per (g, T)
    const int np = (f.dtype == 3) ? 3 : 2;
    if (split16 && f.transpose) {
      // k = c - i0 (8 consecutive columns of a row), n = row - o0
      const int Q = (f.K + 7) >> 3;
      for (int e = tid; e < WNP_ROWS * Q; e += WNP_THREADS) {
        const int r = e & 31, q = e >> 5;
        const int n = row0 + r - f.o0;
        if (r >= nrows || n < 0 || n >= f.N) continue;
        float w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = 8 * q + j;
          w[j] = (k < f.K) ? panel[r * pw + f.i0 + k] : 0.0f;
        }
        uint4 hi, mid, lo;
        const size_t slot = ((size_t)((q >> 1) * NT + (n >> 5)) * np) * 64 + 32 * (q & 1) + (n & 31);
        if (np == 3) {
          wnp_split8(w, hi, mid, lo);
          d4[slot] = hi; d4[slot + 64] = mid; d4[slot + 128] = lo;
        } else {
          wnp_split8_f16(w, hi, lo);
          d4[slot] = hi; d4[slot + 64] = lo;
        }
      }
    } else if (split16 && (f.o0 & 7) == 0) {
      // k = row - o0 (8 consecutive rows of a column: row0 and o0 are multiples of 8), n = c - i0
      for (int e = tid; e < 4 * f.N; e += WNP_THREADS) {
        const int t = e / f.N, n = e - t * f.N;
        const int k0 = row0 - f.o0 + 8 * t;
        if (k0 < 0 || k0 >= f.K) continue;
        float w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = (k0 + j < f.K) ? panel[(8 * t + j) * pw + f.i0 + n] : 0.0f;
        uint4 hi, mid, lo;
        const size_t slot = ((size_t)((k0 >> 4) * NT + (n >> 5)) * np) * 64 + 32 * ((k0 >> 3) & 1) + (n & 31);
        if (np == 3) {
          wnp_split8(w, hi, mid, lo);
          d4[slot] = hi; d4[slot + 64] = mid; d4[slot + 128] = lo;
        } else {
          wnp_split8_f16(w, hi, lo);
          d4[slot] = hi; d4[slot + 64] = lo;
        }
      }
    } else {
      for (int e = tid; e < nrows * L.in; e += WNP_THREADS) {
        const int r = e / L.in, c = e - r * L.in;
        frag_store(f, row0 + r, c, panel[r * pw + c]);
      }
    }
  }
}
extern "C" int nudf_weightnorm_pack_multi(const NudfPackMulti* args, void* stream) {
  if (args->n_layers <= 0 || args->total_rows <= 0) return 0;
  if (args->n_layers > NUDF_PACK_MAX_LAYERS) {
    nudf_set_error("nudf_weightnorm_pack_multi: too many layers", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  int blocks = 0, in_max = 0;
  for (int i = 0; i < args->n_layers; ++i) {
    blocks += (args->layer[i].out + WNP_ROWS - 1) / WNP_ROWS;
    in_max = max(in_max, args->layer[i].in);
    for (int f = 0; f < args->layer[i].nfrag; ++f)      // (the panel is addressed up to i0 + K - 1 <= in - 1)
      if ((args->layer[i].frag[f].dtype == 3 || args->layer[i].frag[f].dtype == 4) && (((uintptr_t)args->layer[i].frag[f].dst) & 15)) {
        nudf_set_error("nudf_weightnorm_pack_multi: split (bf16x3 / f16x2) fragment buffers must be 16-byte aligned", hipErrorInvalidValue);
        return (int)hipErrorInvalidValue;
      }
  }
  const int pw = (in_max + 31) / 32 * 32 + 4;
  const size_t lds = (size_t)WNP_ROWS * pw * sizeof(float);
  if (lds > 64 * 1024) {
    nudf_set_error("nudf_weightnorm_pack_multi: layer wider than 508 inputs", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  hipLaunchKernelGGL(wn_pack_multi_kernel, dim3(blocks), dim3(WNP_THREADS), lds, (hipStream_t)stream, *args, pw);
  NUDF_CHECK_LAUNCH("nudf_weightnorm_pack_multi");
  return 0;
}

__global__ __launch_bounds__(256) void wn_unpack_multi_kernel(NudfUnpackMulti a) {
  const int grow = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (grow >= a.total_rows) return;
  int li = 0;
  while (li + 1 < a.n_layers && a.layer[li + 1].row_start <= grow) ++li;
  const NudfUnpackLayer& L = a.layer[li];
  const int row = grow - L.row_start;
  const int l = threadIdx.x & 63;
  if (l == 0 && L.db_out) L.db_out[row] = L.db_in[row];
  if (!L.g) {
    for (int i = l; i < L.in; i += 64)
      L.dv[(size_t)row * L.in + i] = L.dW[(size_t)row * L.ldw + (L.perm ? L.perm[i] : i)];
    return;
  }
  float dot = 0.f;
  for (int i = l; i < L.in; i += 64)
    dot += L.dW[(size_t)row * L.ldw + (L.perm ? L.perm[i] : i)] * L.v[(size_t)row * L.in + i];
  dot = wave_sum(dot);
  const float inv = L.inv_norm[row];
  const float gg = L.g[row];
  if (l == 0) L.dg[row] = dot * inv;
  const float c1 = gg * inv, c2 = gg * inv * inv * inv * dot;
  for (int i = l; i < L.in; i += 64)
    L.dv[(size_t)row * L.in + i] =
        c1 * L.dW[(size_t)row * L.ldw + (L.perm ? L.perm[i] : i)] - c2 * L.v[(size_t)row * L.in + i];
}
extern "C" int nudf_weightnorm_unpack_grad_multi(const NudfUnpackMulti* args, void* stream) {
  if (args->n_layers <= 0 || args->total_rows <= 0) return 0;
  if (args->n_layers > NUDF_PACK_MAX_LAYERS) {
    nudf_set_error("nudf_weightnorm_unpack_grad_multi: too many layers", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  hipLaunchKernelGGL(wn_unpack_multi_kernel, dim3((args->total_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, *args);
  NUDF_CHECK_LAUNCH("nudf_weightnorm_unpack_grad_multi");
  return 0;
}

// ---------------------------------------------------------------------------------------
// renderer scalars (one thread each) and the L1 colour-loss numerator
// ---------------------------------------------------------------------------------------
__global__ void scalars_fwd_kernel(const float* variance, const float* beta, const float* gamma, float beta_hi,
                                   float* scal, float* recip) {
  if (threadIdx.x != 0) return;
  const float s = fminf(fmaxf(expf(10.0f * variance[0]), 1e-6f), 1e6f);
  const float b = fminf(fmaxf(fminf(fmaxf(expf(10.0f * beta[0]), 0.0f), beta_hi), 1e-6f), 1e6f);
  const float g = fminf(fmaxf(expf(10.0f * gamma[0]), 1e-6f), 1e6f);
  scal[0] = s; scal[1] = b; scal[2] = g;
  if (recip) { recip[0] = 1.0f / s; recip[1] = 1.0f / b; }
}
extern "C" int nudf_scalars_fwd(const float* variance, const float* beta, const float* gamma, float beta_hi,
                                float* scal, float* recip, void* stream) {
  hipLaunchKernelGGL(scalars_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, variance, beta, gamma, beta_hi,
                     scal, recip);
  NUDF_CHECK_LAUNCH("nudf_scalars_fwd");
  return 0;
}
__global__ void scalars_bwd_kernel(const float* variance, const float* beta, const float* gamma, float beta_hi,
                                   const float* d_scal, float* d_param) {
  if (threadIdx.x != 0) return;
  const float s = expf(10.0f * variance[0]);
  d_param[0] = (s >= 1e-6f && s <= 1e6f) ? d_scal[0] * 10.0f * s : 0.0f;
  const float b = expf(10.0f * beta[0]);
  const float b1 = fminf(fmaxf(b, 0.0f), beta_hi);
  d_param[1] = (b >= 0.0f && b <= beta_hi && b1 >= 1e-6f && b1 <= 1e6f) ? d_scal[1] * 10.0f * b : 0.0f;
  const float g = expf(10.0f * gamma[0]);
  d_param[2] = (g >= 1e-6f && g <= 1e6f) ? d_scal[2] * 10.0f * g : 0.0f;
}
extern "C" int nudf_scalars_bwd(const float* variance, const float* beta, const float* gamma, float beta_hi,
                                const float* d_scal, float* d_param, void* stream) {
  hipLaunchKernelGGL(scalars_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, variance, beta, gamma, beta_hi,
                     d_scal, d_param);
  NUDF_CHECK_LAUNCH("nudf_scalars_bwd");
  return 0;
}

__global__ __launch_bounds__(1024) void l1_sum_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                          int n, float* out) {
  __shared__ float red[16];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) acc += fabsf(pred[i] - gt[i]);
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += red[w];
    out[0] = t;
  }
}
extern "C" int nudf_l1_sum_fwd(const float* pred, const float* gt, int n, float* out, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(l1_sum_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, pred, gt, n, out);
  NUDF_CHECK_LAUNCH("nudf_l1_sum_fwd");
  return 0;
}
__global__ void l1_sum_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt, int n,
                                  const float* d_out, float* __restrict__ d_pred) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float d = pred[i] - gt[i];
  d_pred[i] = d_out[0] * ((d > 0.0f) ? 1.0f : ((d < 0.0f) ? -1.0f : 0.0f));
}
extern "C" int nudf_l1_sum_bwd(const float* pred, const float* gt, int n, const float* d_out, float* d_pred,
                               void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(l1_sum_bwd_kernel, dim3(nblocks(n, 256)), dim3(256), 0, (hipStream_t)stream, pred, gt, n, d_out,
                     d_pred);
  NUDF_CHECK_LAUNCH("nudf_l1_sum_bwd");
  return 0;
}

__global__ void sums_errors_fwd_kernel(const float* sums, float n_rays, float* err) {
  if (threadIdx.x != 0) return;
  err[0] = sums[0] / (sums[1] + 1e-5f);
  err[1] = sums[2] / (sums[3] + 1e-5f);
  err[2] = sums[4] / n_rays;
}
extern "C" int nudf_sums_errors_fwd(const float* sums, float n_rays, float* err, void* stream) {
  hipLaunchKernelGGL(sums_errors_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, n_rays, err);
  NUDF_CHECK_LAUNCH("nudf_sums_errors_fwd");
  return 0;
}
__global__ void sums_errors_bwd_kernel(const float* sums, float n_rays, const float* d_err, float* d_sums) {
  if (threadIdx.x != 0) return;
  const float a = sums[1] + 1e-5f, b = sums[3] + 1e-5f;
  d_sums[0] = d_err[0] / a;
  d_sums[1] = -d_err[0] * sums[0] / (a * a);
  d_sums[2] = d_err[1] / b;
  d_sums[3] = -d_err[1] * sums[2] / (b * b);
  d_sums[4] = d_err[2] / n_rays;
}
extern "C" int nudf_sums_errors_bwd(const float* sums, float n_rays, const float* d_err, float* d_sums, void* stream) {
  hipLaunchKernelGGL(sums_errors_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, n_rays, d_err, d_sums);
  NUDF_CHECK_LAUNCH("nudf_sums_errors_bwd");
  return 0;
}

// Loss weights from device memory (include/nudf.h NUDF_LW_*): a graph-captured step replays the kernel ARGUMENTS of
// its capture, so the weights the runner's schedules move every iteration (adjust_color_loss_weights,
// regularization_weights_schedule: exp_runner_blending.py:199-211, 230-251) are read from a device vector when one is
// given and override the by-value floats.  Uniform scalar loads, once per kernel.
#define NUDF_LOSS_WEIGHTS3(w_dev, w_b, w_c, w_px)                                                  \
  if (w_dev) { w_b = (w_dev)[NUDF_LW_COLOR_BASE]; w_c = (w_dev)[NUDF_LW_COLOR]; w_px = (w_dev)[NUDF_LW_COLOR_PIXEL]; }
#define NUDF_LOSS_WEIGHTS6(w_dev, w_b, w_c, w_px, w_igr, w_igr_ns, w_sparse)                         \
  NUDF_LOSS_WEIGHTS3(w_dev, w_b, w_c, w_px)                                                         \
  if (w_dev) { w_igr = (w_dev)[NUDF_LW_IGR]; w_igr_ns = (w_dev)[NUDF_LW_IGR_NS]; w_sparse = (w_dev)[NUDF_LW_SPARSE]; }

// ColorLoss (two L1 terms) in one workgroup: the inputs are [N,3] per-ray tensors
__global__ __launch_bounds__(1024) void color_loss_fwd_kernel(const float* __restrict__ cb, const float* __restrict__ c,
                                                              const float* __restrict__ gt, int n,
                                                              const float* __restrict__ mask, int n_mask, float w_b,
                                                              float w_c, float w_px, const float* __restrict__ w_dev,
                                                              float* out, float* den_out) {
  __shared__ float red[3][16];
  NUDF_LOSS_WEIGHTS3(w_dev, w_b, w_c, w_px);
  float sb = 0.f, sc = 0.f, sm = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) {
    const float g = gt[i];
    sb += fabsf(cb[i] - g);
    sc += fabsf(c[i] - g);
  }
  if (mask)
    for (int i = threadIdx.x; i < n_mask; i += 1024) sm += mask[i];
  sb = wave_sum(sb); sc = wave_sum(sc); sm = wave_sum(sm);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = sb; red[1][threadIdx.x >> 6] = sc; red[2][threadIdx.x >> 6] = sm;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float tb = 0.f, tc = 0.f, tm = 0.f;
    for (int w = 0; w < 16; ++w) { tb += red[0][w]; tc += red[1][w]; tm += red[2][w]; }
    const float den = mask ? (tm + 1e-4f) : (float)n;
    const float Lb = tb / den, Lc = tc / den;
    out[0] = (Lb * w_b + Lc * w_c) / (w_b + w_c + w_px);
    out[1] = Lb;
    out[2] = Lc;
    den_out[0] = den;
  }
}
extern "C" int nudf_color_loss_fwd(const float* cb, const float* c, const float* gt, int n, const float* mask, int n_mask,
                                   float w_b, float w_c, float w_px, const float* w_dev, float* out, float* den_out,
                                   void* stream) {
  hipLaunchKernelGGL(color_loss_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, cb, c, gt, n, mask, n_mask, w_b,
                     w_c, w_px, w_dev, out, den_out);
  NUDF_CHECK_LAUNCH("nudf_color_loss_fwd");
  return 0;
}
// ray-sharded variant: local sums -> (caller all-reduces the 3 floats) -> finish.  sums = [sum|cb-gt|, sum|c-gt|, D]
// with D = sum(mask) or the element count; the same arithmetic as color_loss_fwd_kernel split at the exchange step.
__global__ __launch_bounds__(1024) void color_loss_sums_kernel(const float* __restrict__ cb, const float* __restrict__ c,
                                                               const float* __restrict__ gt, int n,
                                                               const float* __restrict__ mask, int n_mask, float* sums) {
  __shared__ float red[3][16];
  float sb = 0.f, sc = 0.f, sm = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) {
    const float g = gt[i];
    sb += fabsf(cb[i] - g);
    sc += fabsf(c[i] - g);
  }
  if (mask)
    for (int i = threadIdx.x; i < n_mask; i += 1024) sm += mask[i];
  sb = wave_sum(sb); sc = wave_sum(sc); sm = wave_sum(sm);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = sb; red[1][threadIdx.x >> 6] = sc; red[2][threadIdx.x >> 6] = sm;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float tb = 0.f, tc = 0.f, tm = 0.f;
    for (int w = 0; w < 16; ++w) { tb += red[0][w]; tc += red[1][w]; tm += red[2][w]; }
    sums[0] = tb; sums[1] = tc; sums[2] = mask ? tm : (float)n;
  }
}
extern "C" int nudf_color_loss_sums(const float* cb, const float* c, const float* gt, int n, const float* mask, int n_mask,
                                    float* sums, void* stream) {
  hipLaunchKernelGGL(color_loss_sums_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, cb, c, gt, n, mask, n_mask, sums);
  NUDF_CHECK_LAUNCH("nudf_color_loss_sums");
  return 0;
}
__global__ void color_loss_finish_kernel(const float* __restrict__ sums, int has_mask, float w_b, float w_c, float w_px,
                                         const float* __restrict__ w_dev, float* out, float* den_out) {
  NUDF_LOSS_WEIGHTS3(w_dev, w_b, w_c, w_px);
  const float den = has_mask ? (sums[2] + 1e-4f) : sums[2];
  const float Lb = sums[0] / den, Lc = sums[1] / den;
  out[0] = (Lb * w_b + Lc * w_c) / (w_b + w_c + w_px);
  out[1] = Lb;
  out[2] = Lc;
  den_out[0] = den;
}
extern "C" int nudf_color_loss_finish(const float* sums, int has_mask, float w_b, float w_c, float w_px,
                                      const float* w_dev, float* out, float* den_out, void* stream) {
  hipLaunchKernelGGL(color_loss_finish_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sums, has_mask, w_b, w_c, w_px,
                     w_dev, out, den_out);
  NUDF_CHECK_LAUNCH("nudf_color_loss_finish");
  return 0;
}
__global__ void color_loss_bwd_kernel(const float* __restrict__ cb, const float* __restrict__ c,
                                      const float* __restrict__ gt, int n, const float* den, float w_b, float w_c,
                                      float w_px, const float* __restrict__ w_dev, const float* d_out,
                                      float* __restrict__ d_cb, float* __restrict__ d_c) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  NUDF_LOSS_WEIGHTS3(w_dev, w_b, w_c, w_px);
  const float W = w_b + w_c + w_px;
  const float kb = (d_out[0] * w_b / W + d_out[1]) / den[0];
  const float kc = (d_out[0] * w_c / W + d_out[2]) / den[0];
  const float g = gt[i];
  const float a = cb[i] - g, b = c[i] - g;
  d_cb[i] = kb * ((a > 0.f) ? 1.f : ((a < 0.f) ? -1.f : 0.f));
  d_c[i] = kc * ((b > 0.f) ? 1.f : ((b < 0.f) ? -1.f : 0.f));
}
extern "C" int nudf_color_loss_bwd(const float* cb, const float* c, const float* gt, int n, const float* den, float w_b,
                                   float w_c, float w_px, const float* w_dev, const float* d_out, float* d_cb, float* d_c,
                                   void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(color_loss_bwd_kernel, dim3(nblocks(n, 256)), dim3(256), 0, (hipStream_t)stream, cb, c, gt, n, den,
                     w_b, w_c, w_px, w_dev, d_out, d_cb, d_c);
  NUDF_CHECK_LAUNCH("nudf_color_loss_bwd");
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// The loss assembly of a train step in ONE launch each way (exp_runner_blending.py:330-371 with only the two L1 colour
// terms and the three regularisers active -- every shipped conf outside the *_ft blending ones):
//     ColorLoss (loss/loss.py:105-133)            cl = (Lb w_b + Lc w_c) / (w_b + w_c + w_px)
//     regularisers from the composite sums (:531-536, 553)   ge, gens, sparse
//     total (:367-371)                             ((cl + gens * w_igr_ns) + sparse * w_sparse) + ge * w_igr
// Same arithmetic as nudf_color_loss_fwd + nudf_sums_errors_fwd followed by the runner's chain of scalar torch ops (each
// product and sum rounded on its own); it replaces ~20 one-element launches of that chain and its autograd backward,
// ~4.5 us each on the critical path of a 5.5 ms step.  out[8] = {total, cl, Lb, Lc, ge, gens, sparse, 0}.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void step_loss_fwd_kernel(const float* __restrict__ cb, const float* __restrict__ c,
                                                             const float* __restrict__ gt, int n,
                                                             const float* __restrict__ mask, int n_mask,
                                                             float* sums, float n_rays, float w_b,
                                                             float w_c, float w_px, float w_igr, float w_igr_ns,
                                                             float w_sparse, const float* __restrict__ w_dev, float* out,
                                                             float* den_out, const float* __restrict__ sums_ws,
                                                             int sums_nblk, int32_t* status) {
  __shared__ float red[3][16];
  __shared__ float red5[5][16];
  __shared__ float s5[5];
  NUDF_LOSS_WEIGHTS6(w_dev, w_b, w_c, w_px, w_igr, w_igr_ns, w_sparse);
  if (sums_ws) {
    // the composite kernel left its per-block partials (NudfComposite.defer_sums): partial_sums_kernel's reduction here,
    // same thread layout and order (composite.hip), the five sums written out for the backward
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b = threadIdx.x; b < sums_nblk; b += 1024) {
      const float* row = sums_ws + (size_t)b * 5;
#pragma unroll
      for (int k = 0; k < 5; ++k) acc[k] += row[k];
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const float t = wave_sum(acc[k]);
      if ((threadIdx.x & 63) == 0) red5[k][threadIdx.x >> 6] = t;
    }
    __syncthreads();
    if (threadIdx.x < 5) {
      float t = 0.f;
      for (int w = 0; w < 16; ++w) t += red5[threadIdx.x][w];
      sums[threadIdx.x] = t;
      s5[threadIdx.x] = t;
    }
    __syncthreads();
  }
  const float* sv = sums_ws ? s5 : sums;
  float sb = 0.f, sc = 0.f, sm = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) {
    const float g = gt[i];
    sb += fabsf(cb[i] - g);
    sc += fabsf(c[i] - g);
  }
  if (mask)
    for (int i = threadIdx.x; i < n_mask; i += 1024) sm += mask[i];
  sb = wave_sum(sb); sc = wave_sum(sc); sm = wave_sum(sm);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = sb; red[1][threadIdx.x >> 6] = sc; red[2][threadIdx.x >> 6] = sm;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float tb = 0.f, tc = 0.f, tm = 0.f;
    for (int w = 0; w < 16; ++w) { tb += red[0][w]; tc += red[1][w]; tm += red[2][w]; }
    const float den = mask ? (tm + 1e-4f) : (float)n;
    const float Lb = tb / den, Lc = tc / den;
    const float cl = (Lb * w_b + Lc * w_c) / (w_b + w_c + w_px);
    const float ge = sv[0] / (sv[1] + 1e-5f);
    const float gens = sv[2] / (sv[3] + 1e-5f);
    const float sp = sv[4] / n_rays;
    float total = __fadd_rn(cl, __fmul_rn(gens, w_igr_ns));
    total = __fadd_rn(total, __fmul_rn(sp, w_sparse));
    total = __fadd_rn(total, __fmul_rn(ge, w_igr));
    out[0] = total; out[1] = cl; out[2] = Lb; out[3] = Lc; out[4] = ge; out[5] = gens; out[6] = sp; out[7] = 0.f;
    den_out[0] = den;
    // (udf_renderer_blending.py:543-544 stops on a NaN eikonal term; every term of the step's loss ends in `total`)
    if (status && !(fabsf(total) <= 3.0e38f)) atomicOr(status, NUDF_STATUS_NONFINITE_LOSS);
  }
}
extern "C" int nudf_step_loss_fwd(const float* cb, const float* c, const float* gt, int n, const float* mask, int n_mask,
                                  float* sums, float n_rays, float w_b, float w_c, float w_px, float w_igr,
                                  float w_igr_ns, float w_sparse, const float* w_dev, float* out, float* den_out,
                                  const float* sums_ws, int sums_nblk, void* stream) {
  hipLaunchKernelGGL(step_loss_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, cb, c, gt, n, mask, n_mask, sums,
                     n_rays, w_b, w_c, w_px, w_igr, w_igr_ns, w_sparse, w_dev, out, den_out, sums_ws, sums_nblk, nudf_status_flag());
  NUDF_CHECK_LAUNCH("nudf_step_loss_fwd");
  return 0;
}
// upstream: d_total (device scalar; NULL = 1) and, optionally, d_extra[8] for the other outputs (index as out[]; NULL = 0)
// -> d cb, d c [n], d sums[5]
__global__ void step_loss_bwd_kernel(const float* __restrict__ cb, const float* __restrict__ c,
                                     const float* __restrict__ gt, int n, const float* __restrict__ den,
                                     const float* __restrict__ sums, float n_rays, float w_b, float w_c, float w_px,
                                     float w_igr, float w_igr_ns, float w_sparse, const float* __restrict__ w_dev,
                                     const float* __restrict__ d_total, const float* __restrict__ d_extra,
                                     float* __restrict__ d_cb, float* __restrict__ d_c, float* __restrict__ d_sums) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  NUDF_LOSS_WEIGHTS6(w_dev, w_b, w_c, w_px, w_igr, w_igr_ns, w_sparse);
  const float g = d_total ? d_total[0] : 1.0f;
  const float g_cl = g + (d_extra ? d_extra[1] : 0.0f);
  if (i == 0) {
    const float d_ge = __fmul_rn(g, w_igr) + (d_extra ? d_extra[4] : 0.0f);
    const float d_gens = __fmul_rn(g, w_igr_ns) + (d_extra ? d_extra[5] : 0.0f);
    const float d_sp = __fmul_rn(g, w_sparse) + (d_extra ? d_extra[6] : 0.0f);
    const float a = sums[1] + 1e-5f, b = sums[3] + 1e-5f;
    d_sums[0] = d_ge / a;
    d_sums[1] = -d_ge * sums[0] / (a * a);
    d_sums[2] = d_gens / b;
    d_sums[3] = -d_gens * sums[2] / (b * b);
    d_sums[4] = d_sp / n_rays;
  }
  if (i >= n) return;
  const float W = w_b + w_c + w_px;
  const float kb = (g_cl * w_b / W + (d_extra ? d_extra[2] : 0.0f)) / den[0];
  const float kc = (g_cl * w_c / W + (d_extra ? d_extra[3] : 0.0f)) / den[0];
  const float t = gt[i];
  const float a = cb[i] - t, b = c[i] - t;
  d_cb[i] = kb * ((a > 0.f) ? 1.f : ((a < 0.f) ? -1.f : 0.f));
  d_c[i] = kc * ((b > 0.f) ? 1.f : ((b < 0.f) ? -1.f : 0.f));
}
extern "C" int nudf_step_loss_bwd(const float* cb, const float* c, const float* gt, int n, const float* den,
                                  const float* sums, float n_rays, float w_b, float w_c, float w_px, float w_igr,
                                  float w_igr_ns, float w_sparse, const float* w_dev, const float* d_total,
                                  const float* d_extra, float* d_cb, float* d_c, float* d_sums, void* stream) {
  hipLaunchKernelGGL(step_loss_bwd_kernel, dim3(nblocks(max(n, 1), 256)), dim3(256), 0, (hipStream_t)stream, cb, c, gt, n,
                     den, sums, n_rays, w_b, w_c, w_px, w_igr, w_igr_ns, w_sparse, w_dev, d_total, d_extra, d_cb, d_c,
                     d_sums);
  NUDF_CHECK_LAUNCH("nudf_step_loss_bwd");
  return 0;
}

// out4 [P_pad, 4]: column 0 = sign * d * scale (d = NULL: sign * scale), the rest (and rows >= P) zero; out4_sign (optional,
// same shape): column 0 = sign * scale in the same launch (the second-order weight gradient's operand) -- the 4-wide
// column-0 operand of the UDF head's adjoint / second-order weight gradient (mlp.UDFEngine.backward), in one launch
// instead of a zero fill, two products and a strided copy.  (sign * d) * scale, each product rounded, as torch does.
__global__ void col0_seed4_kernel(const float* __restrict__ sign, const float* __restrict__ d, float scale, int P, int P_pad,
                                  float* __restrict__ out4, float* __restrict__ out4_sign) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= P_pad) return;
  float v = 0.0f, vs = 0.0f;
  if (r < P) {
    vs = __fmul_rn(sign[r], scale);
    v = d ? __fmul_rn(__fmul_rn(sign[r], d[r]), scale) : vs;
  }
  reinterpret_cast<float4*>(out4)[r] = make_float4(v, 0.0f, 0.0f, 0.0f);
  if (out4_sign) reinterpret_cast<float4*>(out4_sign)[r] = make_float4(vs, 0.0f, 0.0f, 0.0f);
}
extern "C" int nudf_col0_seed4(const float* sign, const float* d, float scale, int P, int P_pad, float* out4,
                               float* out4_sign, void* stream) {
  if (P_pad <= 0) return 0;
  hipLaunchKernelGGL(col0_seed4_kernel, dim3(nblocks(P_pad, 256)), dim3(256), 0, (hipStream_t)stream, sign, d, scale, P,
                     P_pad, out4, out4_sign);
  NUDF_CHECK_LAUNCH("nudf_col0_seed4");
  return 0;
}


// ---------------------------------------------------------------------------------------------------------------
// The BLENDING step's loss in three launches around torch's sort (BASELINE config 3: ColorLoss with the pixel and the trimmed
// patch term, loss/loss.py:105-133, :66-84, the patch-mask algebra of exp_runner_blending.py:313-315, the three regularisers
// and the runner's weighted total, :330-371).  The generic path is ~90 one-element / per-ray torch launches of ~4.7 us each
// between the two patch-blend kernels (profiles/r04_step_sequence_garment_blend.txt).
//   prepare : m[i] = ((patch_mask[i] * (weight_sum[i] > 0.5)) > 0), err_masked[i] = err[i] * m[i]
//   (torch.sort(err_masked, descending=True) -> err_sorted, order: the trimmed mean needs order statistics)
//   fwd     : out[12] = {total, cl, Lb, Lc, Lpix, Lpatch, ge, gens, sparse, den_pix, k, keep_sum}
//   bwd     : d cb, d c, d pix [3 N], d err [N], d sums [5]
// Every product / sum of the weighted totals is rounded on its own, as the chain of 0-d torch ops rounds them.
// ---------------------------------------------------------------------------------------------------------------
__global__ void blend_loss_prepare_kernel(NudfBlendLoss p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.N) return;
  const float mk = (p.patch_mask[i] * ((p.weight_sum[i] > 0.5f) ? 1.0f : 0.0f)) > 0.0f ? 1.0f : 0.0f;
  p.m[i] = mk;
  p.err_masked[i] = p.err[i] * mk;
}
extern "C" int nudf_blend_loss_prepare(const NudfBlendLoss* args, void* stream) {
  if (args->N <= 0) return 0;
  hipLaunchKernelGGL(blend_loss_prepare_kernel, dim3(nblocks(args->N, 256)), dim3(256), 0, (hipStream_t)stream, *args);
  NUDF_CHECK_LAUNCH("nudf_blend_loss_prepare");
  return 0;
}

__global__ __launch_bounds__(1024) void blend_loss_fwd_kernel(NudfBlendLoss p, int32_t* status) {
  __shared__ float red[8][16];
  __shared__ float s5[5];
  __shared__ float tot[8];
  const int tid = threadIdx.x;
  if (p.sums_ws) {      // the composite launch left its per-block partials: partial_sums_kernel's reduction, same order
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b = tid; b < p.sums_nblk; b += 1024) {
      const float* row = p.sums_ws + (size_t)b * 5;
#pragma unroll
      for (int k = 0; k < 5; ++k) acc[k] += row[k];
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const float t = wave_sum(acc[k]);
      if ((tid & 63) == 0) red[k][tid >> 6] = t;
    }
    __syncthreads();
    if (tid < 5) {
      float t = 0.f;
      for (int w = 0; w < 16; ++w) t += red[tid][w];
      p.sums[tid] = t;
      s5[tid] = t;
    }
    __syncthreads();
  }
  const float* sv = p.sums_ws ? s5 : p.sums;
  // pass 1: the three L1 sums and the mask count
  float sb = 0.f, sc = 0.f, sp = 0.f, cnt = 0.f;
  const int n3 = 3 * p.N;
  for (int i = tid; i < n3; i += 1024) {
    const float g = p.gt[i];
    sb += fabsf(p.cb[i] - g);
    sc += fabsf(p.c[i] - g);
    sp += fabsf(p.pix[i] - g);
  }
  for (int i = tid; i < p.N; i += 1024) cnt += p.m[i];
  sb = wave_sum(sb); sc = wave_sum(sc); sp = wave_sum(sp); cnt = wave_sum(cnt);
  if ((tid & 63) == 0) { red[0][tid >> 6] = sb; red[1][tid >> 6] = sc; red[2][tid >> 6] = sp; red[3][tid >> 6] = cnt; }
  __syncthreads();
  if (tid < 4) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += red[tid][w];
    tot[tid] = t;
  }
  __syncthreads();
  // pass 2: trimmed mean over the descending-sorted masked errors: the first k = floor(ratio * count) positions lose their
  // mask (loss/loss.py:79-84), mean of the errors still masked
  const float kf = floorf(p.trim_ratio * tot[3]);
  float es = 0.f, ks = 0.f;
  for (int j = tid; j < p.N; j += 1024) {
    const float keep = (p.m[p.order[j]] > 0.f && (float)j >= kf) ? 1.0f : 0.0f;
    es += p.err_sorted[j] * keep;
    ks += keep;
  }
  es = wave_sum(es); ks = wave_sum(ks);
  __syncthreads();
  if ((tid & 63) == 0) { red[4][tid >> 6] = es; red[5][tid >> 6] = ks; }
  __syncthreads();
  if (tid == 0) {
    float tes = 0.f, tks = 0.f;
    for (int w = 0; w < 16; ++w) { tes += red[4][w]; tks += red[5][w]; }
    const float* w = p.w_dev;
    const float w_b = w[NUDF_LW_COLOR_BASE], w_c = w[NUDF_LW_COLOR], w_px = w[NUDF_LW_COLOR_PIXEL], w_pa = w[NUDF_LW_COLOR_PATCH];
    const float w_sum = w[NUDF_LW_COLOR_SUM];
    const float Lb = tot[0] / (float)n3, Lc = tot[1] / (float)n3;
    const float den_pix = tot[3] + 1e-4f;
    const float Lpix = tot[2] / den_pix;
    const float Lpatch = tes / tks;
    float cl = __fadd_rn(__fadd_rn(__fmul_rn(Lb, w_b), __fmul_rn(Lc, w_c)), __fmul_rn(Lpix, w_px));
    cl = __fadd_rn(__fdiv_rn(cl, w_sum), __fmul_rn(Lpatch, w_pa));
    const float ge = sv[0] / (sv[1] + 1e-5f);
    const float gens = sv[2] / (sv[3] + 1e-5f);
    const float spr = sv[4] / p.n_rays;
    float total = __fadd_rn(cl, __fmul_rn(gens, w[NUDF_LW_IGR_NS]));
    total = __fadd_rn(total, __fmul_rn(spr, w[NUDF_LW_SPARSE]));
    total = __fadd_rn(total, __fmul_rn(ge, w[NUDF_LW_IGR]));
    float* o = p.out;
    o[0] = total; o[1] = cl; o[2] = Lb; o[3] = Lc; o[4] = Lpix; o[5] = Lpatch; o[6] = ge; o[7] = gens; o[8] = spr;
    o[9] = den_pix; o[10] = kf; o[11] = tks;
    if (status && !(fabsf(total) <= 3.0e38f)) atomicOr(status, NUDF_STATUS_NONFINITE_LOSS);
  }
}
extern "C" int nudf_blend_loss_fwd(const NudfBlendLoss* args, void* stream) {
  if (args->N <= 0 || !args->w_dev) {
    nudf_set_error("nudf_blend_loss_fwd: N >= 1 and the device weight vector required", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  hipLaunchKernelGGL(blend_loss_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, *args, nudf_status_flag());
  NUDF_CHECK_LAUNCH("nudf_blend_loss_fwd");
  return 0;
}

__global__ void blend_loss_bwd_kernel(NudfBlendLoss p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const float* w = p.w_dev;
  const float g = p.d_total ? p.d_total[0] : 1.0f;
  const float w_sum = w[NUDF_LW_COLOR_SUM];
  if (i == 0) {
    const float d_ge = __fmul_rn(g, w[NUDF_LW_IGR]), d_gens = __fmul_rn(g, w[NUDF_LW_IGR_NS]), d_sp = __fmul_rn(g, w[NUDF_LW_SPARSE]);
    const float a = p.sums[1] + 1e-5f, b = p.sums[3] + 1e-5f;
    p.d_sums[0] = d_ge / a;
    p.d_sums[1] = -d_ge * p.sums[0] / (a * a);
    p.d_sums[2] = d_gens / b;
    p.d_sums[3] = -d_gens * p.sums[2] / (b * b);
    p.d_sums[4] = d_sp / p.n_rays;
  }
  const int n3 = 3 * p.N;
  if (i < n3) {
    const float kb = g * w[NUDF_LW_COLOR_BASE] / w_sum / (float)n3;
    const float kc = g * w[NUDF_LW_COLOR] / w_sum / (float)n3;
    const float kp = g * w[NUDF_LW_COLOR_PIXEL] / w_sum / p.out[9];
    const float t = p.gt[i];
    const float a = p.cb[i] - t, b = p.c[i] - t, c = p.pix[i] - t;
    p.d_cb[i] = kb * ((a > 0.f) ? 1.f : ((a < 0.f) ? -1.f : 0.f));
    p.d_c[i] = kc * ((b > 0.f) ? 1.f : ((b < 0.f) ? -1.f : 0.f));
    p.d_pix[i] = kp * ((c > 0.f) ? 1.f : ((c < 0.f) ? -1.f : 0.f));
  }
  if (i < p.N) {      // position i of the sorted list -> ray order[i]: d err = d Lpatch * keep / keep_sum (* m)
    const long long r = p.order[i];
    const float keep = (p.m[r] > 0.f && (float)i >= p.out[10]) ? 1.0f : 0.0f;
    p.d_err[r] = g * w[NUDF_LW_COLOR_PATCH] * keep / p.out[11];
  }
}
extern "C" int nudf_blend_loss_bwd(const NudfBlendLoss* args, void* stream) {
  if (args->N <= 0) return 0;
  hipLaunchKernelGGL(blend_loss_bwd_kernel, dim3(nblocks(3 * args->N, 256)), dim3(256), 0, (hipStream_t)stream, *args);
  NUDF_CHECK_LAUNCH("nudf_blend_loss_bwd");
  return 0;
}
