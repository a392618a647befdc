// Pixel / patch blending (image-based rendering branch) and the SSIM patch loss.
//
//   nudf_pixel_blend_{fwd,bwd}  PatchProjector.pixel_warp -> sample_ptsFeatures_from_featureMaps -> cam2pixel
//                               + color_blend (pixel part)        models/patch_projector.py:21-43,
//                               models/projector_utils.py:8-85, models/fields.py:498-519
//   nudf_pixel_composite_{fwd,bwd}  inside/background mix + weighted sum   models/udf_renderer_blending.py:503-518
//   nudf_patch_blend_{fwd,bwd}  PatchProjector.patch_warp + patch_homography + color_blend (patch part) +
//                               the weighted sum over samples          models/patch_projector.py:45-164,
//                               models/fields.py:521-535, models/udf_renderer_blending.py:520-524
//   nudf_ssim_patch             SSIM with a full-patch Gaussian window   loss/patch_metric.py:21-41, 76-84
//
// The reference materialises [V, N*S*Npx, 3] warped colours (616 MB at 1024x128x49); here the homography,
// the taps, the view softmax and the sum over samples stay in registers and only [N,Npx,3] is written.
// Gather-bound on the source images (L2 / Infinity-Cache resident).  Gradients flow to the blending logits
// and to the compositing weights only (normals are detached and the homographies are built under no_grad in
// the reference; sample positions carry no gradient to any parameter).
#include "nudf_common.h"
#include "../../include/nudf.h"

#define MAXV 10   // blending_cand_views of the shipped confs; 8 source views are used (dataset.py:129-149)
// fully unrolled loop over the views so that per-view arrays stay in registers
#define FORV(v) _Pragma("unroll") for (int v = 0; v < MAXV; ++v) if (v < V)

struct __attribute__((packed, aligned(4))) Texel3 { float r, g, b; };

// F.grid_sample(bilinear, padding_mode='zeros', align_corners=True) on pixel coordinates.  HWC = channel-interleaved
// image (one 12-byte load per texel); otherwise three planes.  Loads are clamped and unconditional, the validity of
// each tap goes into its weight (no divergent load groups); same products and summation order in both layouts.
template <bool HWC>
__device__ __forceinline__ void bilinear3(const float* __restrict__ img, int H, int W, float ix, float iy, float out[3]) {
  const float x0f = floorf(ix), y0f = floorf(iy);
  const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
  const float wx1 = ix - x0f, wx0 = (x0f + 1.0f) - ix;
  const float wy1 = iy - y0f, wy0 = (y0f + 1.0f) - iy;
  const bool vx0 = (x0 >= 0) && (x0 <= W - 1), vx1 = (x1 >= 0) && (x1 <= W - 1);
  const bool vy0 = (y0 >= 0) && (y0 <= H - 1), vy1 = (y1 >= 0) && (y1 <= H - 1);
  const int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x1, 0), W - 1);
  const int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y1, 0), H - 1);
  const bool v00 = vx0 && vy0, v10 = vx1 && vy0, v01 = vx0 && vy1, v11 = vx1 && vy1;
  const float w00 = wx0 * wy0, w10 = wx1 * wy0, w01 = wx0 * wy1, w11 = wx1 * wy1;
  const unsigned o00 = (unsigned)cy0 * W + cx0, o10 = (unsigned)cy0 * W + cx1;
  const unsigned o01 = (unsigned)cy1 * W + cx0, o11 = (unsigned)cy1 * W + cx1;
  float t[4][3];
  if (HWC) {
    const Texel3* q = reinterpret_cast<const Texel3*>(img);
    const Texel3 a = q[o00], b = q[o10], c = q[o01], d = q[o11];
    t[0][0] = a.r; t[0][1] = a.g; t[0][2] = a.b;
    t[1][0] = b.r; t[1][1] = b.g; t[1][2] = b.b;
    t[2][0] = c.r; t[2][1] = c.g; t[2][2] = c.b;
    t[3][0] = d.r; t[3][1] = d.g; t[3][2] = d.b;
  } else {
    const size_t plane = (size_t)H * W;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float* pl = img + ch * plane;
      t[0][ch] = pl[o00]; t[1][ch] = pl[o10]; t[2][ch] = pl[o01]; t[3][ch] = pl[o11];
    }
  }
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    float v = 0.0f;
    if (v00) v += t[0][ch] * w00;
    if (v10) v += t[1][ch] * w10;
    if (v01) v += t[2][ch] * w01;
    if (v11) v += t[3][ch] * w11;
    out[ch] = v;
  }
}

// masked softmax blend weights over V views (fields.py:510-515 / 525-530)
__device__ __forceinline__ void blend_weights(const float* lg, const bool* m, int V, float* sm, float* wn, float& Te) {
  float mx = -1e30f;
  FORV(v) mx = fmaxf(mx, lg[v]);
  float den = 0.f;
  FORV(v) {
    sm[v] = expf(lg[v] - mx);
    den += sm[v];
  }
  float T = 0.f;
  FORV(v) {
    sm[v] /= den;
    wn[v] = m[v] ? sm[v] : 0.0f;
    T += wn[v];
  }
  Te = T + 1e-8f;
  FORV(v) wn[v] /= Te;
}
// backward of blend_weights: dwn -> dlogits
__device__ __forceinline__ void blend_weights_bwd(const float* sm, const float* wn, const bool* m, int V, float Te,
                                                  const float* dwn, float* dl) {
  float dot = 0.f;
  FORV(v) dot += dwn[v] * wn[v];
  float dsm[MAXV];
  float sdot = 0.f;
  FORV(v) {
    const float dw = (dwn[v] - dot) / Te;
    dsm[v] = m[v] ? dw : 0.0f;
    sdot += dsm[v] * sm[v];
  }
  FORV(v) dl[v] = sm[v] * (dsm[v] - sdot);
}

// ------------------------------------------------------------------------------------------
// pixel blending: one thread per sample
// ------------------------------------------------------------------------------------------
template <bool BWD>
__global__ void pixel_blend_kernel(NudfPixelBlend p, const float* __restrict__ d_pix, float* __restrict__ d_logits) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.P) return;
  const float x = p.pts[(size_t)i * 3 + 0], y = p.pts[(size_t)i * 3 + 1], z = p.pts[(size_t)i * 3 + 2];
  const int V = p.V, H = p.H, W = p.W;
  float col[MAXV][3], lg[MAXV], sm[MAXV], wn[MAXV];
  bool m[MAXV];
  FORV(v) {
    const float* pr = p.proj + v * 12;
    const float X = pr[0] * x + pr[1] * y + pr[2] * z + pr[3];
    const float Y = pr[4] * x + pr[5] * y + pr[6] * z + pr[7];
    const float Z = fmaxf(pr[8] * x + pr[9] * y + pr[10] * z + pr[11], 1e-3f);
    float xn = 2.0f * (X / Z) / (float)(W - 1) - 1.0f;
    float yn = 2.0f * (Y / Z) / (float)(H - 1) - 1.0f;
    if (xn > 1.0f || xn < -1.0f) xn = 2.0f;   // projector_utils.py:39-43
    if (yn > 1.0f || yn < -1.0f) yn = 2.0f;
    m[v] = (fabsf(xn) < 1.0f) && (fabsf(yn) < 1.0f);
    const float ix = (xn + 1.0f) * 0.5f * (float)(W - 1);
    const float iy = (yn + 1.0f) * 0.5f * (float)(H - 1);
    if (p.img_layout) bilinear3<true>(p.imgs + (size_t)v * 3 * H * W, H, W, ix, iy, col[v]);
    else bilinear3<false>(p.imgs + (size_t)v * 3 * H * W, H, W, ix, iy, col[v]);
    lg[v] = p.logits[(size_t)i * p.nl + v];
  }
  float Te;
  blend_weights(lg, m, V, sm, wn, Te);
  if (!BWD) {
    float o[3] = {0.f, 0.f, 0.f};
    FORV(v)
      for (int c = 0; c < 3; ++c) o[c] += col[v][c] * wn[v];
    for (int c = 0; c < 3; ++c) p.pix[(size_t)i * 3 + c] = o[c];
  } else {
    float dwn[MAXV], dl[MAXV];
    FORV(v)
      dwn[v] = d_pix[(size_t)i * 3] * col[v][0] + d_pix[(size_t)i * 3 + 1] * col[v][1] + d_pix[(size_t)i * 3 + 2] * col[v][2];
    blend_weights_bwd(sm, wn, m, V, Te, dwn, dl);
#pragma unroll
    for (int v = 0; v < MAXV; ++v)
      if (v < p.nl) d_logits[(size_t)i * p.nl + v] = (v < V) ? dl[v] : 0.0f;
  }
}

extern "C" int nudf_pixel_blend_fwd(const NudfPixelBlend* a, void* stream) {
  if (a->P <= 0) return 0;
  if (a->V > MAXV || a->V > a->nl) {
    nudf_set_error("nudf_pixel_blend: too many views", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  hipLaunchKernelGGL(pixel_blend_kernel<false>, dim3((a->P + 127) / 128), dim3(128), 0, (hipStream_t)stream, *a, nullptr,
                     nullptr);
  NUDF_CHECK_LAUNCH("nudf_pixel_blend_fwd");
  return 0;
}
extern "C" int nudf_pixel_blend_bwd(const NudfPixelBlend* a, const float* d_pix, float* d_logits, void* stream) {
  if (a->P <= 0) return 0;
  hipLaunchKernelGGL(pixel_blend_kernel<true>, dim3((a->P + 127) / 128), dim3(128), 0, (hipStream_t)stream, *a, d_pix,
                     d_logits);
  NUDF_CHECK_LAUNCH("nudf_pixel_blend_bwd");
  return 0;
}

// ------------------------------------------------------------------------------------------
// un-fused warps: PatchProjector.pixel_warp / .patch_warp return the PER-VIEW samples (the reference's projector API,
// models/patch_projector.py:21-43, 45-164).  The training path never materialises them (see the fused kernels);
// these two forward-only kernels exist for callers of the projector interface itself.
// ------------------------------------------------------------------------------------------
__global__ void pixel_warp_kernel(NudfPixelBlend p, float* __restrict__ colors, float* __restrict__ mask) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.P) return;
  const float x = p.pts[(size_t)i * 3 + 0], y = p.pts[(size_t)i * 3 + 1], z = p.pts[(size_t)i * 3 + 2];
  const int V = p.V, H = p.H, W = p.W;
  FORV(v) {
    const float* pr = p.proj + v * 12;
    const float X = pr[0] * x + pr[1] * y + pr[2] * z + pr[3];
    const float Y = pr[4] * x + pr[5] * y + pr[6] * z + pr[7];
    const float Z = fmaxf(pr[8] * x + pr[9] * y + pr[10] * z + pr[11], 1e-3f);
    float xn = 2.0f * (X / Z) / (float)(W - 1) - 1.0f;
    float yn = 2.0f * (Y / Z) / (float)(H - 1) - 1.0f;
    if (xn > 1.0f || xn < -1.0f) xn = 2.0f;   // projector_utils.py:39-43
    if (yn > 1.0f || yn < -1.0f) yn = 2.0f;
    const bool m = (fabsf(xn) < 1.0f) && (fabsf(yn) < 1.0f);
    const float ix = (xn + 1.0f) * 0.5f * (float)(W - 1);
    const float iy = (yn + 1.0f) * 0.5f * (float)(H - 1);
    float col[3];
    if (p.img_layout) bilinear3<true>(p.imgs + (size_t)v * 3 * H * W, H, W, ix, iy, col);
    else bilinear3<false>(p.imgs + (size_t)v * 3 * H * W, H, W, ix, iy, col);
    const size_t o = (size_t)i * V + v;
    colors[o * 3] = col[0]; colors[o * 3 + 1] = col[1]; colors[o * 3 + 2] = col[2];
    mask[o] = m ? 1.0f : 0.0f;
  }
}
extern "C" int nudf_pixel_warp(const NudfPixelBlend* a, float* colors, float* mask, void* stream) {
  if (a->P <= 0) return 0;
  if (a->V > MAXV) {
    nudf_set_error("nudf_pixel_warp: too many views", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  hipLaunchKernelGGL(pixel_warp_kernel, dim3((a->P + 127) / 128), dim3(128), 0, (hipStream_t)stream, *a, colors, mask);
  NUDF_CHECK_LAUNCH("nudf_pixel_warp");
  return 0;
}

// ------------------------------------------------------------------------------------------
// composite of the blended pixel colours (with the inside/background mix of :503-506): one wave per ray
//   x_s = pix_s*inside_s + bg_in_s*(1-inside_s)  (s < S, only when a background exists) ; x_s = bg_tail (s >= S)
//   out = sum_s w_s x_s
// ------------------------------------------------------------------------------------------
template <bool BWD>
__global__ __launch_bounds__(256) void pixel_composite_kernel(NudfPixelComposite p, const float* __restrict__ d_out,
                                                              float* d_w, float* d_pix, float* d_bg_in,
                                                              float* d_bg_tail) {
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + wave;
  if (ray >= p.N) return;
  const int S = p.S, NO = p.n_out, ST = S + NO;
  float acc[3] = {0.f, 0.f, 0.f};
  float dor = 0.f, dog = 0.f, dob = 0.f;
  if (BWD) {
    dor = d_out[ray * 3];
    dog = d_out[ray * 3 + 1];
    dob = d_out[ray * 3 + 2];
  }
  for (int i = l; i < ST; i += 64) {
    const float w = p.w[(size_t)ray * ST + i];
    float xr, xg, xb, ins = 1.0f;
    if (i < S) {
      const size_t b = ((size_t)ray * S + i) * 3;
      xr = p.pix[b]; xg = p.pix[b + 1]; xb = p.pix[b + 2];
      if (p.bg_in) {
        const float px = p.pts[b], py = p.pts[b + 1], pz = p.pts[b + 2];
        ins = (sqrtf(px * px + py * py + pz * pz) < 1.0f) ? 1.0f : 0.0f;
        xr = xr * ins + p.bg_in[b] * (1.0f - ins);
        xg = xg * ins + p.bg_in[b + 1] * (1.0f - ins);
        xb = xb * ins + p.bg_in[b + 2] * (1.0f - ins);
      }
      if (BWD) {
        d_pix[b] = w * dor * ins; d_pix[b + 1] = w * dog * ins; d_pix[b + 2] = w * dob * ins;
        if (d_bg_in) {
          d_bg_in[b] = w * dor * (1.0f - ins); d_bg_in[b + 1] = w * dog * (1.0f - ins);
          d_bg_in[b + 2] = w * dob * (1.0f - ins);
        }
      }
    } else {
      const size_t b = ((size_t)ray * NO + (i - S)) * 3;
      xr = p.bg_tail[b]; xg = p.bg_tail[b + 1]; xb = p.bg_tail[b + 2];
      if (BWD && d_bg_tail) {
        d_bg_tail[b] = w * dor; d_bg_tail[b + 1] = w * dog; d_bg_tail[b + 2] = w * dob;
      }
    }
    if (BWD) d_w[(size_t)ray * ST + i] = dor * xr + dog * xg + dob * xb;
    else { acc[0] += w * xr; acc[1] += w * xg; acc[2] += w * xb; }
  }
  if (!BWD) {
    for (int c = 0; c < 3; ++c) acc[c] = wave_sum(acc[c]);
    if (l == 0) { p.out[ray * 3] = acc[0]; p.out[ray * 3 + 1] = acc[1]; p.out[ray * 3 + 2] = acc[2]; }
  }
}
extern "C" int nudf_pixel_composite_fwd(const NudfPixelComposite* a, void* stream) {
  if (a->N <= 0) return 0;
  hipLaunchKernelGGL(pixel_composite_kernel<false>, dim3((a->N + 3) / 4), dim3(256), 0, (hipStream_t)stream, *a, nullptr,
                     nullptr, nullptr, nullptr, nullptr);
  NUDF_CHECK_LAUNCH("nudf_pixel_composite_fwd");
  return 0;
}
extern "C" int nudf_pixel_composite_bwd(const NudfPixelComposite* a, const float* d_out, float* d_w, float* d_pix,
                                        float* d_bg_in, float* d_bg_tail, void* stream) {
  if (a->N <= 0) return 0;
  hipLaunchKernelGGL(pixel_composite_kernel<true>, dim3((a->N + 3) / 4), dim3(256), 0, (hipStream_t)stream, *a, d_out, d_w,
                     d_pix, d_bg_in, d_bg_tail);
  NUDF_CHECK_LAUNCH("nudf_pixel_composite_bwd");
  return 0;
}

// ------------------------------------------------------------------------------------------
// patch blending: one wave per ray, lanes over the (2h+1)^2 patch pixels (PC chunks of 64)
// cam layout (floats): ref = [K_ref_inv 9 | R_ref 9 | t_ref 3 | cam_loc 3] ; per view
//                      [K_src 9 | R_rel 9 | t_rel 3 | c2 3]   with c2 = -R_rel^T t_rel
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void mat3_mul(const float* A, const float* B, float* C) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}

// One WORKGROUP per ray; its PB_WAVES waves take the samples round-robin (the per-sample work -- V homographies,
// V x Npx bilinear gathers, the view softmax -- is independent across samples; only the forward's sum over samples
// is shared and is reduced through LDS in a fixed order at the end).  One wave per ray left a single wave per SIMD
// walking S x V dependent gather rounds: latency bound at ~0.7 TB/s of texel traffic.
// PB_WAVES = 4 for >= 1024 rays (4 waves/SIMD resident across the chip in one round), 8 below.
template <bool BWD, int PC, int PB_WAVES>
__global__ __launch_bounds__(64 * PB_WAVES) void patch_blend_kernel(NudfPatchBlend p, const float* __restrict__ d_patch,
                                                                    float* __restrict__ d_logits,
                                                                    float* __restrict__ d_w) {
  __shared__ float red[BWD ? 1 : PB_WAVES][BWD ? 1 : (PC * 64 * 3 + 1)];
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int ray = blockIdx.x;
  const int S = p.S, V = p.V, H = p.H, W = p.W, h = p.hps, ws = 2 * p.hps + 1, Npx = ws * ws;
  const float* ref = p.ref_cam;
  const float dxr = p.rays_d[ray * 3], dyr = p.rays_d[ray * 3 + 1], dzr = p.rays_d[ray * 3 + 2];
  const float u0 = p.uv[ray * 2], v0 = p.uv[ray * 2 + 1];

  float hx[PC], hy[PC];
  bool act[PC];
  float accp[PC][3], dpc[PC][3];
#pragma unroll
  for (int c = 0; c < PC; ++c) {
    const int pi = c * 64 + l;
    act[c] = pi < Npx;
    hx[c] = u0 + (float)((pi % ws) - h);      // offsets: (dx, dy), row by row (patch_projector.py:211-214)
    hy[c] = v0 + (float)((pi / ws) - h);
    accp[c][0] = accp[c][1] = accp[c][2] = 0.f;
    if (BWD && act[c]) {
      const size_t b = ((size_t)ray * Npx + pi) * 3;
      dpc[c][0] = d_patch[b]; dpc[c][1] = d_patch[b + 1]; dpc[c][2] = d_patch[b + 2];
    } else {
      dpc[c][0] = dpc[c][1] = dpc[c][2] = 0.f;
    }
  }
  float pmask_acc = 0.f;

  for (int s = wave; s < S; s += PB_WAVES) {
    const size_t sb = (size_t)ray * S + s;
    const float px = p.pts[sb * 3], py = p.pts[sb * 3 + 1], pz = p.pts[sb * 3 + 2];
    float gx = p.grad[sb * 3], gy = p.grad[sb * 3 + 1], gz = p.grad[sb * 3 + 2];
    const float gme = sqrtf(gx * gx + gy * gy + gz * gz) + 1e-5f;
    gx /= gme; gy /= gme; gz /= gme;
    const float cs = dxr * gx + dyr * gy + dzr * gz;
    const float flip = (cs > 0.0f) ? -1.0f : 1.0f;                 // normals = flip_sign * gradients_norm (:447)
    const float nx = flip * gx, ny = flip * gy, nz = flip * gz;
    const float* Rr = ref + 9;
    const float rn[3] = {Rr[0] * nx + Rr[1] * ny + Rr[2] * nz, Rr[3] * nx + Rr[4] * ny + Rr[5] * nz,
                         Rr[6] * nx + Rr[7] * ny + Rr[8] * nz};
    const float pr[3] = {Rr[0] * px + Rr[1] * py + Rr[2] * pz + ref[18], Rr[3] * px + Rr[4] * py + Rr[5] * pz + ref[19],
                         Rr[6] * px + Rr[7] * py + Rr[8] * pz + ref[20]};
    const float d1 = rn[0] * pr[0] + rn[1] * pr[1] + rn[2] * pr[2];
    const float sgn = (d1 < 0.0f) ? -1.0f : 1.0f;                    // sign, 0 -> 1 (:112-114)
    const float dd = fmaxf(fabsf(d1), 1e-8f) * sgn;
    const float ex = px - ref[21], ey = py - ref[22], ez = pz - ref[23];
    const float sdist = sqrtf(ex * ex + ey * ey + ez * ez);

    float col[PC][MAXV][3];
    bool vvalid[MAXV];
    float lg[MAXV], sm[MAXV], wn[MAXV];
    FORV(v) {
      const float* cam = p.src_cam + v * 24;
      const float* Ks = cam;
      const float* Rl = cam + 9;
      const float* tl = cam + 18;
      const float* c2 = cam + 21;
      const float d2 = rn[0] * c2[0] + rn[1] * c2[1] + rn[2] * c2[2];
      const bool valid_h = (fabsf(d1) > 1e-3f) && (fabsf(d1 - d2) > 1e-3f) && ((d2 / d1) < 1.0f);
      // plane normal / distance of the homography; invalid -> fronto-parallel at the sample's distance (:117-131)
      const float q0 = valid_h ? rn[0] / dd : 0.0f, q1 = valid_h ? rn[1] / dd : 0.0f,
                  q2 = valid_h ? rn[2] / dd : 1.0f / sdist;
      float M1[9], M2[9], Hm[9];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        M1[i * 3 + 0] = Rl[i * 3 + 0] + tl[i] * q0;
        M1[i * 3 + 1] = Rl[i * 3 + 1] + tl[i] * q1;
        M1[i * 3 + 2] = Rl[i * 3 + 2] + tl[i] * q2;
      }
      mat3_mul(Ks, M1, M2);
      mat3_mul(M2, ref, Hm);
      bool allv = true;
#pragma unroll
      for (int c = 0; c < PC; ++c) {
        float t0 = Hm[0] * hx[c] + Hm[1] * hy[c] + Hm[2];
        float t1 = Hm[3] * hx[c] + Hm[4] * hy[c] + Hm[5];
        float t2 = Hm[6] * hx[c] + Hm[7] * hy[c] + Hm[8];
        const float den = fmaxf(t2, 1e-8f);
        const float gxp = t0 / den, gyp = t1 / den;
        bool mk = (t2 > 0.0f) && (gxp < (float)(W - h)) && (gyp < (float)(H - h)) && (gxp >= (float)h) && (gyp >= (float)h);
        float xn = fminf(fmaxf(2.0f * gxp / (float)(W - 1) - 1.0f, -10.0f), 10.0f);
        float yn = fminf(fmaxf(2.0f * gyp / (float)(H - 1) - 1.0f, -10.0f), 10.0f);
        const float ix = (xn + 1.0f) * 0.5f * (float)(W - 1), iy = (yn + 1.0f) * 0.5f * (float)(H - 1);
        if (act[c]) {
          if (p.img_layout) bilinear3<true>(p.imgs + (size_t)v * 3 * H * W, H, W, ix, iy, col[c][v]);
          else bilinear3<false>(p.imgs + (size_t)v * 3 * H * W, H, W, ix, iy, col[c][v]);
        }
        else { col[c][v][0] = col[c][v][1] = col[c][v][2] = 0.f; mk = true; }
        allv = allv && mk;
      }
      // view is usable for this sample only if every patch pixel is valid (fields.py:525)
      vvalid[v] = (__ballot(allv) == ~0ull);
      lg[v] = p.logits[sb * p.nl + v];
    }
    float Te;
    blend_weights(lg, vvalid, V, sm, wn, Te);
    bool anyv = false;
    FORV(v) anyv = anyv || vvalid[v];
    const float w = p.w[(size_t)ray * p.ldw + s];
    if (!BWD) {
#pragma unroll
      for (int c = 0; c < PC; ++c)
        FORV(v) {
          accp[c][0] += w * wn[v] * col[c][v][0];
          accp[c][1] += w * wn[v] * col[c][v][1];
          accp[c][2] += w * wn[v] * col[c][v][2];
        }
      if (anyv) pmask_acc += w;
    } else {
      // d w_s = sum_{px,c} dpc * blended ; d wn_v = w * sum_{px,c} dpc * col_v
      float dwn[MAXV], dl[MAXV];
      float dws = 0.f;
      FORV(v) {
        float t = 0.f;
#pragma unroll
        for (int c = 0; c < PC; ++c) t += dpc[c][0] * col[c][v][0] + dpc[c][1] * col[c][v][1] + dpc[c][2] * col[c][v][2];
        t = wave_sum(t);
        dwn[v] = w * t;
        dws += wn[v] * t;
      }
      blend_weights_bwd(sm, wn, vvalid, V, Te, dwn, dl);
      if (l == 0) {
        d_w[(size_t)ray * S + s] = dws;
#pragma unroll
        for (int v = 0; v < MAXV; ++v)
          if (v < p.nl) d_logits[sb * p.nl + v] = (v < V) ? dl[v] : 0.0f;
      }
    }
  }
  if (!BWD) {
#pragma unroll
    for (int c = 0; c < PC; ++c) {
      red[wave][(c * 64 + l) * 3 + 0] = accp[c][0];
      red[wave][(c * 64 + l) * 3 + 1] = accp[c][1];
      red[wave][(c * 64 + l) * 3 + 2] = accp[c][2];
    }
    if (l == 0) red[wave][PC * 64 * 3] = pmask_acc;
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int c = 0; c < PC; ++c) {
        const int pi = c * 64 + l;
        float t0 = 0.f, t1 = 0.f, t2 = 0.f;
        for (int w8 = 0; w8 < PB_WAVES; ++w8) {
          t0 += red[w8][pi * 3]; t1 += red[w8][pi * 3 + 1]; t2 += red[w8][pi * 3 + 2];
        }
        if (pi < Npx) {
          const size_t b = ((size_t)ray * Npx + pi) * 3;
          p.patch_colors[b] = t0; p.patch_colors[b + 1] = t1; p.patch_colors[b + 2] = t2;
        }
      }
      if (l == 0) {
        float t = 0.f;
        for (int w8 = 0; w8 < PB_WAVES; ++w8) t += red[w8][PC * 64 * 3];
        p.patch_mask[ray] = t;
      }
    }
  }
}

extern "C" int nudf_patch_blend_fwd(const NudfPatchBlend* a, void* stream) {
  if (a->N <= 0) return 0;
  const int npx = (2 * a->hps + 1) * (2 * a->hps + 1);
  if (a->V > MAXV || a->V > a->nl || npx > 128) {
    nudf_set_error("nudf_patch_blend: V <= 16 and h_patch_size <= 5 required", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  const bool w8 = a->N < 1024;
  dim3 grid(a->N), block(w8 ? 512 : 256);
  hipStream_t st = (hipStream_t)stream;
  if (npx <= 64) {
    if (w8) hipLaunchKernelGGL((patch_blend_kernel<false, 1, 8>), grid, block, 0, st, *a, nullptr, nullptr, nullptr);
    else hipLaunchKernelGGL((patch_blend_kernel<false, 1, 4>), grid, block, 0, st, *a, nullptr, nullptr, nullptr);
  } else {
    if (w8) hipLaunchKernelGGL((patch_blend_kernel<false, 2, 8>), grid, block, 0, st, *a, nullptr, nullptr, nullptr);
    else hipLaunchKernelGGL((patch_blend_kernel<false, 2, 4>), grid, block, 0, st, *a, nullptr, nullptr, nullptr);
  }
  NUDF_CHECK_LAUNCH("nudf_patch_blend_fwd");
  return 0;
}
extern "C" int nudf_patch_blend_bwd(const NudfPatchBlend* a, const float* d_patch, float* d_logits, float* d_w,
                                    void* stream) {
  if (a->N <= 0) return 0;
  const int npx = (2 * a->hps + 1) * (2 * a->hps + 1);
  const bool w8 = a->N < 1024;
  dim3 grid(a->N), block(w8 ? 512 : 256);
  hipStream_t st = (hipStream_t)stream;
  if (npx <= 64) {
    if (w8) hipLaunchKernelGGL((patch_blend_kernel<true, 1, 8>), grid, block, 0, st, *a, d_patch, d_logits, d_w);
    else hipLaunchKernelGGL((patch_blend_kernel<true, 1, 4>), grid, block, 0, st, *a, d_patch, d_logits, d_w);
  } else {
    if (w8) hipLaunchKernelGGL((patch_blend_kernel<true, 2, 8>), grid, block, 0, st, *a, d_patch, d_logits, d_w);
    else hipLaunchKernelGGL((patch_blend_kernel<true, 2, 4>), grid, block, 0, st, *a, d_patch, d_logits, d_w);
  }
  NUDF_CHECK_LAUNCH("nudf_patch_blend_bwd");
  return 0;
}

// one wave per (ray, sample); lanes over the (2h+1)^2 patch pixels (PC chunks of 64).  Same plane-induced
// homographies as patch_blend_kernel, but the plane normal is the caller's `normals` (patch_projector.py:99-131).
template <int PC>
__global__ __launch_bounds__(256) void patch_warp_kernel(NudfPatchWarp p) {
  const int l = threadIdx.x & 63;
  const long long sidx = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (sidx >= (long long)p.N * p.S) return;
  const int ray = (int)(sidx / p.S);
  const int V = p.V, H = p.H, W = p.W, h = p.hps, ws = 2 * p.hps + 1, Npx = ws * ws;
  const float* ref = p.ref_cam;
  const float u0 = p.uv[ray * 2], v0 = p.uv[ray * 2 + 1];
  const float px = p.pts[sidx * 3], py = p.pts[sidx * 3 + 1], pz = p.pts[sidx * 3 + 2];
  const float nx = p.normals[sidx * 3], ny = p.normals[sidx * 3 + 1], nz = p.normals[sidx * 3 + 2];
  const float* Rr = ref + 9;
  const float rn[3] = {Rr[0] * nx + Rr[1] * ny + Rr[2] * nz, Rr[3] * nx + Rr[4] * ny + Rr[5] * nz,
                       Rr[6] * nx + Rr[7] * ny + Rr[8] * nz};
  const float pr[3] = {Rr[0] * px + Rr[1] * py + Rr[2] * pz + ref[18], Rr[3] * px + Rr[4] * py + Rr[5] * pz + ref[19],
                       Rr[6] * px + Rr[7] * py + Rr[8] * pz + ref[20]};
  const float d1 = rn[0] * pr[0] + rn[1] * pr[1] + rn[2] * pr[2];
  const float sgn = (d1 < 0.0f) ? -1.0f : 1.0f;
  const float dd = fmaxf(fabsf(d1), 1e-8f) * sgn;
  const float ex = px - ref[21], ey = py - ref[22], ez = pz - ref[23];
  const float sdist = sqrtf(ex * ex + ey * ey + ez * ez);
  FORV(v) {
    const float* cam = p.src_cam + v * 24;
    const float* Ks = cam;
    const float* Rl = cam + 9;
    const float* tl = cam + 18;
    const float* c2 = cam + 21;
    const float d2 = rn[0] * c2[0] + rn[1] * c2[1] + rn[2] * c2[2];
    const bool valid_h = (fabsf(d1) > 1e-3f) && (fabsf(d1 - d2) > 1e-3f) && ((d2 / d1) < 1.0f);
    const float q0 = valid_h ? rn[0] / dd : 0.0f, q1 = valid_h ? rn[1] / dd : 0.0f, q2 = valid_h ? rn[2] / dd : 1.0f / sdist;
    float M1[9], M2[9], Hm[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      M1[i * 3 + 0] = Rl[i * 3 + 0] + tl[i] * q0;
      M1[i * 3 + 1] = Rl[i * 3 + 1] + tl[i] * q1;
      M1[i * 3 + 2] = Rl[i * 3 + 2] + tl[i] * q2;
    }
    mat3_mul(Ks, M1, M2);
    mat3_mul(M2, ref, Hm);
#pragma unroll
    for (int c = 0; c < PC; ++c) {
      const int pi = c * 64 + l;
      if (pi >= Npx) continue;
      const float hx = u0 + (float)((pi % ws) - h), hy = v0 + (float)((pi / ws) - h);
      const float t0 = Hm[0] * hx + Hm[1] * hy + Hm[2];
      const float t1 = Hm[3] * hx + Hm[4] * hy + Hm[5];
      const float t2 = Hm[6] * hx + Hm[7] * hy + Hm[8];
      const float den = fmaxf(t2, 1e-8f);
      const float gxp = t0 / den, gyp = t1 / den;
      const bool mk = (t2 > 0.0f) && (gxp < (float)(W - h)) && (gyp < (float)(H - h)) && (gxp >= (float)h) && (gyp >= (float)h);
      const float xn = fminf(fmaxf(2.0f * gxp / (float)(W - 1) - 1.0f, -10.0f), 10.0f);
      const float yn = fminf(fmaxf(2.0f * gyp / (float)(H - 1) - 1.0f, -10.0f), 10.0f);
      const float ix = (xn + 1.0f) * 0.5f * (float)(W - 1), iy = (yn + 1.0f) * 0.5f * (float)(H - 1);
      float col[3];
      if (p.img_layout) bilinear3<true>(p.imgs + (size_t)v * 3 * H * W, H, W, ix, iy, col);
      else bilinear3<false>(p.imgs + (size_t)v * 3 * H * W, H, W, ix, iy, col);
      const size_t o = ((size_t)sidx * V + v) * Npx + pi;
      p.colors[o * 3] = col[0]; p.colors[o * 3 + 1] = col[1]; p.colors[o * 3 + 2] = col[2];
      p.mask[o] = mk ? 1.0f : 0.0f;
    }
  }
}
extern "C" int nudf_patch_warp(const NudfPatchWarp* a, void* stream) {
  if (a->N <= 0 || a->S <= 0) return 0;
  const int npx = (2 * a->hps + 1) * (2 * a->hps + 1);
  if (a->V > MAXV || npx > 128) {
    nudf_set_error("nudf_patch_warp: V <= 10 and h_patch_size <= 5 required", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  const long long n = (long long)a->N * a->S;
  dim3 grid((unsigned)((n + 3) / 4)), block(256);
  if (npx <= 64) hipLaunchKernelGGL(patch_warp_kernel<1>, grid, block, 0, (hipStream_t)stream, *a);
  else hipLaunchKernelGGL(patch_warp_kernel<2>, grid, block, 0, (hipStream_t)stream, *a);
  NUDF_CHECK_LAUNCH("nudf_patch_warp");
  return 0;
}

// ------------------------------------------------------------------------------------------
// SSIM patch error: one wave per ray, lanes over patch pixels; out[n] = sum_c (1 - ssim_c) / 2.
// With d_out/d_pred non-NULL also writes d out / d pred * d_out.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ssim_patch_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                         const float* __restrict__ win, int N, int Npx,
                                                         float* __restrict__ out, const float* __restrict__ d_out,
                                                         float* __restrict__ d_pred) {
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + wave;
  if (ray >= N) return;
  const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
  float total = 0.f;
  const float dout = d_out ? d_out[ray] : 0.f;
  for (int c = 0; c < 3; ++c) {
    float m1 = 0, m2 = 0, e11 = 0, e22 = 0, e12 = 0;
    for (int i = l; i < Npx; i += 64) {
      const float w = win[i];
      const float a = pred[((size_t)ray * Npx + i) * 3 + c], b = gt[((size_t)ray * Npx + i) * 3 + c];
      m1 += w * a; m2 += w * b; e11 += w * a * a; e22 += w * b * b; e12 += w * a * b;
    }
    m1 = wave_sum(m1); m2 = wave_sum(m2); e11 = wave_sum(e11); e22 = wave_sum(e22); e12 = wave_sum(e12);
    const float s1 = e11 - m1 * m1, s2 = e22 - m2 * m2, s12 = e12 - m1 * m2;
    const float A = 2.f * m1 * m2 + C1, B = 2.f * s12 + C2, Cc = m1 * m1 + m2 * m2 + C1, Dd = s1 + s2 + C2;
    total += 1.0f - (A * B) / (Cc * Dd);
    if (d_pred) {
      const float inv = 1.0f / (Cc * Dd);
      for (int i = l; i < Npx; i += 64) {
        const float w = win[i];
        const float a = pred[((size_t)ray * Npx + i) * 3 + c], b = gt[((size_t)ray * Npx + i) * 3 + c];
        const float dA = 2.f * m2 * w, dB = 2.f * w * (b - m2), dC = 2.f * m1 * w, dD = 2.f * w * (a - m1);
        const float dval = -((dA * B + A * dB) * inv - (A * B) * (dC * Dd + Cc * dD) * inv * inv);
        d_pred[((size_t)ray * Npx + i) * 3 + c] = 0.5f * dval * dout;
      }
    }
  }
  if (l == 0) out[ray] = total * 0.5f;
}
extern "C" int nudf_ssim_patch(const float* pred, const float* gt, const float* window, int N, int Npx, float* out,
                               const float* d_out, float* d_pred, void* stream);

// The other patch errors of ColorPatchLoss (loss/loss.py:66-73): TYPE 1 'l1' = sum_px mean_c |pred - gt|, 2 'ssd' =
// sum_px mean_c (pred - gt)^2, 3 'ncc' = 1 - mean_c NCC_c with the full-patch Gaussian window (loss/patch_metric.py:44-67:
// NCC_c = sum_px w (a - mu1)(b - mu2) / ((sqrt(var1 + 1e-4) + 1e-8)(sqrt(var2 + 1e-4) + 1e-8)), sum w = 1).
// One wave per ray, lanes over patch pixels; with d_out / d_pred non-NULL also d out / d pred * d_out.
template <int TYPE>
__global__ __launch_bounds__(256) void patch_metric_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                           const float* __restrict__ win, int N, int Npx,
                                                           float* __restrict__ out, const float* __restrict__ d_out,
                                                           float* __restrict__ d_pred) {
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + wave;
  if (ray >= N) return;
  const float dout = d_out ? d_out[ray] : 0.f;
  const size_t base = (size_t)ray * Npx;
  if (TYPE == 1 || TYPE == 2) {
    float acc = 0.f;
    for (int i = l; i < Npx; i += 64)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float d = pred[(base + i) * 3 + c] - gt[(base + i) * 3 + c];
        acc += (TYPE == 1) ? fabsf(d) : d * d;
        if (d_pred)
          d_pred[(base + i) * 3 + c] = ((TYPE == 1) ? ((d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f)) : 2.f * d) * (dout * (1.0f / 3.0f));
      }
    acc = wave_sum(acc);
    if (l == 0) out[ray] = acc * (1.0f / 3.0f);
    return;
  }
  float total = 0.f;
  for (int c = 0; c < 3; ++c) {
    float m1 = 0, m2 = 0, e11 = 0, e22 = 0, e12 = 0;
    for (int i = l; i < Npx; i += 64) {
      const float w = win[i];
      const float a = pred[(base + i) * 3 + c], b = gt[(base + i) * 3 + c];
      m1 += w * a; m2 += w * b; e11 += w * a * a; e22 += w * b * b; e12 += w * a * b;
    }
    m1 = wave_sum(m1); m2 = wave_sum(m2); e11 = wave_sum(e11); e22 = wave_sum(e22); e12 = wave_sum(e12);
    const float s1 = sqrtf(e11 - m1 * m1 + 1e-4f), s2 = sqrtf(e22 - m2 * m2 + 1e-4f);
    const float d1 = s1 + 1e-8f, d2 = s2 + 1e-8f;
    const float cov = e12 - m1 * m2;
    total += cov / (d1 * d2);
    if (d_pred) {
      // d cov / d a_i = w_i (b_i - mu2); d s1 / d a_i = w_i (a_i - mu1) / s1
      const float k1 = 1.0f / (d1 * d2), k2 = cov / (d1 * d1 * d2 * s1);
      for (int i = l; i < Npx; i += 64) {
        const float w = win[i];
        const float a = pred[(base + i) * 3 + c], b = gt[(base + i) * 3 + c];
        d_pred[(base + i) * 3 + c] = -(w * (b - m2) * k1 - w * (a - m1) * k2) * (dout * (1.0f / 3.0f));
      }
    }
  }
  if (l == 0) out[ray] = 1.0f - total * (1.0f / 3.0f);
}

extern "C" int nudf_patch_metric(int type, const float* pred, const float* gt, const float* window, int N, int Npx,
                                 float* out, const float* d_out, float* d_pred, void* stream) {
  if (N <= 0) return 0;
  if (type == 0) return nudf_ssim_patch(pred, gt, window, N, Npx, out, d_out, d_pred, stream);
  const dim3 grid((N + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (type == 1) hipLaunchKernelGGL(patch_metric_kernel<1>, grid, block, 0, st, pred, gt, window, N, Npx, out, d_out, d_pred);
  else if (type == 2) hipLaunchKernelGGL(patch_metric_kernel<2>, grid, block, 0, st, pred, gt, window, N, Npx, out, d_out, d_pred);
  else if (type == 3) hipLaunchKernelGGL(patch_metric_kernel<3>, grid, block, 0, st, pred, gt, window, N, Npx, out, d_out, d_pred);
  else {
    nudf_set_error("nudf_patch_metric: type must be 0 (ssim), 1 (l1), 2 (ssd) or 3 (ncc)", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  NUDF_CHECK_LAUNCH("nudf_patch_metric");
  return 0;
}

extern "C" int nudf_ssim_patch(const float* pred, const float* gt, const float* window, int N, int Npx, float* out,
                               const float* d_out, float* d_pred, void* stream) {
  if (N <= 0) return 0;
  hipLaunchKernelGGL(ssim_patch_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, pred, gt, window, N, Npx, out,
                     d_out, d_pred);
  NUDF_CHECK_LAUNCH("nudf_ssim_patch");
  return 0;
}
