// GPU-resident ray / patch batch generation (the step immediately before the hot path):
// one launch replaces the ~30 small torch ops + one grid_sample of Dataset.gen_random_rays_patches_at
// (dataset/dataset.py:228-294) and Dataset.near_far_from_sphere (:329-335).  One thread per ray for the ray
// record, one thread per (ray, patch pixel) for the ground-truth patch crop.  HBM / latency bound (tiny).
#include "nudf_common.h"
#include "../../include/nudf.h"

// fp contraction is switched off in both kernels: the sampling positions go through a normalise / un-normalise round
// trip at |x| ~ 1e3 where one fused rounding moves the bilinear weights by 1e-4; with separate roundings the results
// are the reference's fp32 sequence bit for bit (up to the order of the 3-term dot products).
__global__ __launch_bounds__(256) void ray_batch_kernel(NudfRayBatch p) {
#pragma clang fp contract(off)
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= p.N) return;
  const long long px = p.pixels_x[r], py = p.pixels_y[r];
  const float x = (float)px, y = (float)py;
  // p = K^-1 (x, y, 1)   (dataset.py:277-278)
  const float* Ki = p.intrinsics_inv;  // row-major 4x4, upper-left 3x3 used
  const float c0 = Ki[0] * x + Ki[1] * y + Ki[2];
  const float c1 = Ki[4] * x + Ki[5] * y + Ki[6];
  const float c2 = Ki[8] * x + Ki[9] * y + Ki[10];
  const float nrm = sqrtf(c0 * c0 + c1 * c1 + c2 * c2);
  const float v0 = c0 / nrm, v1 = c1 / nrm, v2 = c2 / nrm;   // :279
  const float* T = p.pose;             // row-major 4x4 camera-to-world
  const float d0 = T[0] * v0 + T[1] * v1 + T[2] * v2;        // :280
  const float d1 = T[4] * v0 + T[5] * v1 + T[6] * v2;
  const float d2 = T[8] * v0 + T[9] * v1 + T[10] * v2;
  const float o0 = T[3], o1 = T[7], o2 = T[11];              // :281
  const size_t pix = ((size_t)py * p.W + (size_t)px) * 3;
  float* ray = p.rays + (size_t)r * 10;                       // :283  [o 3 | v 3 | colour 3 | mask 1]
  ray[0] = o0; ray[1] = o1; ray[2] = o2;
  ray[3] = d0; ray[4] = d1; ray[5] = d2;
  ray[6] = p.image[pix + 0]; ray[7] = p.image[pix + 1]; ray[8] = p.image[pix + 2];   // :275
  ray[9] = (p.mask[pix] > 0.0f) ? 1.0f : 0.0f;                                        // :276, mask[:, :1]
  if (p.ndc_uv) {                                             // :270-272  2 * x / (W - 1) - 1
    p.ndc_uv[r * 2 + 0] = (float)(2 * px) / (float)(p.W - 1) - 1.0f;
    p.ndc_uv[r * 2 + 1] = (float)(2 * py) / (float)(p.H - 1) - 1.0f;
  }
  if (p.xyz_cam) {
    p.xyz_cam[r * 3 + 0] = c0; p.xyz_cam[r * 3 + 1] = c1; p.xyz_cam[r * 3 + 2] = c2;
  }
  if (p.near) {                                               // near_far_from_sphere, :329-335
    const float a = d0 * d0 + d1 * d1 + d2 * d2;
    const float b = 2.0f * (o0 * d0 + o1 * d1 + o2 * d2);
    const float mid = 0.5f * (-b) / a;
    p.near[r] = mid - 1.0f;
    p.far[r] = mid + 1.0f;
  }
  if (p.patch_mask) {                                         // :257-260 (strict inequalities)
    const int h = p.h_patch_size;
    p.patch_mask[r] = (px > h && px < (p.W - h) && py > h && py < (p.H - h)) ? 1 : 0;
  }
}

// ground-truth patch colours: F.grid_sample(image, uv, bilinear, zeros, align_corners=False) at the patch pixels
// (dataset.py:254-266); the normalise / un-normalise round trip of the reference is kept so that the sampling
// positions are the reference's (x W / (W - 1) - 0.5, not x).
__global__ __launch_bounds__(256) void patch_crop_kernel(NudfRayBatch p) {
#pragma clang fp contract(off)
  const int npx = (2 * p.h_patch_size + 1) * (2 * p.h_patch_size + 1);
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)p.N * npx) return;
  const int r = (int)(idx / npx), q = (int)(idx - (long long)r * npx);
  const int side = 2 * p.h_patch_size + 1;
  const int dx = q % side - p.h_patch_size, dy = q / side - p.h_patch_size;   // build_patch_offset: x fastest
  const float gx = (float)p.pixels_x[r] + (float)dx, gy = (float)p.pixels_y[r] + (float)dy;
  const float u = 2.0f * gx / (float)(p.W - 1) - 1.0f, v = 2.0f * gy / (float)(p.H - 1) - 1.0f;
  const float ix = ((u + 1.0f) * (float)p.W - 1.0f) * 0.5f, iy = ((v + 1.0f) * (float)p.H - 1.0f) * 0.5f;
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy;
  const float tx = ix - fx, ty = iy - fy;
  float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int xx = x0 + i, yy = y0 + j;
      if (xx < 0 || yy < 0 || xx >= p.W || yy >= p.H) continue;
      const float w = (i ? tx : 1.0f - tx) * (j ? ty : 1.0f - ty);
      const float* s = p.image + ((size_t)yy * p.W + xx) * 3;
      acc[0] += w * s[0]; acc[1] += w * s[1]; acc[2] += w * s[2];
    }
  float* o = p.patch_color + (size_t)idx * 3;
  o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2];
}

extern "C" int nudf_gen_ray_batch(const NudfRayBatch* args, void* stream) {
  const NudfRayBatch& p = *args;
  if (p.N <= 0) return 0;
  if (p.W < 2 || p.H < 2 || p.h_patch_size < 0) {
    nudf_set_error("nudf_gen_ray_batch: W, H >= 2 and h_patch_size >= 0 required", hipErrorInvalidValue);
    return (int)hipErrorInvalidValue;
  }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ray_batch_kernel, dim3((p.N + 255) / 256), dim3(256), 0, st, p);
  if (p.patch_color) {
    const long long total = (long long)p.N * (2 * p.h_patch_size + 1) * (2 * p.h_patch_size + 1);
    hipLaunchKernelGGL(patch_crop_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p);
  }
  NUDF_CHECK_LAUNCH("nudf_gen_ray_batch");
  return 0;
}
