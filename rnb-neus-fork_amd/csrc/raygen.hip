// Per-step ray / target generation of train_rnb on the device.
//
// The reference does this on the host every step: two CPU randint draws, CPU fancy-indexing of the
// [n_images, n_lights, H, W, 3] image / light tensors, three H2D copies and a D2H of the pixel indices
// (models/dataset.py:351-376, exp_runner.py:174-180, :214-220).  At ~10^5 rays/s that round trip is longer than
// the render step.  Here the image stack stays in HBM and one launch produces everything a step consumes.
#include "rnb_internal.h"

namespace rnb {

struct RayGenArgs {
  const float* kinv;       // [4,4] inverse intrinsics of the view (row-major)
  const float* pose;       // [4,4] camera-to-world pose of the view
  const float* images;     // [L,H,W,3] or NULL
  const float* images_wu;  // [L,H,W,3] or NULL
  const float* mask;       // [H,W,Cm]
  const float* lights;     // [L,H,W,3] or NULL
  const int64_t* px;       // [B]
  const int64_t* py;       // [B]
  int64_t B;
  int L, H, W, Cm;
  float* data;             // [B,7] = rays_o | rays_v | mask[..., :1]      (dataset.py:376)
  float* rgb;              // [L,B,3] or NULL
  float* rgb_wu;           // [L,B,3] or NULL
  float* lights_out;       // [L,B,3] or NULL
  float* near;             // [B] or NULL                                 (dataset.py:448-458)
  float* far;              // [B] or NULL
};

__global__ void raygen_kernel(RayGenArgs g) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= g.B) return;
  const int64_t x = g.px[b], y = g.py[b];
  // p = Kinv[:3,:3] (x, y, 1)   (dataset.py:365-367); same left-to-right accumulation as a 3-term dot product
  const float fx = (float)x, fy = (float)y;
  float p[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) p[r] = g.kinv[r * 4 + 0] * fx + g.kinv[r * 4 + 1] * fy + g.kinv[r * 4 + 2] * 1.f;
  // rays_v = p / ||p||          (dataset.py:369)
  const float nrm = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
  float v[3] = {p[0] / nrm, p[1] / nrm, p[2] / nrm};
  // rays_v = R v ; rays_o = t   (dataset.py:371-373)
  float d[3], o[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    d[r] = g.pose[r * 4 + 0] * v[0] + g.pose[r * 4 + 1] * v[1] + g.pose[r * 4 + 2] * v[2];
    o[r] = g.pose[r * 4 + 3];
  }
  const int64_t pix = y * g.W + x;
  float* row = g.data + b * 7;
  row[0] = o[0]; row[1] = o[1]; row[2] = o[2];
  row[3] = d[0]; row[4] = d[1]; row[5] = d[2];
  row[6] = g.mask[pix * g.Cm];
  if (g.near) {
    const float a = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    const float bq = 2.f * (o[0] * d[0] + o[1] * d[1] + o[2] * d[2]);
    const float mid = 0.5f * (-bq) / a;
    g.near[b] = mid - 1.f;
    g.far[b] = mid + 1.f;
  }
  const int64_t plane = (int64_t)g.H * g.W * 3;
  for (int l = 0; l < g.L; ++l) {
    const int64_t src = l * plane + pix * 3;
    const int64_t dst = ((int64_t)l * g.B + b) * 3;
    if (g.rgb) { g.rgb[dst] = g.images[src]; g.rgb[dst + 1] = g.images[src + 1]; g.rgb[dst + 2] = g.images[src + 2]; }
    if (g.rgb_wu) {
      g.rgb_wu[dst] = g.images_wu[src]; g.rgb_wu[dst + 1] = g.images_wu[src + 1]; g.rgb_wu[dst + 2] = g.images_wu[src + 2];
    }
    if (g.lights_out) {
      g.lights_out[dst] = g.lights[src]; g.lights_out[dst + 1] = g.lights[src + 1]; g.lights_out[dst + 2] = g.lights[src + 2];
    }
  }
}

}  // namespace rnb

#define RNB_API extern "C" __attribute__((visibility("default")))

RNB_API int rnb_gen_rays_at_view(const float* intrinsics_inv, const float* pose, const float* images,
                                 const float* images_warmup, const float* mask, int32_t mask_channels,
                                 const float* light_directions, const int64_t* pixels_x, const int64_t* pixels_y,
                                 int64_t B, int32_t n_lights, int32_t H, int32_t W, float* data, float* true_rgb,
                                 float* true_rgb_warmup, float* lights_dir, float* near, float* far,
                                 rnb_stream_t stream) {
  using namespace rnb;
  if (!intrinsics_inv || !pose || !mask || !pixels_x || !pixels_y || !data)
    RNB_FAIL(RNB_E_NULL, "rnb_gen_rays_at_view: NULL pointer");
  if ((true_rgb && !images) || (true_rgb_warmup && !images_warmup) || (lights_dir && !light_directions) ||
      ((near == nullptr) != (far == nullptr)))
    RNB_FAIL(RNB_E_NULL, "rnb_gen_rays_at_view: output requested without its source");
  if (B < 1 || n_lights < 0 || H < 1 || W < 1 || mask_channels < 1)
    RNB_FAIL(RNB_E_INVALID, "rnb_gen_rays_at_view: bad shape (B %lld, L %d, H %d, W %d)", (long long)B, n_lights, H, W);
  RayGenArgs g{intrinsics_inv, pose, images, images_warmup, mask, light_directions, pixels_x, pixels_y, B,
               n_lights, H, W, mask_channels, data, true_rgb, true_rgb_warmup, lights_dir, near, far};
  hipLaunchKernelGGL(raygen_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g);
  RNB_CHECK_LAUNCH();
  return RNB_OK;
}
