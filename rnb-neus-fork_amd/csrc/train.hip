// The two small steps of train_rnb that sit either side of the renderer (exp_runner.py:241-263 and the Adam
// update of exp_runner.py:115/:262): the loss with its input gradients in one launch, and Adam over a flat
// parameter buffer in one launch.  In the reference each is a chain of ~25 / ~40 tiny PyTorch kernels; at a
// 6 ms step those chains are 3 % of the wall time.
#include "rnb_internal.h"

namespace rnb {

__device__ inline double block_sum(double v, double* red) {
  // 1024 threads = 16 waves
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
  return s;
}

// loss = L1(color_fine - true_rgb | mask) / (mask_sum * n_lights) + igr_w * gradient_error
//        + mask_w * BCE(clip(weight_sum, 1e-3, 1 - 1e-3), mask)                      (exp_runner.py:241-258)
// loss[0]; parts[0..2] = color_loss, eikonal_loss, mask_loss; d* = d loss / d input.
// Shard form (batch_global != nullptr): normalisers of the WHOLE data-parallel batch, see rnb_loss_rnb_shard.
__global__ __launch_bounds__(1024) void rnb_loss_kernel(const float* __restrict__ color, const float* __restrict__ rgb,
                                                        const float* __restrict__ mask, const float* __restrict__ wsum,
                                                        const float* __restrict__ gerr, int L, int64_t B, int Cd,
                                                        float igr_w, float mask_w,
                                                        const float* __restrict__ batch_global, float eik_share, float* __restrict__ loss,
                                                        float* __restrict__ parts,
                                                        float* __restrict__ dcolor, float* __restrict__ dwsum,
                                                        float* __restrict__ dgerr) {
  __shared__ double red[16];
  const int tid = threadIdx.x, nt = blockDim.x;
  // rays of the whole batch: from the collective in the shard form (batch_global = {mask count, ray count})
  const float B_global = batch_global ? batch_global[1] : (float)B;
  // ---- mask sum and the BCE term -------------------------------------------------------------------
  double msum = 0.0, bce = 0.0;
  for (int64_t b = tid; b < B; b += nt) {
    const float m = mask_w > 0.f ? (mask[b] > 0.5f ? 1.f : 0.f) : 1.f;
    msum += m;
    const float w = wsum[b];
    const float x = clamp_nan(w, 1e-3f, 1.f - 1e-3f);
    // torch.nn.functional.binary_cross_entropy clamps both logs at -100
    const float lx = max_nan(logf(x), -100.f), l1x = max_nan(logf(1.f - x), -100.f);
    bce -= (double)(m * lx + (1.f - m) * l1x);
    const bool pass = w >= 1e-3f && w <= 1.f - 1e-3f;   // clip's sub-gradient (inclusive, like torch.clamp)
    const float d = (x - m) / fmaxf((1.f - x) * x, 1e-12f);
    dwsum[b] = pass ? mask_w * d / B_global : 0.f;
  }
  msum = block_sum(msum, red);
  bce = block_sum(bce, red);
  const float mask_sum = (batch_global ? batch_global[0] : (float)msum) + 1e-5f;   // fp32 `mask.sum() + 1e-5`
  const float inv = 1.f / (mask_sum * (float)L);
  // ---- masked L1 colour term -----------------------------------------------------------------------
  double l1 = 0.0;
  const int64_t n = (int64_t)L * B * Cd;
  for (int64_t i = tid; i < n; i += nt) {
    const int64_t b = (i / Cd) % B;
    const float m = mask_w > 0.f ? (mask[b] > 0.5f ? 1.f : 0.f) : 1.f;
    const float e = (color[i] - rgb[i]) * m;
    l1 += (double)fabsf(e);
    const float sg = e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f);
    dcolor[i] = sg * m * inv;
  }
  l1 = block_sum(l1, red);
  if (tid == 0) {
    const float color_loss = (float)l1 * inv;
    const float mask_loss = (float)(bce / (double)B_global);
    const float eik = gerr[0] * eik_share;
    loss[0] = color_loss + eik * igr_w + mask_loss * mask_w;
    parts[0] = color_loss;
    parts[1] = eik;
    parts[2] = mask_loss;
    dgerr[0] = igr_w;
  }
}

// torch.optim.Adam (amsgrad=False, maximize=False): the arithmetic of PyTorch's fused implementation.
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps, float wd,
                            float bc1, float bc2_sqrt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float grad = g[i];
  const float param = p[i];
  if (wd != 0.f) grad += wd * param;
  float ea = m[i], es = v[i];
  ea = ea + (1.f - b1) * (grad - ea);
  es = b2 * es + (1.f - b2) * grad * grad;
  const float step_size = lr / bc1;
  const float denom = sqrtf(es) / bc2_sqrt + eps;
  p[i] = param - step_size * ea / denom;
  m[i] = ea;
  v[i] = es;
}

}  // namespace rnb

#define RNB_API extern "C" __attribute__((visibility("default")))

RNB_API int rnb_loss_rnb(const float* color_fine, const float* true_rgb, const float* mask, const float* weight_sum,
                         const float* gradient_error, int32_t n_lights, int64_t B, int32_t color_depth,
                         float igr_weight, float mask_weight, float* loss, float* parts,
                         float* d_color_fine, float* d_weight_sum, float* d_gradient_error, rnb_stream_t stream) {
  using namespace rnb;
  if (!color_fine || !true_rgb || !mask || !weight_sum || !gradient_error || !loss || !parts || !d_color_fine ||
      !d_weight_sum || !d_gradient_error)
    RNB_FAIL(RNB_E_NULL, "rnb_loss_rnb: NULL pointer");
  if (n_lights < 1 || B < 1 || color_depth < 1) RNB_FAIL(RNB_E_INVALID, "rnb_loss_rnb: empty shape");
  hipLaunchKernelGGL(rnb_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, color_fine, true_rgb, mask,
                     weight_sum, gradient_error, n_lights, B, color_depth, igr_weight, mask_weight,
                     (const float*)nullptr, 1.f, loss, parts, d_color_fine, d_weight_sum, d_gradient_error);
  RNB_CHECK_LAUNCH();
  return RNB_OK;
}

RNB_API int rnb_loss_rnb_shard(const float* color_fine, const float* true_rgb, const float* mask,
                               const float* weight_sum, const float* gradient_error, int32_t n_lights, int64_t B,
                               int32_t color_depth, float igr_weight, float mask_weight, const float* batch_global,
                               float eik_share, float* loss, float* parts, float* d_color_fine,
                               float* d_weight_sum, float* d_gradient_error, rnb_stream_t stream) {
  using namespace rnb;
  if (!color_fine || !true_rgb || !mask || !weight_sum || !gradient_error || !loss || !parts || !d_color_fine ||
      !d_weight_sum || !d_gradient_error || !batch_global)
    RNB_FAIL(RNB_E_NULL, "rnb_loss_rnb_shard: NULL pointer");
  if (n_lights < 1 || B < 1 || color_depth < 1) RNB_FAIL(RNB_E_INVALID, "rnb_loss_rnb_shard: bad shape");
  hipLaunchKernelGGL(rnb_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, color_fine, true_rgb, mask,
                     weight_sum, gradient_error, n_lights, B, color_depth, igr_weight, mask_weight, batch_global,
                     eik_share, loss, parts, d_color_fine, d_weight_sum, d_gradient_error);
  RNB_CHECK_LAUNCH();
  return RNB_OK;
}

RNB_API int rnb_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr,
                          double beta1, double beta2, double eps, double weight_decay, int64_t step,
                          rnb_stream_t stream) {
  using namespace rnb;
  if (!param || !grad || !exp_avg || !exp_avg_sq) RNB_FAIL(RNB_E_NULL, "rnb_adam_step: NULL pointer");
  if (n < 0 || step < 1) RNB_FAIL(RNB_E_INVALID, "rnb_adam_step: n %lld, step %lld", (long long)n, (long long)step);
  if (n == 0) return RNB_OK;
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, param, grad,
                     exp_avg, exp_avg_sq, n, (float)lr, (float)beta1, (float)beta2, (float)eps, (float)weight_decay,
                     (float)bc1, (float)sqrt(bc2));
  RNB_CHECK_LAUNCH();
  return RNB_OK;
}
