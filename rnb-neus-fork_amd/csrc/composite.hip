// Per-ray part of the render cores and wrappers: one wave64 per ray.
//
//   mid-points / dists                          models/renderer.py:209-214 (== :478-483)
//   cos annealing, sigmoid CDFs, alpha, weights  models/renderer.py:228-262 (== :503-534)
//   colour composite                            render: :265-267 ; render_rnb[_warmup]: :905-914 / :1009-1017
//   eikonal term                                 models/renderer.py:270-272 (== :538-540)
// and the explicit backward of all of it (oracle/explicit.py::composite_backward is the statement).
#include "rnb_internal.h"

namespace rnb {

constexpr int kMaxS = 512;
constexpr int kMaxL = 8;

__device__ inline float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ inline float inv_s_from_variance(const float* variance, bool* inside_clip) {
  const float raw = expf(variance[0] * 10.0f);
  if (inside_clip) *inside_clip = (raw >= 1e-6f) && (raw <= 1e6f);
  return clamp_nan(raw, 1e-6f, 1e6f);
}

// pts[b,j] = o + d * (z + dists/2),  dists[b,j] = z[j+1]-z[j] (last: sample_dist)
__global__ void fine_points_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                   const float* __restrict__ z, int64_t B, int S, float sample_dist,
                                   float* __restrict__ pts, float* __restrict__ dists, unsigned* __restrict__ smax) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // first kernel of a render forward: the maxima of the saved state (PointBufs::smax) start from zero
  if (smax != nullptr && i < SMAX_SLOTS) smax[i] = 0u;
  if (i >= B * S) return;
  const int64_t b = i / S;
  const int j = (int)(i - b * S);
  const float zz = z[i];
  const float dd = j + 1 < S ? z[i + 1] - zz : sample_dist;
  const float mid = zz + dd * 0.5f;
  dists[i] = dd;
#pragma unroll
  for (int d = 0; d < 3; ++d) pts[i * 3 + d] = rays_o[b * 3 + d] + rays_d[b * 3 + d] * mid;
}


__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max_nan(v, __shfl_xor(v, o, 64));
  return v;
}
// inclusive product scan across the wave
__device__ inline float wave_scan_mul(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_up(v, o, 64);
    if (lane >= o) v *= t;
  }
  return v;
}

struct SampleState {
  float alpha, raw, pc, nc, e_prev, e_next, tc;
};

__device__ inline SampleState eval_sample(float s, float n0, float n1, float n2, float d0, float d1, float d2,
                                          float delta, float inv_s, float c) {
  SampleState st;
  st.tc = d0 * n0 + d1 * n1 + d2 * n2;
  const float ic = -(relu_nan(-st.tc * 0.5f + 0.5f) * (1.0f - c) + relu_nan(-st.tc) * c);
  st.e_next = s + ic * delta * 0.5f;
  st.e_prev = s - ic * delta * 0.5f;
  st.pc = sigm(st.e_prev * inv_s);
  st.nc = sigm(st.e_next * inv_s);
  st.raw = (st.pc - st.nc + 1e-5f) / (st.pc + 1e-5f);
  st.alpha = clamp_nan(st.raw, 0.f, 1.f);
  return st;
}

__device__ inline void load_light(const CompArgs& a, int l, int64_t b, float (&Lv)[3]) {
  const float* p = (a.flags & RNB_FLAG_LIGHT_PER_RAY) ? a.lights + ((int64_t)l * a.B + b) * 3 : a.lights + l * 3;
  Lv[0] = p[0]; Lv[1] = p[1]; Lv[2] = p[2];
}

__global__ __launch_bounds__(64) void composite_fwd_kernel(CompArgs a) {
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.x;
  const int S = a.S;
  const bool mvps = (a.flags & RNB_MODE_MVPS) != 0;
  const bool relu_sh = (a.flags & RNB_FLAG_RELU_SHADING) != 0;
  const bool no_alb = (a.flags & RNB_FLAG_NO_ALBEDO) != 0;
  const float inv_s = inv_s_from_variance(a.variance, nullptr);
  const float d0 = a.rays_d[b * 3], d1 = a.rays_d[b * 3 + 1], d2 = a.rays_d[b * 3 + 2];
  float Lv[kMaxL][3];
  if (mvps)
    for (int l = 0; l < a.L; ++l) load_light(a, l, b, Lv[l]);

  float carry = 1.0f;              // transmittance entering the current 64-sample chunk
  float col[kMaxL][4];
#pragma unroll
  for (int l = 0; l < kMaxL; ++l)
#pragma unroll
    for (int c = 0; c < 4; ++c) col[l][c] = 0.f;
  float wsum = 0.f, wmax = -1.f, gnum = 0.f, gden = 0.f;

  for (int j0 = 0; j0 < S; j0 += 64) {
    const int j = j0 + lane;
    const bool on = j < S;
    const int64_t p = b * S + (on ? j : S - 1);
    const float s = a.sdf[p];
    const float n0 = a.nrm[p * 4], n1 = a.nrm[p * 4 + 1], n2 = a.nrm[p * 4 + 2];
    const SampleState st = eval_sample(s, n0, n1, n2, d0, d1, d2, a.dists[p], inv_s, a.cos_anneal);
    const float x = on ? (1.0f - st.alpha + 1e-7f) : 1.0f;
    const float incl = wave_scan_mul(x, lane);
    float excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 1.0f;
    const float T = carry * excl;
    carry = carry * __shfl(incl, 63, 64);
    const float w = on ? st.alpha * T : 0.f;
    const float px = a.pts[p * 3], py = a.pts[p * 3 + 1], pz = a.pts[p * 3 + 2];
    const float pn = sqrtf(px * px + py * py + pz * pz);
    const float nn = sqrtf(n0 * n0 + n1 * n1 + n2 * n2);
    if (on) {
      a.weights[p] = w;
      a.cdf[p] = st.pc;
      a.gradients[p * 3] = n0; a.gradients[p * 3 + 1] = n1; a.gradients[p * 3 + 2] = n2;
      a.inside[p] = pn < 1.0f ? 1.f : 0.f;
      if (a.sdf_out) a.sdf_out[p] = s;
      if (a.albedo_out)
        for (int c = 0; c < a.C; ++c) a.albedo_out[p * a.C + c] = a.alb[p * 4 + c];
      wsum += w;
      wmax = max_nan(wmax, w);
      {   // relax_inside_sphere as a FACTOR (models/renderer.py:538-540): 0 * NaN stays NaN, as in the reference
        const float relax = pn < 1.2f ? 1.f : 0.f;
        gnum += relax * ((nn - 1.0f) * (nn - 1.0f));
        gden += relax;
      }
      float al[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) al[c] = no_alb ? 1.0f : a.alb[p * 4 + c];
      if (mvps) {
        for (int l = 0; l < a.L; ++l) {
          float sh = n0 * Lv[l][0] + n1 * Lv[l][1] + n2 * Lv[l][2];
          if (relu_sh) sh = relu_nan(sh);
          const float ws = w * sh;
#pragma unroll
          for (int c = 0; c < 4; ++c) col[l][c] = fmaf(al[c], ws, col[l][c]);
        }
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) col[0][c] = fmaf(a.alb[p * 4 + c], w, col[0][c]);
      }
    }
  }
  wsum = wave_sum(wsum);
  wmax = wave_max(wmax);
  gnum = wave_sum(gnum);
  gden = wave_sum(gden);
  const int nl = mvps ? a.L : 1;
  for (int l = 0; l < nl; ++l)
    for (int c = 0; c < a.C; ++c) {
      float v = wave_sum(col[l][c]);
      if (lane == 0) {
        if (mvps) a.color_fine[((int64_t)l * a.B + b) * a.C + c] = v;
        else if (c < 3) {
          if (a.bg) v += a.bg[c] * (1.0f - wsum);
          a.color_fine[b * 3 + c] = v;
        }
      }
    }
  if (lane == 0) {
    a.weight_sum[b] = wsum;
    a.weight_max[b] = wmax;
    a.s_val[b] = 1.0f / inv_s;
    a.gerr_part[b * 2] = gnum;
    a.gerr_part[b * 2 + 1] = gden;
  }
}

// gradient_error = sum_b num_b / (sum_b den_b + 1e-5); keeps the denominator for the backward.
// (Round 5 folded this into composite_fwd_kernel — the last workgroup to arrive did the sum — and measured it: the ticket
// atomic that every one of the 512 one-wave workgroups must wait for took the kernel from 15 to 36 us; a 5 us launch is cheaper.)
__global__ __launch_bounds__(256) void gerr_finalize_kernel(const float* __restrict__ part, int64_t B,
                                                            float* __restrict__ gerr, float* __restrict__ den_out,
                                                            float* __restrict__ partial) {
  __shared__ float rn[4], rd[4];
  float n = 0.f, d = 0.f;
  for (int64_t i = threadIdx.x; i < B; i += 256) { n += part[i * 2]; d += part[i * 2 + 1]; }
  n = wave_sum(n);
  d = wave_sum(d);
  if ((threadIdx.x & 63) == 0) { rn[threadIdx.x >> 6] = n; rd[threadIdx.x >> 6] = d; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float nn = rn[0] + rn[1] + rn[2] + rn[3];
    const float ds = rd[0] + rd[1] + rd[2] + rd[3];
    const float dd = ds + 1e-5f;
    gerr[0] = nn / dd;
    den_out[0] = dd;
    if (partial) { partial[0] = nn; partial[1] = ds; }   // this shard's sums, for the data-parallel exact loss
  }
}


__global__ __launch_bounds__(64) void composite_bwd_kernel(CompBwdArgs g) {
  __shared__ float sAlpha[kMaxS], sT[kMaxS], sWbar[kMaxS], sSuf[kMaxS];
  // first kernel of a backward: the maxima of the adjoint tensors (PointBufs::amax) start from zero (was a memset launch)
  if (g.amax_to_zero != nullptr && blockIdx.x == 0) {
    static_assert(AMAX_SLOTS <= 64, "one wave zeroes the slots");
    if ((int)threadIdx.x < AMAX_SLOTS) g.amax_to_zero[threadIdx.x] = 0u;
  }
  const CompArgs& a = g.f;
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.x;
  const int S = a.S;
  const bool mvps = (a.flags & RNB_MODE_MVPS) != 0;
  const bool relu_sh = (a.flags & RNB_FLAG_RELU_SHADING) != 0;
  const bool no_alb = (a.flags & RNB_FLAG_NO_ALBEDO) != 0;
  const float inv_s = inv_s_from_variance(a.variance, nullptr);
  const float c_an = a.cos_anneal;
  const float d0 = a.rays_d[b * 3], d1 = a.rays_d[b * 3 + 1], d2 = a.rays_d[b * 3 + 2];
  float Lv[kMaxL][3], Cb[kMaxL][4];
#pragma unroll
  for (int l = 0; l < kMaxL; ++l)
#pragma unroll
    for (int c = 0; c < 4; ++c) Cb[l][c] = 0.f;
  const int nl = mvps ? a.L : 1;
  if (mvps)
    for (int l = 0; l < a.L; ++l) load_light(a, l, b, Lv[l]);
  if (g.g_color) {
    for (int l = 0; l < nl; ++l)
      for (int c = 0; c < (mvps ? a.C : 3); ++c)
        Cb[l][c] = mvps ? g.g_color[((int64_t)l * a.B + b) * a.C + c] : g.g_color[b * 3 + c];
  }
  const float g_wsum = g.g_weight_sum ? g.g_weight_sum[b] : 0.f;
  const float g_wmax = g.g_weight_max ? g.g_weight_max[b] : 0.f;
  float bg_dot = 0.f;
  if (!mvps && a.bg) bg_dot = Cb[0][0] * a.bg[0] + Cb[0][1] * a.bg[1] + Cb[0][2] * a.bg[2];

  // pass 1: recompute alpha, T ; wbar ; direct albedo / shading cotangents
  float carry = 1.0f;
  float wmax = -1.f;
  int wmax_j = 0;
  for (int j0 = 0; j0 < S; j0 += 64) {
    const int j = j0 + lane;
    const bool on = j < S;
    const int64_t p = b * S + (on ? j : S - 1);
    const float n0 = a.nrm[p * 4], n1 = a.nrm[p * 4 + 1], n2 = a.nrm[p * 4 + 2];
    const SampleState st = eval_sample(a.sdf[p], n0, n1, n2, d0, d1, d2, a.dists[p], inv_s, c_an);
    const float x = on ? (1.0f - st.alpha + 1e-7f) : 1.0f;
    const float incl = wave_scan_mul(x, lane);
    float excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 1.0f;
    const float T = carry * excl;
    carry = carry * __shfl(incl, 63, 64);
    if (on) {
      const float w = st.alpha * T;
      float wb = g_wsum + (g.g_weights ? g.g_weights[p] : 0.f);
      float nb0 = 0.f, nb1 = 0.f, nb2 = 0.f;
      float ab[4] = {0.f, 0.f, 0.f, 0.f};
      float al[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) al[c] = (mvps && no_alb) ? 1.0f : a.alb[p * 4 + c];
      if (mvps) {
        for (int l = 0; l < a.L; ++l) {
          const float sh_raw = n0 * Lv[l][0] + n1 * Lv[l][1] + n2 * Lv[l][2];
          const float sh = relu_sh ? fmaxf(sh_raw, 0.f) : sh_raw;
          float ca = 0.f;                       // sum_c Cb * albedo
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            ca = fmaf(Cb[l][c], al[c], ca);
            ab[c] = fmaf(Cb[l][c], w * sh, ab[c]);
          }
          wb = fmaf(ca, sh, wb);
          float shb = ca * w;
          if (relu_sh && !(sh_raw > 0.f)) shb = 0.f;
          nb0 = fmaf(shb, Lv[l][0], nb0); nb1 = fmaf(shb, Lv[l][1], nb1); nb2 = fmaf(shb, Lv[l][2], nb2);
        }
        if (no_alb) { ab[0] = ab[1] = ab[2] = ab[3] = 0.f; }
      } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) { wb = fmaf(Cb[0][c], al[c], wb); ab[c] = Cb[0][c] * w; }
        wb -= bg_dot;
      }
      sAlpha[j] = st.alpha;
      sT[j] = T;
      sWbar[j] = wb;
      g.nbar[p * 4] = nb0; g.nbar[p * 4 + 1] = nb1; g.nbar[p * 4 + 2] = nb2; g.nbar[p * 4 + 3] = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) g.albbar[p * 4 + c] = ab[c];
      if (w > wmax) { wmax = w; wmax_j = j; }
    }
  }
  // weight_max cotangent goes to the (first) arg-max sample
  if (g.g_weight_max) {
    const float m = wave_max(wmax);
    int cand = (wmax == m) ? wmax_j : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
    __syncthreads();
    if (lane == 0 && cand < S) sWbar[cand] += g_wmax;
  }
  __syncthreads();
  // suffix sums  suf[j] = sum_{k>j} wbar_k * w_k   (w_k = alpha_k T_k)
  if (lane == 0) {
    double run = 0.0;
    for (int j = S - 1; j >= 0; --j) {
      sSuf[j] = (float)run;
      run += (double)sWbar[j] * (double)(sAlpha[j] * sT[j]);
    }
  }
  __syncthreads();
  // pass 2: back through alpha -> cdfs -> sdf / cos -> normal ; eikonal ; inv_s
  const float gerr_coef = g.g_gerr ? g.g_gerr[0] / (g.gerr_den_global ? g.gerr_den_global[0] : g.gerr_den[0]) : 0.f;
  float invs_bar = 0.f;
  for (int j0 = 0; j0 < S; j0 += 64) {
    const int j = j0 + lane;
    if (j >= S) continue;
    const int64_t p = b * S + j;
    const float n0 = a.nrm[p * 4], n1 = a.nrm[p * 4 + 1], n2 = a.nrm[p * 4 + 2];
    const float delta = a.dists[p];
    const SampleState st = eval_sample(a.sdf[p], n0, n1, n2, d0, d1, d2, delta, inv_s, c_an);
    float alphabar = sWbar[j] * sT[j] - sSuf[j] / (1.0f - st.alpha + 1e-7f);
    const float rawbar = (st.raw >= 0.f && st.raw <= 1.f) ? alphabar : 0.f;
    const float den = st.pc + 1e-5f;
    float pcbar = rawbar * (1.0f / den - (st.pc - st.nc + 1e-5f) / (den * den));
    const float ncbar = -rawbar / den;
    if (g.g_cdf) pcbar += g.g_cdf[p];
    float epb = pcbar * st.pc * (1.0f - st.pc);
    float enb = ncbar * st.nc * (1.0f - st.nc);
    invs_bar += epb * st.e_prev + enb * st.e_next;
    epb *= inv_s;
    enb *= inv_s;
    g.sbar[p] = epb + enb;
    const float icbar = (enb - epb) * delta * 0.5f;
    const float tcbar = icbar * (((-st.tc * 0.5f + 0.5f) > 0.f ? 0.5f * (1.0f - c_an) : 0.f) +
                                 ((-st.tc) > 0.f ? c_an : 0.f));
    float nb0 = g.nbar[p * 4] + tcbar * d0, nb1 = g.nbar[p * 4 + 1] + tcbar * d1, nb2 = g.nbar[p * 4 + 2] + tcbar * d2;
    if (g.g_gerr) {
      const float px = a.pts[p * 3], py = a.pts[p * 3 + 1], pz = a.pts[p * 3 + 2];
      if (sqrtf(px * px + py * py + pz * pz) < 1.2f) {
        const float nn = sqrtf(n0 * n0 + n1 * n1 + n2 * n2);
        if (nn > 0.f) {
          const float k = gerr_coef * 2.0f * (nn - 1.0f) / nn;
          nb0 = fmaf(k, n0, nb0); nb1 = fmaf(k, n1, nb1); nb2 = fmaf(k, n2, nb2);
        }
      }
    }
    if (g.g_gradients) {
      nb0 += g.g_gradients[p * 3]; nb1 += g.g_gradients[p * 3 + 1]; nb2 += g.g_gradients[p * 3 + 2];
    }
    g.nbar[p * 4] = nb0; g.nbar[p * 4 + 1] = nb1; g.nbar[p * 4 + 2] = nb2;
  }
  invs_bar = wave_sum(invs_bar);
  if (lane == 0) {
    if (g.g_s_val) invs_bar -= g.g_s_val[b] / (inv_s * inv_s);
    g.invs_part[b] = invs_bar;
  }
}

// d loss / d variance = (sum_b invs_part) * 10 * inv_s * [raw inv_s inside the clip range]
__global__ __launch_bounds__(256) void variance_grad_kernel(const float* __restrict__ part, int64_t B,
                                                            const float* __restrict__ variance,
                                                            float* __restrict__ dvar) {
  __shared__ float red[4];
  float v = 0.f;
  for (int64_t i = threadIdx.x; i < B; i += 256) v += part[i];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    bool inside;
    const float inv_s = inv_s_from_variance(variance, &inside);
    dvar[0] = inside ? (red[0] + red[1] + red[2] + red[3]) * 10.0f * inv_s : 0.f;
  }
}

int launch_fine_points(const float* rays_o, const float* rays_d, const float* z, int64_t B, int S, float sample_dist,
                       float* pts, float* dists, unsigned* smax_to_zero, hipStream_t s) {
  const int64_t tot = B * S;
  static_assert(SMAX_SLOTS <= 256, "one workgroup zeroes the slots");
  hipLaunchKernelGGL(fine_points_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, rays_o, rays_d, z, B, S,
                     sample_dist, pts, dists, smax_to_zero);
  RNB_CHECK_LAUNCH();
  return RNB_OK;
}

int launch_composite_fwd(const CompArgs& a, hipStream_t s) {
  if (a.S > kMaxS) RNB_FAIL(RNB_E_INVALID, "samples per ray %d > %d", a.S, kMaxS);
  if (a.L > kMaxL) RNB_FAIL(RNB_E_INVALID, "n_lights %d > %d", a.L, kMaxL);
  if (a.gerr == nullptr || a.gerr_den == nullptr) RNB_FAIL(RNB_E_NULL, "composite: missing reduction buffers");
  hipLaunchKernelGGL(composite_fwd_kernel, dim3((unsigned)a.B), dim3(64), 0, s, a);
  RNB_CHECK_LAUNCH();
  hipLaunchKernelGGL(gerr_finalize_kernel, dim3(1), dim3(256), 0, s, a.gerr_part, a.B, a.gerr, a.gerr_den, a.gerr_partial);
  RNB_CHECK_LAUNCH();
  return RNB_OK;
}

int launch_composite_bwd(const CompBwdArgs& g, hipStream_t s) {
  if (g.dvar == nullptr) RNB_FAIL(RNB_E_NULL, "composite backward: missing reduction buffers");
  hipLaunchKernelGGL(composite_bwd_kernel, dim3((unsigned)g.f.B), dim3(64), 0, s, g);
  RNB_CHECK_LAUNCH();
  hipLaunchKernelGGL(variance_grad_kernel, dim3(1), dim3(256), 0, s, g.invs_part, g.f.B, g.f.variance, g.dvar);
  RNB_CHECK_LAUNCH();
  return RNB_OK;
}

}  // namespace rnb
