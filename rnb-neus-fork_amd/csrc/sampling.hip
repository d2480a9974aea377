// Hierarchical importance sampling along rays: one wave64 per ray.
//
// Replaces (no autograd, models/renderer.py is under torch.no_grad() here):
//   z initialisation + perturbation       models/renderer.py:558-573
//   NeuSRenderer.up_sample                models/renderer.py:132-176
//   sample_pdf(det=True)                  models/renderer.py:39-69
//   cat_z_vals (concat + sort + gather)   models/renderer.py:178-192
//
// This file is compiled with -ffp-contract=off: every product/sum below rounds exactly where the
// reference's separate PyTorch ops round, so that searchsorted / sort decisions can be bit-exact.
// cumsum / cumprod follow the CPU reference's semantics (running value kept in double, each output
// rounded to fp32); torch.linspace's fused multiply-add is reproduced with explicit fmaf.
#include "rnb_internal.h"

namespace rnb {

constexpr int kMaxZ = 512;    // n + n_new upper bound
constexpr int kMaxNew = 64;


__device__ inline float sigmoidf_ref(float x) { return 1.0f / (1.0f + expf(-x)); }

// `f` is what the double running value `v` of a scan rounds to.  True when every double within 2e-13 relative of v rounds
// to the same fp32 value — the distance by which a differently associated scan of <= 512 terms can differ from the
// sequential one (each of the <= 512 double operations contributes <= 2^-53 relative).  NaN: never safe.
__device__ inline bool scan_safe(double v, float f) {
  return (float)(v * (1.0 - 2e-13)) == f && (float)(v * (1.0 + 2e-13)) == f;
}

// The order torch.sort uses: NaN compares greater than every number (and equal to NaN).  A strict weak order on ALL
// floats, so the rank-count merge below writes every output slot exactly once whatever the depths are.
__device__ inline bool lt_total(float a, float b) { return a < b || (a == a && b != b); }

// z[b,j] = near + (far-near)*linspace(0,1,n)[j]  (+ (t_rand-0.5)*2/n) ; pts = o + d*z
__global__ void z_init_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                              const float* __restrict__ near, const float* __restrict__ far,
                              const float* __restrict__ t_rand, int64_t B, int n, float* __restrict__ z,
                              float* __restrict__ pts) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * n) return;
  const int64_t b = i / n;
  const int j = (int)(i - b * n);
  const float lin = linspace_at(0.f, 1.f, n, j);
  float zz = near[b] + (far[b] - near[b]) * lin;
  if (t_rand != nullptr) {
    const float tr = t_rand[b] - 0.5f;
    zz = zz + (tr * 2.0f) / (float)n;
  }
  z[i] = zz;
  if (pts != nullptr) {
#pragma unroll
    for (int d = 0; d < 3; ++d) pts[i * 3 + d] = rays_o[b * 3 + d] + rays_d[b * 3 + d] * zz;
  }
}

struct UpArgs {
  const float* rays_o;
  const float* rays_d;
  const float* z_in;          // [B,n] sorted
  const float* sdf_old;       // [B,n_old]   (n_old == n when gather_index == nullptr)
  const float* sdf_new;       // [B,n-n_old] or nullptr
  const int32_t* gather_index;// [B,n] or nullptr
  int n_old;
  int n, n_new;
  float inv_s;
  float* new_z;               // [B,n_new]            (optional)
  int32_t* inds;              // [B,n_new]            (optional)
  float* z_out;               // [B,n+n_new]
  int32_t* sort_index;        // [B,n+n_new]          (optional)
  float* new_pts;             // [B*n_new,3]          (optional)
  float* sdf_sorted_out;      // [B,n] the gathered sdf row (optional)
};

__global__ __launch_bounds__(64) void up_sample_kernel(UpArgs a) {
  __shared__ float z[kMaxZ], sd[kMaxZ], cs[kMaxZ], w[kMaxZ], cdf[kMaxZ];
  __shared__ unsigned char ins[kMaxZ];
  __shared__ float nz[kMaxNew];
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.x;
  const int n = a.n, n_new = a.n_new;
  const float o0 = a.rays_o[b * 3], o1 = a.rays_o[b * 3 + 1], o2 = a.rays_o[b * 3 + 2];
  const float d0 = a.rays_d[b * 3], d1 = a.rays_d[b * 3 + 1], d2 = a.rays_d[b * 3 + 2];

  for (int j = lane; j < n; j += 64) {
    const float zz = a.z_in[b * n + j];
    z[j] = zz;
    float s;
    if (a.gather_index != nullptr) {
      // clamped: the index comes from caller memory through the per-step C entry points
      const int idx = min(max(a.gather_index[b * n + j], 0), n - 1);
      s = idx < a.n_old ? a.sdf_old[b * a.n_old + idx] : a.sdf_new[b * (n - a.n_old) + (idx - a.n_old)];
    } else {
      s = a.sdf_old[b * n + j];
    }
    sd[j] = s;
    if (a.sdf_sorted_out) a.sdf_sorted_out[b * n + j] = s;
    const float px = o0 + d0 * zz, py = o1 + d1 * zz, pz = o2 + d2 * zz;
    const float r = sqrtf(px * px + py * py + pz * pz);
    ins[j] = r < 1.0f ? 1 : 0;
  }
  __syncthreads();
  // section cosines (finite differences of the SDF along the ray)
  for (int j = lane; j < n - 1; j += 64) cs[j] = (sd[j + 1] - sd[j]) / (z[j + 1] - z[j] + 1e-5f);
  __syncthreads();
  for (int j = lane; j < n - 1; j += 64) {
    const float prev = j == 0 ? 0.f : cs[j - 1];
    float c = min_nan(prev, cs[j]);
    c = clamp_nan(c, -1e3f, 0.0f);
    c = c * ((ins[j] | ins[j + 1]) ? 1.0f : 0.0f);
    const float dist = z[j + 1] - z[j];
    const float mid = (sd[j] + sd[j + 1]) * 0.5f;
    const float half = c * dist * 0.5f;
    const float prev_cdf = sigmoidf_ref((mid - half) * a.inv_s);
    const float next_cdf = sigmoidf_ref((mid + half) * a.inv_s);
    w[j] = (prev_cdf - next_cdf + 1e-5f) / (prev_cdf + 1e-5f);   // alpha
  }
  __syncthreads();
  // weights = alpha * cumprod([1, 1-alpha+1e-7])[:-1]; then w + 1e-5 (sample_pdf).  torch.cumprod / torch.cumsum on
  // the CPU keep the running value in double and round every output to fp32, one element after the other.  Here every
  // lane scans a contiguous segment and the segments are chained by a wave scan — a different association of the same
  // double products, i.e. a running value within ~n 2^-52 of the sequential one.  That can only change an fp32 OUTPUT
  // when the running value lies that close to an fp32 rounding boundary: every output is therefore checked
  // (scan_safe) and a ray with a single unsafe element is redone serially, so the results are bit-identical to the
  // sequential scan in all cases.
  const int m = n - 1;
  const int per = (m + 63) >> 6;
  const int j0 = min(lane * per, m), j1 = min(j0 + per, m);
  {
    double loc = 1.0;
    for (int j = j0; j < j1; ++j) loc *= (double)(1.0f - w[j] + 1e-7f);
    double inc = loc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const double t = __shfl_up(inc, o, 64);
      if (lane >= o) inc *= t;
    }
    double run = __shfl_up(inc, 1, 64);
    if (lane == 0) run = 1.0;
    bool safe = true;
    for (int j = j0; j < j1; ++j) {
      const float al = w[j];
      const float T = (float)run;
      safe = safe && scan_safe(run, T);
      run *= (double)(1.0f - al + 1e-7f);
      cs[j] = al * T + 1e-5f;             // (cs is free: the section cosines were consumed above)
    }
    if (__any(!safe)) {                    // wave-uniform; practically never taken
      if (lane == 0) {
        double r = 1.0;
        for (int j = 0; j < m; ++j) {
          const float al = w[j];
          const float T = (float)r;
          r *= (double)(1.0f - al + 1e-7f);
          cs[j] = al * T + 1e-5f;
        }
      }
    }
  }
  __syncthreads();
  // pdf = w / sum(w); cdf = [0, cumsum(pdf)]
  double part = 0.0;
  for (int j = lane; j < m; j += 64) part += (double)cs[j];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  const float wsum = (float)part;
  for (int j = lane; j < m; j += 64) w[j] = cs[j] / wsum;
  __syncthreads();
  {
    double loc = 0.0;
    for (int j = j0; j < j1; ++j) loc += (double)w[j];
    double inc = loc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const double t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    double run = __shfl_up(inc, 1, 64);
    if (lane == 0) { run = 0.0; cdf[0] = 0.f; }
    // The pdf values are fp32 numbers in [2^-27, 4) (each raw weight is >= 1e-5 of a sum <= ~1): every partial sum of any
    // subset then fits 53 bits, i.e. the double additions are EXACT in any order and the chained scan equals the
    // sequential one bit for bit (checked per element; anything else — NaN, a degenerate row — goes the serial way).
    // (scan_safe would be wrong here: an exact sum of two floats is often exactly half-way between two fp32 values.)
    bool safe = true;
    for (int j = j0; j < j1; ++j) {
      const float wj = w[j];
      safe = safe && (wj >= 7.450580596923828e-09f) && (wj < 4.0f);
      run += (double)wj;
      cdf[j + 1] = (float)run;
    }
    if (__any(!safe)) {
      if (lane == 0) {
        double r = 0.0;
        for (int j = 0; j < m; ++j) {
          r += (double)w[j];
          cdf[j + 1] = (float)r;
        }
      }
    }
  }
  __syncthreads();
  // inverse-CDF sampling at u = linspace(0.5/n_new, 1-0.5/n_new, n_new); searchsorted(right=True)
  if (lane < n_new) {
    const float u = linspace_at((float)(0.5 / (double)n_new), (float)(1.0 - 0.5 / (double)n_new), n_new, lane);
    int lo = 0, hi = n;                 // first index with cdf[idx] > u; `!(c > u)` as ATen's upper bound writes it,
    while (lo < hi) {                   // so a NaN CDF (diverged model) sends the search right: ind = n, new_z = NaN
      const int mid = (lo + hi) >> 1;
      if (!(cdf[mid] > u)) lo = mid + 1; else hi = mid;
    }
    const int ind = lo;
    const int below = max(ind - 1, 0);
    const int above = min(n - 1, ind);
    const float c0 = cdf[below], c1 = cdf[above];
    const float b0 = z[below], b1 = z[above];
    float denom = c1 - c0;
    if (denom < 1e-5f) denom = 1.0f;
    const float t = (u - c0) / denom;
    const float v = b0 + t * (b1 - b0);
    nz[lane] = v;
    if (a.inds) a.inds[b * n_new + lane] = ind;
    if (a.new_z) a.new_z[b * n_new + lane] = v;
    if (a.new_pts) {
      float* p = a.new_pts + (b * n_new + lane) * 3;
      p[0] = o0 + d0 * v; p[1] = o1 + d1 * v; p[2] = o2 + d2 * v;
    }
  }
  __syncthreads();
  // stable merge of the sorted old depths with the new ones (== torch.sort over cat[z, new_z]) by rank counting under
  // lt_total: old depth j lands at j + #{new < it}, new depth k at #{old <= it} + #{new before it} — a permutation of
  // 0..nt-1 for ANY depths sorted under that order, NaNs included (they go last, as torch.sort puts them).  The row is
  // pre-filled first, so that even depths that violate the sortedness precondition (caller memory through the per-step
  // entry point) leave no slot of z_out / sort_index unwritten.
  const int nt = n + n_new;
  for (int j = lane; j < nt; j += 64) {
    a.z_out[b * nt + j] = __builtin_nanf("");
    if (a.sort_index) a.sort_index[b * nt + j] = 0;
  }
  __syncthreads();
  for (int j = lane; j < n; j += 64) {
    const float zz = z[j];
    int cnt = 0;
    for (int k = 0; k < n_new; ++k) cnt += lt_total(nz[k], zz) ? 1 : 0;
    const int pos = j + cnt;
    a.z_out[b * nt + pos] = zz;
    if (a.sort_index) a.sort_index[b * nt + pos] = j;
  }
  if (lane < n_new) {
    const float v = nz[lane];
    int lo = 0, hi = n;                 // number of old depths <= v (total order)
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (!lt_total(v, z[mid])) lo = mid + 1; else hi = mid;
    }
    int cnt = lo;
    for (int k = 0; k < n_new; ++k) cnt += (lt_total(nz[k], v) || (!lt_total(v, nz[k]) && k < lane)) ? 1 : 0;
    a.z_out[b * nt + cnt] = v;
    if (a.sort_index) a.sort_index[b * nt + cnt] = n + lane;
  }
}

__global__ void gather_sdf_kernel(const float* __restrict__ sdf_old, const float* __restrict__ sdf_new,
                                  const int32_t* __restrict__ index, int64_t B, int n, int n_new,
                                  float* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int nt = n + n_new;
  if (i >= B * nt) return;
  const int64_t b = i / nt;
  const int idx = min(max(index[i], 0), nt - 1);   // clamped: the index comes from caller memory
  out[i] = idx < n ? sdf_old[b * n + idx] : sdf_new[b * n_new + (idx - n)];
}

int launch_z_init(const float* rays_o, const float* rays_d, const float* near, const float* far,
                  const float* t_rand, int64_t B, int n, float* z, float* pts, hipStream_t s) {
  const int64_t tot = B * n;
  hipLaunchKernelGGL(z_init_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, rays_o, rays_d, near, far,
                     t_rand, B, n, z, pts);
  RNB_CHECK_LAUNCH();
  return RNB_OK;
}

int launch_up_sample_step(const float* rays_o, const float* rays_d, const float* z_in, const float* sdf_old,
                          const float* sdf_new, const int32_t* gather_index, int n_old_for_gather, int64_t B,
                          int n, int n_new, float inv_s, float* new_z, int32_t* inds, float* z_out,
                          int32_t* sort_index, float* new_pts, float* sdf_sorted_out, hipStream_t s) {
  if (n < 2 || n_new < 1 || n_new > kMaxNew || n + n_new > kMaxZ)
    RNB_FAIL(RNB_E_INVALID, "up_sample: unsupported sizes n=%d n_new=%d (n_new <= %d, n+n_new <= %d)", n, n_new,
             kMaxNew, kMaxZ);
  UpArgs a{rays_o, rays_d, z_in, sdf_old, sdf_new, gather_index, gather_index ? n_old_for_gather : n, n, n_new,
           inv_s, new_z, inds, z_out, sort_index, new_pts, sdf_sorted_out};
  hipLaunchKernelGGL(up_sample_kernel, dim3((unsigned)B), dim3(64), 0, s, a);
  RNB_CHECK_LAUNCH();
  return RNB_OK;
}

int launch_gather_sdf(const float* sdf_old, const float* sdf_new, const int32_t* index, int64_t B, int n, int n_new,
                      float* out, hipStream_t s) {
  const int64_t tot = B * (n + n_new);
  hipLaunchKernelGGL(gather_sdf_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, sdf_old, sdf_new, index,
                     B, n, n_new, out);
  RNB_CHECK_LAUNCH();
  return RNB_OK;
}

}  // namespace rnb
