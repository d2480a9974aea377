// Weight-norm materialisation W = g * v / ||v||_row (torch.nn.utils.weight_norm as applied at
// models/fields.py:72-74 and :168-170) into the packed layout, and its backward.
#include "rnb_internal.h"

namespace rnb {

struct WnEntry {
  const float* g;   // [rows_src] or nullptr (no weight norm)
  const float* v;   // [rows_src, K]
  const float* b;   // [rows_src]
  float* dg;        // backward outputs (same shapes) — nullptr in the forward
  float* dv;
  float* db;
  int src_row0;     // first source row handled by this entry
  int N;            // real rows handled
  int Nrows;        // rows of the packed block to write (>= N; the rest is zero)
  int K, Kp;
  int row_begin;    // first global row (block index) of this entry
  int cmap_f;       // >0: albedo layer 0 column permutation, value = F
  int cmap_2pev;
  long long w_off, b_off;
  long long wT_off;  // transposed copy [Kp x Nrows] or -1
  float scale;
  int skip;          // this is the skip layer: `scale` stands for 1 / sqrt(2), taken at fp64 precision by both kernels
};
__device__ inline double wn_scale(const WnEntry& en) { return en.skip ? 0.70710678118654752440 : (double)en.scale; }
constexpr int kMaxEntries = 2 * RNB_MAX_LIN + 2;
struct WnTable {
  int n;
  int total_rows;
  WnEntry e[kMaxEntries];
};

__device__ inline int cmap(const WnEntry& en, int i) {
  if (en.cmap_f <= 0) return i;
  return i < en.cmap_2pev ? en.cmap_f + i : i - en.cmap_2pev;
}

__device__ inline float block_sum(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
  return t;
}

__device__ inline double block_sum_f64(double v, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
  return t;
}

// The row factor g / ||v|| multiplies a whole row: an fp32 rounding error in it is the SAME relative error in every
// product of that row, for every point — for the SDF output row a common offset of the whole SDF field (measured:
// -1.1e-7 with the factor in fp32, profiles/r04_sdf_bias.txt), which a sharp surface (inv_s ~ 400) turns into twice the
// weight_sum error of the fp32 CPU oracle.  So the factor is formed in fp64 and every W element is rounded once, from
// the fp64 product (675k elements per step: nothing).
__global__ __launch_bounds__(256) void wn_fwd_kernel(WnTable tab, float* __restrict__ packed, unsigned* __restrict__ wmax_zero) {
  __shared__ double red[4];
  // x2h: the per-matrix maxima of the mirror's scale table start from zero (x3_pack_kernel, the next launch, grows them)
  if (wmax_zero != nullptr && blockIdx.x == 0 && threadIdx.x < kH2TabSlots) wmax_zero[threadIdx.x] = 0u;
  int ei = 0;
  while (ei + 1 < tab.n && (int)blockIdx.x >= tab.e[ei + 1].row_begin) ++ei;
  const WnEntry& en = tab.e[ei];
  const int r = blockIdx.x - en.row_begin;
  float* wrow = packed + en.w_off + (long long)r * en.Kp;
  float* wT = en.wT_off >= 0 ? packed + en.wT_off + r : nullptr;   // column r of W^T, stride Nrows
  if (r >= en.N) {
    for (int c = threadIdx.x; c < en.Kp; c += blockDim.x) {
      wrow[c] = 0.f;
      if (wT) wT[(long long)c * en.Nrows] = 0.f;
    }
    if (threadIdx.x == 0) packed[en.b_off + r] = 0.f;
    return;
  }
  const int src = en.src_row0 + r;
  const float* vrow = en.v + (long long)src * en.K;
  // the skip layer's 1/sqrt(2) (models/fields.py:93 divides the activations by np.sqrt(2)) at fp64 precision
  const double scale = wn_scale(en);
  double mult = scale;
  if (en.g != nullptr) {
    double ss = 0.0;
    for (int i = threadIdx.x; i < en.K; i += blockDim.x) ss = fma((double)vrow[i], (double)vrow[i], ss);
    ss = block_sum_f64(ss, red);
    mult = scale * ((double)en.g[src] / sqrt(ss));
  }
  for (int i = threadIdx.x; i < en.K; i += blockDim.x) {
    const int c = cmap(en, i);
    const float w = (float)((double)vrow[i] * mult);
    wrow[c] = w;
    if (wT) wT[(long long)c * en.Nrows] = w;
  }
  for (int c = en.K + threadIdx.x; c < en.Kp; c += blockDim.x) {
    wrow[c] = 0.f;
    if (wT) wT[(long long)c * en.Nrows] = 0.f;
  }
  if (threadIdx.x == 0) packed[en.b_off + r] = en.b[src];
}

__global__ __launch_bounds__(256) void wn_bwd_kernel(WnTable tab, const float* __restrict__ pgrad) {
  __shared__ double red[4];
  int ei = 0;
  while (ei + 1 < tab.n && (int)blockIdx.x >= tab.e[ei + 1].row_begin) ++ei;
  const WnEntry& en = tab.e[ei];
  const int r = blockIdx.x - en.row_begin;
  if (r >= en.N) return;
  const int src = en.src_row0 + r;
  const float* vrow = en.v + (long long)src * en.K;
  const float* dwrow = pgrad + en.w_off + (long long)r * en.Kp;
  float* dvrow = en.dv + (long long)src * en.K;
  // the same row factor as the forward's (fp64: scale * g / ||v||), so that W and dW/dv, dW/dg describe one function
  const double scale = wn_scale(en);
  if (en.g == nullptr) {
    for (int i = threadIdx.x; i < en.K; i += blockDim.x) dvrow[i] = (float)(scale * (double)dwrow[cmap(en, i)]);
  } else {
    double ss = 0.0, t = 0.0;
    for (int i = threadIdx.x; i < en.K; i += blockDim.x) {
      const double vv = (double)vrow[i];
      ss = fma(vv, vv, ss);
      t = fma((double)dwrow[cmap(en, i)], vv, t);
    }
    ss = block_sum_f64(ss, red);
    t = block_sum_f64(t, red);
    const double inv_norm = 1.0 / sqrt(ss);
    const double coef = scale * (double)en.g[src] * inv_norm;
    const double proj = t * inv_norm * inv_norm;   // (v . dW) / ||v||^2
    for (int i = threadIdx.x; i < en.K; i += blockDim.x)
      dvrow[i] = (float)(coef * ((double)dwrow[cmap(en, i)] - (double)vrow[i] * proj));
    if (threadIdx.x == 0) en.dg[src] = (float)(scale * t * inv_norm);
  }
  if (threadIdx.x == 0) en.db[src] = pgrad[en.b_off + r];
}

static void add_entry(WnTable& t, const rnb_mlp_params* p, const rnb_mlp_grads* g, int lin, bool wn, int src_row0,
                      int N, int Nrows, int K, int Kp, long long w_off, long long b_off, float scale, int cmap_f,
                      int cmap_2pev, long long wT_off = -1, int skip = 0) {
  WnEntry& e = t.e[t.n];
  e.g = wn ? p->g[lin] : nullptr;
  e.v = p->v[lin];
  e.b = p->b[lin];
  e.dg = (g && wn) ? g->g[lin] : nullptr;
  e.dv = g ? g->v[lin] : nullptr;
  e.db = g ? g->b[lin] : nullptr;
  e.src_row0 = src_row0;
  e.N = N;
  e.Nrows = Nrows;
  e.K = K;
  e.Kp = Kp;
  e.row_begin = t.total_rows;
  e.cmap_f = cmap_f;
  e.cmap_2pev = cmap_2pev;
  e.w_off = w_off;
  e.wT_off = wT_off;
  e.b_off = b_off;
  e.scale = scale;
  e.skip = skip;
  t.total_rows += Nrows;
  t.n++;
}

static int build_table(const rnb_model_desc* d, const Layout& L, const rnb_mlp_params* sdf,
                       const rnb_mlp_params* color, const rnb_mlp_grads* gs, const rnb_mlp_grads* gc, WnTable& t) {
  t.n = 0;
  t.total_rows = 0;
  if (!sdf && !color) RNB_FAIL(RNB_E_NULL, "both parameter sets are NULL");
  if (sdf) {
    if (sdf->n_lin != L.nh + 1) RNB_FAIL(RNB_E_INVALID, "sdf params: n_lin %d != %d", sdf->n_lin, L.nh + 1);
    const bool wn = d->sdf_weight_norm != 0;
    for (int l = 0; l <= L.nh; ++l)
      if (!sdf->v[l] || !sdf->b[l] || (wn && !sdf->g[l])) RNB_FAIL(RNB_E_NULL, "sdf lin%d has a NULL leaf", l);
    for (int l = 0; l < L.nh; ++l) {
      const Lin& ln = L.hid[l];
      add_entry(t, sdf, gs, l, wn, 0, ln.N, ln.Np, ln.K, ln.Kp, ln.w_off, ln.b_off, ln.scale, 0, 0, ln.wT_off, l == L.skip ? 1 : 0);
    }
    // output layer: row 0 -> sdf head, rows 1.. -> feature head
    add_entry(t, sdf, gs, L.nh, wn, 0, 1, 1, L.H, L.Hp, L.wsdf_off, L.bsdf_off, 1.f, 0, 0);
    if (L.F > 0)
      add_entry(t, sdf, gs, L.nh, wn, 1, L.F, L.feat.Np, L.H, L.feat.Kp, L.feat.w_off, L.feat.b_off, 1.f, 0, 0,
                L.feat.wT_off);
  }
  if (color) {
    if (L.F <= 0) RNB_FAIL(RNB_E_INVALID, "albedo network needs a feature head");
    if (color->n_lin != L.nc + 1) RNB_FAIL(RNB_E_INVALID, "color params: n_lin %d != %d", color->n_lin, L.nc + 1);
    const bool wc = d->col_weight_norm != 0;
    for (int l = 0; l <= L.nc; ++l)
      if (!color->v[l] || !color->b[l] || (wc && !color->g[l])) RNB_FAIL(RNB_E_NULL, "color lin%d has a NULL leaf", l);
    for (int l = 0; l < L.nc; ++l) {
      const Lin& ln = L.col[l];
      add_entry(t, color, gc, l, wc, 0, ln.N, ln.Np, ln.K, ln.Kp, ln.w_off, ln.b_off, 1.f, l == 0 ? L.F : 0,
                l == 0 ? 2 * L.pev : 0, ln.wT_off);
    }
    add_entry(t, color, gc, L.nc, wc, 0, L.colo.N, L.colo.Np, L.colo.K, L.colo.Kp, L.colo.w_off, L.colo.b_off, 1.f, 0, 0);
  }
  return RNB_OK;
}

int weightnorm_fwd(const rnb_model_desc* d, const Layout& L, const rnb_mlp_params* sdf, const rnb_mlp_params* color,
                   float* packed, hipStream_t s) {
  WnTable t;
  RNB_TRY(build_table(d, L, sdf, color, nullptr, nullptr, t));
  H2Tab* h2 = h2_tab(L, packed);
  hipLaunchKernelGGL(wn_fwd_kernel, dim3(t.total_rows), dim3(256), 0, s, t, packed, h2 ? h2->wmax : (unsigned*)nullptr);
  RNB_CHECK_LAUNCH();
  return RNB_OK;
}

int weightnorm_bwd(const rnb_model_desc* d, const Layout& L, const rnb_mlp_params* sdf, const rnb_mlp_params* color,
                   const float* pgrad, const rnb_mlp_grads* gs, const rnb_mlp_grads* gc, hipStream_t s) {
  if (sdf && !gs) RNB_FAIL(RNB_E_NULL, "sdf grads are NULL");
  if (color && !gc) RNB_FAIL(RNB_E_NULL, "color grads are NULL");
  WnTable t;
  RNB_TRY(build_table(d, L, sdf, color, gs, gc, t));
  for (int i = 0; i < t.n; ++i)
    if (!t.e[i].dv || !t.e[i].db || (t.e[i].g && !t.e[i].dg)) RNB_FAIL(RNB_E_NULL, "a gradient leaf is NULL");
  hipLaunchKernelGGL(wn_bwd_kernel, dim3(t.total_rows), dim3(256), 0, s, t, pgrad);
  RNB_CHECK_LAUNCH();
  return RNB_OK;
}

}  // namespace rnb
