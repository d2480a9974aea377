// C-ABI entry points of librnbneus_hip.so (declared in include/rnbneus.h).
#include <stdlib.h>

#include "rnb_internal.h"

using namespace rnb;

#define RNB_API extern "C" __attribute__((visibility("default")))

#define RNB_REQUIRE(p, name) \
  if (!(p)) RNB_FAIL(RNB_E_NULL, name " is NULL")

#if __has_include("../build/build_id.h")
#include "../build/build_id.h"   // written by __graft_entry__.build_native: hash of csrc/, include/ and the compiler flags
#endif
#ifndef RNB_BUILD_ID
#define RNB_BUILD_ID "unknown"
#endif

RNB_API int rnb_abi_version(void) { return RNB_ABI_VERSION; }
RNB_API const char* rnb_build_id(void) { return RNB_BUILD_ID; }
RNB_API const char* rnb_last_error_string(void) { return rnb::last_error(); }

RNB_API int rnb_packed_floats(const rnb_model_desc* desc, int64_t* n_floats) {
  RNB_REQUIRE(n_floats, "n_floats");
  Layout L;
  RNB_TRY(make_layout(desc, &L));
  *n_floats = L.total_all;
  return RNB_OK;
}

RNB_API int rnb_weightnorm_fwd(const rnb_model_desc* desc, const rnb_mlp_params* sdf, const rnb_mlp_params* color,
                               float* packed, rnb_stream_t stream) {
  RNB_REQUIRE(packed, "packed");
  Layout L;
  RNB_TRY(make_layout(desc, &L));
  RNB_TRY(weightnorm_fwd(desc, L, sdf, color, packed, (hipStream_t)stream));
  if (is_bf16(L)) RNB_TRY(bf16_pack_weights(L, packed, (hipStream_t)stream));   // bf16 mirror behind the fp32 weights
  if (is_x3(L)) RNB_TRY(x3_pack_weights(L, packed, (hipStream_t)stream));       // hi / mid / lo mirror
  return RNB_OK;
}

RNB_API int rnb_weightnorm_bwd(const rnb_model_desc* desc, const rnb_mlp_params* sdf, const rnb_mlp_params* color,
                               const float* packed_grad, const rnb_mlp_grads* sdf_grads,
                               const rnb_mlp_grads* color_grads, rnb_stream_t stream) {
  RNB_REQUIRE(packed_grad, "packed_grad");
  Layout L;
  RNB_TRY(make_layout(desc, &L));
  return weightnorm_bwd(desc, L, sdf, color, packed_grad, sdf_grads, color_grads, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------
// point-wise evaluation
// ---------------------------------------------------------------------------------------------------
static const int kPointsMode = PM_WITH_NORMAL | PM_WITH_COLOR;

RNB_API int rnb_points_workspace_bytes(const rnb_model_desc* desc, int64_t n_points, int64_t* bytes) {
  RNB_REQUIRE(bytes, "bytes");
  if (n_points < 0) RNB_FAIL(RNB_E_INVALID, "n_points < 0");
  Layout L;
  RNB_TRY(make_layout(desc, &L));
  Carver c(nullptr, 0);
  PointBufs pb;
  carve_points(L, c, n_points, kPointsMode, &pb);
  *bytes = (int64_t)c.off;
  return RNB_OK;
}

static int points_setup(const rnb_model_desc* desc, int64_t n, void* ws, size_t ws_bytes, Layout* L, PointBufs* pb) {
  RNB_TRY(make_layout(desc, L));
  RNB_REQUIRE(ws, "workspace");
  Carver c(ws, ws_bytes);
  carve_points(*L, c, n, kPointsMode, pb);
  if (!c.ok) RNB_FAIL(RNB_E_WORKSPACE, "workspace too small: need %zu bytes, have %zu", c.off, ws_bytes);
  pb->smax = nullptr;   // (state maxima are kept for a render's backward only: nothing zeroes them here)
  return RNB_OK;
}

// positional encoding + forward sweep: fused single-launch kernel for the 256-wide network, generic
// per-layer GEMM chain otherwise.  RNB_VARIANT_GENERIC in the descriptor forces the generic path (A/B testing).
static bool use_fused(const Layout& L) { return !(L.variant & RNB_VARIANT_GENERIC) && fused_supported(L); }
static int forward_points(const Layout& L, const float* packed, const float* pts, int64_t n, PointBufs& pb,
                          bool save, bool need_feat, bool need_gz_last, float* feat_dense, hipStream_t s) {
  if (is_bf16(L)) {
    RNB_TRY(bf16_forward(L, packed, pts, n, pb, save, need_feat, s));
    if (need_feat && feat_dense) RNB_TRY(launch_copy_cols(pb.cin, L.Cinp, L.F, n, feat_dense, s));
    return RNB_OK;
  }
  if (use_fused(L)) {
    RNB_TRY(fused_forward(L, packed, pts, n, pb, save, need_feat, need_gz_last, s));
    if (need_feat && feat_dense) RNB_TRY(launch_copy_cols(pb.cin, L.Cinp, L.F, n, feat_dense, s));
    return RNB_OK;
  }
  RNB_TRY(launch_pe_points(L, pts, n, pb, s));
  return sweep_forward(L, packed, pb, need_feat, need_gz_last, feat_dense, s);
}

// reverse-mode normal: fused sweep (seeds itself from D_last) or the generic chain (seeded by the forward)
static int reverse_points(const Layout& L, const float* packed, PointBufs& pb, hipStream_t s) {
  if (is_bf16(L)) return bf16_reverse(L, packed, pb, s);
  if (use_fused(L)) return fused_reverse(L, packed, pb, s);
  return sweep_reverse(L, packed, pb, s);
}

RNB_API int rnb_sdf_forward(const rnb_model_desc* desc, const float* packed, const float* pts, int64_t n,
                            float* sdf_out, float* feat_out, void* ws, size_t ws_bytes, rnb_stream_t stream) {
  RNB_REQUIRE(packed, "packed");
  RNB_REQUIRE(pts, "pts");
  RNB_REQUIRE(sdf_out, "sdf_out");
  if (n <= 0) return n == 0 ? RNB_OK : (set_error("n < 0"), RNB_E_INVALID);
  hipStream_t s = (hipStream_t)stream;
  Layout L;
  PointBufs pb;
  RNB_TRY(points_setup(desc, n, ws, ws_bytes, &L, &pb));
  RNB_TRY(forward_points(L, packed, pts, n, pb, false, feat_out != nullptr, false, feat_out, s));
  RNB_CHECK_HIP(hipMemcpyAsync(sdf_out, pb.sdf, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, s));
  return RNB_OK;
}

RNB_API int rnb_sdf_gradient(const rnb_model_desc* desc, const float* packed, const float* pts, int64_t n,
                             float* grad_out, float* sdf_out, void* ws, size_t ws_bytes, rnb_stream_t stream) {
  RNB_REQUIRE(packed, "packed");
  RNB_REQUIRE(pts, "pts");
  RNB_REQUIRE(grad_out, "grad_out");
  if (n <= 0) return n == 0 ? RNB_OK : (set_error("n < 0"), RNB_E_INVALID);
  hipStream_t s = (hipStream_t)stream;
  Layout L;
  PointBufs pb;
  RNB_TRY(points_setup(desc, n, ws, ws_bytes, &L, &pb));
  RNB_TRY(forward_points(L, packed, pts, n, pb, true, false, !use_fused(L) && !is_bf16(L), nullptr, s));
  RNB_TRY(reverse_points(L, packed, pb, s));
  RNB_TRY(launch_copy_cols(pb.nrm, 4, 3, n, grad_out, s));
  if (sdf_out) RNB_CHECK_HIP(hipMemcpyAsync(sdf_out, pb.sdf, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, s));
  return RNB_OK;
}

RNB_API int rnb_color_forward(const rnb_model_desc* desc, const float* packed, const float* pts, const float* normals,
                              const float* feats, int64_t n, float* out, void* ws, size_t ws_bytes,
                              rnb_stream_t stream) {
  RNB_REQUIRE(packed, "packed");
  RNB_REQUIRE(pts, "pts");
  RNB_REQUIRE(normals, "normals");
  RNB_REQUIRE(feats, "feats");
  RNB_REQUIRE(out, "out");
  if (n <= 0) return n == 0 ? RNB_OK : (set_error("n < 0"), RNB_E_INVALID);
  hipStream_t s = (hipStream_t)stream;
  Layout L;
  PointBufs pb;
  RNB_TRY(points_setup(desc, n, ws, ws_bytes, &L, &pb));
  if (L.F <= 0) RNB_FAIL(RNB_E_INVALID, "model has no feature head");
  RNB_TRY(launch_fill_cols(feats, L.F, n, pb.Mp, L.Cinp, pb.cin, s));
  RNB_TRY(sweep_color(L, packed, pb, pts, normals, 3, s));
  RNB_TRY(launch_copy_cols(pb.alb, 4, L.Co, n, out, s));
  return RNB_OK;
}

// ---------------------------------------------------------------------------------------------------
// SDF grid (extract_fields)
// ---------------------------------------------------------------------------------------------------
constexpr int64_t kGridChunk = 1 << 20;   // points per pass of the generic (per-layer GEMM) path

static int check_grid(const rnb_grid_desc* gd, int64_t* n_points) {
  RNB_REQUIRE(gd, "grid");
  if (gd->resolution < 1 || gd->resolution > 4096) RNB_FAIL(RNB_E_INVALID, "grid resolution out of range (%d)", gd->resolution);
  if (gd->x_begin < 0 || gd->x_end > gd->resolution || gd->x_begin > gd->x_end)
    RNB_FAIL(RNB_E_INVALID, "bad x-slab [%d, %d) of a %d grid", gd->x_begin, gd->x_end, gd->resolution);
  *n_points = (int64_t)(gd->x_end - gd->x_begin) * gd->resolution * gd->resolution;
  return RNB_OK;
}

static GridGen grid_gen_of(const rnb_grid_desc* gd) {
  GridGen g;
  g.on = 1;
  g.res = gd->resolution;
  g.x_begin = gd->x_begin;
  for (int d = 0; d < 3; ++d) { g.bmin[d] = gd->bound_min[d]; g.bmax[d] = gd->bound_max[d]; }
  g.out_scale = gd->out_scale;
  return g;
}

RNB_API int rnb_sdf_grid_workspace_bytes(const rnb_model_desc* desc, const rnb_grid_desc* grid, int64_t* bytes) {
  RNB_REQUIRE(bytes, "bytes");
  Layout L;
  RNB_TRY(make_layout(desc, &L));
  int64_t n;
  RNB_TRY(check_grid(grid, &n));
  if (use_fused(L) || is_bf16(L)) { *bytes = 256; return RNB_OK; }   // the fused sweep keeps everything in LDS
  Carver c(nullptr, 0);
  PointBufs pb;
  c.take<float>(kGridChunk * 3);
  carve_points(L, c, n < kGridChunk ? n : kGridChunk, PM_SDF_ONLY, &pb);
  *bytes = (int64_t)c.off;
  return RNB_OK;
}

RNB_API int rnb_sdf_grid(const rnb_model_desc* desc, const float* packed, const rnb_grid_desc* grid, float* volume,
                         void* ws, size_t ws_bytes, rnb_stream_t stream) {
  RNB_REQUIRE(packed, "packed");
  RNB_REQUIRE(volume, "volume");
  hipStream_t s = (hipStream_t)stream;
  Layout L;
  RNB_TRY(make_layout(desc, &L));
  int64_t n;
  RNB_TRY(check_grid(grid, &n));
  if (n == 0) return RNB_OK;
  const GridGen gg = grid_gen_of(grid);
  if (use_fused(L) || is_bf16(L)) {
    PointBufs pb;
    memset(&pb, 0, sizeof(pb));
    pb.M = n;
    pb.Mp = pad_rows(n);
    pb.sdf = volume;   // grid mode writes rows < M only
    if (is_bf16(L)) return bf16_forward(L, packed, nullptr, n, pb, false, false, s, &gg);
    return fused_forward(L, packed, nullptr, n, pb, false, false, false, s, &gg);
  }
  RNB_REQUIRE(ws, "workspace");
  for (int64_t first = 0; first < n; first += kGridChunk) {
    const int64_t m = n - first < kGridChunk ? n - first : kGridChunk;
    Carver c(ws, ws_bytes);
    float* pts = c.take<float>(kGridChunk * 3);
    PointBufs pb;
    carve_points(L, c, n < kGridChunk ? n : kGridChunk, PM_SDF_ONLY, &pb);
    if (!c.ok) RNB_FAIL(RNB_E_WORKSPACE, "workspace too small: need %zu bytes, have %zu", c.off, ws_bytes);
    pb.M = m;
    pb.Mp = pad_rows(m);
    RNB_TRY(launch_grid_points(gg, first, m, pts, s));
    RNB_TRY(forward_points(L, packed, pts, m, pb, false, false, false, nullptr, s));
    RNB_TRY(launch_scale_copy(pb.sdf, gg.out_scale, m, volume + first, s));
  }
  return RNB_OK;
}

// ---------------------------------------------------------------------------------------------------
// hierarchical sampling
// ---------------------------------------------------------------------------------------------------
RNB_API int rnb_up_sample_step(const float* rays_o, const float* rays_d, const float* z_in, const float* sdf_in,
                               int64_t B, int32_t n, int32_t n_new, float inv_s, float* new_z, int32_t* inds,
                               float* z_out, int32_t* sort_index, rnb_stream_t stream) {
  RNB_REQUIRE(rays_o, "rays_o");
  RNB_REQUIRE(rays_d, "rays_d");
  RNB_REQUIRE(z_in, "z_in");
  RNB_REQUIRE(sdf_in, "sdf_in");
  RNB_REQUIRE(z_out, "z_out");
  if (B <= 0) return B == 0 ? RNB_OK : (set_error("B < 0"), RNB_E_INVALID);
  return launch_up_sample_step(rays_o, rays_d, z_in, sdf_in, nullptr, nullptr, n, B, n, n_new, inv_s, new_z, inds,
                               z_out, sort_index, nullptr, nullptr, (hipStream_t)stream);
}

RNB_API int rnb_gather_sdf(const float* sdf_old, const float* sdf_new, const int32_t* sort_index, int64_t B,
                           int32_t n, int32_t n_new, float* sdf_out, rnb_stream_t stream) {
  RNB_REQUIRE(sdf_old, "sdf_old");
  RNB_REQUIRE(sdf_new, "sdf_new");
  RNB_REQUIRE(sort_index, "sort_index");
  RNB_REQUIRE(sdf_out, "sdf_out");
  if (B <= 0) return B == 0 ? RNB_OK : (set_error("B < 0"), RNB_E_INVALID);
  return launch_gather_sdf(sdf_old, sdf_new, sort_index, B, n, n_new, sdf_out, (hipStream_t)stream);
}

struct SampleBufs {
  float* z[2];
  float* sdf[2];
  int32_t* index[2];   // sort index of a step (read by the next step's kernel while that one writes its own: ping-pong)
  float* pts;      // [B*n_samples,3] coarse points, later [B*n_new,3]
  PointBufs pb;    // sized for B*n_samples points
  size_t pb_off;   // carve offset of the point buffers
};

static void carve_sample(const Layout& L, const rnb_model_desc* d, Carver& c, int64_t B, SampleBufs* sb) {
  const int S = d->n_samples + d->n_importance;
  for (int i = 0; i < 2; ++i) sb->z[i] = c.take<float>(B * S);
  for (int i = 0; i < 2; ++i) sb->sdf[i] = c.take<float>(B * S);
  for (int i = 0; i < 2; ++i) sb->index[i] = c.take<int32_t>(B * S);
  sb->pts = c.take<float>(B * d->n_samples * 3);
  sb->pb_off = c.off;
  carve_points(L, c, B * d->n_samples, PM_SDF_ONLY, &sb->pb);
}

static int check_sampling_desc(const rnb_model_desc* d) {
  if (d->n_samples < 2) RNB_FAIL(RNB_E_INVALID, "n_samples must be >= 2");
  if (d->n_importance < 0) RNB_FAIL(RNB_E_INVALID, "n_importance < 0");
  if (d->n_importance > 0) {
    if (d->up_sample_steps < 1 || d->n_importance % d->up_sample_steps != 0)
      RNB_FAIL(RNB_E_INVALID, "n_importance (%d) must be a positive multiple of up_sample_steps (%d)",
               d->n_importance, d->up_sample_steps);
    if (d->n_importance / d->up_sample_steps > d->n_samples)
      RNB_FAIL(RNB_E_INVALID, "n_importance/up_sample_steps must not exceed n_samples");
  }
  return RNB_OK;
}

RNB_API int rnb_sample_workspace_bytes(const rnb_model_desc* desc, int64_t B, int64_t* bytes) {
  RNB_REQUIRE(bytes, "bytes");
  Layout L;
  RNB_TRY(make_layout(desc, &L));
  RNB_TRY(check_sampling_desc(desc));
  if (B < 0) RNB_FAIL(RNB_E_INVALID, "B < 0");
  Carver c(nullptr, 0);
  SampleBufs sb;
  carve_sample(L, desc, c, B, &sb);
  *bytes = (int64_t)c.off;
  return RNB_OK;
}

RNB_API int rnb_sample_rays(const rnb_model_desc* desc, const float* packed, const float* rays_o, const float* rays_d,
                            const float* near, const float* far, const float* t_rand, int64_t B, float* z_vals_out,
                            void* ws, size_t ws_bytes, rnb_stream_t stream) {
  RNB_REQUIRE(packed, "packed");
  RNB_REQUIRE(rays_o, "rays_o");
  RNB_REQUIRE(rays_d, "rays_d");
  RNB_REQUIRE(near, "near");
  RNB_REQUIRE(far, "far");
  RNB_REQUIRE(z_vals_out, "z_vals_out");
  if (B <= 0) return B == 0 ? RNB_OK : (set_error("B < 0"), RNB_E_INVALID);
  hipStream_t s = (hipStream_t)stream;
  Layout L;
  RNB_TRY(make_layout(desc, &L));
  RNB_TRY(check_sampling_desc(desc));
  const int n0 = desc->n_samples;
  if (desc->n_importance == 0) return launch_z_init(rays_o, rays_d, near, far, t_rand, B, n0, z_vals_out, nullptr, s);
  RNB_REQUIRE(ws, "workspace");
  Carver c(ws, ws_bytes);
  SampleBufs sb;
  carve_sample(L, desc, c, B, &sb);
  if (!c.ok) RNB_FAIL(RNB_E_WORKSPACE, "workspace too small: need %zu bytes, have %zu", c.off, ws_bytes);
  const int steps = desc->up_sample_steps;
  const int n_new = desc->n_importance / steps;

  RNB_TRY(launch_z_init(rays_o, rays_d, near, far, t_rand, B, n0, sb.z[0], sb.pts, s));
  RNB_TRY(forward_points(L, packed, sb.pts, B * n0, sb.pb, false, false, false, nullptr, s));
  // The SDF row travels through the loop inside the up-sampling kernel itself (renderer.py:185-190: sdf = cat[sdf,
  // new_sdf] gathered by the sort index): step i reads the row step i - 1 left sorted, the SDF of the points step i - 1
  // proposed and its sort index, and leaves the merged row for step i + 1 — no separate gather launches, no copy of the
  // coarse row out of the point buffers (which the next forward re-carves).
  const float* sdf_old = sb.pb.sdf;    // step 0: the coarse row, as the forward wrote it
  const float* sdf_new = nullptr;
  const int32_t* gidx = nullptr;
  int n_old = n0;
  int cur = 0;
  int n = n0;
  for (int i = 0; i < steps; ++i) {
    const bool last = (i + 1 == steps);
    float* z_next = last ? z_vals_out : sb.z[cur ^ 1];
    RNB_TRY(launch_up_sample_step(rays_o, rays_d, sb.z[cur], sdf_old, sdf_new, gidx, n_old, B, n, n_new,
                                  (float)(64 << i), nullptr, nullptr, z_next, last ? nullptr : sb.index[i & 1],
                                  last ? nullptr : sb.pts, last ? nullptr : sb.sdf[cur ^ 1], s));
    if (!last) {
      // SDF of the new points (renderer.py:185)
      Carver c2((char*)ws + sb.pb_off, ws_bytes - sb.pb_off);
      PointBufs pbn;
      carve_points(L, c2, B * n_new, PM_SDF_ONLY, &pbn);
      RNB_TRY(forward_points(L, packed, sb.pts, B * n_new, pbn, false, false, false, nullptr, s));
      sdf_old = sb.sdf[cur ^ 1];       // this step's input row in sorted order (n entries)
      sdf_new = pbn.sdf;
      gidx = sb.index[i & 1];
      n_old = n;
    }
    cur ^= 1;
    n += n_new;
  }
  return RNB_OK;
}

// ---------------------------------------------------------------------------------------------------
// fine pass
// ---------------------------------------------------------------------------------------------------
struct RenderBufs {
  float* pts;
  float* dists;
  float* gerr_part;
  float* gerr_den;
  float* invs_part;
  PointBufs pb;
};

static int render_mode_of(int flags, const Layout& L) {
  int mode = PM_WITH_NORMAL;
  const bool use_color = !((flags & RNB_MODE_MVPS) && (flags & RNB_FLAG_NO_ALBEDO));
  if (use_color) mode |= PM_WITH_COLOR;
  if (!(flags & RNB_FLAG_FORWARD_ONLY)) mode |= PM_WITH_BACKWARD;
  (void)L;
  return mode;
}

static void carve_render(const Layout& L, Carver& c, int64_t B, int S, int flags, RenderBufs* rb) {
  rb->pts = c.take<float>(B * S * 3);
  rb->dists = c.take<float>(B * S);
  rb->gerr_part = c.take<float>(B * 2);
  rb->gerr_den = c.take<float>(1);
  rb->invs_part = c.take<float>(B);
  carve_points(L, c, B * S, render_mode_of(flags, L), &rb->pb);
  if (!(render_mode_of(flags, L) & PM_WITH_COLOR)) rb->pb.alb = c.take<float>(rb->pb.Mp * 4);
}

RNB_API int rnb_render_workspace_bytes(const rnb_model_desc* desc, int64_t B, int32_t S, int32_t flags,
                                       int64_t* bytes) {
  RNB_REQUIRE(bytes, "bytes");
  Layout L;
  RNB_TRY(make_layout(desc, &L));
  if (B < 0 || S < 1) RNB_FAIL(RNB_E_INVALID, "bad B/S");
  Carver c(nullptr, 0);
  RenderBufs rb;
  carve_render(L, c, B, S, flags, &rb);
  *bytes = (int64_t)c.off;
  return RNB_OK;
}

static int render_setup(const rnb_model_desc* desc, const rnb_render_args* a, void* ws, size_t ws_bytes, Layout* L,
                        RenderBufs* rb) {
  RNB_TRY(make_layout(desc, L));
  RNB_REQUIRE(a, "args");
  RNB_REQUIRE(ws, "workspace");
  if (a->B <= 0 || a->S < 1) RNB_FAIL(RNB_E_INVALID, "bad B/S");
  const bool mvps = (a->flags & RNB_MODE_MVPS) != 0;
  if (mvps && (a->n_lights < 1 || !a->lights_dir)) RNB_FAIL(RNB_E_INVALID, "MVPS mode needs lights");
  if (L->F <= 0) RNB_FAIL(RNB_E_INVALID, "model has no feature head");
  RNB_REQUIRE(a->rays_o, "rays_o");
  RNB_REQUIRE(a->rays_d, "rays_d");
  RNB_REQUIRE(a->z_vals, "z_vals");
  RNB_REQUIRE(a->variance, "variance");
  Carver c(ws, ws_bytes);
  carve_render(*L, c, a->B, a->S, a->flags, rb);
  if (!c.ok) RNB_FAIL(RNB_E_WORKSPACE, "workspace too small: need %zu bytes, have %zu", c.off, ws_bytes);
  return RNB_OK;
}

static CompArgs comp_args_of(const Layout& L, const rnb_render_args* a, const RenderBufs& rb) {
  CompArgs c;
  memset(&c, 0, sizeof(c));
  c.B = a->B;
  c.S = a->S;
  c.L = (a->flags & RNB_MODE_MVPS) ? a->n_lights : 1;
  c.C = L.Co;
  c.flags = a->flags;
  c.cos_anneal = a->cos_anneal_ratio;
  c.rays_d = a->rays_d;
  c.pts = rb.pts;
  c.dists = rb.dists;
  c.sdf = rb.pb.sdf;
  c.nrm = rb.pb.nrm;
  c.alb = rb.pb.alb;
  c.lights = a->lights_dir;
  c.bg = (a->flags & RNB_MODE_MVPS) ? nullptr : a->background_rgb;
  c.variance = a->variance;
  c.color_fine = a->color_fine;
  c.weights = a->weights;
  c.cdf = a->cdf_fine;
  c.gradients = a->gradients;
  c.inside = a->inside_sphere;
  c.weight_sum = a->weight_sum;
  c.weight_max = a->weight_max;
  c.s_val = a->s_val;
  c.gerr_part = rb.gerr_part;
  c.sdf_out = a->sdf;
  c.albedo_out = (render_mode_of(a->flags, L) & PM_WITH_COLOR) ? a->sampled_albedo : nullptr;
  return c;
}

RNB_API int rnb_render_fwd(const rnb_model_desc* desc, const float* packed, const rnb_render_args* a, void* ws,
                           size_t ws_bytes, rnb_stream_t stream) {
  RNB_REQUIRE(packed, "packed");
  hipStream_t s = (hipStream_t)stream;
  Layout L;
  RenderBufs rb;
  RNB_TRY(render_setup(desc, a, ws, ws_bytes, &L, &rb));
  RNB_REQUIRE(a->color_fine, "color_fine");
  RNB_REQUIRE(a->weights, "weights");
  RNB_REQUIRE(a->cdf_fine, "cdf_fine");
  RNB_REQUIRE(a->gradients, "gradients");
  RNB_REQUIRE(a->inside_sphere, "inside_sphere");
  RNB_REQUIRE(a->weight_sum, "weight_sum");
  RNB_REQUIRE(a->weight_max, "weight_max");
  RNB_REQUIRE(a->s_val, "s_val");
  RNB_REQUIRE(a->gradient_error, "gradient_error");
  const int mode = render_mode_of(a->flags, L);
  const bool use_color = (mode & PM_WITH_COLOR) != 0;
  RNB_TRY(launch_fine_points(a->rays_o, a->rays_d, a->z_vals, a->B, a->S, 2.0f / (float)desc->n_samples, rb.pts,
                             rb.dists, rb.pb.smax, s));
  const bool color_bf16 = is_bf16(L) && use_color && bf16_color_supported(L);
  if (color_bf16) {   // the feature head writes bf16 K8 straight into the albedo net's input
    RNB_TRY(bf16_forward(L, packed, rb.pts, a->B * a->S, rb.pb, true, true, s, nullptr, true));
  } else {
    RNB_TRY(forward_points(L, packed, rb.pts, a->B * a->S, rb.pb, true, use_color, !use_fused(L) && !is_bf16(L), nullptr, s));
  }
  RNB_TRY(reverse_points(L, packed, rb.pb, s));
  if (color_bf16) RNB_TRY(bf16_color_forward(L, packed, rb.pb, rb.pts, s));
  else if (use_color && use_fused(L) && color_h2_supported(L)) RNB_TRY(color_h2_forward(L, packed, rb.pb, rb.pts, rb.pb.nrm, s));
  else if (use_color) RNB_TRY(sweep_color(L, packed, rb.pb, rb.pts, rb.pb.nrm, 4, s));
  CompArgs c = comp_args_of(L, a, rb);
  c.gerr = a->gradient_error;
  c.gerr_den = rb.gerr_den;
  c.gerr_partial = a->gerr_partial;
  RNB_TRY(launch_composite_fwd(c, s));
  return RNB_OK;
}

RNB_API int rnb_render_bwd(const rnb_model_desc* desc, const float* packed, const rnb_render_args* a,
                           const rnb_render_grads* gout, float* packed_grad, float* variance_grad, void* ws,
                           size_t ws_bytes, rnb_stream_t stream) {
  RNB_REQUIRE(packed, "packed");
  RNB_REQUIRE(gout, "gout");
  RNB_REQUIRE(packed_grad, "packed_grad");
  RNB_REQUIRE(variance_grad, "variance_grad");
  hipStream_t s = (hipStream_t)stream;
  Layout L;
  RenderBufs rb;
  RNB_TRY(render_setup(desc, a, ws, ws_bytes, &L, &rb));
  if (a->flags & RNB_FLAG_FORWARD_ONLY) RNB_FAIL(RNB_E_INVALID, "forward-only render has no backward state");
  const int mode = render_mode_of(a->flags, L);
  const bool use_color = (mode & PM_WITH_COLOR) != 0;
  CompBwdArgs g;
  memset(&g, 0, sizeof(g));
  g.f = comp_args_of(L, a, rb);
  g.weights = a->weights;
  g.g_color = gout->color_fine;
  g.g_weights = gout->weights;
  g.g_cdf = gout->cdf_fine;
  g.g_gradients = gout->gradients;
  g.g_weight_sum = gout->weight_sum;
  g.g_weight_max = gout->weight_max;
  g.g_s_val = gout->s_val;
  g.g_gerr = gout->gradient_error;
  g.gerr_den = rb.gerr_den;
  g.gerr_den_global = a->gerr_den_global;
  g.sbar = rb.pb.sbar;
  g.nbar = rb.pb.nbar;
  g.albbar = rb.pb.albbar;
  g.invs_part = rb.invs_part;
  g.dvar = variance_grad;
  g.amax_to_zero = rb.pb.amax;
  RNB_TRY(launch_composite_bwd(g, s));
  RNB_CHECK_HIP(hipMemsetAsync(packed_grad, 0, (size_t)L.total * sizeof(float), s));
  RNB_TRY(sweep_backward(L, packed, rb.pb, use_color, packed_grad, use_fused(L), s));
  return RNB_OK;
}

RNB_API int rnb_render_range(const rnb_model_desc* desc, const float* packed, const void* ws, size_t ws_bytes, int64_t B, int32_t S,
                             int32_t flags, float* out, rnb_stream_t stream) {
  RNB_REQUIRE(packed, "packed");
  RNB_REQUIRE(ws, "workspace");
  RNB_REQUIRE(out, "out");
  if (B <= 0 || S < 1) RNB_FAIL(RNB_E_INVALID, "bad B/S");
  hipStream_t s = (hipStream_t)stream;
  Layout L;
  RNB_TRY(make_layout(desc, &L));
  Carver c(const_cast<void*>(ws), ws_bytes);
  RenderBufs rb;
  carve_render(L, c, B, S, flags, &rb);
  if (!c.ok) RNB_FAIL(RNB_E_WORKSPACE, "workspace too small: need %zu bytes, have %zu", c.off, ws_bytes);
  return launch_range_report(L, packed, rb.pb, (render_mode_of(flags, L) & PM_WITH_COLOR) != 0,
                             !(flags & RNB_FLAG_FORWARD_ONLY), out, s);
}

RNB_API int rnb_profile_enable(int on) { return profile_enable(on); }
RNB_API int rnb_profile_collect(double* gemm_ms, int64_t* gemm_launches, double* gemm_flops) {
  return profile_collect(gemm_ms, gemm_launches, gemm_flops);
}
RNB_API int64_t rnb_profile_report(char* out, int64_t capacity) { return profile_report(out, capacity); }

RNB_API int rnb_algorithmic_bytes(const rnb_model_desc* desc, int64_t B, int32_t flags, double* train_bytes) {
  RNB_REQUIRE(train_bytes, "train_bytes");
  Layout L;
  RNB_TRY(make_layout(desc, &L));
  const bool use_color = !((flags & RNB_MODE_MVPS) && (flags & RNB_FLAG_NO_ALBEDO));
  const double e = is_bf16(L) ? 2.0 : 4.0;
  const int S = desc->n_samples + desc->n_importance;
  // SDF sweeps: a, D, gz, u, zR, zb written once; D read by R, RA, FB; gz by RA, dW; zR by FB; u, zb, a by dW
  double per_pt = 15.0 * L.nh * L.Hp * e;
  per_pt += 2.0 * L.Ep * e * 2.0;   // e and u_0: written once, read once (layer 0's weight gradient)
  if (use_color) {
    // albedo network: cin (1 write, 2 reads), ac_l (1 write; read by the next layer, the relu mask and a dW), zc_l
    // (1 write, 2 reads), cinb (1 write; read by FB and by the normal's adjoint).  bf16 too when the bf16 albedo
    // kernels apply (RNB_VARIANT_BF16 with the shipped shape), else fp32.
    const double ec = (is_bf16(L) && bf16_color_supported(L)) ? 2.0 : 4.0;
    per_pt += ec * (3.0 * L.Cinp + 4.0 * L.Hcp + 3.0 * L.Hcp * (L.nc - 1) + 3.0 * L.Hcp * L.nc + 2.0 * L.Cinp);
    per_pt += e * 2.0 * L.Hp;         // feature head's weight gradient: fbar and a_last
  }
  *train_bytes = per_pt * (double)B * S;
  return RNB_OK;
}

// Algorithmic MLP FLOPs: multiply-accumulate counts of the real (unpadded) layer shapes.
RNB_API int rnb_algorithmic_flops(const rnb_model_desc* desc, int64_t B, int32_t flags, double* train_flops,
                                  double* forward_flops) {
  Layout L;
  RNB_TRY(make_layout(desc, &L));
  double mac_s = 0;   // one SDF forward
  for (int l = 0; l < L.nh; ++l) mac_s += (double)L.hid[l].N * L.hid[l].K;
  mac_s += (double)(L.F + 1) * L.H;
  double mac_c = 0;
  for (int l = 0; l < L.nc; ++l) mac_c += (double)L.col[l].N * L.col[l].K;
  mac_c += (double)L.colo.N * L.colo.K;
  const bool use_color = !((flags & RNB_MODE_MVPS) && (flags & RNB_FLAG_NO_ALBEDO));
  const double Fs = 2.0 * mac_s, Fc = 2.0 * mac_c;
  const int S = desc->n_samples + desc->n_importance;
  const int n_new = desc->n_importance > 0 ? desc->n_importance / desc->up_sample_steps : 0;
  const double coarse_pts = desc->n_importance > 0 ? desc->n_samples + (double)n_new * (desc->up_sample_steps - 1) : 0;
  const double per_ray_train = coarse_pts * Fs + S * 6.0 * Fs + (use_color ? S * 3.0 * Fc : 0.0);
  const double per_ray_fwd = coarse_pts * Fs + S * 2.0 * Fs + (use_color ? S * Fc : 0.0);
  if (train_flops) *train_flops = per_ray_train * (double)B;
  if (forward_flops) *forward_flops = per_ray_fwd * (double)B;
  return RNB_OK;
}
