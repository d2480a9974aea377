// Host-side layout of the packed weight buffer and of the per-point workspaces.
#include <stdarg.h>

#include "rnb_internal.h"

namespace rnb {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* last_error() { return g_err; }

static void place(Lin& l, int N, int K, float scale, int64_t& off) {
  l.N = N;
  l.K = K;
  l.Np = pad32(N);
  l.Kp = pad32(K);
  l.scale = scale;
  l.w_off = off;
  off += (int64_t)l.Np * l.Kp;
  l.b_off = off;
  off += l.Np;
  l.wT_off = -1;
}
// transposed copy W^T [Kp x Np]: lets the reverse-shaped sweeps (R, FB) stream weight rows exactly like
// the forward-shaped ones
static void place_transpose(Lin& l, int64_t& off) {
  l.wT_off = off;
  off += (int64_t)l.Kp * l.Np;
}

int make_layout(const rnb_model_desc* d, Layout* L) {
  if (!d || !L) RNB_FAIL(RNB_E_NULL, "null desc");
  memset(L, 0, sizeof(*L));
  if (d->sdf_d_in != 3) RNB_FAIL(RNB_E_INVALID, "sdf_d_in must be 3 (got %d)", d->sdf_d_in);
  if (d->sdf_n_layers < 1 || d->sdf_n_layers + 1 > RNB_MAX_LIN)
    RNB_FAIL(RNB_E_INVALID, "sdf_n_layers out of range (%d)", d->sdf_n_layers);
  if (d->sdf_d_hidden < 1 || d->sdf_d_out < 1) RNB_FAIL(RNB_E_INVALID, "bad sdf widths");
  if (d->sdf_multires < 0 || d->sdf_multires > 16) RNB_FAIL(RNB_E_INVALID, "bad sdf_multires");
  if (!(d->sdf_scale > 0.f)) RNB_FAIL(RNB_E_INVALID, "sdf_scale must be positive");
  L->variant = d->variant;
  if (d->variant & ~0xFFF7F) RNB_FAIL(RNB_E_INVALID, "unknown bits in rnb_model_desc.variant (0x%x)", d->variant);
  L->nh = d->sdf_n_layers;
  L->multires = d->sdf_multires;
  L->pe = 3 * (1 + 2 * d->sdf_multires);
  L->Ep = pad32(L->pe);
  L->H = d->sdf_d_hidden;
  L->Hp = pad32(L->H);
  L->skip = d->sdf_skip_in;
  L->sdf_scale = d->sdf_scale;
  if (L->skip >= 0 && (L->skip < 1 || L->skip > L->nh - 1))
    RNB_FAIL(RNB_E_INVALID, "sdf_skip_in must be in [1, n_layers-1] or -1 (got %d)", L->skip);
  if (L->skip >= 0 && L->H - L->pe < 1) RNB_FAIL(RNB_E_INVALID, "d_hidden too small for the skip connection");
  int64_t off = 0;
  for (int l = 0; l < L->nh; ++l) {
    const int K = l == 0 ? L->pe : L->H;
    const int N = (l + 1 == L->skip) ? L->H - L->pe : L->H;
    // every hidden layer is stored Hp rows tall: the layer feeding the skip connection writes the
    // positional encoding into columns [N, N+pe) of its output buffer from its GEMM epilogue
    L->hid[l].N = N;
    L->hid[l].K = K;
    L->hid[l].Np = L->Hp;
    L->hid[l].Kp = pad32(K);
    L->hid[l].scale = l == L->skip ? 0.70710678118654752440f : 1.f;
    L->hid[l].w_off = off;
    off += (int64_t)L->Hp * L->hid[l].Kp;
    L->hid[l].b_off = off;
    off += L->Hp;
    place_transpose(L->hid[l], off);
  }
  L->F = d->sdf_d_out - 1;
  L->Fp = pad32(L->F > 0 ? L->F : 1);
  place(L->feat, L->F > 0 ? L->F : 0, L->H, 1.f, off);
  if (L->F <= 0) { L->feat.Np = 0; }
  else place_transpose(L->feat, off);
  L->wsdf_off = off;
  off += L->Hp;
  L->bsdf_off = off;
  off += kPad;
  // albedo network (mode no_view_dir: input = [pe(p), pe(n), feature])
  L->nc = d->col_n_layers;
  L->multires_view = d->col_multires_view;
  if (L->F > 0) {
    if (d->col_n_layers < 1 || d->col_n_layers + 1 > RNB_MAX_LIN) RNB_FAIL(RNB_E_INVALID, "bad col_n_layers");
    if (d->col_d_feature != L->F) RNB_FAIL(RNB_E_INVALID, "col_d_feature (%d) != sdf_d_out-1 (%d)", d->col_d_feature, L->F);
    if (d->col_d_in != 6) RNB_FAIL(RNB_E_INVALID, "col_d_in must be 6 in no_view_dir mode");
    if (d->col_d_out < 1 || d->col_d_out > 4) RNB_FAIL(RNB_E_INVALID, "col_d_out must be 1..4");
    L->pev = 3 * (1 + 2 * d->col_multires_view);
    L->Cin = L->F + 2 * L->pev;
    L->Cinp = pad32(L->Cin);
    L->Hc = d->col_d_hidden;
    L->Hcp = pad32(L->Hc);
    L->Co = d->col_d_out;
    L->Cop = kPad;
    L->squeeze = d->col_squeeze_out;
    for (int l = 0; l < L->nc; ++l) {
      place(L->col[l], L->Hc, l == 0 ? L->Cin : L->Hc, 1.f, off);
      place_transpose(L->col[l], off);   // the albedo net's backward products also run as k-contiguous GEMMs
    }
    place(L->colo, L->Co, L->Hc, 1.f, off);
  }
  L->total = (off + 31) / 32 * 32;
  L->total_all = L->total;
  if (d->variant & RNB_VARIANT_BF16) {
    if (!fused_supported(*L))
      RNB_FAIL(RNB_E_INVALID, "RNB_VARIANT_BF16 needs the 256-wide SDF network shape (d_hidden 256, feature width <= 256)");
    if (d->variant & RNB_VARIANT_GENERIC) RNB_FAIL(RNB_E_INVALID, "RNB_VARIANT_BF16 and RNB_VARIANT_GENERIC exclude each other");
    L->total_all = L->total + L->total / 2;   // bf16 mirror of the weights behind the fp32 ones
  }
  if (d->variant & RNB_VARIANT_X3) {
    if (d->variant & (RNB_VARIANT_BF16 | RNB_VARIANT_GENERIC | RNB_VARIANT_F32_MFMA))
      RNB_FAIL(RNB_E_INVALID, "RNB_VARIANT_X3 excludes RNB_VARIANT_BF16, RNB_VARIANT_GENERIC and RNB_VARIANT_F32_MFMA");
    if (!fused_supported(*L)) RNB_FAIL(RNB_E_INVALID, "RNB_VARIANT_X3 needs the 256-wide SDF network shape");
  }
  // the fused fp32 path multiplies on the bf16 matrix pipe (x3) unless the native fp32 MFMA is asked for
  if (!(d->variant & (RNB_VARIANT_BF16 | RNB_VARIANT_GENERIC | RNB_VARIANT_F32_MFMA)) && fused_supported(*L))
    L->variant |= RNB_VARIANT_X3;
  if (L->variant & RNB_VARIANT_X3)
    L->total_all = L->total + L->total / 2 * 3;   // hi / mid / lo bf16 mirror of the weights behind the fp32 ones
  if ((d->variant & RNB_VARIANT_REG_TILE) && (d->variant & RNB_VARIANT_LDS_TILE))
    RNB_FAIL(RNB_E_INVALID, "RNB_VARIANT_REG_TILE and RNB_VARIANT_LDS_TILE exclude each other");
  if ((d->variant & RNB_VARIANT_X2H) && (d->variant & RNB_VARIANT_NO_X2H))
    RNB_FAIL(RNB_E_INVALID, "RNB_VARIANT_X2H and RNB_VARIANT_NO_X2H exclude each other");
  if ((d->variant & RNB_VARIANT_X2H) && !(L->variant & RNB_VARIANT_X3))
    RNB_FAIL(RNB_E_INVALID, "RNB_VARIANT_X2H is a form of the x3 path (256-wide SDF network, fp32)");
  // forward-type sweeps of the x3 path: three fp16 terms unless switched off
  if ((L->variant & RNB_VARIANT_X3) && !(d->variant & RNB_VARIANT_NO_X2H)) L->variant |= RNB_VARIANT_X2H;
  L->h2tab_off = -1;
  if (L->variant & RNB_VARIANT_X2H) {
    L->total_all += L->total;   // + hi / lo fp16 mirror (W and W^T of the SDF network used)
    L->h2tab_off = L->total_all;   // + the mirror's scale table (H2Tab)
    L->total_all += 256;
    if (L->nh + 1 + L->nc > kH2TabSlots) RNB_FAIL(RNB_E_INVALID, "too many layers for the x2h scale table");
  }
  return RNB_OK;
}

void carve_points(const Layout& L, Carver& c, int64_t M, int mode, PointBufs* pb) {
  memset(pb, 0, sizeof(*pb));
  pb->M = M;
  pb->Mp = pad_rows(M);
  const int64_t Mp = pb->Mp;
  const bool bf = (L.variant & RNB_VARIANT_BF16) != 0;
  // RNB_VARIANT_BF16: the per-point state of the SDF sweeps is bf16 (K8 layout, bf16.hip): half the bytes
  auto take_state = [&](int64_t n) { return bf ? reinterpret_cast<float*>(c.take<uint16_t>(n)) : c.take<float>(n); };
  pb->x = c.take<float>(Mp * 4);
  pb->e = c.take<float>(Mp * L.Ep);
  for (int l = 0; l < L.nh; ++l) pb->a[l] = take_state(Mp * L.Hp);
  pb->sdf = c.take<float>(Mp);
  pb->smax = c.take<unsigned>(SMAX_SLOTS);
  if (mode & (PM_WITH_NORMAL | PM_WITH_COLOR | PM_WITH_BACKWARD)) {
    for (int l = 0; l < L.nh; ++l) pb->gz[l] = take_state(Mp * L.Hp);
    for (int l = 0; l < L.nh; ++l) pb->D[l] = take_state(Mp * L.Hp);
    pb->ge = c.take<float>(Mp * L.Ep);
    pb->nrm = c.take<float>(Mp * 4);
  }
  if (mode & PM_WITH_COLOR) {
    pb->cin = c.take<float>(Mp * L.Cinp);
    for (int l = 0; l < L.nc; ++l) pb->ac[l] = c.take<float>(Mp * L.Hcp);
    pb->alb = c.take<float>(Mp * 4);
    if (color_h2_supported(L)) pb->ac0_mask = c.take<unsigned>(Mp / 64 * 256 * 2);
    if (bf && bf16_color_supported(L)) {
      pb->cin8 = c.take<uint16_t>(Mp * L.Cinp);
      for (int l = 0; l < L.nc; ++l) pb->ac8[l] = c.take<uint16_t>(Mp * L.Hcp);
    }
  }
  if (mode & PM_WITH_BACKWARD) {
    for (int l = 1; l <= L.nh; ++l) pb->u[l] = take_state(Mp * L.Hp);
    for (int l = 0; l < L.nh; ++l) pb->zR[l] = take_state(Mp * L.Hp);
    for (int l = 0; l < L.nh; ++l) pb->zb[l] = take_state(Mp * L.Hp);
    if (bf) {
      pb->u0_k8 = c.take<uint16_t>(Mp * L.Ep);
      pb->fbar_k8 = c.take<uint16_t>(Mp * L.Hp);
    }
    pb->geb = c.take<float>(Mp * L.Ep);
    pb->amax = c.take<unsigned>(AMAX_SLOTS);
    pb->sbar = c.take<float>(Mp);
    pb->nbar = c.take<float>(Mp * 4);
    pb->albbar = c.take<float>(Mp * 4);
    if (mode & PM_WITH_COLOR) {
      for (int l = 0; l < L.nc; ++l) pb->zc[l] = c.take<float>(Mp * L.Hcp);
      pb->cinb = c.take<float>(Mp * L.Cinp);
      if (bf && bf16_color_supported(L))
        for (int l = 0; l < L.nc; ++l) pb->zc8[l] = c.take<uint16_t>(Mp * L.Hcp);
    }
    {
      const bool wc = (mode & PM_WITH_COLOR) != 0;
      pb->dw_part_floats = dw_staged_floats(L, M, wc);
      if (L.variant & RNB_VARIANT_DETERMINISTIC)
        pb->dw_part_floats += dw_partial_floats(L, M, wc) + (bf ? bf16_dw_partial_floats(L, M, wc) : 0);
      pb->dw_part = c.take<float>(pb->dw_part_floats > 0 ? pb->dw_part_floats : 64);
      if (wc && color_h2_supported(L)) pb->col_part = c.take<float>(color_h2_part_floats(L, M));
      pb->sdfh_part = c.take<float>((int64_t)kSdfHeadSlabs * (L.Hp + 1));
    }
  }
}

}  // namespace rnb
