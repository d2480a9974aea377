// Internal declarations shared by the translation units of librnbneus_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/rnbneus.h"

namespace rnb {

constexpr int kPad = 32;          // every feature width is padded to the MFMA tile width
constexpr int kRowPad = 128;      // point counts are padded to the GEMM block height
inline int pad32(int x) { return (x + kPad - 1) / kPad * kPad; }
inline int64_t pad_rows(int64_t m) { return (m + kRowPad - 1) / kRowPad * kRowPad; }

void set_error(const char* fmt, ...);
#define RNB_FAIL(code, ...)        \
  do {                             \
    ::rnb::set_error(__VA_ARGS__); \
    return (code);                 \
  } while (0)
#define RNB_CHECK_HIP(expr)                                                          \
  do {                                                                               \
    hipError_t e_ = (expr);                                                          \
    if (e_ != hipSuccess) RNB_FAIL(RNB_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)
#define RNB_CHECK_LAUNCH() RNB_CHECK_HIP(hipGetLastError())
#define RNB_TRY(expr)          \
  do {                         \
    int rc_ = (expr);          \
    if (rc_ != RNB_OK) return rc_; \
  } while (0)

// One linear layer inside the packed (effective-weight) buffer: W [Np x Kp] row-major, then b [Np].
struct Lin {
  int N, K;        // real out / in widths
  int Np, Kp;      // padded
  int64_t w_off;   // float offset of W in the packed buffer
  int64_t wT_off;  // float offset of the transposed copy W^T [Kp x Np] (reverse-shaped sweeps), or -1
  int64_t b_off;   // float offset of b
  float scale;     // factor folded into W (1/sqrt2 for the skip layer)
};

// Packed layout of both networks (see weightnorm.hip for how leaves map onto it).
struct Layout {
  // SDF network: nh hidden layers (softplus) + output layer split into sdf row and feature rows
  int nh;                 // number of hidden (softplus) layers = sdf_n_layers
  int pe;                 // positional-encoding width (39)
  int Ep;                 // padded pe width
  int H, Hp;              // hidden width, padded
  int skip;               // skip layer index or -1
  Lin hid[RNB_MAX_LIN];   // hidden layers 0..nh-1
  int F, Fp;              // feature width (d_out-1), padded
  Lin feat;               // rows 1.. of the last layer (feature head)
  int64_t wsdf_off;       // row 0 of the last layer [Hp]
  int64_t bsdf_off;       // its bias [1] (padded to 32)
  float sdf_scale;
  int multires;
  // albedo network: nc hidden (relu) layers + output layer (d_out rows, sigmoid)
  int nc;
  int pev;                // pe width of multires_view (27)
  int Cin, Cinp;          // input width (F + 2*pev), padded
  int Hc, Hcp;
  int Co, Cop;            // d_out, padded to 32
  Lin col[RNB_MAX_LIN];   // hidden layers 0..nc-1
  Lin colo;               // output layer
  int multires_view;
  int squeeze;
  int64_t total;          // floats of fp32 packed weights (also the length of a gradient buffer)
  int64_t total_all;      // floats of the packed buffer = total (+ total / 2 for the bf16 mirror, RNB_VARIANT_BF16)
  int64_t h2tab_off;      // RNB_VARIANT_X2H: float offset of the scale table of the fp16 mirror (H2Tab), else -1
  int variant;            // rnb_model_desc.variant (RNB_VARIANT_* bits)
  int knob(int shift) const { return (variant >> shift) & 3; }
};

int make_layout(const rnb_model_desc* d, Layout* L);

#if defined(__HIPCC__)
// torch.clamp / torch.minimum / torch.max / F.relu PROPAGATE NaN; fmaxf / fminf (v_max_f32 / v_min_f32) return the
// other operand.  A diverged model must come back as NaN exactly where the reference's does, so the element-wise
// min / max of the path go through these.  Identical to fminf / fmaxf on numbers.
__device__ inline float max_nan(float a, float b) { return a != a ? a : (b != b ? b : fmaxf(a, b)); }
__device__ inline float min_nan(float a, float b) { return a != a ? a : (b != b ? b : fminf(a, b)); }
__device__ inline float clamp_nan(float x, float lo, float hi) { return x != x ? x : fminf(fmaxf(x, lo), hi); }
__device__ inline float relu_nan(float x) { return x < 0.f ? 0.f : x; }

// torch.linspace(start, end, steps)[i] as ATen's CPU kernel computes it: step = (end-start)/(steps-1); first half
// start + step*i, second half end - step*(steps-1-i), each as one fused multiply-add (explicit fmaf: independent of
// the translation unit's contraction mode).
__device__ inline float linspace_at(float start, float end, int steps, int i) {
  if (steps == 1) return start;
  const float step = (end - start) / (float)(steps - 1);
  return i < steps / 2 ? fmaf(step, (float)i, start) : fmaf(-step, (float)(steps - 1 - i), end);
}
#endif

// regular grid of extract_fields, generated inside the forward kernel (rnb_sdf_grid)
struct GridGen {
  int on;
  int res, x_begin;
  float bmin[3], bmax[3];
  float out_scale;
};

// ---- workspace carving ------------------------------------------------------------------------
struct Carver {
  char* base;
  size_t cap;
  size_t off = 0;
  bool ok = true;
  Carver(void* p, size_t bytes) : base((char*)p), cap(bytes) {}
  template <class T>
  T* take(int64_t n) {
    size_t bytes = ((size_t)n * sizeof(T) + 255) / 256 * 256;
    T* p = (T*)(base ? base + off : nullptr);
    off += bytes;
    if (base && off > cap) ok = false;
    return p;
  }
};

// ---- RNB_VARIANT_X2H: scales of the fp16 mirror -----------------------------------------------------------------------
// Every matrix of the fp16 mirror is stored times a power of two chosen from the matrix's own maximum: 2^8 while
// max |w| < 64 (the round-4 constant), else the power of two that puts the maximum in [2^13, 2^14) — so NO weight is out of
// range.  The table sits behind the mirror in the packed buffer (256 floats): per matrix id the float bits of max |w|
// (atomicMax by x3_pack_kernel, zeroed by wn_fwd_kernel), the scale and its inverse (written by x2h_pack_kernel, read by
// every kernel that multiplies with the mirror).  ids: hidden layer l -> l, feature head -> nh, albedo hidden layer l ->
// nh + 1 + l; W and W^T share an id.
constexpr int kH2TabSlots = 48;
struct H2Tab {
  unsigned wmax[kH2TabSlots];
  float ws[kH2TabSlots];
  float iws[kH2TabSlots];
};
static_assert(sizeof(H2Tab) <= 256 * sizeof(float), "the packed buffer reserves 256 floats for the table");
inline int h2_id_hid(int l) { return l; }

// State of one batch of points going through the SDF (+albedo) network(s).  All activation matrices
// are [Mp x width_padded] row-major fp32.
struct PointBufs {
  int64_t M, Mp;
  float* x;       // [Mp,4]   scaled points (x,y,z,0)
  float* e;       // [Mp,Ep]  positional encoding
  float* a[RNB_MAX_LIN];   // hidden activations a_l  [Mp,Hp]
  float* D[RNB_MAX_LIN];   // softplus'(z_l) = sigmoid(100 z_l)  [Mp,Hp]   (only with_normal; else nullptr)
  float* gz[RNB_MAX_LIN];  // reverse sweep state gz_l  [Mp,Hp]   (only with_normal)
  float* ge;      // [Mp,Ep]  d sdf / d e
  float* sdf;     // [Mp]
  float* nrm;     // [Mp,4]   d sdf / d x
  float* cin;     // [Mp,Cinp] albedo-net input  [feat | pe(p) | pe(n) | 0]
  float* ac[RNB_MAX_LIN];  // albedo hidden activations [Mp,Hcp]
  float* alb;     // [Mp,4]   albedo (network output)
  // backward-only
  float* u[RNB_MAX_LIN];   // RA sweep inputs u_l  (u[0] is [Mp,Ep])
  float* zR[RNB_MAX_LIN];  // second-order term entering layer l's pre-activation adjoint
  float* zb[RNB_MAX_LIN];  // pre-activation adjoints of F
  float* geb;     // [Mp,Ep]  adjoint of ge
  float* zc[RNB_MAX_LIN];  // albedo-net pre-activation adjoints
  float* cinb;    // [Mp,Cinp] adjoint of cin
  float* sbar;    // [Mp]
  float* nbar;    // [Mp,4]
  float* albbar;  // [Mp,4]
  void* u0_k8;              // RNB_VARIANT_BF16: u_0 = J_pe nbar as bf16 K8 [Mp,Ep]   (written by the RA sweep)
  void* cin8;               // RNB_VARIANT_BF16, bf16 albedo path: albedo-net input as bf16 K8 [Mp,Cinp]
  void* ac8[RNB_MAX_LIN];   //   hidden activations as bf16 K8 [Mp,256]
  void* zc8[RNB_MAX_LIN];   //   pre-activation adjoints as bf16 K8 [Mp,256]
  void* fbar_k8;            // RNB_VARIANT_BF16: feature part of cinb as bf16 K8 [Mp,256] (written by the FB sweep)
  unsigned* amax;           // [AMAX_SLOTS] max |.| of the adjoint tensors (float bits; zeroed at the start of a backward)
  unsigned* smax;           // [SMAX_SLOTS] max |.| of the saved forward state the x2h weight-gradient jobs take as operands
                            // (float bits; zeroed by the first kernel of a render forward, grown by the forward sweeps)
  float* dw_part;           // partial slabs of the split-K weight-gradient GEMMs: [deterministic variant | staged kernel]
  int64_t dw_part_floats;
  unsigned* ac0_mask;       // fused albedo kernels: relu'(ac_0) as bits [tiles][256][2] (color_h2.hip)
  float* col_part;          // fused albedo backward: per-tile column sums of the output layer's gradient [tiles][Co][256] + [tiles][Co]
  float* sdfh_part;         // sdf-head row gradient: per-slab column sums [kSdfHeadSlabs][Hp] + [kSdfHeadSlabs]
};

// slots of PointBufs::amax: zb_l, u_l (u_0 = geb), zc_l, cinb
enum { AMAX_ZB = 0, AMAX_U = RNB_MAX_LIN, AMAX_ZC = 2 * RNB_MAX_LIN + 1, AMAX_CINB = 3 * RNB_MAX_LIN + 1, AMAX_SLOTS = 3 * RNB_MAX_LIN + 2 };
// slots of PointBufs::smax: a_l, gz_l (hidden layers), e (positional encoding), cin (albedo-net input), ac_l (its hidden layers)
enum { SMAX_A = 0, SMAX_GZ = RNB_MAX_LIN, SMAX_E = 2 * RNB_MAX_LIN, SMAX_CIN = 2 * RNB_MAX_LIN + 1, SMAX_AC = 2 * RNB_MAX_LIN + 2,
       SMAX_SLOTS = 3 * RNB_MAX_LIN + 2 };
enum PointMode { PM_SDF_ONLY = 0, PM_WITH_NORMAL = 1, PM_WITH_COLOR = 2, PM_WITH_BACKWARD = 4 };
void carve_points(const Layout& L, Carver& c, int64_t M, int mode, PointBufs* pb);

// ---- device sweeps (mlp.hip) -------------------------------------------------------------------
int launch_pe_points(const Layout& L, const float* pts, int64_t M, PointBufs& pb, hipStream_t s);
int sweep_forward(const Layout& L, const float* packed, PointBufs& pb, bool need_feat, bool need_gz_last,
                  float* feat_dense, hipStream_t s);
int sweep_reverse(const Layout& L, const float* packed, PointBufs& pb, hipStream_t s);
int sweep_color(const Layout& L, const float* packed, PointBufs& pb, const float* pts, const float* nrm, int nrm_ld,
                hipStream_t s);
int launch_copy_cols(const float* src, int ld, int ncols, int64_t M, float* out, hipStream_t s);
int launch_range_report(const Layout& L, const float* packed, const PointBufs& pb, bool with_color, bool with_backward, float* out,
                        hipStream_t s);
int launch_fill_cols(const float* src, int ncols, int64_t M, int64_t Mp, int ld, float* dst, hipStream_t s);
int64_t dw_partial_floats(const Layout& L, int64_t M, bool with_color);
int64_t dw_staged_floats(const Layout& L, int64_t M, bool with_color);
int sweep_backward(const Layout& L, const float* packed, PointBufs& pb, bool with_color, float* packed_grad,
                   bool fused, hipStream_t s);
int fused_reverse(const Layout& L, const float* packed, PointBufs& pb, hipStream_t s);
int fused_ra(const Layout& L, const float* packed, PointBufs& pb, hipStream_t s, int* u_tiles = nullptr);
int fused_fb(const Layout& L, const float* packed, PointBufs& pb, bool with_color, hipStream_t s);

// ---- fused sweeps for hidden width 256 (fused.hip) ---------------------------------------------------
bool fused_supported(const Layout& L);
int fused_forward(const Layout& L, const float* packed, const float* pts, int64_t M, PointBufs& pb, bool save,
                  bool need_feat, bool need_gz_last, hipStream_t s, const GridGen* grid = nullptr);
// ---- M/V sweeps (sweep_mv.hip): x3 arithmetic, matrix waves (32 points each, transposed product, weights through an
// LDS-DMA ring) + vector waves (epilogues, saved state, operand split) ----
bool sweep_mv_supported(const Layout& L);
int sweep_mv_forward(const Layout& L, const float* packed, const float* pts, int64_t M, PointBufs& pb, bool save,
                     bool need_feat, bool need_gz_last, hipStream_t s, const GridGen* grid = nullptr);
// family of a sweep over Mp points: the M/V kernels need >= one 128-point workgroup per CU to fill the chip
constexpr bool kRegTileDefault = false;
inline bool use_reg_tile(const Layout& L, int64_t Mp) {
  if (!sweep_mv_supported(L) || (L.variant & RNB_VARIANT_LDS_TILE)) return false;
  if (L.variant & RNB_VARIANT_REG_TILE) return true;
  return kRegTileDefault && Mp >= 128 * 200;
}
int launch_grid_points(const GridGen& g, int64_t first, int64_t n, float* pts, hipStream_t s);
int launch_scale_copy(const float* src, float scale, int64_t n, float* dst, hipStream_t s);

// ---- RNB_VARIANT_X3: fp32 products as six bf16 MFMA terms (fused_common.hip.h); the split weight mirror ----
inline bool is_x3(const Layout& L) { return (L.variant & RNB_VARIANT_X3) != 0; }
inline bool is_x2h(const Layout& L) { return is_x3(L) && (L.variant & RNB_VARIANT_X2H) != 0; }
// the fp16 two-plane mirror (RNB_VARIANT_X2H) behind the three bf16 planes: matrix at 2 x its float offset, in 2-byte units
inline unsigned short* x2h_mirror(const Layout& L, float* packed) {
  return reinterpret_cast<unsigned short*>(packed + L.total + L.total / 2 * 3);
}
inline const unsigned short* x2h_mirror(const Layout& L, const float* packed) {
  return reinterpret_cast<const unsigned short*>(packed + L.total + L.total / 2 * 3);
}
inline const H2Tab* h2_tab(const Layout& L, const float* packed) {
  return L.h2tab_off >= 0 ? reinterpret_cast<const H2Tab*>(packed + L.h2tab_off) : nullptr;
}
inline H2Tab* h2_tab(const Layout& L, float* packed) {
  return L.h2tab_off >= 0 ? reinterpret_cast<H2Tab*>(packed + L.h2tab_off) : nullptr;
}
int x3_pack_weights(const Layout& L, float* packed, hipStream_t s);

// ---- the albedo network as two fused sweeps in the x2h arithmetic (color_h2.hip) ----
bool color_h2_supported(const Layout& L);
int color_h2_forward(const Layout& L, const float* packed, PointBufs& pb, const float* pts, const float* nrm, hipStream_t s);
int color_h2_backward(const Layout& L, const float* packed, PointBufs& pb, hipStream_t s);
int64_t color_h2_part_floats(const Layout& L, int64_t M);
constexpr int kSdfHeadSlabs = 64;   // row slabs of sdf_head_bwd_kernel's partial sums (summed in slab order: no atomics)

// ---- RNB_VARIANT_BF16 (bf16.hip): bf16-operand sweeps of the 256-wide network, saved state in bf16 "K8" layout ----
inline bool is_bf16(const Layout& L) { return (L.variant & RNB_VARIANT_BF16) != 0; }
int bf16_pack_weights(const Layout& L, float* packed, hipStream_t s);
int bf16_forward(const Layout& L, const float* packed, const float* pts, int64_t M, PointBufs& pb, bool save, bool need_feat,
                 hipStream_t s, const GridGen* grid = nullptr, bool feat_k8 = false);
int bf16_reverse(const Layout& L, const float* packed, PointBufs& pb, hipStream_t s);
bool bf16_color_supported(const Layout& L);
int bf16_color_forward(const Layout& L, const float* packed, PointBufs& pb, const float* pts, hipStream_t s);
int bf16_color_backward(const Layout& L, const float* packed, PointBufs& pb, float* packed_grad, hipStream_t s);
int bf16_backward(const Layout& L, const float* packed, PointBufs& pb, bool with_color, bool color_bf16, float* packed_grad,
                  hipStream_t s);
int64_t bf16_dw_partial_floats(const Layout& L, int64_t M, bool with_color);

// ---- sampling / composite ------------------------------------------------------------------------
int launch_up_sample_step(const float* rays_o, const float* rays_d, const float* z_in, const float* sdf_old,
                          const float* sdf_new, const int32_t* gather_index, int n_old_for_gather,
                          int64_t B, int n, int n_new, float inv_s, float* new_z, int32_t* inds,
                          float* z_out, int32_t* sort_index, float* new_pts, float* sdf_sorted_out,
                          hipStream_t s);

// ---- composite (composite.hip) -------------------------------------------------------------------
struct CompArgs {
  int64_t B;
  int S, L, C;
  int flags;
  float cos_anneal;
  const float* rays_d;
  const float* pts;       // [B*S,3]
  const float* dists;     // [B,S]
  const float* sdf;       // [Mp]
  const float* nrm;       // [Mp,4]
  const float* alb;       // [Mp,4] (network output)
  const float* lights;    // [L,3] or [L,B,3]
  const float* bg;        // [3] or nullptr
  const float* variance;
  // forward outputs
  float* color_fine;
  float* weights;
  float* cdf;
  float* gradients;
  float* inside;
  float* weight_sum;
  float* weight_max;
  float* s_val;
  float* gerr_part;       // [B,2]
  float* gerr;            // [1] gradient_error, gerr_den [1] its denominator, gerr_partial [2] or nullptr: this shard's sums
  float* gerr_den;
  float* gerr_partial;
  float* sdf_out;         // optional copies
  float* albedo_out;
};

struct CompBwdArgs {
  CompArgs f;
  const float* weights;      // saved forward weights [B,S]
  const float* g_color;      // cotangents (nullable)
  const float* g_weights;
  const float* g_cdf;
  const float* g_gradients;
  const float* g_weight_sum;
  const float* g_weight_max;
  const float* g_s_val;
  const float* g_gerr;
  const float* gerr_den;     // [1]
  const float* gerr_den_global;   // [1] or nullptr: denominator of the whole data-parallel batch
  float* sbar;               // [Mp]
  float* nbar;               // [Mp,4]
  float* albbar;             // [Mp,4]
  float* invs_part;          // [B] partial d loss / d inv_s
  float* dvar;               // [1] d loss / d variance
  unsigned* amax_to_zero;    // PointBufs::amax (AMAX_SLOTS words) zeroed by the first workgroup, or nullptr
};

int launch_fine_points(const float* rays_o, const float* rays_d, const float* z, int64_t B, int S, float sample_dist,
                       float* pts, float* dists, unsigned* smax_to_zero, hipStream_t s);
int launch_composite_fwd(const CompArgs& a, hipStream_t s);
int launch_composite_bwd(const CompBwdArgs& g, hipStream_t s);

// ---- sampling (sampling.hip) ---------------------------------------------------------------------
int launch_z_init(const float* rays_o, const float* rays_d, const float* near, const float* far,
                  const float* t_rand, int64_t B, int n, float* z, float* pts, hipStream_t s);
int launch_gather_sdf(const float* sdf_old, const float* sdf_new, const int32_t* index, int64_t B, int n, int n_new,
                      float* out, hipStream_t s);

// ---- weight norm (weightnorm.hip) ------------------------------------------------------------------
int weightnorm_fwd(const rnb_model_desc* d, const Layout& L, const rnb_mlp_params* sdf, const rnb_mlp_params* color,
                   float* packed, hipStream_t s);
int weightnorm_bwd(const rnb_model_desc* d, const Layout& L, const rnb_mlp_params* sdf, const rnb_mlp_params* color,
                   const float* pgrad, const rnb_mlp_grads* gs, const rnb_mlp_grads* gc, hipStream_t s);
const char* last_error();

// ---- optional GEMM event instrumentation (prof.hip) ----------------------------------------------
bool prof_enabled();
void prof_begin(double flops, hipStream_t s, const char* tag);
void prof_end(hipStream_t s);
int profile_enable(int on);
int profile_collect(double* ms, int64_t* launches, double* flops);
int64_t profile_report(char* out, int64_t cap);
struct ProfScope {
  hipStream_t s;
  bool on;
  // tag: kernel class for the per-class report (a string literal: it is kept by pointer)
  ProfScope(double flops, hipStream_t st, const char* tag = nullptr) : s(st), on(prof_enabled()) {
    if (on) prof_begin(flops, s, tag);
  }
  ~ProfScope() {
    if (on) prof_end(s);
  }
};

}  // namespace rnb
