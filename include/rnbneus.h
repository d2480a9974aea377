/*
 * rnbneus.h — C ABI of librnbneus_hip.so, the MI355X (gfx950) implementation of the RNb-NeuS
 * volumetric SDF renderer hot path.
 *
 * The reference (rti-team-imvia/RNb-NeuS-fork) has no FFI layer: its boundary is the Python object
 * protocol between exp_runner.py and models/{embedder,fields,renderer}.py.  Each entry point below
 * names the reference interface it replaces (file:line relative to the reference root).  The Python
 * drop-in classes in rnb-neus-fork_amd/ bind these symbols with ctypes (INTEGRATION.md shows the
 * stub); nothing in the signatures depends on PyTorch: plain device pointers, sizes, a hipStream_t.
 *
 * Conventions
 *   - every function returns 0 on success or a negative RNB_E_* code; it never throws, never calls
 *     exit and never synchronises the device.  rnb_last_error_string() describes the last failure on
 *     the calling thread.
 *   - all pointers are DEVICE pointers to fp32 (or int32 where stated) unless marked host.
 *   - the caller owns every buffer (inputs, outputs, workspaces); the library allocates nothing that
 *     outlives a call and keeps no mutable global state.  Work is enqueued on the given stream only.
 *   - tensors are dense row-major with the shapes given in brackets.
 */
#ifndef RNBNEUS_H
#define RNBNEUS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RNB_ABI_VERSION 5
#define RNB_MAX_LIN 16 /* linear layers per MLP */

enum {
  RNB_OK = 0,
  RNB_E_INVALID = -1,     /* bad argument / unsupported configuration */
  RNB_E_WORKSPACE = -2,   /* workspace too small */
  RNB_E_HIP = -3,         /* a HIP runtime call or launch failed */
  RNB_E_NULL = -4         /* required pointer is NULL */
};

typedef void* rnb_stream_t; /* hipStream_t */

/* Constructor arguments of the three networks and the renderer, i.e. the `model { ... }` block of
 * confs/wmask_rnb.conf:53-90 as splatted into SDFNetwork(**conf) (models/fields.py:9-20),
 * RenderingNetwork(**conf) (models/fields.py:132-141) and NeuSRenderer(**conf)
 * (models/renderer.py:73-82). */
typedef struct rnb_model_desc {
  /* SDFNetwork */
  int32_t sdf_d_in;       /* must be 3 */
  int32_t sdf_d_out;      /* 1 + feature width (257) */
  int32_t sdf_d_hidden;   /* 256 */
  int32_t sdf_n_layers;   /* 8 hidden layers => 9 linear layers */
  int32_t sdf_skip_in;    /* single skip layer index (4) or -1 */
  int32_t sdf_multires;   /* 6 */
  float   sdf_scale;      /* 1.0 */
  int32_t sdf_weight_norm;
  /* RenderingNetwork, mode "no_view_dir" only */
  int32_t col_d_feature;  /* 256 */
  int32_t col_d_in;       /* 6 (points + normals; view dirs are ignored in this mode) */
  int32_t col_d_out;      /* 3 */
  int32_t col_d_hidden;   /* 256 */
  int32_t col_n_layers;   /* 2 hidden layers => 3 linear layers */
  int32_t col_multires_view; /* 4 */
  int32_t col_squeeze_out;   /* 1: sigmoid on the output */
  int32_t col_weight_norm;
  /* NeuSRenderer */
  int32_t n_samples;      /* 64 */
  int32_t n_importance;   /* 64 */
  int32_t up_sample_steps;/* 4 */
  int32_t variant;        /* RNB_VARIANT_* bits: arithmetic / kernel variant of this model instance (0 = default) */
} rnb_model_desc;

/* rnb_model_desc.variant.  Every switch is an explicit field of the descriptor that accompanies each call: the
 * library reads no environment variables and keeps no process-global tuning state.
 *   RNB_VARIANT_BF16          BASELINE config 5: the SDF-network sweeps (forward, reverse normal, their adjoints and
 *                             the weight gradients) take bf16 operands on v_mfma_f32_32x32x16_bf16 with fp32
 *                             accumulators; the per-point saved state is bf16.  Master weights, gradients, sampling,
 *                             the albedo network and the composite stay fp32.  Needs the 256-wide network shape.
 *   RNB_VARIANT_DETERMINISTIC split-K partial sums go to workspace slabs and are reduced in a fixed order instead of
 *                             through fp32 atomics: gradients are bit-reproducible from run to run and across ranks.
 *   RNB_VARIANT_GENERIC       per-layer GEMM chain for every sweep even when the fused kernels support the shape.
 *   RNB_VARIANT_DW_LDS        register-staged LDS weight-gradient GEMMs (the guarded generic kernel) for every job.
 *   RNB_VARIANT_DW_STAGED     256 x 256 weight gradients through the LDS-DMA staged kernel (one 16-wave workgroup owns a
 *                             whole gradient of a point range: every operand byte is fetched once, partial gradients
 *                             leave through slabs + an ordered reduction, no atomics) instead of the register-direct
 *                             128 x 128-tile kernel.  Halves the operand traffic of that launch; measured 2 % slower
 *                             per step, hence not the default.
 *   RNB_VARIANT_X3            fp32 products of the fused sweeps and of the 256 x 256 weight gradients on the bf16 matrix
 *                             pipe: each fp32 operand as three bf16 terms (hi + mid + lo = x exactly), six of the nine
 *                             cross terms per product, fp32 accumulate.  Same fp32 state, same results to fp32 rounding
 *                             (the dropped terms are < 2^-26 relative); rnb_packed_floats grows by the split weight
 *                             mirror (1.5 x).  This is what variant 0 selects for the 256-wide network shape; the bit
 *                             only makes the request explicit (an unsupported shape is then an error).
 *   RNB_VARIANT_F32_MFMA      the same kernels on v_mfma_f32_32x32x2_f32 (the round-1 arithmetic; A/B switch).
 *   RNB_VARIANT_*_TI/_NW      tile height (1: 32 points, 2: 64 points) / waves per workgroup (4 or 8) of the fused
 *                             backward sweeps (BWD) and of the fused forward (FWD); 0 = the measured default.
 *   RNB_VARIANT_REG_TILE /    which family of fused x3 sweeps runs: REG_TILE = the M/V kernels (sweep_mv.hip: matrix waves
 *   RNB_VARIANT_LDS_TILE      with 32 points each and the weights through an LDS-DMA ring + vector waves for the
 *                             epilogues), LDS_TILE = the 64-point LDS-tile kernels (fused.hip / fused_bwd.hip).  Neither
 *                             bit: the measured default per sweep and batch size.  A/B switches.
 *   RNB_VARIANT_X2H /         (with X3; ON by default, NO_X2H switches it off) every fp32 product of the fused path except the
 *   RNB_VARIANT_NO_X2H        RA sweep's is taken as THREE fp16 matrix terms (x = hi + lo in fp16 after a power-of-two scale)
 *                             instead of six bf16 ones: half the matrix time; operands represented to 2^-22 (rms 2^-23.6;
 *                             the measured SDF error against fp64 is below the six-term scheme's).  Every scale
 *                             is TAKEN FROM THE DATA, so there is no operand range [round 5; through ABI 4 weights beyond 255
 *                             and activations beyond 1023 gave inf / NaN]: the weights' per matrix (2^8 while max |w| < 64,
 *                             else the power of two that puts the maximum in [2^13, 2^14); table behind the mirror in the
 *                             packed buffer); activations', network inputs' and Jacobian rows' per 64-point tile and layer (2^6
 *                             while the tile stays below 256); the saved state's that the weight gradients read per launch
 *                             (from maxima the forward leaves in the workspace); loss adjoints' per tile / per launch from
 *                             their recorded maxima, so any loss scale works (a loss times 2^k gives gradients times 2^k bit
 *                             for bit, tested at k = +-40).  In the old range the SDF network's sweeps give the same bits as ABI 4's.  The RA
 *                             sweep is bound by its saved-state traffic and keeps the six bf16 terms. */
enum {
  RNB_VARIANT_BF16 = 1,
  RNB_VARIANT_DETERMINISTIC = 2,
  RNB_VARIANT_GENERIC = 4,
  RNB_VARIANT_DW_LDS = 8,
  RNB_VARIANT_DW_STAGED = 16,
  RNB_VARIANT_X3 = 32,
  RNB_VARIANT_F32_MFMA = 64,
  RNB_VARIANT_BWD_TI_SHIFT = 8,   /* 2 bits: 0 default, 1, 2 */
  RNB_VARIANT_BWD_NW_SHIFT = 10,  /* 2 bits: 0 default, 1 = 4 waves, 2 = 8 waves */
  RNB_VARIANT_FWD_TI_SHIFT = 12,
  RNB_VARIANT_FWD_NW_SHIFT = 14,
  RNB_VARIANT_REG_TILE = 1 << 16,
  RNB_VARIANT_LDS_TILE = 1 << 17,
  RNB_VARIANT_X2H = 1 << 18,
  RNB_VARIANT_NO_X2H = 1 << 19
};

/* Trainable leaves of one MLP in the reference's state_dict naming (linN.weight_g [out,1],
 * linN.weight_v [out,in], linN.bias [out]).  With weight_norm == 0, v holds linN.weight and g is
 * ignored. */
typedef struct rnb_mlp_params {
  int32_t n_lin;
  int32_t pad_;
  const float* g[RNB_MAX_LIN];
  const float* v[RNB_MAX_LIN];
  const float* b[RNB_MAX_LIN];
} rnb_mlp_params;

typedef struct rnb_mlp_grads { /* same shapes as rnb_mlp_params, written (not accumulated) */
  int32_t n_lin;
  int32_t pad_;
  float* g[RNB_MAX_LIN];
  float* v[RNB_MAX_LIN];
  float* b[RNB_MAX_LIN];
} rnb_mlp_grads;

int rnb_abi_version(void);
/* Identity of this build of the library: 16 hex digits hashed from every source file and the compiler flags
 * (rnb-neus-fork_amd/buildid.py), or "unknown" for a library built without the project's build script.  Profiles and
 * bench lines carry it, so that stored measurements are only quoted for the build they were taken on.  [ABI 4] */
const char* rnb_build_id(void);
const char* rnb_last_error_string(void);

/* ---- weight norm ------------------------------------------------------------------------------
 * Replaces torch.nn.utils.weight_norm's forward/backward as applied at models/fields.py:72-74 and
 * :168-170.  The forward materialises W = g * v / ||v||_row for every layer of both MLPs into one
 * "packed" buffer (tile-padded, skip-layer 1/sqrt(2) folded in, layout private to the library) that
 * all compute entry points consume; the backward maps a gradient buffer of the same layout back to
 * the leaves. `color`/`color_grads` may be NULL (no_albedo training, exp_runner.py:111-112); `sdf` may be
 * NULL when only the albedo network is evaluated (rnb_color_forward).
 * rnb_packed_floats is the size of `packed` in floats: the fp32 weights plus, behind them, the MFMA-operand mirror of
 * the variant in use (written by rnb_weightnorm_fwd: hi / mid / lo bf16 planes for the default x3 arithmetic, a bf16
 * copy for RNB_VARIANT_BF16).  A gradient buffer (`packed_grad`) needs the same allocation size; only its fp32 part is
 * written and read. */
int rnb_packed_floats(const rnb_model_desc* desc, int64_t* n_floats);
int rnb_weightnorm_fwd(const rnb_model_desc* desc, const rnb_mlp_params* sdf, const rnb_mlp_params* color,
                       float* packed, rnb_stream_t stream);
int rnb_weightnorm_bwd(const rnb_model_desc* desc, const rnb_mlp_params* sdf, const rnb_mlp_params* color,
                       const float* packed_grad, const rnb_mlp_grads* sdf_grads,
                       const rnb_mlp_grads* color_grads, rnb_stream_t stream);

/* ---- point-wise network evaluation (no autograd) ---------------------------------------------
 * rnb_sdf_forward   : SDFNetwork.forward / .sdf      (models/fields.py:82-108)
 *                     sdf_out [n]; feat_out [n, d_out-1] or NULL
 * rnb_sdf_gradient  : SDFNetwork.gradient            (models/fields.py:114-127), grad_out [n,3];
 *                     sdf_out optional
 * rnb_color_forward : RenderingNetwork.forward       (models/fields.py:177-215), out [n, d_out]
 * Used by NeuSRenderer.extract_geometry's query_func (models/renderer.py:1219-1224), by the
 * up-sampling loop and by tests. */
int rnb_points_workspace_bytes(const rnb_model_desc* desc, int64_t n_points, int64_t* bytes);
int rnb_sdf_forward(const rnb_model_desc* desc, const float* packed, const float* pts, int64_t n,
                    float* sdf_out, float* feat_out, void* ws, size_t ws_bytes, rnb_stream_t stream);
int rnb_sdf_gradient(const rnb_model_desc* desc, const float* packed, const float* pts, int64_t n,
                     float* grad_out, float* sdf_out, void* ws, size_t ws_bytes, rnb_stream_t stream);
int rnb_color_forward(const rnb_model_desc* desc, const float* packed, const float* pts,
                      const float* normals, const float* feats, int64_t n, float* out, void* ws,
                      size_t ws_bytes, rnb_stream_t stream);

/* ---- SDF grid of validate_mesh --------------------------------------------------------------------
 * extract_fields (models/renderer.py:10-25) with query_func = -sdf_network.sdf (models/renderer.py:1219-1224):
 * volume[ix - x_begin, iy, iz] = out_scale * sdf(X[ix], Y[iy], Z[iz]) for x_begin <= ix < x_end, with
 * X = torch.linspace(bound_min[0], bound_max[0], resolution) etc. generated INSIDE the forward kernel (no point
 * buffer, no per-point workspace for the 256-wide network: 512^3 = 1.3e8 evaluations are one launch).  A rank of
 * a data-parallel job asks for its own x-slab.  volume: device [x_end - x_begin, resolution, resolution]. */
typedef struct rnb_grid_desc {
  float bound_min[3];
  float bound_max[3];
  int32_t resolution;
  int32_t x_begin, x_end;
  float out_scale;        /* -1: the reference negates the SDF */
} rnb_grid_desc;
int rnb_sdf_grid_workspace_bytes(const rnb_model_desc* desc, const rnb_grid_desc* grid, int64_t* bytes);
int rnb_sdf_grid(const rnb_model_desc* desc, const float* packed, const rnb_grid_desc* grid, float* volume,
                 void* ws, size_t ws_bytes, rnb_stream_t stream);

/* ---- hierarchical sampling ---------------------------------------------------------------------
 * rnb_up_sample_step: one iteration of the loop at models/renderer.py:970-982 WITHOUT the network
 *   call: NeuSRenderer.up_sample (:132-176) + sample_pdf(det=True) (:39-69) + the concat/sort half of
 *   cat_z_vals (:178-183).  z_in [B,n] sorted, sdf_in [B,n];
 *   outputs new_z [B,n_new], inds int32 [B,n_new] (= torch.searchsorted(cdf,u,right=True)),
 *   z_out [B,n+n_new] sorted, sort_index int32 [B,n+n_new] (= index of torch.sort over cat[z,new_z]).
 * rnb_gather_sdf: sdf_out[b,k] = cat[sdf_old, sdf_new][b, sort_index[b,k]]   (renderer.py:185-190)
 * rnb_sample_rays: the whole no-grad prologue shared by render / render_rnb / render_rnb_warmup
 *   (models/renderer.py:557-608 == :829-880 == :933-984): z = near + (far-near)*linspace(0,1,n_samples)
 *   (+ (t_rand-0.5)*2/n_samples when t_rand != NULL), coarse SDF, up_sample_steps x (up_sample,
 *   cat_z_vals incl. SDF evaluation of the new points).  z_vals_out [B, n_samples+n_importance].
 *   `t_rand` [B] is the torch.rand([B,1]) draw of renderer.py:572, made by the caller. */
int rnb_up_sample_step(const float* rays_o, const float* rays_d, const float* z_in, const float* sdf_in,
                       int64_t B, int32_t n, int32_t n_new, float inv_s, float* new_z, int32_t* inds,
                       float* z_out, int32_t* sort_index, rnb_stream_t stream);
int rnb_gather_sdf(const float* sdf_old, const float* sdf_new, const int32_t* sort_index, int64_t B,
                   int32_t n, int32_t n_new, float* sdf_out, rnb_stream_t stream);
int rnb_sample_workspace_bytes(const rnb_model_desc* desc, int64_t B, int64_t* bytes);
int rnb_sample_rays(const rnb_model_desc* desc, const float* packed, const float* rays_o,
                    const float* rays_d, const float* near, const float* far, const float* t_rand,
                    int64_t B, float* z_vals_out, void* ws, size_t ws_bytes, rnb_stream_t stream);

/* ---- fine pass: render cores + wrappers ---------------------------------------------------------
 * rnb_render_fwd evaluates NeuSRenderer.render_core (models/renderer.py:194-285; mode RNB_MODE_CORE,
 * the composite of NeuSRenderer.render :632-648) or render_core_mvps (:466-554) followed by the
 * multi-light shading composite of render_rnb (:1009-1033; RNB_MODE_MVPS) / render_rnb_warmup
 * (:905-930; RNB_MODE_MVPS | RNB_FLAG_RELU_SHADING) on given sorted z_vals, and keeps what the
 * backward needs in `ws`.  rnb_render_bwd is loss.backward() through that graph
 * (exp_runner.py:261): it consumes gradients w.r.t. the returned tensors and writes gradients w.r.t.
 * the packed weights and the variance scalar. */
enum {
  RNB_MODE_CORE = 0,            /* render(): colour = sum_s c*w (+ background_rgb) */
  RNB_MODE_MVPS = 1,            /* render_rnb*: colour[l] = sum_s albedo*w*(n.l)    */
  RNB_FLAG_RELU_SHADING = 2,    /* warm-up: relu on n.l (renderer.py:913)           */
  RNB_FLAG_NO_ALBEDO = 4,       /* albedo := 1 (renderer.py:905-906, :1009-1010)    */
  RNB_FLAG_LIGHT_PER_RAY = 8,   /* lights_dir is [L,B,3] instead of [L,3]           */
  RNB_FLAG_FORWARD_ONLY = 16    /* no backward will follow                          */
};

typedef struct rnb_render_args {
  int64_t B;              /* rays */
  int32_t S;              /* samples per ray (n_samples + n_importance) */
  int32_t n_lights;       /* L (MVPS) ; ignored for CORE */
  int32_t flags;          /* RNB_MODE_* | RNB_FLAG_* */
  float   cos_anneal_ratio;
  const float* rays_o;    /* [B,3] */
  const float* rays_d;    /* [B,3] */
  const float* z_vals;    /* [B,S] sorted */
  const float* lights_dir;/* [L,3] or [L,B,3] */
  const float* background_rgb; /* [3] or NULL (CORE only, renderer.py:266-267) */
  const float* variance;  /* [1] SingleVarianceNetwork.variance (models/fields.py:317-325) */
  /* outputs (keys of the dict returned at renderer.py:638-648 / :1023-1033) */
  float* color_fine;      /* MVPS: [L,B,C]  CORE: [B,3] ; C = col_d_out */
  float* weights;         /* [B,S] */
  float* cdf_fine;        /* [B,S] */
  float* gradients;       /* [B,S,3] */
  float* inside_sphere;   /* [B,S] */
  float* weight_sum;      /* [B] */
  float* weight_max;      /* [B] */
  float* s_val;           /* [B] */
  float* gradient_error;  /* [1] */
  /* optional extra outputs of the cores (may be NULL) */
  float* sdf;             /* [B*S] */
  float* sampled_albedo;  /* [B*S,C] (network output, before the no_albedo override) */
  /* data parallelism with the exact large-batch loss: the eikonal term of models/renderer.py:538-540 is
   * a ratio of two batch-global sums.  rnb_render_fwd writes this shard's sums (numerator, count) to
   * gerr_partial [2] when it is non-NULL; rnb_render_bwd divides by *gerr_den_global (the all-reduced count + 1e-5)
   * instead of the shard's own denominator when it is non-NULL.  Both NULL: single-process behaviour. */
  float* gerr_partial;
  const float* gerr_den_global;
} rnb_render_args;

typedef struct rnb_render_grads { /* d loss / d <output>; NULL = zero */
  const float* color_fine;
  const float* weights;
  const float* cdf_fine;
  const float* gradients;
  const float* weight_sum;
  const float* weight_max;
  const float* s_val;
  const float* gradient_error;
} rnb_render_grads;

int rnb_render_workspace_bytes(const rnb_model_desc* desc, int64_t B, int32_t S, int32_t flags,
                               int64_t* bytes);
int rnb_render_fwd(const rnb_model_desc* desc, const float* packed, const rnb_render_args* args,
                   void* ws, size_t ws_bytes, rnb_stream_t stream);
int rnb_render_bwd(const rnb_model_desc* desc, const float* packed, const rnb_render_args* args,
                   const rnb_render_grads* gout, float* packed_grad, float* variance_grad, void* ws,
                   size_t ws_bytes, rnb_stream_t stream);

/* Name/duration of the heaviest kernel family, for bench.py's roofline line: fills `flops` with the
 * algorithmic MLP FLOPs of one rnb_render_fwd+bwd (+ sampling) at the given shape. */
int rnb_algorithmic_flops(const rnb_model_desc* desc, int64_t B, int32_t flags, double* train_flops,
                          double* forward_flops);

/* Algorithmic HBM bytes of the per-point saved state one training step (rnb_render_fwd + rnb_render_bwd) moves when
 * every saved matrix is written once and read once by each kernel that consumes it — the traffic floor of the
 * "store, don't recompute" design, which is what bounds the RNB_VARIANT_BF16 step.  Per point:
 * SDF sweeps 15 * n_layers * d_hidden elements (6 matrices written, 9 matrix reads per layer; 2 bytes each with
 * RNB_VARIANT_BF16, else 4) + the albedo network's fp32 activations. */
int rnb_algorithmic_bytes(const rnb_model_desc* desc, int64_t B, int32_t flags, double* train_bytes);

/* ---- Marching cubes of validate_mesh --------------------------------------------------------------
 * Replaces `mcubes.marching_cubes(u, threshold)` (models/renderer.py:31 inside extract_geometry :27-36; called by
 * validate_mesh, exp_runner.py:561-581) on a volume resident in device memory: volume [nx, ny, nz] fp32, C order (the
 * array extract_fields returns).  PyMCubes (third party, un-vendored, not importable here) is the arithmetic being
 * replaced: PARITY UNPINNED; kept from its published behaviour are the classic corner / edge numbering, "inside" :=
 * value <= threshold, ONE vertex per crossed grid edge shared by the cells around it, linear interpolation in double
 * precision and vertices in grid-index coordinates (the caller rescales them, models/renderer.py:34).  Triangles are
 * oriented so that their right-hand normal points towards smaller values.  NaN samples count as outside; an edge
 * between a NaN and an inside sample yields a NaN vertex.
 * Two calls, because the output sizes are data dependent:
 *   rnb_marching_cubes_count  fills counts[0..1] (device int64: vertices, triangles) and the workspace;
 *   -- the caller reads counts, allocates vertices [n_vertices, 3] double and triangles [n_triangles, 3] int32 --
 *   rnb_marching_cubes_emit   same volume / threshold / workspace; writes both arrays.  Deterministic order: vertices
 *                             by owning grid point (x slowest) then edge axis; triangles by cell, then table order.
 * Fails with RNB_E_INVALID beyond 2^30 vertices (vertex ids are 30-bit) or 2^32 grid points. */
int rnb_marching_cubes_workspace_bytes(int32_t nx, int32_t ny, int32_t nz, int64_t* bytes);
int rnb_marching_cubes_count(const float* volume, int32_t nx, int32_t ny, int32_t nz, float threshold,
                             void* workspace, size_t workspace_bytes, int64_t* counts, rnb_stream_t stream);
int rnb_marching_cubes_emit(const float* volume, int32_t nx, int32_t ny, int32_t nz, float threshold,
                            void* workspace, size_t workspace_bytes, int64_t n_vertices, int64_t n_triangles,
                            double* vertices, int32_t* triangles, rnb_stream_t stream);

/* Ray / target generation of one train_rnb step on the device: what Dataset.ps_gen_random_rays_at_view_on_all_lights
 * (models/dataset.py:351-376), the per-pixel light gather (exp_runner.py:214-220) and near_far_from_sphere
 * (models/dataset.py:448-458) compute on the host, for one view whose tensors are resident in device memory.
 *   intrinsics_inv, pose [4,4] row-major (intrinsics_all_inv[img_idx], pose_all[img_idx]);
 *   images, images_warmup, light_directions [n_lights, H, W, 3] of the view (each may be NULL with its output);
 *   mask [H, W, mask_channels] (channel 0 is used); pixels_x, pixels_y [B] int64, 0 <= x < W, 0 <= y < H.
 * Outputs: data [B,7] = rays_o | rays_v | mask; true_rgb, true_rgb_warmup, lights_dir [n_lights, B, 3];
 * near, far [B] (both or neither). */
int rnb_gen_rays_at_view(const float* intrinsics_inv, const float* pose, const float* images,
                         const float* images_warmup, const float* mask, int32_t mask_channels,
                         const float* light_directions, const int64_t* pixels_x, const int64_t* pixels_y, int64_t B,
                         int32_t n_lights, int32_t H, int32_t W, float* data, float* true_rgb,
                         float* true_rgb_warmup, float* lights_dir, float* near, float* far, rnb_stream_t stream);

/* The loss of train_rnb (exp_runner.py:241-258) and its gradients with respect to the renderer outputs, one
 * launch:  loss = sum|(color_fine - true_rgb) * mask| / ((sum(mask) + 1e-5) * n_lights)
 *               + igr_weight * gradient_error
 *               + mask_weight * mean BCE(clip(weight_sum, 1e-3, 1 - 1e-3), mask),   mask := (mask > 0.5)
 * (mask := 1 when mask_weight == 0, exp_runner.py:233-236).  color_fine, true_rgb [n_lights, B, color_depth];
 * mask, weight_sum [B]; gradient_error [1].  loss [1]; parts [3] = color_loss, eikonal_loss, mask_loss;
 * d_color_fine / d_weight_sum / d_gradient_error = d loss / d input (same shapes as the inputs). */
int rnb_loss_rnb(const float* color_fine, const float* true_rgb, const float* mask, const float* weight_sum,
                 const float* gradient_error, int32_t n_lights, int64_t B, int32_t color_depth, float igr_weight,
                 float mask_weight, float* loss, float* parts, float* d_color_fine, float* d_weight_sum,
                 float* d_gradient_error, rnb_stream_t stream);
/* The same loss for one shard of a data-parallel batch, normalised by the GLOBAL batch (exp_runner.py:194
 * mask_sum and :251 BCE mean): batch_global [2] device floats = the all-reduced { sum of (mask > 0.5) (without the
 * 1e-5), number of rays } of the whole batch — the ray count is taken from the collective, not assumed to be
 * B * world, so unequal shards stay exact; gradient_error [1] = the GLOBAL eikonal ratio; eik_share = this shard's
 * share of it in the returned loss value (1/world).  loss/parts are this shard's ADDITIVE share: their sum over the
 * shards is the loss of the whole batch, and the sum over the shards of the gradients is its gradient.  (ABI 3: the
 * ray count moved from a host integer into batch_global[1].) */
int rnb_loss_rnb_shard(const float* color_fine, const float* true_rgb, const float* mask, const float* weight_sum,
                       const float* gradient_error, int32_t n_lights, int64_t B, int32_t color_depth, float igr_weight,
                       float mask_weight, const float* batch_global, float eik_share,
                       float* loss, float* parts, float* d_color_fine, float* d_weight_sum, float* d_gradient_error,
                       rnb_stream_t stream);

/* torch.optim.Adam.step() of exp_runner.py:115/:262 (amsgrad off) over ONE flat buffer of n parameters:
 * exp_avg / exp_avg_sq are the optimizer state, `step` the 1-based step count.  lr..weight_decay are host
 * scalars (the learning-rate schedule of exp_runner.py:327-337 changes lr every step). */
int rnb_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr,
                  double beta1, double beta2, double eps, double weight_decay, int64_t step, rnb_stream_t stream);

/* Diagnostic (not part of the reference's interface): the largest magnitudes of one render's operands, read back from the
 * workspace of rnb_render_fwd (after it) and, for out[4], of rnb_render_bwd (after it) — same desc / B / S / flags / ws as those
 * calls.  out: device [8] floats:
 *   [0] max |effective weight| over the matrices of both networks (0 unless the default x2h arithmetic keeps its scale table)
 *   [1] max |hidden activation or encoded input| of the SDF network      [2] max |Jacobian row| of the normal's reverse sweep
 *   [3] max |input or hidden activation| of the albedo network           [4] max |loss adjoint| recorded by the backward (x2h)
 *   [5..7] 0 (reserved).
 * The default arithmetic (RNB_VARIANT_X2H) takes every fp16 operand scale from the data — per matrix, per tile and layer, per
 * launch — so none of these has a limit; the numbers document how far a model is from the fixed scales' old range (255 /
 * 1023), e.g. over a training run (tools/soak.py, profiles/r05_x2h_range.txt).  Costs one pass over the saved state: not
 * for the hot path.  [ABI 5] */
int rnb_render_range(const rnb_model_desc* desc, const float* packed, const void* ws, size_t ws_bytes, int64_t B, int32_t S,
                     int32_t flags, float* out, rnb_stream_t stream);

/* Measurement aid for bench.py (not part of the reference's interface): while enabled, every launch of
 * the fp32-MFMA kernels (fused_*_kernel, gemm_dw_*_kernel, gemm_rows_kernel<...>) is bracketed by HIP events on
 * its launch stream.  After the caller has synchronised, rnb_profile_collect returns the summed device
 * time (ms), the number of launches and their summed algorithmic FLOPs (real layer shapes) since the
 * last enable/collect.  This is the only mutable global state of the library; it is off by default. */
int rnb_profile_enable(int on);
int rnb_profile_collect(double* gemm_ms, int64_t* gemm_launches, double* gemm_flops);
/* Per kernel class of the last rnb_profile_collect: one text line "<class> <ms> <launches> <flops>" each, written to
 * `out` (NUL-terminated, at most `capacity` bytes; out may be NULL); returns the number of bytes the full text needs. */
int64_t rnb_profile_report(char* out, int64_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* RNBNEUS_H */
