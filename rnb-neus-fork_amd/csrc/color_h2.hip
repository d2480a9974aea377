// The albedo network (RenderingNetwork, mode no_view_dir: models/fields.py:177-215) of the default fp32 path as TWO fused
// sweeps, in the arithmetic of the SDF network's (x2h: three fp16 matrix terms per fp32 product, gemm.hip.h):
//
//   color_fwd_h2_kernel   [feature | pe(p) | pe(n)] -> lin0 + relu -> lin1 + relu -> lin2 + sigmoid     (fields.py:179-214)
//   color_bwd_h2_kernel   its backward down to the adjoint of the input, the Jacobian product of the normal's encoding
//                         and geb = J_pe(x) nbar, the input of the RA sweep (what nbar_geb_kernel does per point)
//
// One 64-point tile per workgroup stays in LDS across the layers (same skeleton as fused_forward_kernel / fused_fb_h2_kernel:
// 4 waves of 64 rows x 64 columns, two workgroups per CU, weights as fragments from the fp16 mirror, saved state out through
// fire-and-forget buffer stores).  Replaces, per step: color_input_kernel, two gemm_rows launches and color_out_kernel in the
// forward; color_out_bwd_kernel, two gemm_rows launches and nbar_geb_kernel in the backward — every 65,536 x 256 activation
// made one HBM round trip between each pair of them.
// The 320 input columns pass through the 256-column tile in two parts: the 64 encoding columns first (K = 64), then the 256
// feature columns accumulate onto them (X3Mma::run_at) — a 320-column fp32 tile would leave room for one workgroup per CU.
// Operand ranges: none assumed.  Tiles of activations carry a per-tile power-of-two scale (kH2ActLimit, fused_common.hip.h),
// tiles of adjoints a per-tile, per-layer one taken from the values about to be stored (as fused_fb_h2_kernel), the weights
// the per-matrix scale of the mirror (H2Tab).
// The gradient of the output layer (<= 4 rows) leaves as per-tile column sums (plain stores into a slab, summed in tile order
// by dw_reduce_kernel): no atomics anywhere — the albedo network's gradients are bit-reproducible.
#include "fused_common.hip.h"

namespace rnb {

struct ColH2Args {
  const float* pts;        // [M,3]
  const float* nrm;        // [Mp,4]
  long long M;
  const float* packed;
  const x3raw* w2;         // fp16 mirror of the weight matrices (matrix at 2 x its float offset)
  const H2Tab* h2tab;
  int id0;                 // table id of the albedo net's first hidden layer
  int F, pev, multires_view, Cinp, Co, squeeze;
  long long w_off[2], wT_off[2], b_off[2];
  long long wo_off, bo_off;
  int ldwo;
  float* cin;              // [Mp,Cinp]: feature columns written by the F sweep; the forward adds the encoding columns
  float* ac[2];            // [Mp,256]
  unsigned* ac0_mask;      // [tiles][256 threads][2]: relu'(ac_0) as one bit per element in the threads' accumulator layout —
                           // all the backward needs of ac_0 (2 MB instead of a 67 MB tile read with its latency exposed)
  float* alb;              // [Mp,4]
  unsigned* smax;          // PointBufs::smax (forward: slots SMAX_CIN, SMAX_AC + l grown) or nullptr
  // backward
  const float* albbar;     // [Mp,4]
  float* zc[2];            // [Mp,256]
  float* cinb;             // [Mp,Cinp]: the feature columns are written (FB sweep, feature head's weight gradient)
  const float* x4;         // [Mp,4]
  const float* nbar;       // [Mp,4]
  float* geb;              // [Mp,Ep]
  int multires, Ep;
  unsigned* amax;          // PointBufs::amax (slots AMAX_ZC + l, AMAX_CINB grown)
  float* dwo_part;         // [tiles][Co][256] column sums of zo^T ac_1 per tile
  float* dbo_part;         // [tiles][Co]
};

constexpr int CT = 64;     // points per tile

// relu with torch's NaN propagation (rnb_internal.h: relu_nan)
__device__ inline float relu_keep_nan(float x) { return x < 0.f ? 0.f : x; }

// hidden-layer epilogue of the forward: a = relu(acc * inv + b) -> tile (times kH2ActScale), HBM; returns the thread's max
// m = 2 m + (a > 0): appends one bit of the relu mask — a compare into VCC and an add-with-carry.  After 32 elements the
// first one appended sits in bit 31 (col_mask_bit).
__device__ inline void col_mask_push(unsigned& m, float a) {
  asm("v_cmp_gt_f32 vcc, %1, 0\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(a) : "vcc");
}
// all-ones / zero from the bit of element (ti, r) of a column tile's mask word
__device__ inline unsigned col_mask_bit(unsigned m, int ti, int r) { return (unsigned)(((int)(m << (ti * 16 + r))) >> 31); }

template <bool MASK>
__device__ inline float col_fwd_epilogue(const v16f (&acc)[2][2], float inv, const float* __restrict__ bias, float* X,
                                         float* __restrict__ out, int64_t row0, int n0, int lane, unsigned* __restrict__ mask_out) {
  const int h = lane >> 5, cl = lane & 31;
  const BufRsrc ro = tile_rsrc(out + (size_t)row0 * FH, CT * FH * 4);
  int amb = 0;   // (relu outputs are >= +0: their maximum on the bit patterns, h2_track2)
#pragma unroll
  for (int tj = 0; tj < 2; ++tj) {
    const int col = n0 + tj * 32 + cl;
    const float bc = bias[col];
    [[maybe_unused]] unsigned mk = 0u;
    const unsigned voff = (unsigned)(4 * h * FH + col) * 4u;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const int rowc = ti * 32 + (r & 3) + 8 * (r >> 2);
        const int row = rowc + 4 * h;
        const float a0 = relu_keep_nan(__builtin_fmaf(acc[ti][tj][r], inv, bc));
        const float a1 = relu_keep_nan(__builtin_fmaf(acc[ti][tj][r + 1], inv, bc));
        X[row * FP + col] = a0 * kH2ActScale;
        X[(row + 1) * FP + col] = a1 * kH2ActScale;
        h2_track2(amb, a0, a1);
        if constexpr (MASK) { col_mask_push(mk, a0); col_mask_push(mk, a1); }
        bstore(ro, voff, rowc * FH * 4, a0);
        bstore(ro, voff, (rowc + 1) * FH * 4, a1);
      }
    if constexpr (MASK) mask_out[tj] = mk;
  }
  return __builtin_bit_cast(float, amb);
}

// after a layer's closing barrier: the tile's scale for the next product.  Common case: the flag is down, the tile keeps
// kH2ActScale.  Flag up (some value reached kH2ActLimit): the waves exchange their maxima, rescale what they wrote and leave
// the tile's maximum in PointBufs::smax (fused_common.hip.h).
__device__ inline void col_tile_rescale(int* flag, float am, float* wmx, float* X, int n0, int lane, int wave, unsigned* smax_slot,
                                        int tid, float& sa, float& isa) {
  sa = kH2ActScale;
  isa = 1.f / kH2ActScale;
  if (h2_flag_up(flag)) {   // (workgroup-uniform)
    am = wave_max(am);
    if (lane == 0) wmx[wave] = am;
    lds_barrier();
    const float tm = tile_max<4>(wmx);
    if (smax_slot != nullptr && tid == 0) amax_tile_commit(smax_slot, tm);
    if (tid == 0) *reinterpret_cast<volatile int*>(flag) = 0;
    x2h_dyn_scale(__builtin_bit_cast(unsigned, tm), sa, isa);
    const float f = sa * (1.f / kH2ActScale);
    for_each_acc<2, 2>(n0, lane, [&](int, int, int, int col, int, int row) { X[row * FP + col] *= f; });
    lds_barrier();
  }
}

__global__ __launch_bounds__(256, 2) void color_fwd_h2_kernel(ColH2Args g) {
  constexpr int NT = 256;
  __shared__ __attribute__((aligned(16))) float X[CT * FP];
  __shared__ float wmx[4];
  __shared__ int ovf[2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
  const int64_t row0 = (int64_t)blockIdx.x * CT;
  const int n0 = wave * 64;
  const int nks0 = g.Cinp >> 4, ksF = g.F >> 4;   // weight steps per row of layer 0's matrix; first step of the encoding columns
  const x3raw* W0 = g.w2 + 2 * g.w_off[0];
  const x3raw* W1 = g.w2 + 2 * g.w_off[1];
  X3Mma<2, 2, 2> mm;
  const float iwsv = h2_iws_load(g.h2tab, lane);
  mm.request_at(W0, nks0, ksF, n0, lane);
  const int PW = g.Cinp - g.F;   // encoding columns (64)
  // the feature tile (written by the F sweep) is requested now and lands while the encoding is computed and multiplied
  vf4 fr[CT * (FH / 4) / NT];
#pragma unroll
  for (int i = 0; i < CT * (FH / 4) / NT; ++i) {
    const int idx = tid + NT * i;
    fr[i] = *reinterpret_cast<const vf4*>(g.cin + (size_t)(row0 + (idx >> 6)) * g.Cinp + (idx & 63) * 4);
  }

  // ---- pe(p), pe(n) (fp32 math, models/embedder.py:40-46) -> tile columns 0 .. PW: 4 threads per point = (which vector,
  //      even / odd octaves) ----
  float pm;
  {
    const int p = tid & 63, part = tid >> 6, which = part >> 1, sub = part & 1;
    const int64_t row = row0 + p;
    float v[3] = {0.f, 0.f, 0.f};
    if (row < g.M) {
      if (which == 0) { v[0] = g.pts[row * 3]; v[1] = g.pts[row * 3 + 1]; v[2] = g.pts[row * 3 + 2]; }
      else { v[0] = g.nrm[row * 4]; v[1] = g.nrm[row * 4 + 1]; v[2] = g.nrm[row * 4 + 2]; }
    }
    float* xr = X + p * FP + which * g.pev;
    if (sub == 0) {
#pragma unroll
      for (int d = 0; d < 3; ++d) xr[d] = v[d] * kH2ActScale;
      if (which == 1)
        for (int c = 2 * g.pev; c < PW; ++c) X[p * FP + c] = 0.f;
    }
    for (int k = sub; k < g.multires_view; k += 2) {
      const float f = (float)(1 << k);
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        float sn, co;
        sincosf(v[d] * f, &sn, &co);
        xr[3 + 6 * k + d] = sn * kH2ActScale;
        xr[3 + 6 * k + 3 + d] = co * kH2ActScale;
      }
    }
    pm = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fabsf(v[2]));   // (the only unbounded entries of the encoding)
    // (every wave leaves its own word: nothing to initialise; the layers' flags start from zero behind the same barrier)
    if (lane == 0) wmx[wave] = __builtin_amdgcn_ballot_w64(pm >= kH2ActLimit) != 0 ? 1.f : 0.f;
    if (tid < 2) ovf[tid] = 0;
  }
  __syncthreads();
  float sa_p = kH2ActScale, isa_p = 1.f / kH2ActScale;
  if (tile_max<4>(wmx) != 0.f) {   // (workgroup-uniform) a coordinate or a normal component beyond 256
    __syncthreads();
    pm = wave_max(pm);
    if (lane == 0) wmx[wave] = pm;
    __syncthreads();
    const float tm_p = fmaxf(tile_max<4>(wmx), 1.f);
    if (g.smax != nullptr && tid == 0) amax_tile_commit(g.smax + SMAX_CIN, tm_p);
    x2h_dyn_scale(__builtin_bit_cast(unsigned, tm_p), sa_p, isa_p);
    const float f = sa_p * (1.f / kH2ActScale);
    for (int idx = tid; idx < CT * PW; idx += NT) {
      const int r = idx / PW, c = idx - r * PW;
      X[r * FP + c] *= f;
    }
    __syncthreads();
  }
  // the encoding columns of the input go to HBM too: Y operand of layer 0's weight gradient
  for (int idx = tid; idx < CT * (PW / 4); idx += NT) {
    const int r = idx / (PW / 4), c4 = idx - r * (PW / 4);
    const vf4 v = *reinterpret_cast<const vf4*>(X + r * FP + c4 * 4) * isa_p;
    *reinterpret_cast<vf4*>(g.cin + (size_t)(row0 + r) * g.Cinp + g.F + c4 * 4) = v;
  }
  v16f acc[2][2];
  mm.run_at<false>(X, W0, nks0, ksF, PW >> 4, n0, lane, acc, W0, nks0, 0, n0);   // acc = pe . W0[:, F ..]^T
  lds_barrier();   // every wave has finished reading the encoding columns
  // ---- features -> tile ----
  float fm = 0.f;
#pragma unroll
  for (int i = 0; i < CT * (FH / 4) / NT; ++i) {
    const int idx = tid + NT * i;
    const vf4 v = fr[i];
    *reinterpret_cast<vf4*>(X + (idx >> 6) * FP + (idx & 63) * 4) = v * kH2ActScale;
    fm = fmaxf(fmaxf(fm, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  h2_raise_flag(fm, &ovf[0], lane);
  __syncthreads();
  float sa = kH2ActScale, isa = 1.f / kH2ActScale;
  if (h2_flag_up(&ovf[0])) {   // (workgroup-uniform) a feature beyond 256
    fm = wave_max(fm);
    if (lane == 0) wmx[wave] = fm;
    __syncthreads();
    const float tm = tile_max<4>(wmx);
    if (g.smax != nullptr && tid == 0) amax_tile_commit(g.smax + SMAX_CIN, tm);
    if (tid == 0) ovf[0] = 0;
    x2h_dyn_scale(__builtin_bit_cast(unsigned, tm), sa, isa);
    const float f = sa * (1.f / kH2ActScale);
    for (int idx = tid; idx < CT * (FH / 4); idx += NT) {
      vf4* q = reinterpret_cast<vf4*>(X + (idx >> 6) * FP + (idx & 63) * 4);
      *q = *q * f;
    }
    __syncthreads();
  }
  if (sa != sa_p) {   // (workgroup-uniform) the two parts of the product share one scale: that of the feature tile
    const float f = sa * isa_p;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = acc[ti][tj] * f;
  }
  mm.run_at<true>(X, W0, nks0, 0, ksF, n0, lane, acc, W1, FH >> 4, 0, n0);   // acc += feature . W0[:, 0 .. F]^T
  lds_barrier();   // in-place update: every wave has finished reading the tile
  {
    const float inv = isa * h2_iws_at(iwsv, g.id0);
    const float am = col_fwd_epilogue<true>(acc, inv, g.packed + g.b_off[0], X, g.ac[0], row0, n0, lane,
                                            g.ac0_mask + ((size_t)blockIdx.x * NT + tid) * 2);
    h2_raise_flag(am, &ovf[1], lane);
    lds_barrier();
    col_tile_rescale(&ovf[1], am, wmx, X, n0, lane, wave, g.smax ? g.smax + SMAX_AC : nullptr, tid, sa, isa);
  }
  mm.run(X, W1, FH, n0, lane, acc, nullptr, 0, 0);
  lds_barrier();
  {
    const float inv = isa * h2_iws_at(iwsv, g.id0 + 1);
    const float am = col_fwd_epilogue<false>(acc, inv, g.packed + g.b_off[1], X, g.ac[1], row0, n0, lane, nullptr);
    h2_raise_flag(am, &ovf[0], lane);
    lds_barrier();
    col_tile_rescale(&ovf[0], am, wmx, X, n0, lane, wave, g.smax ? g.smax + SMAX_AC + 1 : nullptr, tid, sa, isa);
  }
  // ---- output layer + sigmoid: fp32 weights on the VALU, 16 rows per wave ----
  {
    float w[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int u = 0; u < 4; ++u) w[c][u] = c < g.Co ? g.packed[g.wo_off + (long long)c * g.ldwo + lane + 64 * u] : 0.f;
    for (int rr = 0; rr < CT / 4; ++rr) {
      const int row = wave * (CT / 4) + rr;
      float sc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float a = X[row * FP + lane + 64 * u];
#pragma unroll
        for (int c = 0; c < 4; ++c) sc[c] = fmaf(a, w[c][u], sc[c]);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sc[c] += __shfl_xor(sc[c], o, 64);
      if (lane < 4) {
        float v = 0.f;
        if (lane < g.Co) {
          const float s = lane == 0 ? sc[0] : lane == 1 ? sc[1] : lane == 2 ? sc[2] : sc[3];
          v = __builtin_fmaf(s, isa, g.packed[g.bo_off + lane]);
          if (g.squeeze) v = 1.f / (1.f + expf(-v));
        }
        g.alb[(row0 + row) * 4 + lane] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void color_bwd_h2_kernel(ColH2Args g) {
  __shared__ __attribute__((aligned(16))) float X[CT * FP];
  __shared__ __attribute__((aligned(16))) float ZO[CT * 4];
  __shared__ float wm[2][4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
  const int64_t row0 = (int64_t)blockIdx.x * CT;
  const int n0 = wave * 64;
  const long long left = g.M - row0;
  const int rows_ok = left >= CT ? CT : (int)(left < 0 ? 0 : left);
  const x3raw* W1T = g.w2 + 2 * g.wT_off[1];
  const x3raw* W0T = g.w2 + 2 * g.wT_off[0];
  const float iwsv = h2_iws_load(g.h2tab, lane);

  // zo = albbar * sigmoid'  (rows >= M: 0)
  if (tid < CT) {
    const int64_t row = row0 + tid;
    const vf4 a4 = *reinterpret_cast<const vf4*>(g.alb + row * 4);
    const vf4 g4 = *reinterpret_cast<const vf4*>(g.albbar + row * 4);
    vf4 z;
#pragma unroll
    for (int c = 0; c < 4; ++c) z[c] = (c < g.Co && tid < rows_ok) ? g4[c] * (g.squeeze ? a4[c] * (1.f - a4[c]) : 1.f) : 0.f;
    *reinterpret_cast<vf4*>(ZO + tid * 4) = z;
  }
  AuxTile<2, 2> aA;
  prefetch_tile<2, 2>(g.ac[1], row0, n0, lane, aA);
  const vu2 mk0 = *reinterpret_cast<const vu2*>(g.ac0_mask + ((size_t)blockIdx.x * 256 + tid) * 2);   // relu'(ac_0), this thread's bits
  __syncthreads();
  if (tid < g.Co) {   // d b_out of this tile
    float t = 0.f;
    for (int r = 0; r < CT; ++r) t += ZO[r * 4 + tid];
    g.dbo_part[(size_t)blockIdx.x * g.Co + tid] = t;
  }
  v16f acc[2][2];
  X3Mma<2, 2, 2> mm;
  // ---- zc_1 = (zo W_out) * relu'(ac_1);  d W_out of this tile = zo^T ac_1 ----
  {
    const int lane_a = opaque_lane(lane);
    const int h = lane_a >> 5, cl = lane_a & 31;
    const BufRsrc rz = tile_rsrc(g.zc[1] + (size_t)row0 * FH, CT * FH * 4);
    float ds[2][4], wo[2][4];
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        wo[tj][c] = c < g.Co ? g.packed[g.wo_off + (long long)c * g.ldwo + n0 + tj * 32 + cl] : 0.f;
        ds[tj][c] = 0.f;
      }
    // (rows outermost: a row's zo is read once and used for both column tiles — with the column tiles outermost the
    // compiler keeps all 32 rows' zo live across them: 128 registers, spilled)
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rowc = ti * 32 + (r & 3) + 8 * (r >> 2);
        const vf4 zo = *reinterpret_cast<const vf4*>(ZO + (rowc + 4 * h) * 4);
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
          const int col = n0 + tj * 32 + cl;
          const float a = aA.v[ti][tj][r];
          const float t = fmaf(zo[0], wo[tj][0], fmaf(zo[1], wo[tj][1], fmaf(zo[2], wo[tj][2], zo[3] * wo[tj][3])));
          const float z = a > 0.f ? t : 0.f;
          acc[ti][tj][r] = z;
          bstore(rz, (unsigned)(4 * h * FH + col) * 4u, rowc * FH * 4, z);
#pragma unroll
          for (int c = 0; c < 4; ++c) ds[tj][c] = fmaf(zo[c], a, ds[tj][c]);
        }
      }
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float tot = ds[tj][c] + __shfl_xor(ds[tj][c], 32, 64);
        if (lane < 32 && c < g.Co) g.dwo_part[((size_t)blockIdx.x * g.Co + c) * FH + n0 + tj * 32 + lane] = tot;
      }
    const float mr = wave_max(acc_absmax<2, 2>(acc, lane, rows_ok));
    if (lane == 0) wm[1][wave] = mr;
  }
  __syncthreads();
  float s, inv;
  {
    const float tmax = tile_scale<4>(wm[1], s, inv);
    if (tid == 0 && g.amax != nullptr) amax_tile_commit(g.amax + AMAX_ZC + 1, tmax);
  }
  for_each_acc<2, 2>(n0, lane, [&](int tj, int ti, int r, int col, int, int row) { X[row * FP + col] = acc[ti][tj][r] * s; });
  mm.request(W1T, FH, n0, lane);
  lds_barrier();
  // ---- zc_0 = (zc_1 W_1) * relu'(ac_0) ----
  mm.run(X, W1T, FH, n0, lane, acc, W0T, FH, n0);   // (its tail requests the first weight steps of the product after it)
  {
    const float unscale = inv * h2_iws_at(iwsv, g.id0 + 1);
    const int lane_e = opaque_lane(lane);   // (per-lane offsets rebuilt here, not carried through the matrix loop: fused_bwd.hip)
    const int h = lane_e >> 5;
    const BufRsrc rz = tile_rsrc(g.zc[0] + (size_t)row0 * FH, CT * FH * 4);
    for_each_acc<2, 2>(n0, lane_e, [&](int tj, int ti, int r, int col, int rowc, int) {
      const float z = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, acc[ti][tj][r] * unscale) & col_mask_bit(mk0[tj], ti, r));
      bstore(rz, (unsigned)(4 * h * FH + col) * 4u, rowc * FH * 4, z);
      acc[ti][tj][r] = z;
    });
    const float mr = wave_max(acc_absmax<2, 2>(acc, lane_e, rows_ok));
    if (lane_e == 0) wm[0][wave] = mr;
  }
  lds_barrier();   // every wave has finished reading the tile; the maxima are visible
  {
    const float tmax = tile_scale<4>(wm[0], s, inv);
    if (tid == 0 && g.amax != nullptr) amax_tile_commit(g.amax + AMAX_ZC, tmax);
  }
  for_each_acc<2, 2>(n0, lane, [&](int tj, int ti, int r, int col, int, int row) { X[row * FP + col] = acc[ti][tj][r] * s; });
  lds_barrier();
  // ---- cinb = zc_0 W_0: the feature columns 0 .. 255 (the FB sweep's input, the feature head's weight gradient) ----
  const float unscale0 = inv * h2_iws_at(iwsv, g.id0);
  mm.run(X, W0T, FH, n0, lane, acc, nullptr, 0, 0);
  {
    const int lane_e = opaque_lane(lane);
    const int h = lane_e >> 5;
    const BufRsrc rc = tile_rsrc(g.cinb + (size_t)row0 * g.Cinp, CT * g.Cinp * 4);
    const unsigned rowb = (unsigned)g.Cinp * 4u;
    for_each_acc<2, 2>(n0, lane_e, [&](int tj, int ti, int r, int col, int rowc, int) {
      const float v = acc[ti][tj][r] * unscale0;
      bstore(rc, (unsigned)(4 * h) * rowb + (unsigned)col * 4u, rowc * rowb, v);
      acc[ti][tj][r] = v;
    });
    if (g.amax != nullptr) amax_commit(g.amax + AMAX_CINB, acc_absmax<2, 2>(acc, lane_e, rows_ok), lane_e);
  }
  // ---- ... and the 64 encoding columns F .. F + 63: four 32 x 32 blocks, one per wave (row half, column half) ----
  v16f pacc[1][1];
  {
    X3Mma<1, 1, 2> mq;
    const int rt = wave & 1, ct = wave >> 1;
    mq.request(W0T, FH, g.F + 32 * ct, lane);
    mq.run(X + rt * 32 * FP, W0T, FH, g.F + 32 * ct, lane, pacc, nullptr, 0, 0);
  }
  lds_barrier();   // every wave has finished reading zc_0: the tile becomes scratch
  {
    // G[row][0 .. 63] = cinb[:, F ..] (the adjoint of [pe(p) | pe(n)]) at tile columns 0 .. 63
    const int rt = wave & 1, ct = wave >> 1, h = lane >> 5, cl = lane & 31;
#pragma unroll
    for (int r = 0; r < 16; ++r) X[(rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * FP + 32 * ct + cl] = pacc[0][0][r] * unscale0;
  }
  lds_barrier();
  // ---- nbar_total = nbar + J_pe(n)^T cinb[pe(n)];  geb = J_pe(x) nbar_total -> tile columns 64 ..: four threads per point,
  //      each one octave in four of both encodings (one thread per point left three waves idle through 30 sincos) ----
  {
    const int p = tid & 63, q = tid >> 6;
    const int64_t row = row0 + p;
    const bool ok = p < rows_ok;
    float* part = X + p * FP + 128;          // [4][3] partial sums of J_pe(n)^T g at tile columns 128 .. 139
    {
      const float* gq = X + p * FP + g.pev;   // the pe(n) block of the adjoint
      float t[3] = {0.f, 0.f, 0.f};
      if (ok) {
        for (int k = q; k < g.multires_view; k += 4) {
          const float f = (float)(1 << k);
          const int c = 3 + 6 * k;
#pragma unroll
          for (int d = 0; d < 3; ++d) {
            float sn, co;
            sincosf(g.nrm[row * 4 + d] * f, &sn, &co);
            t[d] += f * (gq[c + d] * co - gq[c + 3 + d] * sn);
          }
        }
      }
#pragma unroll
      for (int d = 0; d < 3; ++d) part[q * 3 + d] = t[d];
    }
    __syncthreads();
    float nb[3] = {0.f, 0.f, 0.f};
    if (ok) {
      const float* gq = X + p * FP + g.pev;
#pragma unroll
      for (int d = 0; d < 3; ++d) nb[d] = g.nbar[row * 4 + d] + gq[d] + ((part[d] + part[3 + d]) + (part[6 + d] + part[9 + d]));
    }
    float* o = X + p * FP + 64;
    if (q == 0) {
      o[0] = nb[0]; o[1] = nb[1]; o[2] = nb[2];
      for (int c = 3 + 6 * g.multires; c < g.Ep; ++c) o[c] = 0.f;
    }
    for (int k = q; k < g.multires; k += 4) {
      const float f = (float)(1 << k);
      const int c = 3 + 6 * k;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        float sn, co;
        sincosf(g.x4[row * 4 + d] * f, &sn, &co);
        o[c + d] = f * co * nb[d];
        o[c + 3 + d] = -f * sn * nb[d];
      }
    }
  }
  __syncthreads();
  for (int idx = tid; idx < CT * (g.Ep / 4); idx += 256) {
    const int r = idx / (g.Ep / 4), c4 = idx - r * (g.Ep / 4);
    *reinterpret_cast<vf4*>(g.geb + (size_t)(row0 + r) * g.Ep + c4 * 4) = *reinterpret_cast<const vf4*>(X + r * FP + 64 + c4 * 4);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
bool color_h2_supported(const Layout& L) {
  return is_x2h(L) && !is_bf16(L) && L.F == FH && L.nc == 2 && L.Hc == FH && L.Hcp == FH && L.Cinp - L.F == 64 && L.Co >= 1 &&
         L.Co <= 4 && L.Ep == 64 && 2 * L.pev <= 64 && L.col[0].Kp == L.Cinp && L.col[1].Kp == FH;
}

static void col_fill(const Layout& L, const float* packed, PointBufs& pb, ColH2Args& g) {
  memset(&g, 0, sizeof(g));
  g.M = pb.M;
  g.packed = packed;
  g.w2 = x2h_mirror(L, packed);
  g.h2tab = h2_tab(L, packed);
  g.id0 = L.nh + 1;
  g.F = L.F;
  g.pev = L.pev;
  g.multires_view = L.multires_view;
  g.Cinp = L.Cinp;
  g.Co = L.Co;
  g.squeeze = L.squeeze;
  for (int l = 0; l < 2; ++l) {
    g.w_off[l] = L.col[l].w_off;
    g.wT_off[l] = L.col[l].wT_off;
    g.b_off[l] = L.col[l].b_off;
    g.ac[l] = pb.ac[l];
    g.zc[l] = pb.zc[l];
  }
  g.wo_off = L.colo.w_off;
  g.bo_off = L.colo.b_off;
  g.ldwo = L.colo.Kp;
  g.cin = pb.cin;
  g.alb = pb.alb;
  g.ac0_mask = pb.ac0_mask;
}

static double col_flops(const Layout& L, int64_t M) {
  double fl = 0;
  for (int l = 0; l < L.nc; ++l) fl += 2.0 * (double)M * L.col[l].N * L.col[l].K;
  return fl + 2.0 * (double)M * L.colo.N * L.colo.K;
}

int color_h2_forward(const Layout& L, const float* packed, PointBufs& pb, const float* pts, const float* nrm, hipStream_t s) {
  ColH2Args g;
  col_fill(L, packed, pb, g);
  g.pts = pts;
  g.nrm = nrm;
  g.smax = pb.smax;
  ProfScope prof(col_flops(L, pb.M), s, "albedo_fwd");
  hipLaunchKernelGGL(color_fwd_h2_kernel, dim3((unsigned)(pb.Mp / CT)), dim3(256), 0, s, g);
  RNB_CHECK_LAUNCH();
  return RNB_OK;
}

// floats of the per-tile slabs of the output layer's gradient
int64_t color_h2_part_floats(const Layout& L, int64_t M) { return (pad_rows(M) / CT) * (int64_t)L.Co * (FH + 1); }

int color_h2_backward(const Layout& L, const float* packed, PointBufs& pb, hipStream_t s) {
  ColH2Args g;
  col_fill(L, packed, pb, g);
  g.nrm = pb.nrm;
  g.albbar = pb.albbar;
  g.cinb = pb.cinb;
  g.x4 = pb.x;
  g.nbar = pb.nbar;
  g.geb = pb.geb;
  g.multires = L.multires;
  g.Ep = L.Ep;
  g.amax = pb.amax;
  const int64_t tiles = pb.Mp / CT;
  g.dwo_part = pb.col_part;
  g.dbo_part = pb.col_part + tiles * L.Co * FH;
  ProfScope prof(col_flops(L, pb.M), s, "albedo_bwd");
  hipLaunchKernelGGL(color_bwd_h2_kernel, dim3((unsigned)tiles), dim3(256), 0, s, g);
  RNB_CHECK_LAUNCH();
  return RNB_OK;
}

}  // namespace rnb
