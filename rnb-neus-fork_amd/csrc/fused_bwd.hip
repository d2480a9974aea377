// Fused reverse-shaped sweeps of the 256-wide SDF network (same skeleton as fused_forward_kernel: a tile of
// points per workgroup stays in LDS across all layers, weights stream from L2, saved state leaves through
// fire-and-forget buffer stores).  The per-element operands of the epilogues (D_l, gz_l, zR_l) are
// prefetched into registers with buffer loads issued inside the layer's matrix loop (after the first weight
// block has been requested), so they land while the matrix cores run.  Default shape: 32-point tiles, 8 waves
// of 32 x 32 outputs, two workgroups per CU = 4 waves per SIMD within 128 registers (see the variant table
// near the end of the file).
//
//   fused_reverse_kernel  R : gz_l = g_l * D_l, g_{l-1} = gz_l W_l, normal = J_pe^T g_e   (fields.py:114-127)
//   fused_ra_kernel       RA: adjoint of R (second-order terms zR_l and the u_l operands of dW)
//   fused_fb_kernel       FB: backward of the forward sweep, zb_{l-1} = (zb_l W_l) * D_{l-1} + zR_{l-1}
// Mathematical statement: oracle/explicit.py.  The reverse-shaped products use the transposed weight
// copies W^T that rnb_weightnorm_fwd emits, so every sweep streams weight rows the same way.
#include "fused_common.hip.h"

namespace rnb {

// TI = row tiles per workgroup (template parameter of every kernel below).  TI == 1: 32-point tiles, two
// activation tiles in LDS (a layer reads one and writes the other: one barrier per layer).  TI == 2: 64-point
// tiles, ONE tile updated in place behind a second barrier (two would leave room for only one workgroup per
// CU); the matrix loop then runs 64 MFMAs per weight block instead of 32, which is what lifts it from ~77 %
// to ~90 % of the matrix peak (tools/mfma_probe).

struct FusedBwdArgs {
  const float* packed;
  const x3raw* w3;      // RNB_VARIANT_X3: split mirror of the weight matrices (matrix at 3 x its float offset)
  int nh, skip, pe, multires, Ep;
  float inv_scale;
  int n_real[RNB_MAX_LIN];   // real output width of hidden layer l
  int Kp[RNB_MAX_LIN];       // padded input width of hidden layer l
  long long w_off[RNB_MAX_LIN], wT_off[RNB_MAX_LIN];
  long long wsdf_off, wfT_off;
  float* D[RNB_MAX_LIN];
  float* gz[RNB_MAX_LIN];
  float* u[RNB_MAX_LIN + 1];
  long long M;          // real points (rows >= M of the last tile are padding)
  float* ucol;          // RA: [tiles][FH] column sums of u_nh per point tile INSTEAD of the matrix u_nh (its only reader is
                        // the sdf-head row gradient, which wants exactly these sums), or nullptr: store u_nh
  float* zR[RNB_MAX_LIN];
  float* zb[RNB_MAX_LIN];
  const float* x4;      // [Mp,4]
  float* nrm;           // [Mp,4]      (R)
  const float* geb;     // [Mp,Ep]     (RA)
  const float* sbar;    // [Mp]        (FB)
  const float* fbar;    // [Mp,ld_fbar] first 256 columns, or nullptr (FB, no_albedo)
  int ld_fbar;
  unsigned* amax;       // PointBufs::amax (RA: slots AMAX_U + l of the u_l it writes; FB: AMAX_ZB + l), or nullptr
  const H2Tab* h2tab;   // x2h: scales of the fp16 mirror's matrices (hidden layer l: id l, feature head: id nh)
  unsigned* smax;       // x2h R sweep of a render forward: PointBufs::smax (slots SMAX_GZ + l grown by the tile maxima), or nullptr
};

// matrix loop of one layer (weights at float offset `off` of the packed buffer): two alternating weight-register
// sets, except for 64-point tiles with 64-column waves, which use the one-set ring (register budget).  X3: the same
// product as six bf16 MFMA terms per 16 k (fused_common.hip.h).
template <int TI, int TJ = 2, bool X3 = false, class Hook = NoHook, int NP = 3>
__device__ inline void layer_mma(const float* __restrict__ X, const FusedBwdArgs& g, long long off, int K, int n0, int lane,
                                 v16f (&acc)[TI][TJ], X3Mma<TI, TJ, NP>& mm, long long off_next, Hook hook = Hook(),
                                 int hook_late = 0) {
  // X3: this product's first weight steps were requested through `mm` (before the previous epilogue); off_next >= 0
  // names the product that follows (K = 256), whose first steps are requested as this one finishes
  // (64 x 64-output waves keep no weight registers across the epilogue: its two operand tiles need them)
  if constexpr (X3 && TI == 2 && TJ == 2) {
    mm.request(g.w3 + NP * off, K, n0, lane);
    mm.run(X, g.w3 + NP * off, K, n0, lane, acc, nullptr, 0, 0, hook);
  } else if constexpr (X3) mm.run(X, g.w3 + NP * off, K, n0, lane, acc, off_next >= 0 ? g.w3 + NP * off_next : nullptr, FH, n0, hook);
  else if constexpr (TI == 2 && TJ == 2) layer_mma_nt_ring<TI>(X, g.packed + off, K, n0, lane, acc, hook);
  else layer_mma_nt<TI, Hook, TJ>(X, g.packed + off, K, n0, lane, acc, hook, hook_late);
}

// ---------------------------------------------------------------------------------------------------------
// R sweep
// ---------------------------------------------------------------------------------------------------------
// H2 (with X3): the products as three fp16 terms (gemm.hip.h, "x2h"); the LDS tile then holds gz times SG and g.w3 is the
// fp16 mirror.  The operands are Jacobian rows of the SDF (d sdf / d a_l), not loss adjoints: the fixed scale 2^6 serves unless a
// tile holds a row beyond 256, which the writers notice (flag word in LDS) and answer with a scale from the tile's own maximum.
template <int TI, int NW = 4, bool X3 = false, bool H2 = false>
__global__ __launch_bounds__(64 * NW, (NW == 8 && TI == 2) ? 1 : 2) void fused_reverse_kernel(FusedBwdArgs g) {
  static_assert(!H2 || X3, "x2h is a form of the split-operand path");
  constexpr float SG = H2 ? kH2ActScale : 1.f;   // what a writer multiplies by; the tile's scale `sg` is per tile and layer (kH2ActLimit)
  constexpr int WP = H2 ? 2 : 3;
  constexpr int BT = 32 * TI;
  constexpr int NT = 64 * NW;   // threads
  constexpr int TJ = 8 / NW;    // 32-column tiles per wave
  // both operand tiles of an epilogue fit next to the weight fragments unless the wave owns 64 x 64 outputs
  [[maybe_unused]] constexpr bool BOTH_IN_LOOP = !(TI == 2 && TJ == 2);
  constexpr int NBUF = TI == 1 ? 2 : 1;
  __shared__ __attribute__((aligned(16))) float lds[NBUF * BT * FP + BT * FEP];
  __shared__ float wmx[8];                 // x2h: the waves' words (seed: "reached the limit"; rare path: maxima)
  __shared__ int ovf[2];                   // x2h: "a value of the tile just written reached kH2ActLimit", by layer parity
  float* X = lds;                          // input of the current layer
  float* Y = lds + (NBUF - 1) * BT * FP;   // output of the current layer (== X when updated in place)
  float* GE = lds + NBUF * BT * FP;        // d sdf / d e of the tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
  const int64_t row0 = (int64_t)blockIdx.x * BT;
  const int n0 = wave * 32 * TJ;
  [[maybe_unused]] float sg = SG, isg = 1.f / SG;   // x2h: the tile holds gz times sg
  [[maybe_unused]] float iwsv = 0.f;
  if constexpr (H2) iwsv = h2_iws_load(g.h2tab, lane);

  // seed: gz_{nh-1} = w_sdf * D_{nh-1}  (row 0 of the output layer is d sdf / d a_last)
  [[maybe_unused]] float sm = 0.f;
  {
    const float* Dl = g.D[g.nh - 1] + (size_t)row0 * FH;
    float* gzl = g.gz[g.nh - 1] + (size_t)row0 * FH;
    const float* ws = g.packed + g.wsdf_off;
    float m = 0.f;
    for (int idx = tid; idx < BT * FH / 4; idx += NT) {
      const int r = idx >> 6, c4 = idx & 63;
      const vf4 d = *reinterpret_cast<const vf4*>(Dl + r * FH + c4 * 4);
      const vf4 w = *reinterpret_cast<const vf4*>(ws + c4 * 4);
      const vf4 v = d * w;
      *reinterpret_cast<vf4*>(X + r * FP + c4 * 4) = v * SG;
      *reinterpret_cast<vf4*>(gzl + r * FH + c4 * 4) = v;
      if constexpr (H2) m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    for (int idx = tid; idx < BT * FEP; idx += NT) GE[idx] = 0.f;
    if constexpr (H2) {   // (every wave leaves its own word: nothing to initialise; see fused_forward_kernel)
      if (lane == 0) wmx[wave] = __builtin_amdgcn_ballot_w64(m >= kH2ActLimit) != 0 ? 1.f : 0.f;
      if (tid < 2) ovf[tid] = 0;
    }
    if constexpr (H2) sm = m;
  }
  __syncthreads();
  if constexpr (H2) {
    if (tile_max<NW>(wmx) != 0.f) {   // (workgroup-uniform)
      __syncthreads();
      sm = wave_max(sm);
      if (lane == 0) wmx[wave] = sm;
      __syncthreads();
      const float tm = tile_max<NW>(wmx);
      if (g.smax != nullptr && tid == 0) amax_tile_commit(g.smax + SMAX_GZ + g.nh - 1, tm);
      x2h_dyn_scale(__builtin_bit_cast(unsigned, tm), sg, isg);
      const float f = sg * (1.f / SG);
      for (int idx = tid; idx < BT * FH / 4; idx += NT) {
        vf4* q = reinterpret_cast<vf4*>(X + (idx >> 6) * FP + (idx & 63) * 4);
        *q = *q * f;
      }
      __syncthreads();
    }
  }

  v16f acc[TI][TJ];
  AuxTile<TI, TJ> aD;
  [[maybe_unused]] X3Mma<TI, TJ, WP> mm;
  if constexpr (X3 && !(TI == 2 && TJ == 2)) {
    if (g.nh > 1 || n0 < 64) mm.request(g.w3 + WP * g.wT_off[g.nh - 1], FH, n0, lane);
  }
  for (int l = g.nh - 1; l >= 1; --l) {
    // x2h: accumulator -> g: 1 / (scale of the tile x scale of this layer's matrix in the mirror)
    [[maybe_unused]] const float INV = H2 ? isg * h2_iws_at(iwsv, l) : 1.f;
    const long long nxt = (l > 1 || n0 < 64) ? g.wT_off[l - 1] : -1;   // layer 0's product: the waves of columns 0..63
    // x3, 64 x 64-output waves: the operand tile is requested AFTER the matrix loop, into the registers the loop's
    // fragments leave behind; the other workgroup of the CU multiplies while it travels
    constexpr bool LATE = X3 && TI == 2 && TJ == 2;
    layer_mma<TI, TJ, X3>(X, g, g.wT_off[l], FH, n0, lane, acc, mm, nxt,   // g = gz_l W_l  (columns = inputs of layer l)
                     [&]() { if constexpr (!LATE) prefetch_tile<TI, TJ>(g.D[l - 1], row0, n0, lane, aD); });
    const int lane_e = opaque_lane(lane);   // the epilogue's per-lane offsets are rebuilt here, not carried through the loop
    const int h = lane_e >> 5;
    if constexpr (LATE) prefetch_tile<TI, TJ>(g.D[l - 1], row0, n0, lane_e, aD);
    if constexpr (NBUF == 1) lds_barrier();   // every wave has finished reading the tile
    const bool is_skip = (l == g.skip);
    const int ksplit = is_skip ? FH - g.pe : FH;   // columns that belong to layer l-1's output
    const BufRsrc rg = tile_rsrc(g.gz[l - 1] + (size_t)row0 * FH, BT * FH * 4);
    [[maybe_unused]] float gm[2] = {0.f, 0.f};   // x2h: max |gz_{l-1}| of this thread
    for_each_acc_split<TI, TJ>(
        n0, lane_e, ksplit,
        [&](int tj, int ti, int r, int col, int rowc, int row) {
          const float gzv = acc[ti][tj][r] * INV * aD.v[ti][tj][r];
          Y[row * FP + col] = gzv * SG;
          bstore(rg, (unsigned)(4 * h * FH + col) * 4u, rowc * FH * 4, gzv);
          if constexpr (H2) gm[r & 1] = fmaxf(gm[r & 1], fabsf(gzv));
        },
        [&](int tj, int ti, int r, int col, int rowc, int row) {
          const float v = acc[ti][tj][r] * INV;
          float gzv;
          if (col < ksplit) {
            gzv = v * aD.v[ti][tj][r];
          } else {
            if (col < ksplit + g.pe) GE[row * FEP + (col - ksplit)] = v;   // skip connection: straight to g_e
            gzv = 0.f;
          }
          Y[row * FP + col] = gzv * SG;
          bstore(rg, (unsigned)(4 * h * FH + col) * 4u, rowc * FH * 4, gzv);
          if constexpr (H2) gm[r & 1] = fmaxf(gm[r & 1], fabsf(gzv));
        });
    if constexpr (H2) h2_raise_flag(fmaxf(gm[0], gm[1]), &ovf[l & 1], lane_e);
    lds_barrier();
    if constexpr (H2) {
      sg = SG;
      isg = 1.f / SG;
      if (h2_flag_up(&ovf[l & 1])) {   // (workgroup-uniform; a Jacobian row beyond 256: this tile carries a smaller scale)
        const float m = wave_max(fmaxf(gm[0], gm[1]));
        if (lane_e == 0) wmx[wave] = m;
        lds_barrier();
        const float tm = tile_max<NW>(wmx);
        if (g.smax != nullptr && tid == 0) amax_tile_commit(g.smax + SMAX_GZ + l - 1, tm);
        if (tid == 0) ovf[l & 1] = 0;
        x2h_dyn_scale(__builtin_bit_cast(unsigned, tm), sg, isg);
        const float f = sg * (1.f / SG);
        for_each_acc<TI, TJ>(n0, lane_e, [&](int tj, int ti, int r, int col, int rowc, int row) { Y[row * FP + col] *= f; });
        lds_barrier();
      }
    }
    if constexpr (NBUF == 2) { float* t = X; X = Y; Y = t; }
  }
  // layer 0: g_e += gz_0 W_0 (Ep = 64 columns: the waves that own columns 0..63)
  if (n0 < 64) {
    [[maybe_unused]] const float INV = H2 ? isg * h2_iws_at(iwsv, 0) : 1.f;
    layer_mma<TI, TJ, X3>(X, g, g.wT_off[0], FH, n0, lane, acc, mm, -1);
    for_each_acc<TI, TJ>(n0, lane, [&](int tj, int ti, int r, int col, int rowc, int row) {
      if (col < g.pe) GE[row * FEP + col] += acc[ti][tj][r] * INV;
    });
  }
  __syncthreads();
  // normal = J_pe(x)^T g_e
  if (tid < BT) {
    const int64_t row = row0 + tid;
    const float* ge = GE + tid * FEP;
    float n[3] = {ge[0], ge[1], ge[2]};
    float f = 1.f;
    int c = 3;
    for (int k = 0; k < g.multires; ++k) {
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        float s, co;
        sincosf(g.x4[row * 4 + d] * f, &s, &co);
        n[d] += f * (ge[c + d] * co - ge[c + 3 + d] * s);
      }
      c += 6;
      f *= 2.f;
    }
    g.nrm[row * 4] = n[0]; g.nrm[row * 4 + 1] = n[1]; g.nrm[row * 4 + 2] = n[2]; g.nrm[row * 4 + 3] = 0.f;
  }
}

// ---------------------------------------------------------------------------------------------------------
// RA sweep
// ---------------------------------------------------------------------------------------------------------
template <int TI, int NW = 4, bool X3 = false>
__global__ __launch_bounds__(64 * NW, (NW == 8 && TI == 2) ? 1 : 2) void fused_ra_kernel(FusedBwdArgs g) {
  constexpr int BT = 32 * TI;
  constexpr int NT = 64 * NW;   // threads
  constexpr int TJ = 8 / NW;    // 32-column tiles per wave
  // both operand tiles of an epilogue fit next to the weight fragments unless the wave owns 64 x 64 outputs
  [[maybe_unused]] constexpr bool BOTH_IN_LOOP = !(TI == 2 && TJ == 2);
  constexpr int NBUF = TI == 1 ? 2 : 1;
  __shared__ __attribute__((aligned(16))) float lds[NBUF * BT * FP + BT * FEP];
  float* X = lds;
  float* Y = lds + (NBUF - 1) * BT * FP;
  float* E = lds + NBUF * BT * FP;   // adjoint of g_e of the tile (re-enters at the skip connection)
  __shared__ float wmx[2][8];        // per-wave maxima of u_{l+1} by layer parity (x2h weight-gradient scales)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
  const int64_t row0 = (int64_t)blockIdx.x * BT;
  const int n0 = wave * 32 * TJ;

  float gm = 0.f;   // max |geb| of the tile (its padding rows are written as zeros by nbar_geb_kernel: no mask)
  for (int idx = tid; idx < BT * g.Ep; idx += NT) {
    const int r = idx / g.Ep, c = idx - r * g.Ep;
    const float v = g.geb[(row0 + r) * g.Ep + c];
    X[r * FP + c] = v;
    if (c < FEP) E[r * FEP + c] = v;
    gm = fmaxf(gm, fabsf(v));
  }
  if (g.amax != nullptr) {   // u_0 = geb: the scale of layer 0's weight-gradient job (x2h); one conditional atomic per tile
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) gm = fmaxf(gm, __shfl_xor(gm, o, 64));
    if (lane == 0) wmx[1][wave] = gm;
  }
  __syncthreads();
  if (g.amax != nullptr && tid == 0) {
    float m = wmx[1][0];
    for (int w = 1; w < NW; ++w) m = fmaxf(m, wmx[1][w]);
    amax_tile_commit(g.amax + AMAX_U, m);
  }

  v16f acc[TI][TJ];
  AuxTile<TI, TJ> aD, aG;
  [[maybe_unused]] X3Mma<TI, TJ> mm;
  if constexpr (X3 && !(TI == 2 && TJ == 2)) mm.request(g.w3 + 3 * g.w_off[0], g.Kp[0], n0, lane);
  for (int l = 0; l < g.nh; ++l) {
    // TI == 2: only one operand tile fits next to the weight fragments during the matrix loop; the second
    // one is requested right after it, into the registers the weight fragments leave behind
    layer_mma<TI, TJ, X3>(X, g, g.w_off[l], g.Kp[l], n0, lane, acc, mm, l + 1 < g.nh ? g.w_off[l + 1] : -1,   // gzb = u_l W_l^T
                     [&]() {
                       if constexpr (!(X3 && !BOTH_IN_LOOP)) prefetch_tile<TI, TJ>(g.D[l], row0, n0, lane, aD);
                       if constexpr (BOTH_IN_LOOP) prefetch_tile<TI, TJ>(g.gz[l], row0, n0, lane, aG);
                     });
    const int lane_e = opaque_lane(lane);   // (see fused_reverse_kernel)
    const int h = lane_e >> 5;
    if constexpr (X3 && !BOTH_IN_LOOP) prefetch_tile<TI, TJ>(g.D[l], row0, n0, lane_e, aD);   // (x3: both after the loop)
    if constexpr (!BOTH_IN_LOOP) prefetch_tile<TI, TJ>(g.gz[l], row0, n0, lane_e, aG);
    if constexpr (NBUF == 1) lds_barrier();
    const int n_real = g.n_real[l];
    const bool pe_tail = (l + 1 == g.skip);
    const BufRsrc rzR = tile_rsrc(g.zR[l] + (size_t)row0 * FH, BT * FH * 4);
    if (l + 1 == g.nh && g.ucol != nullptr) {
      // last layer: u_nh feeds nothing but the column sums of the sdf-head row gradient.  Every wave owns its columns
      // for all rows of the tile: sum the lane's 16 TI values per column tile, fold the two lane halves, one store per
      // column — 64 MB less to write here and to read there.  (No tile is written to LDS: nothing follows.)
      float cs[TJ];
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj) cs[tj] = 0.f;
      for_each_acc<TI, TJ>(n0, lane_e, [&](int tj, int ti, int r, int col, int rowc, int row) {
        const float v = acc[ti][tj][r];
        const float un = col < n_real ? v * aD.v[ti][tj][r] : 0.f;
        const float zr = col < n_real ? ((v - un) * aG.v[ti][tj][r]) * 100.f : 0.f;
        bstore(rzR, (unsigned)(4 * h * FH + col) * 4u, rowc * FH * 4, zr);
        cs[tj] += (row0 + row < g.M) ? un : 0.f;     // padding rows hold whatever the workspace held
      });
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj) {
        const float tot = cs[tj] + __shfl_xor(cs[tj], 32, 64);
        if (lane_e < 32) g.ucol[(size_t)blockIdx.x * FH + n0 + tj * 32 + lane_e] = tot;
      }
      break;
    }
    const BufRsrc ru = tile_rsrc(g.u[l + 1] + (size_t)row0 * FH, BT * FH * 4);
    // max |u_{l+1}| rides in the epilogue.  No row mask: the padding rows of geb are written as zeros (nbar_geb_kernel) and D,
    // gz of padding rows are the finite state of the point 0 the forward evaluated there, so every u of a padding row is 0.
    float um[2] = {0.f, 0.f};
    for_each_acc_split<TI, TJ>(
        n0, lane_e, n_real,
        [&](int tj, int ti, int r, int col, int rowc, int row) {
          const unsigned voff = (unsigned)(4 * h * FH + col) * 4u;
          const unsigned soff = rowc * FH * 4;
          const float v = acc[ti][tj][r];
          const float un = v * aD.v[ti][tj][r];
          const float zr = ((v - un) * aG.v[ti][tj][r]) * 100.f;   // 100 v gz (1 - D)
          Y[row * FP + col] = un;
          bstore(rzR, voff, soff, zr);
          bstore(ru, voff, soff, un);
          um[r & 1] = fmaxf(um[r & 1], fabsf(un));
        },
        [&](int tj, int ti, int r, int col, int rowc, int row) {
          const unsigned voff = (unsigned)(4 * h * FH + col) * 4u;
          const unsigned soff = rowc * FH * 4;
          const float v = acc[ti][tj][r];
          float zr, un;
          if (col < n_real) {
            un = v * aD.v[ti][tj][r];
            zr = ((v - un) * aG.v[ti][tj][r]) * 100.f;
          } else {
            zr = 0.f;
            un = (pe_tail && col < n_real + g.pe) ? E[row * FEP + (col - n_real)] : 0.f;
          }
          Y[row * FP + col] = un;
          bstore(rzR, voff, soff, zr);
          bstore(ru, voff, soff, un);
          um[r & 1] = fmaxf(um[r & 1], fabsf(un));
        });
    if (g.amax != nullptr) {   // max |u_{l+1}| of the real rows: the scale of its weight-gradient job (x2h).  The waves'
      // maxima meet in LDS behind the barrier that ends the layer: one atomic per tile, and only when the slot would grow
      float m = fmaxf(um[0], um[1]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
      if (lane_e == 0) wmx[l & 1][wave] = m;
    }
    lds_barrier();
    if (g.amax != nullptr && tid == 0) {
      float m = wmx[l & 1][0];
      for (int w = 1; w < NW; ++w) m = fmaxf(m, wmx[l & 1][w]);
      amax_tile_commit(g.amax + AMAX_U + l + 1, m);
    }
    if constexpr (NBUF == 2) { float* t = X; X = Y; Y = t; }
  }
}

// ---------------------------------------------------------------------------------------------------------
// FB sweep
// ---------------------------------------------------------------------------------------------------------
template <int TI, int NW = 4, bool X3 = false>
__global__ __launch_bounds__(64 * NW, (NW == 8 && TI == 2) ? 1 : 2) void fused_fb_kernel(FusedBwdArgs g) {
  constexpr int BT = 32 * TI;
  constexpr int NT = 64 * NW;   // threads
  constexpr int TJ = 8 / NW;    // 32-column tiles per wave
  // both operand tiles of an epilogue fit next to the weight fragments unless the wave owns 64 x 64 outputs
  [[maybe_unused]] constexpr bool BOTH_IN_LOOP = !(TI == 2 && TJ == 2);
  constexpr int NBUF = TI == 1 ? 2 : 1;
  __shared__ __attribute__((aligned(16))) float lds[NBUF * BT * FP];
  float* X = lds;
  float* Y = lds + (NBUF - 1) * BT * FP;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
  const int64_t row0 = (int64_t)blockIdx.x * BT;
  const int n0 = wave * 32 * TJ;

  v16f acc[TI][TJ];
  AuxTile<TI, TJ> aD, aZ;
  [[maybe_unused]] X3Mma<TI, TJ> mm;
  if constexpr (X3 && !(TI == 2 && TJ == 2)) {
    if (g.fbar != nullptr) mm.request(g.w3 + 3 * g.wfT_off, FH, n0, lane);
    else if (g.nh > 1) mm.request(g.w3 + 3 * g.wT_off[g.nh - 1], FH, n0, lane);
  }
  constexpr bool LATE = X3 && TI == 2 && TJ == 2;   // operand tiles requested after the matrix loop (see fused_reverse_kernel)
  if constexpr (!LATE) {
    prefetch_tile<TI, TJ>(g.D[g.nh - 1], row0, n0, lane, aD);
    prefetch_tile<TI, TJ>(g.zR[g.nh - 1], row0, n0, lane, aZ);
  }
  for (int ti_ = 0; ti_ < TI; ++ti_) for (int tj_ = 0; tj_ < TJ; ++tj_) for (int r_ = 0; r_ < 16; ++r_) acc[ti_][tj_][r_] = 0.f;
  if (g.fbar != nullptr) {   // ab_{nh-1} = fbar W_feat (+ the sdf-head term below)
    const float* fb = g.fbar + (size_t)row0 * g.ld_fbar;
    for (int idx = tid; idx < BT * FH / 4; idx += NT) {
      const int r = idx >> 6, c4 = idx & 63;
      *reinterpret_cast<vf4*>(X + r * FP + c4 * 4) = *reinterpret_cast<const vf4*>(fb + (size_t)r * g.ld_fbar + c4 * 4);
    }
    __syncthreads();
    layer_mma<TI, TJ, X3>(X, g, g.wfT_off, FH, n0, lane, acc, mm, g.nh > 1 ? g.wT_off[g.nh - 1] : -1);
    if constexpr (NBUF == 1) lds_barrier();
  }
  if constexpr (LATE) {
    prefetch_tile<TI, TJ>(g.D[g.nh - 1], row0, n0, lane, aD);
    prefetch_tile<TI, TJ>(g.zR[g.nh - 1], row0, n0, lane, aZ);
  }
  for (int l = g.nh - 1; l >= 0; --l) {
    // epilogue of the product that produced ab_l: zb_l = ab_l * D_l + zR_l
    const int lane_e = opaque_lane(lane);   // (see fused_reverse_kernel)
    const int h = lane_e >> 5;
    const int n_real = g.n_real[l];
    const bool head = (l == g.nh - 1);
    const BufRsrc rzb = tile_rsrc(g.zb[l] + (size_t)row0 * FH, BT * FH * 4);
    if (head) {   // + sbar / scale * w_sdf  (the sdf head's contribution to ab_{nh-1}); once per launch
      for_each_acc<TI, TJ>(n0, lane_e, [&](int tj, int ti, int r, int col, int rowc, int row) {
        acc[ti][tj][r] = fmaf(g.sbar[row0 + row] * g.inv_scale, g.packed[g.wsdf_off + col], acc[ti][tj][r]);
      });
    }
    for_each_acc_split<TI, TJ>(
        n0, lane_e, n_real,
        [&](int tj, int ti, int r, int col, int rowc, int row) {
          const float zb = fmaf(acc[ti][tj][r], aD.v[ti][tj][r], aZ.v[ti][tj][r]);
          Y[row * FP + col] = zb;
          bstore(rzb, (unsigned)(4 * h * FH + col) * 4u, rowc * FH * 4, zb);
          acc[ti][tj][r] = zb;   // (kept for the maximum below)
        },
        [&](int tj, int ti, int r, int col, int rowc, int row) {
          const float zb = col < n_real ? fmaf(acc[ti][tj][r], aD.v[ti][tj][r], aZ.v[ti][tj][r]) : 0.f;
          Y[row * FP + col] = zb;
          bstore(rzb, (unsigned)(4 * h * FH + col) * 4u, rowc * FH * 4, zb);
          acc[ti][tj][r] = zb;
        });
    if (g.amax != nullptr) {   // max |zb_l| of the real rows: the scale of its weight-gradient job (x2h)
      const long long left = g.M - row0;
      amax_commit(g.amax + AMAX_ZB + l, acc_absmax<TI, TJ>(acc, lane_e, left >= BT ? BT : (int)(left < 0 ? 0 : left)), lane_e);
    }
    if (l == 0) break;
    lds_barrier();
    if constexpr (NBUF == 2) { float* t = X; X = Y; Y = t; }
    layer_mma<TI, TJ, X3>(X, g, g.wT_off[l], FH, n0, lane, acc, mm, l > 1 ? g.wT_off[l - 1] : -1,   // ab_{l-1} = zb_l W_l
                     [&]() {
                       if constexpr (!(X3 && !BOTH_IN_LOOP)) prefetch_tile<TI, TJ>(g.D[l - 1], row0, n0, lane, aD);
                       if constexpr (BOTH_IN_LOOP) prefetch_tile<TI, TJ>(g.zR[l - 1], row0, n0, lane, aZ);
                     });
    if constexpr (X3 && !BOTH_IN_LOOP) prefetch_tile<TI, TJ>(g.D[l - 1], row0, n0, opaque_lane(lane), aD);
    if constexpr (!BOTH_IN_LOOP) prefetch_tile<TI, TJ>(g.zR[l - 1], row0, n0, opaque_lane(lane), aZ);
    if constexpr (NBUF == 1) lds_barrier();   // every wave has finished reading the tile
  }
}

// ---------------------------------------------------------------------------------------------------------
// x2h forms of RA and FB (64-point tiles updated in place): the products as three fp16 terms (gemm.hip.h).  The A operand
// is a loss adjoint, whose magnitude nothing bounds a priori: the tile in LDS carries a power-of-two scale chosen PER TILE
// AND LAYER from the actual maximum of the values about to be stored (every wave leaves the maximum of its block in LDS
// before the barrier that already separates the matrix loop from the in-place update, so the exchange costs no barrier) —
// the largest element of a tile always lands in [2^13, 2^14): no overflow whatever the loss scale, and every element
// within 2^-16 of its tile's maximum keeps the full two-plane precision.  The same maxima (real rows only), folded
// over the tiles with one atomic per tile, are the scales of the weight-gradient jobs (PointBufs::amax).
// ---------------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void fused_fb_h2_kernel(FusedBwdArgs g) {
  constexpr int TI = 2, BT = 64, NT = 64 * NW, TJ = 8 / NW;
  __shared__ __attribute__((aligned(16))) float lds[BT * FP];
  __shared__ float wm[2][8];   // per-wave maxima over the real rows, by layer parity (padding rows hold workspace garbage:
                               // they may overflow the scaled tile — their products are never read)
  float* X = lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
  const int64_t row0 = (int64_t)blockIdx.x * BT;
  const int n0 = wave * 32 * TJ;
  const long long left = g.M - row0;
  const int rows_ok = left >= BT ? BT : (int)(left < 0 ? 0 : left);

  v16f acc[TI][TJ];
  AuxTile<TI, TJ> aD, aZ;
  X3Mma<TI, TJ, 2> mm;
  constexpr bool LATE = TJ == 2;   // 64 x 64-output waves: operand tiles requested after the matrix loop
  if constexpr (!LATE) {
    if (g.fbar != nullptr) mm.request(g.w3 + 2 * g.wfT_off, FH, n0, lane);
    else if (g.nh > 1) mm.request(g.w3 + 2 * g.wT_off[g.nh - 1], FH, n0, lane);
    prefetch_tile<TI, TJ>(g.D[g.nh - 1], row0, n0, lane, aD);
    prefetch_tile<TI, TJ>(g.zR[g.nh - 1], row0, n0, lane, aZ);
  }
  for (int ti_ = 0; ti_ < TI; ++ti_) for (int tj_ = 0; tj_ < TJ; ++tj_) for (int r_ = 0; r_ < 16; ++r_) acc[ti_][tj_][r_] = 0.f;
  float unscale = 1.f;   // accumulator -> ab (true units)
  const float iwsv = h2_iws_load(g.h2tab, lane);
  if (g.fbar != nullptr) {   // ab_{nh-1} = fbar W_feat (+ the sdf-head term below)
    const float* fb = g.fbar + (size_t)row0 * g.ld_fbar;
    float m = 0.f;
    for (int idx = tid; idx < BT * FH / 4; idx += NT) {
      const int r = idx >> 6, c4 = idx & 63;
      const vf4 v = *reinterpret_cast<const vf4*>(fb + (size_t)r * g.ld_fbar + c4 * 4);
      *reinterpret_cast<vf4*>(X + r * FP + c4 * 4) = v;
      if (r < rows_ok) m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    m = wave_max(m);
    if (lane == 0) wm[1][wave] = m;
    __syncthreads();
    float s, inv;
    tile_scale<NW>(wm[1], s, inv);
    for (int idx = tid; idx < BT * FH / 4; idx += NT) {   // every thread rescales the elements it wrote
      const int r = idx >> 6, c4 = idx & 63;
      vf4* q = reinterpret_cast<vf4*>(X + r * FP + c4 * 4);
      *q = *q * s;
    }
    __syncthreads();
    if constexpr (LATE) mm.request(g.w3 + 2 * g.wfT_off, FH, n0, lane);
    mm.run(X, g.w3 + 2 * g.wfT_off, FH, n0, lane, acc, (!LATE && g.nh > 1) ? g.w3 + 2 * g.wT_off[g.nh - 1] : nullptr, FH, n0);
    unscale = inv * h2_iws_at(iwsv, g.nh);   // (tile scale x the feature head's scale in the mirror)
  }
  if constexpr (LATE) {
    prefetch_tile<TI, TJ>(g.D[g.nh - 1], row0, n0, lane, aD);
    prefetch_tile<TI, TJ>(g.zR[g.nh - 1], row0, n0, lane, aZ);
  }
  for (int l = g.nh - 1; l >= 0; --l) {
    // phase 1 (registers and HBM only): zb_l = ab_l * D_l + zR_l, its maxima
    const int lane_e = opaque_lane(lane);
    const int h = lane_e >> 5;
    const int n_real = g.n_real[l];
    const bool head = (l == g.nh - 1);
    const int par = l & 1;
    const BufRsrc rzb = tile_rsrc(g.zb[l] + (size_t)row0 * FH, BT * FH * 4);
    if (head) {   // + sbar / scale * w_sdf  (the sdf head's contribution to ab_{nh-1}); once per launch
      for_each_acc<TI, TJ>(n0, lane_e, [&](int tj, int ti, int r, int col, int rowc, int row) {
        acc[ti][tj][r] = fmaf(g.sbar[row0 + row] * g.inv_scale, g.packed[g.wsdf_off + col], acc[ti][tj][r] * unscale);
      });
      unscale = 1.f;
    }
    for_each_acc_split<TI, TJ>(
        n0, lane_e, n_real,
        [&](int tj, int ti, int r, int col, int rowc, int row) {
          const float zb = fmaf(acc[ti][tj][r] * unscale, aD.v[ti][tj][r], aZ.v[ti][tj][r]);
          bstore(rzb, (unsigned)(4 * h * FH + col) * 4u, rowc * FH * 4, zb);
          acc[ti][tj][r] = zb;
        },
        [&](int tj, int ti, int r, int col, int rowc, int row) {
          const float zb = col < n_real ? fmaf(acc[ti][tj][r] * unscale, aD.v[ti][tj][r], aZ.v[ti][tj][r]) : 0.f;
          bstore(rzb, (unsigned)(4 * h * FH + col) * 4u, rowc * FH * 4, zb);
          acc[ti][tj][r] = zb;
        });
    {
      const float mr = wave_max(acc_absmax<TI, TJ>(acc, lane_e, rows_ok));
      if (lane_e == 0) wm[par][wave] = mr;
    }
    lds_barrier();   // every wave has finished reading the tile; the maxima are visible
    float s, inv;
    const float tmax = tile_scale<NW>(wm[par], s, inv);
    if (tid == 0 && g.amax != nullptr) amax_tile_commit(g.amax + AMAX_ZB + l, tmax);
    if (l == 0) break;
    // phase 2: the tile for the next product, scaled by this layer's own maximum
    for_each_acc<TI, TJ>(n0, lane_e, [&](int tj, int ti, int r, int col, int rowc, int row) {
      X[row * FP + col] = acc[ti][tj][r] * s;
    });
    lds_barrier();
    if constexpr (LATE) {
      mm.request(g.w3 + 2 * g.wT_off[l], FH, n0, lane);
      mm.run(X, g.w3 + 2 * g.wT_off[l], FH, n0, lane, acc, nullptr, 0, 0);   // ab_{l-1} = zb_l W_l
      prefetch_tile<TI, TJ>(g.D[l - 1], row0, n0, opaque_lane(lane), aD);
      prefetch_tile<TI, TJ>(g.zR[l - 1], row0, n0, opaque_lane(lane), aZ);
    } else {
      mm.run(X, g.w3 + 2 * g.wT_off[l], FH, n0, lane, acc, l > 1 ? g.w3 + 2 * g.wT_off[l - 1] : nullptr, FH, n0, [&]() {
        prefetch_tile<TI, TJ>(g.D[l - 1], row0, n0, lane, aD);
        prefetch_tile<TI, TJ>(g.zR[l - 1], row0, n0, lane, aZ);
      });
    }
    unscale = inv * h2_iws_at(iwsv, l);
  }
}

// ---------------------------------------------------------------------------------------------------------
static void fill_args(const Layout& L, const float* packed, PointBufs& pb, FusedBwdArgs& g) {
  memset(&g, 0, sizeof(g));
  g.packed = packed;
  g.w3 = reinterpret_cast<const x3raw*>(packed + L.total);
  g.nh = L.nh;
  g.skip = L.skip;
  g.pe = L.pe;
  g.multires = L.multires;
  g.Ep = L.Ep;
  g.inv_scale = 1.f / L.sdf_scale;
  for (int l = 0; l < L.nh; ++l) {
    g.n_real[l] = L.hid[l].N;
    g.Kp[l] = L.hid[l].Kp;
    g.w_off[l] = L.hid[l].w_off;
    g.wT_off[l] = L.hid[l].wT_off;
    g.D[l] = pb.D[l];
    g.gz[l] = pb.gz[l];
    g.zR[l] = pb.zR[l];
    g.zb[l] = pb.zb[l];
  }
  for (int l = 1; l <= L.nh; ++l) g.u[l] = pb.u[l];
  g.wsdf_off = L.wsdf_off;
  g.wfT_off = L.feat.wT_off;
  g.M = pb.M;
  g.ucol = nullptr;
  g.x4 = pb.x;
  g.nrm = pb.nrm;
  g.geb = pb.geb;
  g.sbar = pb.sbar;
  g.amax = is_x2h(L) ? pb.amax : nullptr;
  g.h2tab = is_x2h(L) ? h2_tab(L, packed) : nullptr;
  g.smax = nullptr;
}

// Variant of the three sweeps: tile height TI (32 / 64 points) x waves per workgroup NW (4: 64 columns per wave;
// 8: 32 columns per wave).  Measured on 65,536 points (us; R / RA / FB):
//   TI=1 NW=4: 651 / 717 / 700      two workgroups per CU, 2 waves per SIMD
//   TI=2 NW=4: 628 / 693 / 801      64-point tiles in place; 49-65 spilled registers, FB's 2nd tile exposed
//   TI=2 NW=8: 618 / 774 / 702      one workgroup per CU, no spills
//   TI=1 NW=8: 603 / 671 / 669      two workgroups per CU, 4 waves per SIMD (<= 128 registers)   <- default
// More resident waves beat larger tiles: what limits these kernels is waiting (operand tiles from HBM, weight
// fragments from L2), which only other waves' MFMAs can fill.  RNB_VARIANT_BWD_TI / _NW (rnb_model_desc.variant) override all three.
// x3 (the default arithmetic) runs TI=2 NW=4: 436 / 536 / 480 us against 503 / 602 / 555 (TI=1 NW=4), 483 / 619 / 666
// (TI=1 NW=8) and 493 / 630 / 590 (TI=2 NW=8).  64 x 64 outputs per wave halve both the weight bytes pulled from L2 per
// point (32-point tiles ran at the CU's 64 B/clk fill rate, the matrix pipe 28-34 % busy) and the split work per MFMA;
// the operand tiles of the epilogue are requested after the matrix loop, into the registers it frees, and travel while
// the CU's other workgroup multiplies.
static int bwd_nw(const Layout& L, int dflt) {
  const int v = L.knob(RNB_VARIANT_BWD_NW_SHIFT);
  return v == 1 ? 4 : v == 2 ? 8 : dflt;
}

static int bwd_ti(const Layout& L, int dflt) {
  const int v = L.knob(RNB_VARIANT_BWD_TI_SHIFT);
  return v == 1 || v == 2 ? v : dflt;
}

static double hidden_flops(const Layout& L, int64_t M, int first) {
  double fl = 0;
  for (int l = first; l < L.nh; ++l) fl += 2.0 * (double)M * L.hid[l].N * L.hid[l].K;
  return fl;
}

int fused_reverse(const Layout& L, const float* packed, PointBufs& pb, hipStream_t s) {
  FusedBwdArgs g;
  fill_args(L, packed, pb, g);
  ProfScope prof(hidden_flops(L, pb.M, 0), s, "R_sweep");
  const int ti = bwd_ti(L, is_x3(L) ? 2 : 1), nw = bwd_nw(L, is_x3(L) ? 4 : 8);
  const dim3 grid((unsigned)(pb.Mp / (32 * ti))), block(64 * nw);
  if (is_x2h(L)) {
    g.w3 = x2h_mirror(L, packed);
    g.smax = pb.smax;   // (nullptr outside a render forward)
    if (ti == 2 && nw == 8) hipLaunchKernelGGL((fused_reverse_kernel<2, 8, true, true>), grid, block, 0, s, g);
    else if (ti == 2) hipLaunchKernelGGL((fused_reverse_kernel<2, 4, true, true>), grid, block, 0, s, g);
    else if (nw == 8) hipLaunchKernelGGL((fused_reverse_kernel<1, 8, true, true>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((fused_reverse_kernel<1, 4, true, true>), grid, block, 0, s, g);
  } else if (is_x3(L)) {
    if (ti == 2 && nw == 8) hipLaunchKernelGGL((fused_reverse_kernel<2, 8, true>), grid, block, 0, s, g);
    else if (ti == 2) hipLaunchKernelGGL((fused_reverse_kernel<2, 4, true>), grid, block, 0, s, g);
    else if (nw == 8) hipLaunchKernelGGL((fused_reverse_kernel<1, 8, true>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((fused_reverse_kernel<1, 4, true>), grid, block, 0, s, g);
  } else if (ti == 2 && nw == 8) hipLaunchKernelGGL((fused_reverse_kernel<2, 8>), grid, block, 0, s, g);
  else if (ti == 2) hipLaunchKernelGGL(fused_reverse_kernel<2>, grid, block, 0, s, g);
  else if (nw == 8) hipLaunchKernelGGL((fused_reverse_kernel<1, 8>), grid, block, 0, s, g);
  else hipLaunchKernelGGL(fused_reverse_kernel<1>, grid, block, 0, s, g);
  RNB_CHECK_LAUNCH();
  return RNB_OK;
}

// u_tiles != nullptr: the last layer leaves per-tile column sums of u_nh in pb.u[nh] ([tiles][256], *u_tiles tiles)
// instead of the matrix (see FusedBwdArgs::ucol)
int fused_ra(const Layout& L, const float* packed, PointBufs& pb, hipStream_t s, int* u_tiles) {
  FusedBwdArgs g;
  fill_args(L, packed, pb, g);
  ProfScope prof(hidden_flops(L, pb.M, 0), s, "RA_sweep");
  const int ti = bwd_ti(L, is_x3(L) ? 2 : 1), nw = bwd_nw(L, is_x3(L) ? 4 : 8);
  if (u_tiles != nullptr) {
    g.ucol = pb.u[L.nh];
    *u_tiles = (int)(pb.Mp / (32 * ti));
  }
  const dim3 grid((unsigned)(pb.Mp / (32 * ti))), block(64 * nw);
  // RA stays on the six bf16 terms: it reads D_l, gz_l and writes zR_l, u_{l+1} — 2.1 GB per 65,536 points, 0.42 ms at
  // 5 TB/s against 0.45 ms measured: the matrix time hides under the state traffic.  (An x2h form with FB's two-phase
  // epilogue was built and measured in round 4 — 0.505 / 0.545 ms against 0.475, profiles/r04_ab_experiments.txt — and removed.)
  if (is_x3(L)) {
    if (ti == 2 && nw == 8) hipLaunchKernelGGL((fused_ra_kernel<2, 8, true>), grid, block, 0, s, g);
    else if (ti == 2) hipLaunchKernelGGL((fused_ra_kernel<2, 4, true>), grid, block, 0, s, g);
    else if (nw == 8) hipLaunchKernelGGL((fused_ra_kernel<1, 8, true>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((fused_ra_kernel<1, 4, true>), grid, block, 0, s, g);
  } else if (ti == 2 && nw == 8) hipLaunchKernelGGL((fused_ra_kernel<2, 8>), grid, block, 0, s, g);
  else if (ti == 2) hipLaunchKernelGGL(fused_ra_kernel<2>, grid, block, 0, s, g);
  else if (nw == 8) hipLaunchKernelGGL((fused_ra_kernel<1, 8>), grid, block, 0, s, g);
  else hipLaunchKernelGGL(fused_ra_kernel<1>, grid, block, 0, s, g);
  RNB_CHECK_LAUNCH();
  return RNB_OK;
}

int fused_fb(const Layout& L, const float* packed, PointBufs& pb, bool with_color, hipStream_t s) {
  FusedBwdArgs g;
  fill_args(L, packed, pb, g);
  g.fbar = with_color ? pb.cinb : nullptr;
  g.ld_fbar = L.Cinp;
  ProfScope prof(hidden_flops(L, pb.M, 1) + (with_color ? 2.0 * (double)pb.M * L.F * L.H : 0.0), s, "FB_sweep");
  const int ti = bwd_ti(L, is_x3(L) ? 2 : 1), nw = bwd_nw(L, is_x3(L) ? 4 : 8);
  const dim3 grid((unsigned)(pb.Mp / (32 * ti))), block(64 * nw);
  if (is_x2h(L) && ti == 2) {   // three fp16 terms, per-tile scales (64-point tiles only: the in-place form)
    g.w3 = x2h_mirror(L, packed);
    if (nw == 8) hipLaunchKernelGGL((fused_fb_h2_kernel<8>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((fused_fb_h2_kernel<4>), grid, block, 0, s, g);
  } else if (is_x3(L)) {
    if (ti == 2 && nw == 8) hipLaunchKernelGGL((fused_fb_kernel<2, 8, true>), grid, block, 0, s, g);
    else if (ti == 2) hipLaunchKernelGGL((fused_fb_kernel<2, 4, true>), grid, block, 0, s, g);
    else if (nw == 8) hipLaunchKernelGGL((fused_fb_kernel<1, 8, true>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((fused_fb_kernel<1, 4, true>), grid, block, 0, s, g);
  } else if (ti == 2 && nw == 8) hipLaunchKernelGGL((fused_fb_kernel<2, 8>), grid, block, 0, s, g);
  else if (ti == 2) hipLaunchKernelGGL(fused_fb_kernel<2>, grid, block, 0, s, g);
  else if (nw == 8) hipLaunchKernelGGL((fused_fb_kernel<1, 8>), grid, block, 0, s, g);
  else hipLaunchKernelGGL(fused_fb_kernel<1>, grid, block, 0, s, g);
  RNB_CHECK_LAUNCH();
  return RNB_OK;
}

}  // namespace rnb
