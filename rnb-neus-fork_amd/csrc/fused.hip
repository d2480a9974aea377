// Fused SDF-network sweeps for the shipped network shape (hidden width 256): one workgroup carries a tile
// of 64 (or 32) points through ALL layers.  Activations stay in LDS between layers, weights stream from L2
// straight into MFMA B-fragments (each 128-byte weight line is fetched once per workgroup and consumed by
// four back-to-back 16-byte loads), and what the backward pass needs is written to HBM with fire-and-forget
// stores that overlap the next layer's matrix work.  Replaces, per sweep, the chain of per-layer GEMM
// launches of mlp.hip (which remain the generic path for other widths).
//
//   fused_forward_kernel   positional encoding + F sweep (+ sdf head, + feature head)
//                          models/embedder.py:40-46, models/fields.py:82-104
#include <type_traits>

#include "fused_common.hip.h"

// A/B switches (compile time): scalar instead of packed fp32 math in the forward epilogue (same operations element by element:
// the same bits).  Six bf16 terms (round 2, one box, two runs each): packed 3.694 / 3.684 ms per step, scalar 3.71 / 3.78.
// Three fp16 terms (round 5, profiles/r05_ab_experiments.txt §11): scalar is the faster one, F(save) 0.323 / 0.324 against
// 0.335 / 0.330 ms — a v_pk_*_f32 holds the vector port for two passes, saves no port time over the two instructions it
// replaces, and cannot take the |x| modifier (two v_or per pair instead); with half as many MFMAs per layer the epilogue's
// port time is what a layer's cycle is made of (§11: stamps).
#ifndef RNB_X3_SCALAR_EPI
#define RNB_X3_SCALAR_EPI 0
#endif
#ifndef RNB_H2_SCALAR_EPI
#define RNB_H2_SCALAR_EPI 1
#endif
// A/B switches (compile time, tools/build_variant.sh): what the x2h range guard of the forward sweep costs.
// 1: no maximum tracking and no flag (the tile is assumed in range: round 4's behaviour); 2: tracking, but the flag is not read
#ifndef RNB_H2_GUARD_AB
#define RNB_H2_GUARD_AB 0
#endif

namespace rnb {


// TI = row tiles per workgroup (64 points for TI = 2; 32 points for TI = 1, used for small batches so that
// every CU still gets a workgroup).  NW = waves per workgroup: 4 (each wave 64 output columns) or 8 (32 columns
// each) — the latter for batches so small that a CU holds a single workgroup: two waves per SIMD instead of one
// hide each other's LDS / L2 waits.
// H2 (with X3): the products on the fp16 matrix pipe in three terms instead of six bf16 ones (gemm.hip.h, "x2h"): the
// LDS tile then holds the activations times SA, the mirror the weights times kH2WScale, and the accumulators come out
// scaled by the product — undone by the fma that adds the bias.
template <int TI, bool SAVE, int NW = 4, bool X3 = false, bool H2 = false>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void fused_forward_kernel(FusedFwdArgs g) {
  static_assert(!H2 || X3, "x2h is a form of the split-operand path");
  constexpr float SA = H2 ? kH2ActScale : 1.f;                         // what a writer multiplies by (see kH2ActLimit)
  constexpr int WP = H2 ? 2 : 3;                                       // planes of the weight mirror
  constexpr int FT = 32 * TI;
  constexpr int NT = 64 * NW;     // threads
  constexpr int TJ = 8 / NW;      // 32-column tiles per wave
  // TI == 1: two activation tiles (a layer reads one, writes the other: one barrier per layer);
  // TI == 2: one tile updated in place behind a second barrier (two tiles would not leave room for two
  // workgroups per CU)
  constexpr int NBUF = TI == 1 ? 2 : 1;
  __shared__ __attribute__((aligned(16))) float lds[NBUF * FT * FP + FT * FEP];
  __shared__ float wmx[8];        // x2h, rare path: the waves' maxima of the values just written
  __shared__ int ovf[2];          // x2h: "a value of the tile just written reached kH2ActLimit", by layer parity
  float* X = lds;
  float* Y = lds + (NBUF - 1) * FT * FP;
  float* E = lds + NBUF * FT * FP;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
  const int64_t row0 = (int64_t)blockIdx.x * FT;
  const int n0 = wave * 32 * TJ;
  // x2h: the tile in LDS holds its values times `sa` (a power of two, per tile and layer; isa = 1 / sa)
  [[maybe_unused]] float sa = SA, isa = 1.f / SA;
  [[maybe_unused]] float iwsv = 0.f;
  if constexpr (H2) iwsv = h2_iws_load(g.h2tab, lane);

  // ---- positional encoding of the tile: X[:, 0:Ep] = [x, sin(2^k x), cos(2^k x)], zero padded -------
  [[maybe_unused]] float xm = 0.f;
  {
    constexpr int PARTS = NT / FT;
    const int p = tid % FT, part = tid / FT;
    const int64_t row = row0 + p;
    float x[3] = {0.f, 0.f, 0.f};
    if (row < g.M) {
      if (g.grid.on) {   // row = ((ix - x_begin) * res + iy) * res + iz of the slab
        const int res = g.grid.res;
        int64_t r = row;
        const int iz = (int)(r % res);
        r /= res;
        const int iy = (int)(r % res);
        const int ix = (int)(r / res) + g.grid.x_begin;
        x[0] = linspace_at(g.grid.bmin[0], g.grid.bmax[0], res, ix) * g.scale;
        x[1] = linspace_at(g.grid.bmin[1], g.grid.bmax[1], res, iy) * g.scale;
        x[2] = linspace_at(g.grid.bmin[2], g.grid.bmax[2], res, iz) * g.scale;
      } else {
        x[0] = g.pts[row * 3] * g.scale;
        x[1] = g.pts[row * 3 + 1] * g.scale;
        x[2] = g.pts[row * 3 + 2] * g.scale;
      }
    }
    float* xr = X + p * FP;
    float* er = E + p * FEP;
    if (part == 0) {
      xr[0] = x[0] * SA; xr[1] = x[1] * SA; xr[2] = x[2] * SA;
      er[0] = x[0]; er[1] = x[1]; er[2] = x[2];
      for (int c = g.pe; c < g.Ep; ++c) xr[c] = 0.f;
      if (SAVE) {
        g.x4[row * 4] = x[0]; g.x4[row * 4 + 1] = x[1]; g.x4[row * 4 + 2] = x[2]; g.x4[row * 4 + 3] = 0.f;
      }
    }
    for (int k = part; k < g.multires; k += PARTS) {
      const float f = (float)(1 << k);
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        float s, co;
        sincosf(x[d] * f, &s, &co);
        const int c = 3 + 6 * k + d;
        xr[c] = s * SA; xr[c + 3] = co * SA;
        er[c] = s; er[c + 3] = co;
      }
    }
    if constexpr (H2) {   // the only unbounded entries of the encoding are the coordinates themselves
      xm = fmaxf(fmaxf(fabsf(x[0]), fabsf(x[1])), fabsf(x[2]));
      // (every wave leaves its own word: nothing to initialise; the layers' flags start from zero behind the same barrier)
      if (lane == 0) wmx[wave] = __builtin_amdgcn_ballot_w64(xm >= kH2ActLimit) != 0 ? 1.f : 0.f;
      if (tid < 2) ovf[tid] = 0;
    }
  }
  __syncthreads();
  if constexpr (H2) {
    if (tile_max<NW>(wmx) != 0.f) {   // (workgroup-uniform) coordinates beyond 256: this tile carries a smaller scale
      __syncthreads();                // (every wave has read the words)
      xm = wave_max(xm);
      if (lane == 0) wmx[wave] = xm;
      __syncthreads();
      const float tm = fmaxf(tile_max<NW>(wmx), 1.f);
      if (SAVE && g.smax != nullptr && tid == 0) amax_tile_commit(g.smax + SMAX_E, tm);
      x2h_dyn_scale(__builtin_bit_cast(unsigned, tm), sa, isa);
      const float f = sa * (1.f / SA);
      for (int idx = tid; idx < FT * g.Ep; idx += NT) {
        const int r = idx / g.Ep, c = idx - r * g.Ep;
        X[r * FP + c] *= f;
      }
      __syncthreads();
    }
  }
  if (SAVE) {   // e is an operand of the backward (dW of layer 0) and of the R sweep: FT x Ep floats
    for (int idx = tid; idx < FT * g.Ep; idx += NT) {
      const int r = idx / g.Ep, c = idx - r * g.Ep;
      g.e[(row0 + r) * g.Ep + c] = X[r * FP + c] * isa;
    }
  }

  const int h = lane >> 5, cl = lane & 31;
  v16f acc[TI][TJ];
  [[maybe_unused]] X3Mma<TI, TJ, WP> mm;
  if constexpr (X3) mm.request(g.w3 + WP * g.w_off[0], g.Kp[0], n0, lane);
  // x2h, rare path: the tile at X (written by layer l_written, times SA) holds a value beyond the fixed scale — the waves
  // exchange their maxima, every thread rescales the elements it wrote, the tile's maximum goes to PointBufs::smax
  [[maybe_unused]] auto rescale_input = [&](float am_thread, int l_written) {
    am_thread = wave_max(am_thread);
    if (lane == 0) wmx[wave] = am_thread;
    lds_barrier();
    const float tm = tile_max<NW>(wmx);
    if (SAVE && g.smax != nullptr && tid == 0) amax_tile_commit(g.smax + SMAX_A + l_written, tm);
    if (tid == 0) ovf[l_written & 1] = 0;
    x2h_dyn_scale(__builtin_bit_cast(unsigned, tm), sa, isa);
    const float f = sa * (1.f / SA);
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
      for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int r = 0; r < 16; ++r) X[(ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * FP + n0 + tj * 32 + cl] *= f;
    lds_barrier();
  };
  [[maybe_unused]] float am_prev = 0.f;   // x2h: this thread's maximum of what it wrote to the tile in the previous layer
  for (int l = 0; l < g.nh; ++l) {
    // x2h: the flag of the tile this layer reads is REQUESTED here and looked at after the matrix loop (its LDS round trip
    // hides under the loop; read right behind the previous layer's barrier it cost F(save) 7 us): the product runs
    // speculatively on the fixed scale and is redone on the rare tile that needed another one
    [[maybe_unused]] int pend = 0;
    if constexpr (H2 && (RNB_H2_GUARD_AB == 0 || RNB_H2_GUARD_AB == 3)) {
      if (l > 0) pend = *reinterpret_cast<const volatile int*>(&ovf[(l - 1) & 1]);
    }
    // x2h: accumulator -> pre-activation: 1 / (scale of the tile x scale of this layer's matrix in the mirror)
#if RNB_H2_GUARD_AB == 3   // (timing experiment only: the round-4 literal instead of the runtime factor)
    [[maybe_unused]] float inv = H2 ? 1.f / (kH2ActScale * kH2WScale) : 1.f;
#else
    [[maybe_unused]] float inv = H2 ? isa * h2_iws_at(iwsv, l) : 1.f;
#endif
    if constexpr (X3) {   // the next product's first weight steps are requested before this layer's epilogue
      const x3raw* wn = l + 1 < g.nh ? g.w3 + WP * g.w_off[l + 1] : (g.with_feat ? g.w3 + WP * g.wf_off : nullptr);
      mm.run(X, g.w3 + WP * g.w_off[l], g.Kp[l], n0, lane, acc, wn, FH, n0);
      if constexpr (H2 && (RNB_H2_GUARD_AB == 0 || RNB_H2_GUARD_AB == 3)) {
        if (__builtin_expect(__builtin_amdgcn_readfirstlane(pend) != 0, 0)) {   // (workgroup-uniform)
          lds_barrier();   // every wave has finished its (void) pass over the tile
          rescale_input(am_prev, l - 1);
#if RNB_H2_GUARD_AB != 3
          inv = isa * h2_iws_at(iwsv, l);
#endif
          mm.request(g.w3 + WP * g.w_off[l], g.Kp[l], n0, lane);
          mm.run(X, g.w3 + WP * g.w_off[l], g.Kp[l], n0, lane, acc, wn, FH, n0);
        }
      }
    } else layer_mma_nt<TI, NoHook, TJ>(X, g.packed + g.w_off[l], g.Kp[l], n0, lane, acc);
    if constexpr (NBUF == 1) lds_barrier();   // every wave has finished reading the input activations
    const float* bias = g.packed + g.b_off[l];
    // saved state goes out through buffer stores: one 32-bit lane offset per column tile plus a
    // compile-time row offset in the scalar operand (plain pointer stores cost a 64-bit VGPR address
    // pair per element, i.e. 128 extra registers and spills)
    const BufRsrc ra = tile_rsrc(SAVE ? g.a[l] + (size_t)row0 * FH : nullptr, FT * FH * 4);
    const BufRsrc rD = tile_rsrc(SAVE ? g.D[l] + (size_t)row0 * FH : nullptr, FT * FH * 4);
    const int n_real = g.n_real[l];
    const bool pe_tail = (l + 1 == g.skip);
    // x2h: max |.| of what this thread writes to the tile.  Softplus outputs are >= +0: ONE v_max3_i32 per pair of values
    // (h2_track2), no branch; the signed encoding columns of the one tile that carries the skip connection are tracked where
    // they are written (a per-pair choice between the two forms cost a scalar branch per pair: +5 % on this kernel).
    [[maybe_unused]] int amb = 0;
    [[maybe_unused]] float am = 0.f;
    // One column tile of the wave.  FULL (wave-uniform, decided per tile outside): every column is a real output of the layer —
    // no per-element column check.  The choice is made ONCE per tile between two instantiations: as a condition inside the
    // loops the compiler kept it as a scalar branch per value pair (32 taken branches per tile and layer).
    auto column_tile = [&](auto full_c, int tj) {
      constexpr bool FULL = decltype(full_c)::value;
      const int col = n0 + tj * 32 + cl;
      const float bc = bias[col];
      const unsigned voff = (unsigned)(4 * h * FH + col) * 4u;
#pragma unroll
      for (int ti = 0; ti < TI; ++ti) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          // registers r, r + 1 are rows rowc, rowc + 1 of the same column: softplus on the pair (packed math)
          const int rowc = ti * 32 + (r & 3) + 8 * (r >> 2);   // compile-time part of the row
          const int row = rowc + 4 * h;
          vf2 a, D;
          const vf2 z = H2 ? vf2{__builtin_fmaf(acc[ti][tj][r], inv, bc), __builtin_fmaf(acc[ti][tj][r + 1], inv, bc)}
                           : vf2{acc[ti][tj][r] + bc, acc[ti][tj][r + 1] + bc};
          constexpr bool SCALAR_EPI = H2 ? RNB_H2_SCALAR_EPI != 0 : (X3 && RNB_X3_SCALAR_EPI != 0);
          if constexpr (SAVE) softplus_aD_sel<SCALAR_EPI>(z, a, D);
          else a = softplus_a_sel<SCALAR_EPI>(z);
          if constexpr (!FULL) {
            if (col >= n_real) {   // only the tile straddling the skip connection's PE columns
              const bool pe_col = pe_tail && col < n_real + g.pe;
              a = vf2{pe_col ? E[row * FEP + (col - n_real)] : 0.f, pe_col ? E[(row + 1) * FEP + (col - n_real)] : 0.f};
              D = vf2{0.f, 0.f};
              if constexpr (H2) am = fmaxf(am, fmaxf(fabsf(a.x), fabsf(a.y)));
            }
          }
          Y[row * FP + col] = a.x * SA;
          Y[(row + 1) * FP + col] = a.y * SA;
          if constexpr (H2 && RNB_H2_GUARD_AB != 1) h2_track2(amb, a.x, a.y);
          if (SAVE) {
            bstore(ra, voff, rowc * FH * 4, a.x);
            bstore(ra, voff, (rowc + 1) * FH * 4, a.y);
            bstore(rD, voff, rowc * FH * 4, D.x);
            bstore(rD, voff, (rowc + 1) * FH * 4, D.y);
          }
        }
      }
    };
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
      if (n0 + tj * 32 + 32 <= n_real) column_tile(std::true_type{}, tj);   // (wave-uniform)
      else column_tile(std::false_type{}, tj);
    }
    if constexpr (H2 && RNB_H2_GUARD_AB != 1) {
      am = fmaxf(am, __builtin_bit_cast(float, amb));
      h2_raise_flag(am, &ovf[l & 1], lane);
    }
    lds_barrier();   // the new activations are visible to every wave
    if constexpr (H2) {   // (the fixed scale again, unless the next layer's look at the flag says otherwise)
      sa = SA;
      isa = 1.f / SA;
      am_prev = am;
    }
    if constexpr (NBUF == 2) { float* t = X; X = Y; Y = t; }
  }
  if constexpr (H2 && RNB_H2_GUARD_AB == 0) {   // the last hidden layer's tile, read by the two heads below
    if (h2_flag_up(&ovf[(g.nh - 1) & 1])) rescale_input(am_prev, g.nh - 1);
  }

  // ---- sdf head: row 0 of the output layer (models/fields.py:104, :106-108) -----------------------------
  {
    const float* ws = g.packed + g.wsdf_off;
    float w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) w[u] = ws[lane + 64 * u];
    const float bs = g.packed[g.bsdf_off];
    for (int rr = 0; rr < FT / NW; ++rr) {
      const int row = wave * (FT / NW) + rr;
      float s = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) s = fmaf(X[row * FP + lane + 64 * u], w[u], s);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      if (lane == 0) {
        const float v = (H2 ? __builtin_fmaf(s, isa, bs) : s + bs) / g.scale;
        if (!g.grid.on) g.sdf[row0 + row] = v;
        else if (row0 + row < g.M) g.sdf[row0 + row] = v * g.grid.out_scale;   // the volume has exactly M entries
      }
    }
  }
  // ---- feature head: rows 1.. of the output layer, written into the albedo network's input ------------
  if (g.with_feat) {
    [[maybe_unused]] const float inv = H2 ? isa * h2_iws_at(iwsv, g.nh) : 1.f;
    if constexpr (X3) mm.run(X, g.w3 + WP * g.wf_off, FH, n0, lane, acc, nullptr, 0, 0);   // (requested by the last hidden layer)
    else layer_mma_nt<TI, NoHook, TJ>(X, g.packed + g.wf_off, FH, n0, lane, acc);
    const float* bias = g.packed + g.bf_off;
    const BufRsrc rc = tile_rsrc(g.cin + (size_t)row0 * g.Cinp, FT * g.Cinp * 4);
    const unsigned rowb = (unsigned)g.Cinp * 4u;   // bytes per row of the albedo-net input
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
      const int col = n0 + tj * 32 + cl;
      if (col < g.F) {
        const float bc = bias[col];
        const unsigned voff = (unsigned)(4 * h) * rowb + (unsigned)col * 4u;
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rowc = ti * 32 + (r & 3) + 8 * (r >> 2);
            bstore(rc, voff, rowc * rowb, H2 ? __builtin_fmaf(acc[ti][tj][r], inv, bc) : acc[ti][tj][r] + bc);
          }
        }
      }
    }
  }
}

// ---- RNB_VARIANT_X3: the split weight mirror ----------------------------------------------------------------
constexpr int kMaxX3 = 4 * RNB_MAX_LIN + 2;
// id: slot of the matrix in the x2h scale table (H2Tab; W and W^T share it), or -1; tr: this entry is the transposed copy
struct X3Entry { long long off; int N, K, unit_begin, tperm, id, tr; };
struct X3Table { int n, total_units; X3Entry e[kMaxX3]; };
// one thread per 16-byte unit of one plane-triple: W[32 nt + c][16 ks + 8 h .. +8] -> hi, mid, lo.
// tperm (the SDF network's matrices): the lane's 8 k of a step are 16 ks + 4 h + {0..3, 8..11} instead — the order in
// which the accumulator of a TRANSPOSED product (sweep_mv.hip: mv_kfeat) hands its features to the next layer; the
// LDS-tile kernels read their activation rows in the same order (x3_read_a), so one mirror serves both families.
// tab != nullptr (x2h): the maximum |w| of every matrix is left in tab->wmax[id] on the way (float bits, zeroed by
// wn_fwd_kernel; W entries only: W^T holds the same values) — x2h_pack_kernel, which follows, takes the matrix's scale from it.
__global__ __launch_bounds__(256) void x3_pack_kernel(const float* __restrict__ src, X3Table t, x3raw* __restrict__ dst,
                                                      H2Tab* __restrict__ tab) {
  __shared__ float wmx[4];
  const int u = blockIdx.x * 256 + threadIdx.x;
  const bool live = u < t.total_units;
  int ei = 0;
  while (ei + 1 < t.n && u >= t.e[ei + 1].unit_begin) ++ei;
  const X3Entry en = t.e[ei];
  float m = 0.f;
  if (live) {
    const int lu = u - en.unit_begin;            // fragment lu / 64, lane lu % 64
    const int frag = lu >> 6, lane = lu & 63;
    const int nks = en.K >> 4;
    const int nt = frag / nks, ks = frag - nt * nks;
    const int c = lane & 31, h = lane >> 5;
    const float* sp = src + en.off + (size_t)(nt * 32 + c) * en.K + ks * 16 + h * (en.tperm ? 4 : 8);
    const vf4 x0 = *reinterpret_cast<const vf4*>(sp), x1 = *reinterpret_cast<const vf4*>(sp + (en.tperm ? 8 : 4));
    vu4x hi, mid, lo;
    x3_split8(x0, x1, hi, mid, lo);
    x3raw* dp = dst + 3 * en.off + ((size_t)frag * 3 * 64 + lane) * 8;
    *reinterpret_cast<vu4x*>(dp) = hi;
    *reinterpret_cast<vu4x*>(dp + 512) = mid;
    *reinterpret_cast<vu4x*>(dp + 1024) = lo;
    m = fmaxf(fmaxf(fmaxf(fabsf(x0.x), fabsf(x0.y)), fmaxf(fabsf(x0.z), fabsf(x0.w))),
              fmaxf(fmaxf(fabsf(x1.x), fabsf(x1.y)), fmaxf(fabsf(x1.z), fabsf(x1.w))));
  }
  if (tab == nullptr) return;   // (uniform)
  // x2h: max |w| of the matrix -> tab->wmax[id] (W entries only).  A wave lies inside one entry (units are multiples of 64);
  // when the whole workgroup does (the shipped shapes), its four waves' maxima meet in LDS and ONE conditional atomic leaves
  // — one per wave put ~260 same-address atomics in a row on every slot: +16 us per step.
  const bool rec = live && en.id >= 0 && !en.tr;
  const int blk0 = blockIdx.x * 256;
  const bool whole = en.unit_begin <= blk0 && blk0 + 256 <= en.unit_begin + en.N * en.K / 8;   // (workgroup-uniform)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if (!whole) {
    if (rec && (threadIdx.x & 63) == 0) amax_tile_commit(tab->wmax + en.id, m);
    return;
  }
  if ((threadIdx.x & 63) == 0) wmx[threadIdx.x >> 6] = m;
  __syncthreads();
  if (rec && threadIdx.x == 0) amax_tile_commit(tab->wmax + en.id, fmaxf(fmaxf(wmx[0], wmx[1]), fmaxf(wmx[2], wmx[3])));
}
// the scale of a matrix of the fp16 mirror from the float bits of its max |w|: 2^8 (the round-4 constant: results unchanged)
// while the maximum is below 64; beyond, the power of two that puts the maximum in [2^13, 2^14) — no finite weight overflows
// fp16.  (A non-finite maximum keeps 2^8: the mirror then carries the inf / NaN into every product, as fp32 arithmetic would.)
__device__ inline float x2h_weight_scale(unsigned mbits, float& inv) {
  float s = kH2WScale;
  inv = 1.f / kH2WScale;
  const int ef = (int)(mbits >> 23);
  if (ef >= 127 + 6 && ef < 255) x2h_dyn_scale(mbits, s, inv);
  return s;
}
// the fp16 mirror of the forward-type kernels (x2h): same fragments, two planes, weights times the matrix's scale
__global__ void x2h_pack_kernel(const float* __restrict__ src, X3Table t, x3raw* __restrict__ dst, H2Tab* __restrict__ tab) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= t.total_units) return;
  int ei = 0;
  while (ei + 1 < t.n && u >= t.e[ei + 1].unit_begin) ++ei;
  const X3Entry en = t.e[ei];
  const int lu = u - en.unit_begin;
  const int frag = lu >> 6, lane = lu & 63;
  const int nks = en.K >> 4;
  const int nt = frag / nks, ks = frag - nt * nks;
  const int c = lane & 31, h = lane >> 5;
  float inv;
  const float sw = x2h_weight_scale(tab->wmax[en.id], inv);
  if (lu == 0 && !en.tr) { tab->ws[en.id] = sw; tab->iws[en.id] = inv; }   // what the consumers read
  const float* sp = src + en.off + (size_t)(nt * 32 + c) * en.K + ks * 16 + h * (en.tperm ? 4 : 8);
  vu4x hi, lo;
  x2h_split8(*reinterpret_cast<const vf4*>(sp) * sw, *reinterpret_cast<const vf4*>(sp + (en.tperm ? 8 : 4)) * sw, hi, lo);
  x3raw* dp = dst + 2 * en.off + ((size_t)frag * 2 * 64 + lane) * 8;
  *reinterpret_cast<vu4x*>(dp) = hi;
  *reinterpret_cast<vu4x*>(dp + 512) = lo;
}
int x3_pack_weights(const Layout& L, float* packed, hipStream_t s) {
  x3raw* dst = reinterpret_cast<x3raw*>(packed + L.total);
  X3Table t;
  t.n = 0;
  t.total_units = 0;
  auto add = [&](long long off, int N, int K, int tperm, int id, int tr) {
    if (off < 0 || N <= 0 || K <= 0) return;
    X3Entry& e = t.e[t.n++];
    e.off = off; e.N = N; e.K = K; e.unit_begin = t.total_units; e.tperm = tperm; e.id = id; e.tr = tr;
    t.total_units += N * K / 8;
  };
  for (int l = 0; l < L.nh; ++l) {
    add(L.hid[l].w_off, L.hid[l].Np, L.hid[l].Kp, 1, l, 0);
    add(L.hid[l].wT_off, L.hid[l].Kp, L.hid[l].Np, 1, l, 1);
  }
  if (L.F > 0) {
    add(L.feat.w_off, L.feat.Np, L.feat.Kp, 1, L.nh, 0);
    add(L.feat.wT_off, L.feat.Kp, L.feat.Np, 1, L.nh, 1);
  }
  for (int l = 0; l < L.nc; ++l) {   // the albedo network's hidden layers (read as fragments by its kernels)
    add(L.col[l].w_off, L.col[l].Np, L.col[l].Kp, 0, L.nh + 1 + l, 0);
    add(L.col[l].wT_off, L.col[l].Kp, L.col[l].Np, 0, L.nh + 1 + l, 1);
  }
  H2Tab* tab = is_x2h(L) ? h2_tab(L, packed) : nullptr;
  hipLaunchKernelGGL(x3_pack_kernel, dim3((unsigned)((t.total_units + 255) / 256)), dim3(256), 0, s, packed, t, dst, tab);
  RNB_CHECK_LAUNCH();
  if (is_x2h(L)) {   // every matrix once more as two fp16 planes (the scales come from the maxima just taken)
    // the fp16 planes of the albedo network's matrices are read by its fused kernels (color_h2.hip) through the same
    // product loop as the SDF network's (X3Mma): same k order inside a step
    for (int q = 0; q < t.n; ++q) t.e[q].tperm = 1;
    hipLaunchKernelGGL(x2h_pack_kernel, dim3((unsigned)((t.total_units + 255) / 256)), dim3(256), 0, s, packed, t,
                       x2h_mirror(L, packed), tab);
    RNB_CHECK_LAUNCH();
  }
  return RNB_OK;
}

bool fused_supported(const Layout& L) {
  if (L.Hp != FH || L.H != FH) return false;
  if (L.Ep > 64 || L.pe > FEP) return false;
  if (L.nh < 1) return false;
  for (int l = 0; l < L.nh; ++l)
    if (L.hid[l].Np != FH || (L.hid[l].Kp != FH && l != 0)) return false;
  if (L.F > FH) return false;
  return true;
}

// Fused replacement of launch_pe_points + sweep_forward (same outputs; pb.a / pb.D only when `save`).
int fused_forward(const Layout& L, const float* packed, const float* pts, int64_t M, PointBufs& pb, bool save,
                  bool need_feat, bool need_gz_last, hipStream_t s, const GridGen* grid) {
  if (!save && use_reg_tile(L, pb.Mp)) return sweep_mv_forward(L, packed, pts, M, pb, save, need_feat, need_gz_last, s, grid);
  FusedFwdArgs g;
  memset(&g, 0, sizeof(g));
  if (grid) g.grid = *grid;
  g.pts = pts;
  g.M = M;
  g.packed = packed;
  const bool h2 = is_x2h(L);
  g.w3 = h2 ? x2h_mirror(L, packed) : reinterpret_cast<const x3raw*>(packed + L.total);
  g.nh = L.nh;
  g.skip = L.skip;
  g.pe = L.pe;
  g.multires = L.multires;
  g.Ep = L.Ep;
  g.scale = L.sdf_scale;
  for (int l = 0; l < L.nh; ++l) {
    g.n_real[l] = L.hid[l].N;
    g.Kp[l] = L.hid[l].Kp;
    g.w_off[l] = L.hid[l].w_off;
    g.b_off[l] = L.hid[l].b_off;
    g.a[l] = pb.a[l];
    g.D[l] = pb.D[l];
  }
  g.wsdf_off = L.wsdf_off;
  g.bsdf_off = L.bsdf_off;
  g.with_feat = need_feat ? 1 : 0;
  g.F = L.F;
  g.Cinp = L.Cinp;
  g.wf_off = L.feat.w_off;
  g.bf_off = L.feat.b_off;
  g.cin = pb.cin;
  g.sdf = pb.sdf;
  g.x4 = pb.x;
  g.e = pb.e;
  // (the fused reverse sweep seeds itself from D_last: nobody asks this kernel for the seed any more)
  if (need_gz_last) RNB_FAIL(RNB_E_INVALID, "fused forward sweep: the reverse sweep's seed is formed by fused_reverse_kernel");
  g.gz_last = nullptr;
  g.h2tab = h2 ? h2_tab(L, packed) : nullptr;
  g.smax = (save && h2) ? pb.smax : nullptr;
  // algorithmic FLOPs of the sweep (real layer shapes), for the optional event instrumentation
  double fl = 0;
  for (int l = 0; l < L.nh; ++l) fl += 2.0 * (double)M * L.hid[l].N * L.hid[l].K;
  fl += 2.0 * (double)M * L.H;
  if (need_feat) fl += 2.0 * (double)M * L.F * L.H;
  ProfScope prof(fl, s, save ? "F_sweep(save)" : "F_sweep(forward_only)");
  // 64-point tiles when that still gives every CU >= 2 workgroups, 32-point tiles for small batches
  const int force_ti = L.knob(RNB_VARIANT_FWD_TI_SHIFT);   // tuning knob: 1 or 2 forces the tile height
  const bool small = force_ti ? (force_ti == 1) : (pb.Mp / 64 < 512);
  const int force_nw = L.knob(RNB_VARIANT_FWD_NW_SHIFT);   // tuning knob: 1 = 4 waves, 2 = 8 waves (small batches)
  const bool x3 = is_x3(L);
  // (Measured and not kept: the 32-point form of the pre-split kernel for the sampling passes, and weight fragments four
  // steps ahead in the small-batch kernel: both 60 us per 8,192-point pass like the default — 256 workgroups each stream the
  // whole 3.5 MB of weight planes from L2, 0.9 GB per pass at the ~16 TB/s the L2s deliver for shared rows.)
  if (small) {
    const unsigned blocks = (unsigned)(pb.Mp / 32);
    const bool wide = force_nw ? (force_nw == 2) : (blocks <= 256);   // at most one workgroup per CU
    if (h2) {
      if (save && wide) hipLaunchKernelGGL((fused_forward_kernel<1, true, 8, true, true>), dim3(blocks), dim3(512), 0, s, g);
      else if (save) hipLaunchKernelGGL((fused_forward_kernel<1, true, 4, true, true>), dim3(blocks), dim3(256), 0, s, g);
      else if (wide) hipLaunchKernelGGL((fused_forward_kernel<1, false, 8, true, true>), dim3(blocks), dim3(512), 0, s, g);
      else hipLaunchKernelGGL((fused_forward_kernel<1, false, 4, true, true>), dim3(blocks), dim3(256), 0, s, g);
    } else if (x3) {
      if (save && wide) hipLaunchKernelGGL((fused_forward_kernel<1, true, 8, true>), dim3(blocks), dim3(512), 0, s, g);
      else if (save) hipLaunchKernelGGL((fused_forward_kernel<1, true, 4, true>), dim3(blocks), dim3(256), 0, s, g);
      else if (wide) hipLaunchKernelGGL((fused_forward_kernel<1, false, 8, true>), dim3(blocks), dim3(512), 0, s, g);
      else hipLaunchKernelGGL((fused_forward_kernel<1, false, 4, true>), dim3(blocks), dim3(256), 0, s, g);
    } else if (save && wide) hipLaunchKernelGGL((fused_forward_kernel<1, true, 8>), dim3(blocks), dim3(512), 0, s, g);
    else if (save) hipLaunchKernelGGL((fused_forward_kernel<1, true>), dim3(blocks), dim3(256), 0, s, g);
    else if (wide) hipLaunchKernelGGL((fused_forward_kernel<1, false, 8>), dim3(blocks), dim3(512), 0, s, g);
    else hipLaunchKernelGGL((fused_forward_kernel<1, false>), dim3(blocks), dim3(256), 0, s, g);
  } else {
    const unsigned blocks = (unsigned)(pb.Mp / 64);
    if (h2) {
      if (save) hipLaunchKernelGGL((fused_forward_kernel<2, true, 4, true, true>), dim3(blocks), dim3(256), 0, s, g);
      else hipLaunchKernelGGL((fused_forward_kernel<2, false, 4, true, true>), dim3(blocks), dim3(256), 0, s, g);
    } else if (x3) {
      if (save) hipLaunchKernelGGL((fused_forward_kernel<2, true, 4, true>), dim3(blocks), dim3(256), 0, s, g);
      else hipLaunchKernelGGL((fused_forward_kernel<2, false, 4, true>), dim3(blocks), dim3(256), 0, s, g);
    } else if (save) hipLaunchKernelGGL((fused_forward_kernel<2, true>), dim3(blocks), dim3(256), 0, s, g);
    else hipLaunchKernelGGL((fused_forward_kernel<2, false>), dim3(blocks), dim3(256), 0, s, g);
  }
  RNB_CHECK_LAUNCH();
  return RNB_OK;
}

}  // namespace rnb
