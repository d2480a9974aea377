// Fused SDF-network sweeps for the shipped network shape (hidden width 256): one workgroup carries a tile
// of 64 (or 32) points through ALL layers.  Activations stay in LDS between layers, weights stream from L2
// straight into MFMA B-fragments (each 128-byte weight line is fetched once per workgroup and consumed by
// four back-to-back 16-byte loads), and what the backward pass needs is written to HBM with fire-and-forget
// stores that overlap the next layer's matrix work.  Replaces, per sweep, the chain of per-layer GEMM
// launches of mlp.hip (which remain the generic path for other widths).
//
//   fused_forward_kernel   positional encoding + F sweep (+ sdf head, + feature head)
//                          models/embedder.py:40-46, models/fields.py:82-104
#include "fused_common.hip.h"

// A/B switch (compile time): scalar instead of packed fp32 math in the x3 forward epilogue.  Measured on one box, two
// runs each: packed 3.694 / 3.684 ms per step, scalar 3.71 / 3.78 — the epilogue runs beside ANOTHER wave's MFMAs, where
// the packed forms keep their halved issue count.
#ifndef RNB_X3_SCALAR_EPI
#define RNB_X3_SCALAR_EPI 0
#endif

namespace rnb {

struct FusedFwdArgs {
  const float* pts;     // [M,3]
  int64_t M;
  const float* packed;
  const x3raw* w3;      // RNB_VARIANT_X3: split mirror of the weight matrices (matrix at 3 x its float offset)
  int nh, skip, pe, multires, Ep;
  float scale;
  int n_real[RNB_MAX_LIN];
  int Kp[RNB_MAX_LIN];
  long long w_off[RNB_MAX_LIN], b_off[RNB_MAX_LIN];
  long long wsdf_off, bsdf_off;
  int with_feat, F, Cinp;
  long long wf_off, bf_off;
  float* cin;           // [Mp,Cinp] feature block destination (with_feat)
  float* sdf;           // [Mp]
  // saved state (SAVE only)
  float* x4;            // [Mp,4]
  float* e;             // [Mp,Ep]
  float* a[RNB_MAX_LIN];
  float* D[RNB_MAX_LIN];
  float* gz_last;       // [Mp,256] seed of the reverse sweep: w_sdf * D_last (optional)
  GridGen grid;         // on: points come from the regular grid, sdf (scaled) goes to rows < M only
};

// TI = row tiles per workgroup (64 points for TI = 2; 32 points for TI = 1, used for small batches so that
// every CU still gets a workgroup).  NW = waves per workgroup: 4 (each wave 64 output columns) or 8 (32 columns
// each) — the latter for batches so small that a CU holds a single workgroup: two waves per SIMD instead of one
// hide each other's LDS / L2 waits.
template <int TI, bool SAVE, int NW = 4, bool X3 = false>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void fused_forward_kernel(FusedFwdArgs g) {
  constexpr int FT = 32 * TI;
  constexpr int NT = 64 * NW;     // threads
  constexpr int TJ = 8 / NW;      // 32-column tiles per wave
  // TI == 1: two activation tiles (a layer reads one, writes the other: one barrier per layer);
  // TI == 2: one tile updated in place behind a second barrier (two tiles would not leave room for two
  // workgroups per CU)
  constexpr int NBUF = TI == 1 ? 2 : 1;
  __shared__ __attribute__((aligned(16))) float lds[NBUF * FT * FP + FT * FEP];
  float* X = lds;
  float* Y = lds + (NBUF - 1) * FT * FP;
  float* E = lds + NBUF * FT * FP;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
  const int64_t row0 = (int64_t)blockIdx.x * FT;
  const int n0 = wave * 32 * TJ;

  // ---- positional encoding of the tile: X[:, 0:Ep] = [x, sin(2^k x), cos(2^k x)], zero padded -------
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
      xr[0] = x[0]; xr[1] = x[1]; xr[2] = x[2];
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
        xr[c] = s; xr[c + 3] = co;
        er[c] = s; er[c + 3] = co;
      }
    }
  }
  __syncthreads();
  if (SAVE) {   // e is an operand of the backward (dW of layer 0) and of the R sweep: FT x Ep floats
    for (int idx = tid; idx < FT * g.Ep; idx += NT) {
      const int r = idx / g.Ep, c = idx - r * g.Ep;
      g.e[(row0 + r) * g.Ep + c] = X[r * FP + c];
    }
  }

  const int h = lane >> 5, cl = lane & 31;
  v16f acc[TI][TJ];
  [[maybe_unused]] X3Mma<TI, TJ> mm;
  if constexpr (X3) mm.request(g.w3 + 3 * g.w_off[0], g.Kp[0], n0, lane);
  for (int l = 0; l < g.nh; ++l) {
    if constexpr (X3) {   // the next product's first weight steps are requested before this layer's epilogue
      const x3raw* wn = l + 1 < g.nh ? g.w3 + 3 * g.w_off[l + 1] : (g.with_feat ? g.w3 + 3 * g.wf_off : nullptr);
      mm.run(X, g.w3 + 3 * g.w_off[l], g.Kp[l], n0, lane, acc, wn, FH, n0);
    } else layer_mma_nt<TI, NoHook, TJ>(X, g.packed + g.w_off[l], g.Kp[l], n0, lane, acc);
    if constexpr (NBUF == 1) lds_barrier();   // every wave has finished reading the input activations
    const float* bias = g.packed + g.b_off[l];
    // saved state goes out through buffer stores: one 32-bit lane offset per column tile plus a
    // compile-time row offset in the scalar operand (plain pointer stores cost a 64-bit VGPR address
    // pair per element, i.e. 128 extra registers and spills)
    const BufRsrc ra = tile_rsrc(SAVE ? g.a[l] + (size_t)row0 * FH : nullptr, FT * FH * 4);
    const BufRsrc rD = tile_rsrc(SAVE ? g.D[l] + (size_t)row0 * FH : nullptr, FT * FH * 4);
    const BufRsrc rg = tile_rsrc((SAVE && g.gz_last) ? g.gz_last + (size_t)row0 * FH : nullptr, FT * FH * 4);
    const int n_real = g.n_real[l];
    const bool pe_tail = (l + 1 == g.skip);
    const bool last = (l + 1 == g.nh);
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
      const int col = n0 + tj * 32 + cl;
      const float bc = bias[col];
      const float ws = (SAVE && last && g.gz_last) ? g.packed[g.wsdf_off + col] : 0.f;
      const bool tile_full = n0 + tj * 32 + 32 <= n_real;   // wave-uniform: no per-element column checks
      const unsigned voff = (unsigned)(4 * h * FH + col) * 4u;
#pragma unroll
      for (int ti = 0; ti < TI; ++ti) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          // registers r, r + 1 are rows rowc, rowc + 1 of the same column: softplus on the pair (packed math)
          const int rowc = ti * 32 + (r & 3) + 8 * (r >> 2);   // compile-time part of the row
          const int row = rowc + 4 * h;
          vf2 a, D;
          if constexpr (SAVE) softplus_aD_sel<X3 && RNB_X3_SCALAR_EPI>(vf2{acc[ti][tj][r] + bc, acc[ti][tj][r + 1] + bc}, a, D);
          else a = softplus_a_sel<X3 && RNB_X3_SCALAR_EPI>(vf2{acc[ti][tj][r] + bc, acc[ti][tj][r + 1] + bc});
          if (!tile_full && col >= n_real) {   // only the tile straddling the skip connection's PE columns
            const bool pe_col = pe_tail && col < n_real + g.pe;
            a = vf2{pe_col ? E[row * FEP + (col - n_real)] : 0.f, pe_col ? E[(row + 1) * FEP + (col - n_real)] : 0.f};
            D = vf2{0.f, 0.f};
          }
          Y[row * FP + col] = a.x;
          Y[(row + 1) * FP + col] = a.y;
          if (SAVE) {
            bstore(ra, voff, rowc * FH * 4, a.x);
            bstore(ra, voff, (rowc + 1) * FH * 4, a.y);
            bstore(rD, voff, rowc * FH * 4, D.x);
            bstore(rD, voff, (rowc + 1) * FH * 4, D.y);
            if (last && g.gz_last) {
              bstore(rg, voff, rowc * FH * 4, ws * D.x);
              bstore(rg, voff, (rowc + 1) * FH * 4, ws * D.y);
            }
          }
        }
      }
    }
    lds_barrier();   // the new activations are visible to every wave
    if constexpr (NBUF == 2) { float* t = X; X = Y; Y = t; }
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
        const float v = (s + bs) / g.scale;
        if (!g.grid.on) g.sdf[row0 + row] = v;
        else if (row0 + row < g.M) g.sdf[row0 + row] = v * g.grid.out_scale;   // the volume has exactly M entries
      }
    }
  }
  // ---- feature head: rows 1.. of the output layer, written into the albedo network's input ------------
  if (g.with_feat) {
    if constexpr (X3) mm.run(X, g.w3 + 3 * g.wf_off, FH, n0, lane, acc, nullptr, 0, 0);   // (requested by the last hidden layer)
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
            bstore(rc, voff, rowc * rowb, acc[ti][tj][r] + bc);
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// x3 forward with the activations PRE-SPLIT in LDS ("p3"; A/B variant: fwd_ti = 2, fwd_nw = 8)
// ---------------------------------------------------------------------------------------------------------------
// fused_forward_kernel<.., X3> keeps the tile fp32 in LDS and lets each of the four waves of a workgroup split the A rows
// it reads: 4 x redundant, and the term that saturates the vector-issue port.  Here the producer of an
// activation splits it once and the tile lives in LDS as three bf16 planes [64][264] (6 bytes per element: ONE 8-wave
// workgroup per CU, each wave 64 rows x 32 columns), so the matrix loop is fragment reads + MFMAs only.
constexpr int PP = 264;            // plane pitch in bf16 elements (528 B = 132 dwords = 4 mod 64: conflict-free b128 reads)

// rows r and r + 1 of one column (what an accumulator register pair holds): split the pair, six 2-byte stores
template <int PPLANE>
__device__ inline void p3_put2(x3raw* __restrict__ P, int idx, float a, float b) {
  unsigned uh = x3_pack2(a, b);
  asm("" : "+v"(uh));
  const float ra = a - __builtin_bit_cast(float, uh << 16);
  const float rb = __builtin_fmaf(__builtin_bit_cast(float, uh & 0xffff0000u), -1.f, b);
  unsigned um = x3_pack2(ra, rb);
  asm("" : "+v"(um));
  const float sa = ra - __builtin_bit_cast(float, um << 16);
  const float sb = __builtin_fmaf(__builtin_bit_cast(float, um & 0xffff0000u), -1.f, rb);
  const unsigned ul = x3_pack2(sa, sb);
  P[idx] = (x3raw)uh;               P[idx + PP] = (x3raw)(uh >> 16);
  P[idx + PPLANE] = (x3raw)um;      P[idx + PPLANE + PP] = (x3raw)(um >> 16);
  P[idx + 2 * PPLANE] = (x3raw)ul;  P[idx + 2 * PPLANE + PP] = (x3raw)(ul >> 16);
}
template <int PPLANE>
__device__ inline void p3_put(x3raw* __restrict__ P, int idx, float a) {
  const unsigned uh = x3_pack2(a, 0.f);
  const float ra = a - __builtin_bit_cast(float, uh << 16);
  const unsigned um = x3_pack2(ra, 0.f);
  const float sa = ra - __builtin_bit_cast(float, um << 16);
  P[idx] = (x3raw)uh;
  P[idx + PPLANE] = (x3raw)um;
  P[idx + 2 * PPLANE] = (x3raw)x3_pack2(sa, 0.f);
}
template <int PPLANE>
__device__ inline float p3_get(const x3raw* __restrict__ P, int idx) {
  return (__builtin_bit_cast(float, (unsigned)P[idx] << 16) + __builtin_bit_cast(float, (unsigned)P[idx + PPLANE] << 16)) +
         __builtin_bit_cast(float, (unsigned)P[idx + 2 * PPLANE] << 16);
}
// acc = X W^T for one wave: rows 0 .. 32 TI of the plane tile (plane stride PL elements), columns n0 .. n0 + 32;
// fragments one 16-k step ahead
template <int TI>
__device__ inline void layer_mma_p3(const x3raw* __restrict__ P, const x3raw* __restrict__ W3, int K, int n0, int lane,
                                    v16f (&acc)[TI][1]) {
  constexpr int PL = 32 * TI * PP;
  const int i = lane & 31, h = lane >> 5;
  const x3raw* ap = P + i * PP + h * 8;
  const int nks = K >> 4;   // even
  vu4x a0[TI][3], a1[TI][3], b0[1][3], b1[1][3];
  auto read_a = [&](int ks, vu4x (&a)[TI][3]) {
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) a[ti][pl] = *reinterpret_cast<const vu4x*>(ap + ti * 32 * PP + pl * PL + ks * 16);
  };
  x3_load_b<1>(W3, nks, n0, 0, lane, b0);
  read_a(0, a0);
  x3_load_b<1>(W3, nks, n0, 1, lane, b1);
  read_a(1, a1);
  __builtin_amdgcn_sched_barrier(0);
  x3_mfma<TI, 1, true>(a0, b0, acc);
  __builtin_amdgcn_sched_barrier(0);
  const int last = nks - 1;
  for (int ks = 1; ks + 1 < nks; ks += 2) {
    x3_load_b<1>(W3, nks, n0, ks + 1, lane, b0);
    read_a(ks + 1, a0);
    __builtin_amdgcn_sched_barrier(0);
    x3_mfma<TI, 1, false>(a1, b1, acc);
    __builtin_amdgcn_sched_barrier(0);
    x3_load_b<1>(W3, nks, n0, min(ks + 2, last), lane, b1);
    read_a(min(ks + 2, last), a1);
    __builtin_amdgcn_sched_barrier(0);
    x3_mfma<TI, 1, false>(a0, b0, acc);
    __builtin_amdgcn_sched_barrier(0);
  }
  x3_mfma<TI, 1, false>(a1, b1, acc);
}

// ---------------------------------------------------------------------------------------------------------------
// "p3s": the pre-split kernel WITHOUT workgroup barriers between the layers.  The 8 waves form two groups, A = waves 0-3
// (activation columns 0..127) and B = waves 4-7 (columns 128..255); waves w and w + 4 share a SIMD.  Every wave contracts
// over the A columns first, then over the B columns; B starts a layer's matrix loop when A has finished the FIRST HALF of
// its own — so the groups run half a layer apart, and while a wave of A is in its epilogue (softplus, split, stores) its
// SIMD partner of B multiplies, and vice versa: what two workgroups per CU give the fp32-tile kernel, inside ONE workgroup
// whose plane tile (6 bytes per element) fills the LDS.  The tile is updated in place under five counters in LDS:
//   WA / WB  waves of A / B that have WRITTEN their columns of layer l          (read-after-write: 4 (l + 1))
//   RAa, RAb waves of A / B that have READ the A columns in layer l's loop      (write-after-read for A's epilogue, and RAa
//            is B's start signal: the half-layer offset)
//   RB       waves (all 8) that have READ the B columns in layer l's loop       (write-after-read for B's epilogue)
// An LDS instruction of one wave executes after the LDS instructions that wave issued before it, so a counter add needs
// no wait in front of it; the readers poll (ds_read + s_sleep).
struct P3Sync { int WA, WB, RAa, RAb, RB, pad[3]; };
__device__ inline void p3_wait(const int* c, int target) {
  asm volatile("" ::: "memory");
  // bounded: a wave that has waited ~1 s gives up (the counters only ever lag by one layer; this is the exit condition
  // every wave reaches whatever happens, so the grid always drains)
  for (int spin = 0; spin < (1 << 22); ++spin) {
    if (__builtin_amdgcn_readfirstlane(*(const volatile int*)c) >= target) break;
    __builtin_amdgcn_s_sleep(1);
  }
  asm volatile("" ::: "memory");
}
__device__ inline void p3_signal(int* c, int lane) {
  asm volatile("" ::: "memory");
  if (lane == 0) __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
}
// the matrix loop of layer_mma_p3 with the two hand-offs: `before_b` runs before the first fragment read of the second
// half of the k range is issued, `after_a` once every read of the first half has been issued.  The weight fragments of the
// first two steps are requested by `request` — before the caller waits for its input columns.
#ifndef RNB_P3S_MM_PRIO
#define RNB_P3S_MM_PRIO 1
#define RNB_P3S_EP_PRIO 0
#endif
template <int TI>
struct P3sMma {
  vu4x b[4][1][3];   // weight fragments of four 16-k steps in flight (one step of this wave is 12 MFMAs = 384 clocks: one
                     // step ahead does not cover an L2 round trip)
  __device__ inline void request(const x3raw* __restrict__ W3, int K, int n0, int lane) {
#pragma unroll
    for (int q = 0; q < 4; ++q) x3_load_b<1>(W3, K >> 4, n0, q, lane, b[q]);   // (K >= 64: at least four steps)
  }
  template <class FB, class FA>
  __device__ inline void run(const x3raw* __restrict__ P, const x3raw* __restrict__ W3, int K, int n0, int lane,
                             v16f (&acc)[TI][1], FB before_b, FA after_a) {
    constexpr int PL = 32 * TI * PP;
    const int i = lane & 31, h = lane >> 5;
    const x3raw* ap = P + i * PP + h * 8;
    const int nks = K >> 4;      // 4 or 16
    const int half = nks >> 1;   // even
    const int last = nks - 1;
    vu4x a0[TI][3], a1[TI][3];
    auto read_a = [&](int ks, vu4x (&a)[TI][3]) {
#pragma unroll
      for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) a[ti][pl] = *reinterpret_cast<const vu4x*>(ap + ti * 32 * PP + pl * PL + ks * 16);
    };
    read_a(0, a0);
    __builtin_amdgcn_s_setprio(RNB_P3S_MM_PRIO);
    for (int s4 = 0; s4 < nks; s4 += 4) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int st = s4 + q;
        vu4x (&ac)[TI][3] = (q & 1) ? a1 : a0;
        vu4x (&an)[TI][3] = (q & 1) ? a0 : a1;
        if (st + 1 == half) { __builtin_amdgcn_s_setprio(RNB_P3S_EP_PRIO); before_b(); __builtin_amdgcn_s_setprio(RNB_P3S_MM_PRIO); }
        read_a(min(st + 1, last), an);
        if (st + 1 == half) after_a();
        __builtin_amdgcn_sched_barrier(0);
        if (st == 0) x3_mfma<TI, 1, true>(ac, b[q], acc);
        else x3_mfma<TI, 1, false>(ac, b[q], acc);
        __builtin_amdgcn_sched_barrier(0);
        if (st + 4 < nks) x3_load_b<1>(W3, nks, n0, st + 4, lane, b[q]);
      }
    }
    __builtin_amdgcn_s_setprio(RNB_P3S_EP_PRIO);
  }
};

template <bool SAVE, int TI>
__global__ __launch_bounds__(512, TI == 1 ? 2 : 1) void fused_forward_p3_kernel(FusedFwdArgs g) {
  constexpr int FT = 32 * TI, NT = 512, NW = 8;
  constexpr int PPLANE = FT * PP;    // elements of one plane
  __shared__ __attribute__((aligned(16))) x3raw P[3 * PPLANE];   // 101,376 B (TI = 2) / 50,688 B (TI = 1)
  __shared__ float E[FT * FEP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
  const int64_t row0 = (int64_t)blockIdx.x * FT;
  const int n0 = wave * 32;

  // ---- positional encoding of the tile ---------------------------------------------------------------------
  {
    constexpr int PARTS = NT / FT;
    const int p = tid % FT, part = tid / FT;
    const int64_t row = row0 + p;
    float x[3] = {0.f, 0.f, 0.f};
    if (row < g.M) {
      if (g.grid.on) {
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
    x3raw* xr = P + p * PP;
    float* er = E + p * FEP;
    if (part == 0) {
#pragma unroll
      for (int d = 0; d < 3; ++d) { p3_put<PPLANE>(xr, d, x[d]); er[d] = x[d]; }
      for (int c = g.pe; c < g.Ep; ++c) { xr[c] = 0; xr[c + PPLANE] = 0; xr[c + 2 * PPLANE] = 0; }
      if (SAVE) {
        g.x4[row * 4] = x[0]; g.x4[row * 4 + 1] = x[1]; g.x4[row * 4 + 2] = x[2]; g.x4[row * 4 + 3] = 0.f;
      }
    }
    for (int k = part; k < g.multires; k += PARTS) {
      const float f = (float)(1 << k);
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        float sn, co;
        sincosf(x[d] * f, &sn, &co);
        const int c = 3 + 6 * k + d;
        p3_put<PPLANE>(xr, c, sn); p3_put<PPLANE>(xr, c + 3, co);
        er[c] = sn; er[c + 3] = co;
      }
    }
  }
  __syncthreads();
  if (SAVE) {   // e (fp32): the PE columns live in E, the padding is zero
    for (int idx = tid; idx < FT * g.Ep; idx += NT) {
      const int r = idx / g.Ep, c = idx - r * g.Ep;
      g.e[(row0 + r) * g.Ep + c] = c < g.pe ? E[r * FEP + c] : 0.f;
    }
  }

  const int h = lane >> 5, cl = lane & 31;
  v16f acc[TI][1];
  for (int l = 0; l < g.nh; ++l) {
    layer_mma_p3<TI>(P, g.w3 + 3 * g.w_off[l], g.Kp[l], n0, lane, acc);
    lds_barrier();   // every wave has finished reading the input activations (the tile is updated in place)
    const float* bias = g.packed + g.b_off[l];
    const BufRsrc ra = tile_rsrc(SAVE ? g.a[l] + (size_t)row0 * FH : nullptr, FT * FH * 4);
    const BufRsrc rD = tile_rsrc(SAVE ? g.D[l] + (size_t)row0 * FH : nullptr, FT * FH * 4);
    const BufRsrc rg = tile_rsrc((SAVE && g.gz_last) ? g.gz_last + (size_t)row0 * FH : nullptr, FT * FH * 4);
    const int n_real = g.n_real[l];
    const bool pe_tail = (l + 1 == g.skip);
    const bool last = (l + 1 == g.nh);
    const int col = n0 + cl;
    const float bc = bias[col];
    const float ws = (SAVE && last && g.gz_last) ? g.packed[g.wsdf_off + col] : 0.f;
    const bool tile_full = n0 + 32 <= n_real;   // wave-uniform
    const unsigned voff = (unsigned)(4 * h * FH + col) * 4u;
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) {
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const int rowc = ti * 32 + (r & 3) + 8 * (r >> 2);
        const int row = rowc + 4 * h;
        vf2 a, D;
        if constexpr (SAVE) softplus_aD(vf2{acc[ti][0][r] + bc, acc[ti][0][r + 1] + bc}, a, D);
        else a = softplus_a(vf2{acc[ti][0][r] + bc, acc[ti][0][r + 1] + bc});
        if (!tile_full && col >= n_real) {
          const bool pe_col = pe_tail && col < n_real + g.pe;
          a = vf2{pe_col ? E[row * FEP + (col - n_real)] : 0.f, pe_col ? E[(row + 1) * FEP + (col - n_real)] : 0.f};
          D = vf2{0.f, 0.f};
        }
        p3_put2<PPLANE>(P, row * PP + col, a.x, a.y);
        if (SAVE) {
          bstore(ra, voff, rowc * FH * 4, a.x);
          bstore(ra, voff, (rowc + 1) * FH * 4, a.y);
          bstore(rD, voff, rowc * FH * 4, D.x);
          bstore(rD, voff, (rowc + 1) * FH * 4, D.y);
          if (last && g.gz_last) {
            bstore(rg, voff, rowc * FH * 4, ws * D.x);
            bstore(rg, voff, (rowc + 1) * FH * 4, ws * D.y);
          }
        }
      }
    }
    lds_barrier();   // the new activations are visible to every wave
  }

  // ---- sdf head -----------------------------------------------------------------------------------------------
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
      for (int u = 0; u < 4; ++u) s = fmaf(p3_get<PPLANE>(P, row * PP + lane + 64 * u), w[u], s);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      if (lane == 0) {
        const float v = (s + bs) / g.scale;
        if (!g.grid.on) g.sdf[row0 + row] = v;
        else if (row0 + row < g.M) g.sdf[row0 + row] = v * g.grid.out_scale;
      }
    }
  }
  // ---- feature head ---------------------------------------------------------------------------------------------
  if (g.with_feat) {
    layer_mma_p3<TI>(P, g.w3 + 3 * g.wf_off, FH, n0, lane, acc);
    const float* bias = g.packed + g.bf_off;
    const BufRsrc rc = tile_rsrc(g.cin + (size_t)row0 * g.Cinp, FT * g.Cinp * 4);
    const unsigned rowb = (unsigned)g.Cinp * 4u;
    const int col = n0 + cl;
    if (col < g.F) {
      const float bc = bias[col];
      const unsigned voff = (unsigned)(4 * h) * rowb + (unsigned)col * 4u;
#pragma unroll
      for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rowc = ti * 32 + (r & 3) + 8 * (r >> 2);
          bstore(rc, voff, rowc * rowb, acc[ti][0][r] + bc);
        }
    }
  }
}

template <bool SAVE>
__global__ __launch_bounds__(512, 1) void fused_forward_p3s_kernel(FusedFwdArgs g) {
  constexpr int TI = 2;
  constexpr int FT = 32 * TI, NT = 512, NW = 8;
  constexpr int PPLANE = FT * PP;    // elements of one plane
  __shared__ __attribute__((aligned(16))) x3raw P[3 * PPLANE];   // 101,376 B (TI = 2) / 50,688 B (TI = 1)
  __shared__ float E[FT * FEP];
  __shared__ P3Sync sy;
  if (threadIdx.x < 8) reinterpret_cast<int*>(&sy)[threadIdx.x] = 0;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
  const int64_t row0 = (int64_t)blockIdx.x * FT;
  const int n0 = wave * 32;

  // ---- positional encoding of the tile ---------------------------------------------------------------------
  {
    constexpr int PARTS = NT / FT;
    const int p = tid % FT, part = tid / FT;
    const int64_t row = row0 + p;
    float x[3] = {0.f, 0.f, 0.f};
    if (row < g.M) {
      if (g.grid.on) {
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
    x3raw* xr = P + p * PP;
    float* er = E + p * FEP;
    if (part == 0) {
#pragma unroll
      for (int d = 0; d < 3; ++d) { p3_put<PPLANE>(xr, d, x[d]); er[d] = x[d]; }
      for (int c = g.pe; c < g.Ep; ++c) { xr[c] = 0; xr[c + PPLANE] = 0; xr[c + 2 * PPLANE] = 0; }
      if (SAVE) {
        g.x4[row * 4] = x[0]; g.x4[row * 4 + 1] = x[1]; g.x4[row * 4 + 2] = x[2]; g.x4[row * 4 + 3] = 0.f;
      }
    }
    for (int k = part; k < g.multires; k += PARTS) {
      const float f = (float)(1 << k);
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        float sn, co;
        sincosf(x[d] * f, &sn, &co);
        const int c = 3 + 6 * k + d;
        p3_put<PPLANE>(xr, c, sn); p3_put<PPLANE>(xr, c + 3, co);
        er[c] = sn; er[c + 3] = co;
      }
    }
  }
  __syncthreads();
  if (SAVE) {   // e (fp32): the PE columns live in E, the padding is zero
    for (int idx = tid; idx < FT * g.Ep; idx += NT) {
      const int r = idx / g.Ep, c = idx - r * g.Ep;
      g.e[(row0 + r) * g.Ep + c] = c < g.pe ? E[r * FEP + c] : 0.f;
    }
  }

  const int h = lane >> 5, cl = lane & 31;
  v16f acc[TI][1];
  const int grp = wave >> 2;   // 0: group A (columns 0..127), 1: group B
  P3sMma<TI> mm;
  for (int l = 0; l < g.nh; ++l) {
    mm.request(g.w3 + 3 * g.w_off[l], g.Kp[l], n0, lane);
    if (l > 0) p3_wait(&sy.WA, 4 * l);              // the A columns of layer l - 1 are written
    if (grp == 1) p3_wait(&sy.RAa, 4 * (l + 1));     // B runs half a layer behind A
    mm.run(P, g.w3 + 3 * g.w_off[l], g.Kp[l], n0, lane, acc,
           [&]() { if (l > 0) p3_wait(&sy.WB, 4 * l); },                       // the B columns of layer l - 1 are written
           [&]() { p3_signal(grp == 0 ? &sy.RAa : &sy.RAb, lane); });           // this wave has read the A columns
    p3_signal(&sy.RB, lane);                                                    // ... and the B columns
    // write-after-read: this group's columns are overwritten in place once EVERY wave has read them (layer 0 reads the
    // PE columns 0..63, inside A's range, during its whole loop)
    if (grp == 0) {
      p3_wait(&sy.RAa, 4 * (l + 1));
      p3_wait(&sy.RAb, 4 * (l + 1));
      if (l == 0) p3_wait(&sy.RB, 8);
    } else {
      p3_wait(&sy.RB, 8 * (l + 1));
    }
    const float* bias = g.packed + g.b_off[l];
    const BufRsrc ra = tile_rsrc(SAVE ? g.a[l] + (size_t)row0 * FH : nullptr, FT * FH * 4);
    const BufRsrc rD = tile_rsrc(SAVE ? g.D[l] + (size_t)row0 * FH : nullptr, FT * FH * 4);
    const BufRsrc rg = tile_rsrc((SAVE && g.gz_last) ? g.gz_last + (size_t)row0 * FH : nullptr, FT * FH * 4);
    const int n_real = g.n_real[l];
    const bool pe_tail = (l + 1 == g.skip);
    const bool last = (l + 1 == g.nh);
    const int col = n0 + cl;
    const float bc = bias[col];
    const float ws = (SAVE && last && g.gz_last) ? g.packed[g.wsdf_off + col] : 0.f;
    const bool tile_full = n0 + 32 <= n_real;   // wave-uniform
    const unsigned voff = (unsigned)(4 * h * FH + col) * 4u;
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) {
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const int rowc = ti * 32 + (r & 3) + 8 * (r >> 2);
        const int row = rowc + 4 * h;
        vf2 a, D;
        if constexpr (SAVE) softplus_aD(vf2{acc[ti][0][r] + bc, acc[ti][0][r + 1] + bc}, a, D);
        else a = softplus_a(vf2{acc[ti][0][r] + bc, acc[ti][0][r + 1] + bc});
        if (!tile_full && col >= n_real) {
          const bool pe_col = pe_tail && col < n_real + g.pe;
          a = vf2{pe_col ? E[row * FEP + (col - n_real)] : 0.f, pe_col ? E[(row + 1) * FEP + (col - n_real)] : 0.f};
          D = vf2{0.f, 0.f};
        }
        p3_put2<PPLANE>(P, row * PP + col, a.x, a.y);
        if (SAVE) {
          bstore(ra, voff, rowc * FH * 4, a.x);
          bstore(ra, voff, (rowc + 1) * FH * 4, a.y);
          bstore(rD, voff, rowc * FH * 4, D.x);
          bstore(rD, voff, (rowc + 1) * FH * 4, D.y);
          if (last && g.gz_last) {
            bstore(rg, voff, rowc * FH * 4, ws * D.x);
            bstore(rg, voff, (rowc + 1) * FH * 4, ws * D.y);
          }
        }
      }
    }
    p3_signal(grp == 0 ? &sy.WA : &sy.WB, lane);    // this wave's columns of layer l are written
  }
  __syncthreads();   // the heads read every column

  // ---- sdf head -----------------------------------------------------------------------------------------------
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
      for (int u = 0; u < 4; ++u) s = fmaf(p3_get<PPLANE>(P, row * PP + lane + 64 * u), w[u], s);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      if (lane == 0) {
        const float v = (s + bs) / g.scale;
        if (!g.grid.on) g.sdf[row0 + row] = v;
        else if (row0 + row < g.M) g.sdf[row0 + row] = v * g.grid.out_scale;
      }
    }
  }
  // ---- feature head ---------------------------------------------------------------------------------------------
  if (g.with_feat) {
    layer_mma_p3<TI>(P, g.w3 + 3 * g.wf_off, FH, n0, lane, acc);
    const float* bias = g.packed + g.bf_off;
    const BufRsrc rc = tile_rsrc(g.cin + (size_t)row0 * g.Cinp, FT * g.Cinp * 4);
    const unsigned rowb = (unsigned)g.Cinp * 4u;
    const int col = n0 + cl;
    if (col < g.F) {
      const float bc = bias[col];
      const unsigned voff = (unsigned)(4 * h) * rowb + (unsigned)col * 4u;
#pragma unroll
      for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rowc = ti * 32 + (r & 3) + 8 * (r >> 2);
          bstore(rc, voff, rowc * rowb, acc[ti][0][r] + bc);
        }
    }
  }
}

// ---- RNB_VARIANT_X3: the split weight mirror ----------------------------------------------------------------
constexpr int kMaxX3 = 4 * RNB_MAX_LIN + 2;
struct X3Entry { long long off; int N, K, unit_begin; };
struct X3Table { int n, total_units; X3Entry e[kMaxX3]; };
// one thread per 16-byte unit of one plane-triple: W[32 nt + c][16 ks + 8 h .. +8] -> hi, mid, lo
__global__ void x3_pack_kernel(const float* __restrict__ src, X3Table t, x3raw* __restrict__ dst) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= t.total_units) return;
  int ei = 0;
  while (ei + 1 < t.n && u >= t.e[ei + 1].unit_begin) ++ei;
  const X3Entry en = t.e[ei];
  const int lu = u - en.unit_begin;            // fragment lu / 64, lane lu % 64
  const int frag = lu >> 6, lane = lu & 63;
  const int nks = en.K >> 4;
  const int nt = frag / nks, ks = frag - nt * nks;
  const int c = lane & 31, h = lane >> 5;
  const float* sp = src + en.off + (size_t)(nt * 32 + c) * en.K + ks * 16 + h * 8;
  vu4x hi, mid, lo;
  x3_split8(*reinterpret_cast<const vf4*>(sp), *reinterpret_cast<const vf4*>(sp + 4), hi, mid, lo);
  x3raw* dp = dst + 3 * en.off + ((size_t)frag * 3 * 64 + lane) * 8;
  *reinterpret_cast<vu4x*>(dp) = hi;
  *reinterpret_cast<vu4x*>(dp + 512) = mid;
  *reinterpret_cast<vu4x*>(dp + 1024) = lo;
}
int x3_pack_weights(const Layout& L, float* packed, hipStream_t s) {
  x3raw* dst = reinterpret_cast<x3raw*>(packed + L.total);
  X3Table t;
  t.n = 0;
  t.total_units = 0;
  auto add = [&](long long off, int N, int K) {
    if (off < 0 || N <= 0 || K <= 0) return;
    X3Entry& e = t.e[t.n++];
    e.off = off; e.N = N; e.K = K; e.unit_begin = t.total_units;
    t.total_units += N * K / 8;
  };
  for (int l = 0; l < L.nh; ++l) {
    add(L.hid[l].w_off, L.hid[l].Np, L.hid[l].Kp);
    add(L.hid[l].wT_off, L.hid[l].Kp, L.hid[l].Np);
  }
  if (L.F > 0) {
    add(L.feat.w_off, L.feat.Np, L.feat.Kp);
    add(L.feat.wT_off, L.feat.Kp, L.feat.Np);
  }
  for (int l = 0; l < L.nc; ++l) {   // the albedo network's hidden layers (gemm_rows_x3m_kernel reads them as fragments)
    add(L.col[l].w_off, L.col[l].Np, L.col[l].Kp);
    add(L.col[l].wT_off, L.col[l].Kp, L.col[l].Np);
  }
  hipLaunchKernelGGL(x3_pack_kernel, dim3((unsigned)((t.total_units + 255) / 256)), dim3(256), 0, s, packed, t, dst);
  RNB_CHECK_LAUNCH();
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
  FusedFwdArgs g;
  memset(&g, 0, sizeof(g));
  if (grid) g.grid = *grid;
  g.pts = pts;
  g.M = M;
  g.packed = packed;
  g.w3 = reinterpret_cast<const x3raw*>(packed + L.total);
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
  g.gz_last = need_gz_last ? pb.gz[L.nh - 1] : nullptr;
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
  if (x3 && force_ti == 2 && force_nw == 2) {   // A/B: activations pre-split in LDS, one 8-wave workgroup per CU
    const unsigned blocks = (unsigned)(pb.Mp / 64);
    if (save) hipLaunchKernelGGL((fused_forward_p3s_kernel<true>), dim3(blocks), dim3(512), 0, s, g);
    else hipLaunchKernelGGL((fused_forward_p3s_kernel<false>), dim3(blocks), dim3(512), 0, s, g);
    RNB_CHECK_LAUNCH();
    return RNB_OK;
  }
  // (Measured and not kept: the 32-point form of the pre-split kernel for the sampling passes, and weight fragments four
  // steps ahead in the small-batch kernel: both 60 us per 8,192-point pass like the default — 256 workgroups each stream the
  // whole 3.5 MB of weight planes from L2, 0.9 GB per pass at the ~16 TB/s the L2s deliver for shared rows.)
  if (small) {
    const unsigned blocks = (unsigned)(pb.Mp / 32);
    const bool wide = force_nw ? (force_nw == 2) : (blocks <= 256);   // at most one workgroup per CU
    if (x3) {
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
    if (x3) {
      if (save) hipLaunchKernelGGL((fused_forward_kernel<2, true, 4, true>), dim3(blocks), dim3(256), 0, s, g);
      else hipLaunchKernelGGL((fused_forward_kernel<2, false, 4, true>), dim3(blocks), dim3(256), 0, s, g);
    } else if (save) hipLaunchKernelGGL((fused_forward_kernel<2, true>), dim3(blocks), dim3(256), 0, s, g);
    else hipLaunchKernelGGL((fused_forward_kernel<2, false>), dim3(blocks), dim3(256), 0, s, g);
  }
  RNB_CHECK_LAUNCH();
  return RNB_OK;
}

}  // namespace rnb
