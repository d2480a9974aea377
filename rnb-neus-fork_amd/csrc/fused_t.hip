// "Register-tile" sweeps of the 256-wide SDF network (x3 arithmetic): the activations of a point never leave the
// registers of the wave that owns it.
//
// fused.hip / fused_bwd.hip keep a 64-point tile in LDS and give each of the four waves of a workgroup 64 output
// COLUMNS: every wave then splits the same fp32 rows into bf16 planes (4 x redundant vector work), every layer ends
// in two workgroup barriers, and each 64-point tile streams the whole 393 KB plane mirror of a layer from L2
// (three co-limits of the same size — matrix pipe, vector issue, L2 -> CU weight stream).
//
// Here the product is taken TRANSPOSED: acc[feature][point] = W[feature][k] * act[k][point].  The weights are the
// MFMA's A operand, the activations its B operand, and a wave owns 32 POINTS and all 256 features:
//   * the accumulator of v_mfma_f32_32x32x16_bf16 holds, per lane, ONE point (lane & 31) and 16 features of each
//     32-feature block — which is exactly the shape of the B operand of the next layer (one point per lane, 8 k per
//     16-k step), up to a permutation of k that is folded into the fragment order of the weight mirror
//     (x3_pack_kernel, `t_kfeat`).  Layer output -> activation -> split -> next layer's operand, all in registers:
//     no LDS round trip, no barrier between layers, and every activation is split ONCE;
//   * LDS, free of activations, holds a ring of weight k-steps filled by LDS-DMA (buffer_load ... lds): the mirror of
//     a layer is fetched once per 128-point workgroup (half the L2 -> CU stream per point of the 64-point tiles) and
//     read by the four waves as ready-made fragments (ds_read_b128, conflict-free: 1 KB contiguous per instruction);
//   * one workgroup = 4 waves = one wave per SIMD with the whole register file (128 accumulator + 128 activation
//     registers + fragments); the only synchronisation is one s_barrier per 16-k step for the ring.
//
//   fused_forward_t_kernel   positional encoding + F sweep (+ sdf head, + feature head)
//                            models/embedder.py:40-46, models/fields.py:82-104
#include <type_traits>

#include "fused_common.hip.h"

namespace rnb {

constexpr int TW = 4;                    // waves per workgroup (one per SIMD)
constexpr int TPT = 32 * TW;             // points per workgroup
constexpr int T_PIECE = 1024;            // one plane of one fragment: one LDS-DMA instruction (64 lanes x 16 B)
constexpr int T_SLOT = 8 * 3 * T_PIECE;  // one 16-k step of a 256-row matrix: 8 row blocks x 3 planes
constexpr int T_NSLOT = 4;               // ring depth (k-steps)
constexpr int T_EP = 68;                 // pitch (floats) of the per-point PE copy (conflict-free b128 reads)
constexpr int T_MAXM = RNB_MAX_LIN + 1;  // matrices of one weight stream
constexpr int T_MAXB = 12;               // bias rows kept in LDS (hidden layers + feature head)
#ifndef T_AUX_ST
#define T_AUX_ST 0   // cache policy of the state stores (0: default write-back, 2: non-temporal)
#endif

// feature (k index of the next layer) held by accumulator register r of 32-feature block j in lane half h
__host__ __device__ constexpr int t_kfeat(int j, int r, int h) { return 32 * j + 4 * h + (r & 3) + 8 * (r >> 2); }

// the matrices streamed through the ring, in order; every one has 256 rows and a multiple of 64 columns
struct TStream {
  int nmat;
  int nks[T_MAXM];         // 16-k steps of matrix i (a multiple of 4: every matrix starts at ring slot 0)
  unsigned boff[T_MAXM];   // byte offset of its mirror in the split mirror
};

// One LDS-DMA piece: 64 lanes x 16 bytes from (rs, voff + soff) to LDS bytes [lds_addr, lds_addr + 1024).  Inline
// assembly on purpose: hipcc's wait-count pass treats the builtin form as a store to LDS that may alias every later
// ds_read and puts s_waitcnt vmcnt(0) in front of them — the ring's whole point is that the DMAs of later steps stay in
// flight while the current step is read (their completion is waited for explicitly, counted, before the barrier that
// publishes a slot).
__device__ __attribute__((always_inline)) inline void t_dma16(vu4x rs, unsigned lds_addr, unsigned voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff)
               : "memory");
}
__device__ inline unsigned lds_addr_of(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
// issue side of the ring.  Everything about a request is scalar and known from the position in the (unrolled) code:
// the consumer at step s of a matrix requests step s + 3 — of the same matrix, or of the next one (`nxt`); past the end of
// the stream (`nxt.clamp`) the last step is requested again, into a slot nobody reads any more, so that the vmcnt
// arithmetic of the consumer stays uniform.
struct TRing {
  vu4x rs;           // buffer resource over the split mirror
  unsigned lds;      // LDS byte address of the ring
};
struct TMat {
  unsigned base;     // byte offset of the matrix in the mirror
  int nks;           // its 16-k steps
  int clamp;         // 1: there is no such matrix (request its last step again)
};
__device__ inline TRing t_ring_init(const x3raw* w3, const char* ring) {
  const unsigned long long a = (unsigned long long)w3;
  TRing q;
  q.rs = vu4x{(unsigned)a, (unsigned)(a >> 32) & 0xffffu, 0x7fffffffu, 0x00020000u};
  q.lds = lds_addr_of(ring);
  return q;
}
// this wave's share of k-step ks of matrix m into ring slot `slot`: row blocks 2 wave, 2 wave + 1 (6 pieces of 1 KB)
__device__ __attribute__((always_inline)) inline void t_issue(const TRing& q, const TMat& m, int ks, int slot, int wave, unsigned lane16) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int nt = 2 * wave + u;
    const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((int)(m.base + (unsigned)((nt * m.nks + ks) * 3) * (unsigned)T_PIECE));
    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(q.lds + (unsigned)(slot * T_SLOT + nt * 3 * T_PIECE)));
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) t_dma16(q.rs, dst + pl * T_PIECE, lane16, soff + pl * T_PIECE);
  }
}

// the six terms of one (32 features x 32 points x 16 k) block, small ones first (x3_mfma).  `out`: where the LAST term
// leaves the block (the MFMA's destination need not be its C operand): the last step of a product hands each finished
// block over to the epilogue's accumulator set for free.
template <bool FIRST>
__device__ __attribute__((always_inline)) inline void t_block(const vu4x (&a)[3], const vu4x (&b)[3], v16f& acc, v16f& out) {
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
  constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
  v16f c = acc;
#pragma unroll
  for (int t = 0; t < 6; ++t) {
    if (FIRST && t == 0) {
      const v16f zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(x3bf8, a[PA[t]]), __builtin_bit_cast(x3bf8, b[PB[t]]), zero, 0, 0, 0);
    } else {
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(x3bf8, a[PA[t]]), __builtin_bit_cast(x3bf8, b[PB[t]]), c, 0, 0, 0);
    }
  }
  out = c;
}
// one pair of the activation split (x3_split8, one pair at a time so that it can be spread over the MFMA gaps)
__device__ __attribute__((always_inline)) inline void t_split_pair(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  unsigned uh = x3_pack2(a, b);
  asm("" : "+v"(uh));
  const float ra = a - __builtin_bit_cast(float, uh << 16);
  const float rb = __builtin_fmaf(__builtin_bit_cast(float, uh & 0xffff0000u), -1.f, b);
  unsigned um = x3_pack2(ra, rb);
  asm("" : "+v"(um));
  const float sa = ra - __builtin_bit_cast(float, um << 16);
  const float sb = __builtin_fmaf(__builtin_bit_cast(float, um & 0xffff0000u), -1.f, rb);
  hi = uh;
  mid = um;
  lo = x3_pack2(sa, sb);
}

// ---------------------------------------------------------------------------------------------------------------
// The software pipeline.  A product (layer) of NKS 16-k steps runs k-step-major: step s multiplies the 8 row blocks of
// the weights by the step's B operand P[s & 3] (three bf16 planes of 8 k per lane).  Its accumulators are complete
// only after the last step — so the EPILOGUE of a layer (bias, softplus, saved state, split into the next layer's B
// planes) is cut into 16 TASKS, task k = the 8 values per lane that become step k of the next layer (accumulator block
// k >> 1, registers 8 (k & 1) .. + 8), and the tasks ride in the matrix loop of the NEXT product: task k + 1 inside
// step k (vector instructions in the MFMA gaps; the planes it writes are the ones step k + 1 reads), task 0 inside the
// layer's own last step (block 0 of that step is multiplied first, its accumulator is final after six MFMAs).  Two
// accumulator sets alternate between consecutive products.  Nothing of the epilogue is exposed but the tail of the
// very last layer; the activations exist in fp32 only inside a task.
// Ring protocol, per step s (global step t):
//   blocks 0..3 | s_waitcnt vmcnt: this wave's pieces of step t + 1 have landed | s_barrier: so have everybody's, and
//   everybody has finished reading step t - 1 | request step t + 3 into the slot of t - 1 | blocks 4..7 (whose
//   fragment prefetch already reaches into step t + 1).
// ---------------------------------------------------------------------------------------------------------------
struct TEpi {            // one layer's epilogue (wave-uniform unless stated)
  const float* Bl;       // LDS: bias row of the layer + 4 h (per lane half)
  const float* WS;       // LDS: sdf row + 4 h when this is the last hidden layer (the sdf head rides in its tasks), else a row of zeros
  const float* Ep;       // LDS: PE row of the lane's point (skip connection)
  int n_real, pe;        // real output width; PE width
  bool pe_tail;          // the outputs beyond n_real are the PE columns of the skip connection
  BufRsrc ra, rD;        // SAVE: tile resources of a_l, D_l
  unsigned voff;         // SAVE: the lane's byte offset inside the tile (point row + 4 h)
  int h;
};
struct TTmp {
  float a[8];       // the task's activations (kept for its split, one step later)
};
// state of one activation quad between its slices
struct TAct {
  float z[4], tt[4], pp[4], qq[4], w0[4], w[4], u[4], r[4], lg[4], a[4], D[4];
  vf4 wv;
};

// Values 4 HALF .. 4 HALF + 3 of task K (one accumulator quad q = 2 (K & 1) + HALF of block K >> 1): bias, softplus
// (+ derivative), skip-connection columns, saved state (one 16-byte store per array), sdf head — four independent
// dependency chains advanced in lockstep, cut into four SLICES of ~20 instructions; slice i rides behind block i of the
// half-step (its own scheduling region: 6 MFMAs with the slice's instructions spread over their gaps).  The code rides
// in the wave's OWN matrix loop, where issue is in order: a single chain exposes the latency of every instruction to
// the matrix pipe, and hipcc's group scheduler left to itself bunches a half-step's 110 vector instructions behind a few
// MFMAs (measured: 447 of 768 gaps empty, clumps of 30-140).  The transcendentals close a slice, their consumers open
// the next.  Scalar fp32 math: beside MFMAs a packed fp32 instruction costs more than the two scalar ones it replaces.
template <bool SAVE, int K, int HALF, int SL>
__device__ __attribute__((always_inline)) inline void t_act_slice(TAct& x, TTmp& t, const v16f (&acc)[8], const TEpi& e, float& sacc) {
#ifdef T_EXP_NOACT
  return;
#endif
  constexpr int b = K >> 1, q = 2 * (K & 1) + HALF, r0 = 4 * q;
  constexpr float L2E = 1.44269504088896341f, LN2 = 0.693147180559945309f;
  if constexpr (SL == 0) {
    const vf4 bv = *reinterpret_cast<const vf4*>(e.Bl + 32 * b + 8 * q);
    x.wv = *reinterpret_cast<const vf4*>(e.WS + 32 * b + 8 * q);
#pragma unroll
    for (int c = 0; c < 4; ++c) x.z[c] = acc[b][r0 + c] + bv[c];
#pragma unroll
    for (int c = 0; c < 4; ++c) x.tt[c] = x.z[c] * 100.f;
    float nt[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) nt[c] = -fabsf(x.tt[c]);
#pragma unroll
    for (int c = 0; c < 4; ++c) x.pp[c] = nt[c] * L2E;
#pragma unroll
    for (int c = 0; c < 4; ++c) x.qq[c] = __builtin_fmaf(nt[c], L2E, -x.pp[c]) * LN2;
#pragma unroll
    for (int c = 0; c < 4; ++c) x.w0[c] = __builtin_amdgcn_exp2f(x.pp[c]);
  } else if constexpr (SL == 1) {
#pragma unroll
    for (int c = 0; c < 4; ++c) x.w[c] = __builtin_fmaf(x.w0[c], x.qq[c], x.w0[c]);
#pragma unroll
    for (int c = 0; c < 4; ++c) x.u[c] = 1.f + x.w[c];
    if constexpr (SAVE) {
#pragma unroll
      for (int c = 0; c < 4; ++c) x.r[c] = __builtin_amdgcn_rcpf(x.u[c]);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) x.lg[c] = __builtin_amdgcn_logf(x.u[c]);
  } else if constexpr (SL == 2) {
    float l1p[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float d = x.w[c] - (x.u[c] - 1.f);
      if constexpr (SAVE) l1p[c] = __builtin_fmaf(x.lg[c], LN2, d * x.r[c]);          // (softplus_aD)
      else l1p[c] = __builtin_fmaf(x.lg[c], LN2, __builtin_fmaf(-d, x.w[c], d));      // (softplus_a: no reciprocal)
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) x.a[c] = __builtin_fmaf(l1p[c], 0.01f, fmaxf(x.z[c], 0.f));
    if constexpr (SAVE) {
#pragma unroll
      for (int c = 0; c < 4; ++c) x.D[c] = x.tt[c] >= 0.f ? x.r[c] : x.w[c] * x.r[c];
    }
  } else {
    if constexpr (b >= 6) {
      // Columns beyond the layer's real width (the layer that feeds the skip connection: the PE columns; n_real >= 192 as
      // the PE is at most 64 wide, so only blocks 6 and 7 can hold them).  Branch-free on purpose: a branch here would cut
      // the scheduling region in two and the slice would no longer ride between the MFMAs.
      const int f0 = 32 * b + 8 * q + 4 * opaque_lane(e.h) - e.n_real;   // (rebuilt here: hoisted, these per-lane values spill)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int idx = f0 + c;                                  // >= 0: beyond the real width
        const int ic = idx < 0 ? 0 : (idx > 63 ? 63 : idx);
        const float pv = e.Ep[ic];
        const float v = (e.pe_tail && idx < e.pe) ? pv : 0.f;
        x.a[c] = idx >= 0 ? v : x.a[c];
        if constexpr (SAVE) x.D[c] = idx >= 0 ? 0.f : x.D[c];
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) t.a[4 * HALF + c] = x.a[c];
#ifndef T_EXP_NOSTORE
    if constexpr (SAVE) {
      constexpr int f0 = 32 * b + 8 * q;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4x, vf4{x.a[0], x.a[1], x.a[2], x.a[3]}), e.ra, e.voff, f0 * 4, T_AUX_ST);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4x, vf4{x.D[0], x.D[1], x.D[2], x.D[3]}), e.rD, e.voff, f0 * 4, T_AUX_ST);
    }
#else
    if constexpr (SAVE) sacc += x.D[0] + x.D[1] + x.D[2] + x.D[3];
#endif
    // sdf head (the sdf row, or zeros for the other layers).  Unconditional and pinned: behind a wave-uniform test hipcc
    // sinks all 128 products of a layer to the end of the matrix loop and keeps every activation alive until then.
    sacc = fmaf(x.a[1], x.wv[1], fmaf(x.a[0], x.wv[0], sacc));
    sacc = fmaf(x.a[3], x.wv[3], fmaf(x.a[2], x.wv[2], sacc));
    asm volatile("" : "+v"(sacc));
  }
}
template <bool SAVE, int K, int HALF>
__device__ __attribute__((always_inline)) inline void t_task_act4(TTmp& t, const v16f (&acc)[8], const TEpi& e, float& sacc) {
  TAct x;
  t_act_slice<SAVE, K, HALF, 0>(x, t, acc, e, sacc);
  t_act_slice<SAVE, K, HALF, 1>(x, t, acc, e, sacc);
  t_act_slice<SAVE, K, HALF, 2>(x, t, acc, e, sacc);
  t_act_slice<SAVE, K, HALF, 3>(x, t, acc, e, sacc);
}
// pairs 2 HALF, 2 HALF + 1 of task K: split into the B planes of step K of the next product (two chains in lockstep, three slices)
struct TSplit {
  float ra[2], rb[2];
  unsigned uh[2], um[2];
};
template <int K, int HALF, int SL>
__device__ __attribute__((always_inline)) inline void t_split_slice(TSplit& x, const TTmp& t, vu4x (&P)[4][3]) {
#ifdef T_EXP_NOSPLIT
  return;
#endif
  if constexpr (SL == 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      x.uh[i] = x3_pack2(t.a[4 * HALF + 2 * i], t.a[4 * HALF + 2 * i + 1]);
      asm("" : "+v"(x.uh[i]));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) x.ra[i] = t.a[4 * HALF + 2 * i] - __builtin_bit_cast(float, x.uh[i] << 16);
#pragma unroll
    for (int i = 0; i < 2; ++i) x.rb[i] = __builtin_fmaf(__builtin_bit_cast(float, x.uh[i] & 0xffff0000u), -1.f, t.a[4 * HALF + 2 * i + 1]);
  } else if constexpr (SL == 1) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      x.um[i] = x3_pack2(x.ra[i], x.rb[i]);
      asm("" : "+v"(x.um[i]));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) x.ra[i] = x.ra[i] - __builtin_bit_cast(float, x.um[i] << 16);
#pragma unroll
    for (int i = 0; i < 2; ++i) x.rb[i] = __builtin_fmaf(__builtin_bit_cast(float, x.um[i] & 0xffff0000u), -1.f, x.rb[i]);
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      P[K & 3][0][2 * HALF + i] = x.uh[i];
      P[K & 3][1][2 * HALF + i] = x.um[i];
      P[K & 3][2][2 * HALF + i] = x3_pack2(x.ra[i], x.rb[i]);
    }
  }
}
template <int K, int HALF>
__device__ __attribute__((always_inline)) inline void t_task_split2(const TTmp& t, vu4x (&P)[4][3]) {
  TSplit x;
  t_split_slice<K, HALF, 0>(x, t, P);
  t_split_slice<K, HALF, 1>(x, t, P);
  t_split_slice<K, HALF, 2>(x, t, P);
}

template <int B, int E, class F>
__device__ __attribute__((always_inline)) inline void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}
// One product.  accN: the working accumulators; accO: the FINISHED accumulators of the previous product, which the epilogue
// tasks read (PREV) — and into which this product's blocks are handed over by the last MFMA of its last step (all of the
// previous product's tasks have read their values by then).  One instance of this body serves every 16-step product: two
// instances with the accumulator sets swapped did not fit the instruction cache (the matrix loop then ran at the speed of
// the instruction fetch: every added instruction cost its bytes).  Schedule of the epilogue tasks (each half-step = one
// scheduling region of 24 MFMAs):
//   step s of a PREV product:  activation of task s + 2 (values 0..3 in the first half, 4..7 in the second), split of
//                              task s + 1 (two pairs per half) -> P[(s + 1) & 3], read by step s + 1;
//   last step, has_own:        this layer's own tasks 0 and 1 (block 0 is final, in accO[0], after six MFMAs): activation of task 0 in
//                              the first half; split of task 0 and activation of task 1 in the second.  (Task 1's split
//                              is step 0 of the next product.)
// SAVE: every activation quad issues two 16-byte stores; they sit between the DMAs in the in-order vmcnt queue.
template <int NKS, bool PREV, bool SAVE>
__device__ __attribute__((always_inline)) inline void t_product(v16f (&accN)[8], v16f (&accO)[8], vu4x (&P)[4][3], const char* ring,
                                                                const TRing& rq, const TMat& cur, const TMat& nxt, const TEpi& prev,
                                                                const TEpi& own, bool has_own, TTmp (&tm)[2], float& sacc, int wave,
                                                                int lane) {
  const char* fr = ring + lane * 16;
  const unsigned lane16 = (unsigned)lane * 16u;
  vu4x a[3][3];
  auto rd = [&](int slot, int j, vu4x (&d)[3]) __attribute__((always_inline)) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) d[pl] = *reinterpret_cast<const vu4x*>(fr + slot * T_SLOT + (j * 3 + pl) * T_PIECE);
  };
  rd(0, 0, a[0]);
  rd(0, 1, a[1]);
  __builtin_amdgcn_sched_barrier(0);
  static_for<0, NKS>([&](auto sc) __attribute__((always_inline)) {
    constexpr int s = decltype(sc)::value;
    static_for<0, 2>([&](auto hc) __attribute__((always_inline)) {
      constexpr int half = decltype(hc)::value;
      TAct xa, xo0, xo1;
      TSplit xs, xs1;
      static_for<0, 4>([&](auto jc) __attribute__((always_inline)) {
        constexpr int jl = decltype(jc)::value;      // block inside the half-step = slice index
        constexpr int j = 4 * half + jl;
        constexpr int blk = s * 8 + j;
        constexpr int nj = (j + 2) & 7, ns = s + ((j + 2) >> 3);
        if constexpr (ns < NKS) rd(ns & (T_NSLOT - 1), nj, a[(blk + 2) % 3]);
        if constexpr (s == NKS - 1) t_block<s == 0>(a[blk % 3], P[s & 3], accN[j], accO[j]);
        else t_block<s == 0>(a[blk % 3], P[s & 3], accN[j], accN[j]);
        // the epilogue slices that ride behind this block
        if constexpr (PREV) {
          if constexpr (s + 1 < 16 && jl < 3) t_split_slice<s + 1, half, jl>(xs, tm[(s + 1) & 1], P);
          if constexpr (s + 2 < 16) t_act_slice<SAVE, s + 2, half, jl>(xa, tm[s & 1], accO, prev, sacc);
        }
        if constexpr (s == NKS - 1) {
          if (has_own) {   // (wave-uniform) this layer's own tasks 0 and 1: block 0 is final (in accO[0]) after its six MFMAs
            if constexpr (half == 0) {
              // blocks 1..3: the two activation quads of task 0 (slices 0..3 of each over three blocks)
              if constexpr (jl == 1) { t_act_slice<SAVE, 0, 0, 0>(xo0, tm[0], accO, own, sacc); t_act_slice<SAVE, 0, 1, 0>(xo1, tm[0], accO, own, sacc); }
              if constexpr (jl == 2) { t_act_slice<SAVE, 0, 0, 1>(xo0, tm[0], accO, own, sacc); t_act_slice<SAVE, 0, 1, 1>(xo1, tm[0], accO, own, sacc); }
              if constexpr (jl == 3) {
                t_act_slice<SAVE, 0, 0, 2>(xo0, tm[0], accO, own, sacc); t_act_slice<SAVE, 0, 1, 2>(xo1, tm[0], accO, own, sacc);
                t_act_slice<SAVE, 0, 0, 3>(xo0, tm[0], accO, own, sacc); t_act_slice<SAVE, 0, 1, 3>(xo1, tm[0], accO, own, sacc);
              }
            } else {
              // blocks 4..7: split of task 0 (both halves), the two activation quads of task 1
              if constexpr (jl < 3) { t_split_slice<0, 0, jl>(xs, tm[0], P); t_split_slice<0, 1, jl>(xs1, tm[0], P); }
              t_act_slice<SAVE, 1, 0, jl>(xo0, tm[1], accO, own, sacc);
              t_act_slice<SAVE, 1, 1, jl>(xo1, tm[1], accO, own, sacc);
            }
          }
        }
        if constexpr (ns < NKS) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
        for (int m = 0; m < 6; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, (s == NKS - 1) ? 10 : 5, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (half == 0) {
        // The in-order vmcnt queue behind the DMAs of step t + 1 (requested in the middle of step s - 2), steady state of a
        // SAVE product: 2 + 2 stores, the 6 DMAs of step t + 2, 2 + 2 stores = 14.  The count waited for must never EXCEED
        // the operations really issued since (pieces of step t + 1 would slip through); the schedule above issues at
        // least 12 in step 14 (no activation rides there), 8 in the last step, 16 in steps 0 and 1: 10 everywhere but the
        // last step only ever waits for stores that are two half-steps old.
        constexpr int W = !(SAVE && PREV) ? 6 : (s == NKS - 1 ? 8 : 10);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W) : "memory");
        __builtin_amdgcn_s_barrier();
        if constexpr (s + 3 < NKS) t_issue(rq, cur, s + 3, (s + 3) & (T_NSLOT - 1), wave, lane16);
        else t_issue(rq, nxt, nxt.clamp ? nxt.nks - 1 : s + 3 - NKS, (s + 3) & (T_NSLOT - 1), wave, lane16);
        __builtin_amdgcn_sched_barrier(0);
      }
    });
  });
}

struct TFwdArgs {
  FusedFwdArgs f;
  TStream st;
  unsigned long long* stamps;   // tools/t_bench (RNB_T_STAMP builds): [workgroup][64] shader clocks of wave 0; else unused
};
#ifdef RNB_T_STAMP
#define T_STAMP(i) do { if (tid == 0) ga.stamps[(size_t)blockIdx.x * 64 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define T_STAMP_REAL(i) do { if (tid == 0) ga.stamps[(size_t)blockIdx.x * 64 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define T_STAMP(i) do { } while (0)
#define T_STAMP_REAL(i) do { } while (0)
#endif

#ifdef T_EXP_MV
#define T_LAUNCH_THREADS (128 * TW)
#else
#define T_LAUNCH_THREADS (64 * TW)
#endif
template <bool SAVE>
__global__ __launch_bounds__(T_LAUNCH_THREADS, 1) void fused_forward_t_kernel(TFwdArgs ga) {
  const FusedFwdArgs& g = ga.f;
  const TStream& st = ga.st;
  __shared__ __attribute__((aligned(1024))) char ring[T_NSLOT * T_SLOT];      // 96 KB
  __shared__ __attribute__((aligned(16))) float Esh[TPT * T_EP];              // 34 KB: PE of the tile (skip connection)
  __shared__ __attribute__((aligned(16))) float Bsh[T_MAXB * FH];             // 12 KB: biases
  __shared__ __attribute__((aligned(16))) float WSsh[2 * FH];                 // sdf head row | zeros
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
#ifdef T_EXP_MV
  if (wave >= TW) {   // dummy vector waves: T_EXP_MV instructions per step beside the matrix waves, same barriers
    float v[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
    const float m = ga.f.scale * 1.0001f;
    __syncthreads();
    const int nsteps = 4 + 16 * (ga.f.nh - 1 + (ga.f.with_feat ? 1 : 0));
    for (int st2 = 0; st2 < nsteps; ++st2) {
#pragma unroll
      for (int it = 0; it < T_EXP_MV / 8; ++it)
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = __builtin_fmaf(v[c], m, 0.5f);
      asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
      __builtin_amdgcn_s_barrier();
    }
    if (v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7] == 12345.f) ga.f.sdf[0] = 1.f;
    return;
  }
#endif
  const int p = lane & 31, h = lane >> 5;
  const int64_t row0 = (int64_t)blockIdx.x * TPT;
  const int prow = wave * 32 + p;          // the lane's point inside the tile
  const int64_t row = row0 + prow;

  T_STAMP(0);
  T_STAMP_REAL(60);
  // ---- the weight stream starts first: steps 0..2 travel during the prologue ----------------------------------------
  const TRing rq = t_ring_init(g.w3, ring);
  {
    const TMat m0 = {st.boff[0], st.nks[0], 0};
#pragma unroll
    for (int t = 0; t < 3; ++t) t_issue(rq, m0, t, t, wave, (unsigned)lane * 16u);
  }

  // ---- biases and the sdf row -> LDS ---------------------------------------------------------------------------
  for (int l = 0; l < g.nh; ++l) Bsh[l * FH + tid] = g.packed[g.b_off[l] + tid];
  if (g.with_feat) Bsh[g.nh * FH + tid] = tid < g.F ? g.packed[g.bf_off + tid] : 0.f;
  WSsh[tid] = g.packed[g.wsdf_off + tid];
  WSsh[FH + tid] = 0.f;

  // ---- positional encoding of the lane's point: Esh[prow][0:64] = [x, sin(2^k x), cos(2^k x)], zero padded ------------
  float* er = Esh + prow * T_EP;
  {
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
    if (h == 0) {
      er[0] = x[0]; er[1] = x[1]; er[2] = x[2];
      for (int c = g.pe; c < 64; ++c) er[c] = 0.f;
      if (SAVE) *reinterpret_cast<vf4*>(g.x4 + row * 4) = vf4{x[0], x[1], x[2], 0.f};
    }
    for (int k = h; k < g.multires; k += 2) {   // the two lanes of a point share the frequencies
      const float f = (float)(1 << k);
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        float s, co;
        sincosf(x[d] * f, &s, &co);
        const int c = 3 + 6 * k + d;
        er[c] = s;
        er[c + 3] = co;
      }
    }
  }
  __syncthreads();   // (also: steps 0..2 of the weight stream have landed)
  if (SAVE) {   // e is an operand of the backward (dW of layer 0) and of the R sweep: [points][64]
    for (int idx = tid; idx < TPT * 16; idx += 64 * TW) {
      const int r = idx >> 4, c4 = idx & 15;
      *reinterpret_cast<vf4*>(g.e + (row0 + r) * 64 + c4 * 4) = *reinterpret_cast<const vf4*>(Esh + r * T_EP + c4 * 4);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (keeps the ring's vmcnt arithmetic exact from the first step on)
  }

  // ---- B planes of layer 0's four steps: the PE in t_kfeat order -------------------------------------------------
  vu4x P[4][3];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const vf4 v0 = *reinterpret_cast<const vf4*>(er + 32 * (s >> 1) + 16 * (s & 1) + 4 * h);
    const vf4 v1 = *reinterpret_cast<const vf4*>(er + 32 * (s >> 1) + 16 * (s & 1) + 8 + 4 * h);
    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
      unsigned uh, um, ul;
      t_split_pair(v[2 * pr], v[2 * pr + 1], uh, um, ul);
      P[s][0][pr] = uh; P[s][1][pr] = um; P[s][2][pr] = ul;
    }
  }

  T_STAMP(1);
  v16f accN[8], accO[8];
  float sacc = 0.f;
  const unsigned voff = (unsigned)(prow * FH + 4 * h) * 4u;
  const int nprod = g.nh + (g.with_feat ? 1 : 0);
  auto mat = [&](int i, bool clamp) __attribute__((always_inline)) {
    TMat m = {(unsigned)__builtin_amdgcn_readfirstlane((int)st.boff[i]), __builtin_amdgcn_readfirstlane(st.nks[i]), clamp ? 1 : 0};
    return m;
  };
  auto epi = [&](int l) __attribute__((always_inline)) {   // epilogue of hidden layer l
    TEpi e;
    e.Bl = Bsh + l * FH + 4 * h;
    e.WS = WSsh + ((l + 1 == g.nh) ? 0 : FH) + 4 * h;
    e.Ep = er;
    e.n_real = g.n_real[l];
    e.pe = g.pe;
    e.pe_tail = (l + 1 == g.skip);
    e.ra = tile_rsrc(SAVE ? g.a[l] + (size_t)row0 * FH : nullptr, TPT * FH * 4);
    e.rD = tile_rsrc(SAVE ? g.D[l] + (size_t)row0 * FH : nullptr, TPT * FH * 4);
    e.voff = voff;
    e.h = h;
    return e;
  };
  TEpi eprev = epi(0), eown = epi(0);
  TTmp tm[2];
  // product 0 (layer 0, 4 steps): no previous tasks; its own tasks 0 and 1 ride in its last step
  t_product<4, false, SAVE>(accN, accO, P, ring, rq, mat(0, false), mat(nprod > 1 ? 1 : 0, nprod <= 1), eprev, eown, true, tm, sacc, wave, lane);
  T_STAMP(2);
  for (int i = 1; i < nprod; ++i) {
    eprev = eown;
    const bool hidden = i < g.nh;
    if (hidden) eown = epi(i);
    t_product<16, true, SAVE>(accN, accO, P, ring, rq, mat(i, false), mat(i + 1 < nprod ? i + 1 : i, i + 1 >= nprod), eprev, eown, hidden, tm,
                              sacc, wave, lane);
    T_STAMP(2 + i);
  }
  if (!g.with_feat) {
    // ---- exposed tail: tasks 2..15 of the last hidden layer (no product follows: activation, saved state, sdf head)
    static_for<2, 16>([&](auto kc) __attribute__((always_inline)) {
      constexpr int k = decltype(kc)::value;
      t_task_act4<SAVE, k, 0>(tm[0], accO, eown, sacc);
      t_task_act4<SAVE, k, 1>(tm[0], accO, eown, sacc);
    });
  } else {
    // ---- feature head: rows 1.. of the output layer, written into the albedo network's input --------------------
    const BufRsrc rc = tile_rsrc(g.cin + (size_t)row0 * g.Cinp, (unsigned)TPT * g.Cinp * 4);
    const unsigned vc = (unsigned)(prow * g.Cinp + 4 * h) * 4u;
    const float* Bf = Bsh + g.nh * FH + 4 * h;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int f0 = 32 * j + 8 * q;
        const vf4 bv = *reinterpret_cast<const vf4*>(Bf + f0);
        const vf4 o = {accO[j][4 * q] + bv[0], accO[j][4 * q + 1] + bv[1], accO[j][4 * q + 2] + bv[2], accO[j][4 * q + 3] + bv[3]};
        if (f0 + 4 * h < g.F) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4x, o), rc, vc, f0 * 4, T_AUX_ST);
      }
  }
  // ---- sdf head: row 0 of the output layer (models/fields.py:104, :106-108); its dot product rode in the tasks ----
  sacc += __shfl_xor(sacc, 32, 64);
  if (h == 0) {
    const float v = (sacc + g.packed[g.bsdf_off]) / g.scale;
    if (!g.grid.on) g.sdf[row] = v;
    else if (row < g.M) g.sdf[row] = v * g.grid.out_scale;   // the volume has exactly M entries
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the over-requested steps have landed before the LDS is released
  T_STAMP(40);
  T_STAMP_REAL(61);
}

bool fused_t_supported(const Layout& L) {
  if (!is_x3(L) || !fused_supported(L)) return false;
  if (L.Ep != 64 || L.hid[0].Kp != 64) return false;
  if (L.nh + 1 > T_MAXB || L.nh < 2 || L.nh > 12) return false;
  for (int l = 0; l < L.nh; ++l)
    if (L.hid[l].N < 192) return false;   // (the skip-connection columns live in the last two 32-feature blocks)
  if (L.F > 0 && (L.feat.Np != FH || L.feat.Kp != FH || (L.F & 3))) return false;
  return true;
}

int fused_forward_t(const Layout& L, const float* packed, const float* pts, int64_t M, PointBufs& pb, bool save,
                    bool need_feat, bool need_gz_last, hipStream_t s, const GridGen* grid) {
  if (need_gz_last) RNB_FAIL(RNB_E_INVALID, "register-tile forward: the fused reverse sweep seeds itself (no gz_last)");
  TFwdArgs ga;
  memset(&ga, 0, sizeof(ga));
  FusedFwdArgs& g = ga.f;
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
    ga.st.nks[l] = L.hid[l].Kp / 16;
    ga.st.boff[l] = (unsigned)(6 * L.hid[l].w_off);
  }
  ga.st.nmat = L.nh;
  if (need_feat) {
    ga.st.nks[L.nh] = L.feat.Kp / 16;
    ga.st.boff[L.nh] = (unsigned)(6 * L.feat.w_off);
    ga.st.nmat = L.nh + 1;
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
  double fl = 0;
  for (int l = 0; l < L.nh; ++l) fl += 2.0 * (double)M * L.hid[l].N * L.hid[l].K;
  fl += 2.0 * (double)M * L.H;
  if (need_feat) fl += 2.0 * (double)M * L.F * L.H;
  ProfScope prof(fl, s, save ? "F_sweep(save)" : "F_sweep(forward_only)");
  const unsigned blocks = (unsigned)(pb.Mp / TPT);   // (Mp is a multiple of 128)
  if (save) hipLaunchKernelGGL((fused_forward_t_kernel<true>), dim3(blocks), dim3(64 * TW), 0, s, ga);
  else hipLaunchKernelGGL((fused_forward_t_kernel<false>), dim3(blocks), dim3(64 * TW), 0, s, ga);
  RNB_CHECK_LAUNCH();
  return RNB_OK;
}

}  // namespace rnb
