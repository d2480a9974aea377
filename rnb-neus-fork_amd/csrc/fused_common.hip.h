// Shared device helpers of the fused SDF-network sweeps (fused.hip, fused_bwd.hip).
#pragma once
#include <stdlib.h>

#include "gemm.hip.h"
#include "rnb_internal.h"

namespace rnb {

constexpr int FH = 256;       // hidden width of the fused path
constexpr int FP = FH + 4;    // LDS pitch of the activation tile
constexpr int FEP = 40;       // LDS pitch of the positional-encoding copy kept for the skip connection

// arguments of the forward sweep kernels (fused.hip, sweep_mv.hip)
struct FusedFwdArgs {
  const float* pts;     // [M,3]
  int64_t M;
  const float* packed;
  const x3raw* w3;      // RNB_VARIANT_X3: split mirror of the weight matrices (matrix at 3 x its float offset;
                        // the fp16 mirror of the x2h kernels: at 2 x)
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
  const H2Tab* h2tab;   // RNB_VARIANT_X2H: scales of the fp16 mirror's matrices (hidden layer l: id l, feature head: id nh)
  unsigned* smax;       // SAVE (render forward only): PointBufs::smax, grown by the tile maxima of e, a_l and the features; or nullptr
};

// ---- x2h: per-tile scale of the operand tile in LDS -----------------------------------------------------------------
// The tile holds its values times a power of two: kH2ActScale (2^6, the round-4 constant: results unchanged) while every
// value of the tile stays below kH2ActLimit, else the power of two that puts the tile's maximum in [2^13, 2^14) — chosen
// per tile and layer from the values just written, so NO activation, network input or Jacobian row is out of range.
// Writers store times kH2ActScale and keep the maximum of what they wrote (one v_max3 per pair of values); a wave in which
// any lane reached the limit (one ballot) raises a flag in LDS before the barrier that ends the layer, every wave reads the
// flag behind it — that is all the common case costs.  Only when the flag is up do the waves exchange their maxima, every
// thread rescales the elements it wrote (an exact multiplication) and the tile's maximum is left in PointBufs::smax for the
// weight-gradient kernel (whose state operands then take their scale from it): a path a sane network never takes.
constexpr float kH2ActLimit = 256.f;   // 2^8 * 2^6 = 2^14: the first maximum that leaves [.., 2^14)
__device__ inline float wave_max(float m) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  return m;
}
template <int NW>
__device__ inline float tile_max(const float* wm) {
  float m = wm[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) m = fmaxf(m, wm[w]);
  return m;
}
// maximum of the NW per-wave values of a layer -> (tile maximum, the scale that puts it in [2^13, 2^14), 1 / scale): tiles of
// loss adjoints, whose magnitude nothing bounds a priori, are always scaled from their own maximum
template <int NW>
__device__ inline float tile_scale(const float* wm, float& s, float& inv_s) {
  const float m = tile_max<NW>(wm);
  x2h_dyn_scale(__builtin_bit_cast(unsigned, m), s, inv_s);
  return m;
}
// m = max(m, a, b) on the BIT PATTERNS of two floats, compared as SIGNED integers: for values >= +0 that is the float order, a
// negative value reads as a negative integer and never wins (callers whose values can be negative keep those apart), a NaN
// with the sign bit clear reads as a large integer and raises the flag, harmlessly.  One v_max3_i32 for two values.
// (Written as asm: from the nested max the compiler made a float maximum plus an integer one.)
__device__ inline void h2_track2(int& m, float a, float b) {
  asm("v_max3_i32 %0, %1, %2, %3" : "=v"(m) : "v"(m), "v"(a), "v"(b));
}
// the waves of a workgroup agree on whether the tile just written needs a smaller scale: `mine` = this thread's maximum;
// flag = one LDS word (zero unless raised; reset by the slow path)
__device__ inline void h2_raise_flag(float mine, int* flag, int lane) {
  if (__builtin_amdgcn_ballot_w64(mine >= kH2ActLimit) != 0 && lane == 0) *reinterpret_cast<volatile int*>(flag) = 1;
}
__device__ inline bool h2_flag_up(const int* flag) {   // (expected down: the rescaling path is laid out off the hot path)
  return __builtin_expect(__builtin_amdgcn_readfirstlane(*reinterpret_cast<const volatile int*>(flag)) != 0, 0);
}
// The inverse scales of the mirror's matrices, one per lane (lane id = table id), loaded ONCE at the top of a kernel: a load
// per layer would sit in the in-order vector-memory queue in front of that layer's weight fragments.  h2_iws_at picks a
// layer's value out of the register (v_readlane with a uniform index).
__device__ inline float h2_iws_load(const H2Tab* tab, int lane) { return tab->iws[lane < kH2TabSlots ? lane : 0]; }
__device__ inline float h2_iws_at(float iwsv, int id) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, iwsv), id));
}
// One atomic per tile (and layer): fire and forget.  The result is not used, so the instruction does not return and the wave
// does not wait for it — a "does the slot grow?" load in front of it (round 4) made the committing wave sit out a global round
// trip before every layer's matrix loop, with its partners waiting for it at the next barrier.  A thousand tiles spread over a
// launch put a few atomics per microsecond on a slot: nothing for the L2.
__device__ inline void amax_tile_commit(unsigned* slot, float m) {
  (void)__hip_atomic_fetch_max(slot, __builtin_bit_cast(unsigned, m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One (32*TI) x (32*TJ) output block per wave (TJ = 2 unless stated): C[rows][n0..] = X[rows][K] * W[n][K]^T, K a
// multiple of 32.
// k-permutation inside each 32-k block: lane half h takes k = 32Q + 16h + 4q + c, so every lane streams 64
// contiguous bytes of its weight row per block.
template <int TJ>
__device__ inline void load_b_block(const float* __restrict__ W, int K, int n0, int Q, int lane, vf4 (&b)[TJ][4]) {
  const int j = lane & 31, h = lane >> 5;
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj) {
    const float* p = W + (size_t)(n0 + tj * 32 + j) * K + Q * 32 + h * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) b[tj][q] = *reinterpret_cast<const vf4*>(p + q * 4);
  }
}

// A value the optimiser cannot see through: per-lane offsets derived from it are rebuilt where they are used instead of
// being hoisted out of the layer loop and kept live across the matrix loop (where they spill).
__device__ inline int opaque_lane(int v) { asm volatile("" : "+v"(v)); return v; }

struct NoHook {
  __device__ void operator()() const {}
};

// the 4 k-groups (x 4 MFMA k-steps) of one 32-k block with the block's weight fragments in `b`.
// FIRST: the very first MFMA of every accumulator takes the constant 0 as its C operand (an inline
// constant of the instruction) instead of a zeroed register tile: saves the 16 v_mov per accumulator that
// would otherwise be paid in matrix time at the top of every layer.
template <int TI, bool FIRST = false, int TJ = 2>
__device__ inline void mma_block(const float* __restrict__ X, int Q, int i, int h, const vf4 (&b)[TJ][4],
                                 v16f (&acc)[TI][TJ]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    vf4 a[TI];
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
      a[ti] = *reinterpret_cast<const vf4*>(X + (ti * 32 + i) * FP + Q * 32 + h * 16 + q * 4);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) {
          if (FIRST && q == 0 && c == 0) {
            const v16f zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ti][c], b[tj][q][c], zero, 0, 0, 0);
          } else {
            acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ti][c], b[tj][q][c], acc[ti][tj], 0, 0, 0);
          }
        }
    }
    // keep each k-group's fragment reads behind the previous group's MFMAs (hoisting them all costs
    // registers and, measured, 15 % of the matrix-pipe utilisation)
    __builtin_amdgcn_sched_barrier(0);
  }
}

// C[rows][n0..n0+63] += X[rows][K] * W[n][K]^T for one wave; K a multiple of 64.  Weight fragments are
// prefetched one 32-k block ahead into two alternating register sets (no copies: on gfx950 the fp32 MFMA
// shares the SIMD's issue bandwidth with ordinary vector instructions, so every v_mov in this loop is
// matrix time lost).  `hook` runs once inside the loop, after a block's weight prefetch has been issued:
// the place to issue the epilogue's operand loads (the vector-memory counter retires in order, so loads
// issued before the first weight block would have to land before the first MFMA).  hook_late: run it in
// the second-to-last block instead of the second.
template <int TI, class Hook = NoHook, int TJ = 2>
__device__ inline void layer_mma_nt(const float* __restrict__ X, const float* __restrict__ W, int K, int n0, int lane,
                                    v16f (&acc)[TI][TJ], Hook hook = Hook(), int hook_late = 0) {
  // acc = X W^T (overwritten, not accumulated: the first MFMA of every accumulator starts from the constant 0)
  const int i = lane & 31, h = lane >> 5;
  const int nQ = K / 32;   // even
  const int hookQ = hook_late ? (nQ >= 2 ? nQ - 2 : 0) : (nQ > 2 ? 1 : 0);
  vf4 b0[TJ][4], b1[TJ][4];
  load_b_block<TJ>(W, K, n0, 0, lane, b0);
  load_b_block<TJ>(W, K, n0, 1, lane, b1);
  if (hookQ == 0) hook();
  mma_block<TI, true, TJ>(X, 0, i, h, b0, acc);
  if (2 < nQ) load_b_block<TJ>(W, K, n0, 2, lane, b0);
  if (hookQ == 1) hook();
  mma_block<TI, false, TJ>(X, 1, i, h, b1, acc);
  for (int Q = 2; Q < nQ; Q += 2) {
    load_b_block<TJ>(W, K, n0, Q + 1, lane, b1);
    if (Q == hookQ) hook();
    mma_block<TI, false, TJ>(X, Q, i, h, b0, acc);
    if (Q + 2 < nQ) load_b_block<TJ>(W, K, n0, Q + 2, lane, b0);
    if (Q + 1 == hookQ) hook();
    mma_block<TI, false, TJ>(X, Q + 1, i, h, b1, acc);
  }
}

// Same product with ONE set of weight registers (32 instead of 64): a k-group's fragments are re-requested
// for the next 32-k block as soon as the group's MFMAs have been issued, i.e. the prefetch runs exactly one
// block (4 k-groups) ahead.  L2-resident weights need no more; the 64-point sweeps need the registers.
template <int TI, bool FIRST>
__device__ inline void ring_block(const float* __restrict__ X, const float* __restrict__ p0,
                                  const float* __restrict__ p1, int Q, int Qn, int i, int h, vf4 (&b)[2][4],
                                  v16f (&acc)[TI][2]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    vf4 a[TI];
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
      a[ti] = *reinterpret_cast<const vf4*>(X + (ti * 32 + i) * FP + Q * 32 + h * 16 + q * 4);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) {
          if (FIRST && q == 0 && c == 0) {
            const v16f zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ti][c], b[tj][q][c], zero, 0, 0, 0);
          } else {
            acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ti][c], b[tj][q][c], acc[ti][tj], 0, 0, 0);
          }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    b[0][q] = *reinterpret_cast<const vf4*>(p0 + Qn * 32 + q * 4);
    b[1][q] = *reinterpret_cast<const vf4*>(p1 + Qn * 32 + q * 4);
    __builtin_amdgcn_sched_barrier(0);
  }
}
template <int TI, class Hook = NoHook>
__device__ inline void layer_mma_nt_ring(const float* __restrict__ X, const float* __restrict__ W, int K, int n0,
                                         int lane, v16f (&acc)[TI][2], Hook hook = Hook()) {
  // acc = X W^T (overwritten, like layer_mma_nt)
  const int i = lane & 31, h = lane >> 5;
  const int nQ = K / 32;
  const float* p0 = W + (size_t)(n0 + i) * K + h * 16;
  const float* p1 = p0 + (size_t)32 * K;
  vf4 b[2][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    b[0][q] = *reinterpret_cast<const vf4*>(p0 + q * 4);
    b[1][q] = *reinterpret_cast<const vf4*>(p1 + q * 4);
  }
  if (nQ == 1) hook();
  ring_block<TI, true>(X, p0, p1, 0, nQ > 1 ? 1 : 0, i, h, b, acc);
  for (int Q = 1; Q < nQ; ++Q) {
    const int Qn = Q + 1 < nQ ? Q + 1 : Q;   // the last block re-requests itself (harmless, keeps the loop uniform)
    if (Q == 1) hook();
    ring_block<TI, false>(X, p0, p1, Q, Qn, i, h, b, acc);
  }
}


// ---------------------------------------------------------------------------------------------------------------
// fp32 products on the bf16 matrix pipe ("x3": every fp32 operand as THREE bf16 terms)
// ---------------------------------------------------------------------------------------------------------------
// x = hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): three round-to-nearest bf16 values
// whose sum is x exactly (8 + 8 + 8 mantissa bits; |mid| <= 2^-9 |x|, |lo| <= 2^-18 |x|).  a b is then taken as the six
// terms  a_hi b_hi + a_hi b_mid + a_mid b_hi + a_hi b_lo + a_lo b_hi + a_mid b_mid, accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16.  The three dropped terms are below 2^-26 |a b| together: a quarter of the rounding error
// fp32 itself commits on the product, so the result is an fp32 product in every sense the parity tests can see — at
// 6 x 32 cycles per 16 k (192) against 8 x 64 = 512 for v_mfma_f32_32x32x2_f32, and, unlike the fp32 MFMA, the bf16
// MFMA co-executes with the VALU work of the epilogues (profiles/r02_overlap_probe_*).
// Weights are split once per step into a fragment-ordered mirror (x3_pack_weights): for W [N][K], fragment (nt, ks)
// = rows 32 nt .. +32, k = 16 ks .. +16 is three consecutive 1 KB blocks (hi, mid, lo), each 64 lanes x 16 bytes with
// lane (c, h) = W[32 nt + c][16 ks + 4 h + {0..3, 8..11}] (SDF network; the albedo net's matrices keep 16 ks + 8 h .. + 8).
// Activations stay fp32 in LDS and are split as they are read.
// (buffer loads: per-lane offset lane * 16 in one VGPR, the fragment's offset in the scalar operand, the plane in the
// immediate — plain pointer arithmetic cost five 64-bit vector adds per step in front of the loads)
template <int TJ, int NP = 3>
__device__ inline void x3_load_b(const x3raw* __restrict__ W3, int nks, int n0, int ks, int lane, vu4x (&b)[TJ][NP]) {
  const BufRsrc rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<x3raw*>(W3), 0, 0x4000000, 0x00020000);
  const unsigned voff = (unsigned)lane * 16u;
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj) {
    const unsigned soff = (unsigned)(((n0 >> 5) + tj) * nks + ks) * (NP * 1024u);
#pragma unroll
    for (int pl = 0; pl < NP; ++pl)
      b[tj][pl] = __builtin_bit_cast(vu4x, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff + pl * 1024, 0));
  }
}
template <int TI>
__device__ inline void x3_read_a(const float* __restrict__ xp, int ks, vf4 (&a)[TI][2]) {
#pragma unroll
  for (int ti = 0; ti < TI; ++ti) {
    a[ti][0] = *reinterpret_cast<const vf4*>(xp + ti * 32 * FP + ks * 16);       // k = 16 ks + 4 h + 0..3
    a[ti][1] = *reinterpret_cast<const vf4*>(xp + ti * 32 * FP + ks * 16 + 8);   //     16 ks + 4 h + 8..11  (the mirror's order)
  }
}
template <int TI, int NP = 3>
__device__ inline void x3_split(const vf4 (&raw)[TI][2], vu4x (&a)[TI][NP]) {
#pragma unroll
  for (int ti = 0; ti < TI; ++ti) xn_split8<NP>(raw[ti][0], raw[ti][1], a[ti]);
}
// One pipeline stage: the MFMAs of step s (operands `a`, `b`) with, between them, the split of step s + 1's rows
// (`raw` -> `an`): the bf16 MFMA leaves about four vector-instruction issue slots free per instruction
// (tools/overlap_probe), which is where the 36 TI split instructions go.  Then the raw rows of step s + 2 are
// requested into `raw` (the caller clamps the step index: no branch inside the pipeline, the last re-read is unused).
template <int TI, int TJ, bool FIRST, int NP = 3>
__device__ inline void x3_stage(const float* __restrict__ xp, int ks_next_raw, vf4 (&raw)[TI][2],
                                const vu4x (&a)[TI][NP], vu4x (&an)[TI][NP], const vu4x (&b)[TJ][NP], v16f (&acc)[TI][TJ]) {
  x3_mfma<TI, TJ, FIRST, NP>(a, b, acc);
  x3_split<TI, NP>(raw, an);
  constexpr int NM = (NP == 3 ? 6 : 3) * TI * TJ;
  constexpr int PER = ((NP == 3 ? 36 : 16) * TI + NM - 1) / NM;   // vector instructions behind each MFMA
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
    __builtin_amdgcn_sched_group_barrier(0x002, PER, 0);   // PER VALU
  }
  __builtin_amdgcn_sched_barrier(0);
  x3_read_a<TI>(xp, ks_next_raw, raw);
  __builtin_amdgcn_sched_barrier(0);
}
// acc = X W^T (overwritten) for one wave: rows = the TI 32-row tiles at X, columns n0 .. n0 + 32 TJ.  X: LDS fp32, pitch
// FP.  W3: the layer's x3 mirror.  K a multiple of 32.  Weight fragments run one 16-k step ahead in a second register
// set; the split A operands likewise.  `hook` runs once, inside step 1 (the epilogue's operand loads).
// The weight fragments of steps 0 and 1 are REQUESTED by the caller (`request`) — for the next layer's product before
// the epilogue of this one, so that they land during the epilogue instead of costing an exposed L2 round trip (and, the
// memory counter retiring in order, a wait for the epilogue's stores) at the top of every layer.
// NP = 3: the bf16 scheme (mirror: three planes per fragment); NP = 2: the fp16 scheme (x2h: two planes; X holds the
// activations times kH2ActScale, the mirror the weights times kH2WScale, acc comes out scaled by their product).
template <int TI, int TJ = 2, int NP = 3>
struct X3Mma {
  vu4x b0[TJ][NP], b1[TJ][NP];
  __device__ inline void request(const x3raw* __restrict__ W3, int K, int n0, int lane) {
    x3_load_b<TJ, NP>(W3, K >> 4, n0, 0, lane, b0);
    x3_load_b<TJ, NP>(W3, K >> 4, n0, 1, lane, b1);
  }
  // W3n / Kn / n0n: the product that follows (nullptr: none); its first two weight steps are requested as soon as the
  // registers are free
  template <class Hook = NoHook>
  __device__ inline void run(const float* __restrict__ X, const x3raw* __restrict__ W3, int K, int n0, int lane,
                             v16f (&acc)[TI][TJ], const x3raw* __restrict__ W3n, int Kn, int n0n, Hook hook = Hook()) {
    const int i = lane & 31, h = lane >> 5;
    const float* xp = X + i * FP + h * 4;   // (k order of the SDF mirror: x3_pack_kernel, tperm)
    const int nks = K >> 4;   // even
    vu4x a0[TI][NP], a1[TI][NP];
    vf4 raw[TI][2];
    x3_read_a<TI>(xp, 0, raw);
    x3_split<TI, NP>(raw, a0);
    x3_read_a<TI>(xp, 1, raw);
    __builtin_amdgcn_sched_barrier(0);
    const int last = nks - 1;
    // the matrix loop outranks the other workgroup's epilogue on this SIMD: its MFMAs and the split between them then never
    // queue behind the epilogue's vector work (3.97 -> 3.92 ms per step, two runs each on one box)
    __builtin_amdgcn_s_setprio(1);
    x3_stage<TI, TJ, true, NP>(xp, min(2, last), raw, a0, a1, b0, acc);   // step 0 (splits step 1)
    hook();
    for (int ks = 1; ks + 1 < nks; ks += 2) {   // (nks even: ks + 2 <= last inside the loop)
      x3_load_b<TJ, NP>(W3, nks, n0, ks + 1, lane, b0);
      __builtin_amdgcn_sched_barrier(0);
      x3_stage<TI, TJ, false, NP>(xp, ks + 2, raw, a1, a0, b1, acc);
      x3_load_b<TJ, NP>(W3, nks, n0, ks + 2, lane, b1);
      __builtin_amdgcn_sched_barrier(0);
      x3_stage<TI, TJ, false, NP>(xp, min(ks + 3, last), raw, a0, a1, b0, acc);
    }
    if (W3n) x3_load_b<TJ, NP>(W3n, Kn >> 4, n0n, 0, lane, b0);
    __builtin_amdgcn_sched_barrier(0);
    x3_mfma<TI, TJ, false, NP>(a1, b1, acc);   // the last step: nothing left to split
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(0);
    if (W3n) x3_load_b<TJ, NP>(W3n, Kn >> 4, n0n, 1, lane, b1);
  }
  // The same product over a RANGE of k: weight steps ks0 .. ks0 + cnt of a matrix whose rows hold nks_tot steps, against the
  // first 16 cnt columns of the tile at X (cnt even, >= 2).  ACC: the accumulators are added to instead of overwritten.
  // The product that follows is named by (W3n, nksn_tot, ks0n, n0n).  Used by the albedo network's first layer, whose 320
  // input columns pass through the 256-column tile in two parts.
  __device__ inline void request_at(const x3raw* __restrict__ W3, int nks_tot, int ks0, int n0, int lane) {
    x3_load_b<TJ, NP>(W3, nks_tot, n0, ks0, lane, b0);
    x3_load_b<TJ, NP>(W3, nks_tot, n0, ks0 + 1, lane, b1);
  }
  template <bool ACC, class Hook = NoHook>
  __device__ inline void run_at(const float* __restrict__ X, const x3raw* __restrict__ W3, int nks_tot, int ks0, int cnt, int n0,
                                int lane, v16f (&acc)[TI][TJ], const x3raw* __restrict__ W3n, int nksn_tot, int ks0n, int n0n,
                                Hook hook = Hook()) {
    const int i = lane & 31, h = lane >> 5;
    const float* xp = X + i * FP + h * 4;
    vu4x a0[TI][NP], a1[TI][NP];
    vf4 raw[TI][2];
    x3_read_a<TI>(xp, 0, raw);
    x3_split<TI, NP>(raw, a0);
    x3_read_a<TI>(xp, 1, raw);
    __builtin_amdgcn_sched_barrier(0);
    const int last = cnt - 1;
    __builtin_amdgcn_s_setprio(1);
    x3_stage<TI, TJ, !ACC, NP>(xp, min(2, last), raw, a0, a1, b0, acc);
    hook();
    for (int ks = 1; ks + 1 < cnt; ks += 2) {
      x3_load_b<TJ, NP>(W3, nks_tot, n0, ks0 + ks + 1, lane, b0);
      __builtin_amdgcn_sched_barrier(0);
      x3_stage<TI, TJ, false, NP>(xp, ks + 2, raw, a1, a0, b1, acc);
      x3_load_b<TJ, NP>(W3, nks_tot, n0, ks0 + ks + 2, lane, b1);
      __builtin_amdgcn_sched_barrier(0);
      x3_stage<TI, TJ, false, NP>(xp, min(ks + 3, last), raw, a0, a1, b0, acc);
    }
    if (W3n) x3_load_b<TJ, NP>(W3n, nksn_tot, n0n, ks0n, lane, b0);
    __builtin_amdgcn_sched_barrier(0);
    x3_mfma<TI, TJ, false, NP>(a1, b1, acc);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(0);
    if (W3n) x3_load_b<TJ, NP>(W3n, nksn_tot, n0n, ks0n + 1, lane, b1);
  }
};
template <int TI, int TJ = 2, class Hook = NoHook>
__device__ inline void layer_mma_x3(const float* __restrict__ X, const x3raw* __restrict__ W3, int K, int n0, int lane,
                                    v16f (&acc)[TI][TJ], Hook hook = Hook()) {
  X3Mma<TI, TJ> m;
  m.request(W3, K, n0, lane);
  m.run(X, W3, K, n0, lane, acc, nullptr, 0, 0, hook);
}

// AuxTile<TI, TJ>: one value per accumulator element of the wave's (32 TI) x (32 TJ) block
template <int TI, int TJ = 2>
struct AuxTile {
  float v[TI][TJ][16];
};

// visits the wave's BT x 64 accumulator block: f(tj, ti, r, col, rowc, row) with rowc the lane-independent
// part of the row (compile-time after unrolling) and row = rowc + 4*(lane>>5)
template <int TI, int TJ = 2, class F>
__device__ inline void for_each_acc(int n0, int lane, F f) {
  const int h = lane >> 5, cl = lane & 31;
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj) {
    const int col = n0 + tj * 32 + cl;
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rowc = ti * 32 + (r & 3) + 8 * (r >> 2);
        f(tj, ti, r, col, rowc, rowc + 4 * h);
      }
    }
  }
}

// Same visit, split per 32-column tile into a branch-free body for tiles that lie entirely below `limit`
// (wave-uniform test: n0 and limit are scalars) and a general body for the one tile that may straddle it.
// On gfx950 the fp32 MFMA and ordinary vector instructions exclude each other on a SIMD (tools/overlap_probe),
// so every per-element compare / exec-mask round trip in an epilogue is matrix time lost.
template <int TI, int TJ = 2, class FF, class FS>
__device__ inline void for_each_acc_split(int n0, int lane, int limit, FF fast, FS slow) {
  const int h = lane >> 5, cl = lane & 31;
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj) {
    const int col = n0 + tj * 32 + cl;
    if (n0 + tj * 32 + 32 <= limit) {
#pragma unroll
      for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rowc = ti * 32 + (r & 3) + 8 * (r >> 2);
          fast(tj, ti, r, col, rowc, rowc + 4 * h);
        }
    } else {
#pragma unroll
      for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rowc = ti * 32 + (r & 3) + 8 * (r >> 2);
          slow(tj, ti, r, col, rowc, rowc + 4 * h);
        }
    }
  }
}

// issues the buffer loads of one [BT x 256] tile in accumulator layout (no wait: consumed after the MFMA loop)
template <int TI, int TJ = 2>
__device__ inline void prefetch_tile(const float* base, int64_t row0, int n0, int lane, AuxTile<TI, TJ>& t) {
  const BufRsrc rs = tile_rsrc(base + (size_t)row0 * FH, 32 * TI * FH * 4);
  const int h = lane >> 5;
  for_each_acc<TI, TJ>(n0, lane, [&](int tj, int ti, int r, int col, int rowc, int row) {
    t.v[ti][tj][r] = bload(rs, (unsigned)(4 * h * FH + col) * 4u, rowc * FH * 4);
  });
}

template <int TI>
__device__ inline void zero_acc2(v16f (&acc)[TI][2]) {
#pragma unroll
  for (int a = 0; a < TI; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
}


}  // namespace rnb
