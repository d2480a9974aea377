// fp32 MFMA GEMM building blocks for gfx950 (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fma chain).
//
// One workgroup = 256 threads = 4 wave64 arranged 2(M) x 2(N); block tile 128 x BN (BN = 128 or 256),
// K-step 32; each wave owns a 64 x BN/2 sub-tile = 2 x TN MFMA tiles of 32x32 (16 accumulator registers
// per tile per lane).  Operand tiles are staged through LDS with a register prefetch of the next K-step
// (global loads in flight while the matrix cores run).  fp32 MFMA retires 2 k per 64 cycles per SIMD, so
// operand bandwidth is far from binding; the layouts are chosen for conflict-free ds_read_b128 /
// ds_read_b32.  All tile/wave indices are kept in SGPRs (readfirstlane) so every MFMA runs with full
// EXEC and no per-instruction branching; partially filled column blocks take a separate guarded path.
//
// The reduction index inside a K-step is permuted: within each group of 8 k, lane half h (= lane>>5)
// supplies k = 8q+4h+c for MFMA step c (one 16-byte LDS read feeds 4 MFMAs).  A and B use the same
// permutation, so the product is unchanged up to fp32 summation order.
#pragma once
#include <hip/hip_runtime.h>

namespace rnb {

typedef float v16f __attribute__((ext_vector_type(16)));
// native 4-vector for register staging (HIP's float4 struct is not split by SROA when it sits in an
// array: the staging arrays then live in scratch memory)
typedef float vf4 __attribute__((ext_vector_type(4)));
__device__ inline vf4 make_vf4(float a, float b, float c, float d) {
  vf4 r = {a, b, c, d};
  return r;
}

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LDK = BK + 4;   // pitch (floats) of a k-contiguous tile  [rows][36]

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory counter,
// i.e. it would wait for prefetch loads and fire-and-forget global stores that are still in flight.
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ inline int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// Raw buffer access to one tile of a row-major matrix: resource = tile base + byte size, per-lane 32-bit
// byte offset in a VGPR, wave-uniform byte offset in the scalar operand (no per-access vector address
// arithmetic; out-of-range reads return 0, out-of-range writes are dropped).
#ifndef RNB_AUX_ST
#define RNB_AUX_ST 2   // nt: the saved state is written once and read much later (or once): keep it out of L2's way
#endif
#ifndef RNB_AUX_LD
#define RNB_AUX_LD 2   // nt: epilogue operand tiles are read exactly once
#endif
typedef __amdgpu_buffer_rsrc_t BufRsrc;
typedef float vf2 __attribute__((ext_vector_type(2)));
typedef unsigned vu2 __attribute__((ext_vector_type(2)));
__device__ inline BufRsrc tile_rsrc(const float* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, p ? bytes : 0, 0x00020000);
}
__device__ inline void bstore(BufRsrc r, unsigned voff, unsigned soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, RNB_AUX_ST);
}
__device__ inline float bload(BufRsrc r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, RNB_AUX_LD));
}
__device__ inline vf2 bload2(BufRsrc r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(vf2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}

// ---- global -> register staging ------------------------------------------------------------------
// k-contiguous source: element (r, k) at src[r*ld + k]; tile = ROWS rows from r0, k0..k0+31.
template <int ROWS, bool GUARD>
__device__ inline void load_rows(const float* __restrict__ src, int ld, int r0, int k0, int rmax, int tid,
                                 vf4 (&v)[ROWS / 32]) {
#pragma unroll
  for (int i = 0; i < ROWS / 32; ++i) {
    const int idx = tid + 256 * i;
    const int r = idx >> 3, c4 = idx & 7;
    if constexpr (!GUARD) {
      v[i] = *reinterpret_cast<const vf4*>(src + (size_t)(r0 + r) * ld + k0 + c4 * 4);
    } else {   // branch-free: load from a clamped (valid) row, then select
      const bool ok = r0 + r < rmax;
      const int rr = ok ? r0 + r : rmax - 1;
      const vf4 t = *reinterpret_cast<const vf4*>(src + (size_t)rr * ld + k0 + c4 * 4);
      v[i] = make_vf4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
    }
  }
}
template <int ROWS>
__device__ inline void store_rows(float* __restrict__ T, int tid, const vf4 (&v)[ROWS / 32]) {
#pragma unroll
  for (int i = 0; i < ROWS / 32; ++i) {
    const int idx = tid + 256 * i;
    const int r = idx >> 3, c4 = idx & 7;
    *reinterpret_cast<vf4*>(T + r * LDK + c4 * 4) = v[i];
  }
}
// k-major source: element (k, c) at src[k*ld + c]; tile = 32 k rows from k0, COLS columns from c0.
template <int COLS, bool GUARD>
__device__ inline void load_kmajor(const float* __restrict__ src, int ld, int k0, int c0, int kmax, int cmax,
                                   int tid, vf4 (&v)[COLS / 32]) {
  constexpr int C4 = COLS / 4;   // vf4 per k row
#pragma unroll
  for (int i = 0; i < COLS / 32; ++i) {
    const int idx = tid + 256 * i;
    const int kk = idx / C4, c4 = idx % C4;
    if constexpr (!GUARD) {
      v[i] = *reinterpret_cast<const vf4*>(src + (size_t)(k0 + kk) * ld + c0 + c4 * 4);
    } else {   // branch-free: clamp to a valid element, then select
      const bool ok = (k0 + kk < kmax) && (c0 + c4 * 4 < cmax);
      const int kr = k0 + kk < kmax ? k0 + kk : kmax - 1;
      const int cc = c0 + c4 * 4 < cmax ? c0 + c4 * 4 : cmax - 4;
      const vf4 t = *reinterpret_cast<const vf4*>(src + (size_t)kr * ld + cc);
      v[i] = make_vf4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
    }
  }
}
template <int COLS>
__device__ inline void store_kmajor(float* __restrict__ T, int tid, const vf4 (&v)[COLS / 32]) {
  constexpr int C4 = COLS / 4;
#pragma unroll
  for (int i = 0; i < COLS / 32; ++i) {
    const int idx = tid + 256 * i;
    const int kk = idx / C4, c4 = idx % C4;
    *reinterpret_cast<vf4*>(T + kk * COLS + c4 * 4) = v[i];
  }
}

// ---- LDS -> MFMA fragments ---------------------------------------------------------------------
// PITCH: LDK for k-contiguous tiles, the tile width for k-major tiles.
template <bool KMAJOR, int PITCH>
__device__ inline void frag4(const float* __restrict__ T, int idx, int q, int h, float (&o)[4]) {
  if constexpr (!KMAJOR) {
    const vf4 t = *reinterpret_cast<const vf4*>(T + idx * PITCH + q * 8 + h * 4);
    o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = T[(q * 8 + h * 4 + c) * PITCH + idx];
  }
}

// One K-step (32 k) of a wave's 64 x (32*TN) sub-tile.  `mask` bit j = column tile j is inside N
// (wave-uniform, only consulted when GUARD).
template <bool A_KMAJOR, int A_PITCH, bool B_KMAJOR, int B_PITCH, int TN, bool GUARD>
__device__ inline void mma_step(const float* __restrict__ As, const float* __restrict__ Bs, int a_base, int b_base,
                                int lane, unsigned mask, v16f (&acc)[2][TN]) {
  const int i = lane & 31, h = lane >> 5;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float a[2][4], b[TN][4];
    frag4<A_KMAJOR, A_PITCH>(As, a_base + i, q, h, a[0]);
    frag4<A_KMAJOR, A_PITCH>(As, a_base + 32 + i, q, h, a[1]);
#pragma unroll
    for (int tj = 0; tj < TN; ++tj) frag4<B_KMAJOR, B_PITCH>(Bs, b_base + 32 * tj + i, q, h, b[tj]);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int tj = 0; tj < TN; ++tj) {
        if (!GUARD || ((mask >> tj) & 1u)) {
          acc[0][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][c], b[tj][c], acc[0][tj], 0, 0, 0);
          acc[1][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1][c], b[tj][c], acc[1][tj], 0, 0, 0);
        }
      }
    }
    // keep the fragment loads of the next k-group behind this group's MFMAs: hoisting all four groups'
    // ds_reads to the top costs 4x the fragment registers and spills the staging registers
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Accumulator element (tile ti ; register r) of lane `lane` -> row inside the wave's 64-row band.
__device__ inline int acc_row(int ti, int r, int lane) { return ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// Epilogue through LDS: the MFMA accumulator layout (one column per lane, rows spread over registers)
// would make every global access a 4-byte-per-lane access of two 128-byte row segments.  Each wave
// instead transposes its sub-tile through a private LDS strip, 16 rows at a time, and then walks it
// row-major with 16 bytes per lane: the epilogue functor sees 4 consecutive columns of one row
// (`epi.apply4(row, col, v)`, col % 4 == 0) and issues dwordx4 loads/stores (512 contiguous bytes per
// half-wave).  `strip` = this wave's [16][32*TN + 4] floats of the (now idle) staging LDS.
template <int TN, class Epi, int TI = 2>
__device__ inline void run_epilogue(const v16f (&acc)[TI][TN], float* __restrict__ strip, int row0, int col0,
                                    int lane, unsigned mask, const Epi& epi) {
  constexpr int W = 32 * TN;   // columns of the wave's sub-tile
  constexpr int P = W + 4;     // strip pitch (floats)
  constexpr int C4 = W / 4;    // 16-byte groups per row
  const int h = lane >> 5, cl = lane & 31;
#pragma unroll
  for (int ti = 0; ti < TI; ++ti) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      // registers 8*half .. 8*half+7 of every column tile hold rows 16*half .. 16*half+15
#pragma unroll
      for (int tj = 0; tj < TN; ++tj) {
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
          const int rl = (rr & 3) + 8 * (rr >> 2) + 4 * h;
          strip[rl * P + tj * 32 + cl] = acc[ti][tj][half * 8 + rr];
        }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < W / 16; ++it) {
        const int idx = it * 64 + lane;
        const int rl = idx / C4, c4 = idx % C4;
        const vf4 v = *reinterpret_cast<const vf4*>(strip + rl * P + c4 * 4);
        if ((mask >> (c4 >> 3)) & 1u) epi.apply4(row0 + ti * 32 + half * 16 + rl, col0 + c4 * 4, v);
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

template <int TN>
__device__ inline void zero_acc(v16f (&acc)[2][TN]) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
}

// ---- C[M x N] = A[M x K] * op(B) ------------------------------------------------------------------
//   B_KMAJOR == false ("NT"): B given as W[N][K] (k contiguous): C = A W^T      (forward-shaped layers)
//   B_KMAJOR == true  ("NN"): B given as W[K][N] (n contiguous): C = A W        (reverse-shaped layers)
// M is a multiple of 128 (padded buffers), N and K multiples of 32.  grid = (M/128, ceil(N/BN)).
template <bool B_KMAJOR, int BN, bool GUARD>
__device__ inline void rows_main_loop(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
                                      int N, int K, int m_blk, int n_blk, int wm, int wn, unsigned mask,
                                      float* __restrict__ As, float* __restrict__ Bs, v16f (&acc)[2][BN / 64]) {
  constexpr int TN = BN / 64;
  constexpr int B_PITCH = B_KMAJOR ? BN : LDK;
  const int tid = threadIdx.x, lane = tid & 63;
  vf4 ra[BM / 32], rb[BN / 32];
  const int nk = K / BK;
  if (nk > 0) {
    load_rows<BM, false>(A, lda, m_blk, 0, 0, tid, ra);
    if constexpr (!B_KMAJOR) load_rows<BN, GUARD>(W, ldw, n_blk, 0, N, tid, rb);
    else load_kmajor<BN, GUARD>(W, ldw, 0, n_blk, K, N, tid, rb);
  }
  for (int kt = 0; kt < nk; ++kt) {
    store_rows<BM>(As, tid, ra);
    if constexpr (!B_KMAJOR) store_rows<BN>(Bs, tid, rb);
    else store_kmajor<BN>(Bs, tid, rb);
    lds_barrier();
    if (kt + 1 < nk) {
      const int k0 = (kt + 1) * BK;
      load_rows<BM, false>(A, lda, m_blk, k0, 0, tid, ra);
      if constexpr (!B_KMAJOR) load_rows<BN, GUARD>(W, ldw, n_blk, k0, N, tid, rb);
      else load_kmajor<BN, GUARD>(W, ldw, k0, n_blk, K, N, tid, rb);
    }
    mma_step<false, LDK, B_KMAJOR, B_PITCH, TN, GUARD>(As, Bs, wm * 64, wn * (BN / 2), lane, mask, acc);
    lds_barrier();
  }
}

// GUARD (chosen by the host: N % BN != 0) selects the variant whose B staging is bounds-checked and whose
// MFMAs are skipped for column tiles outside N.  One variant per kernel: two inlined copies of the main
// loop in one kernel exceed the backend's alloca-promotion budget and push the staging registers to scratch.
template <bool B_KMAJOR, int BN, bool GUARD, class Epi>
__global__ __launch_bounds__(256, BN == 256 ? 2 : 3) void gemm_rows_kernel(const float* __restrict__ A, int lda,
                                                                           const float* __restrict__ W, int ldw,
                                                                           int N, int K, Epi epi) {
  constexpr int TN = BN / 64;
  constexpr int B_FLOATS = B_KMAJOR ? BK * BN : BN * LDK;
  __shared__ __attribute__((aligned(16))) float smem[BM * LDK + B_FLOATS];
  float* As = smem;
  float* Bs = smem + BM * LDK;
  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int wm = wave >> 1, wn = wave & 1;
  const int m_blk = blockIdx.x * BM, n_blk = blockIdx.y * BN;
  // column tiles of this wave that lie inside N (wave-uniform)
  unsigned mask = 0;
#pragma unroll
  for (int tj = 0; tj < TN; ++tj)
    if (n_blk + wn * (BN / 2) + tj * 32 < N) mask |= 1u << tj;

  v16f acc[2][TN];
  zero_acc<TN>(acc);
  rows_main_loop<B_KMAJOR, BN, GUARD>(A, lda, W, ldw, N, K, m_blk, n_blk, wm, wn, mask, As, Bs, acc);
  // the main loop ends with a barrier: the staging tiles are idle and become the transpose strips
  static_assert(4 * 16 * (32 * TN + 4) <= BM * LDK + B_FLOATS, "epilogue strips must fit in the staging LDS");
  run_epilogue<TN, Epi>(acc, smem + wave * 16 * (32 * TN + 4), m_blk + wm * 64, n_blk + wn * (BN / 2), lane, mask,
                        epi);
}

// ---- dW[N x K] += X1^T Y1 (+ X2^T Y2), reduction over the M points, split-K over workgroups ---------
//   X* [M x N] (ldx), Y* [M x K] (ldy); rows >= M are masked.  1-D grid over (job, tile, split).
//   Partial tiles are accumulated into dW with float atomics (dW zero-initialised by the caller);
//   colsum(X) of pair `bias_pair` over the same rows is added to db when db != nullptr (by the
//   tile_k == 0 blocks).
struct DwPair {
  const float* X;
  int ldx;
  const float* Y;
  int ldy;
  // x2h form of the 256-row kernel (gemm_dw_x3_kernel<., 2>): which operand is the loss adjoint (0: X, 1: Y; the other one
  // is saved forward state of known range) and where its producer left max |.| over the real rows (float bits)
  int adj = 0;
  const unsigned* amax = nullptr;
  // ... and where the forward left max |.| of the OTHER operand (saved state: PointBufs::smax), or nullptr: the fixed 2^6
  const unsigned* smax = nullptr;
};

template <bool GUARD, int KT>
__device__ inline void dw_main_loop(const DwPair& p, int m_begin, int m_end, int N, int K, int n_blk, int k_blk,
                                    int wm, int wn, unsigned mask, bool do_bias, double& bsum,
                                    float* __restrict__ Xs, float* __restrict__ Ys, v16f (&acc)[2][KT / 64]) {
  const int tid = threadIdx.x, lane = tid & 63;
  vf4 rx[4], ry[KT / 32];
  load_kmajor<128, GUARD>(p.X, p.ldx, m_begin, n_blk, m_end, N, tid, rx);
  load_kmajor<KT, GUARD>(p.Y, p.ldy, m_begin, k_blk, m_end, K, tid, ry);
  for (int m0 = m_begin; m0 < m_end; m0 += BK) {
    store_kmajor<128>(Xs, tid, rx);
    store_kmajor<KT>(Ys, tid, ry);
    lds_barrier();
    if (m0 + BK < m_end) {
      load_kmajor<128, GUARD>(p.X, p.ldx, m0 + BK, n_blk, m_end, N, tid, rx);
      load_kmajor<KT, GUARD>(p.Y, p.ldy, m0 + BK, k_blk, m_end, K, tid, ry);
    }
    if (do_bias) {
#pragma unroll 8
      for (int kk = 0; kk < BK; ++kk) bsum += (double)Xs[kk * 128 + tid];
    }
    mma_step<true, 128, true, KT, KT / 64, GUARD>(Xs, Ys, wm * 64, wn * (KT / 2), lane, mask, acc);
    lds_barrier();
  }
}

// ---- fp32 products as bf16 MFMA terms ("x3"): primitives (the scheme is described in fused_common.hip.h) ----------
typedef unsigned short x3raw;
typedef unsigned vu4x __attribute__((ext_vector_type(4)));
typedef __bf16 x3bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 x3bf2 __attribute__((ext_vector_type(2)));

__device__ inline unsigned x3_pack2(float a, float b) {
  x3bf2 p = {(__bf16)a, (__bf16)b};   // v_cvt_pk_bf16_f32, round to nearest even
  return __builtin_bit_cast(unsigned, p);
}
// 8 consecutive k of one row -> the row's hi / mid / lo operand registers: 3 conversions, 4 unpacks and 4 subtractions
// per pair.  The subtractions (exact) must stay scalar: beside MFMAs a v_pk_add_f32 costs ~13 cycles more than the two
// v_sub_f32 it replaces (MI355X_MICROARCH.md, packed f32 VALU), and hipcc's SLP pass would pack two adjacent
// subtractions — so one of each pair is written as an fma, which it cannot merge with the other.
__device__ inline void x3_split8(const vf4& x0, const vf4& x1, vu4x& hi, vu4x& mid, vu4x& lo) {
  const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float a = x[2 * p], b = x[2 * p + 1];
    unsigned uh = x3_pack2(a, b);
    asm("" : "+v"(uh));   // one conversion per pair (otherwise the low half is converted a second time for `uh << 16`)
    const float ra = a - __builtin_bit_cast(float, uh << 16);
    const float rb = __builtin_fmaf(__builtin_bit_cast(float, uh & 0xffff0000u), -1.f, b);
    unsigned um = x3_pack2(ra, rb);
    asm("" : "+v"(um));
    const float sa = ra - __builtin_bit_cast(float, um << 16);
    const float sb = __builtin_fmaf(__builtin_bit_cast(float, um & 0xffff0000u), -1.f, rb);
    hi[p] = uh;
    mid[p] = um;
    lo[p] = x3_pack2(sa, sb);
  }
}
// ---- maxima of adjoint tensors (scales of the x2h weight-gradient jobs) --------------------------------------------
// A producer of an adjoint tensor leaves max |.| over the REAL rows (padding rows hold workspace garbage) in a slot of
// PointBufs::amax as float bits (non-negative floats order like unsigned integers): one atomic per wave.
__device__ inline void amax_commit(unsigned* slot, float m, int lane) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  // fire and forget (the result is unused: the instruction does not return, the wave does not wait).  Callers keep the number
  // of atomics per slot and launch in the thousands: every wave of a 65,536-row GEMM sending one at the same moment (round 4's
  // first version) serialised in the L2 for 35 us.
  if (lane == 0 && slot != nullptr)
    (void)__hip_atomic_fetch_max(slot, __builtin_bit_cast(unsigned, m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// max |acc| over the accumulator rows below `rows_ok` (relative to the wave's first row) of a (32 TI) x (32 TJ) block
template <int TI, int TJ>
__device__ inline float acc_absmax(const v16f (&acc)[TI][TJ], int lane, int rows_ok) {
  float m0 = 0.f, m1 = 0.f;
  if (rows_ok >= 32 * TI) {
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          m0 = fmaxf(m0, fabsf(acc[ti][tj][r]));
          m1 = fmaxf(m1, fabsf(acc[ti][tj][r + 1]));
        }
  } else {
    const int h = lane >> 5;
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          m0 = fmaxf(m0, row < rows_ok ? fabsf(acc[ti][tj][r]) : 0.f);
        }
  }
  return fmaxf(m0, m1);
}

// ---- "x2h": the same idea on the fp16 matrix pipe with TWO planes and THREE terms ----------------------------------
// xs = x * S (S a power of two: exact), hi = fp16(xs), lo = fp16(xs - hi): two 11-bit roundings, |xs - hi - lo| <= 2^-22 |xs|
// (rms 2^-23.6: about four times the rounding noise of storing x in fp32 at all) as long as lo is a normal fp16 number
// (|xs| >= 2^-3), and <= 2^-25 absolutely below that; hi overflows at |xs| >= 65520.  a b is taken as
// a_hi b_hi + a_hi b_lo + a_lo b_hi (the dropped a_lo b_lo is below 2^-22 |a b|), at HALF the matrix time of the six bf16
// terms.  These per-product errors are independent and far below what the fp32 accumulation of a K = 256 dot product
// commits (2.7 ulp rms, profiles/r04_sdf_bias.txt): measured on the full network, the SDF comes out closer to fp64 than
// with the six bf16 terms (rms 8.3e-8 against 1.07e-7, profiles/r04_x2h_check.txt).  What bf16 gave for free — range — is
// bought with the scales, and every scale is taken from the data it scales (round 5; fused_common.hip.h, "per-tile scale"):
// per matrix for the weights (2^8 while max |w| < 64), per 64-point tile and layer for activations, network inputs and the
// reverse sweep's Jacobian rows (2^6 while the tile stays below 256), from recorded maxima for loss adjoints and for the saved
// state the weight-gradient kernel reads.  No operand has a range to respect.  (The RA sweep alone stays on the bf16 scheme.)
typedef _Float16 x2h8 __attribute__((ext_vector_type(8)));
typedef _Float16 x2h2 __attribute__((ext_vector_type(2)));
constexpr float kH2ActScale = 64.f, kH2WScale = 256.f;
__device__ inline unsigned x2h_pack2(float a, float b) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f2{a, b}, x2h2));   // v_cvt_pk_f16_f32, round to nearest even
}
// x - (float)(low / high half of h) in one instruction (v_fma_mix_f32 reads the fp16 half directly)
__device__ inline float x2h_resid_lo(float x, unsigned h) {
  float r;
  asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(h));
  return r;
}
__device__ inline float x2h_resid_hi(float x, unsigned h) {
  float r;
  asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(h));
  return r;
}
// the power-of-two scale of the adjoint operand of an x2h job: its maximum (float bits `mbits`) goes to [2^13, 2^14)
__device__ inline void x2h_dyn_scale(unsigned mbits, float& s, float& inv_s) {
  int ef = (int)(mbits >> 23);                        // biased exponent of the maximum (0: all zero)
  ef = ef < 24 ? 24 : (ef > 250 ? 250 : ef);          // both factors stay normal numbers
  s = __builtin_bit_cast(float, (unsigned)(267 - ef) << 23);        // 2^(13 - e)
  inv_s = __builtin_bit_cast(float, (unsigned)(ef - 13) << 23);     // 2^(e - 13)
}
// 8 consecutive k of one row (already scaled) -> hi / lo operand registers: 4 instructions per pair
__device__ inline void x2h_split8(const vf4& x0, const vf4& x1, vu4x& hi, vu4x& lo) {
  const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const unsigned uh = x2h_pack2(x[2 * p], x[2 * p + 1]);
    hi[p] = uh;
    lo[p] = x2h_pack2(x2h_resid_lo(x[2 * p], uh), x2h_resid_hi(x[2 * p + 1], uh));
  }
}
// planes -> operand registers of one row, either scheme
template <int NP>
__device__ inline void xn_split8(const vf4& x0, const vf4& x1, vu4x (&p)[NP]) {
  if constexpr (NP == 3) x3_split8(x0, x1, p[0], p[1], p[2]);
  else x2h_split8(x0, x1, p[0], p[1]);
}
// the six (three) terms of one 16-k step, small ones first, the (ti, tj) accumulators in rotation
template <int TI, int TJ, bool FIRST, int NP = 3>
__device__ inline void x3_mfma(const vu4x (&a)[TI][NP], const vu4x (&b)[TJ][NP], v16f (&acc)[TI][TJ]) {
  constexpr int NT = NP == 3 ? 6 : 3;
  // NP == 3: a_lo b_hi, a_hi b_lo, a_mid b_mid, a_mid b_hi, a_hi b_mid, a_hi b_hi;  NP == 2: a_lo b_hi, a_hi b_lo, a_hi b_hi
  constexpr int PA[6] = {NP == 3 ? 2 : 1, 0, NP == 3 ? 1 : 0, 1, 0, 0};
  constexpr int PB[6] = {0, NP == 3 ? 2 : 1, NP == 3 ? 1 : 0, 0, 1, 0};
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
      for (int ti = 0; ti < TI; ++ti) {
        const v16f zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const v16f c = (FIRST && t == 0) ? zero : acc[ti][tj];
        if constexpr (NP == 3)
          acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(x3bf8, a[ti][PA[t]]),
                                                                __builtin_bit_cast(x3bf8, b[tj][PB[t]]), c, 0, 0, 0);
        else
          acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(x2h8, a[ti][PA[t]]),
                                                               __builtin_bit_cast(x2h8, b[tj][PB[t]]), c, 0, 0, 0);
      }
}

// One dW job = one weight matrix; a launch carries a group of jobs so that the atomic tail of one matrix
// overlaps the main loop of the next (a single 256x256 job is ONE resident wave of workgroups: all of them
// would reach their atomics together).  block_end = exclusive prefix sum of the jobs' block counts, each a
// multiple of 8 when splits is (keeps the XCD decode below valid inside the group).
struct DwJob {
  DwPair p1, p2;
  float* dW;
  float* db;
  // RNB_VARIANT_DETERMINISTIC: per-split partial tiles [splits][N][lddw] / column sums [splits][N] written with plain
  // stores (zero-initialised by the caller) and summed in split order by dw_reduce_kernel; nullptr: fp32 atomics
  float* part;
  float* partb;
  int npairs, N, K, lddw, bias_pair, splits, rows_per_split, block_end;
};
constexpr int kMaxDwJobs = 12;
constexpr int kMaxDwExtra = 2;   // reduce-only jobs (slabs written by other kernels) that ride in the reduction launch
struct DwGroup {
  DwJob job[kMaxDwJobs + kMaxDwExtra];
  int njobs, M;
};

// GUARD (host: N % 128 || K % KT || M % 32) as for gemm_rows_kernel, and the width KT (128 or 64) of the
// output tile along K: all jobs of a group share both.  KT = 64 keeps all four waves busy on narrow or ragged
// K (the PE-input layer has K = 64, the albedo net's first layer K = 320).
template <bool GUARD, int KT>
__global__ __launch_bounds__(256, 3) void gemm_dw_kernel(const DwGroup g) {
  constexpr int TN = KT / 64;
  __shared__ __attribute__((aligned(16))) float smem[BK * 128 + BK * KT];
  float* Xs = smem;
  float* Ys = smem + BK * 128;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
  const int wm = wave >> 1, wn = wave & 1;
  int ji = 0, begin = 0;
  for (int i = 0; i + 1 < g.njobs; ++i)
    if ((int)blockIdx.x >= g.job[i].block_end) { ji = i + 1; begin = g.job[i].block_end; }
  const DwJob& J = g.job[ji];
  const int blk = (int)blockIdx.x - begin;
  const int M = g.M, N = J.N, K = J.K, splits = J.splits;
  float* __restrict__ dW = J.dW;
  float* __restrict__ db = J.db;
  // Blocks of a job: tiles_n * tiles_k * splits.  Workgroups are dealt round-robin over the 8 XCDs
  // (blocks b and b+8 share an L2), so the tiles of one point-split are placed on ONE XCD in consecutive
  // dispatch slots: the X / Y chunks they share are then served by that XCD's L2 instead of being fetched
  // once per tile.  Pure placement heuristic: any mapping is correct.
  const int tiles_n = (N + 127) / 128, tiles_k = (K + KT - 1) / KT;
  const int nt = tiles_n * tiles_k;
  if (blk >= nt * splits) return;   // padding blocks that align the next job to 8
  int tile, split;
  if (splits % 8 == 0) {
    const int xcd = blk & 7, j = blk >> 3;
    tile = j % nt;
    split = (j / nt) * 8 + xcd;
  } else {
    tile = blk % nt;
    split = blk / nt;
  }
  const int tile_n = tile % tiles_n, tile_k = tile / tiles_n;
  const int n_blk = tile_n * 128, k_blk = tile_k * KT;
  const int m_begin = split * J.rows_per_split;
  const int m_end = min(M, m_begin + J.rows_per_split);
  unsigned mask = 0;
#pragma unroll
  for (int tj = 0; tj < TN; ++tj)
    if (k_blk + wn * (KT / 2) + tj * 32 < K) mask |= 1u << tj;

  v16f acc[2][TN];
  zero_acc<TN>(acc);
  double bsum = 0.0;   // bias gradients are long signed sums: keep the per-block partial in fp64
  const bool bias_blk = (db != nullptr) && tile_k == 0 && tid < 128 && (n_blk + tid < N);

  if (m_begin < m_end) {
    for (int pi = 0; pi < J.npairs; ++pi) {
      const DwPair p = pi == 0 ? J.p1 : J.p2;
      const bool do_bias = bias_blk && pi == J.bias_pair;
      dw_main_loop<GUARD, KT>(p, m_begin, m_end, N, K, n_blk, k_blk, wm, wn, mask, do_bias, bsum, Xs, Ys, acc);
    }
  }
  // atomics: each register of a 32x32 accumulator is two 128-byte row segments per wave instruction
  const int lddw = J.lddw;
  float* __restrict__ pdst = J.part ? J.part + (size_t)split * N * lddw : nullptr;
#pragma unroll
  for (int tj = 0; tj < TN; ++tj) {
    if (!((mask >> tj) & 1u)) continue;
    const int col = k_blk + wn * (KT / 2) + tj * 32 + (lane & 31);
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = n_blk + wm * 64 + acc_row(ti, r, lane);
        if (row < N) {
          if (pdst) pdst[(size_t)row * lddw + col] = acc[ti][tj][r];
          else atomicAdd(dW + (size_t)row * lddw + col, acc[ti][tj][r]);
        }
      }
    }
  }
  if (bias_blk) {
    if (J.partb) J.partb[(size_t)split * N + n_blk + tid] = (float)bsum;
    else atomicAdd(db + n_blk + tid, (float)bsum);
  }
}

// ---- dW without LDS ----------------------------------------------------------------------------------
// Both operands of dW = X^T Y are point-major in memory, and v_mfma_f32_32x32x2_f32 wants exactly that: lane
// (i, h) supplies A[row i][k = h] and B[k = h][col i], i.e. for a pair of consecutive points the two lane
// halves read the two rows X[m + h][...] — a coalesced global load IS the fragment.  No staging, no
// barriers: the four waves of a workgroup (2 x 2 sub-tiles of 64 x KT/2) run independently and only share
// L1/L2 lines.  X comes in as 8-byte loads (lane i holds columns 2i, 2i+1 -> the wave's two row tiles are
// the even and the odd rows of its 64-row band), Y as 4-byte loads (columns i and 32 + i), so a pair of
// points costs 1 + TN loads for 2 * TN MFMAs.  Loads run two 16-point chunks ahead in a 3-slot register
// ring (<= 63 in flight per wave).  Exact shapes only: N % 128 == 0, K % KT == 0, point ranges % 16 == 0.
template <int KT, int OCC>
__global__ __launch_bounds__(256, OCC) void gemm_dw_direct_kernel(const DwGroup g) {
  constexpr int TN = KT / 64;
  constexpr int CH = 8;   // point pairs per chunk
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
  const int wm = wave >> 1, wn = wave & 1;
  int ji = 0, begin = 0;
  for (int q = 0; q + 1 < g.njobs; ++q)
    if ((int)blockIdx.x >= g.job[q].block_end) { ji = q + 1; begin = g.job[q].block_end; }
  const DwJob& J = g.job[ji];
  const int blk = (int)blockIdx.x - begin;
  const int M = g.M, N = J.N, K = J.K, splits = J.splits;
  const int tiles_n = N / 128, tiles_k = K / KT;
  const int nt = tiles_n * tiles_k;
  if (blk >= nt * splits) return;   // padding blocks that align the next job to 8
  int tile, split;
  if (splits % 8 == 0) {            // XCD-aware placement, see gemm_dw_kernel
    const int xcd = blk & 7, j = blk >> 3;
    tile = j % nt;
    split = (j / nt) * 8 + xcd;
  } else {
    tile = blk % nt;
    split = blk / nt;
  }
  const int tile_n = tile % tiles_n, tile_k = tile / tiles_n;
  const int m_begin = split * J.rows_per_split;
  const int m_end = min(M, m_begin + J.rows_per_split);
  if (m_begin >= m_end) return;
  const int i = lane & 31, h = lane >> 5;
  const int n_w = tile_n * 128 + wm * 64;          // first row (of dW) of this wave
  const int k_w = tile_k * KT + wn * (KT / 2);     // first column
  const int nch = (m_end - m_begin) / (2 * CH);
  const bool bias_wave = J.db != nullptr && tile_k == 0 && wn == 0;

  v16f acc[2][TN];
  zero_acc<TN>(acc);
  double bs0 = 0.0, bs1 = 0.0;   // column sums of X (bias gradient), fp64 partials

  for (int pi = 0; pi < J.npairs; ++pi) {
    const DwPair p = pi == 0 ? J.p1 : J.p2;
    const bool do_bias = bias_wave && pi == J.bias_pair;
    const unsigned xrow = (unsigned)p.ldx * 4u, yrow = (unsigned)p.ldy * 4u;   // row pitch in bytes
    const BufRsrc rx = tile_rsrc(p.X + (size_t)m_begin * p.ldx, (unsigned)(m_end - m_begin) * xrow);
    const BufRsrc ry = tile_rsrc(p.Y + (size_t)m_begin * p.ldy, (unsigned)(m_end - m_begin) * yrow);
    const unsigned vx = (unsigned)h * xrow + (unsigned)(n_w + 2 * i) * 4u;
    const unsigned vy = (unsigned)h * yrow + (unsigned)(k_w + i) * 4u;
    vf2 a[3][CH];
    float b[3][CH][TN];
#define RNB_DW_LOAD(slot, chunk)                                                     \
    {                                                                                  \
      const int c_ = min((chunk), nch - 1);                                            \
      const unsigned sx = (unsigned)c_ * (2 * CH) * xrow, sy = (unsigned)c_ * (2 * CH) * yrow; \
      _Pragma("unroll") for (int q = 0; q < CH; ++q) {                                 \
        a[slot][q] = bload2(rx, vx, sx + (unsigned)(2 * q) * xrow);                    \
        _Pragma("unroll") for (int tj = 0; tj < TN; ++tj)                              \
          b[slot][q][tj] = bload(ry, vy + (unsigned)tj * 128u, sy + (unsigned)(2 * q) * yrow); \
      }                                                                                \
    }
#define RNB_DW_MMA(slot)                                                               \
    {                                                                                  \
      _Pragma("unroll") for (int q = 0; q < CH; ++q) {                                 \
        _Pragma("unroll") for (int tj = 0; tj < TN; ++tj) {                            \
          acc[0][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[slot][q].x, b[slot][q][tj], acc[0][tj], 0, 0, 0); \
          acc[1][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[slot][q].y, b[slot][q][tj], acc[1][tj], 0, 0, 0); \
        }                                                                              \
      }                                                                                \
      if (do_bias) {   /* 16 points in fp32, then into the fp64 partial */                 \
        float t0 = 0.f, t1 = 0.f;                                                      \
        _Pragma("unroll") for (int q = 0; q < CH; ++q) { t0 += a[slot][q].x; t1 += a[slot][q].y; } \
        bs0 += (double)t0;                                                             \
        bs1 += (double)t1;                                                             \
      }                                                                                \
    }
    RNB_DW_LOAD(0, 0)
    RNB_DW_LOAD(1, 1)
    for (int c = 0; c < nch; c += 3) {
      RNB_DW_LOAD(2, c + 2)
      __builtin_amdgcn_sched_barrier(0);
      RNB_DW_MMA(0)
      __builtin_amdgcn_sched_barrier(0);
      RNB_DW_LOAD(0, c + 3)
      __builtin_amdgcn_sched_barrier(0);
      if (c + 1 < nch) RNB_DW_MMA(1)
      __builtin_amdgcn_sched_barrier(0);
      RNB_DW_LOAD(1, c + 4)
      __builtin_amdgcn_sched_barrier(0);
      if (c + 2 < nch) RNB_DW_MMA(2)
      __builtin_amdgcn_sched_barrier(0);
    }
#undef RNB_DW_LOAD
#undef RNB_DW_MMA
  }
  // atomics: accumulator (ti, tj, r) of lane (i, h) is dW[n_w + 2 * rho + ti][k_w + 32 * tj + i],
  // rho = (r & 3) + 8 * (r >> 2) + 4 * h
  float* __restrict__ dW = J.dW;
  const int lddw = J.lddw;
  float* __restrict__ pdst = J.part ? J.part + (size_t)split * N * lddw : nullptr;
#pragma unroll
  for (int tj = 0; tj < TN; ++tj) {
    const int col = k_w + tj * 32 + i;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = n_w + 2 * ((r & 3) + 8 * (r >> 2) + 4 * h) + ti;
        if (pdst) pdst[(size_t)row * lddw + col] = acc[ti][tj][r];
        else atomicAdd(dW + (size_t)row * lddw + col, acc[ti][tj][r]);
      }
    }
  }
  if (bias_wave) {
    bs0 += __shfl_xor(bs0, 32, 64);
    bs1 += __shfl_xor(bs1, 32, 64);
    if (h == 0) {
      if (J.partb) {
        J.partb[(size_t)split * N + n_w + 2 * i] = (float)bs0;
        J.partb[(size_t)split * N + n_w + 2 * i + 1] = (float)bs1;
      } else {
        atomicAdd(J.db + n_w + 2 * i, (float)bs0);
        atomicAdd(J.db + n_w + 2 * i + 1, (float)bs1);
      }
    }
  }
}

// RNB_VARIANT_DETERMINISTIC: dW = sum over splits (in split order, fp64 running sum) of the partial tiles; grid =
// (blocks over N * lddw elements, job).
template <int DUMMY>
__global__ __launch_bounds__(256) void dw_reduce_kernel(const DwGroup g) {
  const DwJob& J = g.job[blockIdx.y];
  // slabs [split][N][K] (K == lddw for whole-matrix jobs; a column range of a wider matrix has K < lddw)
  const size_t n = (size_t)J.N * J.K;
  const bool dense = J.K == J.lddw;
  if (J.splits > 64) {
    // many small slabs (the per-tile column sums of the fused albedo backward: a thousand slabs of a few hundred elements):
    // 16 elements x 16 split phases per workgroup pass, each thread a fixed subsequence of the slabs, the 16 phases summed in
    // phase order — as reproducible as one chain, 256 loads in flight per workgroup instead of one per element.  The bias
    // slabs ride as elements n .. n + N.
    __shared__ double red[16][17];
    const size_t ntot = n + ((J.db != nullptr && J.partb != nullptr) ? (size_t)J.N : 0);
    const int e = threadIdx.x & 15, ph = threadIdx.x >> 4;
    for (size_t base = (size_t)blockIdx.x * 16; base < ntot; base += (size_t)gridDim.x * 16) {   // (uniform per workgroup)
      const size_t idx = base + e;
      double s[4] = {0.0, 0.0, 0.0, 0.0};
      if (idx < ntot) {
        const float* src = idx < n ? J.part + idx : J.partb + (idx - n);
        const size_t stride = idx < n ? n : (size_t)J.N;
        int sp = ph;
        for (; sp + 48 < J.splits; sp += 64) {
#pragma unroll
          for (int u = 0; u < 4; ++u) s[u] += (double)src[(size_t)(sp + 16 * u) * stride];
        }
        for (int u = 0; sp < J.splits; sp += 16, ++u) s[u] += (double)src[(size_t)sp * stride];
      }
      red[ph][e] = (s[0] + s[1]) + (s[2] + s[3]);
      __syncthreads();
      if (threadIdx.x < 16 && base + threadIdx.x < ntot) {
        const size_t i2 = base + threadIdx.x;
        double t = 0.0;
        for (int q = 0; q < 16; ++q) t += red[q][threadIdx.x];
        if (i2 < n) J.dW[dense ? i2 : (i2 / (size_t)J.K) * (size_t)J.lddw + i2 % (size_t)J.K] = (float)t;
        else J.db[i2 - n] = (float)t;
      }
      __syncthreads();
    }
    return;
  }
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (size_t)gridDim.x * 256) {
    // four interleaved running sums (splits 0, 4, 8 .. / 1, 5, .. / ..) combined in a fixed order: as reproducible as one
    // chain, but four loads in flight instead of one
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    int sp = 0;
    for (; sp + 4 <= J.splits; sp += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) s[u] += (double)J.part[(size_t)(sp + u) * n + idx];
    }
    for (int u = 0; sp < J.splits; ++sp, ++u) s[u] += (double)J.part[(size_t)sp * n + idx];
    const size_t dst = dense ? idx : (idx / (size_t)J.K) * (size_t)J.lddw + idx % (size_t)J.K;
    J.dW[dst] = (float)((s[0] + s[1]) + (s[2] + s[3]));
  }
  if (J.db != nullptr && J.partb != nullptr) {
    for (int r = blockIdx.x * 256 + threadIdx.x; r < J.N; r += gridDim.x * 256) {
      double s = 0.0;
      for (int sp = 0; sp < J.splits; ++sp) s += (double)J.partb[(size_t)sp * J.N + r];
      J.db[r] = (float)s;
    }
  }
}

// ---- dW for 256 x 256 weight matrices, operands staged in LDS by LDS-DMA ---------------------------------------------
// The register-direct kernel above gives every 128 x 128 tile its own workgroup, so each operand half is fetched by two
// workgroups that are not synchronised: 2.0x the unique bytes at the memory side (profiles/hbm_traffic.json, round 1).
// Here ONE workgroup of 16 waves owns the whole 256 x 256 gradient of a point range: per 32-point chunk the two operand
// slabs (32 x 256 fp32 = 32 KB each, rows of the point-major matrices as they lie in memory) are copied global -> LDS by
// global_load_lds_dwordx4 (no VGPR round trip) into two alternating buffers: one raw barrier per 32-point chunk, the
// next chunk's DMAs in flight while this one is multiplied.  The fragments are what the direct kernel loads from global: lane (i, h)
// reads X[m + h][2i, 2i + 1] (8 bytes: the wave's two row tiles are the even / odd rows of its 64-row band) and
// Y[m + h][i], Y[m + h][32 + i] — conflict-free ds_read_b64 / ds_read_b32.  Wave (wm, wn) = rows 64 wm.., columns 64 wn...
constexpr int kStChunk = 32;                    // points per chunk
constexpr int kStOpBytes = kStChunk * 256 * 4;  // one operand slab (32 KB)
constexpr int kStBufs = 2;                      // chunk c + 1 is copied while chunk c is multiplied

__device__ inline void dw_staged_issue(const DwPair& p, int m_begin, int nchunks, int chunk, char* buf, int wave, int lane) {
  const int c = chunk < nchunks ? chunk : nchunks - 1;   // past the end: harmless re-fetch, keeps vmcnt uniform
#pragma unroll
  for (int q = 0; q < kStChunk / 16; ++q) {
    const int u = (q * 16 + wave) * 64 + lane;           // 16-byte unit of the slab: row u / 64, columns 4 (u % 64)..
    const int row = m_begin + c * kStChunk + (u >> 6);
    const float* xs = p.X + (size_t)row * p.ldx + (u & 63) * 4;
    const float* ys = p.Y + (size_t)row * p.ldy + (u & 63) * 4;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)xs,
                                     (__attribute__((address_space(3))) void*)(buf + (q * 16 + wave) * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ys,
                                     (__attribute__((address_space(3))) void*)(buf + kStOpBytes + (q * 16 + wave) * 1024), 16, 0, 0);
  }
}

template <int DUMMY>
__global__ __launch_bounds__(1024, 1) void gemm_dw_staged_kernel(const DwGroup g) {
  __shared__ __attribute__((aligned(16))) char lds[kStBufs * 2 * kStOpBytes];   // 128 KB: the only shared object
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
  const int wm = wave >> 2, wn = wave & 3;
  int ji = 0, begin = 0;
  for (int q = 0; q + 1 < g.njobs; ++q)
    if ((int)blockIdx.x >= g.job[q].block_end) { ji = q + 1; begin = g.job[q].block_end; }
  const DwJob& J = g.job[ji];
  const int split = (int)blockIdx.x - begin;
  if (split >= J.splits) return;
  const int m_begin = split * J.rows_per_split;
  const int m_end = min(g.M, m_begin + J.rows_per_split);
  if (m_begin >= m_end) return;
  const int nchunks = (m_end - m_begin) / kStChunk;   // ranges are multiples of the chunk (host)
  const int i = lane & 31, h = lane >> 5;
  v16f acc[2][2];
  zero_acc<2>(acc);
  double bs0 = 0.0, bs1 = 0.0;
  const bool bias_wave = J.db != nullptr && wn == 0;
  for (int pi = 0; pi < J.npairs; ++pi) {
    const DwPair p = pi == 0 ? J.p1 : J.p2;
    const bool do_bias = bias_wave && pi == J.bias_pair;
    __builtin_amdgcn_s_barrier();   // every wave is done with the buffers of the previous pair
    dw_staged_issue(p, m_begin, nchunks, 0, lds, wave, lane);
    for (int c = 0; c < nchunks; ++c) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part of chunk c has landed
      __builtin_amdgcn_s_barrier();                      // ... every wave's; and chunk c - 1 has been read by all
      dw_staged_issue(p, m_begin, nchunks, c + 1, lds + ((c + 1) & 1) * 2 * kStOpBytes, wave, lane);
      const char* bx = lds + (c & 1) * 2 * kStOpBytes;
      const char* by = bx + kStOpBytes;
      float t0 = 0.f, t1 = 0.f;
      // fragments of point pair q + 1 are read while the four MFMAs of pair q run (explicit two-deep rotation: left to
      // itself the compiler waits for each pair's reads right in front of its MFMAs)
      const char* ax = bx + h * 1024 + (wm * 64 + 2 * i) * 4;
      const char* ay = by + h * 1024 + (wn * 64 + i) * 4;
      vf2 a0 = *reinterpret_cast<const vf2*>(ax);
      float b00 = *reinterpret_cast<const float*>(ay), b01 = *reinterpret_cast<const float*>(ay + 128);
#pragma unroll
      for (int q = 0; q < kStChunk / 2; q += 2) {
        const vf2 a1 = *reinterpret_cast<const vf2*>(ax + (q + 1) * 2048);
        const float b10 = *reinterpret_cast<const float*>(ay + (q + 1) * 2048);
        const float b11 = *reinterpret_cast<const float*>(ay + (q + 1) * 2048 + 128);
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b00, acc[0][0], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b00, acc[1][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b01, acc[0][1], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b01, acc[1][1], 0, 0, 0);
        if (do_bias) { t0 += a0.x; t1 += a0.y; }
        __builtin_amdgcn_sched_barrier(0);
        if (q + 2 < kStChunk / 2) {
          a0 = *reinterpret_cast<const vf2*>(ax + (q + 2) * 2048);
          b00 = *reinterpret_cast<const float*>(ay + (q + 2) * 2048);
          b01 = *reinterpret_cast<const float*>(ay + (q + 2) * 2048 + 128);
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b10, acc[0][0], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b10, acc[1][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b11, acc[0][1], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b11, acc[1][1], 0, 0, 0);
        if (do_bias) { t0 += a1.x; t1 += a1.y; }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (do_bias) { bs0 += (double)t0; bs1 += (double)t1; }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS reads of chunk c are complete
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the over-fetched chunk before the buffers are reused
  }
  // accumulator (ti, tj, r) of lane (i, h) is dW[64 wm + 2 rho + ti][64 wn + 32 tj + i], rho = (r & 3) + 8 (r >> 2) + 4 h
  const int lddw = J.lddw;
  float* __restrict__ pdst = J.part ? J.part + (size_t)split * J.N * lddw : nullptr;
#pragma unroll
  for (int tj = 0; tj < 2; ++tj) {
    const int col = wn * 64 + tj * 32 + i;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * h) + ti;
        if (pdst) pdst[(size_t)row * lddw + col] = acc[ti][tj][r];
        else atomicAdd(J.dW + (size_t)row * lddw + col, acc[ti][tj][r]);
      }
  }
  if (bias_wave) {
    bs0 += __shfl_xor(bs0, 32, 64);
    bs1 += __shfl_xor(bs1, 32, 64);
    if (h == 0) {
      if (J.partb) {
        J.partb[(size_t)split * J.N + wm * 64 + 2 * i] = (float)bs0;
        J.partb[(size_t)split * J.N + wm * 64 + 2 * i + 1] = (float)bs1;
      } else {
        atomicAdd(J.db + wm * 64 + 2 * i, (float)bs0);
        atomicAdd(J.db + wm * 64 + 2 * i + 1, (float)bs1);
      }
    }
  }
}

// ---- C[M x N] = A[M x K] W[N][K]^T as six bf16 MFMA terms (RNB_VARIANT_X3; the albedo network's GEMMs) -------------
// Same tiling and epilogues as gemm_rows_kernel (128 x BN output tile, 4 waves of 64 x BN/2), but the 16-k staging
// tiles are split into hi / mid / lo bf16 planes on their way into LDS (once per workgroup: every staged fp32 value
// is split by the thread that loaded it), and the waves multiply plane fragments: lane (row i, half h) reads 16 bytes
// = k 8h .. 8h + 7 of its row.  Plane tile: [rows][48 bytes] (16 bf16 + pad: conflict-free ds_read_b128).
constexpr int XK = 16;   // k per staging tile = one MFMA k-step
constexpr int XP = 48;   // bytes per row of a plane tile

typedef unsigned vu2x __attribute__((ext_vector_type(2)));
__device__ inline void x3_split4(const vf4& x, vu2x& hi, vu2x& mid, vu2x& lo) {
  const float v[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const float a = v[2 * p], b = v[2 * p + 1];
    unsigned uh = x3_pack2(a, b);
    asm("" : "+v"(uh));
    const float ra = a - __builtin_bit_cast(float, uh << 16);
    const float rb = __builtin_fmaf(__builtin_bit_cast(float, uh & 0xffff0000u), -1.f, b);
    unsigned um = x3_pack2(ra, rb);
    asm("" : "+v"(um));
    const float sa = ra - __builtin_bit_cast(float, um << 16);
    const float sb = __builtin_fmaf(__builtin_bit_cast(float, um & 0xffff0000u), -1.f, rb);
    hi[p] = uh;
    mid[p] = um;
    lo[p] = x3_pack2(sa, sb);
  }
}
__device__ inline void x2h_split4(const vf4& x, vu2x& hi, vu2x& lo) {   // x already scaled
  const float v[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const unsigned uh = x2h_pack2(v[2 * p], v[2 * p + 1]);
    hi[p] = uh;
    lo[p] = x2h_pack2(x2h_resid_lo(v[2 * p], uh), x2h_resid_hi(v[2 * p + 1], uh));
  }
}
// ROWS x 16 k of a k-contiguous source -> ROWS / 64 float4 per thread (row idx >> 2, k quad idx & 3).  Buffer loads:
// the thread's (row, k quad) offset is loop-invariant (one VGPR per float4), the k offset of the tile is scalar.
template <int ROWS, bool GUARD>
__device__ inline void x3_rows_offsets(int ld, int r0, int rmax, int tid, unsigned (&off)[ROWS / 64], bool (&ok)[ROWS / 64]) {
#pragma unroll
  for (int i = 0; i < ROWS / 64; ++i) {
    const int idx = tid + 256 * i;
    const int r = idx >> 2, c4 = idx & 3;
    ok[i] = !GUARD || r0 + r < rmax;
    const int rr = ok[i] ? r0 + r : rmax - 1;   // GUARD: a clamped (valid) row, selected away below
    off[i] = ((unsigned)rr * (unsigned)ld + (unsigned)c4 * 4u) * 4u;
  }
}
template <int ROWS, bool GUARD>
__device__ inline void x3_load_rows(BufRsrc rs, const unsigned (&off)[ROWS / 64], const bool (&ok)[ROWS / 64], int k0,
                                    vf4 (&v)[ROWS / 64]) {
#pragma unroll
  for (int i = 0; i < ROWS / 64; ++i) {
    const vf4 t = __builtin_bit_cast(vf4, __builtin_amdgcn_raw_buffer_load_b128(rs, off[i], (unsigned)k0 * 4u, 0));
    if constexpr (!GUARD) v[i] = t;
    else v[i] = make_vf4(ok[i] ? t.x : 0.f, ok[i] ? t.y : 0.f, ok[i] ? t.z : 0.f, ok[i] ? t.w : 0.f);
  }
}
template <int ROWS>
__device__ inline void x3_store_rows(char* __restrict__ P, int tid, const vf4 (&v)[ROWS / 64]) {
#pragma unroll
  for (int i = 0; i < ROWS / 64; ++i) {
    const int idx = tid + 256 * i;
    const int r = idx >> 2, c4 = idx & 3;
    vu2x hi, mid, lo;
    x3_split4(v[i], hi, mid, lo);
    char* w = P + r * XP + c4 * 8;
    *reinterpret_cast<vu2x*>(w) = hi;
    *reinterpret_cast<vu2x*>(w + ROWS * XP) = mid;
    *reinterpret_cast<vu2x*>(w + 2 * ROWS * XP) = lo;
  }
}

template <int BN, bool GUARD, class Epi>
__global__ __launch_bounds__(256, 2) void gemm_rows_x3_kernel(const float* __restrict__ A, int lda,
                                                              const float* __restrict__ W, int ldw, int N, int K, Epi epi) {
  constexpr int TN = BN / 64;
  __shared__ __attribute__((aligned(16))) char smem[3 * (BM + BN) * XP];
  char* Ap = smem;
  char* Bp = smem + 3 * BM * XP;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
  const int wm = wave >> 1, wn = wave & 1;
  const int m_blk = blockIdx.x * BM, n_blk = blockIdx.y * BN;
  unsigned mask = 0;
#pragma unroll
  for (int tj = 0; tj < TN; ++tj)
    if (n_blk + wn * (BN / 2) + tj * 32 < N) mask |= 1u << tj;
  const int i = lane & 31, h = lane >> 5;
  const char* fa = Ap + (wm * 64 + i) * XP + h * 16;
  const char* fb = Bp + (wn * (BN / 2) + i) * XP + h * 16;

  v16f acc[2][TN];
  zero_acc<TN>(acc);
  vf4 ra[BM / 64], rb[BN / 64];
  const int nk = K / XK;
  // resources based at the tile's first row (32-bit offsets stay inside the tile)
  const BufRsrc rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + (size_t)m_blk * lda), 0, 0xfffffffc, 0x00020000);
  const BufRsrc rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W + (size_t)n_blk * ldw), 0, 0xfffffffc, 0x00020000);
  unsigned oa[BM / 64], ob[BN / 64];
  bool ka[BM / 64], kb[BN / 64];
  x3_rows_offsets<BM, false>(lda, 0, 0, tid, oa, ka);
  x3_rows_offsets<BN, GUARD>(ldw, 0, N - n_blk, tid, ob, kb);
  x3_load_rows<BM, false>(rsA, oa, ka, 0, ra);
  x3_load_rows<BN, GUARD>(rsW, ob, kb, 0, rb);
  for (int kt = 0; kt < nk; ++kt) {
    x3_store_rows<BM>(Ap, tid, ra);
    x3_store_rows<BN>(Bp, tid, rb);
    lds_barrier();
    {
      const int k0 = min(kt + 1, nk - 1) * XK;   // past the end: a harmless re-load
      x3_load_rows<BM, false>(rsA, oa, ka, k0, ra);
      x3_load_rows<BN, GUARD>(rsW, ob, kb, k0, rb);
    }
    vu4x a[2][3];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) a[ti][pl] = *reinterpret_cast<const vu4x*>(fa + pl * BM * XP + ti * 32 * XP);
#pragma unroll
    for (int tj = 0; tj < TN; ++tj) {
      if (GUARD && !((mask >> tj) & 1u)) continue;
      vu4x b[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) b[pl] = *reinterpret_cast<const vu4x*>(fb + pl * BN * XP + tj * 32 * XP);
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0};   // small terms first (x3_mfma)
      constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
          acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(x3bf8, a[ti][PA[t]]),
                                                                __builtin_bit_cast(x3bf8, b[PB[t]]), acc[ti][tj], 0, 0, 0);
    }
    lds_barrier();
  }
  static_assert(4 * 16 * (32 * TN + 4) * 4 <= 3 * (BM + BN) * XP, "epilogue strips must fit in the staging LDS");
  run_epilogue<TN, Epi>(acc, reinterpret_cast<float*>(smem) + wave * 16 * (32 * TN + 4), m_blk + wm * 64,
                        n_blk + wn * (BN / 2), lane, mask, epi);
}

// ---- the same product with the WEIGHTS TAKEN FROM THE SPLIT MIRROR ("x3m"; the albedo network's GEMMs) -----------------
// gemm_rows_x3_kernel splits both operands on their way into LDS — the weight tile again in every one of the M / 128
// workgroups — behind two barriers per 16-k step.  Here only the activations are staged: one float4 per thread and step,
// split once, into one of two plane tiles [3][128][48 B] (one barrier per step; the split of step s + 1 rides beside the
// MFMAs of step s); the weight fragments come straight from the fragment-ordered mirror that rnb_weightnorm_fwd wrote
// (x3_pack_weights; one contiguous 1 KB per load, as in the fused sweeps), two steps ahead in registers.  One 8-wave
// workgroup per 128 rows; wave w owns ALL 128 rows x the column tiles w and (TJ == 2) w + 8 — every weight fragment is
// fetched once per workgroup, every activation split once.  N <= 32 * 8 * TJ, N % 32 == 0, K % 32 == 0.
// amax != nullptr: max |acc| over the rows below m_real is left there (an upper bound of the epilogue's masked outputs)
template <int TJ, class Epi>
__global__ __launch_bounds__(512, 1) void gemm_rows_x3m_kernel(const float* __restrict__ A, int lda,
                                                               const x3raw* __restrict__ W3, int N, int K, Epi epi,
                                                               unsigned* amax = nullptr, long long m_real = 0) {
  constexpr int NP = 3;   // (an fp16 two-plane form of this kernel existed in round 4; the fused albedo kernels replaced it)
  constexpr int ROWS = 128;
  constexpr int PLB = ROWS * XP;              // bytes of one plane
  constexpr int BUFB = NP * PLB;              // one staging buffer (three planes: 18,432 B)
  constexpr int STRIP = 16 * (32 + 4) * 4;    // epilogue strip of one wave (one 32-column tile at a time)
  __shared__ __attribute__((aligned(16))) char smem[2 * BUFB > 8 * STRIP ? 2 * BUFB : 8 * STRIP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
  const int m_blk = blockIdx.x * ROWS;
  const int nks = K >> 4;   // even
  int nt[TJ];
  bool on[TJ];
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj) { nt[tj] = wave + 8 * tj; on[tj] = nt[tj] * 32 < N; }
  // staging role: row tid >> 2, k quad tid & 3 of every 16-k step
  const int sr = tid >> 2, sc = tid & 3;
  const BufRsrc rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + (size_t)m_blk * lda), 0, 0xfffffffc, 0x00020000);
  const unsigned aoff = ((unsigned)sr * (unsigned)lda + (unsigned)sc * 4u) * 4u;
  char* const swr = smem + sr * XP + sc * 8;
  const int i = lane & 31, h = lane >> 5;
  const char* const fa = smem + i * XP + h * 16;
  const BufRsrc rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<x3raw*>(W3), 0, 0x7ffffff0, 0x00020000);
  const unsigned boff = (unsigned)lane * 16u;
  auto load_b = [&](int ks, vu4x (&b)[TJ][NP]) {
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
      const unsigned soff = (unsigned)((on[tj] ? nt[tj] : 0) * nks + ks) * (NP * 1024u);
#pragma unroll
      for (int pl = 0; pl < NP; ++pl)
        b[tj][pl] = __builtin_bit_cast(vu4x, __builtin_amdgcn_raw_buffer_load_b128(rsW, boff, soff + pl * 1024, 0));
    }
  };
  auto load_a = [&](int ks) {
    return __builtin_bit_cast(vf4, __builtin_amdgcn_raw_buffer_load_b128(rsA, aoff, (unsigned)ks * 64u, RNB_AUX_LD));
  };
  auto stage = [&](const vf4& x, int buf) {
    char* w = swr + buf * BUFB;
    vu2x hi, mid, lo;
    x3_split4(x, hi, mid, lo);
    *reinterpret_cast<vu2x*>(w) = hi;
    *reinterpret_cast<vu2x*>(w + PLB) = mid;
    *reinterpret_cast<vu2x*>(w + 2 * PLB) = lo;
  };
  v16f acc[4][TJ];
#pragma unroll
  for (int ti = 0; ti < 4; ++ti)
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;
  vu4x b0[TJ][NP], b1[TJ][NP];
  load_b(0, b0);
  load_b(1, b1);
  vf4 x = load_a(0);
  stage(x, 0);
  x = load_a(1);
  auto step = [&](int ks, vu4x (&b)[TJ][NP]) {
    // buffer ks & 1 holds step ks (written before the barrier); buffer (ks + 1) & 1 was read in step ks - 1: free
    lds_barrier();
    stage(x, (ks + 1) & 1);                        // step ks + 1 (past the end: a harmless re-stage of the last step)
    x = load_a(min(ks + 2, nks - 1));
    const char* f = fa + (ks & 1) * BUFB;
    vu4x a[4][NP];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) a[ti][pl] = *reinterpret_cast<const vu4x*>(f + pl * PLB + ti * 32 * XP);
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};   // small terms first (x3_mfma)
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
      if (TJ == 2 && tj == 1 && !on[1]) continue;   // (wave-uniform)
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
          acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(x3bf8, a[ti][PA[t]]),
                                                                __builtin_bit_cast(x3bf8, b[tj][PB[t]]), acc[ti][tj], 0, 0, 0);
    }
    load_b(min(ks + 2, nks - 1), b);
  };
  for (int ks = 0; ks < nks; ks += 2) {
    step(ks, b0);
    step(ks + 1, b1);
  }
  __syncthreads();   // the staging buffers become the epilogue strips
  if (amax != nullptr) {   // (workgroup-uniform)  the eight waves' maxima meet in LDS: one atomic per workgroup
    __shared__ float wmx[8];
    const long long left = m_real - m_blk;
    float m = 0.f;
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
      if (!on[tj]) continue;
      v16f t[4][1];
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) t[ti][0] = acc[ti][tj];
      m = fmaxf(m, acc_absmax<4, 1>(t, lane, left >= ROWS ? ROWS : (int)(left < 0 ? 0 : left)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) wmx[wave] = m;
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < 8; ++w) m = fmaxf(m, wmx[w]);
      const unsigned b = __builtin_bit_cast(unsigned, m);
      if (b > __atomic_load_n(amax, __ATOMIC_RELAXED)) atomicMax(amax, b);
    }
  }
  float* strip = reinterpret_cast<float*>(smem + wave * STRIP);
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj) {
    if (!on[tj]) continue;
    v16f t[4][1];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) t[ti][0] = acc[ti][tj];
    run_epilogue<1, Epi, 4>(t, strip, m_blk, nt[tj] * 32, lane, 1u, epi);
  }
}

// ---- dW for 256 x 256 weight matrices as six bf16 MFMA terms (RNB_VARIANT_X3) --------------------------------------
// Same ownership as the staged kernel (one workgroup = the whole 256 x 256 gradient of a point range, slabs + ordered
// reduction), but the operands are split ONCE per workgroup on their way into LDS.  Staging: wave (operand o, point
// quad q) loads rows m + 4q .. + 4 of X_o, one dwordx4 per lane = ONE whole 1 KB row per instruction (thread = 4
// consecutive columns x 4 points; 4 loads per 16-point chunk — sixteen dword loads per thread, the first version,
// filled the vector-memory queue: half of every chunk's time went into issuing them, tools/dwx3_bench), splits each
// column's four points into hi / mid / lo (x3_split4) and writes 8-byte half units.  LDS image of a chunk:
// [operand][plane][point half][unit(column)] x 16 bytes with unit(c) = 68 (c & 3) + (c >> 2): the writer's lanes (column
// group c >> 2, fixed c & 3) and the reader's lanes (32 consecutive columns, ds_read_b128 of the MFMA operand of lane
// (column, half)) are both conflict-free.  8 waves: wave (wm, wn) owns rows 64 wm .. + 64, columns 128 wn .. + 128 of dW
// (128 accumulator registers; two waves per SIMD leave each 256).  Per chunk a wave issues 18 fragment reads and 48
// MFMAs; the split of the next chunk rides in the MFMA gaps; the raw rows run TWO chunks ahead in two register sets.
constexpr int kX3Chunk = 16;
constexpr int kX3Half = 4 * 68 * 16;            // one point half of one plane: 272 units (4 column residues x 68)
constexpr int kX3Plane = 2 * kX3Half;
constexpr int kX3OpBytes = 3 * kX3Plane;        // one operand of one chunk: 25.5 KB
constexpr int kX3BufBytes = 2 * kX3OpBytes;     // both operands
// NP planes per operand (3: bf16 hi / mid / lo, six terms; 2: fp16 hi / lo, three terms — "x2h")
template <int NP> constexpr int dw_op_bytes() { return NP * kX3Plane; }
template <int NP> constexpr int dw_buf_bytes() { return 2 * NP * kX3Plane; }

// (buffer loads: the lane's column offset in one VGPR, the wave-uniform row offset in the scalar operand)
__device__ inline void dw_x3_load(BufRsrc rs, unsigned voff, int ld, int row0, vf4 (&x)[4]) {
#pragma unroll
  for (int p = 0; p < 4; ++p)
    x[p] = __builtin_bit_cast(vf4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (unsigned)(row0 + p) * (unsigned)ld * 4u, RNB_AUX_LD));   // read once
}
// 4 columns x 4 points of one thread -> 12 half units at w (+ 68 * 16 per column, + kX3Plane per plane)
// one column (four points) -> its NP plane units; sc: the operand's scale (x2h only)
template <int NP>
__device__ inline void dw_xn_split_col(const vf4& col, float sc, vu2x (&pl)[NP]) {
  if constexpr (NP == 3) x3_split4(col, pl[0], pl[1], pl[2]);
  else x2h_split4(col * sc, pl[0], pl[1]);
}
template <int NP>
__device__ inline void dw_x3_split(const vf4 (&x)[4], float sc, vu2x (&pl)[4][NP]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) dw_xn_split_col<NP>(vf4{x[0][j], x[1][j], x[2][j], x[3][j]}, sc, pl[j]);
}
template <int NP>
__device__ inline void dw_x3_store(char* w, const vu2x (&pl)[4][NP]) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int q = 0; q < NP; ++q) *reinterpret_cast<vu2x*>(w + j * 68 * 16 + q * kX3Plane) = pl[j][q];
}
// the NT terms of one (ti, tj) block, small ones first
template <int NP>
__device__ inline v16f dw_xn_mfma(const vu4x (&a)[NP], const vu4x (&b)[NP], v16f c, int t) {
  constexpr int PA[6] = {NP == 3 ? 2 : 1, 0, NP == 3 ? 1 : 0, 1, 0, 0};
  constexpr int PB[6] = {0, NP == 3 ? 2 : 1, NP == 3 ? 1 : 0, 0, 1, 0};
  if constexpr (NP == 3)
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(x3bf8, a[PA[t]]), __builtin_bit_cast(x3bf8, b[PB[t]]), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(x2h8, a[PA[t]]), __builtin_bit_cast(x2h8, b[PB[t]]), c, 0, 0, 0);
}
// One chunk of one wave: the 48 MFMAs on the fragments at fx / fy (one column tile of Y at a time, the next tile's
// fragments requested before the current tile's MFMAs), and — in the MFMA gaps, three vector instructions behind each
// MFMA — the split of the raw rows `x` of a later chunk, written to `w` at the end.
template <int NP>
__device__ inline void dw_x3_chunk(const char* fx, const char* fy, v16f (&acc)[2][4], const vf4 (&x)[4], float sc, char* w) {
  constexpr int NT = NP == 3 ? 6 : 3;
  vu4x a[2][NP], b[2][NP];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) a[t][pl] = *reinterpret_cast<const vu4x*>(fx + pl * kX3Plane + t * 128);
#pragma unroll
  for (int pl = 0; pl < NP; ++pl) b[0][pl] = *reinterpret_cast<const vu4x*>(fy + pl * kX3Plane);
#pragma unroll
  for (int tj = 0; tj < 4; ++tj) {
    if (tj + 1 < 4) {
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) b[(tj + 1) & 1][pl] = *reinterpret_cast<const vu4x*>(fy + pl * kX3Plane + (tj + 1) * 128);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) acc[ti][tj] = dw_xn_mfma<NP>(a[ti], b[tj & 1], acc[ti][tj], t);
    {   // column tj of this thread's 4 x 4 raw block: split and stored while tile tj multiplies
      vu2x pl[NP];
      dw_xn_split_col<NP>(vf4{x[0][tj], x[1][tj], x[2][tj], x[3][tj]}, sc, pl);
#pragma unroll
      for (int q = 0; q < NP; ++q) *reinterpret_cast<vu2x*>(w + tj * 68 * 16 + q * kX3Plane) = pl[q];
    }
  }
  // schedule of the region: per column tile its fragment reads (of the NEXT tile), its 2 NT MFMAs with the vector work of
  // one raw column between them, then that column's stores
  __builtin_amdgcn_sched_group_barrier(0x100, 3 * NP, 0);   // a and b[0]
#pragma unroll
  for (int tj = 0; tj < 4; ++tj) {
    if (tj + 1 < 4) __builtin_amdgcn_sched_group_barrier(0x100, NP, 0);
#pragma unroll
    for (int m = 0; m < 2 * NT; ++m) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x200, NP, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
}
// The same for a NARROW job (Y operand of 64 columns: the PE-input layer, the tail of the albedo net's 320-wide first
// layer): wave (wm, wn) owns rows 64 wm .. + 64, columns 32 wn .. + 32 — 12 MFMAs per chunk; the staging split of the
// thread's whole 4 x 4 raw block rides between them (st_on: lanes that stage nothing skip the stores).
template <int NP>
__device__ inline void dw_x3_chunk_narrow(const char* fx, const char* fy, v16f (&acc)[2][1], const vf4 (&x)[4], float sc, char* w,
                                          bool st_on) {
  constexpr int NT = NP == 3 ? 6 : 3;
  vu4x a[2][NP], b[NP];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) a[t][pl] = *reinterpret_cast<const vu4x*>(fx + pl * kX3Plane + t * 128);
#pragma unroll
  for (int pl = 0; pl < NP; ++pl) b[pl] = *reinterpret_cast<const vu4x*>(fy + pl * kX3Plane);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) acc[ti][0] = dw_xn_mfma<NP>(a[ti], b, acc[ti][0], t);
  vu2x pl[4][NP];
  dw_x3_split<NP>(x, sc, pl);
  __builtin_amdgcn_sched_group_barrier(0x100, 3 * NP, 0);
#pragma unroll
  for (int m = 0; m < 2 * NT; ++m) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  if (st_on) dw_x3_store<NP>(w, pl);
}
__device__ inline void dw_x3_colsum(const vf4 (&x)[4], bool on, double (&bs)[4]) {
  if (!on) return;
#pragma unroll
  for (int j = 0; j < 4; ++j) bs[j] += (double)((x[0][j] + x[1][j]) + (x[2][j] + x[3][j]));
}

// DUMMY == 1 (tools/dwx3_bench only): wave 0 sums the clocks it spends waiting at the barrier / issuing a chunk's
// reads, MFMAs, split and stores / issuing the next loads, and leaves them in J.db (as uint64[8] per workgroup)
template <int DUMMY, bool NARROW, int NP = 3>
__device__ inline void dw_x3_body(const DwGroup& g, const DwJob& J, int split, char* lds) {
  constexpr int kOp = dw_op_bytes<NP>(), kBuf = dw_buf_bytes<NP>();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
  const int wm = wave >> 1, wn = wave & 1;
  const int m_begin = split * J.rows_per_split;
  const int m_end = min(g.M, m_begin + J.rows_per_split);
  if (m_begin >= m_end) return;   // (workgroup-uniform)
  const int nchunks = (m_end - m_begin) / kX3Chunk;   // even: ranges are multiples of 32 points (host)
  // narrow job: the Y operand has 64 columns (J.K == 64); everything about X and the row split stays
  constexpr bool narrow = NARROW;
  // staging role of this thread: columns 4 cg .. + 4, points 4 pq .. + 4 of operand sop
  const int cg = lane, pq = wave & 3, sop = wave >> 2;
  const bool st_on = !(narrow && sop == 1 && cg >= 16);   // a narrow Y row is 16 column groups
  char* const swr = lds + sop * kOp + (pq >> 1) * kX3Half + cg * 16 + (pq & 1) * 8;   // + buffer, column, plane
  // fragment addresses of this lane: column 64 wm (128 wn) + 32 t + i of the operand, point half h
  const int i = lane & 31, h = lane >> 5;
  const int ui = (i & 3) * 68 + (i >> 2);
  const char* const fx = lds + h * kX3Half + (ui + 16 * wm) * 16;
  const char* const fy = lds + kOp + h * kX3Half + (ui + (narrow ? 8 : 32) * wn) * 16;
  constexpr int NTJ = NARROW ? 1 : 4;
  v16f acc[2][NTJ];
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < NTJ; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;
  double bs[4] = {0.0, 0.0, 0.0, 0.0};
  // x2h: one scale for the adjoint operands of all pairs of the job (they share the accumulators), from the larger of
  // their recorded maxima; the state operands (activations, Jacobian rows, network inputs) carry kH2ActScale
  [[maybe_unused]] float s_adj = 1.f, s_state = kH2ActScale, unscale = 1.f;
  if constexpr (NP == 2) {
    unsigned mb = J.p1.amax ? *J.p1.amax : 0u;
    if (J.npairs > 1 && J.p2.amax) mb = max(mb, *J.p2.amax);
    float inv;
    x2h_dyn_scale(mb, s_adj, inv);
    // the state operands: 2^6 (the round-4 constant: results unchanged) while their recorded maximum stays below 2^8, else
    // the power of two that puts it in [2^13, 2^14) — no saved activation / Jacobian row is out of range
    unsigned sb = J.p1.smax ? *J.p1.smax : 0u;
    if (J.npairs > 1 && J.p2.smax) sb = max(sb, *J.p2.smax);
    float inv_state = 1.f / kH2ActScale;
    if ((sb >> 23) >= 127u + 8u && (sb >> 23) < 255u) x2h_dyn_scale(sb, s_state, inv_state);
    unscale = inv * inv_state;
  }
  [[maybe_unused]] unsigned long long t_bar = 0, t_chunk = 0, t_load = 0, t_all = 0;
  [[maybe_unused]] const unsigned long long t_begin = DUMMY == 1 ? __builtin_amdgcn_s_memtime() : 0;
  [[maybe_unused]] const unsigned long long r_begin = DUMMY == 1 ? __builtin_amdgcn_s_memrealtime() : 0;
  for (int pi = 0; pi < J.npairs; ++pi) {
    const DwPair p = pi == 0 ? J.p1 : J.p2;
    const int ld = sop == 0 ? p.ldx : p.ldy;
    [[maybe_unused]] const float sc = NP == 2 ? (sop == p.adj ? s_adj : s_state) : 1.f;   // this thread's operand
    // resource based at this split's first row: 32-bit offsets stay inside the split whatever the total point count
    const BufRsrc src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((sop == 0 ? p.X : p.Y) + (size_t)m_begin * ld), 0,
                                                          0xfffffffc, 0x00020000);
    const unsigned voff = st_on ? 16u * (unsigned)cg : 0u;   // (lanes that stage nothing re-read column group 0)
    const bool do_bias = DUMMY == 0 && J.db != nullptr && pi == J.bias_pair && sop == 0;
    const int last = nchunks - 1;
    const int r0 = 4 * pq;   // (rows relative to the split)
    vf4 x0[4], x1[4];   // raw rows of an even / odd chunk
    dw_x3_load(src, voff, ld, r0, x0);
    dw_x3_load(src, voff, ld, r0 + min(1, last) * kX3Chunk, x1);
    __builtin_amdgcn_s_barrier();   // every wave is done with the buffers of the previous pair
    {
      vu2x pl[4][NP];
      dw_x3_split<NP>(x0, sc, pl);
      if (st_on) dw_x3_store<NP>(swr, pl);
      dw_x3_colsum(x0, do_bias, bs);
    }
    dw_x3_load(src, voff, ld, r0 + min(2, last) * kX3Chunk, x0);
    for (int c = 0; c < nchunks; c += 2) {
      // even chunk c from buffer 0; chunk c + 1 (x1) -> buffer 1; x1 <- chunk c + 3
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      [[maybe_unused]] const unsigned long long s0 = DUMMY == 1 ? __builtin_amdgcn_s_memtime() : 0;
      __builtin_amdgcn_s_barrier();
      [[maybe_unused]] const unsigned long long s1 = DUMMY == 1 ? __builtin_amdgcn_s_memtime() : 0;
      if constexpr (narrow) dw_x3_chunk_narrow<NP>(fx, fy, acc, x1, sc, swr + kBuf, st_on);
      else dw_x3_chunk<NP>(fx, fy, acc, x1, sc, swr + kBuf);
      dw_x3_colsum(x1, do_bias, bs);   // (nchunks even: chunk c + 1 always exists)
      [[maybe_unused]] const unsigned long long s2 = DUMMY == 1 ? __builtin_amdgcn_s_memtime() : 0;
      dw_x3_load(src, voff, ld, r0 + min(c + 3, last) * kX3Chunk, x1);
      if constexpr (DUMMY == 1) {
        const unsigned long long s3 = __builtin_amdgcn_s_memtime();
        t_bar += s1 - s0; t_chunk += s2 - s1; t_load += s3 - s2;
      }
      // odd chunk c + 1 from buffer 1; chunk c + 2 (x0) -> buffer 0 (past the end: a re-split of the last chunk that
      // nobody reads); x0 <- chunk c + 4
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if constexpr (narrow) dw_x3_chunk_narrow<NP>(fx + kBuf, fy + kBuf, acc, x0, sc, swr, st_on);
      else dw_x3_chunk<NP>(fx + kBuf, fy + kBuf, acc, x0, sc, swr);
      dw_x3_colsum(x0, do_bias && c + 2 < nchunks, bs);
      dw_x3_load(src, voff, ld, r0 + min(c + 4, last) * kX3Chunk, x0);
    }
  }
  // accumulator (ti, tj, r) of lane (i, h) is dW[64 wm + 32 ti + rho][128 wn + 32 tj + i], rho = (r & 3) + 8 (r >> 2) + 4 h
  // slabs are compact [split][N][K] (K = the job's Y columns; dw_reduce_kernel scatters them into dW with lddw)
  const int lddw = J.lddw, Kj = J.K;
  float* __restrict__ pdst = J.part ? J.part + (size_t)split * J.N * Kj : nullptr;
#pragma unroll
  for (int tj = 0; tj < NTJ; ++tj) {
    const int col = narrow ? wn * 32 + i : wn * 128 + tj * 32 + i;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float v = NP == 2 ? acc[ti][tj][r] * unscale : acc[ti][tj][r];
        if (pdst) __builtin_nontemporal_store(v, pdst + (size_t)row * Kj + col);
        else atomicAdd(J.dW + (size_t)row * lddw + col, v);
      }
  }
  if constexpr (DUMMY == 1) {
    if (tid == 0) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(J.db) + 8 * (size_t)blockIdx.x;
      t_all = __builtin_amdgcn_s_memtime() - t_begin;
      o[0] = t_bar; o[1] = t_chunk; o[2] = t_load; o[3] = t_all;
      o[4] = __builtin_amdgcn_s_memrealtime() - r_begin;   // 100 MHz
    }
  }
  if (DUMMY == 0 && J.db != nullptr) {   // column sums of the bias pair's X operand: the four point quads of a column meet in LDS
    double* bx = reinterpret_cast<double*>(lds);
    __syncthreads();
    if (sop == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) bx[pq * 256 + 4 * cg + j] = bs[j];
    }
    __syncthreads();
    if (tid < 256) {
      const float v = (float)((bx[tid] + bx[256 + tid]) + (bx[512 + tid] + bx[768 + tid]));
      if (J.partb) J.partb[(size_t)split * J.N + tid] = v;
      else atomicAdd(J.db + tid, v);
    }
  }
}


template <int DUMMY, int NP = 3>
__global__ __launch_bounds__(512, 1) void gemm_dw_x3_kernel(const DwGroup g) {
  __shared__ __attribute__((aligned(16))) char lds[2 * dw_buf_bytes<NP>() > 8192 ? 2 * dw_buf_bytes<NP>() : 8192];   // 102 KB (NP = 2: 68 KB)
  int ji = 0, begin = 0;
  for (int q = 0; q + 1 < g.njobs; ++q)
    if ((int)blockIdx.x >= g.job[q].block_end) { ji = q + 1; begin = g.job[q].block_end; }
  const DwJob& J = g.job[ji];
  const int split = (int)blockIdx.x - begin;
  if (split >= J.splits) return;
  // two bodies, one per job width (workgroup-uniform): separate accumulator sets, separate register allocation
  if (J.K < 256) dw_x3_body<DUMMY, true, NP>(g, J, split, lds);
  else dw_x3_body<DUMMY, false, NP>(g, J, split, lds);
}

// ---- activation helpers ----------------------------------------------------------------------------
// nn.Softplus(beta=100) with PyTorch's threshold 20 (models/fields.py:80)
__device__ inline float softplus100(float z) {
  const float t = z * 100.f;
  return t > 20.f ? z : log1pf(expf(t)) * 0.01f;
}
// softplus and its derivative in one go: a = softplus_100(z), D = sigmoid(100 z).
// On gfx950 the fp32 MFMA and ordinary vector instructions exclude each other on a SIMD, so this function is
// paid for in matrix time: it is written for instruction count (16 VALU + 3 transcendentals per element,
// most of them packable two elements at a time — see the vf2 overload).
//   w = exp(-|t|), t = 100 z        (never overflows; no range clamp needed)
//   a = max(z, 0) + log1p(w) / 100  (for t > 20 the second term is below half an ulp of z: a == z exactly,
//                                    PyTorch's threshold branch, models/fields.py:80)
//   D = 1 / (1 + w)  (t >= 0)   |   w / (1 + w)  (t < 0)        (== 1 exactly for t > 20)
// exp through the hardware exp2 with the rounding error of the product -|t| log2(e) fed back to first order;
// log1p(w) = log(u) + (w - (u - 1)) / u with u = fl(1 + w) (rounding-compensated, hardware log2 and rcp).
__device__ inline void softplus_aD(float z, float& a, float& D) {
  constexpr float L2E = 1.44269504088896341f, LN2 = 0.693147180559945309f;
  const float t = z * 100.f;
  const float nt = -fabsf(t);
  const float p = nt * L2E;
  const float q = __builtin_fmaf(nt, L2E, -p);
  const float w0 = __builtin_amdgcn_exp2f(p);
  const float w = __builtin_fmaf(w0, q * LN2, w0);
  const float u = 1.f + w;
  const float r = __builtin_amdgcn_rcpf(u);
  const float lg = __builtin_amdgcn_logf(u);
  const float l1p = __builtin_fmaf(lg, LN2, (w - (u - 1.f)) * r);
  a = __builtin_fmaf(l1p, 0.01f, fmaxf(z, 0.f));
  D = t >= 0.f ? r : w * r;
}
// two elements at a time: the multiplies / adds / fmas become v_pk_*_f32 (one issue slot for both)
__device__ inline void softplus_aD(vf2 z, vf2& a, vf2& D) {
  constexpr float L2E = 1.44269504088896341f, LN2 = 0.693147180559945309f;
  const vf2 t = z * 100.f;
  const vf2 nt = {-fabsf(t.x), -fabsf(t.y)};
  const vf2 p = nt * L2E;
  const vf2 q = __builtin_elementwise_fma(nt, vf2{L2E, L2E}, -p);
  const vf2 w0 = {__builtin_amdgcn_exp2f(p.x), __builtin_amdgcn_exp2f(p.y)};
  const vf2 w = __builtin_elementwise_fma(w0, q * LN2, w0);
  const vf2 u = w + 1.f;
  const vf2 r = {__builtin_amdgcn_rcpf(u.x), __builtin_amdgcn_rcpf(u.y)};
  const vf2 lg = {__builtin_amdgcn_logf(u.x), __builtin_amdgcn_logf(u.y)};
  const vf2 l1p = __builtin_elementwise_fma(lg, vf2{LN2, LN2}, (w - (u - 1.f)) * r);
  const vf2 zp = {fmaxf(z.x, 0.f), fmaxf(z.y, 0.f)};
  a = __builtin_elementwise_fma(l1p, vf2{0.01f, 0.01f}, zp);
  const vf2 wr = w * r;
  D = vf2{t.x >= 0.f ? r.x : wr.x, t.y >= 0.f ? r.y : wr.y};
}
// a alone (forward-only evaluations, e.g. the sampling passes): no reciprocal — the compensation term
// (w - (u - 1)) / u is at most half an ulp of u, so 1 / u ~ 1 - w inside it changes a by < eps w / 2.
__device__ inline vf2 softplus_a(vf2 z) {
  constexpr float L2E = 1.44269504088896341f, LN2 = 0.693147180559945309f;
  const vf2 t = z * 100.f;
  const vf2 nt = {-fabsf(t.x), -fabsf(t.y)};
  const vf2 p = nt * L2E;
  const vf2 q = __builtin_elementwise_fma(nt, vf2{L2E, L2E}, -p);
  const vf2 w0 = {__builtin_amdgcn_exp2f(p.x), __builtin_amdgcn_exp2f(p.y)};
  const vf2 w = __builtin_elementwise_fma(w0, q * LN2, w0);
  const vf2 u = w + 1.f;
  const vf2 lg = {__builtin_amdgcn_logf(u.x), __builtin_amdgcn_logf(u.y)};
  const vf2 d = w - (u - 1.f);
  const vf2 l1p = __builtin_elementwise_fma(lg, vf2{LN2, LN2}, __builtin_elementwise_fma(-d, w, d));
  const vf2 zp = {fmaxf(z.x, 0.f), fmaxf(z.y, 0.f)};
  return __builtin_elementwise_fma(l1p, vf2{0.01f, 0.01f}, zp);
}
// The same two functions on scalars, for the x3 kernels: there the epilogue of one wave runs beside the other wave's bf16
// MFMAs, where packed fp32 instructions are an anti-lever (MI355X_MICROARCH.md: a v_pk_* costs ~13 cycles more than the
// two scalar instructions it replaces).  Bit-identical results (same operations, element-wise).
__device__ inline float softplus_a(float z) {
  constexpr float L2E = 1.44269504088896341f, LN2 = 0.693147180559945309f;
  const float t = z * 100.f;
  const float nt = -fabsf(t);
  const float p = nt * L2E;
  const float q = __builtin_fmaf(nt, L2E, -p);
  const float w0 = __builtin_amdgcn_exp2f(p);
  const float w = __builtin_fmaf(w0, q * LN2, w0);
  const float u = w + 1.f;
  const float lg = __builtin_amdgcn_logf(u);
  const float d = w - (u - 1.f);
  const float l1p = __builtin_fmaf(lg, LN2, __builtin_fmaf(-d, w, d));
  return __builtin_fmaf(l1p, 0.01f, fmaxf(z, 0.f));
}
template <bool SCALAR>
__device__ inline vf2 softplus_a_sel(vf2 z) {
  if constexpr (SCALAR) return vf2{softplus_a(z.x), softplus_a(z.y)};
  else return softplus_a(z);
}
template <bool SCALAR>
__device__ inline void softplus_aD_sel(vf2 z, vf2& a, vf2& D) {
  if constexpr (SCALAR) {
    float a0, a1, D0, D1;
    softplus_aD(z.x, a0, D0);
    softplus_aD(z.y, a1, D1);
    a = vf2{a0, a1};
    D = vf2{D0, D1};
  }
  else softplus_aD(z, a, D);
}
// D = d softplus / dz = sigmoid(100 z) expressed through a = softplus(z):  D = 1 - exp(-100 a)
// (exactly 1 above the threshold, where a == z); E = 1 - D, and softplus'' = 100 D E.
__device__ inline void softplus_DE(float a, float& D, float& E) {
  const float t = a * 100.f;
  if (t > 20.f) { D = 1.f; E = 0.f; }
  else { D = -expm1f(-t); E = expf(-t); }
}
__device__ inline float softplus_D(float a) {
  const float t = a * 100.f;
  return t > 20.f ? 1.f : -expm1f(-t);
}

}  // namespace rnb
