// fp32 MFMA GEMM building blocks for gfx950 (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fma chain).
//
// One workgroup = 256 threads = 4 wave64 arranged 2x2; block tile 128x128, K-step 32; each wave owns a
// 64x64 sub-tile = 2x2 MFMA tiles of 32x32 (4 x 16 accumulator registers per lane).  Operand tiles are
// staged through LDS with a register prefetch of the next K-step (global loads in flight while the
// matrix cores run).  fp32 MFMA retires 2 k per 64 cycles per SIMD, so operand bandwidth is far from
// binding; the layouts below are chosen for conflict-free ds_read_b128 / ds_read_b32.
//
// The reduction index inside a K-step is permuted: within each group of 8 k, lane half h (= lane>>5)
// supplies k = 8q+4h+c for MFMA step c (one 16-byte LDS read feeds 4 MFMAs).  A and B use the same
// permutation, so the product is unchanged up to fp32 summation order.
#pragma once
#include <hip/hip_runtime.h>

namespace rnb {

typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int BM = 128;
constexpr int BN = 128;
constexpr int BK = 32;
constexpr int LDK = BK + 4;   // pitch of a k-contiguous tile  [128][36]
constexpr int LDN = 128;      // pitch of a k-major tile       [32][128]
constexpr int TILE_FLOATS = 128 * LDK;  // 4608 floats >= 32*128

// ---- global -> register staging ------------------------------------------------------------------
// k-contiguous source: element (r, k) at src[r*ld + k]; tile = rows r0..r0+127, k0..k0+31.
// Rows >= rmax are zero-filled.
__device__ inline void load_rows(const float* __restrict__ src, int ld, int r0, int k0, int rmax, int tid,
                                 float4 (&v)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int idx = tid + 256 * i;
    int r = idx >> 3, c4 = idx & 7;
    if (r0 + r < rmax)
      v[i] = *reinterpret_cast<const float4*>(src + (size_t)(r0 + r) * ld + k0 + c4 * 4);
    else
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
__device__ inline void store_rows(float* __restrict__ T, int tid, const float4 (&v)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int idx = tid + 256 * i;
    int r = idx >> 3, c4 = idx & 7;
    *reinterpret_cast<float4*>(T + r * LDK + c4 * 4) = v[i];
  }
}
// k-major source: element (k, c) at src[k*ld + c]; tile = k rows k0..k0+31, columns c0..c0+127.
// k >= kmax or c >= cmax are zero-filled.
__device__ inline void load_kmajor(const float* __restrict__ src, int ld, int k0, int c0, int kmax, int cmax,
                                   int tid, float4 (&v)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int idx = tid + 256 * i;
    int kk = idx >> 5, c4 = idx & 31;
    if (k0 + kk < kmax && c0 + c4 * 4 < cmax)
      v[i] = *reinterpret_cast<const float4*>(src + (size_t)(k0 + kk) * ld + c0 + c4 * 4);
    else
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
__device__ inline void store_kmajor(float* __restrict__ T, int tid, const float4 (&v)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int idx = tid + 256 * i;
    int kk = idx >> 5, c4 = idx & 31;
    *reinterpret_cast<float4*>(T + kk * LDN + c4 * 4) = v[i];
  }
}

// ---- LDS -> MFMA fragments ---------------------------------------------------------------------
template <bool KMAJOR>
__device__ inline void frag4(const float* __restrict__ T, int idx, int q, int h, float (&o)[4]) {
  if constexpr (!KMAJOR) {
    float4 t = *reinterpret_cast<const float4*>(T + idx * LDK + q * 8 + h * 4);
    o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = T[(q * 8 + h * 4 + c) * LDN + idx];
  }
}

// One K-step (32 k) of a wave's 64x64 sub-tile.  tile_on[j] says whether column tile j is inside N.
template <bool A_KMAJOR, bool B_KMAJOR>
__device__ inline void mma_step(const float* __restrict__ As, const float* __restrict__ Bs, int wm, int wn,
                                int lane, bool on0, bool on1, v16f (&acc)[2][2]) {
  const int i = lane & 31, h = lane >> 5;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float a0[4], a1[4], b0[4], b1[4];
    frag4<A_KMAJOR>(As, wm * 64 + i, q, h, a0);
    frag4<A_KMAJOR>(As, wm * 64 + 32 + i, q, h, a1);
    frag4<B_KMAJOR>(Bs, wn * 64 + i, q, h, b0);
    frag4<B_KMAJOR>(Bs, wn * 64 + 32 + i, q, h, b1);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (on0) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[c], b0[c], acc[0][0], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[c], b0[c], acc[1][0], 0, 0, 0);
      }
      if (on1) {
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[c], b1[c], acc[0][1], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[c], b1[c], acc[1][1], 0, 0, 0);
      }
    }
  }
}

// Accumulator element (tile ti,tj ; register r) of lane `lane` -> (row, col) inside the 128x128 block.
__device__ inline int acc_row(int wm, int ti, int r, int lane) {
  return wm * 64 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}
__device__ inline int acc_col(int wn, int tj, int lane) { return wn * 64 + tj * 32 + (lane & 31); }

template <class Epi>
__device__ inline void run_epilogue(const v16f (&acc)[2][2], int m_blk, int n_blk, int wm, int wn, int lane,
                                    bool on0, bool on1, const Epi& epi) {
#pragma unroll
  for (int tj = 0; tj < 2; ++tj) {
    if (!(tj == 0 ? on0 : on1)) continue;
    const int col = n_blk + acc_col(wn, tj, lane);
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m_blk + acc_row(wm, ti, r, lane);
        epi(row, col, acc[ti][tj][r]);
      }
    }
  }
}

// ---- C[M x N] = A[M x K] * op(B) ------------------------------------------------------------------
//   B_KMAJOR == false ("NT"): B given as W[N][K] (k contiguous): C = A W^T      (forward-shaped layers)
//   B_KMAJOR == true  ("NN"): B given as W[K][N] (n contiguous): C = A W        (reverse-shaped layers)
// M is a multiple of 128 (padded buffers), N and K multiples of 32.  grid = (M/128, ceil(N/128)).
template <bool B_KMAJOR, class Epi>
__global__ __launch_bounds__(256, 2) void gemm_rows_kernel(const float* __restrict__ A, int lda,
                                                           const float* __restrict__ W, int ldw, int N,
                                                           int K, Epi epi) {
  __shared__ __attribute__((aligned(16))) float smem[2 * TILE_FLOATS];
  float* As = smem;
  float* Bs = smem + TILE_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m_blk = blockIdx.x * BM, n_blk = blockIdx.y * BN;
  const bool on0 = n_blk + wn * 64 < N, on1 = n_blk + wn * 64 + 32 < N;

  v16f acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  float4 ra[4], rb[4];
  const int nk = K / BK;
  if (nk > 0) {
    load_rows(A, lda, m_blk, 0, 0x7fffffff, tid, ra);
    if constexpr (!B_KMAJOR) load_rows(W, ldw, n_blk, 0, N, tid, rb);
    else load_kmajor(W, ldw, 0, n_blk, K, N, tid, rb);
  }
  for (int kt = 0; kt < nk; ++kt) {
    store_rows(As, tid, ra);
    if constexpr (!B_KMAJOR) store_rows(Bs, tid, rb);
    else store_kmajor(Bs, tid, rb);
    __syncthreads();
    if (kt + 1 < nk) {
      const int k0 = (kt + 1) * BK;
      load_rows(A, lda, m_blk, k0, 0x7fffffff, tid, ra);
      if constexpr (!B_KMAJOR) load_rows(W, ldw, n_blk, k0, N, tid, rb);
      else load_kmajor(W, ldw, k0, n_blk, K, N, tid, rb);
    }
    mma_step<false, B_KMAJOR>(As, Bs, wm, wn, lane, on0, on1, acc);
    __syncthreads();
  }
  run_epilogue(acc, m_blk, n_blk, wm, wn, lane, on0, on1, epi);
}

// ---- dW[N x K] += X1^T Y1 (+ X2^T Y2), reduction over the M points, split over blockIdx.z ---------
//   X* [M x N] (ldx), Y* [M x K] (ldy); rows >= M are masked.  grid = (ceil(N/128), ceil(K/128), splits).
//   Partial tiles are accumulated into dW with float atomics (dW zero-initialised by the caller);
//   colsum(Xb) over the same rows is added to db when db != nullptr (by the blockIdx.y == 0 blocks).
struct DwPair {
  const float* X;
  int ldx;
  const float* Y;
  int ldy;
};
__global__ __launch_bounds__(256, 2) void gemm_dw_kernel(DwPair p1, DwPair p2, int npairs, int M, int N, int K,
                                                         int rows_per_split, float* __restrict__ dW, int lddw,
                                                         float* __restrict__ db, int bias_pair) {
  __shared__ __attribute__((aligned(16))) float smem[2 * TILE_FLOATS];
  float* Xs = smem;
  float* Ys = smem + TILE_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int n_blk = blockIdx.x * BM, k_blk = blockIdx.y * BN;
  const int m_begin = blockIdx.z * rows_per_split;
  const int m_end = min(M, m_begin + rows_per_split);
  const bool on0 = k_blk + wn * 64 < K, on1 = k_blk + wn * 64 + 32 < K;
  const bool row_on0 = n_blk + wm * 64 < N, row_on1 = n_blk + wm * 64 + 32 < N;
  (void)row_on0; (void)row_on1;

  v16f acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  double bsum = 0.0;   // bias gradients are long signed sums: keep the per-block partial in fp64
  const bool do_bias = (db != nullptr) && blockIdx.y == 0 && tid < 128 && (n_blk + tid < N);

  if (m_begin < m_end) {
    for (int pi = 0; pi < npairs; ++pi) {
      const DwPair p = pi == 0 ? p1 : p2;
      float4 rx[4], ry[4];
      load_kmajor(p.X, p.ldx, m_begin, n_blk, m_end, N, tid, rx);
      load_kmajor(p.Y, p.ldy, m_begin, k_blk, m_end, K, tid, ry);
      for (int m0 = m_begin; m0 < m_end; m0 += BK) {
        store_kmajor(Xs, tid, rx);
        store_kmajor(Ys, tid, ry);
        __syncthreads();
        if (m0 + BK < m_end) {
          load_kmajor(p.X, p.ldx, m0 + BK, n_blk, m_end, N, tid, rx);
          load_kmajor(p.Y, p.ldy, m0 + BK, k_blk, m_end, K, tid, ry);
        }
        if (do_bias && pi == bias_pair) {
#pragma unroll 8
          for (int kk = 0; kk < BK; ++kk) bsum += (double)Xs[kk * LDN + tid];
        }
        mma_step<true, true>(Xs, Ys, wm, wn, lane, on0, on1, acc);
        __syncthreads();
      }
    }
  }
  // atomics: each register of a 32x32 accumulator is two 128-byte row segments per wave instruction
#pragma unroll
  for (int tj = 0; tj < 2; ++tj) {
    if (!(tj == 0 ? on0 : on1)) continue;
    const int col = k_blk + acc_col(wn, tj, lane);
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = n_blk + acc_row(wm, ti, r, lane);
        if (row < N) atomicAdd(dW + (size_t)row * lddw + col, acc[ti][tj][r]);
      }
    }
  }
  if (do_bias) atomicAdd(db + n_blk + tid, (float)bsum);
}

// ---- activation helpers ----------------------------------------------------------------------------
// nn.Softplus(beta=100) with PyTorch's threshold 20 (models/fields.py:80)
__device__ inline float softplus100(float z) {
  const float t = z * 100.f;
  return t > 20.f ? z : log1pf(expf(t)) * 0.01f;
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
