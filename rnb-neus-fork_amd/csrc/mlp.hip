// SDF / albedo network sweeps on a batch of points: forward (F), reverse-mode normal (R), albedo MLP (C)
// and the explicit backward (C', RA, FB, dW) — see oracle/explicit.py for the mathematical statement and
// the reference lines each stage replaces (models/fields.py:82-127, :177-215; the backward replaces
// autograd's double backward invoked at exp_runner.py:261).
#include "gemm.hip.h"
#include "rnb_internal.h"

namespace rnb {

// =====================================================================================================
// point-wise kernels
// =====================================================================================================

// positional encoding (models/embedder.py:40-46): e = [x, sin(2^k x), cos(2^k x)]_k, padded with zeros
__global__ void pe_points_kernel(const float* __restrict__ pts, int64_t M, int64_t Mp, float scale, int multires,
                                 int Ep, float* __restrict__ x4, float* __restrict__ e) {
  int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= Mp) return;
  float x[3] = {0.f, 0.f, 0.f};
  if (row < M) {
    x[0] = pts[row * 3 + 0] * scale;
    x[1] = pts[row * 3 + 1] * scale;
    x[2] = pts[row * 3 + 2] * scale;
  }
  x4[row * 4 + 0] = x[0]; x4[row * 4 + 1] = x[1]; x4[row * 4 + 2] = x[2]; x4[row * 4 + 3] = 0.f;
  float* er = e + row * Ep;
  er[0] = x[0]; er[1] = x[1]; er[2] = x[2];
  int c = 3;
  float f = 1.f;
  for (int k = 0; k < multires; ++k) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      float s, co;
      sincosf(x[d] * f, &s, &co);
      er[c + d] = s;
      er[c + 3 + d] = co;
    }
    c += 6;
    f *= 2.f;
  }
  for (; c < Ep; ++c) er[c] = 0.f;
}

// sdf head: sdf = (a_last . w_sdf + b_sdf)/scale ; optionally seeds the reverse sweep gz_last = w_sdf * D
// 32 lanes per point.
__global__ void sdf_head_kernel(const float* __restrict__ a, const float* __restrict__ D, int Hp, int H,
                                const float* __restrict__ wsdf, const float* __restrict__ bsdf, float inv_scale,
                                int64_t Mp, float* __restrict__ sdf, float* __restrict__ gz) {
  const int sub = threadIdx.x & 31;
  int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
  if (row >= Mp) return;
  const float* ar = a + row * Hp;
  float acc = 0.f;
  for (int k = sub; k < Hp; k += 32) {
    const float av = ar[k];
    const float w = k < H ? wsdf[k] : 0.f;
    acc = fmaf(av, w, acc);
    if (gz) gz[row * Hp + k] = k < H ? w * D[row * Hp + k] : 0.f;
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 32);
  if (sub == 0) sdf[row] = (acc + bsdf[0]) * inv_scale;
}

// normal = J_pe(x)^T g_e   (d sdf / d pts; models/fields.py:114-127)
__global__ void normal_kernel(const float* __restrict__ x4, const float* __restrict__ ge, int Ep, int multires,
                              int64_t Mp, float* __restrict__ nrm) {
  int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= Mp) return;
  const float* g = ge + row * Ep;
  float n[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) n[d] = g[d];
  float f = 1.f;
  int c = 3;
  for (int k = 0; k < multires; ++k) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      float s, co;
      sincosf(x4[row * 4 + d] * f, &s, &co);
      n[d] += f * (g[c + d] * co - g[c + 3 + d] * s);
    }
    c += 6;
    f *= 2.f;
  }
  nrm[row * 4 + 0] = n[0]; nrm[row * 4 + 1] = n[1]; nrm[row * 4 + 2] = n[2]; nrm[row * 4 + 3] = 0.f;
}

// ---- 64-point row tiles through LDS ------------------------------------------------------------------
// The per-point kernels below compute (or consume) W consecutive columns of one matrix row per lane.
// Touching global memory in that shape makes every access instruction hit 64 different rows; instead one
// wave stages its 64 rows x W columns in LDS (pitch W + 1: conflict-free both ways) and moves them with
// 16 bytes per lane along the rows.  W % 4 == 0, col0 % 4 == 0, ld % 4 == 0; one wave per workgroup.
__device__ inline void tile_store64(float* __restrict__ dst, int ld, int64_t r0, int col0, int W,
                                    const float* __restrict__ tile, int lane) {
  const int g = W >> 2;   // 16-byte groups per row
  for (int idx = lane; idx < 64 * g; idx += 64) {
    const int p = idx / g, c = (idx - p * g) * 4;
    const float* t = tile + p * (W + 1) + c;
    *reinterpret_cast<vf4*>(dst + (r0 + p) * ld + col0 + c) = make_vf4(t[0], t[1], t[2], t[3]);
  }
}
__device__ inline void tile_load64(const float* __restrict__ src, int ld, int64_t r0, int col0, int W,
                                   float* __restrict__ tile, int lane) {
  const int g = W >> 2;
  for (int idx = lane; idx < 64 * g; idx += 64) {
    const int p = idx / g, c = (idx - p * g) * 4;
    const vf4 v = *reinterpret_cast<const vf4*>(src + (r0 + p) * ld + col0 + c);
    float* t = tile + p * (W + 1) + c;
    t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
  }
}

// albedo-net input columns F.. : [pe_v(p) | pe_v(n) | 0]  (feature columns 0..F-1 are written by the
// feature-head GEMM).  models/fields.py:179-191 in the packed column order.  One wave = 64 points;
// dynamic LDS = 64 * (Cinp - F + 1) floats.
__global__ __launch_bounds__(64) void color_input_kernel(const float* __restrict__ pts, const float* __restrict__ nrm,
                                                         int nrm_ld, int64_t M, int64_t Mp, int F, int multires,
                                                         int Cinp, float* __restrict__ cin) {
  extern __shared__ float tile[];
  const int lane = threadIdx.x, W = Cinp - F;
  const int64_t r0 = (int64_t)blockIdx.x * 64, row = r0 + lane;
  float* cr = tile + lane * (W + 1);
  int c = 0;
  for (int which = 0; which < 2; ++which) {
    float v[3] = {0.f, 0.f, 0.f};
    if (row < M) {
      if (which == 0) { v[0] = pts[row * 3]; v[1] = pts[row * 3 + 1]; v[2] = pts[row * 3 + 2]; }
      else { v[0] = nrm[row * nrm_ld]; v[1] = nrm[row * nrm_ld + 1]; v[2] = nrm[row * nrm_ld + 2]; }
    }
    cr[c] = v[0]; cr[c + 1] = v[1]; cr[c + 2] = v[2];
    c += 3;
    float f = 1.f;
    for (int k = 0; k < multires; ++k) {
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        float s, co;
        sincosf(v[d] * f, &s, &co);
        cr[c + d] = s;
        cr[c + 3 + d] = co;
      }
      c += 6;
      f *= 2.f;
    }
  }
  for (; c < W; ++c) cr[c] = 0.f;
  __builtin_amdgcn_wave_barrier();
  tile_store64(cin, Cinp, r0, F, W, tile, lane);
}

// albedo output layer (d_out rows) + sigmoid: 32 lanes per point.
// One wave per 4 rows: lane l reads the float4 l (+ 64, ..) of a row — a whole 1 KB row per load instruction, four rows
// in flight — and keeps the output layer's <= 4 weight rows for its columns in registers (Hcp <= 256 * 4).
__global__ __launch_bounds__(256) void color_out_kernel(const float* __restrict__ ac, int Hcp, int Hc, const float* __restrict__ Wo,
                                 int ldwo, const float* __restrict__ bo, int Co, int squeeze, int64_t Mp,
                                 float* __restrict__ alb) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t row0 = wave * 4;
  if (row0 >= Mp) return;
  float acc[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[r][c] = 0.f;
  for (int k4 = lane; k4 * 4 < Hcp; k4 += 64) {
    vf4 w[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      w[c] = make_vf4(0.f, 0.f, 0.f, 0.f);
      if (c < Co) {
        const float* wp = Wo + (size_t)c * ldwo + k4 * 4;
        w[c] = make_vf4(k4 * 4 < Hc ? wp[0] : 0.f, k4 * 4 + 1 < Hc ? wp[1] : 0.f, k4 * 4 + 2 < Hc ? wp[2] : 0.f,
                        k4 * 4 + 3 < Hc ? wp[3] : 0.f);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (row0 + r < Mp) {
        const vf4 a = *reinterpret_cast<const vf4*>(ac + (row0 + r) * Hcp + k4 * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c)
          acc[r][c] = fmaf(a.x, w[c].x, fmaf(a.y, w[c].y, fmaf(a.z, w[c].z, fmaf(a.w, w[c].w, acc[r][c]))));
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc[r][c] += __shfl_xor(acc[r][c], o, 64);
    }
  if (lane < 4 && row0 + lane < Mp) {   // lane r finishes row r
    const int64_t row = row0 + lane;
    float mine[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) mine[c] = lane == 0 ? acc[0][c] : lane == 1 ? acc[1][c] : lane == 2 ? acc[2][c] : acc[3][c];
    for (int c = 0; c < 4; ++c) {
      float v = 0.f;
      if (c < Co) {
        v = mine[c] + bo[c];
        if (squeeze) v = 1.f / (1.f + expf(-v));
      }
      alb[row * 4 + c] = v;
    }
  }
}

// backward of the albedo output layer: zo = albbar * alb(1-alb); zc_last = (zo Wo) * relu'(ac);
// dWo += zo^T ac ; dbo += sum zo.   A workgroup owns 32 columns (blockIdx.y) and a slab of rows (blockIdx.x):
// thread = (4 columns, one of 64 row phases); 16-byte accesses, one 128-byte line per row and matrix; an output
// address receives one atomic per row slab (same-address atomics serialise in the L2).  Co <= 4.
__global__ __launch_bounds__(512) void color_out_bwd_kernel(const float* __restrict__ albbar,
                                                            const float* __restrict__ alb,
                                                            const float* __restrict__ ac, int Hcp, int Hc,
                                                            const float* __restrict__ Wo, int ldwo, int Co,
                                                            int squeeze, int64_t M, int rows_per_blk,
                                                            float* __restrict__ zc, float* __restrict__ dWo,
                                                            float* __restrict__ dbo, unsigned* __restrict__ amax) {
  __shared__ float red[64][4][33];
  __shared__ float redb[64][4];
  const int tid = threadIdx.x, cg = tid & 7, ph = tid >> 3;
  const int kl = cg * 4, k0 = blockIdx.y * 32 + kl;
  const bool bias_blk = blockIdx.y == 0;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk;
  const int64_t r1 = min(M, r0 + rows_per_blk);
  float w[4][4], dw[4][4], db[4] = {0.f, 0.f, 0.f, 0.f};
  float zmax = 0.f;   // max |zc| written by this thread (rows < M only: the loop stops at r1)
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      w[c][j] = (c < Co && k0 + j < Hc) ? Wo[c * ldwo + k0 + j] : 0.f;
      dw[c][j] = 0.f;
    }
#pragma unroll 4
  for (int64_t row = r0 + ph; row < r1; row += 64) {
    const vf4 a4 = *reinterpret_cast<const vf4*>(alb + row * 4);
    const vf4 g4 = *reinterpret_cast<const vf4*>(albbar + row * 4);
    float zo[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) zo[c] = c < Co ? g4[c] * (squeeze ? a4[c] * (1.f - a4[c]) : 1.f) : 0.f;
    if (cg == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c) db[c] += zo[c];
    }
    const vf4 av = *reinterpret_cast<const vf4*>(ac + row * Hcp + k0);
    vf4 z;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float t = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) { t = fmaf(zo[c], w[c][j], t); dw[c][j] = fmaf(zo[c], av[j], dw[c][j]); }
      z[j] = (k0 + j < Hc && av[j] > 0.f) ? t : 0.f;
      zmax = fmaxf(zmax, fabsf(z[j]));
    }
    *reinterpret_cast<vf4*>(zc + row * Hcp + k0) = z;
  }
  __shared__ float zm[8];
  if (amax != nullptr) {   // (uniform)  wave maxima meet in LDS behind the barrier below: one atomic per workgroup
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) zmax = fmaxf(zmax, __shfl_xor(zmax, o, 64));
    if ((tid & 63) == 0) zm[tid >> 6] = zmax;
  }
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) red[ph][c][kl + j] = dw[c][j];
  if (cg == 0) {
#pragma unroll
    for (int c = 0; c < 4; ++c) redb[ph][c] = db[c];
  }
  __syncthreads();
  if (amax != nullptr && tid == 511) {
    float m = zm[0];
    for (int w = 1; w < 8; ++w) m = fmaxf(m, zm[w]);
    const unsigned b = __builtin_bit_cast(unsigned, m);
    if (b > __atomic_load_n(amax, __ATOMIC_RELAXED)) atomicMax(amax, b);
  }
  if (tid < 128) {            // (c, column) pairs of this chunk
    const int c = tid >> 5, col = tid & 31;
    float t = 0.f;
    for (int q = 0; q < 64; ++q) t += red[q][c][col];
    if (c < Co && blockIdx.y * 32 + col < Hc) atomicAdd(dWo + c * ldwo + blockIdx.y * 32 + col, t);
  } else if (bias_blk && tid < 128 + Co) {
    const int c = tid - 128;
    float t = 0.f;
    for (int q = 0; q < 64; ++q) t += redb[q][c];
    atomicAdd(dbo + c, t);
  }
}

// nbar_total = nbar + J_pe(n)^T cinb[pe(n) block] ;  geb = J_pe(x) nbar_total  (input of the RA sweep)
// One wave = 64 points.  The [pe(p) | pe(n)] block of cinb (columns blk_off .. Cinp) comes in through an LDS
// tile, the geb rows go out through one; dynamic LDS = 64 * (max(Cinp - blk_off, Ep) + 1) floats.
__global__ __launch_bounds__(64) void nbar_geb_kernel(const float* __restrict__ x4, const float* __restrict__ nrm,
                                                      const float* __restrict__ nbar_in,
                                                      const float* __restrict__ cinb, int Cinp, int blk_off,
                                                      int pen_off, int multires_view, int with_color, int multires,
                                                      int Ep, int64_t M, int64_t Mp, float* __restrict__ geb,
                                                      unsigned* __restrict__ amax) {
  extern __shared__ float tile[];
  const int lane = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * 64, row = r0 + lane;
  float nb[3] = {0.f, 0.f, 0.f};
  if (with_color) {
    const int Wc = Cinp - blk_off;
    tile_load64(cinb, Cinp, r0, blk_off, Wc, tile, lane);
    __builtin_amdgcn_wave_barrier();
    if (row < M) {
      const float* g = tile + lane * (Wc + 1) + (pen_off - blk_off);
#pragma unroll
      for (int d = 0; d < 3; ++d) nb[d] = nbar_in[row * 4 + d] + g[d];
      float f = 1.f;
      int c = 3;
      for (int k = 0; k < multires_view; ++k) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          float s, co;
          sincosf(nrm[row * 4 + d] * f, &s, &co);
          nb[d] += f * (g[c + d] * co - g[c + 3 + d] * s);
        }
        c += 6;
        f *= 2.f;
      }
    }
    __builtin_amdgcn_wave_barrier();   // every lane is done with the input tile before it is overwritten
  } else if (row < M) {
#pragma unroll
    for (int d = 0; d < 3; ++d) nb[d] = nbar_in[row * 4 + d];
  }
  float* o = tile + lane * (Ep + 1);
  o[0] = nb[0]; o[1] = nb[1]; o[2] = nb[2];
  float gm = fmaxf(fmaxf(fabsf(nb[0]), fabsf(nb[1])), fabsf(nb[2]));   // max |geb| of this row (rows >= M carry nb = 0)
  int c = 3;
  float f = 1.f;
  for (int k = 0; k < multires; ++k) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      float s, co;
      sincosf(x4[row * 4 + d] * f, &s, &co);
      const float v0 = f * co * nb[d], v1 = -f * s * nb[d];
      o[c + d] = v0;
      o[c + 3 + d] = v1;
      gm = fmaxf(gm, fmaxf(fabsf(v0), fabsf(v1)));
    }
    c += 6;
    f *= 2.f;
  }
  for (; c < Ep; ++c) o[c] = 0.f;
  if (amax != nullptr) amax_commit(amax, gm, lane);   // the scale of layer 0's weight-gradient job (x2h)
  __builtin_amdgcn_wave_barrier();
  tile_store64(geb, Ep, r0, 0, Ep, tile, lane);
}

// gradient of the sdf-head row: dw_sdf[k] += sum_rows ( sbar/scale * a_last + u_last ), db_sdf += sum sbar/scale
// A workgroup owns 32 columns (blockIdx.y) and a slab of rows (blockIdx.x): thread = (4 columns, one of 64 row
// phases), i.e. every row contributes one 128-byte line per matrix, and an output address only receives one
// atomic per row slab (same-address atomics serialise in the L2: with whole-row workgroups every address took
// one atomic from every workgroup).  fp64 partial sums (long signed sums).
// ulast == nullptr: u_nh arrives as per-tile column sums `ucol` [ntiles][Hp] (fused RA sweep), added by the first row slab.
__global__ __launch_bounds__(512) void sdf_head_bwd_kernel(const float* __restrict__ a, const float* __restrict__ ulast,
                                                           const float* __restrict__ ucol, int ntiles,
                                                           int Hp, int H, const float* __restrict__ sbar,
                                                           float inv_scale, int64_t M, int rows_per_blk,
                                                           float* __restrict__ dwsdf, float* __restrict__ dbsdf,
                                                           float* __restrict__ part_w, float* __restrict__ part_b) {
  // part_w != nullptr: this row slab's sums go to part_w[slab][Hp] / part_b[slab] with plain stores (summed in slab order by
  // dw_reduce_kernel: bit-reproducible) instead of into dwsdf / dbsdf through fp32 atomics
  __shared__ double red[64][33];
  __shared__ double redb[64];
  const int tid = threadIdx.x, cg = tid & 7, ph = tid >> 3;
  const int kl = cg * 4, k0 = blockIdx.y * 32 + kl;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk;
  const int64_t r1 = min(M, r0 + rows_per_blk);
  double s[4] = {0.0, 0.0, 0.0, 0.0}, sb = 0.0;
#pragma unroll 4
  for (int64_t row = r0 + ph; row < r1; row += 64) {
    const float t = sbar[row] * inv_scale;
    if (cg == 0) sb += (double)t;
    const vf4 av = *reinterpret_cast<const vf4*>(a + row * Hp + k0);
    if (ulast != nullptr) {
      const vf4 uv = *reinterpret_cast<const vf4*>(ulast + row * Hp + k0);
#pragma unroll
      for (int j = 0; j < 4; ++j) s[j] += (double)(t * av[j] + uv[j]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) s[j] += (double)(t * av[j]);
    }
  }
  if (ulast == nullptr && blockIdx.x == 0) {
    for (int tile = ph; tile < ntiles; tile += 64) {
      const vf4 uv = *reinterpret_cast<const vf4*>(ucol + (size_t)tile * Hp + k0);
#pragma unroll
      for (int j = 0; j < 4; ++j) s[j] += (double)uv[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) red[ph][kl + j] = s[j];
  if (cg == 0) redb[ph] = sb;
  __syncthreads();
  if (tid < 32) {
    double t = 0.0;
    for (int q = 0; q < 64; ++q) t += red[q][tid];
    const int col = blockIdx.y * 32 + tid;
    if (part_w != nullptr) part_w[(size_t)blockIdx.x * Hp + col] = col < H ? (float)t : 0.f;
    else if (col < H) atomicAdd(dwsdf + col, (float)t);
  } else if (tid == 32 && blockIdx.y == 0) {
    double t = 0.0;
    for (int q = 0; q < 64; ++q) t += redb[q];
    if (part_b != nullptr) part_b[blockIdx.x] = (float)t;
    else atomicAdd(dbsdf, (float)t);
  }
}

// points [first, first + n) of the regular grid of extract_fields (generic path of rnb_sdf_grid)
__global__ void grid_points_kernel(GridGen g, int64_t first, int64_t n, float* __restrict__ pts) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t r = first + i;
  const int iz = (int)(r % g.res);
  r /= g.res;
  const int iy = (int)(r % g.res);
  const int ix = (int)(r / g.res) + g.x_begin;
  pts[i * 3] = linspace_at(g.bmin[0], g.bmax[0], g.res, ix);
  pts[i * 3 + 1] = linspace_at(g.bmin[1], g.bmax[1], g.res, iy);
  pts[i * 3 + 2] = linspace_at(g.bmin[2], g.bmax[2], g.res, iz);
}
__global__ void scale_copy_kernel(const float* __restrict__ src, float scale, int64_t n, float* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i] * scale;
}

// max |.| of a buffer into one slot (float bits, atomicMax; the slot only grows): the maxima of the saved state that the
// per-layer albedo path leaves for the x2h weight-gradient jobs (the fused kernels record theirs on the way)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, int64_t n4, unsigned* __restrict__ slot) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const vf4 v = *reinterpret_cast<const vf4*>(x + 4 * i);
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  amax_commit(slot, m, threadIdx.x & 63);
}

// strided [rows, ld] (first ncols columns) -> dense [M, ncols]
__global__ void copy_cols_kernel(const float* __restrict__ src, int ld, int ncols, int64_t M, float* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * ncols) return;
  int64_t row = i / ncols;
  int col = (int)(i - row * ncols);
  out[i] = src[row * ld + col];
}
// dense [M, ncols] -> strided [Mp, ld] (rows >= M zero-filled)
__global__ void fill_cols_kernel(const float* __restrict__ src, int ncols, int64_t M, int64_t Mp, int ld,
                                 float* __restrict__ dst) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Mp * ncols) return;
  int64_t row = i / ncols;
  int col = (int)(i - row * ncols);
  dst[row * ld + col] = row < M ? src[i] : 0.f;
}

// =====================================================================================================
// GEMM epilogues.  apply4(row, col, v): 4 consecutive columns col..col+3 (col % 4 == 0) of one output row.
// All activation matrices have a padded leading dimension (multiple of 32), so 16-byte accesses are
// aligned and in bounds; columns >= the real width are written as zeros (or the skip-connection payload).
// =====================================================================================================
__device__ inline vf4 ld4(const float* p) { return *reinterpret_cast<const vf4*>(p); }
__device__ inline void st4(float* p, vf4 v) { *reinterpret_cast<vf4*>(p) = v; }

// F hidden layer: a = softplus(acc + b), D = softplus'(acc + b); columns >= N_real: PE override (layer
// feeding the skip layer) or 0
struct EpiF {
  const float* b;
  float* out;
  float* outD;     // nullptr when no derivative is needed (no-grad SDF evaluation)
  int ld;
  int n_real;
  const float* e;  // nullptr unless this layer feeds the skip layer
  int Ep, pe;
  __device__ void apply4(int row, int col, vf4 v) const {
    const vf4 bb = ld4(b + col);
    vf4 a, D;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int cc = col + c;
      float ac, Dc;
      if (cc < n_real) softplus_aD(v[c] + bb[c], ac, Dc);
      else {
        ac = (e != nullptr && cc < n_real + pe) ? e[(size_t)row * Ep + (cc - n_real)] : 0.f;
        Dc = 0.f;
      }
      a[c] = ac;
      D[c] = Dc;
    }
    st4(out + (size_t)row * ld + col, a);
    if (outD) st4(outD + (size_t)row * ld + col, D);
  }
};
// plain linear head (+bias) written to a strided buffer for columns < n_real (n_real % 4 == 0 not assumed)
struct EpiBias {
  const float* b;
  float* out;
  int ld;
  int n_real;
  __device__ void apply4(int row, int col, vf4 v) const {
    if (col + 3 < n_real) {
      st4(out + (size_t)row * ld + col, v + ld4(b + col));
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (col + c < n_real) out[(size_t)row * ld + col + c] = v[c] + b[col + c];
    }
  }
};
struct EpiRelu {
  const float* b;
  float* out;
  int ld;
  int n_real;
  __device__ void apply4(int row, int col, vf4 v) const {
    const vf4 bb = ld4(b + col);
    vf4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = col + c < n_real ? relu_nan(v[c] + bb[c]) : 0.f;
    st4(out + (size_t)row * ld + col, o);
  }
};
// R layer l>=1: g = acc ; skip layer: columns [k_split, k_split+pe) go to ge ; gz_{l-1} = g * D_{l-1}
struct EpiR {
  const float* D_prev;
  float* gz_prev;
  int ld;
  int k_split;   // number of columns that belong to the previous layer's output
  float* ge;     // destination of the skip part (or nullptr)
  int Ep, pe;
  __device__ void apply4(int row, int col, vf4 v) const {
    const size_t o = (size_t)row * ld + col;
    if (col + 3 < k_split) {
      st4(gz_prev + o, v * ld4(D_prev + o));
    } else {
      vf4 g;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int cc = col + c;
        if (cc < k_split) g[c] = v[c] * D_prev[o + c];
        else {
          if (ge != nullptr && cc < k_split + pe) ge[(size_t)row * Ep + (cc - k_split)] = v[c];
          g[c] = 0.f;
        }
      }
      st4(gz_prev + o, g);
    }
  }
};
// R layer 0: ge (+)= acc
struct EpiR0 {
  float* ge;
  int Ep, pe;
  int accumulate;
  __device__ void apply4(int row, int col, vf4 v) const {
    const size_t o = (size_t)row * Ep + col;
    vf4 g = accumulate ? ld4(ge + o) : make_vf4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int c = 0; c < 4; ++c) g[c] = col + c < pe ? g[c] + v[c] : 0.f;
    st4(ge + o, g);
  }
};
// RA layer l: gzb = acc ; zR_l = 100 gzb gz_l (1 - D_l) ; u_{l+1} = gzb D_l  (PE-adjoint override when
// feeding the skip layer)
struct EpiRA {
  const float* D;
  const float* gz;
  float* zR;
  float* u_next;
  int ld;
  int n_real;
  const float* geb;  // nullptr unless this layer feeds the skip layer
  int Ep, pe;
  __device__ void apply4(int row, int col, vf4 v) const {
    const size_t o = (size_t)row * ld + col;
    const vf4 Dv = ld4(D + o), gzv = ld4(gz + o);
    vf4 zr, un;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int cc = col + c;
      if (cc < n_real) {
        zr[c] = 100.f * v[c] * gzv[c] * (1.f - Dv[c]);
        un[c] = v[c] * Dv[c];
      } else {
        zr[c] = 0.f;
        un[c] = (geb != nullptr && cc < n_real + pe) ? geb[(size_t)row * Ep + (cc - n_real)] : 0.f;
      }
    }
    st4(zR + o, zr);
    st4(u_next + o, un);
  }
};
// FB: zb_{l-1} = (acc [+ sbar/scale * w_sdf]) * D_{l-1} + zR_{l-1}
struct EpiFB {
  const float* D_prev;
  const float* zR_prev;
  float* zb_prev;
  int ld;
  int n_real;          // real width of layer l-1's output
  const float* sbar;   // only for the head step
  const float* wsdf;
  float inv_scale;
  __device__ void apply4(int row, int col, vf4 v) const {
    const size_t o = (size_t)row * ld + col;
    const vf4 Dv = ld4(D_prev + o), zr = ld4(zR_prev + o);
    if (sbar != nullptr) {
      const float sb = sbar[row] * inv_scale;
      const vf4 w = ld4(wsdf + col);
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = fmaf(sb, w[c], v[c]);
    }
    vf4 zb;
#pragma unroll
    for (int c = 0; c < 4; ++c) zb[c] = col + c < n_real ? fmaf(v[c], Dv[c], zr[c]) : 0.f;
    st4(zb_prev + o, zb);
  }
};
// albedo backward through a relu layer: zc_{l-1} = acc * (ac_{l-1} > 0)
struct EpiReluMask {
  const float* ac_prev;
  float* out;
  int ld;
  int n_real;
  __device__ void apply4(int row, int col, vf4 v) const {
    const size_t o = (size_t)row * ld + col;
    const vf4 a = ld4(ac_prev + o);
    vf4 r;
#pragma unroll
    for (int c = 0; c < 4; ++c) r[c] = (col + c < n_real && a[c] > 0.f) ? v[c] : 0.f;
    st4(out + o, r);
  }
};
struct EpiStore {
  float* out;
  int ld;
  __device__ void apply4(int row, int col, vf4 v) const { st4(out + (size_t)row * ld + col, v); }
};

// =====================================================================================================
// launch helpers
// =====================================================================================================
// algorithmic FLOPs of one layer-shaped GEMM over M points: real (unpadded) layer shape
static inline double mm_flops(int64_t M, const Lin& ln) { return 2.0 * (double)M * ln.N * ln.K; }

// x3: the product as six bf16 MFMA terms (RNB_VARIANT_X3; k-contiguous weights, N >= 256, K % 16 == 0)
template <bool B_KMAJOR, class Epi>
static int launch_rows(const float* A, int lda, const float* W, int ldw, int64_t Mp, int N, int K, const Epi& epi,
                       double flops, hipStream_t s, bool x3 = false, const x3raw* W3 = nullptr, unsigned* amax = nullptr,
                       int64_t m_real = 0, const char* tag = "layer_gemm") {
  ProfScope prof(flops, s, tag);
  if constexpr (!B_KMAJOR) {
    // W3: this matrix in the split mirror (x3_pack_weights): the weights are then read as ready-made fragments
    if (x3 && W3 != nullptr && N % 32 == 0 && N <= 512 && K % 32 == 0 && Mp % 128 == 0) {
      if (N <= 256) hipLaunchKernelGGL((gemm_rows_x3m_kernel<1, Epi>), dim3((unsigned)(Mp / 128)), dim3(512), 0, s, A, lda, W3, N, K, epi, amax, (long long)m_real);
      else hipLaunchKernelGGL((gemm_rows_x3m_kernel<2, Epi>), dim3((unsigned)(Mp / 128)), dim3(512), 0, s, A, lda, W3, N, K, epi, amax, (long long)m_real);
      RNB_CHECK_LAUNCH();
      return RNB_OK;
    }
    // (only the mirror kernels above leave max |.| of their outputs: a weight-gradient job scaled by a slot nobody wrote
    // would overflow — refuse instead of falling through)
    if (amax != nullptr) RNB_FAIL(RNB_E_INVALID, "layer GEMM %d x %d: the x2h maxima were requested from a kernel that does not record them", N, K);
    if (x3 && N >= 256 && K % XK == 0) {
      dim3 grid((unsigned)(Mp / BM), (unsigned)((N + 255) / 256));
      if (N % 256 == 0) hipLaunchKernelGGL((gemm_rows_x3_kernel<256, false, Epi>), grid, dim3(256), 0, s, A, lda, W, ldw, N, K, epi);
      else hipLaunchKernelGGL((gemm_rows_x3_kernel<256, true, Epi>), grid, dim3(256), 0, s, A, lda, W, ldw, N, K, epi);
      RNB_CHECK_LAUNCH();
      return RNB_OK;
    }
  }
  if (N >= 256) {   // 128 x 256 tiles: the 256-wide layers of the full model run as one wave of 2 blocks / CU
    dim3 grid((unsigned)(Mp / BM), (unsigned)((N + 255) / 256));
    if (N % 256 == 0)
      hipLaunchKernelGGL((gemm_rows_kernel<B_KMAJOR, 256, false, Epi>), grid, dim3(256), 0, s, A, lda, W, ldw, N, K, epi);
    else
      hipLaunchKernelGGL((gemm_rows_kernel<B_KMAJOR, 256, true, Epi>), grid, dim3(256), 0, s, A, lda, W, ldw, N, K, epi);
  } else {
    dim3 grid((unsigned)(Mp / BM), (unsigned)((N + 127) / 128));
    if (N % 128 == 0)
      hipLaunchKernelGGL((gemm_rows_kernel<B_KMAJOR, 128, false, Epi>), grid, dim3(256), 0, s, A, lda, W, ldw, N, K, epi);
    else
      hipLaunchKernelGGL((gemm_rows_kernel<B_KMAJOR, 128, true, Epi>), grid, dim3(256), 0, s, A, lda, W, ldw, N, K, epi);
  }
  RNB_CHECK_LAUNCH();
  return RNB_OK;
}

// Collects the dW jobs of one backward pass and launches them as (at most) three grouped GEMMs, one per kernel
// variant (K-tile 128 exact / K-tile 64 exact / K-tile 64 guarded).  Every job reads buffers that stay untouched until the end of sweep_backward, so deferring the
// launch is safe.
// split-K plan of one dW job: kernel variant v ([0] K % 128 == 0, [1] K % 64 == 0, [2] anything: guarded), number of
// point splits and points per split.  Shared by DwBatch::add and by the sizing of the deterministic partial slabs.
static void dw_plan(int64_t M, int N, int K, int* v_out, int* splits_out, int* rows_out) {
  const bool exact = N % 128 == 0 && M % BK == 0;
  const int v = (exact && K % 128 == 0) ? 0 : (exact && K % 64 == 0) ? 1 : 2;
  const int kt = v == 0 ? 128 : 64;                  // tile width along K of the variant (see kernel)
  const int min_rows = v == 0 ? 1024 : 512;          // points per block (half-size tiles: half the rows)
  const int tiles = ((N + 127) / 128) * ((K + kt - 1) / kt);
  int splits = (int)((M + min_rows - 1) / min_rows);
  const int max_splits = (1024 + tiles - 1) / tiles;  // ~1024 blocks per job
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  if (splits >= 8) splits = splits / 8 * 8;   // multiple of 8: enables the XCD-aware placement in the kernel
  int rows = (int)((M + splits - 1) / splits);
  rows = (rows + BK - 1) / BK * BK;
  // (the kernel tolerates empty splits, so the job keeps the multiple-of-8 split count)
  if ((int64_t)rows * splits < M) splits = (int)((M + rows - 1) / rows);
  *v_out = v;
  *splits_out = splits;
  *rows_out = rows;
}

// Point split of the staged 256 x 256 kernel: one workgroup per (job, split) and ONE round of workgroups (<= 256, one per
// CU), each job's share of them proportional to its work (operand pairs), so every CU multiplies for the whole launch.
// Partial gradients leave through plain stores into slabs that dw_reduce_kernel sums in split order: an fp32 atomic tail
// of 256 KB per workgroup would cost ~50 us per round at the chip's ~1.3 TB/s atomic rate, with nothing to hide under.
static void dw_staged_plan(int64_t M, int npairs, int total_pairs, int* splits_out, int* rows_out) {
  int splits = total_pairs > 0 ? (256 * npairs) / total_pairs : 1;
  if (splits < 1) splits = 1;
  int64_t rows = (M + splits - 1) / splits;
  rows = (rows + kStChunk - 1) / kStChunk * kStChunk;
  if (rows < 2 * kStChunk) rows = 2 * kStChunk;
  splits = (int)((M + rows - 1) / rows);     // every split is non-empty: the reduction reads every slab
  *splits_out = splits;
  *rows_out = (int)rows;
}

// Jobs of the one-workgroup-per-gradient kernels.  The LDS-DMA staged kernel takes 256 x 256 matrices only; the x3
// kernel also takes 256 x K with K a multiple of 64 as COLUMN RANGES of the Y operand: 256-column ranges run as whole
// jobs, what is left as narrow (64-column) jobs — the PE-input layer (K = 64) and the albedo net's first layer
// (K = 320 = 256 + 64) then ride in the same launch instead of a separate fp32-MFMA one.  Work units for the split
// plan: a 256-column pair costs about twice a narrow pair (a quarter of the MFMAs, the same staging of X).
static bool x3_job_shape(bool x3, int N, int K) { return N == 256 && (K == 256 || (x3 && K % 64 == 0 && K >= 64 && K <= 1024)); }
static int x3_job_units(int npairs, int width) { return npairs * (width >= 256 ? 2 : 1); }
template <class F>
static void x3_for_each_range(int K, F f) {   // f(first column, width): 256-wide ranges, then 64-wide ones
  int c = 0;
  for (; c + 256 <= K; c += 256) f(c, 256);
  for (; c + 64 <= K; c += 64) f(c, 64);
}

struct DwBatch {
  DwGroup grp[4];     // [0] K % 128 == 0, [1] K % 64 == 0, [2] anything (guarded), [3] 256 x 256 (LDS-DMA staged)
  double flops[4];
  int64_t M;
  hipStream_t s;
  bool lds_path;      // RNB_VARIANT_DW_LDS: staged-through-LDS kernels (A/B switch)
  bool no_staged = true;    // RNB_VARIANT_DW_STAGED clears it: 256 x 256 jobs through the LDS-DMA staged kernel
  bool x3 = false;          // RNB_VARIANT_X3: 256 x 256 jobs through gemm_dw_x3_kernel (same split plan and slabs)
  bool h2 = false;          // RNB_VARIANT_X2H: ... as three fp16 terms, the adjoint operands scaled by their recorded maxima
  float* part;        // RNB_VARIANT_DETERMINISTIC: bump allocator over the zeroed partial-slab workspace (or nullptr)
  int64_t part_left;
  float* slab;        // slabs of the staged 256 x 256 kernel (always; the tail of the same workspace)
  int64_t slab_left;
  float* const slab_base;         // the slab workspace as handed in: every flushed group starts from it again
  const int64_t slab_floats;
  // reduce-only jobs: slabs that OTHER kernels wrote (per-tile column sums of the fused albedo backward, per-slab sums of the
  // sdf-head row): summed by the reduction launch that follows the last group of weight-gradient jobs, no launch of their own
  DwJob extra[kMaxDwExtra];
  int nextra = 0;
  int add_reduce_only(float* dW, int lddw, float* db, float* part, float* partb, int N, int K, int splits) {
    if (nextra == kMaxDwExtra) RNB_FAIL(RNB_E_INVALID, "too many reduce-only jobs");
    DwJob& j = extra[nextra++];
    memset(&j, 0, sizeof(j));
    j.dW = dW; j.db = db; j.part = part; j.partb = partb;
    j.N = N; j.K = K; j.lddw = lddw; j.splits = splits;
    return RNB_OK;
  }
  DwBatch(int64_t M_, hipStream_t s_, bool lds_path_, float* part_, int64_t part_floats, float* slab_, int64_t slab_floats_)
      : M(M_), s(s_), lds_path(lds_path_), part(part_), part_left(part_floats), slab(slab_), slab_left(slab_floats_),
        slab_base(slab_), slab_floats(slab_floats_) {
    for (int v = 0; v < 4; ++v) { grp[v].njobs = 0; grp[v].M = (int)M_; flops[v] = 0.0; }
  }
  // the staged kernel: every job of the group is split the same way, decided when the group is complete
  int flush_staged(bool final = false) {
    DwGroup& g = grp[3];
    if (g.njobs == 0 && !(final && nextra > 0)) return RNB_OK;
    int total_pairs = 0;   // (work units: x3_job_units)
    for (int q = 0; q < g.njobs; ++q) total_pairs += x3_job_units(g.job[q].npairs, g.job[q].K);
    for (int a = 0, b = g.njobs - 1; a < b; ++a, --b) {   // most recently produced operands first (see flush)
      const DwJob t = g.job[a];
      g.job[a] = g.job[b];
      g.job[b] = t;
    }
    int end = 0;
    for (int q = 0; q < g.njobs; ++q) {
      DwJob& j = g.job[q];
      int splits, rows;
      dw_staged_plan(M, x3_job_units(j.npairs, j.K), total_pairs, &splits, &rows);
      {   // never more slabs than the workspace holds (a group smaller than the one the workspace was sized for)
        const int64_t per_split = (int64_t)j.N * j.K + j.N;
        const int64_t room = slab != nullptr ? slab_left / per_split / (g.njobs - q) : 0;
        if (room < 1) RNB_FAIL(RNB_E_WORKSPACE, "weight-gradient slab workspace exhausted");
        if (splits > room) {
          splits = (int)room;
          int64_t r = (M + splits - 1) / splits;
          r = (r + kStChunk - 1) / kStChunk * kStChunk;
          rows = (int)r;
          splits = (int)((M + rows - 1) / rows);
        }
      }
      j.splits = splits;
      j.rows_per_split = rows;
      end += splits;
      j.block_end = end;
      const int64_t need = (int64_t)splits * j.N * j.K + (int64_t)splits * j.N;
      if (slab == nullptr || need > slab_left) RNB_FAIL(RNB_E_WORKSPACE, "weight-gradient slab workspace exhausted");
      j.part = slab;
      j.partb = slab + (int64_t)splits * j.N * j.K;
      slab += need;
      slab_left -= need;
    }
    if (end > 0) {   // (two scopes: the class time of the weight-gradient kernel is then its own launch duration, as a kernel trace shows it)
      ProfScope prof(flops[3], s, "dW(x3: 256x256 + narrow jobs)");
      if (x3 && h2) hipLaunchKernelGGL((gemm_dw_x3_kernel<0, 2>), dim3((unsigned)end), dim3(512), 0, s, g);
      else if (x3) hipLaunchKernelGGL((gemm_dw_x3_kernel<0, 3>), dim3((unsigned)end), dim3(512), 0, s, g);
      else hipLaunchKernelGGL(gemm_dw_staged_kernel<0>, dim3((unsigned)end), dim3(1024), 0, s, g);
    }
    RNB_CHECK_LAUNCH();
    int nred = g.njobs;
    if (final) {   // the reduce-only jobs ride behind the real ones (no blocks of the kernel above: block_end stays `end`)
      for (int q = 0; q < nextra; ++q) {
        g.job[nred] = extra[q];
        g.job[nred].block_end = end;
        ++nred;
      }
      nextra = 0;
    }
    {
      ProfScope prof(0.0, s, "dW(slab reduce)");
      hipLaunchKernelGGL(dw_reduce_kernel<0>, dim3(256, nred), dim3(256), 0, s, g);
    }
    g.njobs = 0;
    flops[3] = 0.0;
    // the reduction above has read every slab of this group and the next group's kernels follow it on the same
    // stream: the workspace (sized for ONE group of kMaxDwJobs, dw_slab_floats) is free again.  Without this a model
    // with more 256-wide gradient jobs than one group holds ran out of slabs on its second group.
    slab = slab_base;
    slab_left = slab_floats;
    RNB_CHECK_LAUNCH();
    return RNB_OK;
  }
  int flush(int v) {
    DwGroup& g = grp[v];
    if (g.njobs == 0) return RNB_OK;
    // Jobs are added in the order the backward produces their operands (layer nh-1 first); launch them
    // most-recent-first so that the operands written last (zb_0, zb_1, ...) are still in the memory-side
    // cache when their job runs.
    for (int a = 0, b = g.njobs - 1; a < b; ++a, --b) {
      const DwJob t = g.job[a];
      g.job[a] = g.job[b];
      g.job[b] = t;
    }
    {
      int end = 0;   // recompute the prefix sums of the block counts for the new order
      for (int q = 0; q < g.njobs; ++q) {
        DwJob& j = g.job[q];
        const int kt = v == 0 ? 128 : 64;
        const int tiles = ((j.N + 127) / 128) * ((j.K + kt - 1) / kt);
        end += (tiles * j.splits + 7) / 8 * 8;
        j.block_end = end;
      }
    }
    const dim3 grid((unsigned)g.job[g.njobs - 1].block_end);
    {
      ProfScope prof(flops[v], s, "dW(other)");
      if (v == 0 && !lds_path) hipLaunchKernelGGL((gemm_dw_direct_kernel<128, 3>), grid, dim3(256), 0, s, g);
      else if (v == 1 && !lds_path) hipLaunchKernelGGL((gemm_dw_direct_kernel<64, 3>), grid, dim3(256), 0, s, g);
      else if (v == 0) hipLaunchKernelGGL((gemm_dw_kernel<false, 128>), grid, dim3(256), 0, s, g);
      else if (v == 1) hipLaunchKernelGGL((gemm_dw_kernel<false, 64>), grid, dim3(256), 0, s, g);
      else hipLaunchKernelGGL((gemm_dw_kernel<true, 64>), grid, dim3(256), 0, s, g);
      if (part != nullptr) {   // ordered reduction of the partial slabs
        RNB_CHECK_LAUNCH();
        hipLaunchKernelGGL(dw_reduce_kernel<0>, dim3(256, g.njobs), dim3(256), 0, s, g);
      }
    }
    g.njobs = 0;
    flops[v] = 0.0;
    RNB_CHECK_LAUNCH();
    return RNB_OK;
  }
  int add(DwPair p1, DwPair p2, int npairs, int N, int K, float* dW, int lddw, float* db, int bias_pair, double fl) {
    int v, splits, rows;
    dw_plan(M, N, K, &v, &splits, &rows);
    if (x3_job_shape(x3, N, K) && M % kStChunk == 0 && !lds_path && !no_staged) {   // -> the one-workgroup-per-gradient kernel
      int rc = RNB_OK;
      x3_for_each_range(K, [&](int c0, int width) {
        if (rc != RNB_OK) return;
        if (grp[3].njobs == kMaxDwJobs) rc = flush_staged();
        if (rc != RNB_OK) return;
        DwJob& j = grp[3].job[grp[3].njobs++];
        j.p1 = p1; j.p2 = p2;
        j.p1.Y += c0; j.p2.Y += c0;              // column range of the Y operands (their leading dimension stays)
        j.dW = dW + c0;
        j.db = c0 == 0 ? db : nullptr;            // the bias sums (columns of X) belong to the first range
        j.part = nullptr; j.partb = nullptr;
        j.npairs = npairs; j.N = N; j.K = width; j.lddw = lddw; j.bias_pair = bias_pair;
        j.splits = 0; j.rows_per_split = 0; j.block_end = 0;
      });
      RNB_TRY(rc);
      flops[3] += fl;
      return RNB_OK;
    }
    if (grp[v].njobs == kMaxDwJobs) RNB_TRY(flush(v));
    const int kt = v == 0 ? 128 : 64;
    const int tiles = ((N + 127) / 128) * ((K + kt - 1) / kt);
    DwGroup& g = grp[v];
    DwJob& j = g.job[g.njobs];
    j.p1 = p1; j.p2 = p2; j.dW = dW; j.db = db;
    j.part = nullptr;
    j.partb = nullptr;
    if (part != nullptr) {
      const int64_t need = (int64_t)splits * N * lddw + (int64_t)splits * N;
      if (need > part_left) RNB_FAIL(RNB_E_WORKSPACE, "deterministic dW: partial-slab workspace exhausted");
      j.part = part;
      j.partb = part + (int64_t)splits * N * lddw;
      part += need;
      part_left -= need;
    }
    j.npairs = npairs; j.N = N; j.K = K; j.lddw = lddw; j.bias_pair = bias_pair;
    j.splits = splits; j.rows_per_split = rows;
    // jobs start on a multiple of 8 blocks so that (block & 7) is the XCD inside every job
    const int begin = g.njobs ? g.job[g.njobs - 1].block_end : 0;
    j.block_end = begin + (tiles * splits + 7) / 8 * 8;
    ++g.njobs;
    flops[v] += fl;
    return RNB_OK;
  }
  int flush_all() {
    RNB_TRY(flush(1));   // holds the first layer's job: its operands are the most recent
    RNB_TRY(flush_staged(true));
    RNB_TRY(flush(0));
    return flush(2);
  }
};

// floats of partial-slab workspace the deterministic variant needs for one backward over M points: the same job list
// as sweep_backward
// floats of slab workspace of the staged kernel for one backward over M points (the 256 x 256 jobs of sweep_backward)
int64_t dw_staged_floats(const Layout& L, int64_t M, bool with_color) {
  const bool x3 = is_x3(L);
  int total_units = 0;
  auto units = [&](const Lin& ln, int npairs) {
    if (!x3_job_shape(x3, ln.Np, ln.Kp)) return;
    x3_for_each_range(ln.Kp, [&](int, int width) { total_units += x3_job_units(npairs, width); });
  };
  for (int l = 0; l < L.nh; ++l) units(L.hid[l], 2);
  if (with_color) {
    units(L.feat, 1);
    for (int l = 0; l < L.nc; ++l) units(L.col[l], 1);
  }
  int64_t total = 0;
  auto job = [&](const Lin& ln, int npairs) {
    if (!x3_job_shape(x3, ln.Np, ln.Kp)) return;
    x3_for_each_range(ln.Kp, [&](int, int width) {
      int splits, rows;
      dw_staged_plan(M, x3_job_units(npairs, width), total_units, &splits, &rows);
      total += (int64_t)splits * 256 * width + (int64_t)splits * 256;
    });
  };
  for (int l = 0; l < L.nh; ++l) job(L.hid[l], 2);
  if (with_color) {
    job(L.feat, 1);
    for (int l = 0; l < L.nc; ++l) job(L.col[l], 1);
  }
  return total;
}

// floats of ordered-reduction workspace of the atomic kernels (RNB_VARIANT_DETERMINISTIC), same job list as sweep_backward
int64_t dw_partial_floats(const Layout& L, int64_t M, bool with_color) {
  int64_t total = 0;
  // (jobs that DwBatch::add hands to the one-workgroup-per-gradient kernels leave through that kernel's own slabs, whatever
  // the variant: no ordered-reduction slabs — and no 200 MB memset per step — for them)
  const bool staged_path = (is_x3(L) || (L.variant & RNB_VARIANT_DW_STAGED) != 0) && !(L.variant & RNB_VARIANT_DW_LDS);
  auto job = [&](int N, int K) {
    if (staged_path && x3_job_shape(is_x3(L), N, K) && M % kStChunk == 0) return;
    int v, splits, rows;
    dw_plan(M, N, K, &v, &splits, &rows);
    total += (int64_t)splits * N * K + (int64_t)splits * N;
  };
  if (with_color) {
    for (int l = 0; l < L.nc; ++l) job(L.col[l].Np, L.col[l].Kp);
    job(L.feat.Np, L.feat.Kp);
  }
  for (int l = 0; l < L.nh; ++l) job(L.hid[l].Np, L.hid[l].Kp);
  return total;
}

// the split mirror of the matrix at float offset `off` of the packed buffer (x3_pack_weights), or nullptr
static inline const x3raw* x3_mirror(const Layout& L, const float* packed, int64_t off) {
  if (!is_x3(L) || off < 0) return nullptr;
  return reinterpret_cast<const x3raw*>(packed + L.total) + 3 * off;
}

static inline unsigned blocks_for(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

int launch_grid_points(const GridGen& g, int64_t first, int64_t n, float* pts, hipStream_t s) {
  hipLaunchKernelGGL(grid_points_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, g, first, n, pts);
  RNB_CHECK_LAUNCH();
  return RNB_OK;
}
int launch_scale_copy(const float* src, float scale, int64_t n, float* dst, hipStream_t s) {
  hipLaunchKernelGGL(scale_copy_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, src, scale, n, dst);
  RNB_CHECK_LAUNCH();
  return RNB_OK;
}
int launch_copy_cols(const float* src, int ld, int ncols, int64_t M, float* out, hipStream_t s) {
  hipLaunchKernelGGL(copy_cols_kernel, dim3(blocks_for(M * ncols, 256)), dim3(256), 0, s, src, ld, ncols, M, out);
  RNB_CHECK_LAUNCH();
  return RNB_OK;
}
int launch_fill_cols(const float* src, int ncols, int64_t M, int64_t Mp, int ld, float* dst, hipStream_t s) {
  hipLaunchKernelGGL(fill_cols_kernel, dim3(blocks_for(Mp * ncols, 256)), dim3(256), 0, s, src, ncols, M, Mp, ld, dst);
  RNB_CHECK_LAUNCH();
  return RNB_OK;
}

static int launch_absmax(const float* x, int64_t n, unsigned* slot, hipStream_t s) {
  hipLaunchKernelGGL(absmax_kernel, dim3(1024), dim3(256), 0, s, x, n / 4, slot);
  RNB_CHECK_LAUNCH();
  return RNB_OK;
}

// out[k] = max over a list of word slots (float bits) — the finishing step of rnb_render_range
__global__ void range_fold_kernel(const unsigned* __restrict__ words, int n, int k, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  unsigned m = 0u;
  for (int i = 0; i < n; ++i) m = max(m, words[i]);
  out[k] = __builtin_bit_cast(float, m);
}

// rnb_render_range: out[0..4] as documented in include/rnbneus.h.  The maxima of the saved state are taken by a pass over the
// buffers themselves (the hot path records them only for tiles beyond the fixed scales' range); out doubles as scratch.
int launch_range_report(const Layout& L, const float* packed, const PointBufs& pb, bool with_color, bool with_backward, float* out,
                        hipStream_t s) {
  RNB_CHECK_HIP(hipMemsetAsync(out, 0, 8 * sizeof(float), s));
  unsigned* w = reinterpret_cast<unsigned*>(out);
  const H2Tab* tab = h2_tab(L, packed);
  if (tab != nullptr) {
    hipLaunchKernelGGL(range_fold_kernel, dim3(1), dim3(64), 0, s, tab->wmax, L.nh + 1 + L.nc, 0, out);
    RNB_CHECK_LAUNCH();
  }
  const int64_t n = pb.Mp * L.Hp;
  const bool bf = is_bf16(L);   // (bf16 state is not fp32: only the fp32 buffers are scanned)
  if (!bf) {
    RNB_TRY(launch_absmax(pb.e, pb.Mp * L.Ep, w + 1, s));
    for (int l = 0; l < L.nh; ++l) RNB_TRY(launch_absmax(pb.a[l], n, w + 1, s));
    if (pb.gz[0] != nullptr)
      for (int l = 0; l < L.nh; ++l) RNB_TRY(launch_absmax(pb.gz[l], n, w + 2, s));
  }
  if (with_color && pb.cin != nullptr) {
    RNB_TRY(launch_absmax(pb.cin, pb.Mp * L.Cinp, w + 3, s));
    for (int l = 0; l < L.nc; ++l) RNB_TRY(launch_absmax(pb.ac[l], pb.Mp * L.Hcp, w + 3, s));
  }
  if (with_backward && pb.amax != nullptr && is_x2h(L)) {
    hipLaunchKernelGGL(range_fold_kernel, dim3(1), dim3(64), 0, s, pb.amax, (int)AMAX_SLOTS, 4, out);
    RNB_CHECK_LAUNCH();
  }
  return RNB_OK;
}

int launch_pe_points(const Layout& L, const float* pts, int64_t M, PointBufs& pb, hipStream_t s) {
  hipLaunchKernelGGL(pe_points_kernel, dim3(blocks_for(pb.Mp, 256)), dim3(256), 0, s, pts, M, pb.Mp, L.sdf_scale,
                     L.multires, L.Ep, pb.x, pb.e);
  RNB_CHECK_LAUNCH();
  return RNB_OK;
}

// F: forward sweep (models/fields.py:82-104).  Needs pb.e; fills pb.a[*], pb.sdf, optionally the feature
// block of pb.cin (need_feat) and the reverse-sweep seed pb.gz[nh-1] (need_gz_last).
int sweep_forward(const Layout& L, const float* packed, PointBufs& pb, bool need_feat, bool need_gz_last,
                  float* feat_dense, hipStream_t s) {
  for (int l = 0; l < L.nh; ++l) {
    const Lin& ln = L.hid[l];
    const float* in = l == 0 ? pb.e : pb.a[l - 1];
    const int lda = l == 0 ? L.Ep : L.Hp;
    EpiF epi{packed + ln.b_off, pb.a[l], pb.D[l], L.Hp, ln.N, (l + 1 == L.skip) ? pb.e : nullptr, L.Ep, L.pe};
    RNB_TRY((launch_rows<false, EpiF>(in, lda, packed + ln.w_off, ln.Kp, pb.Mp, ln.Np, ln.Kp, epi, mm_flops(pb.M, ln), s)));
  }
  hipLaunchKernelGGL(sdf_head_kernel, dim3(blocks_for(pb.Mp * 32, 256)), dim3(256), 0, s, pb.a[L.nh - 1],
                     pb.D[L.nh - 1], L.Hp, L.H, packed + L.wsdf_off, packed + L.bsdf_off, 1.f / L.sdf_scale, pb.Mp, pb.sdf,
                     need_gz_last ? pb.gz[L.nh - 1] : nullptr);
  RNB_CHECK_LAUNCH();
  if (need_feat) {
    EpiBias epi{packed + L.feat.b_off, pb.cin, L.Cinp, L.F};
    RNB_TRY((launch_rows<false, EpiBias>(pb.a[L.nh - 1], L.Hp, packed + L.feat.w_off, L.feat.Kp, pb.Mp, L.feat.Np,
                                         L.feat.Kp, epi, mm_flops(pb.M, L.feat), s)));
    if (feat_dense) {
      RNB_TRY(launch_copy_cols(pb.cin, L.Cinp, L.F, pb.M, feat_dense, s));
    }
  }
  return RNB_OK;
}

// R: reverse sweep for the normal (models/fields.py:114-127 without autograd).  Needs pb.a[*], pb.gz[nh-1].
int sweep_reverse(const Layout& L, const float* packed, PointBufs& pb, hipStream_t s) {
  for (int l = L.nh - 1; l >= 1; --l) {
    const Lin& ln = L.hid[l];
    const bool is_skip = (l == L.skip);
    EpiR epi{pb.D[l - 1], pb.gz[l - 1], L.Hp, is_skip ? ln.K - L.pe : ln.K, is_skip ? pb.ge : nullptr, L.Ep, L.pe};
    RNB_TRY((launch_rows<true, EpiR>(pb.gz[l], L.Hp, packed + ln.w_off, ln.Kp, pb.Mp, ln.Kp, ln.Np, epi, mm_flops(pb.M, ln), s)));
  }
  {
    const Lin& ln = L.hid[0];
    EpiR0 epi{pb.ge, L.Ep, L.pe, L.skip >= 1 ? 1 : 0};
    RNB_TRY((launch_rows<true, EpiR0>(pb.gz[0], L.Hp, packed + ln.w_off, ln.Kp, pb.Mp, ln.Kp, ln.Np, epi, mm_flops(pb.M, ln), s)));
  }
  hipLaunchKernelGGL(normal_kernel, dim3(blocks_for(pb.Mp, 256)), dim3(256), 0, s, pb.x, pb.ge, L.Ep, L.multires,
                     pb.Mp, pb.nrm);
  RNB_CHECK_LAUNCH();
  return RNB_OK;
}

// C: albedo network (models/fields.py:177-215, mode no_view_dir).  Needs the feature block of pb.cin and pb.nrm.
int sweep_color(const Layout& L, const float* packed, PointBufs& pb, const float* pts, const float* nrm, int nrm_ld,
                hipStream_t s) {
  hipLaunchKernelGGL(color_input_kernel, dim3(blocks_for(pb.Mp, 64)), dim3(64),
                     (size_t)64 * (L.Cinp - L.F + 1) * sizeof(float), s, pts, nrm, nrm_ld, pb.M, pb.Mp,
                     L.F, L.multires_view, L.Cinp, pb.cin);
  RNB_CHECK_LAUNCH();
  for (int l = 0; l < L.nc; ++l) {
    const Lin& ln = L.col[l];
    const float* in = l == 0 ? pb.cin : pb.ac[l - 1];
    const int lda = l == 0 ? L.Cinp : L.Hcp;
    EpiRelu epi{packed + ln.b_off, pb.ac[l], L.Hcp, ln.N};
    // (per-layer path of an albedo net the fused kernels do not cover: six bf16 terms — no operand range to look after)
    RNB_TRY((launch_rows<false, EpiRelu>(in, lda, packed + ln.w_off, ln.Kp, pb.Mp, ln.Np, ln.Kp, epi, mm_flops(pb.M, ln), s, is_x3(L),
                                         x3_mirror(L, packed, ln.w_off), nullptr, 0, "layer_gemm(forward)")));
    // x2h weight gradients take this layer's input / output as a state operand: its maximum (PointBufs::smax)
    if (is_x2h(L) && pb.smax != nullptr) {
      if (l == 0) RNB_TRY(launch_absmax(pb.cin, pb.Mp * L.Cinp, pb.smax + SMAX_CIN, s));
      RNB_TRY(launch_absmax(pb.ac[l], pb.Mp * L.Hcp, pb.smax + SMAX_AC + l, s));
    }
  }
  hipLaunchKernelGGL(color_out_kernel, dim3(blocks_for(pb.Mp * 16, 256)), dim3(256), 0, s, pb.ac[L.nc - 1], L.Hcp,
                     L.Hc, packed + L.colo.w_off, L.colo.Kp, packed + L.colo.b_off, L.Co, L.squeeze, pb.Mp, pb.alb);
  RNB_CHECK_LAUNCH();
  return RNB_OK;
}

// Backward of everything above given pb.sbar, pb.nbar, pb.albbar (from the composite backward).
// packed_grad (same layout as `packed`) must be zero on entry; it receives dW_eff / db of every layer.
int sweep_backward(const Layout& L, const float* packed, PointBufs& pb, bool with_color, float* packed_grad,
                   bool fused, hipStream_t s) {
  const int64_t M = pb.M, Mp = pb.Mp;
  const bool det = (L.variant & RNB_VARIANT_DETERMINISTIC) != 0;
  if (pb.dw_part == nullptr) RNB_FAIL(RNB_E_WORKSPACE, "no weight-gradient slab workspace was carved");
  // workspace = [ordered-reduction slabs of the atomic kernels (deterministic variant only) | slabs of the staged kernel]
  const int64_t staged_floats = dw_staged_floats(L, M, with_color);
  const int64_t det_floats = pb.dw_part_floats - staged_floats;
  if (det && det_floats > 0) RNB_CHECK_HIP(hipMemsetAsync(pb.dw_part, 0, (size_t)det_floats * sizeof(float), s));
  DwBatch dw(M, s, (L.variant & RNB_VARIANT_DW_LDS) != 0, det ? pb.dw_part : nullptr, det ? det_floats : 0,
             pb.dw_part + det_floats, staged_floats);
  dw.x3 = is_x3(L);
  dw.no_staged = (L.variant & RNB_VARIANT_DW_STAGED) == 0 && !dw.x3;
  // x2h weight gradients: every adjoint tensor's producer (fused sweeps, albedo backward) records its maximum
  const bool h2 = is_x2h(L) && fused && !is_bf16(L) && pb.amax != nullptr;
  dw.h2 = h2;
  // (pb.amax was zeroed by the composite backward, the first kernel of rnb_render_bwd)
  const bool color_bf16 = is_bf16(L) && with_color && bf16_color_supported(L) && pb.cin8 != nullptr;
  // the albedo network's backward as ONE fused sweep (color_h2.hip), which also forms geb = J_pe(x) nbar_total
  const bool color_h2 = with_color && h2 && color_h2_supported(L) && pb.col_part != nullptr;
  // ---- C': albedo network backward ---------------------------------------------------------------
  if (color_bf16) {
    RNB_TRY(bf16_color_backward(L, packed, pb, packed_grad, s));
  } else if (color_h2) {
    RNB_TRY(color_h2_backward(L, packed, pb, s));
    for (int l = L.nc - 1; l >= 0; --l) {
      const Lin& ln = L.col[l];
      const float* in = l == 0 ? pb.cin : pb.ac[l - 1];
      const int ldin = l == 0 ? L.Cinp : L.Hcp;
      DwPair p{pb.zc[l], L.Hcp, in, ldin, 0, pb.amax + AMAX_ZC + l, pb.smax + (l == 0 ? SMAX_CIN : SMAX_AC + l - 1)};
      RNB_TRY(dw.add(p, p, 1, ln.Np, ln.Kp, packed_grad + ln.w_off, ln.Kp, packed_grad + ln.b_off, 0, mm_flops(M, ln)));
    }
    // the output layer's gradient: per-tile column sums, summed in tile order by the reduction launch of the weight gradients
    const int64_t tiles = Mp / 64;
    RNB_TRY(dw.add_reduce_only(packed_grad + L.colo.w_off, L.colo.Kp, packed_grad + L.colo.b_off, pb.col_part,
                               pb.col_part + tiles * L.Co * L.Hcp, L.Co, L.Hcp, (int)tiles));
  } else if (with_color) {
    const int chunks = L.Hcp / 32;   // 32-column chunks x row slabs, ~256 workgroups, slabs a multiple of 64 rows
    int64_t slabs = det ? 1 : (256 + chunks - 1) / chunks;   // deterministic: ONE slab, i.e. one add per address onto zero
    int rows_per_blk = (int)((M + slabs - 1) / slabs);
    rows_per_blk = (rows_per_blk + 63) / 64 * 64;
    hipLaunchKernelGGL(color_out_bwd_kernel, dim3(blocks_for(M, rows_per_blk), chunks), dim3(512), 0, s, pb.albbar, pb.alb,
                       pb.ac[L.nc - 1], L.Hcp, L.Hc, packed + L.colo.w_off, L.colo.Kp, L.Co, L.squeeze, M,
                       rows_per_blk, pb.zc[L.nc - 1], packed_grad + L.colo.w_off, packed_grad + L.colo.b_off,
                       h2 ? pb.amax + AMAX_ZC + (L.nc - 1) : (unsigned*)nullptr);
    RNB_CHECK_LAUNCH();
    for (int l = L.nc - 1; l >= 0; --l) {
      const Lin& ln = L.col[l];
      const float* in = l == 0 ? pb.cin : pb.ac[l - 1];
      const int ldin = l == 0 ? L.Cinp : L.Hcp;
      DwPair p{pb.zc[l], L.Hcp, in, ldin, 0, h2 ? pb.amax + AMAX_ZC + l : nullptr,
               h2 ? pb.smax + (l == 0 ? SMAX_CIN : SMAX_AC + l - 1) : nullptr};
      RNB_TRY(dw.add(p, p, 1, ln.Np, ln.Kp, packed_grad + ln.w_off, ln.Kp, packed_grad + ln.b_off, 0, mm_flops(M, ln)));
      if (l > 0) {
        EpiReluMask epi{pb.ac[l - 1], pb.zc[l - 1], L.Hcp, L.col[l - 1].N};
        // zc_{l-1} = (zc_l W_l) * relu': k-contiguous product against the transposed copy W_l^T [Kp x Np]
        // (per-layer path: six bf16 terms; the kernel leaves max |acc| for the x2h weight-gradient job of zc_{l-1})
        RNB_TRY((launch_rows<false, EpiReluMask>(pb.zc[l], L.Hcp, packed + ln.wT_off, ln.Np, Mp, ln.Kp, ln.Np, epi, mm_flops(M, ln), s, is_x3(L),
                                                 x3_mirror(L, packed, ln.wT_off), h2 ? pb.amax + AMAX_ZC + (l - 1) : nullptr, M)));
      } else {
        EpiStore epi{pb.cinb, L.Cinp};
        RNB_TRY((launch_rows<false, EpiStore>(pb.zc[0], L.Hcp, packed + ln.wT_off, ln.Np, Mp, ln.Kp, ln.Np, epi, mm_flops(M, ln), s, is_x3(L),
                                              x3_mirror(L, packed, ln.wT_off), h2 ? pb.amax + AMAX_CINB : nullptr, M)));
      }
    }
  }
  // ---- nbar (+ albedo-net contribution) -> geb = u_0 -----------------------------------------------
  if (!color_h2) {
    const int wt = (with_color && L.Cinp - L.F > L.Ep) ? L.Cinp - L.F : L.Ep;
    hipLaunchKernelGGL(nbar_geb_kernel, dim3(blocks_for(Mp, 64)), dim3(64), (size_t)64 * (wt + 1) * sizeof(float), s,
                       pb.x, pb.nrm, pb.nbar, pb.cinb, L.Cinp, L.F, L.F + L.pev, L.multires_view, with_color ? 1 : 0,
                       L.multires, L.Ep, M, Mp, pb.geb, (unsigned*)nullptr);   // (max |geb|: recorded by the RA sweep as it loads the tile)
  }
  RNB_CHECK_LAUNCH();
  if (is_bf16(L)) {
    // RNB_VARIANT_BF16: RA, the sdf-head row, FB and every weight gradient of the SDF network (+ feature head) run as
    // bf16 sweeps on the bf16 saved state; the albedo net's own (fp32) weight-gradient jobs were queued above
    RNB_TRY(dw.flush_all());
    return bf16_backward(L, packed, pb, with_color, color_bf16, packed_grad, s);
  }
  // ---- RA: adjoint of the reverse sweep, forward layer order -----------------------------------------
  int u_tiles = 0;
  if (fused) {
    RNB_TRY(fused_ra(L, packed, pb, s, &u_tiles));
  } else {
    for (int l = 0; l < L.nh; ++l) {
      const Lin& ln = L.hid[l];
      const float* in = l == 0 ? pb.geb : pb.u[l];
      const int lda = l == 0 ? L.Ep : L.Hp;
      EpiRA epi{pb.D[l], pb.gz[l], pb.zR[l], pb.u[l + 1], L.Hp, ln.N, (l + 1 == L.skip) ? pb.geb : nullptr, L.Ep, L.pe};
      RNB_TRY((launch_rows<false, EpiRA>(in, lda, packed + ln.w_off, ln.Kp, Mp, ln.Np, ln.Kp, epi, mm_flops(M, ln), s)));
    }
  }
  // ---- sdf-head row gradient ---------------------------------------------------------------------
  {
    // Hp / 32 column chunks x row slabs, ~256 workgroups in total, slabs a multiple of the 64 row phases
    const int chunks = L.Hp / 32;
    int64_t slabs = det ? 1 : (256 + chunks - 1) / chunks;
    int rows_per_blk = (int)((M + slabs - 1) / slabs);
    rows_per_blk = (rows_per_blk + 63) / 64 * 64;
    // with the one-workgroup-per-gradient kernels (x3 / staged) there is a slab reduction at the end of the backward: the row
    // slabs' sums ride in it (no atomics: bit-reproducible); otherwise fp32 atomics (one slab in the deterministic variant)
    const bool slab_out = !dw.no_staged && !dw.lds_path && pb.sdfh_part != nullptr && M % kStChunk == 0;
    if (slab_out) {
      slabs = kSdfHeadSlabs;
      rows_per_blk = (int)((M + slabs - 1) / slabs);
      rows_per_blk = (rows_per_blk + 63) / 64 * 64;
    }
    const unsigned nslab = blocks_for(M, rows_per_blk);
    float* part_w = slab_out ? pb.sdfh_part : nullptr;
    float* part_b = slab_out ? pb.sdfh_part + (size_t)nslab * L.Hp : nullptr;
    hipLaunchKernelGGL(sdf_head_bwd_kernel, dim3(nslab, chunks), dim3(512), 0, s, pb.a[L.nh - 1],
                       fused ? (const float*)nullptr : (const float*)pb.u[L.nh], (const float*)pb.u[L.nh], u_tiles, L.Hp,
                       L.H, pb.sbar, 1.f / L.sdf_scale, M, rows_per_blk,
                       packed_grad + L.wsdf_off, packed_grad + L.bsdf_off, part_w, part_b);
    RNB_CHECK_LAUNCH();
    if (slab_out)
      RNB_TRY(dw.add_reduce_only(packed_grad + L.wsdf_off, L.Hp, packed_grad + L.bsdf_off, part_w, part_b, 1, L.Hp, (int)nslab));
  }
  // ---- FB head: zb_{nh-1} = (fbar Wf + sbar/scale w_sdf) * D + zR ----------------------------------
  if (fused) RNB_TRY(fused_fb(L, packed, pb, with_color, s));   // all zb_l in one launch
  {
    if (!fused) {
      EpiFB epi{pb.D[L.nh - 1], pb.zR[L.nh - 1], pb.zb[L.nh - 1], L.Hp, L.hid[L.nh - 1].N, pb.sbar,
                packed + L.wsdf_off, 1.f / L.sdf_scale};
      const int K = with_color ? L.feat.Np : 0;   // no_albedo: fbar == 0, the GEMM degenerates to its epilogue
      RNB_TRY((launch_rows<true, EpiFB>(pb.cinb, L.Cinp, packed + L.feat.w_off, L.feat.Kp, Mp, L.feat.Kp, K, epi,
                                        with_color ? mm_flops(M, L.feat) : 0.0, s)));
    }
    if (with_color) {
      DwPair p{pb.cinb, L.Cinp, pb.a[L.nh - 1], L.Hp, 0, h2 ? pb.amax + AMAX_CINB : nullptr, h2 ? pb.smax + SMAX_A + L.nh - 1 : nullptr};
      RNB_TRY(dw.add(p, p, 1, L.feat.Np, L.feat.Kp, packed_grad + L.feat.w_off, L.feat.Kp,
                     packed_grad + L.feat.b_off, 0, mm_flops(M, L.feat)));
    }
  }
  // ---- FB + dW, layers nh-1 .. 0 -------------------------------------------------------------------
  for (int l = L.nh - 1; l >= 0; --l) {
    const Lin& ln = L.hid[l];
    const float* in = l == 0 ? pb.e : pb.a[l - 1];
    const int ldin = l == 0 ? L.Ep : L.Hp;
    const float* uin = l == 0 ? pb.geb : pb.u[l];
    // (x2h: adjoint operand, its recorded maximum; the state operand's recorded maximum)
    DwPair p1{pb.gz[l], L.Hp, uin, ldin, 1, h2 ? pb.amax + AMAX_U + l : nullptr, h2 ? pb.smax + SMAX_GZ + l : nullptr};
    DwPair p2{pb.zb[l], L.Hp, in, ldin, 0, h2 ? pb.amax + AMAX_ZB + l : nullptr,
              h2 ? pb.smax + (l == 0 ? SMAX_E : SMAX_A + l - 1) : nullptr};
    RNB_TRY(dw.add(p1, p2, 2, ln.Np, ln.Kp, packed_grad + ln.w_off, ln.Kp, packed_grad + ln.b_off, 1,
                   2.0 * mm_flops(M, ln)));
    if (l > 0 && !fused) {
      const Lin& lp = L.hid[l - 1];
      EpiFB epi{pb.D[l - 1], pb.zR[l - 1], pb.zb[l - 1], L.Hp, lp.N, nullptr, nullptr, 1.f};
      RNB_TRY((launch_rows<true, EpiFB>(pb.zb[l], L.Hp, packed + ln.w_off, ln.Kp, Mp, ln.Kp, ln.Np, epi, mm_flops(M, ln), s)));
    }
  }
  return dw.flush_all();
}

}  // namespace rnb
