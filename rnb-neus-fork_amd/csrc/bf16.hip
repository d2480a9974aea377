// RNB_VARIANT_BF16 — BASELINE config 5: the sweeps of the 256-wide SDF network with bf16 operands on
// v_mfma_f32_32x32x16_bf16 and fp32 accumulators.  Same mathematics as fused.hip / fused_bwd.hip (oracle/explicit.py is
// the statement); what changes is the arithmetic of the matrix products and the format of the per-point saved state.
//
//   bf_forward_kernel   positional encoding + F sweep (+ sdf head, + feature head)   models/fields.py:82-104
//   bf_reverse_kernel   R : reverse-mode normal                                      models/fields.py:114-127
//   bf_ra_kernel        RA: adjoint of R
//   bf_fb_kernel        FB: backward of F
//   bf_dw_kernel        dW_l = gz_l^T u_l + zb_l^T in_l for all layers (grouped launch, split over points)
//   bf_sdf_head_bwd_kernel  gradient of the sdf-head row
//
// At 1/16 of the fp32 matrix time these sweeps are bound by the HBM traffic of the saved state, not by the matrix
// cores.  Saved state is therefore bf16, in ONE layout that serves every consumer without a transpose:
// "K8" = [points / 8][columns][8 points].  (1) An accumulator tile of v_mfma_f32_32x32x16_bf16 holds, per lane, one
// column and rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5): registers 4g .. 4g+3 are four consecutive points of one
// column = 8 contiguous bytes of a K8 matrix, and the 64 lanes of one store instruction cover 512 contiguous bytes.
// (2) The weight-gradient product sums over points: its MFMA operands are "8 consecutive points of one column" for
// both X^T and Y — exactly one 16-byte K8 unit per lane, coalesced, no LDS, no transposed reads.
// Activations inside a sweep stay in LDS as row-major bf16 [point][256] (pitch 264: conflict-free ds_read_b128 of the
// A fragments); weights stream from L2 as bf16 rows of the mirror that rnb_weightnorm_fwd appends to the packed buffer.
// fp32 master weights, fp32 gradients (split-K partial sums leave through fp32 atomics or ordered slabs), fp32
// epilogue math; only what enters an MFMA or goes to HBM per point is rounded to bf16 (round-to-nearest-even).
#include "fused_common.hip.h"

namespace rnb {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef unsigned vu4 __attribute__((ext_vector_type(4)));
typedef unsigned short bfraw;   // storage type of one bf16 (no arithmetic on it)

constexpr int BP = 264;    // LDS pitch (bf16 elements) of an activation row: 528 bytes
constexpr int BT = 64;     // points per workgroup
// Every sweep kernel is templated on TI = 32-row MFMA tiles per wave.  TI = 2: 4 waves per workgroup, each all 64 rows x
// 64 columns.  TI = 1: 8 waves, wave = (row half, column group), 32 rows x 64 columns each: half the accumulator and
// prefetch registers (<= 128), so two workgroups per CU are 4 waves per SIMD instead of 2.  (Kept as an A/B variant:
// although the TI = 2 sweeps are parked 46-69 % of the time (SQ_WAIT_ANY), TI = 1 is slower — see bf_ti below.)
template <int TI> struct BfCfg { static constexpr int NW = 8 / TI; static constexpr int NT = 64 * NW; };

__device__ inline unsigned pack2(float a, float b) {
  bf2 p = {(__bf16)a, (__bf16)b};   // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
  return __builtin_bit_cast(unsigned, p);
}
// A value the optimiser cannot see through: address arithmetic derived from it is redone where it is used instead of
// being hoisted out of the layer loop (loop-invariant per-lane offsets of ~40 loads and stores, kept live across the
// matrix loop, were what spilled in the sweeps with two epilogue operand tiles).
__device__ inline int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
__device__ inline bfraw to_bf(float a) { return __builtin_bit_cast(bfraw, (__bf16)a); }
__device__ inline float bf_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ inline float bf_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
__device__ inline float bf_f(bfraw v) { return __builtin_bit_cast(float, (unsigned)v << 16); }

// element offset of (row, col) in a K8 matrix with C columns
__device__ inline size_t k8(int64_t row, int col, int C) { return ((size_t)(row >> 3) * C + col) * 8 + (row & 7); }

// ---- matrix loop ------------------------------------------------------------------------------------------
// acc[ti][tj] = X[64 rows][K] * W[n0 + 32 tj + .][K]^T for one wave (rows: all 64 of the tile), K a multiple of 64.
// X: LDS, row-major bf16, pitch BP.  W: global bf16 [N][K] row-major; lane (i, h) streams 16 bytes (k = 8h .. 8h+7 of
// the 16-k step) of weight row n0 + 32 tj + i per step.  Weight fragments run one 64-k block ahead in a second
// register set (two alternating sets, no copies).
// KS = 16-k steps per prefetched weight block: 4 (two register sets of 32) at TI = 2, 2 (two sets of 16) at TI = 1,
// where four resident waves per SIMD cover the L2 latency instead of a deeper per-wave prefetch.
// Weight matrices in the bf16 mirror are stored in MFMA-FRAGMENT ORDER (bf16_pack_weights): for W [N][K], fragment
// (nt, ks) = rows 32 nt .. +32, k = 16 ks .. +16 is 64 consecutive 16-byte units, unit (h, c) = W[32 nt + c][16 ks + 8 h .. +8].
// One B-fragment load of a wave is then ONE contiguous 1 KB read (8 cache lines).  Row-major weights make the same
// load touch 32 lines (32 bytes of each of 32 rows) — at 8 loads per 16 MFMAs that kept the L1 tag pipeline, not the
// matrix cores, busy: the sweeps ran at 14 % of the bf16 MFMA rate.
// (buffer loads: lane * 16 in one VGPR, the fragment's offset in the scalar operand, the step in the immediate — no
// vector address arithmetic in front of the loads)
template <int KS>
__device__ inline void bf_load_b(const bfraw* __restrict__ W, int K, int n0, int Q, int lane, vu4 (&b)[KS][2]) {
  const int nks = K >> 4;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bfraw*>(W), 0, 0x4000000, 0x00020000);
  const unsigned voff = (unsigned)lane * 16u;
#pragma unroll
  for (int tj = 0; tj < 2; ++tj) {
    const unsigned soff = (unsigned)(((n0 >> 5) + tj) * nks + Q * KS) * 1024u;
#pragma unroll
    for (int s = 0; s < KS; ++s)
      b[s][tj] = __builtin_bit_cast(vu4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff + s * 1024, 0));
  }
}
// One 16 KS-k block: the A fragments of step s + 1 are read from LDS while the MFMAs of step s run (explicit rotation +
// a scheduling fence per step: left alone, hipcc parks every ds_read right in front of its MFMAs and waits for it).
template <int TI, int KS, int PITCH>
__device__ inline void bf_mma_block(const bfraw* __restrict__ X, int Q, int lane, const vu4 (&b)[KS][2], v16f (&acc)[TI][2]) {
  const int i = lane & 31, h = lane >> 5;
  const bfraw* xp = X + i * PITCH + Q * (16 * KS) + h * 8;
  vu4 a[2][TI];
#pragma unroll
  for (int ti = 0; ti < TI; ++ti) a[0][ti] = *reinterpret_cast<const vu4*>(xp + ti * 32 * PITCH);
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    if (s + 1 < KS) {
#pragma unroll
      for (int ti = 0; ti < TI; ++ti) a[(s + 1) & 1][ti] = *reinterpret_cast<const vu4*>(xp + ti * 32 * PITCH + (s + 1) * 16);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int ti = 0; ti < TI; ++ti)
        acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a[s & 1][ti]), __builtin_bit_cast(bf8, b[s][tj]),
                                                              acc[ti][tj], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}
template <int TI>
__device__ inline void bf_zero(v16f (&acc)[TI][2]) {
#pragma unroll
  for (int a = 0; a < TI; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
}
// X points at the first of the wave's 32 TI rows.
// `pre` holds weight block 0 of THIS product, requested by the previous call (`Wnext` / `Knext`: the product that
// follows; its block 0 is requested here as soon as this product's last block is in flight, so that it lands during
// the epilogue instead of costing an exposed L2 round trip at the top of every layer).  The block loop is unrolled for
// the compile-time block count NQ (K = 64 NQ / (4 / KS)): two alternating register sets, no copies for even NQ.
template <int TI, int PITCH = BP, int KSV = (TI == 1 ? 2 : 4)>
struct BfMma {
  static constexpr int KS = KSV;
  vu4 pre[KS][2];
  __device__ inline void request(const bfraw* __restrict__ W, int K, int n0, int lane) { bf_load_b<KS>(W, K, n0, 0, lane, pre); }
  template <int NQ>
  __device__ inline void run_fixed(const bfraw* __restrict__ X, const bfraw* __restrict__ W, int K, int n0, int lane,
                                   v16f (&acc)[TI][2], const bfraw* __restrict__ Wnext, int Knext, int n0next) {
    bf_zero<TI>(acc);
    vu4 alt[KS][2];
    __builtin_amdgcn_s_setprio(1);   // (the matrix loop outranks the other workgroup's epilogue on this SIMD)
#pragma unroll
    for (int Q = 0; Q < NQ; ++Q) {
      // request the block after this one into the set that is not being multiplied
      if (Q + 1 < NQ) {
        if (Q & 1) bf_load_b<KS>(W, K, n0, Q + 1, lane, pre); else bf_load_b<KS>(W, K, n0, Q + 1, lane, alt);
      } else if (Wnext) {
        if (Q & 1) bf_load_b<KS>(Wnext, Knext, n0next, 0, lane, pre); else bf_load_b<KS>(Wnext, Knext, n0next, 0, lane, alt);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (Q & 1) bf_mma_block<TI, KS, PITCH>(X, Q, lane, alt, acc); else bf_mma_block<TI, KS, PITCH>(X, Q, lane, pre, acc);
    }
    __builtin_amdgcn_s_setprio(0);
    if ((NQ & 1) && Wnext) {   // odd block count: the next product's block 0 sits in `alt`
#pragma unroll
      for (int s = 0; s < KS; ++s) { pre[s][0] = alt[s][0]; pre[s][1] = alt[s][1]; }
    }
  }
  // K is one of 64 (PE input), 256 (hidden) or 320 (the albedo net's input): the call sites know which
  template <int KK>
  __device__ inline void run(const bfraw* __restrict__ X, const bfraw* __restrict__ W, int n0, int lane, v16f (&acc)[TI][2],
                             const bfraw* __restrict__ Wnext, int Knext, int n0next) {
    run_fixed<KK / (16 * KS)>(X, W, KK, n0, lane, acc, Wnext, Knext, n0next);
  }
};
template <int TI, int PITCH = BP>
__device__ inline void bf_layer_mma(const bfraw* __restrict__ X, const bfraw* __restrict__ W, int K, int n0, int lane,
                                    v16f (&acc)[TI][2]) {
  BfMma<TI, PITCH> m;
  m.request(W, K, n0, lane);
  if (K == 256) m.template run<256>(X, W, n0, lane, acc, nullptr, 0, 0);
  else if (K == 64) m.template run<64>(X, W, n0, lane, acc, nullptr, 0, 0);
  else m.template run<320>(X, W, n0, lane, acc, nullptr, 0, 0);
}

// ---- accumulator-layout access to K8 matrices -------------------------------------------------------------------
// One "quad" = registers 4g .. 4g+3 of one 32 x 32 accumulator tile = points 8g + 4h .. +3 of one column = 8 bytes.
// Buffer accesses: resource based at the wave's first K8 block row (row0 is wave-uniform), the lane's (column, half)
// offset in ONE VGPR, the quad's block row in the scalar operand — no 64-bit vector address per quad (the epilogues of
// these sweeps are what the vector port is busy with).
struct Quad { float v[4]; };
__device__ inline __amdgpu_buffer_rsrc_t k8_rsrc(const bfraw* base, int64_t row0, int C) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<bfraw*>(base) + (size_t)(row0 >> 3) * C * 8, 0, 0x7ffffff0, 0x00020000);
}
__device__ inline Quad k8_load_quad(const bfraw* __restrict__ base, int64_t row0, int ti, int g, int col, int h) {
  const vu2 u = __builtin_bit_cast(vu2, __builtin_amdgcn_raw_buffer_load_b64(k8_rsrc(base, row0, FH), (unsigned)(col * 16 + 8 * h),
                                                                               (unsigned)((ti * 4 + g) * FH * 16), RNB_AUX_LD));
  return Quad{{bf_lo(u.x), bf_hi(u.x), bf_lo(u.y), bf_hi(u.y)}};
}
__device__ inline void k8_store_quad(bfraw* __restrict__ base, int64_t row0, int ti, int g, int col, int h, float a, float b,
                                     float c, float d, int C = FH) {
  const vu2 u = {pack2(a, b), pack2(c, d)};
  __builtin_amdgcn_raw_buffer_store_b64(u, k8_rsrc(base, row0, C), (unsigned)(col * 16 + 8 * h), (unsigned)((ti * 4 + g) * C * 16), RNB_AUX_ST);
}
// 8 rows of one column of an LDS tile (row-major, pitch P) -> one 16-byte K8 unit
template <int P>
__device__ inline vu4 lds_gather8(const bfraw* __restrict__ X, int blk, int c) {
  bfraw v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = X[(blk * 8 + j) * P + c];
  return vu4{(unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16),
             (unsigned)v[4] | ((unsigned)v[5] << 16), (unsigned)v[6] | ((unsigned)v[7] << 16)};
}
// one 16-byte K8 unit -> 8 rows of one column of an LDS tile
template <int P>
__device__ inline void lds_scatter8(bfraw* __restrict__ X, int blk, int c, vu4 u) {
  const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int j = 0; j < 8; ++j) X[(blk * 8 + j) * P + c] = (bfraw)((j & 1) ? (w[j >> 1] >> 16) : (w[j >> 1] & 0xffffu));
}
// a whole [64 x 256] tile of a K8 matrix in accumulator layout (issued early, consumed after the matrix loop)
template <int TI> struct AuxBf { vu2 q[TI][2][4]; };
template <int TI>
__device__ inline void k8_prefetch(const bfraw* __restrict__ base, int64_t row0, int n0, int lane, AuxBf<TI>& t) {
  const int c = lane & 31, h = lane >> 5;
  const __amdgpu_buffer_rsrc_t rs = k8_rsrc(base, row0, FH);
#pragma unroll
  for (int ti = 0; ti < TI; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        t.q[ti][tj][g] = __builtin_bit_cast(vu2, __builtin_amdgcn_raw_buffer_load_b64(
            rs, (unsigned)((n0 + c) * 16 + 8 * h), (unsigned)((ti * 4 + g) * FH * 16 + tj * 512), RNB_AUX_LD));
}
template <int TI>
__device__ inline float aux_at(const AuxBf<TI>& t, int ti, int tj, int r) {
  const vu2 u = t.q[ti][tj][r >> 2];
  const unsigned w = (r & 2) ? u.y : u.x;
  return (r & 1) ? bf_hi(w) : bf_lo(w);
}

// softplus(beta = 100) and its derivative for bf16 consumers: hardware exp2 / log2 / rcp without the compensation
// terms of the fp32 path (their error, ~1e-7 relative, is far below half a bf16 ulp = 2e-3 relative)
__device__ inline void softplus_aD_fast(float z, float& a, float& D) {
  constexpr float L2E = 1.44269504088896341f, LN2 = 0.693147180559945309f;
  const float t = z * 100.f;
  const float w = __builtin_amdgcn_exp2f(-fabsf(t) * L2E);
  const float u = 1.f + w;
  const float r = __builtin_amdgcn_rcpf(u);
  a = __builtin_fmaf(__builtin_amdgcn_logf(u), LN2 * 0.01f, fmaxf(z, 0.f));
  D = t >= 0.f ? r : w * r;
}

// ---------------------------------------------------------------------------------------------------------------
// F sweep
// ---------------------------------------------------------------------------------------------------------------
struct BfFwdArgs {
  const float* pts;        // [M,3]
  int64_t M;
  const float* packed;     // fp32 packed weights (biases, sdf head row)
  const bfraw* wbf;        // bf16 mirror of the packed buffer (same offsets)
  int nh, skip, pe, multires, Ep;
  float scale;
  int n_real[RNB_MAX_LIN];
  int Kp[RNB_MAX_LIN];
  long long w_off[RNB_MAX_LIN], b_off[RNB_MAX_LIN];
  long long wsdf_off, bsdf_off;
  int with_feat, F, Cinp;
  long long wf_off, bf_off;
  float* cin;              // [Mp,Cinp] fp32 feature block destination (with_feat, cin8 == nullptr)
  bfraw* cin8;             // [Mp,Cinp] K8 bf16 feature block destination (bf16 albedo path) or nullptr
  float* sdf;              // [Mp]
  float* x4;               // [Mp,4]            (SAVE)
  bfraw* e;                // [Mp,64]  K8       (SAVE) positional encoding = input of layer 0
  bfraw* a[RNB_MAX_LIN];   // [Mp,256] K8       (SAVE)
  bfraw* D[RNB_MAX_LIN];   // [Mp,256] K8       (SAVE)
  GridGen grid;
};

template <bool SAVE, int TI>
__global__ __launch_bounds__(BfCfg<TI>::NT, TI == 1 ? 4 : 2) void bf_forward_kernel(BfFwdArgs g) {
  constexpr int NW = BfCfg<TI>::NW, NT = BfCfg<TI>::NT;
  __shared__ __attribute__((aligned(16))) bfraw X[BT * BP];
  __shared__ float E[BT * FEP];     // fp32 copy of the positional encoding for the skip connection
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
  const int64_t row0 = (int64_t)blockIdx.x * BT;
  const int n0 = (wave & 3) * 64;          // column group of the wave
  const int rb = (wave >> 2) * 32;         // first row of the wave inside the tile (TI == 1: two row halves)
  const int64_t rowW = row0 + rb;
  const bfraw* Xw = X + rb * BP;
  const int h = lane >> 5, cl = lane & 31;

  // ---- positional encoding of the tile (fp32 math, models/embedder.py:40-46) ---------------------------------
  {
    constexpr int PARTS = NT / BT;   // 4 or 8 threads per point
    const int p = tid % BT, part = tid / BT;
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
    bfraw* xr = X + p * BP;
    float* er = E + p * FEP;
    if (part == 0) {
#pragma unroll
      for (int d = 0; d < 3; ++d) { xr[d] = to_bf(x[d]); er[d] = x[d]; }
      for (int c = g.pe; c < g.Ep; ++c) xr[c] = 0;
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
        xr[c] = to_bf(s); xr[c + 3] = to_bf(co);
        er[c] = s; er[c + 3] = co;
      }
    }
  }
  __syncthreads();
  if (SAVE) {   // e (bf16, K8, 64 columns): the Y operand of layer 0's weight gradient
    for (int u = tid; u < (BT / 8) * g.Ep; u += NT) {
      const int blk = u / g.Ep, c = u - blk * g.Ep;
      bfraw v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = X[(blk * 8 + j) * BP + c];
      vu4 o = {(unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16),
               (unsigned)v[4] | ((unsigned)v[5] << 16), (unsigned)v[6] | ((unsigned)v[7] << 16)};
      *reinterpret_cast<vu4*>(g.e + (((size_t)(row0 >> 3) + blk) * g.Ep + c) * 8) = o;
    }
  }

  v16f acc[TI][2];
  // cross-layer weight prefetch (the next layer's block 0 in flight during the epilogue) only where the registers are
  // free: with SAVE the epilogue also holds the D values and the packed stores (29 spilled registers otherwise)
  constexpr bool XL = true;
  BfMma<TI> mm;
  if (XL) mm.request(g.wbf + g.w_off[0], g.Kp[0], n0, lane);
  for (int l = 0; l < g.nh; ++l) {
    const bfraw* wn = !XL ? nullptr : (l + 1 < g.nh ? g.wbf + g.w_off[l + 1] : (g.with_feat ? g.wbf + g.wf_off : nullptr));
    if (!XL) mm.request(g.wbf + g.w_off[l], g.Kp[l], n0, lane);
    if (l == 0) mm.template run<64>(Xw, g.wbf + g.w_off[0], n0, lane, acc, wn, FH, n0);   // (fused_supported: Ep = 64, hidden 256)
    else mm.template run<256>(Xw, g.wbf + g.w_off[l], n0, lane, acc, wn, FH, n0);
    lds_barrier();   // every wave has finished reading the input activations (the tile is updated in place)
    const int lo = opaque(lane), h = lo >> 5, cl = lo & 31;
    const float* bias = g.packed + g.b_off[l];
    const int n_real = g.n_real[l];
    const bool pe_tail = (l + 1 == g.skip);
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
      const int col = n0 + tj * 32 + cl;
      const float bc = bias[col];
      const bool tile_full = n0 + tj * 32 + 32 <= n_real;   // wave-uniform: no per-element column tests
      const bool real = col < n_real;
      const bool pe_col = pe_tail && !real && col < n_real + g.pe;
#pragma unroll
      for (int ti = 0; ti < TI; ++ti) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float a[4], D[4];
          if (tile_full) {
#pragma unroll
            for (int j = 0; j < 4; ++j) softplus_aD_fast(acc[ti][tj][4 * q + j] + bc, a[j], D[j]);
          } else {   // only the tile straddling the skip connection's PE columns
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              if (real) softplus_aD_fast(acc[ti][tj][4 * q + j] + bc, a[j], D[j]);
              else { a[j] = pe_col ? E[(rb + 4 * h) * FEP + (col - n_real) + (ti * 32 + 8 * q + j) * FEP] : 0.f; D[j] = 0.f; }
            }
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) X[(rb + 4 * h) * BP + col + (ti * 32 + 8 * q + j) * BP] = to_bf(a[j]);
          if (SAVE) {
            k8_store_quad(g.a[l], rowW, ti, q, col, h, a[0], a[1], a[2], a[3]);
            k8_store_quad(g.D[l], rowW, ti, q, col, h, D[0], D[1], D[2], D[3]);
          }
        }
      }
    }
    lds_barrier();   // the new activations are visible to every wave
  }

  // ---- sdf head: row 0 of the output layer, fp32 weights on the bf16 activations ---------------------------------
  {
    const float* ws = g.packed + g.wsdf_off;
    float w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) w[u] = ws[lane + 64 * u];
    const float bs = g.packed[g.bsdf_off];
    for (int rr = 0; rr < BT / NW; ++rr) {
      const int row = wave * (BT / NW) + rr;
      float s = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) s = fmaf(bf_f(X[row * BP + lane + 64 * u]), w[u], s);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      if (lane == 0) {
        const float v = (s + bs) / g.scale;
        if (!g.grid.on) g.sdf[row0 + row] = v;
        else if (row0 + row < g.M) g.sdf[row0 + row] = v * g.grid.out_scale;
      }
    }
  }
  // ---- feature head: rows 1.. of the output layer, written (fp32) into the albedo network's input -------------------
  if (g.with_feat) {
    if (!XL) mm.request(g.wbf + g.wf_off, FH, n0, lane);
    mm.template run<256>(Xw, g.wbf + g.wf_off, n0, lane, acc, nullptr, 0, 0);   // (XL: block 0 was requested by the last hidden layer)
    const float* bias = g.packed + g.bf_off;
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
      const int col = n0 + tj * 32 + cl;
      if (col < g.F) {
        const float bc = bias[col];
        if (g.cin8 != nullptr) {
#pragma unroll
          for (int ti = 0; ti < TI; ++ti)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              k8_store_quad(g.cin8, rowW, ti, q, col, h, acc[ti][tj][4 * q] + bc, acc[ti][tj][4 * q + 1] + bc,
                            acc[ti][tj][4 * q + 2] + bc, acc[ti][tj][4 * q + 3] + bc, g.Cinp);
        } else {
#pragma unroll
          for (int ti = 0; ti < TI; ++ti)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = rb + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
              g.cin[(size_t)(row0 + row) * g.Cinp + col] = acc[ti][tj][r] + bc;
            }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// backward-shaped sweeps
// ---------------------------------------------------------------------------------------------------------------
struct BfBwdArgs {
  const float* packed;
  const bfraw* wbf;
  int64_t M;
  int nh, skip, pe, multires, Ep;
  float inv_scale;
  int n_real[RNB_MAX_LIN];
  int Kp[RNB_MAX_LIN];
  long long w_off[RNB_MAX_LIN], wT_off[RNB_MAX_LIN];
  long long wsdf_off, wfT_off;
  bfraw* D[RNB_MAX_LIN];
  bfraw* gz[RNB_MAX_LIN];
  bfraw* u[RNB_MAX_LIN + 1];   // u[0]: [Mp,64] K8 (written by RA from geb); u[l >= 1]: [Mp,256] K8
  bfraw* zR[RNB_MAX_LIN];
  bfraw* zb[RNB_MAX_LIN];
  bfraw* fbar8;         // [Mp,256] K8: the feature part of the albedo net's input adjoint (written by FB)
  const float* x4;      // [Mp,4]
  float* nrm;           // [Mp,4]      (R)
  const float* geb;     // [Mp,Ep] fp32 row-major (RA)
  const float* sbar;    // [Mp]        (FB)
  const float* fbar;    // [Mp,ld_fbar] fp32 row-major, first 256 columns, or nullptr (FB; fp32 albedo path)
  int ld_fbar;
  int fbar_in_k8;       // 1: fbar8 already holds the feature adjoint (bf16 albedo path), read it instead of `fbar`
};

// R: gz_l = g_l * D_l, g_{l-1} = gz_l W_l, normal = J_pe^T g_e
template <int TI>
__global__ __launch_bounds__(BfCfg<TI>::NT, TI == 1 ? 4 : 2) void bf_reverse_kernel(BfBwdArgs g) {
  constexpr int NT = BfCfg<TI>::NT;
  __shared__ __attribute__((aligned(16))) bfraw X[BT * BP];
  __shared__ float GE[BT * FEP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
  const int64_t row0 = (int64_t)blockIdx.x * BT;
  const int n0 = (wave & 3) * 64;          // column group of the wave
  const int rb = (wave >> 2) * 32;         // first row of the wave inside the tile (TI == 1: two row halves)
  const int64_t rowW = row0 + rb;
  const bfraw* Xw = X + rb * BP;
  const int h = lane >> 5, cl = lane & 31;

  // seed: gz_{nh-1} = w_sdf * D_{nh-1}; one K8 unit (8 points of one column) per thread and step
  {
    const bfraw* Dl = g.D[g.nh - 1] + (size_t)(row0 >> 3) * FH * 8;
    bfraw* gzl = g.gz[g.nh - 1] + (size_t)(row0 >> 3) * FH * 8;
    const float* ws = g.packed + g.wsdf_off;
    for (int u = tid; u < (BT / 8) * FH; u += NT) {
      const int blk = u / FH, c = u - blk * FH;
      const vu4 d = *reinterpret_cast<const vu4*>(Dl + (size_t)u * 8);
      const float w = ws[c];
      const float v[8] = {bf_lo(d.x) * w, bf_hi(d.x) * w, bf_lo(d.y) * w, bf_hi(d.y) * w,
                          bf_lo(d.z) * w, bf_hi(d.z) * w, bf_lo(d.w) * w, bf_hi(d.w) * w};
      const vu4 o = {pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
      *reinterpret_cast<vu4*>(gzl + (size_t)u * 8) = o;
#pragma unroll
      for (int j = 0; j < 8; ++j) X[(blk * 8 + j) * BP + c] = to_bf(v[j]);
    }
    for (int idx = tid; idx < BT * FEP; idx += NT) GE[idx] = 0.f;
  }
  __syncthreads();

  v16f acc[TI][2];
  AuxBf<TI> aD;
  BfMma<TI> mm;
  mm.request(g.wbf + g.wT_off[g.nh - 1], FH, n0, lane);
  for (int l = g.nh - 1; l >= 1; --l) {
    k8_prefetch<TI>(g.D[l - 1], rowW, n0, opaque(lane), aD);
    const bfraw* wn = (l > 1 || n0 < 64) ? g.wbf + g.wT_off[l - 1] : nullptr;   // layer 0's product: wave(s) of columns 0..63
    mm.template run<256>(Xw, g.wbf + g.wT_off[l], n0, lane, acc, wn, FH, n0);   // g = gz_l W_l  (columns = inputs of layer l)
    lds_barrier();
    const int lo = opaque(lane), h = lo >> 5, cl = lo & 31;
    const bool is_skip = (l == g.skip);
    const int ksplit = is_skip ? FH - g.pe : FH;   // columns that belong to layer l-1's output
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
      const int col = n0 + tj * 32 + cl;
      const bool tile_full = n0 + tj * 32 + 32 <= ksplit;   // wave-uniform: no per-element column tests
#pragma unroll
      for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float o[4];
          if (tile_full) {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = acc[ti][tj][4 * q + j] * aux_at(aD, ti, tj, 4 * q + j);
          } else {   // only the tile straddling the skip connection's PE columns
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int r = 4 * q + j;
              const float v = acc[ti][tj][r];
              if (col < ksplit) o[j] = v * aux_at(aD, ti, tj, r);
              else {
                if (col < ksplit + g.pe) GE[(rb + 4 * h) * FEP + (col - ksplit) + (ti * 32 + 8 * q + j) * FEP] = v;   // skip connection: straight to g_e
                o[j] = 0.f;
              }
            }
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) X[(rb + 4 * h) * BP + col + (ti * 32 + 8 * q + j) * BP] = to_bf(o[j]);
          k8_store_quad(g.gz[l - 1], rowW, ti, q, col, h, o[0], o[1], o[2], o[3]);
        }
    }
    lds_barrier();
  }
  // layer 0: g_e += gz_0 W_0 (Ep = 64 columns: wave 0)
  if (n0 < 64) {
    if (g.nh == 1) mm.request(g.wbf + g.wT_off[0], FH, n0, lane);
    mm.template run<256>(Xw, g.wbf + g.wT_off[0], n0, lane, acc, nullptr, 0, 0);
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
      const int col = n0 + tj * 32 + cl;
      if (col < g.pe) {
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
          for (int r = 0; r < 16; ++r) GE[(rb + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * FEP + col] += acc[ti][tj][r];
      }
    }
  }
  __syncthreads();
  if (tid < BT) {   // normal = J_pe(x)^T g_e  (fp32)
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

// RA: u_{l+1} = (u_l W_l^T) * D_l, zR_l = 100 (u_l W_l^T) gz_l (1 - D_l)
template <int TI>
__global__ __launch_bounds__(BfCfg<TI>::NT, TI == 1 ? 4 : 2) void bf_ra_kernel(BfBwdArgs g) {
  constexpr int NT = BfCfg<TI>::NT;
  __shared__ __attribute__((aligned(16))) bfraw X[BT * BP];
  __shared__ float E[BT * FEP];   // adjoint of g_e of the tile (re-enters at the skip connection)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
  const int64_t row0 = (int64_t)blockIdx.x * BT;
  const int n0 = (wave & 3) * 64;          // column group of the wave
  const int rb = (wave >> 2) * 32;         // first row of the wave inside the tile (TI == 1: two row halves)
  const int64_t rowW = row0 + rb;
  const bfraw* Xw = X + rb * BP;

  for (int idx = tid; idx < BT * g.Ep; idx += NT) {
    const int r = idx / g.Ep, c = idx - r * g.Ep;
    const float v = g.geb[(row0 + r) * g.Ep + c];
    X[r * BP + c] = to_bf(v);
    if (c < FEP) E[r * FEP + c] = v;
  }
  __syncthreads();
  // u_0 in K8 (the Y operand of layer 0's weight gradient)
  for (int u = tid; u < (BT / 8) * g.Ep; u += NT) {
    const int blk = u / g.Ep, c = u - blk * g.Ep;
    bfraw v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = X[(blk * 8 + j) * BP + c];
    vu4 o = {(unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16),
             (unsigned)v[4] | ((unsigned)v[5] << 16), (unsigned)v[6] | ((unsigned)v[7] << 16)};
    *reinterpret_cast<vu4*>(g.u[0] + (((size_t)(row0 >> 3) + blk) * g.Ep + c) * 8) = o;
  }

  v16f acc[TI][2];
  AuxBf<TI> aD, aG;
  // no cross-layer weight prefetch and 32-k weight blocks here: the two epilogue operand tiles already fill the registers
  BfMma<TI> mm;
  mm.request(g.wbf + g.w_off[0], g.Kp[0], n0, lane);
  for (int l = 0; l < g.nh; ++l) {
    const int lp = opaque(lane);
    k8_prefetch<TI>(g.D[l], rowW, n0, lp, aD);
    k8_prefetch<TI>(g.gz[l], rowW, n0, lp, aG);
    const bfraw* wn = l + 1 < g.nh ? g.wbf + g.w_off[l + 1] : nullptr;
    if (l == 0) mm.template run<64>(Xw, g.wbf + g.w_off[0], n0, lane, acc, wn, FH, n0);   // gzb = u_l W_l^T
    else mm.template run<256>(Xw, g.wbf + g.w_off[l], n0, lane, acc, wn, FH, n0);
    lds_barrier();
    const int lo = opaque(lane), h = lo >> 5, cl = lo & 31;
    const int n_real = g.n_real[l];
    const bool pe_tail = (l + 1 == g.skip);
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
      const int col = n0 + tj * 32 + cl;
      const bool tile_full = n0 + tj * 32 + 32 <= n_real;   // wave-uniform: no per-element column tests
#pragma unroll
      for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float un[4], zr[4];
          if (tile_full) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int r = 4 * q + j;
              const float v = acc[ti][tj][r];
              un[j] = v * aux_at(aD, ti, tj, r);
              zr[j] = ((v - un[j]) * aux_at(aG, ti, tj, r)) * 100.f;
            }
          } else {   // only the tile straddling the skip connection's PE columns
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int r = 4 * q + j;
              const float v = acc[ti][tj][r];
              if (col < n_real) {
                un[j] = v * aux_at(aD, ti, tj, r);
                zr[j] = ((v - un[j]) * aux_at(aG, ti, tj, r)) * 100.f;
              } else {
                zr[j] = 0.f;
                un[j] = (pe_tail && col < n_real + g.pe) ? E[(rb + 4 * h) * FEP + (col - n_real) + (ti * 32 + 8 * q + j) * FEP] : 0.f;
              }
            }
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) X[(rb + 4 * h) * BP + col + (ti * 32 + 8 * q + j) * BP] = to_bf(un[j]);
          k8_store_quad(g.u[l + 1], rowW, ti, q, col, h, un[0], un[1], un[2], un[3]);
          k8_store_quad(g.zR[l], rowW, ti, q, col, h, zr[0], zr[1], zr[2], zr[3]);
        }
    }
    lds_barrier();
  }
}

// FB: zb_{l-1} = (zb_l W_l) * D_{l-1} + zR_{l-1}, head: ab_{nh-1} = fbar W_feat + sbar / scale * w_sdf
template <int TI>
__global__ __launch_bounds__(BfCfg<TI>::NT, TI == 1 ? 4 : 2) void bf_fb_kernel(BfBwdArgs g) {
  constexpr int NT = BfCfg<TI>::NT;
  __shared__ __attribute__((aligned(16))) bfraw X[BT * BP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
  const int64_t row0 = (int64_t)blockIdx.x * BT;
  const int n0 = (wave & 3) * 64;          // column group of the wave
  const int rb = (wave >> 2) * 32;         // first row of the wave inside the tile (TI == 1: two row halves)
  const int64_t rowW = row0 + rb;
  const bfraw* Xw = X + rb * BP;

  v16f acc[TI][2];
  AuxBf<TI> aD, aZ;
  BfMma<TI> mm;
  const bool has_head = g.fbar_in_k8 || g.fbar != nullptr;
  if (has_head) mm.request(g.wbf + g.wfT_off, FH, n0, lane);
  else if (g.nh > 1) mm.request(g.wbf + g.wT_off[g.nh - 1], FH, n0, lane);
  const bfraw* w_first = g.nh > 1 ? g.wbf + g.wT_off[g.nh - 1] : nullptr;
  bf_zero<TI>(acc);
  if (g.fbar_in_k8) {
    const bfraw* fb = g.fbar8 + (size_t)(row0 >> 3) * FH * 8;
    for (int u = tid; u < (BT / 8) * FH; u += NT)
      lds_scatter8<BP>(X, u / FH, u % FH, *reinterpret_cast<const vu4*>(fb + (size_t)u * 8));
    __syncthreads();
    mm.template run<256>(Xw, g.wbf + g.wfT_off, n0, lane, acc, w_first, FH, n0);
    lds_barrier();
  } else if (g.fbar != nullptr) {
    // fbar (fp32 row-major, from the albedo net's backward) -> LDS bf16, and K8 for the feature head's dW
    for (int idx = tid; idx < BT * FH / 4; idx += NT) {
      const int r = idx >> 6, c4 = idx & 63;
      const vf4 v = *reinterpret_cast<const vf4*>(g.fbar + (size_t)(row0 + r) * g.ld_fbar + c4 * 4);
      const vu2 o = {pack2(v.x, v.y), pack2(v.z, v.w)};
      *reinterpret_cast<vu2*>(X + r * BP + c4 * 4) = o;
    }
    __syncthreads();
    for (int u = tid; u < (BT / 8) * FH; u += NT) {
      const int blk = u / FH, c = u - blk * FH;
      bfraw v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = X[(blk * 8 + j) * BP + c];
      vu4 o = {(unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16),
               (unsigned)v[4] | ((unsigned)v[5] << 16), (unsigned)v[6] | ((unsigned)v[7] << 16)};
      *reinterpret_cast<vu4*>(g.fbar8 + (((size_t)(row0 >> 3) + blk) * FH + c) * 8) = o;
    }
    mm.template run<256>(Xw, g.wbf + g.wfT_off, n0, lane, acc, w_first, FH, n0);
    lds_barrier();
  }
  for (int l = g.nh - 1; l >= 0; --l) {
    const int lo = opaque(lane), h = lo >> 5, cl = lo & 31;
    k8_prefetch<TI>(g.D[l], rowW, n0, lo, aD);
    k8_prefetch<TI>(g.zR[l], rowW, n0, lo, aZ);
    const int n_real = g.n_real[l];
    const bool head = (l == g.nh - 1);
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
      const int col = n0 + tj * 32 + cl;
      const float ws = head ? g.packed[g.wsdf_off + col] : 0.f;
#pragma unroll
      for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float zb[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = 4 * q + j;
            float v = acc[ti][tj][r];
            if (head) v = fmaf(g.sbar[row0 + rb + 4 * h + (ti * 32 + 8 * q + j)] * g.inv_scale, ws, v);   // the sdf head's contribution
            zb[j] = col < n_real ? fmaf(v, aux_at(aD, ti, tj, r), aux_at(aZ, ti, tj, r)) : 0.f;
            X[(rb + 4 * h) * BP + col + (ti * 32 + 8 * q + j) * BP] = to_bf(zb[j]);
          }
          k8_store_quad(g.zb[l], rowW, ti, q, col, h, zb[0], zb[1], zb[2], zb[3]);
        }
    }
    if (l == 0) break;
    lds_barrier();
    mm.template run<256>(Xw, g.wbf + g.wT_off[l], n0, lane, acc, l > 1 ? g.wbf + g.wT_off[l - 1] : nullptr, FH, n0);   // ab_{l-1} = zb_l W_l
    lds_barrier();   // every wave has finished reading the tile
  }
}

// ---------------------------------------------------------------------------------------------------------------
// albedo network (RenderingNetwork, mode no_view_dir: models/fields.py:177-215) in bf16
// ---------------------------------------------------------------------------------------------------------------
// Input [feature | pe(p) | pe(n) | 0] (Cinp = 320 columns, the packed column order of weightnorm.hip), nc hidden ReLU
// layers of width 256, output layer (<= 4 rows, sigmoid) on the VALU with fp32 weights.  Saved for the backward in K8
// bf16: cin8 (all Cinp columns), ac8[l].  One 64-point tile per workgroup, 4 waves of 64 rows x 64 columns.
constexpr int CP = 328;    // LDS pitch (bf16 elements) of a [point][Cinp <= 320] row: 656 bytes, conflict-free ds_read_b128
constexpr int CMAX = 320;

struct BfColArgs {
  const float* pts;        // [M,3]
  const float* nrm;        // [Mp,4]
  int64_t M;
  const float* packed;
  const bfraw* wbf;
  int nc, F, pev, multires_view, Cinp, Co, squeeze;
  int Kp[RNB_MAX_LIN];
  long long w_off[RNB_MAX_LIN], wT_off[RNB_MAX_LIN], b_off[RNB_MAX_LIN];
  long long wo_off, bo_off;
  int ldwo;
  bfraw* cin8;             // [Mp,Cinp] K8: features written by the F sweep; this kernel adds the pe columns
  bfraw* ac8[RNB_MAX_LIN]; // [Mp,256] K8
  float* alb;              // [Mp,4]
  // backward
  const float* albbar;     // [Mp,4]
  bfraw* zc8[RNB_MAX_LIN]; // [Mp,256] K8
  bfraw* fbar8;            // [Mp,256] K8 (out): adjoint of the feature columns
  float* cinb;             // [Mp,Cinp] fp32 row-major: only the pe columns F.. are written (consumed by nbar_geb_kernel)
};

__global__ __launch_bounds__(256, 2) void bf_color_fwd_kernel(BfColArgs g) {
  __shared__ __attribute__((aligned(16))) bfraw X[BT * CP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
  const int64_t row0 = (int64_t)blockIdx.x * BT;
  const int n0 = wave * 64;
  const int h = lane >> 5, cl = lane & 31;
  // features: K8 units of cin8 -> LDS rows
  {
    const bfraw* src = g.cin8 + (size_t)(row0 >> 3) * g.Cinp * 8;
    for (int u = tid; u < (BT / 8) * g.F; u += 256) {
      const int blk = u / g.F, c = u - blk * g.F;
      lds_scatter8<CP>(X, blk, c, *reinterpret_cast<const vu4*>(src + ((size_t)blk * g.Cinp + c) * 8));
    }
  }
  // pe(p), pe(n) (fp32 math, models/embedder.py:40-46): 4 threads per point = (which vector, even / odd octaves)
  {
    const int p = tid & 63, part = tid >> 6, which = part >> 1, sub = part & 1;
    const int64_t row = row0 + p;
    float v[3] = {0.f, 0.f, 0.f};
    if (row < g.M) {
      if (which == 0) { v[0] = g.pts[row * 3]; v[1] = g.pts[row * 3 + 1]; v[2] = g.pts[row * 3 + 2]; }
      else { v[0] = g.nrm[row * 4]; v[1] = g.nrm[row * 4 + 1]; v[2] = g.nrm[row * 4 + 2]; }
    }
    bfraw* xr = X + p * CP + g.F + which * g.pev;
    if (sub == 0) {
#pragma unroll
      for (int d = 0; d < 3; ++d) xr[d] = to_bf(v[d]);
      if (which == 1)
        for (int c = g.F + 2 * g.pev; c < g.Cinp; ++c) X[p * CP + c] = 0;
    }
    for (int k = sub; k < g.multires_view; k += 2) {
      const float f = (float)(1 << k);
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        float sn, co;
        sincosf(v[d] * f, &sn, &co);
        xr[3 + 6 * k + d] = to_bf(sn);
        xr[3 + 6 * k + 3 + d] = to_bf(co);
      }
    }
  }
  __syncthreads();
  // the pe columns of the input in K8 (Y operand of layer 0's weight gradient)
  {
    const int W = g.Cinp - g.F;
    for (int u = tid; u < (BT / 8) * W; u += 256) {
      const int blk = u / W, c = g.F + (u - blk * W);
      *reinterpret_cast<vu4*>(g.cin8 + (((size_t)(row0 >> 3) + blk) * g.Cinp + c) * 8) = lds_gather8<CP>(X, blk, c);
    }
  }
  v16f acc[2][2];
  for (int l = 0; l < g.nc; ++l) {
    bf_layer_mma<2, CP>(X, g.wbf + g.w_off[l], g.Kp[l], n0, lane, acc);
    lds_barrier();
    const float* bias = g.packed + g.b_off[l];
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
      const int col = n0 + tj * 32 + cl;
      const float bc = bias[col];
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float a[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            a[j] = relu_nan(acc[ti][tj][4 * q + j] + bc);
            X[(ti * 32 + 8 * q + 4 * h + j) * CP + col] = to_bf(a[j]);
          }
          k8_store_quad(g.ac8[l], row0, ti, q, col, h, a[0], a[1], a[2], a[3]);
        }
    }
    lds_barrier();
  }
  // output layer + sigmoid: fp32 weights on the bf16 activations, 16 rows per wave
  {
    float w[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int u = 0; u < 4; ++u) w[c][u] = c < g.Co ? g.packed[g.wo_off + (long long)c * g.ldwo + lane + 64 * u] : 0.f;
    for (int rr = 0; rr < BT / 4; ++rr) {
      const int row = wave * (BT / 4) + rr;
      float sc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float a = bf_f(X[row * CP + lane + 64 * u]);
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
          v = sc[lane] + g.packed[g.bo_off + lane];
          if (g.squeeze) v = 1.f / (1.f + expf(-v));
        }
        g.alb[(row0 + row) * 4 + lane] = v;
      }
    }
  }
}

// backward: zo = albbar * alb (1 - alb); zc_last = (zo Wo) * relu'; zc_{l-1} = (zc_l W_l) * relu'; cinb = zc_0 W_0
__global__ __launch_bounds__(256, 2) void bf_color_bwd_kernel(BfColArgs g) {
  __shared__ __attribute__((aligned(16))) bfraw X[BT * BP];
  __shared__ float ZO[BT * 4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
  const int64_t row0 = (int64_t)blockIdx.x * BT;
  const int n0 = wave * 64;
  const int h = lane >> 5, cl = lane & 31;
  if (tid < BT) {
    const int64_t row = row0 + tid;
    const vf4 a4 = *reinterpret_cast<const vf4*>(g.alb + row * 4);
    const vf4 g4 = *reinterpret_cast<const vf4*>(g.albbar + row * 4);
#pragma unroll
    for (int c = 0; c < 4; ++c)
      ZO[tid * 4 + c] = (c < g.Co && row < g.M) ? g4[c] * (g.squeeze ? a4[c] * (1.f - a4[c]) : 1.f) : 0.f;
  }
  __syncthreads();
  {   // zc_{nc-1}: one K8 unit (8 points of one column) per thread and step
    const int L = g.nc - 1;
    const bfraw* ac = g.ac8[L] + (size_t)(row0 >> 3) * FH * 8;
    bfraw* zc = g.zc8[L] + (size_t)(row0 >> 3) * FH * 8;
    for (int u = tid; u < (BT / 8) * FH; u += 256) {
      const int blk = u / FH, c = u - blk * FH;
      float w[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) w[k] = k < g.Co ? g.packed[g.wo_off + (long long)k * g.ldwo + c] : 0.f;
      const vu4 a = *reinterpret_cast<const vu4*>(ac + (size_t)u * 8);
      const unsigned aw[4] = {a.x, a.y, a.z, a.w};
      float z[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float av = (j & 1) ? bf_hi(aw[j >> 1]) : bf_lo(aw[j >> 1]);
        const float* zo = ZO + (blk * 8 + j) * 4;
        const float t = fmaf(zo[0], w[0], fmaf(zo[1], w[1], fmaf(zo[2], w[2], zo[3] * w[3])));
        z[j] = av > 0.f ? t : 0.f;
        X[(blk * 8 + j) * BP + c] = to_bf(z[j]);
      }
      *reinterpret_cast<vu4*>(zc + (size_t)u * 8) = vu4{pack2(z[0], z[1]), pack2(z[2], z[3]), pack2(z[4], z[5]), pack2(z[6], z[7])};
    }
  }
  __syncthreads();
  v16f acc[2][2];
  AuxBf<2> aA;
  for (int l = g.nc - 1; l >= 1; --l) {
    k8_prefetch<2>(g.ac8[l - 1], row0, n0, lane, aA);
    bf_layer_mma<2>(X, g.wbf + g.wT_off[l], FH, n0, lane, acc);   // zc_l W_l  (columns = inputs of layer l)
    lds_barrier();
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
      const int col = n0 + tj * 32 + cl;
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float z[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            z[j] = aux_at(aA, ti, tj, 4 * q + j) > 0.f ? acc[ti][tj][4 * q + j] : 0.f;
            X[(ti * 32 + 8 * q + 4 * h + j) * BP + col] = to_bf(z[j]);
          }
          k8_store_quad(g.zc8[l - 1], row0, ti, q, col, h, z[0], z[1], z[2], z[3]);
        }
    }
    lds_barrier();
  }
  // cinb = zc_0 W_0: columns 0 .. F-1 (features) -> fbar8 (K8 bf16, what the FB sweep and the feature head's dW read);
  // columns F .. Cin-1 (pe(p) | pe(n)) -> fp32 row-major cinb (the normal's adjoint, nbar_geb_kernel)
  bf_layer_mma<2>(X, g.wbf + g.wT_off[0], FH, n0, lane, acc);
#pragma unroll
  for (int tj = 0; tj < 2; ++tj) {
    const int col = n0 + tj * 32 + cl;
    if (col < g.F) {
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          k8_store_quad(g.fbar8, row0, ti, q, col, h, acc[ti][tj][4 * q], acc[ti][tj][4 * q + 1], acc[ti][tj][4 * q + 2],
                        acc[ti][tj][4 * q + 3]);
    }
  }
  if (wave == 0) {   // the 64 pe columns: one more 64 x 64 block
    bf_layer_mma<2>(X, g.wbf + g.wT_off[0], FH, g.F, lane, acc);
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
      const int col = g.F + tj * 32 + cl;
      if (col < g.Cinp) {
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            g.cinb[(size_t)(row0 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * g.Cinp + col] = acc[ti][tj][r];
      }
    }
  }
}

// gradient of the albedo output layer: dWo[c][k] += sum_rows zo[row][c] ac_last[row][k], dbo[c] += sum_rows zo[row][c].
// One thread per column k and point slab (one slab in the deterministic variant).
__global__ __launch_bounds__(1024) void bf_color_out_bwd_kernel(const bfraw* __restrict__ ac, const float* __restrict__ alb,
                                                                const float* __restrict__ albbar, int Co, int squeeze,
                                                                int64_t M, int64_t rows_per_blk, int ldwo,
                                                                float* __restrict__ dWo, float* __restrict__ dbo) {
  __shared__ double red[4][4][FH], redb[4][4];
  const int c = threadIdx.x & 255, ph = threadIdx.x >> 8;   // column, one of 4 row phases
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk;
  const int64_t r1 = r0 + rows_per_blk < M ? r0 + rows_per_blk : M;
  double s[4] = {0.0, 0.0, 0.0, 0.0}, sb[4] = {0.0, 0.0, 0.0, 0.0};
  for (int64_t r = r0 + 8 * ph; r < r1; r += 32) {
    const vu4 av = *reinterpret_cast<const vu4*>(ac + ((size_t)(r >> 3) * FH + c) * 8);
    const unsigned aw[4] = {av.x, av.y, av.z, av.w};
    float t[4] = {0.f, 0.f, 0.f, 0.f}, tb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (r + j < r1) {
        const float a = (j & 1) ? bf_hi(aw[j >> 1]) : bf_lo(aw[j >> 1]);
        const vf4 a4 = *reinterpret_cast<const vf4*>(alb + (r + j) * 4);
        const vf4 g4 = *reinterpret_cast<const vf4*>(albbar + (r + j) * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float zo = k < Co ? g4[k] * (squeeze ? a4[k] * (1.f - a4[k]) : 1.f) : 0.f;
          t[k] = fmaf(zo, a, t[k]);
          tb[k] += zo;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { s[k] += (double)t[k]; sb[k] += (double)tb[k]; }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    red[ph][k][c] = s[k];
    if (c == 0) redb[ph][k] = sb[k];
  }
  __syncthreads();
  if (ph == 0) {
    for (int k = 0; k < Co; ++k) {
      atomicAdd(dWo + (size_t)k * ldwo + c, (float)(red[0][k][c] + red[1][k][c] + red[2][k][c] + red[3][k][c]));
      if (c == 0) atomicAdd(dbo + k, (float)(redb[0][k] + redb[1][k] + redb[2][k] + redb[3][k]));
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// dW: every weight-gradient job of one backward in one launch
// ---------------------------------------------------------------------------------------------------------------
// One job = dW[N x K] (+)= sum over pairs X_p^T Y_p, X_p [M x 256] K8 (all N = 256 columns), Y_p [M x Cy] K8 of which
// columns ycol0 .. ycol0 + K are used (K = 64 or 256).  A workgroup = 8 waves owns ALL of dW for one point range, so
// every operand byte is read once per launch: wave (wm, wn) computes rows 64 wm .. +64, columns (K / 2) wn .. of dW.
// Both MFMA operands are one 16-byte K8 unit per lane straight from global memory (see the header of this file).
struct BfDwJob {
  const bfraw* X[2];
  const bfraw* Y[2];
  int Cy[2], ycol0[2];
  int npairs, K, lddw, bias_pair;
  float* dW;        // fp32 [256 x lddw]
  float* db;        // fp32 [256] or nullptr
  float* part;      // deterministic: [splits][256][lddw] slabs (or nullptr: fp32 atomics)
  float* partb;     // deterministic: [splits][256]
};
constexpr int kMaxBfDwJobs = 16;
struct BfDwGroup {
  BfDwJob job[kMaxBfDwJobs];
  int njobs, splits;
  int64_t M, rows_per_split;
};

template <int KW>   // columns of dW per wave: 128 (K = 256) or 32 (K = 64)
__device__ inline void bf_dw_job(const BfDwJob& J, int64_t m_begin, int64_t m_end, int64_t M, int split, int lane, int wave) {
  constexpr int TN = KW / 32;
  const int wm = wave >> 1, wn = wave & 1;
  const int i = lane & 31, h = lane >> 5;
  v16f acc[2][TN];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float bs[2] = {0.f, 0.f};
  const bool bias_wave = J.db != nullptr && wn == 0;
  for (int pi = 0; pi < J.npairs; ++pi) {
    const bfraw* Xb = J.X[pi] + ((size_t)(m_begin >> 3) * FH + wm * 64 + i) * 8;
    const bfraw* Yb = J.Y[pi] + ((size_t)(m_begin >> 3) * J.Cy[pi] + J.ycol0[pi] + wn * KW + i) * 8;
    const size_t xs = (size_t)FH * 8, ys = (size_t)J.Cy[pi] * 8;   // elements per 8-point block
    const bool do_bias = bias_wave && pi == J.bias_pair;
    const int64_t nsteps = (m_end - m_begin + 15) / 16;
    // 3-slot register ring: the loads of step s + 2 are issued before the MFMAs of step s (HBM latency x bandwidth per
    // CU is ~40 KB; one step of one workgroup is 16 KB of operands, two workgroup-steps are in flight per CU)
    vu4 a[3][2], b[3][TN];
    auto load = [&](int slot, int64_t s) {
      const int64_t sc = s < nsteps ? s : nsteps - 1;   // past the end: a harmless re-load of the last step
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) a[slot][ti] = *reinterpret_cast<const vu4*>(Xb + (2 * sc + h) * xs + ti * 32 * 8);
#pragma unroll
      for (int tj = 0; tj < TN; ++tj) b[slot][tj] = *reinterpret_cast<const vu4*>(Yb + (2 * sc + h) * ys + tj * 32 * 8);
    };
    auto compute = [&](int slot, int64_t s) {
      const int64_t m0 = m_begin + s * 16;
      if (m0 + 16 > M) {   // ragged tail: points >= M contribute nothing (their saved state is padding)
        const int64_t first = m0 + 8 * h;
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
          unsigned w[4] = {a[slot][ti].x, a[slot][ti].y, a[slot][ti].z, a[slot][ti].w};
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (first + j >= M) w[j >> 1] &= (j & 1) ? 0x0000ffffu : 0xffff0000u;
          a[slot][ti] = vu4{w[0], w[1], w[2], w[3]};
        }
      }
#pragma unroll
      for (int tj = 0; tj < TN; ++tj)
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
          acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a[slot][ti]),
                                                                __builtin_bit_cast(bf8, b[slot][tj]), acc[ti][tj], 0, 0, 0);
      if (do_bias) {
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
          const vu4 v = a[slot][ti];
          bs[ti] += (bf_lo(v.x) + bf_hi(v.x)) + (bf_lo(v.y) + bf_hi(v.y)) + (bf_lo(v.z) + bf_hi(v.z)) + (bf_lo(v.w) + bf_hi(v.w));
        }
      }
    };
    load(0, 0);
    load(1, 1);
    for (int64_t s = 0; s < nsteps; s += 3) {
      load(2, s + 2);
      compute(0, s);
      if (s + 1 < nsteps) { load(0, s + 3); compute(1, s + 1); }
      if (s + 2 < nsteps) { load(1, s + 4); compute(2, s + 2); }
    }
  }
  // accumulator (ti, tj, r) of lane (i, h) is dW[64 wm + 32 ti + (r & 3) + 8 (r >> 2) + 4 h][KW wn + 32 tj + i]
  const int lddw = J.lddw;
  float* __restrict__ pdst = J.part ? J.part + (size_t)split * FH * lddw : nullptr;
#pragma unroll
  for (int tj = 0; tj < TN; ++tj) {
    const int col = wn * KW + tj * 32 + i;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (pdst) pdst[(size_t)row * lddw + col] = acc[ti][tj][r];
        else atomicAdd(J.dW + (size_t)row * lddw + col, acc[ti][tj][r]);
      }
  }
  if (bias_wave) {
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
      const float t = bs[ti] + __shfl_xor(bs[ti], 32, 64);
      if (h == 0) {
        const int row = wm * 64 + ti * 32 + i;
        if (J.partb) J.partb[(size_t)split * FH + row] = t;
        else atomicAdd(J.db + row, t);
      }
    }
  }
}

// K = 256 jobs: the two operand chunks of 32 points (X 16 KB + Y 16 KB, K8 units in global order) are staged in LDS by
// LDS-DMA (global_load_lds_dwordx4: 16 bytes per lane, no VGPR round trip), three chunks deep, so every operand byte is
// fetched ONCE per workgroup (the register-direct form above lets the two / four waves that share a fragment each
// fetch it: measured 1.7x the unique bytes at the memory side).  One raw barrier per chunk; the DMAs of the next two
// chunks stay in flight across it (counted vmcnt, never 0 inside the loop).
constexpr int kDwChunk = 32;                            // points per chunk
constexpr int kDwOpBytes = (kDwChunk / 8) * FH * 16;    // bytes of one operand chunk (4 blocks x 256 units x 16 B)
constexpr int kDwBufs = 3;

__device__ inline void dw_issue_chunk(const bfraw* __restrict__ Xg, const bfraw* __restrict__ Yg, int64_t chunk,
                                      int64_t nchunks, int CyUnits, char* lds_buf, int wave, int lane) {
  // this wave's share: units [wave * 128, wave * 128 + 128) of each operand chunk = 2 DMA instructions per operand.
  // X chunk: blocks 4 chunk .. +3, all 256 columns: contiguous in global.  Y chunk: 256 of the Cy columns per block.
  const int64_t c = chunk < nchunks ? chunk : nchunks - 1;   // past the end: harmless re-fetch, keeps vmcnt uniform
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int u = wave * 128 + q * 64;     // first unit of this instruction (wave-uniform); blk = u / 256, col = u % 256
    const int blk = u >> 8, col = (u & 255) + lane;
    const bfraw* xs = Xg + ((size_t)(c * 4 + blk) * FH + col) * 8;
    const bfraw* ys = Yg + ((size_t)(c * 4 + blk) * CyUnits + col) * 8;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)xs,
                                     (__attribute__((address_space(3))) void*)(lds_buf + u * 16), 16, 0, RNB_AUX_LD);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ys,
                                     (__attribute__((address_space(3))) void*)(lds_buf + kDwOpBytes + u * 16), 16, 0, RNB_AUX_LD);
  }
}

__device__ inline void bf_dw_job_lds(const BfDwJob& J, int64_t m_begin, int64_t m_end, int64_t M, int split, int lane, int wave,
                                     char* lds) {
  constexpr int TN = 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int i = lane & 31, h = lane >> 5;
  v16f acc[2][TN];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float bs[2] = {0.f, 0.f};
  const bool bias_wave = J.db != nullptr && wn == 0;
  const int64_t nchunks = (m_end - m_begin + kDwChunk - 1) / kDwChunk;   // (ranges are multiples of 64 points)
  for (int pi = 0; pi < J.npairs; ++pi) {
    const bfraw* Xg = J.X[pi] + (size_t)(m_begin >> 3) * FH * 8;
    const bfraw* Yg = J.Y[pi] + ((size_t)(m_begin >> 3) * J.Cy[pi] + J.ycol0[pi]) * 8;
    const bool do_bias = bias_wave && pi == J.bias_pair;
    __builtin_amdgcn_s_barrier();   // every wave is done with the buffers of the previous pair
    dw_issue_chunk(Xg, Yg, 0, nchunks, J.Cy[pi], lds, wave, lane);
    dw_issue_chunk(Xg, Yg, 1, nchunks, J.Cy[pi], lds + 2 * kDwOpBytes, wave, lane);
    for (int64_t c = 0; c < nchunks; ++c) {
      // chunk c has landed (this wave's 4 DMAs of chunk c + 1 may still be in flight) ...
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // ... for every wave; and every wave has finished reading chunk c - 1
      dw_issue_chunk(Xg, Yg, c + 2, nchunks, J.Cy[pi], lds + ((c + 2) % kDwBufs) * 2 * kDwOpBytes, wave, lane);
      const char* bx = lds + (c % kDwBufs) * 2 * kDwOpBytes;
      const char* by = bx + kDwOpBytes;
#pragma unroll
      for (int st = 0; st < 2; ++st) {   // two 16-point MFMA steps per chunk
        vu4 a[2], b[TN];
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
          a[ti] = *reinterpret_cast<const vu4*>(bx + ((2 * st + h) * FH + wm * 64 + ti * 32 + i) * 16);
#pragma unroll
        for (int tj = 0; tj < TN; ++tj)
          b[tj] = *reinterpret_cast<const vu4*>(by + ((2 * st + h) * FH + wn * 128 + tj * 32 + i) * 16);
        const int64_t m0 = m_begin + c * kDwChunk + st * 16;
        if (m0 + 16 > M) {   // ragged tail: points >= M contribute nothing (their saved state is padding)
          const int64_t first = m0 + 8 * h;
#pragma unroll
          for (int ti = 0; ti < 2; ++ti) {
            unsigned w[4] = {a[ti].x, a[ti].y, a[ti].z, a[ti].w};
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (first + j >= M) w[j >> 1] &= (j & 1) ? 0x0000ffffu : 0xffff0000u;
            a[ti] = vu4{w[0], w[1], w[2], w[3]};
          }
        }
#pragma unroll
        for (int tj = 0; tj < TN; ++tj)
#pragma unroll
          for (int ti = 0; ti < 2; ++ti)
            acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a[ti]), __builtin_bit_cast(bf8, b[tj]),
                                                                  acc[ti][tj], 0, 0, 0);
        if (do_bias) {
#pragma unroll
          for (int ti = 0; ti < 2; ++ti) {
            const vu4 v = a[ti];
            bs[ti] += (bf_lo(v.x) + bf_hi(v.x)) + (bf_lo(v.y) + bf_hi(v.y)) + (bf_lo(v.z) + bf_hi(v.z)) + (bf_lo(v.w) + bf_hi(v.w));
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS reads of chunk c are complete
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the two over-fetched chunks before the buffers are reused
  }
  const int lddw = J.lddw;
  float* __restrict__ pdst = J.part ? J.part + (size_t)split * FH * lddw : nullptr;
#pragma unroll
  for (int tj = 0; tj < TN; ++tj) {
    const int col = wn * 128 + tj * 32 + i;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (pdst) pdst[(size_t)row * lddw + col] = acc[ti][tj][r];
        else atomicAdd(J.dW + (size_t)row * lddw + col, acc[ti][tj][r]);
      }
  }
  if (bias_wave) {
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
      const float t = bs[ti] + __shfl_xor(bs[ti], 32, 64);
      if (h == 0) {
        const int row = wm * 64 + ti * 32 + i;
        if (J.partb) J.partb[(size_t)split * FH + row] = t;
        else atomicAdd(J.db + row, t);
      }
    }
  }
}

__global__ __launch_bounds__(512, 1) void bf_dw_kernel(const BfDwGroup g) {
  __shared__ __attribute__((aligned(16))) char lds[kDwBufs * 2 * kDwOpBytes];   // 96 KB: the ONLY shared object
  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int ji = blockIdx.x / g.splits, split = blockIdx.x - ji * g.splits;
  const BfDwJob& J = g.job[ji];
  const int64_t m_begin = (int64_t)split * g.rows_per_split;
  const int64_t m_end = m_begin + g.rows_per_split < g.M ? m_begin + g.rows_per_split : g.M;
  if (m_begin >= m_end) return;
  if (J.K == 256) bf_dw_job_lds(J, m_begin, m_end, g.M, split, lane, wave, lds);
  else bf_dw_job<32>(J, m_begin, m_end, g.M, split, lane, wave);
}

// deterministic variant: ordered reduction of the slabs
__global__ __launch_bounds__(256) void bf_dw_reduce_kernel(const BfDwGroup g) {
  const BfDwJob& J = g.job[blockIdx.y];
  const size_t n = (size_t)FH * J.lddw;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (size_t)gridDim.x * 256) {
    if ((int)(idx % J.lddw) >= J.K) continue;
    double s = 0.0;
    for (int sp = 0; sp < g.splits; ++sp) s += (double)J.part[(size_t)sp * n + idx];
    J.dW[idx] = (float)s;
  }
  if (J.db != nullptr && J.partb != nullptr) {
    for (int r = blockIdx.x * 256 + threadIdx.x; r < FH; r += gridDim.x * 256) {
      double s = 0.0;
      for (int sp = 0; sp < g.splits; ++sp) s += (double)J.partb[(size_t)sp * FH + r];
      J.db[r] = (float)s;
    }
  }
}

// gradient of the sdf-head row: dw_sdf[k] += sum_rows (sbar / scale * a_last + u_last), db_sdf += sum sbar / scale.
// One thread per column and point slab; K8 units (8 points of one column) per load.  One slab per column chunk in the
// deterministic variant (a single add onto zero per address).
__global__ __launch_bounds__(1024) void bf_sdf_head_bwd_kernel(const bfraw* __restrict__ a, const bfraw* __restrict__ ulast,
                                                               const float* __restrict__ sbar, float inv_scale, int64_t M,
                                                               int64_t rows_per_blk, float* __restrict__ dwsdf,
                                                               float* __restrict__ dbsdf) {
  __shared__ double red[4][FH], redb[4];
  const int c = threadIdx.x & 255, ph = threadIdx.x >> 8;   // column, one of 4 row phases (8-point blocks ph, ph + 4, ...)
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk;
  const int64_t r1 = r0 + rows_per_blk < M ? r0 + rows_per_blk : M;
  double s = 0.0, sb = 0.0;
  for (int64_t r = r0 + 8 * ph; r < r1; r += 32) {
    const vu4 av = *reinterpret_cast<const vu4*>(a + ((size_t)(r >> 3) * FH + c) * 8);
    const vu4 uv = *reinterpret_cast<const vu4*>(ulast + ((size_t)(r >> 3) * FH + c) * 8);
    const float af[8] = {bf_lo(av.x), bf_hi(av.x), bf_lo(av.y), bf_hi(av.y), bf_lo(av.z), bf_hi(av.z), bf_lo(av.w), bf_hi(av.w)};
    const float uf[8] = {bf_lo(uv.x), bf_hi(uv.x), bf_lo(uv.y), bf_hi(uv.y), bf_lo(uv.z), bf_hi(uv.z), bf_lo(uv.w), bf_hi(uv.w)};
    float t = 0.f, tb = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (r + j < r1) {
        const float w = sbar[r + j] * inv_scale;
        t += fmaf(w, af[j], uf[j]);
        tb += w;
      }
    }
    s += (double)t;
    sb += (double)tb;
  }
  red[ph][c] = s;
  if (c == 0) redb[ph] = sb;
  __syncthreads();
  if (ph == 0) {
    atomicAdd(dwsdf + c, (float)(red[0][c] + red[1][c] + red[2][c] + red[3][c]));
    if (c == 0) atomicAdd(dbsdf, (float)(redb[0] + redb[1] + redb[2] + redb[3]));
  }
}

// fp32 packed weights -> bf16 mirror: every matrix that serves as an MFMA B operand, at its own element offset, in
// fragment order (see bf_load_b).  One workgroup per 32-row x 16-k fragment... one thread per 16-byte unit.
struct BfPackEntry { long long off; int N, K; int unit_begin; };
constexpr int kMaxPack = 4 * RNB_MAX_LIN + 4;
struct BfPackTable { int n, total_units; BfPackEntry e[kMaxPack]; };
__global__ void bf_pack_kernel(const float* __restrict__ src, BfPackTable t, bfraw* __restrict__ dst) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= t.total_units) return;
  int ei = 0;
  while (ei + 1 < t.n && u >= t.e[ei + 1].unit_begin) ++ei;
  const BfPackEntry en = t.e[ei];
  const int lu = u - en.unit_begin;            // unit inside the matrix: fragment lu / 64, lane lu % 64
  const int frag = lu >> 6, lane = lu & 63;
  const int nks = en.K >> 4;
  const int nt = frag / nks, ks = frag - nt * nks;
  const int c = lane & 31, h = lane >> 5;
  const float* sp = src + en.off + (size_t)(nt * 32 + c) * en.K + ks * 16 + h * 8;
  const vu4 o = {pack2(sp[0], sp[1]), pack2(sp[2], sp[3]), pack2(sp[4], sp[5]), pack2(sp[6], sp[7])};
  *reinterpret_cast<vu4*>(dst + en.off + (size_t)lu * 8) = o;
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
// the albedo network runs in bf16 too when it has the shipped shape; otherwise its fp32 kernels (mlp.hip) are used
bool bf16_color_supported(const Layout& L) {
  if (L.F != FH || L.Hc != FH || L.Hcp != FH) return false;
  if (L.Cinp > CMAX || L.Cinp % 64 != 0 || L.Cinp - L.F > 64) return false;
  if (L.nc < 1 || L.Co < 1 || L.Co > 4) return false;
  return true;
}

static void fill_col(const Layout& L, const float* packed, const float* pts, PointBufs& pb, BfColArgs& g) {
  memset(&g, 0, sizeof(g));
  g.pts = pts;
  g.nrm = pb.nrm;
  g.M = pb.M;
  g.packed = packed;
  g.wbf = reinterpret_cast<const bfraw*>(packed + L.total);
  g.nc = L.nc; g.F = L.F; g.pev = L.pev; g.multires_view = L.multires_view; g.Cinp = L.Cinp; g.Co = L.Co;
  g.squeeze = L.squeeze;
  for (int l = 0; l < L.nc; ++l) {
    g.Kp[l] = L.col[l].Kp;
    g.w_off[l] = L.col[l].w_off;
    g.wT_off[l] = L.col[l].wT_off;
    g.b_off[l] = L.col[l].b_off;
    g.ac8[l] = reinterpret_cast<bfraw*>(pb.ac8[l]);
    g.zc8[l] = reinterpret_cast<bfraw*>(pb.zc8[l]);
  }
  g.wo_off = L.colo.w_off;
  g.bo_off = L.colo.b_off;
  g.ldwo = L.colo.Kp;
  g.cin8 = reinterpret_cast<bfraw*>(pb.cin8);
  g.alb = pb.alb;
  g.albbar = pb.albbar;
  g.fbar8 = reinterpret_cast<bfraw*>(pb.fbar_k8);
  g.cinb = pb.cinb;
}

static double color_flops(const Layout& L, int64_t M, int first) {
  double fl = 0;
  for (int l = first; l < L.nc; ++l) fl += 2.0 * (double)M * L.col[l].N * L.col[l].K;
  return fl;
}

// C: albedo network forward on the tile state the F and R sweeps left (cin8 features, pb.nrm)
int bf16_color_forward(const Layout& L, const float* packed, PointBufs& pb, const float* pts, hipStream_t s) {
  BfColArgs g;
  fill_col(L, packed, pts, pb, g);
  ProfScope prof(color_flops(L, pb.M, 0) + 2.0 * (double)pb.M * L.colo.N * L.colo.K, s, "albedo_fwd");
  hipLaunchKernelGGL(bf_color_fwd_kernel, dim3((unsigned)(pb.Mp / BT)), dim3(256), 0, s, g);
  RNB_CHECK_LAUNCH();
  return RNB_OK;
}

// C': albedo network backward (writes zc8, fbar8, the pe columns of cinb, dWo / dbo); its hidden-layer weight
// gradients join the grouped dW launch of bf16_backward
int bf16_color_backward(const Layout& L, const float* packed, PointBufs& pb, float* packed_grad, hipStream_t s) {
  BfColArgs g;
  fill_col(L, packed, nullptr, pb, g);
  const bool det = (L.variant & RNB_VARIANT_DETERMINISTIC) != 0;
  {
    ProfScope prof(color_flops(L, pb.M, 0) + 2.0 * (double)pb.M * L.colo.N * L.colo.K, s, "albedo_bwd");
    hipLaunchKernelGGL(bf_color_bwd_kernel, dim3((unsigned)(pb.Mp / BT)), dim3(256), 0, s, g);
    RNB_CHECK_LAUNCH();
  }
  int64_t slabs = det ? 1 : 256;
  int64_t rows_per_blk = (pb.M + slabs - 1) / slabs;
  rows_per_blk = (rows_per_blk + 7) / 8 * 8;
  hipLaunchKernelGGL(bf_color_out_bwd_kernel, dim3((unsigned)((pb.M + rows_per_blk - 1) / rows_per_blk)), dim3(1024), 0, s,
                     reinterpret_cast<const bfraw*>(pb.ac8[L.nc - 1]), pb.alb, pb.albbar, L.Co, L.squeeze, pb.M, rows_per_blk,
                     L.colo.Kp, packed_grad + L.colo.w_off, packed_grad + L.colo.b_off);
  RNB_CHECK_LAUNCH();
  return RNB_OK;
}

int bf16_pack_weights(const Layout& L, float* packed, hipStream_t s) {
  bfraw* dst = reinterpret_cast<bfraw*>(packed + L.total);
  BfPackTable t;
  t.n = 0;
  t.total_units = 0;
  auto add = [&](long long off, int N, int K) {
    if (off < 0 || N <= 0 || K <= 0) return;
    BfPackEntry& e = t.e[t.n++];
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
    for (int l = 0; l < L.nc; ++l) {
      add(L.col[l].w_off, L.col[l].Np, L.col[l].Kp);
      add(L.col[l].wT_off, L.col[l].Kp, L.col[l].Np);
    }
  }
  hipLaunchKernelGGL(bf_pack_kernel, dim3((unsigned)((t.total_units + 255) / 256)), dim3(256), 0, s, packed, t, dst);
  RNB_CHECK_LAUNCH();
  return RNB_OK;
}

// rows per wave of the sweeps: RNB_VARIANT_{FWD,BWD}_TI (1: 32 rows x 8 waves, 2: 64 rows x 4 waves).  Default 2:
// measured 4.23 ms / step against 5.04 ms with TI = 1 on both (512 rays x 256 samples) — the 8-wave form doubles the
// weight bytes each CU pulls from L2 per point (every fragment feeds one row tile instead of two), and that stream,
// not latency, is what these sweeps wait for.
static int bf_ti(const Layout& L, int shift) {
  const int v = L.knob(shift);
  return v == 1 ? 1 : 2;
}

static const bfraw* wbf_of(const Layout& L, const float* packed) { return reinterpret_cast<const bfraw*>(packed + L.total); }

static double hidden_flops_bf(const Layout& L, int64_t M, int first) {
  double fl = 0;
  for (int l = first; l < L.nh; ++l) fl += 2.0 * (double)M * L.hid[l].N * L.hid[l].K;
  return fl;
}

int bf16_forward(const Layout& L, const float* packed, const float* pts, int64_t M, PointBufs& pb, bool save, bool need_feat,
                 hipStream_t s, const GridGen* grid, bool feat_k8) {
  BfFwdArgs g;
  memset(&g, 0, sizeof(g));
  if (grid) g.grid = *grid;
  g.pts = pts;
  g.M = M;
  g.packed = packed;
  g.wbf = wbf_of(L, packed);
  g.nh = L.nh; g.skip = L.skip; g.pe = L.pe; g.multires = L.multires; g.Ep = L.Ep;
  g.scale = L.sdf_scale;
  for (int l = 0; l < L.nh; ++l) {
    g.n_real[l] = L.hid[l].N;
    g.Kp[l] = L.hid[l].Kp;
    g.w_off[l] = L.hid[l].w_off;
    g.b_off[l] = L.hid[l].b_off;
    g.a[l] = reinterpret_cast<bfraw*>(pb.a[l]);
    g.D[l] = reinterpret_cast<bfraw*>(pb.D[l]);
  }
  g.wsdf_off = L.wsdf_off;
  g.bsdf_off = L.bsdf_off;
  g.with_feat = need_feat ? 1 : 0;
  g.F = L.F;
  g.Cinp = L.Cinp;
  g.wf_off = L.feat.w_off;
  g.bf_off = L.feat.b_off;
  g.cin = pb.cin;
  g.cin8 = feat_k8 ? reinterpret_cast<bfraw*>(pb.cin8) : nullptr;
  g.sdf = pb.sdf;
  g.x4 = pb.x;
  g.e = reinterpret_cast<bfraw*>(pb.e);
  double fl = hidden_flops_bf(L, M, 0) + 2.0 * (double)M * L.H;
  if (need_feat) fl += 2.0 * (double)M * L.F * L.H;
  ProfScope prof(fl, s, save ? "F_sweep(save)" : "F_sweep(forward_only)");
  const unsigned blocks = (unsigned)(pb.Mp / BT);
  const int ti = bf_ti(L, RNB_VARIANT_FWD_TI_SHIFT);
  if (save && ti == 1) hipLaunchKernelGGL((bf_forward_kernel<true, 1>), dim3(blocks), dim3(512), 0, s, g);
  else if (save) hipLaunchKernelGGL((bf_forward_kernel<true, 2>), dim3(blocks), dim3(256), 0, s, g);
  else if (ti == 1) hipLaunchKernelGGL((bf_forward_kernel<false, 1>), dim3(blocks), dim3(512), 0, s, g);
  else hipLaunchKernelGGL((bf_forward_kernel<false, 2>), dim3(blocks), dim3(256), 0, s, g);
  RNB_CHECK_LAUNCH();
  return RNB_OK;
}

static void fill_bwd(const Layout& L, const float* packed, PointBufs& pb, BfBwdArgs& g) {
  memset(&g, 0, sizeof(g));
  g.packed = packed;
  g.wbf = wbf_of(L, packed);
  g.M = pb.M;
  g.nh = L.nh; g.skip = L.skip; g.pe = L.pe; g.multires = L.multires; g.Ep = L.Ep;
  g.inv_scale = 1.f / L.sdf_scale;
  for (int l = 0; l < L.nh; ++l) {
    g.n_real[l] = L.hid[l].N;
    g.Kp[l] = L.hid[l].Kp;
    g.w_off[l] = L.hid[l].w_off;
    g.wT_off[l] = L.hid[l].wT_off;
    g.D[l] = reinterpret_cast<bfraw*>(pb.D[l]);
    g.gz[l] = reinterpret_cast<bfraw*>(pb.gz[l]);
    g.zR[l] = reinterpret_cast<bfraw*>(pb.zR[l]);
    g.zb[l] = reinterpret_cast<bfraw*>(pb.zb[l]);
  }
  g.u[0] = reinterpret_cast<bfraw*>(pb.u0_k8);
  for (int l = 1; l <= L.nh; ++l) g.u[l] = reinterpret_cast<bfraw*>(pb.u[l]);
  g.fbar8 = reinterpret_cast<bfraw*>(pb.fbar_k8);
  g.wsdf_off = L.wsdf_off;
  g.wfT_off = L.feat.wT_off;
  g.x4 = pb.x;
  g.nrm = pb.nrm;
  g.geb = pb.geb;
  g.sbar = pb.sbar;
}

int bf16_reverse(const Layout& L, const float* packed, PointBufs& pb, hipStream_t s) {
  BfBwdArgs g;
  fill_bwd(L, packed, pb, g);
  ProfScope prof(hidden_flops_bf(L, pb.M, 0), s, "R_sweep");
  if (bf_ti(L, RNB_VARIANT_BWD_TI_SHIFT) == 1) hipLaunchKernelGGL(bf_reverse_kernel<1>, dim3((unsigned)(pb.Mp / BT)), dim3(512), 0, s, g);
  else hipLaunchKernelGGL(bf_reverse_kernel<2>, dim3((unsigned)(pb.Mp / BT)), dim3(256), 0, s, g);
  RNB_CHECK_LAUNCH();
  return RNB_OK;
}

// RA, sdf-head row gradient, FB and every dW job of the SDF network (+ the feature head's) for one backward
int bf16_backward(const Layout& L, const float* packed, PointBufs& pb, bool with_color, bool color_bf16, float* packed_grad,
                  hipStream_t s) {
  const int64_t M = pb.M;
  const bool det = (L.variant & RNB_VARIANT_DETERMINISTIC) != 0;
  BfBwdArgs g;
  fill_bwd(L, packed, pb, g);
  const unsigned blocks = (unsigned)(pb.Mp / BT);
  const int bti = bf_ti(L, RNB_VARIANT_BWD_TI_SHIFT);
  {
    ProfScope prof(hidden_flops_bf(L, M, 0), s, "RA_sweep");
    if (bti == 1) hipLaunchKernelGGL(bf_ra_kernel<1>, dim3(blocks), dim3(512), 0, s, g);
    else hipLaunchKernelGGL(bf_ra_kernel<2>, dim3(blocks), dim3(256), 0, s, g);
    RNB_CHECK_LAUNCH();
  }
  {
    int64_t slabs = det ? 1 : 256;
    int64_t rows_per_blk = (M + slabs - 1) / slabs;
    rows_per_blk = (rows_per_blk + 7) / 8 * 8;
    hipLaunchKernelGGL(bf_sdf_head_bwd_kernel, dim3((unsigned)((M + rows_per_blk - 1) / rows_per_blk)), dim3(1024), 0, s,
                       reinterpret_cast<const bfraw*>(pb.a[L.nh - 1]), reinterpret_cast<const bfraw*>(pb.u[L.nh]), pb.sbar,
                       1.f / L.sdf_scale, M, rows_per_blk, packed_grad + L.wsdf_off, packed_grad + L.bsdf_off);
    RNB_CHECK_LAUNCH();
  }
  {
    g.fbar = (with_color && !color_bf16) ? pb.cinb : nullptr;
    g.ld_fbar = L.Cinp;
    g.fbar_in_k8 = (with_color && color_bf16) ? 1 : 0;
    ProfScope prof(hidden_flops_bf(L, M, 1) + (with_color ? 2.0 * (double)M * L.F * L.H : 0.0), s, "FB_sweep");
    if (bti == 1) hipLaunchKernelGGL(bf_fb_kernel<1>, dim3(blocks), dim3(512), 0, s, g);
    else hipLaunchKernelGGL(bf_fb_kernel<2>, dim3(blocks), dim3(256), 0, s, g);
    RNB_CHECK_LAUNCH();
  }
  // ---- dW jobs ----------------------------------------------------------------------------------------------
  BfDwGroup grp;
  memset(&grp, 0, sizeof(grp));
  grp.M = M;
  // points per workgroup: enough workgroups to fill the chip (njobs x splits >= ~2 per CU), ranges a multiple of 16
  const int njobs_est = L.nh + (with_color ? 1 : 0) + ((with_color && color_bf16) ? L.nc + 1 : 0);
  int splits = (int)((512 + njobs_est - 1) / njobs_est);
  int64_t rows = (M + splits - 1) / splits;
  rows = (rows + 63) / 64 * 64;
  splits = (int)((M + rows - 1) / rows);
  grp.splits = splits;
  grp.rows_per_split = rows;
  // (the staged fp32 kernel's slabs at the tail of the workspace are free again: its reduction was enqueued earlier)
  float* part = det ? pb.dw_part : nullptr;
  int64_t part_left = det ? pb.dw_part_floats : 0;
  double fl = 0;
  auto add = [&](const bfraw* X1, const bfraw* Y1, int Cy1, const bfraw* X2, const bfraw* Y2, int Cy2, int npairs, int K,
                 const Lin& ln, int bias_pair, double f, int ycol0 = 0, bool with_bias = true) -> int {
    if (grp.njobs == kMaxBfDwJobs) RNB_FAIL(RNB_E_INVALID, "too many weight-gradient jobs for one bf16 launch");
    BfDwJob& J = grp.job[grp.njobs++];
    J.X[0] = X1; J.Y[0] = Y1; J.Cy[0] = Cy1; J.ycol0[0] = ycol0;
    J.X[1] = X2; J.Y[1] = Y2; J.Cy[1] = Cy2; J.ycol0[1] = ycol0;
    J.npairs = npairs; J.K = K; J.lddw = ln.Kp; J.bias_pair = bias_pair;
    J.dW = packed_grad + ln.w_off + ycol0;      // a column range [ycol0, ycol0 + K) of the layer's [256 x Kp] gradient
    J.db = with_bias ? packed_grad + ln.b_off : nullptr;
    J.part = nullptr; J.partb = nullptr;
    if (det) {
      const int64_t need = (int64_t)splits * FH * ln.Kp + (int64_t)splits * FH;
      if (need > part_left) RNB_FAIL(RNB_E_WORKSPACE, "deterministic bf16 dW: partial-slab workspace exhausted");
      J.part = part; J.partb = part + (int64_t)splits * FH * ln.Kp;
      part += need; part_left -= need;
    }
    fl += f;
    return RNB_OK;
  };
  if (det) RNB_CHECK_HIP(hipMemsetAsync(pb.dw_part, 0, (size_t)pb.dw_part_floats * sizeof(float), s));
  for (int l = 0; l < L.nh; ++l) {
    const Lin& ln = L.hid[l];
    const bfraw* in = l == 0 ? reinterpret_cast<const bfraw*>(pb.e) : reinterpret_cast<const bfraw*>(pb.a[l - 1]);
    const bfraw* uin = l == 0 ? reinterpret_cast<const bfraw*>(pb.u0_k8) : reinterpret_cast<const bfraw*>(pb.u[l]);
    const int Cy = l == 0 ? L.Ep : FH;
    RNB_TRY(add(reinterpret_cast<const bfraw*>(pb.gz[l]), uin, Cy, reinterpret_cast<const bfraw*>(pb.zb[l]), in, Cy, 2, ln.Kp,
                ln, 1, 4.0 * (double)M * ln.N * ln.K));
  }
  if (with_color) {
    RNB_TRY(add(reinterpret_cast<const bfraw*>(pb.fbar_k8), reinterpret_cast<const bfraw*>(pb.a[L.nh - 1]), FH, nullptr,
                nullptr, 0, 1, L.feat.Kp, L.feat, 0, 2.0 * (double)M * L.feat.N * L.feat.K));
    if (color_bf16) {   // the albedo net's hidden layers: dW_l = zc_l^T in_l
      for (int l = L.nc - 1; l >= 1; --l)
        RNB_TRY(add(reinterpret_cast<const bfraw*>(pb.zc8[l]), reinterpret_cast<const bfraw*>(pb.ac8[l - 1]), FH, nullptr, nullptr,
                    0, 1, L.col[l].Kp, L.col[l], 0, 2.0 * (double)M * L.col[l].N * L.col[l].K));
      // layer 0 reads the Cinp-wide input: its 256 feature columns and its 64 pe columns are two jobs
      RNB_TRY(add(reinterpret_cast<const bfraw*>(pb.zc8[0]), reinterpret_cast<const bfraw*>(pb.cin8), L.Cinp, nullptr, nullptr, 0, 1,
                  FH, L.col[0], 0, 2.0 * (double)M * L.col[0].N * L.col[0].K));
      RNB_TRY(add(reinterpret_cast<const bfraw*>(pb.zc8[0]), reinterpret_cast<const bfraw*>(pb.cin8), L.Cinp, nullptr, nullptr, 0, 1,
                  L.Cinp - FH, L.col[0], 0, 0.0, FH, false));
    }
  }
  {
    ProfScope prof(fl, s, "dW(all)");
    hipLaunchKernelGGL(bf_dw_kernel, dim3((unsigned)(grp.njobs * splits)), dim3(512), 0, s, grp);
    RNB_CHECK_LAUNCH();
    if (det) {
      hipLaunchKernelGGL(bf_dw_reduce_kernel, dim3(64, grp.njobs), dim3(256), 0, s, grp);
      RNB_CHECK_LAUNCH();
    }
  }
  return RNB_OK;
}

// floats of deterministic partial-slab workspace for bf16_backward over M points
int64_t bf16_dw_partial_floats(const Layout& L, int64_t M, bool with_color) {
  const bool cbf = with_color && bf16_color_supported(L);
  const int njobs_est = L.nh + (with_color ? 1 : 0) + (cbf ? L.nc + 1 : 0);
  int splits = (int)((512 + njobs_est - 1) / njobs_est);
  int64_t rows = (M + splits - 1) / splits;
  rows = (rows + 63) / 64 * 64;
  splits = (int)((M + rows - 1) / rows);
  int64_t total = 0;
  for (int l = 0; l < L.nh; ++l) total += (int64_t)splits * FH * L.hid[l].Kp + (int64_t)splits * FH;
  if (with_color) total += (int64_t)splits * FH * L.feat.Kp + (int64_t)splits * FH;
  if (cbf) {
    for (int l = 1; l < L.nc; ++l) total += (int64_t)splits * FH * L.col[l].Kp + (int64_t)splits * FH;
    total += 2 * ((int64_t)splits * FH * L.col[0].Kp + (int64_t)splits * FH);
  }
  return total;
}

}  // namespace rnb
