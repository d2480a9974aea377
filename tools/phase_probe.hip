// Probe (development aid, round 5): the compute-only floor of a forward-type sweep's layer, by arrangement of its two phases.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/phase_probe.hip -o tools/phase_probe && tools/phase_probe
// One "layer" of a wave = M: 192 v_mfma_f32_32x32x16_f16 (64 x 64 outputs, K = 256, three terms) with SPLIT vector instructions
// between them (the operand split of the fp32 tile: 512 of v_cvt_pk_f16_f32 / v_fma_mix_f32, or none for a pre-split tile), then
// E: the epilogue's vector work for 32 value pairs (13 packed + 14 single + 6 transcendental instructions per pair, the static
// mix of fused_forward_kernel<2, true, 4, true, true>).  No memory traffic at all: registers only, every instruction in inline
// asm so that the order is as written.  Arrangements:
//   alt   : M then E, workgroup barrier after each (the shipped structure); 1 or 2 workgroups of 4 waves per CU
//   pp    : "ping-pong in one wave": each phase issues a layer's MFMAs INTERLEAVED with the previous phase's epilogue work
//           (1 MFMA, then its share of the vector instructions), barrier after each phase; one workgroup of 4 waves per CU;
//           a layer of 128 points = two such phases
// Output: clocks per layer and per 128 points (s_memtime of wave 0 of workgroup 0, and wall time).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Regs {
  v16f acc[4];
  h8 ya, yb;
  f2 p[4];
  float s[8];
  float c0, c1;
  unsigned u[2];
};
__device__ inline void init(Regs& r, int lane) {
  for (int a = 0; a < 4; ++a) for (int i = 0; i < 16; ++i) r.acc[a][i] = 0.f;
  for (int j = 0; j < 8; ++j) { r.ya[j] = (_Float16)(0.001f * (lane + j)); r.yb[j] = (_Float16)0.5f; }
  for (int j = 0; j < 4; ++j) r.p[j] = f2{0.5f + 0.001f * lane, 0.25f};
  for (int j = 0; j < 8; ++j) r.s[j] = 0.5f + 0.001f * (lane + j);
  r.c0 = 0.999f; r.c1 = 1e-3f; r.u[0] = 0x3c003c00u; r.u[1] = 0x38003800u;
}
__device__ inline float fold(const Regs& r) {
  float t = 0.f;
  for (int a = 0; a < 4; ++a) for (int i = 0; i < 16; ++i) t += r.acc[a][i];
  for (int j = 0; j < 4; ++j) t += r.p[j].x + r.p[j].y;
  for (int j = 0; j < 8; ++j) t += r.s[j];
  return t + (float)r.u[0] + (float)r.u[1];
}
#define MFMA(a) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(r.acc[a]) : "v"(r.ya), "v"(r.yb))
#define PK(j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r.p[j]) : "v"(r.p[(j + 1) & 3]), "v"(r.p[(j + 2) & 3]))
#define SG(j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r.s[j]) : "v"(r.c0), "v"(r.c1))
#define TR_EXP(j) asm volatile("v_exp_f32 %0, %0" : "+v"(r.s[j]))
#define TR_LOG(j) asm volatile("v_log_f32 %0, %0" : "+v"(r.s[j]))
#define TR_RCP(j) asm volatile("v_rcp_f32 %0, %0" : "+v"(r.s[j]))
#define CVT(j) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r.u[j]) : "v"(r.s[j]), "v"(r.s[j + 2]))
#define MIX(j) asm volatile("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r.s[4 + j]) : "v"(r.s[j]), "v"(r.u[j]))

// the epilogue work of ONE value pair: 13 packed, 14 single, 6 transcendental (exp, log, rcp per value)
__device__ inline void epi_pair(Regs& r) {
  PK(0); PK(1); SG(0); SG(1); TR_EXP(2); PK(2); SG(3); SG(4); TR_EXP(5); PK(3); SG(6); SG(7);
  TR_LOG(2); PK(0); SG(0); SG(1); TR_LOG(5); PK(1); PK(2); SG(3); TR_RCP(2); PK(3); SG(4); SG(6);
  TR_RCP(5); PK(0); PK(1); SG(7); SG(0); PK(2); PK(3); SG(1); PK(0);
}
// eighth of a pair's work (4 of its 33 instructions, transcendentals spread): what fits behind one MFMA when 192 MFMAs carry
// 32 pairs (6 MFMAs per pair -> 5.5 instructions per MFMA); E6 = 6 instructions: 5 or 6 per MFMA alternate below
template <int PART>
__device__ inline void epi_slice(Regs& r) {   // PART 0..5: the six slices of a pair (5 or 6 instructions each: 33 in all)
  if constexpr (PART == 0) { PK(0); PK(1); SG(0); SG(1); TR_EXP(2); PK(2); }
  if constexpr (PART == 1) { SG(3); SG(4); TR_EXP(5); PK(3); SG(6); }
  if constexpr (PART == 2) { SG(7); TR_LOG(2); PK(0); SG(0); SG(1); TR_LOG(5); }
  if constexpr (PART == 3) { PK(1); PK(2); SG(3); TR_RCP(2); PK(3); }
  if constexpr (PART == 4) { SG(4); SG(6); TR_RCP(5); PK(0); PK(1); SG(7); }
  if constexpr (PART == 5) { SG(0); PK(2); PK(3); SG(1); PK(0); }
}
// split work behind one MFMA: SPLIT = 512 instructions per 192 MFMAs: 8 per 3 MFMAs -> 3, 3, 2
template <int N>
__device__ inline void split_n(Regs& r) {
  if constexpr (N >= 1) CVT(0);
  if constexpr (N >= 2) MIX(0);
  if constexpr (N >= 3) CVT(1);
  if constexpr (N >= 4) MIX(1);
}

// M phase: 192 MFMAs; SPLIT: with the operand split between them
template <bool SPLIT>
__device__ inline void phase_m(Regs& r) {
  for (int g = 0; g < 16; ++g) {   // 16 k-steps of 12 MFMAs
#pragma unroll
    for (int m = 0; m < 12; ++m) {
      MFMA(m & 3);
      if constexpr (SPLIT) { if (m % 3 == 2) split_n<2>(r); else split_n<3>(r); }
    }
  }
}
__device__ inline void phase_e(Regs& r) {
  for (int p = 0; p < 32; ++p) epi_pair(r);
}
// ping-pong phase: 192 MFMAs, each followed by its share of the OTHER half's epilogue (and of the split)
template <bool SPLIT>
__device__ inline void phase_pp(Regs& r) {
  for (int g = 0; g < 16; ++g) {   // 12 MFMAs = 2 pairs of epilogue work
#pragma unroll
    for (int m = 0; m < 12; ++m) {
      MFMA(m & 3);
      if constexpr (SPLIT) { if (m % 3 == 2) split_n<2>(r); else split_n<3>(r); }
      if (m % 6 == 0) epi_slice<0>(r);
      if (m % 6 == 1) epi_slice<1>(r);
      if (m % 6 == 2) epi_slice<2>(r);
      if (m % 6 == 3) epi_slice<3>(r);
      if (m % 6 == 4) epi_slice<4>(r);
      if (m % 6 == 5) epi_slice<5>(r);
    }
  }
}
__device__ inline void bar() { asm volatile("s_barrier" ::: "memory"); }

// MODE 0: alt (M, barrier, E, barrier);  1: pp (two ping-pong phases per "layer of 128 points");  2: M only;  3: E only
// PM / PE: wave priority inside M / inside E (alt only)
// STAG: every workgroup draws a ticket from a counter of its CU (HW_ID, XCC_ID); odd ones run one E phase first, so that the
// CU's two workgroups are half a layer apart
__device__ unsigned g_ticket[8 * 256];
template <int MODE, bool SPLIT, int PM, int PE, bool STAG = false>
__global__ __launch_bounds__(256, 2) void layer_kernel(float* out, long long* clk, int layers) {
  Regs r;
  init(r, threadIdx.x & 63);
  if constexpr (STAG) {
    __shared__ int late;
    if (threadIdx.x == 0) {
      const unsigned hw = __builtin_amdgcn_s_getreg((7u << 11) | (8u << 6) | 4u);
      const unsigned xcc = __builtin_amdgcn_s_getreg((3u << 11) | (0u << 6) | 20u);
      late = (int)(atomicAdd(&g_ticket[((xcc & 7u) << 8) | (hw & 255u)], 1u) & 1u);
    }
    __syncthreads();
    if (late) phase_e(r);
  }
  bar();
  const long long t0 = __builtin_readcyclecounter();
  for (int l = 0; l < layers; ++l) {
    if constexpr (MODE == 0) {
      __builtin_amdgcn_s_setprio(PM);
      phase_m<SPLIT>(r);
      bar();
      __builtin_amdgcn_s_setprio(PE);
      phase_e(r);
      bar();
    } else if constexpr (MODE == 1) {
      phase_pp<SPLIT>(r);
      bar();
      phase_pp<SPLIT>(r);
      bar();
    } else if constexpr (MODE == 2) {
      phase_m<SPLIT>(r);
      bar();
    } else {
      phase_e(r);
      bar();
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = fold(r);
}

// same-wave co-issue check: [1 MFMA, NF independent v_fma_f32] repeated; TYPE 0: f16 32x32x16, 1: bf16 32x32x16
typedef short s8v __attribute__((ext_vector_type(8)));
template <int TYPE, int NF>
__global__ __launch_bounds__(256, 2) void fill_kernel(float* out, long long* clk, int iters) {
  Regs r;
  init(r, threadIdx.x & 63);
  s8v ba, bb;
  for (int j = 0; j < 8; ++j) { ba[j] = (short)(0x3c00 + j); bb[j] = (short)0x3f00; }
  bar();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters * 48; ++it) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if constexpr (TYPE == 0) MFMA(a);
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(r.acc[a]) : "v"(ba), "v"(bb));
#pragma unroll
      for (int q = 0; q < NF; ++q) SG(q & 7);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = fold(r);
}

template <class K>
static void run(const char* name, K kern, int blocks, int layers, double points_per_block, float* out, long long* clk) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, clk, layers);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, clk, layers);
  CK(hipEventRecord(b));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  long long c;
  CK(hipMemcpy(&c, clk, sizeof(c), hipMemcpyDeviceToHost));
  const double per_layer = (double)c / layers;
  const double per_cu_points = points_per_block * blocks / 256.0;   // points a CU carries through one layer in that time
  printf("%-58s blocks %4d  %9.0f s_memtime ticks per layer (100 MHz)  wall %8.1f us per layer  -> %7.2f us per layer per 128 points of a CU\n", name, blocks,
         per_layer, ms * 1e3 / layers, ms * 1e3 / layers * 128.0 / per_cu_points);
}

int main() {
  float* out;
  long long* clk;
  CK(hipMalloc(&out, sizeof(float) * 1024 * 256));
  CK(hipMalloc(&clk, 64));
  const int L = 64;
  // 64 points per workgroup of 4 waves (alt); the pp workgroup carries 128 points per "layer" (two phases)
  run("alt  split   1 WG/CU               ", layer_kernel<0, true, 0, 0>, 256, L, 64, out, clk);
  run("alt  split   2 WG/CU               ", layer_kernel<0, true, 0, 0>, 512, L, 64, out, clk);
  run("alt  split   2 WG/CU  prio M1 E0    ", layer_kernel<0, true, 1, 0>, 512, L, 64, out, clk);
  run("alt  split   2 WG/CU  prio M0 E3    ", layer_kernel<0, true, 0, 3>, 512, L, 64, out, clk);
  run("alt  split   2 WG/CU  half a layer apart", layer_kernel<0, true, 0, 0, true>, 512, L, 64, out, clk);
  run("alt  planes  2 WG/CU  half a layer apart", layer_kernel<0, false, 0, 0, true>, 512, L, 64, out, clk);
  run("alt  planes  2 WG/CU  apart, prio M0 E3", layer_kernel<0, false, 0, 3, true>, 512, L, 64, out, clk);
  run("alt  planes  2 WG/CU  apart, prio M1 E0", layer_kernel<0, false, 1, 0, true>, 512, L, 64, out, clk);
  run("alt  planes  1 WG/CU               ", layer_kernel<0, false, 0, 0>, 256, L, 64, out, clk);
  run("alt  planes  2 WG/CU               ", layer_kernel<0, false, 0, 0>, 512, L, 64, out, clk);
  run("pp   split   1 WG/CU (128 points)  ", layer_kernel<1, true, 0, 0>, 256, L, 128, out, clk);
  run("pp   planes  1 WG/CU (128 points)  ", layer_kernel<1, false, 0, 0>, 256, L, 128, out, clk);
  run("pp   planes  2 WG/CU (256 points)  ", layer_kernel<1, false, 0, 0>, 512, L, 128, out, clk);
  run("M only split   1 WG/CU             ", layer_kernel<2, true, 0, 0>, 256, L, 64, out, clk);
  run("M only planes  1 WG/CU             ", layer_kernel<2, false, 0, 0>, 256, L, 64, out, clk);
  run("M only planes  2 WG/CU             ", layer_kernel<2, false, 0, 0>, 512, L, 64, out, clk);
  run("E only         1 WG/CU             ", layer_kernel<3, false, 0, 0>, 256, L, 64, out, clk);
  run("E only         2 WG/CU             ", layer_kernel<3, false, 0, 0>, 512, L, 64, out, clk);
  run("f16  MFMA + 0 v_fma (192 MFMAs per 'layer')", fill_kernel<0, 0>, 256, L, 64, out, clk);
  run("f16  MFMA + 2 v_fma", fill_kernel<0, 2>, 256, L, 64, out, clk);
  run("f16  MFMA + 4 v_fma", fill_kernel<0, 4>, 256, L, 64, out, clk);
  run("f16  MFMA + 6 v_fma", fill_kernel<0, 6>, 256, L, 64, out, clk);
  run("f16  MFMA + 8 v_fma", fill_kernel<0, 8>, 256, L, 64, out, clk);
  run("bf16 MFMA + 0 v_fma", fill_kernel<1, 0>, 256, L, 64, out, clk);
  run("bf16 MFMA + 4 v_fma", fill_kernel<1, 4>, 256, L, 64, out, clk);
  run("bf16 MFMA + 6 v_fma", fill_kernel<1, 6>, 256, L, 64, out, clk);
  run("bf16 MFMA + 8 v_fma", fill_kernel<1, 8>, 256, L, 64, out, clk);
  run("f16  MFMA + 6 v_fma, 2 WG/CU", fill_kernel<0, 6>, 512, L, 64, out, clk);
  return 0;
}
