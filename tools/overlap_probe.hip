// Probe: does ordinary VALU work overlap with MFMA work on one gfx950 SIMD?  (development aid; round 2 rewrite)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/overlap_probe.hip -o tools/overlap_probe
//   tools/overlap_probe                    # wall times (HIP events)
//   rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES
//             -d gpurun_out/ovl -- tools/overlap_probe   # every variant is its own kernel name
//
// Part A  "same wave": ONE wave per SIMD issues [1 MFMA, NF independent v_fma_f32] repeated, all in inline asm so
//         the order is exactly as written.  If the matrix pipe and the VALU are separate, time stays at the bare MFMA
//         time until the fillers' issue cost exceeds the MFMA's shadow, then grows by 4 cycles per filler.
// Part B  "two waves": 512-thread workgroups, waves w and w + 4 share a SIMD (checked: every wave records
//         HW_ID.SIMD_ID); one half streams MFMAs, the other a VALU chain.  ORDER picks which half is the older one;
//         PACED = one accumulator (every MFMA depends on the previous: the wave is NOT always ready to issue).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef short v8s __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ inline unsigned hw_id() { return __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)); }   // HW_REG_HW_ID

#define MFMA_F32(acc) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(xa), "v"(xb))
#define MFMA_BF16(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(ya), "v"(yb))
#define FILL(r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(c0), "v"(c1))

// ---- Part A ---------------------------------------------------------------------------------------------
template <int MTYPE, int NF>
__global__ __launch_bounds__(256) void same_wave(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  v16f acc[4];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float xa = 0.01f * lane, xb = 0.5f, c0 = 0.999f, c1 = 1e-3f;
  v8s ya, yb;
  for (int j = 0; j < 8; ++j) { ya[j] = (short)(0x3c00 + lane + j); yb[j] = (short)0x3f00; }
  float f[8];
  for (int j = 0; j < 8; ++j) f[j] = 0.001f * (lane + j);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if (MTYPE == 0) MFMA_F32(acc[a]); else MFMA_BF16(acc[a]);
#pragma unroll
      for (int q = 0; q < NF; ++q) FILL(f[q & 7]);
    }
  }
  float s = 0.f;
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  for (int j = 0; j < 8; ++j) s += f[j];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

// ---- Part B ---------------------------------------------------------------------------------------------
// RUN: 1 = MFMA half only, 2 = VALU half only, 3 = both.  ORDER 0: waves 0-3 MFMA (older), 4-7 VALU; 1: swapped.
template <int MTYPE, int PACED, int ORDER, int RUN>
__global__ __launch_bounds__(512) void two_waves(float* out, unsigned* ids, int m_iters, int v_iters) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (ids && lane == 0) ids[blockIdx.x * 8 + wave] = hw_id();
  const bool is_m = ORDER == 0 ? wave < 4 : wave >= 4;
  if (is_m) {
    if (!(RUN & 1)) return;
    v16f acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float xa = 0.01f * lane, xb = 0.5f;
    v8s ya, yb;
    for (int j = 0; j < 8; ++j) { ya[j] = (short)(0x3c00 + lane + j); yb[j] = (short)0x3f00; }
    for (int it = 0; it < m_iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          if (MTYPE == 0) MFMA_F32(acc[PACED ? 0 : a]); else MFMA_BF16(acc[PACED ? 0 : a]);
        }
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
  } else {
    if (!(RUN & 2)) return;
    float c0 = 0.999f, c1 = 1e-3f;
    float f[8];
    for (int j = 0; j < 8; ++j) f[j] = 0.001f * (lane + j);
    for (int it = 0; it < v_iters; ++it) {
#pragma unroll
      for (int q = 0; q < 64; ++q) FILL(f[q & 7]);
    }
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += f[j];
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
  }
}

template <class F>
static float time_it(F f, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / iters;
}

template <int MTYPE, int NF>
static float run_a(float* out, int iters, float base) {
  const float t = time_it([&] { hipLaunchKernelGGL((same_wave<MTYPE, NF>), dim3(256), dim3(256), 0, 0, out, iters); }, 5);
  const double cyc_per_mfma = (double)t * 1e-6 * 2.4e9 / (4.0 * iters);   // at the nominal 2.4 GHz
  printf("  %-5s NF=%2d  %8.1f us  = %6.1f clk/MFMA @2.4GHz  (x%.3f of bare)\n", MTYPE ? "bf16" : "f32", NF, t, cyc_per_mfma,
         base > 0 ? t / base : 1.0);
  return t;
}

template <int MTYPE, int PACED, int ORDER>
static void run_b(float* out, unsigned* ids, int m_iters, int v_iters) {
  auto t = [&](auto kern) { return time_it([&] { hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, out, ids, m_iters, v_iters); }, 5); };
  const float tm = t(two_waves<MTYPE, PACED, ORDER, 1>), tv = t(two_waves<MTYPE, PACED, ORDER, 2>), tb = t(two_waves<MTYPE, PACED, ORDER, 3>);
  printf("  %-5s %-6s MFMA-half=%s : MFMA alone %8.1f us | VALU alone %8.1f us | both %8.1f us  (max %.1f, sum %.1f)\n",
         MTYPE ? "bf16" : "f32", PACED ? "paced" : "free", ORDER ? "waves4-7(younger)" : "waves0-3(older)", tm, tv, tb,
         tm > tv ? tm : tv, tm + tv);
}

int main() {
  float* out;
  unsigned* ids;
  CK(hipMalloc(&out, (size_t)256 * 512 * 4));
  CK(hipMalloc(&ids, 256 * 8 * 4));
  CK(hipMemset(ids, 0, 256 * 8 * 4));
  printf("== Part A: one wave per SIMD, [1 MFMA + NF x v_fma_f32] (inline asm, order as written)\n");
  {
    const int it = 4000;
    float b = run_a<0, 0>(out, it, 0);
    run_a<0, 2>(out, it, b); run_a<0, 4>(out, it, b); run_a<0, 8>(out, it, b); run_a<0, 12>(out, it, b);
    run_a<0, 14>(out, it, b); run_a<0, 16>(out, it, b); run_a<0, 24>(out, it, b); run_a<0, 32>(out, it, b);
    const int itb = 8000;
    b = run_a<1, 0>(out, itb, 0);
    run_a<1, 2>(out, itb, b); run_a<1, 4>(out, itb, b); run_a<1, 6>(out, itb, b); run_a<1, 8>(out, itb, b);
    run_a<1, 12>(out, itb, b); run_a<1, 16>(out, itb, b);
  }
  printf("== Part B: two waves per SIMD (512-thread workgroups), one half MFMA, the other a v_fma_f32 chain\n");
  hipLaunchKernelGGL((two_waves<0, 0, 0, 3>), dim3(256), dim3(512), 0, 0, out, ids, 10, 10);
  CK(hipDeviceSynchronize());
  {
    unsigned h[256 * 8];
    CK(hipMemcpy(h, ids, sizeof(h), hipMemcpyDeviceToHost));
    int same = 0, total = 0;
    for (int b = 0; b < 256; ++b)
      for (int w = 0; w < 4; ++w) { ++total; if (((h[b * 8 + w] >> 4) & 3) == ((h[b * 8 + w + 4] >> 4) & 3)) ++same; }
    printf("  HW_ID.SIMD_ID of block 0, waves 0..7:");
    for (int w = 0; w < 8; ++w) printf(" %u", (h[w] >> 4) & 3);
    printf("   | waves w and w+4 on the same SIMD in %d of %d pairs\n", same, total);
  }
  run_b<0, 0, 0>(out, nullptr, 2000, 1200);
  run_b<0, 0, 1>(out, nullptr, 2000, 1200);
  run_b<0, 1, 0>(out, nullptr, 2000, 1200);
  run_b<0, 1, 1>(out, nullptr, 2000, 1200);
  run_b<1, 0, 0>(out, nullptr, 4000, 1200);
  run_b<1, 0, 1>(out, nullptr, 4000, 1200);
  run_b<1, 1, 0>(out, nullptr, 4000, 1200);
  run_b<1, 1, 1>(out, nullptr, 4000, 1200);
  return 0;
}
