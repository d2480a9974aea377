// Probe: can fp32 MFMA work of one wave overlap with ordinary VALU work (softplus-like epilogue) or global
// stores of ANOTHER wave on the same SIMD?  (development aid)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/overlap_probe.hip -o tools/overlap_probe
// One 512-thread workgroup per CU: waves 0-3 ("M") run a pure MFMA stream, waves 4-7 ("V") run a VALU /
// store stream; wave w and w+4 share a SIMD.  Times: M alone, V alone, both.  Perfect overlap: both ~= max;
// mutually exclusive issue: both ~= sum.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int VMODE, int MTYPE = 0, int PRIO = 0>   // PRIO: s_setprio of the non-matrix waves; VMODE 0: exp/log/rcp chain (softplus-like), 1: plain fma chain, 2: global dword
                                      // stores; MTYPE 0: v_mfma_f32_32x32x2_f32, 1: v_mfma_f32_32x32x16_bf16
__global__ __launch_bounds__(512) void k(float* out, int m_iters, int v_iters, int run_m, int run_v) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (wave < 4) {
    if (!run_m) return;
    v16f acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float x = 0.01f * lane, y = 0.5f;
    v8bf xb, yb;
    for (int j = 0; j < 8; ++j) { xb[j] = (__bf16)(0.01f * lane + j); yb[j] = (__bf16)0.5f; }
    for (int it = 0; it < m_iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          if (MTYPE == 0) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
          else acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, acc[a], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
  } else {
    if (!run_v) return;
    if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = 0.001f * (lane + j);
    float* o = out + (size_t)(gridDim.x + blockIdx.x) * 512 * 64;
    for (int it = 0; it < v_iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (VMODE == 0) {
          float t = fminf(fmaxf(v[j] * 100.f, -87.f), 20.f);
          float e = __builtin_amdgcn_exp2f(t * 1.44269504f);
          float u = 1.f + e;
          float lg = __builtin_amdgcn_logf(u) * 0.69314718f;
          v[j] = lg * 0.01f * (e * __builtin_amdgcn_rcpf(u)) + 1e-3f;
        } else if (VMODE == 1) {
#pragma unroll
          for (int q = 0; q < 12; ++q) v[j] = fmaf(v[j], 0.999f, 1e-3f);
        } else {
          o[((size_t)(it & 63) * 8 + j) * 512 + threadIdx.x] = v[j];
        }
      }
    }
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += v[j];
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

template <int VMODE, int MTYPE = 0, int PRIO = 0>
static void run(const char* name, float* out, int m_iters, int v_iters) {
  const int wgs = 256;
  auto t = [&](int rm, int rv) {
    return time_it([&] { hipLaunchKernelGGL((k<VMODE, MTYPE, PRIO>), dim3(wgs), dim3(512), 0, 0, out, m_iters, v_iters, rm, rv); }, 10);
  };
  const float tm = t(1, 0), tv = t(0, 1), tb = t(1, 1);
  printf("%-28s MFMA alone %8.1f us | other alone %8.1f us | both %8.1f us  (max %.1f, sum %.1f)\n", name, tm, tv, tb,
         tm > tv ? tm : tv, tm + tv);
}

int main() {
  float* out;
  CK(hipMalloc(&out, (size_t)2 * 256 * 512 * 64 * 4 + (1 << 20)));
  run<0>("softplus-like VALU", out, 2000, 2600);
  run<0>("softplus-like VALU (half)", out, 2000, 1300);
  run<1>("fma chain VALU", out, 2000, 3400);
  run<2>("global dword stores", out, 2000, 6000);
  printf("-- non-matrix waves at s_setprio 3 --\n");
  run<0, 0, 3>("softplus-like VALU", out, 2000, 2600);
  run<1, 0, 3>("fma chain VALU", out, 2000, 3400);
  run<2, 0, 3>("global dword stores", out, 2000, 6000);
  printf("-- v_mfma_f32_32x32x16_bf16 as the matrix stream --\n");
  run<0, 1>("softplus-like VALU", out, 4000, 2600);
  run<1, 1>("fma chain VALU", out, 4000, 3400);
  run<2, 1>("global dword stores", out, 4000, 6000);
  return 0;
}
