// Probe: one fused-forward-shaped tile loop with fp32 emulated by six bf16 MFMAs per product
// (x = x_hi + x_mid + x_lo, products hh, hm, mh, hl, lh, mm; fp32 accumulation).  (development aid)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/bf16x6_probe.hip -o tools/bf16x6_probe
// One workgroup = 64 points x 256 columns through 8 layers of K = 256; activations live in LDS as three bf16
// planes [64][264]; weights stream from global memory as three bf16 planes [256][256] per layer (64 contiguous
// bytes per lane per 64-k block).  8 waves (64 rows x 32 columns each), one workgroup per CU.
// Compare with tools/mfma_probe ("full loop + softplus-like epilogue (no stores)", same tile, fp32 MFMA).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int PB = 264;   // plane pitch in bf16 elements (528 B: same bank residue as the fp32 pitch 260)

__device__ inline u16 f2bf(float x) { return __builtin_bit_cast(u16, (__bf16)x); }
__device__ inline float bf2f(u16 b) { return __builtin_bit_cast(float, (unsigned)b << 16); }

template <int EPI>   // 0: trivial epilogue, 1: softplus-like + 3-way split + LDS plane writes
__global__ __launch_bounds__(512, 1) void probe(const u16* __restrict__ W, float* __restrict__ out, int layers) {
  __shared__ __attribute__((aligned(16))) u16 P[3][64 * PB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = wave * 32;
  const int i = lane & 31, h = lane >> 5;
  for (int idx = tid; idx < 3 * 64 * PB; idx += 512) (&P[0][0])[idx] = f2bf(0.001f * (idx % 97));
  __syncthreads();
  v16f acc[2];
  for (int l = 0; l < layers; ++l) {
    for (int a = 0; a < 2; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const u16* Wl = W + (size_t)l * 3 * 256 * 256;
    v8bf bn[3][4], b[3][4];
    auto loadb = [&](int Qb, v8bf (&d)[3][4]) {
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const u16* q = Wl + (size_t)p * 256 * 256 + (size_t)(n0 + i) * 256 + Qb * 64 + h * 32;
#pragma unroll
        for (int s = 0; s < 4; ++s) d[p][s] = *reinterpret_cast<const v8bf*>(q + s * 8);
      }
    };
    loadb(0, bn);
    for (int Qb = 0; Qb < 4; ++Qb) {
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int s = 0; s < 4; ++s) b[p][s] = bn[p][s];
      if (Qb + 1 < 4) loadb(Qb + 1, bn);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        v8bf a[2][3];
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
          for (int p = 0; p < 3; ++p)
            a[ti][p] = *reinterpret_cast<const v8bf*>(&P[p][(ti * 32 + i) * PB + Qb * 64 + h * 32 + s * 8]);
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
          acc[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ti][0], b[0][s], acc[ti], 0, 0, 0);   // hh
          acc[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ti][0], b[1][s], acc[ti], 0, 0, 0);   // hm
          acc[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ti][1], b[0][s], acc[ti], 0, 0, 0);   // mh
          acc[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ti][0], b[2][s], acc[ti], 0, 0, 0);   // hl
          acc[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ti][2], b[0][s], acc[ti], 0, 0, 0);   // lh
          acc[ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ti][1], b[1][s], acc[ti], 0, 0, 0);   // mm
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (EPI == 0) {
      P[0][(lane & 31) * PB + n0 + (lane >> 5)] = f2bf(acc[0][0]);
    } else {
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const int col = n0 + i;
          float v = acc[ti][r] * 1e-6f;
          float t = v * 100.f;
          float nt = -fabsf(t);
          float p0 = nt * 1.44269504f;
          float q = __builtin_fmaf(nt, 1.44269504f, -p0);
          float w0 = __builtin_amdgcn_exp2f(p0);
          float w = __builtin_fmaf(w0, q * 0.69314718f, w0);
          float u = 1.f + w;
          float lg = __builtin_amdgcn_logf(u);
          float d = w - (u - 1.f);
          float a = __builtin_fmaf(__builtin_fmaf(lg, 0.69314718f, __builtin_fmaf(-d, w, d)), 0.01f, fmaxf(v, 0.f));
          const u16 hi = f2bf(a);
          const float r1 = a - bf2f(hi);
          const u16 mid = f2bf(r1);
          const u16 lo = f2bf(r1 - bf2f(mid));
          P[0][row * PB + col] = hi;
          P[1][row * PB + col] = mid;
          P[2][row * PB + col] = lo;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  float s = 0.f;
  for (int a = 0; a < 2; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[(size_t)blockIdx.x * 512 + tid] = s + bf2f(P[0][tid]);
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

int main() {
  u16* W; float* out;
  CK(hipMalloc(&W, (size_t)8 * 3 * 256 * 256 * 2));
  CK(hipMalloc(&out, (size_t)2048 * 512 * 4));
  CK(hipMemset(W, 0, (size_t)8 * 3 * 256 * 256 * 2));
  const int wgs = 1024, layers = 8;
  const double fl = (double)wgs * layers * 2.0 * 64 * 256 * 256;   // fp32-equivalent FLOPs
  float t0 = time_it([&] { hipLaunchKernelGGL(probe<0>, dim3(wgs), dim3(512), 0, 0, W, out, layers); }, 10);
  float t1 = time_it([&] { hipLaunchKernelGGL(probe<1>, dim3(wgs), dim3(512), 0, 0, W, out, layers); }, 10);
  printf("bf16x6 loop, trivial epilogue          %8.1f us  %7.1f fp32-equivalent TFLOP/s (%.0f%% of the fp32 MFMA peak)\n", t0, fl / t0 * 1e-6, fl / t0 * 1e-6 / 157.3 * 100);
  printf("bf16x6 loop + softplus + 3-way split   %8.1f us  %7.1f fp32-equivalent TFLOP/s (%.0f%% of the fp32 MFMA peak)\n", t1, fl / t1 * 1e-6, fl / t1 * 1e-6 / 157.3 * 100);
  printf("(fp32 MFMA, same tile and epilogue without the split: tools/mfma_probe, ~555 us)\n");
  return 0;
}
