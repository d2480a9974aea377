// Probe: which ingredient of the fused layer loop costs matrix-pipe utilisation?  (development aid)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_probe.hip -o tools/mfma_probe && tools/mfma_probe
// Variants of one 64-row x 256-col tile run through 8 "layers" of K = 256 per workgroup (4 waves, each
// 64 rows x 64 cols = 2x2 MFMA tiles), 2 workgroups per CU:
//   A_LDS : A fragments come from LDS (ds_read_b128) instead of registers
//   B_GLB : B fragments stream from global memory (L2) with a one-block prefetch instead of registers
//   BAR   : two workgroup barriers per layer
//   SCHED : sched_barrier(0) after every 16-MFMA group
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float vf4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int FP = 260;

template <bool A_LDS, bool B_GLB, bool BAR, bool SCHED, int TI, bool PIPE = false, int EPI = 0>
__global__ __launch_bounds__(256, 2) void probe_kernel(const float* __restrict__ W, float* __restrict__ out, int layers) {
  __shared__ __attribute__((aligned(16))) float X[32 * TI * FP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = wave * 64;
  const int i = lane & 31, h = lane >> 5;
  for (int idx = tid; idx < 32 * TI * FP; idx += 256) X[idx] = 0.001f * (idx % 97);
  __syncthreads();
  v16f acc[TI][2];
  for (int a = 0; a < TI; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  vf4 areg[TI];
  for (int a = 0; a < TI; ++a) areg[a] = vf4{0.01f * lane, 0.02f, 0.03f, 0.04f};
  constexpr bool XPF = (EPI == 4 || EPI == 5);   // next layer's first weight block is loaded before the stores
  constexpr bool WIDE = (EPI == 3 || EPI == 4);  // a / D leave through LDS as 16-byte-per-lane row stores
  vf4 bn[2][4];
  for (int l = 0; l < layers; ++l) {
    const float* Wl = W + (size_t)l * 256 * 256;
    vf4 b[2][4];
    auto loadb_at = [&](const float* Wx, int Q, vf4 (&d)[2][4]) {
      for (int tj = 0; tj < 2; ++tj) {
        const float* p = Wx + (size_t)(n0 + tj * 32 + i) * 256 + Q * 32 + h * 16;
        for (int q = 0; q < 4; ++q) d[tj][q] = *reinterpret_cast<const vf4*>(p + q * 4);
      }
    };
    auto loadb = [&](int Q, vf4 (&d)[2][4]) {
      for (int tj = 0; tj < 2; ++tj) {
        const float* p = Wl + (size_t)(n0 + tj * 32 + i) * 256 + Q * 32 + h * 16;
        for (int q = 0; q < 4; ++q) d[tj][q] = B_GLB ? *reinterpret_cast<const vf4*>(p + q * 4) : vf4{0.1f, 0.2f, 0.3f, 0.4f + 0.001f * Q};
      }
    };
    if (!(XPF && l > 0)) loadb(0, bn);
    for (int Q = 0; Q < 8; ++Q) {
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int q = 0; q < 4; ++q) b[tj][q] = bn[tj][q];
      if (Q + 1 < 8) loadb(Q + 1, bn);
      vf4 an[TI];
      if (PIPE) {
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) an[ti] = *reinterpret_cast<const vf4*>(X + (ti * 32 + i) * FP + Q * 32 + h * 16);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        vf4 a[TI];
        if (PIPE) {
#pragma unroll
          for (int ti = 0; ti < TI; ++ti) a[ti] = an[ti];
          if (q < 3) {
#pragma unroll
            for (int ti = 0; ti < TI; ++ti)
              an[ti] = *reinterpret_cast<const vf4*>(X + (ti * 32 + i) * FP + Q * 32 + h * 16 + (q + 1) * 4);
          }
        } else {
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
          a[ti] = A_LDS ? *reinterpret_cast<const vf4*>(X + (ti * 32 + i) * FP + Q * 32 + h * 16 + q * 4) : areg[ti];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int ti = 0; ti < TI; ++ti)
              acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ti][c], b[tj][q][c], acc[ti][tj], 0, 0, 0);
        if (SCHED) __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (BAR) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (EPI == 0) {
        // light epilogue: touch one LDS element per lane so the barrier pair has something to order
        X[(lane & 31) * FP + n0 + (lane >> 5)] = acc[0][0][0];
      } else {
        if (EPI == 5) loadb_at(W + (size_t)((l + 1) % 8) * 256 * 256, 0, bn);
        // heavy epilogue: ~EPI-flavoured VALU work per accumulator element + LDS write (+ global stores)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
          for (int ti = 0; ti < TI; ++ti)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
              const int col = n0 + tj * 32 + i;
              float v = acc[ti][tj][r] * 1e-6f;
              float t = fminf(fmaxf(v * 100.f, -87.f), 20.f);
              float e = __builtin_amdgcn_exp2f(t * 1.44269504f);
              float u = 1.f + e;
              float lg = __builtin_amdgcn_logf(u) * 0.69314718f;
              float a = (t > 19.f) ? v : lg * 0.01f * (e * __builtin_amdgcn_rcpf(u == 1.f ? 1.f : u - 1.f));
              float D = e * __builtin_amdgcn_rcpf(u);
              X[row * FP + col] = a;
              if (EPI == 2 || EPI == 5) {
                out[((size_t)blockIdx.x * 32 * TI + row) * 256 + col] = a;
                out[((size_t)(gridDim.x + blockIdx.x) * 32 * TI + row) * 256 + col] = D;
              } else {
                acc[ti][tj][r] = D;   // keep D live
              }
            }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (WIDE) {
        if (EPI == 4) loadb_at(W + (size_t)((l + 1) % 8) * 256 * 256, 0, bn);
        // the tile (a_l) is row-major in LDS: every lane moves 16 bytes, a wave one full 1 KB row
        float* o0 = out + (size_t)blockIdx.x * 32 * TI * 256;
        float* o1 = out + (size_t)(gridDim.x + blockIdx.x) * 32 * TI * 256;
#pragma unroll
        for (int it = 0; it < 8 * TI; ++it) {
          const int idx = it * 256 + tid;
          const int row = idx >> 6, c4 = idx & 63;
          const vf4 v = *reinterpret_cast<const vf4*>(X + row * FP + c4 * 4);
          *reinterpret_cast<vf4*>(o0 + (size_t)row * 256 + c4 * 4) = v;
          *reinterpret_cast<vf4*>(o1 + (size_t)row * 256 + c4 * 4) = v;
        }
      }
    }
  }
  float s = 0.f;
  for (int a = 0; a < TI; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
  out[(size_t)blockIdx.x * 256 + tid] = s;
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

template <bool A, bool B, bool BAR, bool S, int TI, bool PIPE = false, int EPI = 0>
static void run(const char* name, const float* W, float* out, int wgs) {
  const int layers = 8;
  float us = time_it([&] { hipLaunchKernelGGL((probe_kernel<A, B, BAR, S, TI, PIPE, EPI>), dim3(wgs), dim3(256), 0, 0, W, out, layers); }, 10);
  const double fl = (double)wgs * layers * 2.0 * (32 * TI) * 256 * 256;
  printf("%-44s TI=%d wgs=%5d %9.1f us %7.1f TFLOP/s (%.1f%%)\n", name, TI, wgs, us, fl / us * 1e-6, fl / us * 1e-6 / 157.3 * 100);
}

int main() {
  float *W, *out;
  CK(hipMalloc(&W, (size_t)8 * 256 * 256 * 4));
  CK(hipMalloc(&out, (size_t)2 * 2048 * 64 * 256 * 4));
  CK(hipMemset(W, 0, (size_t)8 * 256 * 256 * 4));
  for (int wgs : {1024}) {
    run<false, false, false, false, 2>("regs only", W, out, wgs);
    run<true, false, false, false, 2>("A from LDS", W, out, wgs);
    run<true, false, false, true, 2>("A from LDS + sched_barrier", W, out, wgs);
    run<false, true, false, false, 2>("B from global", W, out, wgs);
    run<true, true, false, true, 2>("A LDS + B global + sched", W, out, wgs);
    run<true, true, true, true, 2>("A LDS + B global + sched + barriers", W, out, wgs);
    run<true, true, true, false, 2>("A LDS + B global + barriers (no sched)", W, out, wgs);
    run<true, true, true, true, 1>("A LDS + B global + sched + barriers", W, out, wgs * 2);
    run<true, true, true, true, 2, false, 1>("full loop + softplus-like epilogue (no stores)", W, out, wgs);
    run<true, true, true, true, 2, false, 2>("full loop + softplus-like epilogue + 2 stores", W, out, wgs);
    run<true, true, true, true, 2, false, 5>("  ... dword stores + cross-layer B prefetch", W, out, wgs);
    run<true, true, true, true, 2, false, 3>("  ... wide stores from LDS", W, out, wgs);
    run<true, true, true, true, 2, false, 4>("  ... wide stores + cross-layer B prefetch", W, out, wgs);
    run<true, true, true, true, 1, false, 1>("full loop + softplus-like epilogue (no stores)", W, out, wgs * 2);
    run<true, true, true, true, 1, false, 2>("full loop + softplus-like epilogue + 2 stores", W, out, wgs * 2);
    run<true, true, true, true, 1, false, 3>("  ... wide stores from LDS", W, out, wgs * 2);
    run<true, true, true, true, 1, false, 4>("  ... wide stores + cross-layer B prefetch", W, out, wgs * 2);
  }
  return 0;
}
