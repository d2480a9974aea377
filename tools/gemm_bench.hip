// Stand-alone micro-benchmark of the fp32-MFMA GEMM kernels (development aid; not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_bench.hip -o tools/gemm_bench && tools/gemm_bench
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../rnb-neus-fork_amd/csrc/gemm.hip.h"

using namespace rnb;

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e = (x);                                                        \
    if (e != hipSuccess) {                                                     \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

struct EpiStoreB {
  float* out;
  int ld;
  __device__ void apply4(int row, int col, vf4 v) const { *reinterpret_cast<vf4*>(out + (size_t)row * ld + col) = v; }
};
// softplus forward-like epilogue: two stores (a, D)
struct EpiSoft {
  const float* b;
  float* out;
  float* outD;
  int ld;
  __device__ void apply4(int row, int col, vf4 v) const {
    const vf4 bb = *reinterpret_cast<const vf4*>(b + col);
    vf4 a, D;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float ac, Dc;
      softplus_aD(v[c] + bb[c], ac, Dc);
      a[c] = ac;
      D[c] = Dc;
    }
    *reinterpret_cast<vf4*>(out + (size_t)row * ld + col) = a;
    *reinterpret_cast<vf4*>(outD + (size_t)row * ld + col) = D;
  }
};
// RA-like epilogue: two aux reads, two stores
struct EpiHeavy {
  const float* D;
  const float* gz;
  float* zR;
  float* un;
  int ld;
  __device__ void apply4(int row, int col, vf4 v) const {
    const size_t o = (size_t)row * ld + col;
    const vf4 Dv = *reinterpret_cast<const vf4*>(D + o), g = *reinterpret_cast<const vf4*>(gz + o);
    vf4 zr, u;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      zr[c] = 100.f * v[c] * g[c] * (1.f - Dv[c]);
      u[c] = v[c] * Dv[c];
    }
    *reinterpret_cast<vf4*>(zR + o) = zr;
    *reinterpret_cast<vf4*>(un + o) = u;
  }
};

// pure matrix-core loops (operands in registers): calibrates the device's sustained fp32 MFMA rate
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void peak32_kernel(float* out, int iters, float a0, float b0) {
  v16f acc[4];
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void peak16_kernel(float* out, int iters, float a0, float b0) {
  v4f acc[8];
  for (int t = 0; t < 8; ++t)
    for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int t = 0; t < 8; ++t)
    for (int r = 0; r < 4; ++r) s += acc[t][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F>
static float time_it(F f, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / iters;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 65536, N = 256, K = 256;
  std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hb(N);
  srand(1);
  for (auto& v : hA) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
  for (auto& v : hW) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
  for (auto& v : hb) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
  float *A, *W, *b, *C, *C2, *X1, *X2, *dW;
  CK(hipMalloc(&A, (size_t)M * K * 4));
  CK(hipMalloc(&W, (size_t)N * K * 4));
  CK(hipMalloc(&b, N * 4));
  CK(hipMalloc(&C, (size_t)M * N * 4));
  CK(hipMalloc(&C2, (size_t)M * N * 4));
  CK(hipMalloc(&X1, (size_t)M * N * 4));
  CK(hipMalloc(&X2, (size_t)M * N * 4));
  CK(hipMalloc(&dW, (size_t)N * K * 4));
  CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(X1, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(X2, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  const double flops = 2.0 * M * N * K;
  const int iters = 20;
  auto report = [&](const char* name, float us, double fl) {
    printf("%-42s %9.1f us  %7.1f TFLOP/s  (%.1f%% of 157.3)\n", name, us, fl / us * 1e-6, fl / us * 1e-6 / 157.3 * 100);
  };
  if (argc > 2) {   // device calibration: sustained fp32 MFMA rate with operands in registers
    const int it = 2000;
    for (int wgs : {256, 512, 1024}) {
      const double fl32 = (double)wgs * 4 * it * 32 * (2.0 * 32 * 32 * 2);
      const double fl16 = (double)wgs * 4 * it * 64 * (2.0 * 16 * 16 * 4);
      char nm[64];
      snprintf(nm, sizeof nm, "peak mfma 32x32x2  (%d WGs of 4 waves)", wgs);
      report(nm, time_it([&] { hipLaunchKernelGGL(peak32_kernel, dim3(wgs), dim3(256), 0, 0, C, it, 0.5f, 0.25f); }, 10), fl32);
      snprintf(nm, sizeof nm, "peak mfma 16x16x4  (%d WGs of 4 waves)", wgs);
      report(nm, time_it([&] { hipLaunchKernelGGL(peak16_kernel, dim3(wgs), dim3(256), 0, 0, C, it, 0.5f, 0.25f); }, 10), fl16);
    }
  }
  {
    EpiStoreB e{C, N};
    report("rows NT  BN=256 store", time_it([&] {
             hipLaunchKernelGGL((gemm_rows_kernel<false, 256, false, EpiStoreB>), dim3(M / 128, 1), dim3(256), 0, 0, A, K, W, K, N, K, e);
           }, iters), flops);
    report("rows NT  BN=128 store", time_it([&] {
             hipLaunchKernelGGL((gemm_rows_kernel<false, 128, false, EpiStoreB>), dim3(M / 128, 2), dim3(256), 0, 0, A, K, W, K, N, K, e);
           }, iters), flops);
    report("rows NN  BN=256 store", time_it([&] {
             hipLaunchKernelGGL((gemm_rows_kernel<true, 256, false, EpiStoreB>), dim3(M / 128, 1), dim3(256), 0, 0, A, K, W, K, N, K, e);
           }, iters), flops);
    report("rows NN  BN=128 store", time_it([&] {
             hipLaunchKernelGGL((gemm_rows_kernel<true, 128, false, EpiStoreB>), dim3(M / 128, 2), dim3(256), 0, 0, A, K, W, K, N, K, e);
           }, iters), flops);
  }
  {
    EpiSoft e{b, C, C2, N};
    report("rows NT  BN=256 softplus", time_it([&] {
             hipLaunchKernelGGL((gemm_rows_kernel<false, 256, false, EpiSoft>), dim3(M / 128, 1), dim3(256), 0, 0, A, K, W, K, N, K, e);
           }, iters), flops);
  }
  {
    EpiHeavy e{X1, X2, C, C2, N};
    report("rows NT  BN=256 RA-like epilogue", time_it([&] {
             hipLaunchKernelGGL((gemm_rows_kernel<false, 256, false, EpiHeavy>), dim3(M / 128, 1), dim3(256), 0, 0, A, K, W, K, N, K, e);
           }, iters), flops);
    report("rows NT  BN=128 RA-like epilogue", time_it([&] {
             hipLaunchKernelGGL((gemm_rows_kernel<false, 128, false, EpiHeavy>), dim3(M / 128, 2), dim3(256), 0, 0, A, K, W, K, N, K, e);
           }, iters), flops);
  }
  {
    DwPair p1{X1, N, A, K}, p2{X2, N, A, K};
    for (int njobs : {1, 8}) {
      for (int splits : {32, 64, 128, 256}) {
        DwGroup g;
        g.njobs = njobs;
        g.M = M;
        for (int j = 0; j < njobs; ++j) {
          DwJob& J = g.job[j];
          J.p1 = p1; J.p2 = p2; J.dW = dW; J.db = nullptr;
          J.npairs = 2; J.N = N; J.K = K; J.lddw = K; J.bias_pair = 1;
          J.splits = splits; J.rows_per_split = M / splits;
          J.block_end = (j + 1) * 4 * splits;
        }
        char name[96];
        snprintf(name, sizeof name, "dW 2 pairs (atomics), %d job(s)/launch, %d splits", njobs, splits);
        report(name, time_it([&] {
                 hipLaunchKernelGGL((gemm_dw_kernel<false, 128>), dim3(g.job[njobs - 1].block_end), dim3(256), 0, 0, g);
               }, iters), 2 * flops * njobs);
        snprintf(name, sizeof name, "dW direct (no LDS),   %d job(s)/launch, %d splits", njobs, splits);
        report(name, time_it([&] {
                 hipLaunchKernelGGL((gemm_dw_direct_kernel<128, 3>), dim3(g.job[njobs - 1].block_end), dim3(256), 0, 0, g);
               }, iters), 2 * flops * njobs);
      }
    }
  }
  // correctness spot check of the NT kernel against a host dot product
  {
    EpiStoreB e{C, N};
    hipLaunchKernelGGL((gemm_rows_kernel<false, 256, false, EpiStoreB>), dim3(M / 128, 1), dim3(256), 0, 0, A, K, W, K, N, K, e);
    std::vector<float> hC((size_t)256 * N);
    CK(hipMemcpy(hC.data(), C + (size_t)(M - 256) * N, hC.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int r = 0; r < 256; r += 37)
      for (int c = 0; c < N; c += 13) {
        double s = 0;
        for (int k = 0; k < K; ++k) s += (double)hA[(size_t)(M - 256 + r) * K + k] * hW[(size_t)c * K + k];
        worst = fmax(worst, fabs(s - hC[(size_t)r * N + c]));
      }
    printf("max |err| vs host fp64 dot: %.3e\n", worst);
  }
  return 0;
}
