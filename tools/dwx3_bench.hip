// Stand-alone timing of gemm_dw_x3_kernel (development aid; not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dwx3_bench.hip -o tools/dwx3_bench && tools/dwx3_bench [points]
// 16 operand pairs of 256 x 256 gradients over M points, split like DwBatch::flush_staged (one round of 256 workgroups),
// random operands.  Prints the launch time, the matrix-pipe share it implies at 2.4 GHz, and (stamped build of the same
// kernel, DUMMY = 1) where wave 0 of a workgroup spends its clocks: barrier wait / chunk issue / load issue.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>
#include <vector>

#include "../rnb-neus-fork_amd/csrc/gemm.hip.h"

using namespace rnb;

#define CK(x)                                                                                      \
  do {                                                                                             \
    hipError_t e = (x);                                                                            \
    if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } \
  } while (0)

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 65536, N = 256, K = 256, njobs = 8;
  std::vector<float> h((size_t)M * 256);
  srand(1);
  for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
  float* op[4];
  for (auto& p : op) {
    CK(hipMalloc(&p, (size_t)M * 256 * 4));
    CK(hipMemcpy(p, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  }
  DwGroup g;
  g.njobs = njobs;
  g.M = M;
  const int splits = 256 / njobs;
  int rows = (M + splits - 1) / splits;
  rows = (rows + 31) / 32 * 32;
  float *slab, *dW;
  unsigned long long* stamps;
  CK(hipMalloc(&slab, (size_t)njobs * splits * (N * K + N) * 4));
  CK(hipMalloc(&dW, (size_t)njobs * N * K * 4));
  CK(hipMalloc(&stamps, 256 * 8 * 8));
  int end = 0;
  for (int q = 0; q < njobs; ++q) {
    DwJob& j = g.job[q];
    j.p1 = DwPair{op[0], 256, op[1], 256};
    j.p2 = DwPair{op[2], 256, op[3], 256};
    j.npairs = 2; j.N = N; j.K = K; j.lddw = K; j.bias_pair = 1;
    j.dW = dW + (size_t)q * N * K;
    j.db = reinterpret_cast<float*>(stamps);
    j.splits = splits; j.rows_per_split = rows;
    end += splits; j.block_end = end;
    j.part = slab + (size_t)q * splits * (N * K + N);
    j.partb = j.part + (size_t)splits * N * K;
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto run = [&](auto np_tag) {
    constexpr int NP = decltype(np_tag)::value;
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((gemm_dw_x3_kernel<0, NP>), dim3(end), dim3(512), 0, 0, g);
    CK(hipEventRecord(e0));
    const int iters = 20;
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL((gemm_dw_x3_kernel<0, NP>), dim3(end), dim3(512), 0, 0, g);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    const int terms = NP == 3 ? 6 : 3;
    const double mfma_cycles_per_simd = 2.0 * njobs * (double)M / 16 * 8 * (8 * terms) * 32 / 1024;   // pairs x chunks x waves x MFMAs x 32 clk
    printf("gemm_dw_x3_kernel<0, %d>: %d points, %d jobs x 2 pairs: %.1f us;  matrix pipe %.0f %% at 2.4 GHz;  operands %.2f TB/s\n", NP, M,
           njobs, ms * 1e3, 100.0 * mfma_cycles_per_simd / (ms * 1e-3 * 2.4e9), 2.0 * njobs * 2.0 * M * 1024 / (ms * 1e-3) / 1e12);
    hipLaunchKernelGGL((gemm_dw_x3_kernel<1, NP>), dim3(end), dim3(512), 0, 0, g);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> st(256 * 8);
    CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
    double a[5] = {0, 0, 0, 0, 0};
    for (int b = 0; b < end; ++b)
      for (int k = 0; k < 5; ++k) a[k] += (double)st[8 * b + k] / end;
    const double nch = 2.0 * rows / 16;   // chunk iterations per workgroup (both pairs)
    printf("  stamped (s_memtime = shader clocks): per chunk  barrier wait %.0f  chunk issue %.0f  load issue %.0f  | whole kernel %.0f clocks"
           " = %.1f us of s_memrealtime => %.2f GHz (matrix work per chunk and SIMD: %d)\n", a[0] / nch * 2, a[1] / nch * 2,
           a[2] / nch * 2, a[3], a[4] / 100.0, a[3] / (a[4] / 100.0) / 1e3, 2 * 8 * terms * 32);
  };
  run(std::integral_constant<int, 3>());
  run(std::integral_constant<int, 2>());   // x2h: three fp16 terms (no recorded maxima here: the clamp scale; timing only)
  return 0;
}
