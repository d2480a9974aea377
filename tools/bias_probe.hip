// GPU dev probe: signed rounding bias of (a) the softplus epilogue functions and (b) one x3 dot product of positive
// numbers, against fp64.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I rnb-neus-fork_amd/csrc tools/bias_probe.hip -o tools/bias_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#include "rnb_internal.h"
#include "gemm.hip.h"
using namespace rnb;

__global__ void sp_kernel(const float* z, float* a1, float* a2, float* a3, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float a, D;
  softplus_aD(z[i], a, D);
  a1[i] = a;
  a2[i] = softplus_a(z[i]);
  a3[i] = softplus100(z[i]);
}

// one wave: C[32x32] = A[32x256] B[256x32]^T through x3 (6 terms, the library's order), k-major operands in registers
__global__ void x3_kernel(const float* A, const float* B, float* C) {
  const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
  v16f acc[1][1];
  for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
  for (int ks = 0; ks < 16; ++ks) {
    vu4x a[1][3], b[1][3];
    const float* ap = A + i * 256 + ks * 16 + h * 8;
    const float* bp = B + i * 256 + ks * 16 + h * 8;
    x3_split8(*reinterpret_cast<const vf4*>(ap), *reinterpret_cast<const vf4*>(ap + 4), a[0][0], a[0][1], a[0][2]);
    x3_split8(*reinterpret_cast<const vf4*>(bp), *reinterpret_cast<const vf4*>(bp + 4), b[0][0], b[0][1], b[0][2]);
    if (ks == 0) x3_mfma<1, 1, true>(a, b, acc);
    else x3_mfma<1, 1, false>(a, b, acc);
  }
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
    C[row * 32 + i] = acc[0][0][r];
  }
}

int main() {
  const int n = 1 << 20;
  std::mt19937 rng(1);
  std::vector<float> z(n);
  std::normal_distribution<float> nd(0.03f, 0.06f);
  for (auto& v : z) v = nd(rng);
  float *dz, *d1, *d2, *d3;
  hipMalloc(&dz, n * 4); hipMalloc(&d1, n * 4); hipMalloc(&d2, n * 4); hipMalloc(&d3, n * 4);
  hipMemcpy(dz, z.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(sp_kernel, dim3(n / 256), dim3(256), 0, 0, dz, d1, d2, d3, n);
  std::vector<float> a1(n), a2(n), a3(n);
  hipMemcpy(a1.data(), d1, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(a2.data(), d2, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(a3.data(), d3, n * 4, hipMemcpyDeviceToHost);
  double s1 = 0, s2 = 0, s3 = 0, sc = 0, q1 = 0, q2 = 0, q3 = 0, qc = 0, sa = 0;
  for (int i = 0; i < n; ++i) {
    const double t = 100.0 * (double)z[i];
    const double ref = t > 20.0 ? (double)z[i] : std::log1p(std::exp(t)) / 100.0;
    const float cpu = t > 20.0 ? z[i] : std::log1p(std::exp((float)t)) / 100.f;
    s1 += a1[i] - ref; s2 += a2[i] - ref; s3 += a3[i] - ref; sc += cpu - ref; sa += ref;
    q1 += (a1[i] - ref) * (a1[i] - ref); q2 += (a2[i] - ref) * (a2[i] - ref); q3 += (a3[i] - ref) * (a3[i] - ref);
    qc += (cpu - ref) * (cpu - ref);
  }
  printf("softplus over %d z ~ N(0.03, 0.06): mean a = %.4e\n", n, sa / n);
  printf("  softplus_aD : mean err %+.3e rms %.3e\n", s1 / n, std::sqrt(q1 / n));
  printf("  softplus_a  : mean err %+.3e rms %.3e\n", s2 / n, std::sqrt(q2 / n));
  printf("  softplus100 : mean err %+.3e rms %.3e (libm expf/log1pf on the device)\n", s3 / n, std::sqrt(q3 / n));
  printf("  host fp32   : mean err %+.3e rms %.3e\n", sc / n, std::sqrt(qc / n));

  // x3 dot products of positive numbers: 32 x 32 outputs, K = 256, repeated over fresh data
  std::uniform_real_distribution<float> ua(0.0f, 0.12f), ub(0.105f, 0.115f);
  float *dA, *dB, *dC;
  hipMalloc(&dA, 32 * 256 * 4); hipMalloc(&dB, 32 * 256 * 4); hipMalloc(&dC, 32 * 32 * 4);
  double sb = 0, qb = 0, sf = 0, qf = 0, mean = 0;
  int cnt = 0;
  for (int rep = 0; rep < 64; ++rep) {
    std::vector<float> A(32 * 256), B(32 * 256), Cc(32 * 32);
    for (auto& v : A) v = ua(rng);
    for (auto& v : B) v = ub(rng);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(x3_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    hipMemcpy(Cc.data(), dC, Cc.size() * 4, hipMemcpyDeviceToHost);
    for (int r = 0; r < 32; ++r)
      for (int c = 0; c < 32; ++c) {
        double ref = 0;
        float f = 0.f;
        for (int k = 0; k < 256; ++k) { ref += (double)A[r * 256 + k] * (double)B[c * 256 + k]; f = fmaf(A[r * 256 + k], B[c * 256 + k], f); }
        const double e = Cc[r * 32 + c] - ref, ef = f - ref;
        sb += e; qb += e * e; sf += ef; qf += ef * ef; mean += ref; ++cnt;
      }
  }
  printf("x3 dot products (K = 256, positive operands, mean value %.4f, ulp %.2e):\n", mean / cnt, std::ldexp(1.0, -23 + (int)std::floor(std::log2(mean / cnt))));
  printf("  x3 MFMA      : mean err %+.3e rms %.3e\n", sb / cnt, std::sqrt(qb / cnt));
  printf("  host fp32 fma: mean err %+.3e rms %.3e (sequential)\n", sf / cnt, std::sqrt(qf / cnt));
  return 0;
}
