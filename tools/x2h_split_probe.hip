#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdio>
#include <cmath>
typedef float vf4 __attribute__((ext_vector_type(4)));
typedef float vf2 __attribute__((ext_vector_type(2)));
typedef _Float16 vh2 __attribute__((ext_vector_type(2)));
typedef unsigned vu4x __attribute__((ext_vector_type(4)));
__device__ inline unsigned pack2(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(vf2{a, b}, vh2)); }
__device__ inline float resid_lo(float x, unsigned h) {
  float r;
  asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(h));
  return r;
}
__device__ inline float resid_hi(float x, unsigned h) {
  float r;
  asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(h));
  return r;
}
__global__ void k(const float* X, unsigned* out) {
  vf4 x0 = *(const vf4*)(X + threadIdx.x * 4);
  unsigned h0 = pack2(x0.x, x0.y), h1 = pack2(x0.z, x0.w);
  unsigned l0 = pack2(resid_lo(x0.x, h0), resid_hi(x0.y, h0));
  unsigned l1 = pack2(resid_lo(x0.z, h1), resid_hi(x0.w, h1));
  out[threadIdx.x * 4] = h0; out[threadIdx.x * 4 + 1] = h1; out[threadIdx.x * 4 + 2] = l0; out[threadIdx.x * 4 + 3] = l1;
}
int main() {
  const int n = 256;
  float hx[n * 4]; unsigned ho[n * 4];
  for (int i = 0; i < n * 4; ++i) hx[i] = (float)(i * 0.37123 - 100.0) * (i % 7 == 0 ? 1e-3f : 1.f);
  float* dx; unsigned* d;
  hipMalloc(&dx, sizeof(hx)); hipMalloc(&d, sizeof(ho));
  hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(n), 0, 0, dx, d);
  hipMemcpy(ho, d, sizeof(ho), hipMemcpyDeviceToHost);
  double worst = 0;
  for (int i = 0; i < n; ++i) for (int e = 0; e < 4; ++e) {
    unsigned hw = ho[i * 4 + e / 2], lw = ho[i * 4 + 2 + e / 2];
    unsigned short hb = e & 1 ? hw >> 16 : hw & 0xffff, lb = e & 1 ? lw >> 16 : lw & 0xffff;
    _Float16 hh, ll; hh = __builtin_bit_cast(_Float16, hb); ll = __builtin_bit_cast(_Float16, lb);
    double x = hx[i * 4 + e], err = fabs((double)hh + (double)ll - x) / fabs(x);
    if (err > worst) worst = err;
  }
  printf("worst relative |hi + lo - x| / |x| = %.3e (2^-24 = %.3e)\n", worst, 1.0 / (1 << 24));
}
