// Stand-alone timing of the M/V forward sweep (development aid; not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DRNB_MV_STAMP tools/mv_bench.hip -o tools/mv_bench && tools/mv_bench [points] [save]
// Random weights (mirror filled with random bf16 planes: timing only), random points.  Prints the launch time, the
// matrix-pipe share it implies at 2.4 GHz and, from s_memtime stamps of wave 0 of every workgroup, where a tile's clocks
// go: prologue / matrix loop / epilogue per layer, and the clock the chip held (s_memrealtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../rnb-neus-fork_amd/csrc/sweep_mv.hip"

using namespace rnb;

// (the library's other translation units are not linked: the two host symbols fused_t.hip refers to)
namespace rnb {
bool fused_supported(const Layout&) { return true; }
bool prof_enabled() { return false; }
void prof_begin(double, hipStream_t, const char*) {}
void prof_end(hipStream_t) {}
void set_error(const char*, ...) {}
}

#define CK(x)                                                                                      \
  do {                                                                                             \
    hipError_t e = (x);                                                                            \
    if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } \
  } while (0)

int main(int argc, char** argv) {
  const int64_t M = argc > 1 ? atoll(argv[1]) : (1 << 20);
  const bool save = argc > 2 && atoi(argv[2]) != 0;
  const int nh = 8;
  // packed: 9 matrices (layer 0: 256 x 64, 7 x 256 x 256, feature head) + biases + sdf row, like make_layout orders them
  std::vector<long long> w_off(nh + 1), b_off(nh + 1);
  long long off = 0;
  for (int l = 0; l <= nh; ++l) {
    const int K = l == 0 ? 64 : 256;
    w_off[l] = off; off += 256LL * K;
    b_off[l] = off; off += 256;
    off += 256LL * K;   // transposed copy
  }
  const long long wsdf = off; off += 256;
  const long long bsdf = off; off += 32;
  const long long total = off;
  std::vector<float> hp((size_t)total + (size_t)total * 3 / 2);
  srand(1);
  for (long long i = 0; i < total; ++i) hp[i] = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
  unsigned short* mir = reinterpret_cast<unsigned short*>(hp.data() + total);
  for (long long i = 0; i < total * 3; ++i) {
    const float v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    unsigned u; memcpy(&u, &v, 4);
    mir[i] = (unsigned short)(u >> 16);
  }
  float* packed;
  CK(hipMalloc(&packed, hp.size() * 4));
  CK(hipMemcpy(packed, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
  std::vector<float> hpts((size_t)M * 3);
  for (auto& v : hpts) v = (rand() / (float)RAND_MAX - 0.5f) * 1.8f;
  float *pts, *sdf, *state = nullptr, *x4 = nullptr, *ebuf = nullptr;
  CK(hipMalloc(&pts, hpts.size() * 4));
  CK(hipMemcpy(pts, hpts.data(), hpts.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&sdf, (size_t)M * 4));
  const unsigned blocks = (unsigned)(M / MV_PT);
  unsigned long long* stamps;
  CK(hipMalloc(&stamps, (size_t)blocks * 64 * 8));
  CK(hipMemset(stamps, 0, (size_t)blocks * 64 * 8));
  MvFwdArgs ga;
  memset(&ga, 0, sizeof(ga));
  FusedFwdArgs& g = ga.f;
  g.pts = pts; g.M = M; g.packed = packed;
  g.w3 = reinterpret_cast<const x3raw*>(packed + total);
  g.nh = nh; g.skip = 4; g.pe = 39; g.multires = 6; g.Ep = 64; g.scale = 1.f;
  if (save) {
    CK(hipMalloc(&state, (size_t)M * 256 * 4 * 2 * nh));
    CK(hipMalloc(&x4, (size_t)M * 16));
    CK(hipMalloc(&ebuf, (size_t)M * 64 * 4));
  }
  for (int l = 0; l < nh; ++l) {
    g.n_real[l] = (l + 1 == 4) ? 217 : 256;
    g.Kp[l] = l == 0 ? 64 : 256;
    g.w_off[l] = w_off[l]; g.b_off[l] = b_off[l];
    ga.st.nks[l] = g.Kp[l] / 16;
    ga.st.boff[l] = (unsigned)(6 * w_off[l]);
    if (save) { g.a[l] = state + (size_t)(2 * l) * M * 256; g.D[l] = state + (size_t)(2 * l + 1) * M * 256; }
  }
  ga.st.nmat = nh;
  g.wsdf_off = wsdf; g.bsdf_off = bsdf;
  g.sdf = sdf; g.x4 = x4; g.e = ebuf;
  ga.stamps = stamps;
  int* errw;
  CK(hipMalloc(&errw, 4));
  CK(hipMemset(errw, 0, 4));
  ga.err = errw;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto launch = [&]() {
    if (save) hipLaunchKernelGGL((sweep_mv_forward_kernel<true>), dim3(blocks), dim3(128 * MV_MW), 0, 0, ga);
    else hipLaunchKernelGGL((sweep_mv_forward_kernel<false>), dim3(blocks), dim3(128 * MV_MW), 0, 0, ga);
  };
  for (int it = 0; it < 3; ++it) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int iters = 10;
  for (int it = 0; it < iters; ++it) launch();
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  { int eh = 0; CK(hipMemcpy(&eh, errw, 4, hipMemcpyDeviceToHost)); if (eh) printf("!! a bounded wait gave up (error word set)\n"); }
  ms /= iters;
  // MFMAs per wave: (4 + 7 * 16) steps x 48; one wave per SIMD; tiles per CU = blocks / 256
  const double mfma_clk = (4 + 7 * 16) * 48 * 32.0 * blocks / 256.0;
  printf("sweep_mv_forward_kernel<%s>: %lld points: %.3f ms;  matrix pipe %.1f %% at 2.4 GHz;  %.1f TFLOP/s algorithmic\n", save ? "save" : "fwd",
         (long long)M, ms, 100.0 * mfma_clk / (ms * 1e-3 * 2.4e9), 1.049e6 * M / (ms * 1e-3) / 1e12);
#ifdef RNB_MV_STAMP
  std::vector<unsigned long long> st((size_t)blocks * 64);
  CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
  double pro = 0, prod[16] = {0}, all = 0, real = 0, tail = 0;
  for (unsigned b = 0; b < blocks; ++b) {
    const unsigned long long* s = &st[(size_t)b * 64];
    pro += (double)(s[1] - s[0]);
    for (int l = 0; l < nh; ++l) prod[l] += (double)(s[2 + l] - s[1 + l]);
    tail += (double)(s[40] - s[1 + nh]);
    all += (double)(s[40] - s[0]);
    real += (double)(s[61] - s[60]);
  }
  printf("per tile (wave 0, shader clocks): prologue %.0f | tail %.0f | whole tile %.0f clocks = %.2f us => %.2f GHz\n", pro / blocks,
         tail / blocks, all / blocks, real / blocks / 100.0, all / real / 10.0);
  for (int l = 0; l < nh; ++l) {
    double a = 0, b = 0, c = 0;
    if (l >= 1)
      for (unsigned bb = 0; bb < blocks; ++bb) {
        const unsigned long long* s = &st[(size_t)bb * 64 + 8 + 4 * l];
        a += (double)(s[1] - s[0]); b += (double)(s[2] - s[1]);
        c += (double)(s[0] - st[(size_t)bb * 64 + 8 + 4 * (l - 1) + 2]);
      }
    printf("  product %d: %.0f clocks (matrix work %d) | steps 0..14 %.0f  last step + hand-over %.0f  wait for first planes %.0f\n", l, prod[l] / blocks,
           (l == 0 ? 4 : 16) * 48 * 32, a / blocks, b / blocks, l >= 2 ? c / blocks : 0.0);
  }
#endif
  return 0;
}
