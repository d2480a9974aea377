// Optional event instrumentation of the GEMM family (bench.py's roofline line): when enabled, every
// layer-GEMM launch is bracketed by two hipEventRecord calls on the launch stream; rnb_profile_collect
// sums the elapsed times after the caller has synchronised.  Off by default: no events, no state.
#include <vector>

#include "rnb_internal.h"

namespace rnb {

struct ProfState {
  bool on = false;
  std::vector<hipEvent_t> pool;   // pairs (start, stop)
  size_t used = 0;                // events handed out
  std::vector<double> flops;      // per pair
};
static ProfState g_prof;

bool prof_enabled() { return g_prof.on; }

void prof_begin(double flops, hipStream_t s) {
  ProfState& p = g_prof;
  if (p.used + 2 > p.pool.size()) {
    for (int i = 0; i < 2; ++i) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) { p.on = false; return; }
      p.pool.push_back(e);
    }
  }
  p.flops.push_back(flops);
  (void)hipEventRecord(p.pool[p.used], s);
}
void prof_end(hipStream_t s) {
  ProfState& p = g_prof;
  (void)hipEventRecord(p.pool[p.used + 1], s);
  p.used += 2;
}

int profile_enable(int on) {
  g_prof.on = on != 0;
  g_prof.used = 0;
  g_prof.flops.clear();
  return RNB_OK;
}

int profile_collect(double* ms, int64_t* launches, double* flops) {
  ProfState& p = g_prof;
  double t = 0, f = 0;
  const size_t n = p.used / 2;
  for (size_t i = 0; i < n; ++i) {
    float e = 0.f;
    RNB_CHECK_HIP(hipEventElapsedTime(&e, p.pool[2 * i], p.pool[2 * i + 1]));
    t += e;
    f += p.flops[i];
  }
  if (ms) *ms = t;
  if (launches) *launches = (int64_t)n;
  if (flops) *flops = f;
  p.used = 0;
  p.flops.clear();
  return RNB_OK;
}

}  // namespace rnb
