// Optional event instrumentation of the GEMM family (bench.py's roofline line): when enabled, every
// layer-GEMM launch is bracketed by two hipEventRecord calls on the launch stream; rnb_profile_collect
// sums the elapsed times after the caller has synchronised.  Off by default: no events, no state.
#include <stdio.h>
#include <string.h>

#include <vector>

#include "rnb_internal.h"

namespace rnb {

struct ProfState {
  bool on = false;
  std::vector<hipEvent_t> pool;   // pairs (start, stop)
  size_t used = 0;                // events handed out
  std::vector<double> flops;      // per pair
  std::vector<const char*> tag;   // per pair: kernel class (string literal) or nullptr
};
static ProfState g_prof;

bool prof_enabled() { return g_prof.on; }

void prof_begin(double flops, hipStream_t s, const char* tag) {
  ProfState& p = g_prof;
  if (p.used + 2 > p.pool.size()) {
    for (int i = 0; i < 2; ++i) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) { p.on = false; return; }
      p.pool.push_back(e);
    }
  }
  p.flops.push_back(flops);
  p.tag.push_back(tag);
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
  g_prof.tag.clear();
  return RNB_OK;
}

// per kernel class of the last collection (profile_report)
struct ProfClass { const char* tag; double ms, flops; int64_t n; };
static std::vector<ProfClass> g_classes;

int profile_collect(double* ms, int64_t* launches, double* flops) {
  ProfState& p = g_prof;
  double t = 0, f = 0;
  const size_t n = p.used / 2;
  g_classes.clear();
  for (size_t i = 0; i < n; ++i) {
    float e = 0.f;
    RNB_CHECK_HIP(hipEventElapsedTime(&e, p.pool[2 * i], p.pool[2 * i + 1]));
    t += e;
    f += p.flops[i];
    const char* tg = p.tag[i] ? p.tag[i] : "other";
    size_t c = 0;
    while (c < g_classes.size() && strcmp(g_classes[c].tag, tg) != 0) ++c;
    if (c == g_classes.size()) g_classes.push_back(ProfClass{tg, 0.0, 0.0, 0});
    g_classes[c].ms += e;
    g_classes[c].flops += p.flops[i];
    g_classes[c].n += 1;
  }
  if (ms) *ms = t;
  if (launches) *launches = (int64_t)n;
  if (flops) *flops = f;
  p.used = 0;
  p.flops.clear();
  p.tag.clear();
  return RNB_OK;
}

// one line per kernel class of the last profile_collect: "<tag> <ms> <launches> <flops>\n"; returns the bytes needed
int64_t profile_report(char* out, int64_t cap) {
  int64_t need = 0;
  for (const ProfClass& c : g_classes) {
    char line[160];
    const int k = snprintf(line, sizeof line, "%s %.6f %lld %.6e\n", c.tag, c.ms, (long long)c.n, c.flops);
    if (out && need + k < cap) memcpy(out + need, line, (size_t)k);
    need += k;
  }
  if (out && cap > 0) out[need < cap ? need : cap - 1] = 0;
  return need + 1;
}

}  // namespace rnb
