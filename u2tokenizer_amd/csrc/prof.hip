// Optional in-stream profiling: when enabled (u2tok_set_option("profile", 1)) every launcher brackets its
// kernel with a pair of hipEvents on the launch stream; u2tok_profile_collect() sums elapsed time, launch
// counts and algorithmic FLOPs per kernel class.  bench.py uses it (in a separate, instrumented pass) to
// get the dominant kernel's measured duration for the roofline line without leaving the process.
#include <vector>
#include "kernels.h"

namespace u2 {

namespace {
struct Rec { hipEvent_t a, b; int cat; double flops, bytes; };
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
size_t g_pool_used = 0;

hipEvent_t get_event() {
  if (g_pool_used == g_pool.size()) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    g_pool.push_back(e);
  }
  return g_pool[g_pool_used++];
}
}  // namespace

void prof_enable(bool on) {
  g_on = on;
  g_recs.clear();
  g_pool_used = 0;
}
bool prof_enabled() { return g_on; }

ProfScope::ProfScope(int cat, double flops, hipStream_t st, double bytes) : idx_(-1), st_(st) {
  if (!g_on) return;
  Rec r{get_event(), get_event(), cat, flops, bytes};
  if (!r.a || !r.b) return;
  hipEventRecord(r.a, st);
  g_recs.push_back(r);
  idx_ = (int)g_recs.size() - 1;
}
ProfScope::~ProfScope() {
  if (idx_ >= 0) hipEventRecord(g_recs[idx_].b, st_);
}

int prof_collect(double* ms, double* flops, double* bytes, int64_t* count, int ncat) {
  for (int i = 0; i < ncat; ++i) { ms[i] = 0; flops[i] = 0; count[i] = 0; if (bytes) bytes[i] = 0; }
  for (auto& r : g_recs) {
    if (hipEventSynchronize(r.b) != hipSuccess) return U2_ERR_LAUNCH;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return U2_ERR_LAUNCH;
    if (r.cat >= 0 && r.cat < ncat) {
      ms[r.cat] += t; flops[r.cat] += r.flops; count[r.cat] += 1;
      if (bytes) bytes[r.cat] += r.bytes;
    }
  }
  g_recs.clear();
  g_pool_used = 0;
  return U2_OK;
}

}  // namespace u2
