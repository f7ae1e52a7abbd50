// Context storage, the thread-local binding and the optional in-stream profiling (see ctx.h).
//
// Profiling: with option "profile" = 1 every launcher brackets its kernel with a pair of hipEvents on the launch stream;
// u2tok_profile_collect() sums elapsed time, launch counts, algorithmic FLOPs and bytes per kernel class.  bench.py uses
// it (in a separate, instrumented pass) to get each kernel class's measured duration for the roofline lines.
#include "kernels.h"

namespace u2 {

namespace {
Context g_default_ctx;
thread_local Context* tl_ctx = nullptr;
}  // namespace

Context& ctx() { return tl_ctx ? *tl_ctx : g_default_ctx; }
void ctx_bind(Context* c) { tl_ctx = c; }
Context* ctx_bound() { return tl_ctx; }

SideStream* Context::side_for(hipStream_t owner) {
  std::lock_guard<std::mutex> lk(mu);
  for (SideStream* s : sides)
    if (s->owner == owner) return s;
  SideStream* s = new SideStream();
  s->owner = owner;
  bool ok = hipStreamCreateWithFlags(&s->s, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&s->fork, hipEventDisableTiming) == hipSuccess;
  for (auto& e : s->done) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    delete s;  // partially created handles are leaked on purpose: this only happens when the runtime is unusable
    return nullptr;
  }
  sides.push_back(s);
  return s;
}

void Context::set_scratch(hipStream_t st, void* p, size_t bytes) {
  std::lock_guard<std::mutex> lk(mu);
  for (auto& s : scratch)
    if (s.st == st) { s.p = p; s.bytes = bytes; return; }
  scratch.push_back({st, p, bytes});
}

Scratch Context::scratch_of(hipStream_t st) {
  std::lock_guard<std::mutex> lk(mu);
  for (auto& s : scratch)
    if (s.st == st) return s;
  return {st, nullptr, 0};
}

void Context::release() {
  std::lock_guard<std::mutex> lk(mu);
  for (SideStream* s : sides) {
    (void)hipStreamDestroy(s->s);
    (void)hipEventDestroy(s->fork);
    for (auto& e : s->done) (void)hipEventDestroy(e);
    delete s;
  }
  sides.clear();
  for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
  ev_pool.clear();
  recs.clear();
  ev_used = 0;
}

// ------------------------------------------------------------------------------------------------ profiling
void prof_enable(bool on) {
  Context& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  c.opt.profile = on ? 1 : 0;
  c.recs.clear();
  c.ev_used = 0;
}
bool prof_enabled() { return opts().profile != 0; }

ProfScope::ProfScope(int cat, double flops, hipStream_t st, double bytes) : idx_(-1), st_(st), c_(nullptr) {
  Context& c = ctx();
  if (!c.opt.profile) return;
  std::lock_guard<std::mutex> lk(c.mu);
  auto get_event = [&]() -> hipEvent_t {
    if (c.ev_used == c.ev_pool.size()) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return nullptr;
      c.ev_pool.push_back(e);
    }
    return c.ev_pool[c.ev_used++];
  };
  ProfRec r{get_event(), get_event(), cat, flops, bytes};
  if (!r.a || !r.b) return;
  (void)hipEventRecord(r.a, st);
  c.recs.push_back(r);
  idx_ = (int)c.recs.size() - 1;
  c_ = &c;
}
ProfScope::~ProfScope() {
  if (idx_ < 0) return;
  std::lock_guard<std::mutex> lk(c_->mu);
  if ((size_t)idx_ < c_->recs.size()) (void)hipEventRecord(c_->recs[idx_].b, st_);
}

int prof_collect(double* ms, double* flops, double* bytes, int64_t* count, int ncat) {
  Context& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  for (int i = 0; i < ncat; ++i) { ms[i] = 0; flops[i] = 0; count[i] = 0; if (bytes) bytes[i] = 0; }
  int rc = U2_OK;
  for (auto& r : c.recs) {
    float t = 0.f;
    if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) { rc = U2_ERR_LAUNCH; break; }
    if (r.cat >= 0 && r.cat < ncat) {
      ms[r.cat] += t; flops[r.cat] += r.flops; count[r.cat] += 1;
      if (bytes) bytes[r.cat] += r.bytes;
    }
  }
  c.recs.clear();
  c.ev_used = 0;
  return rc;
}

}  // namespace u2
