// prof.hip -- optional per-launch timing with HIP events on the launch stream (a bench /
// diagnostic facility: bench.py's live `roofline` numbers come from here).  Disabled by default;
// when enabled every kernel launch of the library is bracketed by two events.  Process-global,
// not thread-safe: enable it from the one thread that drives the stream.
#include <vector>

#include "common.h"

namespace {
struct Rec { hipEvent_t a, b; int cls; double work; };
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
size_t g_pool_next = 0;

hipEvent_t get_event() {
  if (g_pool_next == g_pool.size()) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    g_pool.push_back(e);
  }
  return g_pool[g_pool_next++];
}
}  // namespace

ProfScope::ProfScope(int cls, double work, hipStream_t s) : stream_(s), idx_(-1) {
  if (!g_on) return;
  Rec r;
  r.a = get_event(); r.b = get_event(); r.cls = cls; r.work = work;
  if (!r.a || !r.b) return;
  (void)hipEventRecord(r.a, s);
  idx_ = (int)g_recs.size();
  g_recs.push_back(r);
}
ProfScope::~ProfScope() {
  if (idx_ >= 0) (void)hipEventRecord(g_recs[idx_].b, stream_);
}

extern "C" int vtc_prof_begin(void) {
  g_recs.clear();
  g_pool_next = 0;
  g_on = true;
  return 0;
}

extern "C" int vtc_prof_end(void *stream, double *ms, long long *launches, double *work) {
  g_on = false;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) {
    vtc_set_error("prof_end: stream synchronize failed");
    return 1;
  }
  for (int c = 0; c < VTC_PROF_NCLASS; ++c) { ms[c] = 0; launches[c] = 0; work[c] = 0; }
  for (const Rec &r : g_recs) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
    ms[r.cls] += t; launches[r.cls] += 1; work[r.cls] += r.work;
  }
  g_recs.clear();
  g_pool_next = 0;
  return 0;
}
