// prof.hip -- optional per-launch timing with HIP events on the launch stream (a bench /
// diagnostic facility: bench.py's live `roofline` numbers come from here).  Disabled by default;
// when enabled every kernel launch of the library is bracketed by two events.  Process-global state behind a
// mutex (taken only while profiling is on: the disabled path is one relaxed atomic load), so launches from several
// threads are recorded correctly; meaningful numbers still want one stream.
// Launches carry a REGION besides their kernel class: towers.hip marks the attention branches (LayerNorm + QKV +
// attention core + out-proj / temporal_fc) and the MLP branches, so that bench.py can report the MFMA fraction of the
// TimeSformer attention branches on their own.
#include <atomic>
#include <mutex>
#include <vector>

#include "common.h"

static std::atomic<long long> g_launches{0};
void vtc_count_launch() { g_launches.fetch_add(1, std::memory_order_relaxed); }
extern "C" long long vtc_debug_launch_count(void) { return g_launches.load(std::memory_order_relaxed); }

namespace {
struct Rec { hipEvent_t a, b; int cls, region; double work; const int *m_dev; int tag[3]; };
std::atomic<bool> g_on{false};
std::mutex g_mu;
thread_local int g_region = VTC_PROF_REGION_OTHER;
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

// m_dev != NULL: `work` is the work PER ROW and the row count lives in device memory (the sync-free ragged text tower): read
// back when the records are collected
namespace {
double rec_work(const Rec &r) {
  if (!r.m_dev) return r.work;
  int m = 0;
  if (hipMemcpy(&m, r.m_dev, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return 0.0;
  return r.work * m;
}
}  // namespace

ProfScope::ProfScope(int cls, double work, hipStream_t s, const int *m_dev) : stream_(s), idx_(-1) {
  if (!g_on.load(std::memory_order_relaxed)) return;
  std::lock_guard<std::mutex> lk(g_mu);
  Rec r;
  r.a = get_event(); r.b = get_event(); r.cls = cls; r.region = g_region; r.work = work; r.m_dev = m_dev;
  r.tag[0] = r.tag[1] = r.tag[2] = 0;
  if (!r.a || !r.b) return;
  (void)hipEventRecord(r.a, s);
  idx_ = (int)g_recs.size();
  g_recs.push_back(r);
}
ProfScope::~ProfScope() {
  if (idx_ < 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (idx_ < (int)g_recs.size()) (void)hipEventRecord(g_recs[idx_].b, stream_);
}

void ProfScope::tag(int a, int b, int c) {
  if (idx_ < 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (idx_ < (int)g_recs.size()) { g_recs[idx_].tag[0] = a; g_recs[idx_].tag[1] = b; g_recs[idx_].tag[2] = c; }
}

ProfRegion::ProfRegion(int region) : prev_(g_region) { g_region = region; }
ProfRegion::~ProfRegion() { g_region = prev_; }

extern "C" int vtc_prof_begin(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_recs.clear();
  g_pool_next = 0;
  g_on.store(true);
  return 0;
}

// ms / launches / work: [VTC_PROF_NCLASS * VTC_PROF_NREGION], index cls * VTC_PROF_NREGION + region
extern "C" int vtc_prof_end_regions(void *stream, double *ms, long long *launches, double *work) {
  g_on.store(false);
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) {
    vtc_set_error("prof_end: stream synchronize failed");
    return 1;
  }
  std::lock_guard<std::mutex> lk(g_mu);
  for (int c = 0; c < VTC_PROF_NCLASS * VTC_PROF_NREGION; ++c) { ms[c] = 0; launches[c] = 0; work[c] = 0; }
  for (const Rec &r : g_recs) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
    const int i = r.cls * VTC_PROF_NREGION + r.region;
    ms[i] += t; launches[i] += 1; work[i] += rec_work(r);
  }
  g_recs.clear();
  g_pool_next = 0;
  return 0;
}

extern "C" int vtc_prof_end_records(void *stream, int max, int *n, int *cls, int *region, double *ms, double *work, int *tag) {
  g_on.store(false);
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) {
    vtc_set_error("prof_end: stream synchronize failed");
    return 1;
  }
  std::lock_guard<std::mutex> lk(g_mu);
  int k = 0;
  for (const Rec &r : g_recs) {
    float t = 0.f;
    if (k >= max || hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
    cls[k] = r.cls; region[k] = r.region; ms[k] = t; work[k] = rec_work(r);
    tag[3 * k] = r.tag[0]; tag[3 * k + 1] = r.tag[1]; tag[3 * k + 2] = r.tag[2];
    ++k;
  }
  *n = k;
  g_recs.clear();
  g_pool_next = 0;
  return 0;
}

extern "C" int vtc_prof_end(void *stream, double *ms, long long *launches, double *work) {
  double m[VTC_PROF_NCLASS * VTC_PROF_NREGION], w[VTC_PROF_NCLASS * VTC_PROF_NREGION];
  long long l[VTC_PROF_NCLASS * VTC_PROF_NREGION];
  if (int rc = vtc_prof_end_regions(stream, m, l, w)) return rc;
  for (int c = 0; c < VTC_PROF_NCLASS; ++c) {
    ms[c] = 0; launches[c] = 0; work[c] = 0;
    for (int r = 0; r < VTC_PROF_NREGION; ++r) {
      ms[c] += m[c * VTC_PROF_NREGION + r]; launches[c] += l[c * VTC_PROF_NREGION + r]; work[c] += w[c * VTC_PROF_NREGION + r];
    }
  }
  return 0;
}
