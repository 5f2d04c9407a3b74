// Launch profiler of the C ABI (include/cdseg.h: cdseg_prof_enable / cdseg_prof_summary / cdseg_prof_summary_class).
#include <mutex>
#include <utility>
#include <vector>

#include "common.h"
#include "prof.h"

namespace {
bool g_on = false;
std::mutex g_mu;  // launches may come from several host threads (one per lane)
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_events[CDSEG_PROF_CLASSES];
}  // namespace

bool cdseg_prof_begin(int cls, hipStream_t s, CdsegProfToken* tok) {
  if (!g_on || cls < 0 || cls >= CDSEG_PROF_CLASSES) return false;
  if (hipEventCreate(&tok->e0) != hipSuccess) return false;
  tok->cls = cls;
  (void)hipEventRecord(tok->e0, s);
  return true;
}

void cdseg_prof_end(const CdsegProfToken& tok, hipStream_t s) {
  hipEvent_t e1;
  if (hipEventCreate(&e1) != hipSuccess) return;
  (void)hipEventRecord(e1, s);
  std::lock_guard<std::mutex> lk(g_mu);
  g_events[tok.cls].emplace_back(tok.e0, e1);
}

// enable / disable event timing of the profiled launch classes (drops earlier records)
extern "C" int cdseg_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& v : g_events) {
    for (auto& ev : v) {
      (void)hipEventDestroy(ev.first);
      (void)hipEventDestroy(ev.second);
    }
    v.clear();
  }
  g_on = on != 0;
  return CDSEG_OK;
}

// after a device synchronisation: total milliseconds and number of launches recorded for one class
extern "C" int cdseg_prof_summary_class(int cls, double* total_ms, long* launches) {
  if (cls < 0 || cls >= CDSEG_PROF_CLASSES) return CDSEG_ERR_ARG;
  std::lock_guard<std::mutex> lk(g_mu);
  double t = 0.0;
  for (auto& ev : g_events[cls]) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ev.first, ev.second) != hipSuccess) return CDSEG_ERR_LAUNCH;
    t += ms;
  }
  if (total_ms) *total_ms = t;
  if (launches) *launches = (long)g_events[cls].size();
  return CDSEG_OK;
}

extern "C" int cdseg_prof_summary(double* total_ms, long* launches) {
  return cdseg_prof_summary_class(CDSEG_PROF_ATTENTION, total_ms, launches);
}
