// Optional per-launch timing with HIP events on the launch stream (used by bench.py's roofline leg;
// disabled by default -> zero work on the hot path).  Not capturable: keep it off under hipGraph capture.
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.hpp"
#include "prof.hpp"

namespace ctvae {

static int g_enabled = 0;
static std::mutex g_mu;
struct Rec {
  std::string name;
  hipEvent_t a, b;
  double flops, bytes;
};
static std::vector<Rec> g_recs;

bool prof_enabled() { return g_enabled != 0; }
bool prof_detailed() { return g_enabled >= 2; }

ProfScope::ProfScope(const char* name, hipStream_t st, double flops, double bytes) : idx_(-1), st_(st) {
  if (!g_enabled) return;
  Rec r;
  r.name = name;
  r.flops = flops;
  r.bytes = bytes;
  if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
  (void)hipEventRecord(r.a, st);
  std::lock_guard<std::mutex> lk(g_mu);
  g_recs.push_back(r);
  idx_ = (int)g_recs.size() - 1;
}

ProfScope::~ProfScope() {
  if (idx_ < 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  (void)hipEventRecord(g_recs[idx_].b, st_);
}

}  // namespace ctvae

using namespace ctvae;

extern "C" {

void ctvae_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_enabled = on;
}

// Calibration: n event pairs with nothing between them, logged as "(empty event pair)".  What such a pair measures
// is the cost the pair adds around every timed launch; bench.py subtracts it from the per-kernel averages.
void ctvae_prof_calibrate(void* stream, int n) {
  for (int i = 0; i < n; ++i) { ProfScope ps("(empty event pair)", (hipStream_t)stream, 0.0, 0.0); }
}

// Synchronises the recorded events, aggregates per kernel name and clears the log.  Writes lines
// "name\tcount\ttotal_ms\ttotal_flops\ttotal_bytes\n" into buf (truncated to n bytes); returns bytes needed.
size_t ctvae_prof_report(char* buf, size_t n) {
  std::lock_guard<std::mutex> lk(g_mu);
  struct Agg { long cnt = 0; double ms = 0, flops = 0, bytes = 0; };
  std::map<std::string, Agg> agg;
  for (auto& r : g_recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      Agg& a = agg[r.name];
      a.cnt++; a.ms += ms; a.flops += r.flops; a.bytes += r.bytes;
    }
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  g_recs.clear();
  std::string out;
  char line[512];
  for (auto& kv : agg) {
    snprintf(line, sizeof line, "%s\t%ld\t%.6f\t%.6e\t%.6e\n", kv.first.c_str(), kv.second.cnt, kv.second.ms, kv.second.flops,
             kv.second.bytes);
    out += line;
  }
  if (buf && n) {
    size_t c = out.size() < n - 1 ? out.size() : n - 1;
    for (size_t i = 0; i < c; ++i) buf[i] = out[i];
    buf[c] = 0;
  }
  return out.size() + 1;
}
}
