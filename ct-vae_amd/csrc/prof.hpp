// RAII launch timer (see prof.hip).  Usage in a launcher:  ProfScope ps("kernel<cfg>", stream, flops, bytes);
#pragma once
#include <hip/hip_runtime.h>

namespace ctvae {
bool prof_enabled();
bool prof_detailed();  // level 2: kernel names carry the problem shape
struct ProfScope {
  ProfScope(const char* name, hipStream_t st, double flops, double bytes);
  ~ProfScope();
  int idx_;
  hipStream_t st_;
};
}  // namespace ctvae
