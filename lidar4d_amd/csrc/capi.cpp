// Error reporting + version of the C ABI (include/lidar4d_hip.h).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include "../../include/lidar4d_hip.h"

static thread_local char g_err[512] = "";

extern "C" void l4d_set_error(int code, const char* where) {
  const char* txt = code > 1 ? hipGetErrorString((hipError_t)code) : "invalid argument";
  snprintf(g_err, sizeof(g_err), "%s: %s (code %d)", where, txt, code);
}

extern "C" const char* l4d_last_error(void) { return g_err; }

extern "C" int l4d_version(void) { return L4D_ABI_VERSION; }

// ---- per-kernel timing (bench.py) ------------------------------------------------------------------------------------
#include <vector>
struct ProfRec {
  const char* name;
  hipEvent_t start, stop;
};
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;

extern "C" int l4d_profile_enable(int on) {
  for (auto& r : g_prof) {
    (void)hipEventDestroy(r.start);
    (void)hipEventDestroy(r.stop);
  }
  g_prof.clear();
  g_prof_on = on != 0;
  return 0;
}

extern "C" int l4d_prof_begin(const char* kernel, void* stream) {
  if (!g_prof_on) return -1;
  ProfRec r;
  r.name = kernel;
  if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return -1;
  (void)hipEventRecord(r.start, (hipStream_t)stream);
  g_prof.push_back(r);
  return (int)g_prof.size() - 1;
}

extern "C" void l4d_prof_end(int idx, void* stream) { (void)hipEventRecord(g_prof[idx].stop, (hipStream_t)stream); }

extern "C" int l4d_profile_count(void) { return (int)g_prof.size(); }

// record i -> kernel name (as written at the launch site, template arguments included) and its duration in ms
extern "C" int l4d_profile_get(int i, const char** name, float* ms) {
  if (i < 0 || i >= (int)g_prof.size()) { l4d_set_error(1, "l4d_profile_get: index out of range"); return 1; }
  hipError_t e = hipEventSynchronize(g_prof[i].stop);
  if (e == hipSuccess) e = hipEventElapsedTime(ms, g_prof[i].start, g_prof[i].stop);
  if (e != hipSuccess) { l4d_set_error((int)e, "l4d_profile_get"); return (int)e; }
  *name = g_prof[i].name;
  return 0;
}
