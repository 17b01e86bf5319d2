// Error reporting + version of the C ABI (include/lidar4d_hip.h).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define L4D_INTERNAL extern "C" __attribute__((visibility("hidden")))  // (as in common.h: helpers the kernels' translation units call)

#include "../../include/lidar4d_hip.h"

static thread_local char g_err[512] = "";

L4D_INTERNAL void l4d_set_error(int code, const char* where) {
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

// L4D_TRACE=1 (debugging): every launch is announced on stderr and waited for, so that a faulting kernel is the last one named
static int g_trace = -1;
L4D_INTERNAL void l4d_trace_sync(const char* kernel, void* stream) {
  hipError_t e = hipStreamSynchronize((hipStream_t)stream);
  if (e == hipSuccess) e = hipGetLastError();
  fprintf(stderr, "[l4d] done   %s: %s\n", kernel, e == hipSuccess ? "ok" : hipGetErrorString(e));
  fflush(stderr);
}
L4D_INTERNAL int l4d_prof_begin(const char* kernel, void* stream) {
  if (g_trace < 0) {
    const char* e = getenv("L4D_TRACE");
    g_trace = (e && e[0] == '1') ? 1 : 0;
  }
  if (g_trace) {
    fprintf(stderr, "[l4d] launch %s\n", kernel);
    fflush(stderr);
    return -2;
  }
  if (!g_prof_on) return -1;
  ProfRec r;
  r.name = kernel;
  if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return -1;
  (void)hipEventRecord(r.start, (hipStream_t)stream);
  g_prof.push_back(r);
  return (int)g_prof.size() - 1;
}

L4D_INTERNAL void l4d_prof_end(int idx, void* stream) { (void)hipEventRecord(g_prof[idx].stop, (hipStream_t)stream); }

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

// ---- side streams: independent kernels of one call run concurrently -----------------------------------------------------
// A training step is a chain of kernels that each fill the chip but are bound by DIFFERENT resources (gather issue rate, VALU,
// LDS atomics, HBM streaming), and many of them do not depend on each other: the hash part of the field encode does not need
// the LDS evaluation of the xz / yz stacks until its last columns, the sorted scatter of the static grid does not need the
// plane adjoints ...  Such kernels are forked onto side streams of the launch stream (event record + wait: capturable into a
// hipGraph like any other stream dependency) and joined back before anything reads their results.
//   l4d_streams_config(mask)   bit 0: field encode forward, bit 1: field adjoint (l4d_density_encode_bwd's defer_join leaves the
//                              join to the caller, who overlaps the flow field's backward with the side streams)
//   l4d_streams_join(stream)   the launch stream waits for everything outstanding on the side streams
#include <stdlib.h>
#define L4D_N_SIDE 3
#define L4D_N_EVENTS 32
// one pool of side streams / events PER DEVICE (a process may drive several GPUs: a stream belongs to the device it was created on)
#define L4D_MAX_DEVICES 16
struct SidePool {
  hipStream_t side[L4D_N_SIDE];
  hipEvent_t events[L4D_N_EVENTS];
  bool ready, busy[L4D_N_SIDE];
  int event_next;
};
static SidePool g_pools[L4D_MAX_DEVICES];
static int g_streams_mask = -1;
static SidePool* cur_pool() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= L4D_MAX_DEVICES) return nullptr;
  return &g_pools[dev];
}
#define g_side (pool->side)
#define g_events (pool->events)
#define g_side_ready (pool->ready)
#define g_side_busy (pool->busy)
#define g_event_next (pool->event_next)

extern "C" int l4d_streams_mask(void) {
  // (set through l4d_streams_config only; default 0.  Measured, DESIGN.md section 4: the kernels of this path share their
  // bottlenecks; concurrency buys 0 +- 0.4 ms)
  if (g_streams_mask < 0) g_streams_mask = 0;
  return g_streams_mask;
}
extern "C" int l4d_streams_config(int32_t mask) {
  g_streams_mask = mask;
  return 0;
}
static int side_init(SidePool* pool) {
  if (!pool) { l4d_set_error(1, "side streams: no current device"); return 1; }
  if (g_side_ready) return 0;
  for (int i = 0; i < L4D_N_SIDE; ++i) {
    hipError_t e = hipStreamCreateWithFlags(&g_side[i], hipStreamNonBlocking);
    if (e != hipSuccess) { l4d_set_error((int)e, "side stream"); return (int)e; }
  }
  for (int i = 0; i < L4D_N_EVENTS; ++i) {
    hipError_t e = hipEventCreateWithFlags(&g_events[i], hipEventDisableTiming);
    if (e != hipSuccess) { l4d_set_error((int)e, "side event"); return (int)e; }
  }
  g_side_ready = true;
  return 0;
}
static hipEvent_t next_event(SidePool* pool) {
  hipEvent_t ev = g_events[g_event_next];
  g_event_next = (g_event_next + 1) % L4D_N_EVENTS;
  return ev;
}
// side stream i continues from the current end of `from` (main stream or another side stream); returns it (null on failure)
extern "C" void* l4d_side_fork(void* from, int32_t i) {
  SidePool* pool = cur_pool();
  if (i < 0 || i >= L4D_N_SIDE || side_init(pool)) return nullptr;
  hipEvent_t ev = next_event(pool);
  if (hipEventRecord(ev, (hipStream_t)from) != hipSuccess || hipStreamWaitEvent(g_side[i], ev, 0) != hipSuccess) {
    l4d_set_error(1, "l4d_side_fork");
    return nullptr;
  }
  g_side_busy[i] = true;
  return (void*)g_side[i];
}
// `into` waits for side stream i
extern "C" int l4d_side_join(void* into, int32_t i) {
  SidePool* pool = cur_pool();
  if (!pool || i < 0 || i >= L4D_N_SIDE || !g_side_ready || !g_side_busy[i]) return 0;
  hipEvent_t ev = next_event(pool);
  hipError_t e = hipEventRecord(ev, g_side[i]);
  if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)into, ev, 0);
  if (e != hipSuccess) { l4d_set_error((int)e, "l4d_side_join"); return (int)e; }
  g_side_busy[i] = false;
  return 0;
}
extern "C" int l4d_streams_join(void* stream) {
  for (int i = 0; i < L4D_N_SIDE; ++i) {
    int rc = l4d_side_join(stream, i);
    if (rc) return rc;
  }
  return 0;
}
