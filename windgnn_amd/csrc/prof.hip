// Optional per-kernel timing for bench.py's roofline: hipEvent pairs recorded on the launch stream
// around each kernel while profiling is enabled.  Debug facility: it is the only global mutable
// state in the library, is off by default, and is not thread-safe.
#include <string.h>

#include <string>
#include <vector>

#include "common.h"

bool g_prof_on = false;

namespace {
struct Rec { int kind; hipEvent_t a, b; };
struct Kind { std::string name; double flops = 0, bytes = 0, ms = 0; int64_t n = 0; };
std::vector<Rec> g_recs;
std::vector<Kind> g_kinds;
std::vector<hipEvent_t> g_pool;
size_t g_pool_used = 0;
bool g_folded = true;

hipEvent_t get_event() {
  if (g_pool_used == g_pool.size()) {
    hipEvent_t e;
    (void)hipEventCreate(&e);
    g_pool.push_back(e);
  }
  return g_pool[g_pool_used++];
}

int kind_of(const char* name) {
  for (size_t i = 0; i < g_kinds.size(); ++i)
    if (g_kinds[i].name == name) return (int)i;
  Kind k;
  k.name = name;
  g_kinds.push_back(k);
  return (int)g_kinds.size() - 1;
}

void fold() {
  if (g_folded) return;
  (void)hipDeviceSynchronize();
  for (const Rec& r : g_recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) g_kinds[r.kind].ms += ms;
  }
  g_recs.clear();
  g_pool_used = 0;
  g_folded = true;
}
}  // namespace

void prof_begin(const char* kernel, double flops, double bytes, hipStream_t st) {
  Rec r;
  r.kind = kind_of(kernel);
  Kind& k = g_kinds[r.kind];
  k.flops += flops;
  k.bytes += bytes;
  k.n += 1;
  r.a = get_event();
  r.b = get_event();
  (void)hipEventRecord(r.a, st);
  g_recs.push_back(r);
  g_folded = false;
}

void prof_end(hipStream_t st) { (void)hipEventRecord(g_recs.back().b, st); }

extern "C" {

int wgnn_profile_enable(int on) {
  if (on) {
    fold();
    g_kinds.clear();
  }
  g_prof_on = on != 0;
  return WGNN_OK;
}

int wgnn_profile_read(int idx, char* name, size_t name_len, double* total_ms, int64_t* launches, double* flops,
                      double* bytes) {
  fold();
  if (idx < 0 || idx >= (int)g_kinds.size()) return WGNN_ERR_SHAPE;
  const Kind& k = g_kinds[idx];
  if (name && name_len > 0) {
    strncpy(name, k.name.c_str(), name_len - 1);
    name[name_len - 1] = 0;
  }
  if (total_ms) *total_ms = k.ms;
  if (launches) *launches = k.n;
  if (flops) *flops = k.flops;
  if (bytes) *bytes = k.bytes;
  return WGNN_OK;
}
}
