// libgcmi.so: version, per-thread error string, graph checks, optional kernel timing.
#include <atomic>
#include <cstdlib>
#include <stdarg.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace gcmi {

static thread_local char g_err[512] = "";

// Streaming kernels that walk rows in workgroup order take turns walking them forwards and backwards: a consumer
// launched right after its producer then starts on the rows the producer wrote LAST, which are the ones still in the
// 256 MB Infinity Cache (an N x 64 float array of the benchmark batch is 0.31 GB), instead of evicting them while it
// re-reads the oldest rows from HBM.
// GCMI_SWEEP: 0 = always forwards, 1 = the row-linear kernels alternate, 2 (default) = the window gathers take part
// too.  Same box, back to back: 4.73 / 4.67 / 4.66 ms per step.
static int sweep_mode() {
  static const int mode = getenv("GCMI_SWEEP") ? atoi(getenv("GCMI_SWEEP")) : 2;
  return mode;
}
static std::atomic<unsigned> g_sweep_counter{0};
int next_sweep_direction() {
  if (sweep_mode() == 0) return 0;
  return (int)(g_sweep_counter.fetch_add(1, std::memory_order_relaxed) & 1u);
}
int next_sweep_direction_windows() {
  if (sweep_mode() != 2) return 0;
  return (int)(g_sweep_counter.fetch_add(1, std::memory_order_relaxed) & 1u);
}

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_graph(const gcmi_graph* g, bool need_cols) {
  GCMI_CHECK_ARG(g != nullptr, "graph is NULL");
  GCMI_CHECK_ARG(g->max_deg >= 0 && g->max_deg <= GCMI_MAX_DEG, "max_deg %d outside [0,%d]",
                 g->max_deg, GCMI_MAX_DEG);
  GCMI_CHECK_ARG(g->n_atoms >= 0 && g->n_edges >= 0 && g->n_mols >= 0, "negative graph size");
  GCMI_CHECK_ARG(g->deg_start[0] == 0 && g->edge_start[0] == 0, "deg_start/edge_start must begin at 0");
  int64_t e = 0;
  for (int d = 0; d <= g->max_deg; ++d) {
    int64_t nd = (int64_t)g->deg_start[d + 1] - g->deg_start[d];
    GCMI_CHECK_ARG(nd >= 0, "deg_start not ascending at degree %d", d);
    GCMI_CHECK_ARG(g->edge_start[d] == e, "edge_start[%d]=%d, expected %lld", d, g->edge_start[d],
                   (long long)e);
    e += nd * d;
  }
  GCMI_CHECK_ARG(g->deg_start[g->max_deg + 1] == g->n_atoms, "deg_start[max_deg+1]=%d != n_atoms=%d",
                 g->deg_start[g->max_deg + 1], g->n_atoms);
  GCMI_CHECK_ARG(e == g->n_edges && g->edge_start[g->max_deg + 1] == g->n_edges,
                 "n_edges=%d inconsistent with the degree blocks (%lld)", g->n_edges, (long long)e);
  if (need_cols) GCMI_CHECK_ARG(g->n_edges == 0 || g->d_col_idx != nullptr, "d_col_idx is NULL");
  return GCMI_OK;
}

// ---------------------------------------------------------------- timing
struct KernelTimer {
  bool on = false;
  std::vector<hipEvent_t> pool;  // start,end,start,end ...
  size_t used = 0;               // events in use
};
static KernelTimer g_timers[GCMI_K_COUNT];
static std::mutex g_timer_mu;

void timing_begin(int id, hipStream_t s) {
  if (id < 0 || id >= GCMI_K_COUNT || !g_timers[id].on) return;
  std::lock_guard<std::mutex> lk(g_timer_mu);
  KernelTimer& t = g_timers[id];
  if (t.used + 2 > t.pool.size()) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
    t.pool.push_back(a);
    t.pool.push_back(b);
  }
  (void)hipEventRecord(t.pool[t.used], s);
}

void timing_end(int id, hipStream_t s) {
  if (id < 0 || id >= GCMI_K_COUNT || !g_timers[id].on) return;
  std::lock_guard<std::mutex> lk(g_timer_mu);
  KernelTimer& t = g_timers[id];
  if (t.used + 2 > t.pool.size()) return;
  (void)hipEventRecord(t.pool[t.used + 1], s);
  t.used += 2;
}

}  // namespace gcmi

extern "C" {

int gcmi_version(void) { return GCMI_VERSION; }

const char* gcmi_last_error(void) { return gcmi::g_err; }

int gcmi_timing_enable(int32_t id, int32_t on) {
  GCMI_CHECK_ARG(id >= 0 && id < GCMI_K_COUNT, "kernel id %d out of range", id);
  std::lock_guard<std::mutex> lk(gcmi::g_timer_mu);
  gcmi::g_timers[id].on = on != 0;
  return GCMI_OK;
}

int gcmi_timing_read(int32_t id, int64_t* n_launches, double* total_ms, int32_t reset) {
  GCMI_CHECK_ARG(id >= 0 && id < GCMI_K_COUNT, "kernel id %d out of range", id);
  GCMI_CHECK_ARG(n_launches && total_ms, "NULL output");
  std::lock_guard<std::mutex> lk(gcmi::g_timer_mu);
  gcmi::KernelTimer& t = gcmi::g_timers[id];
  double ms = 0;
  for (size_t i = 0; i + 1 < t.used; i += 2) {
    if (hipEventSynchronize(t.pool[i + 1]) != hipSuccess) {
      gcmi::set_error("hipEventSynchronize failed");
      return GCMI_ERR_LAUNCH;
    }
    float e = 0;
    if (hipEventElapsedTime(&e, t.pool[i], t.pool[i + 1]) == hipSuccess) ms += e;
  }
  *n_launches = (int64_t)(t.used / 2);
  *total_ms = ms;
  if (reset) t.used = 0;
  return GCMI_OK;
}

}  // extern "C"
