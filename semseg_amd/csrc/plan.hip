// Step plan: launch sequencing BELOW the C ABI (round 5).
//
// The train step of the reference (tool/train.py:269-276: forward, loss, backward, optimizer step) is ~1 200 launches of the
// entry points of include/semseg_hip.h on two or three HIP streams.  Their arguments — device pointers of buffers the engine
// owns for its whole life, shapes, tile codes, stream handles — are the same every step, so the sequence is RECORDED once
// (the host driver appends every call it makes: entry-point id + a row of 64-bit argument slots) and afterwards REPLAYED
// from here: one C loop over call thunks (plan_thunks.inc, generated from the header), or — where the recorded range holds
// no collective — one hipGraph captured from that loop.  What changes from step to step (learning rates, the dropout
// counter) lives in device memory written by semseg_step_state_set before the replay; inputs are copied into buffers the
// driver owns.  Cross-stream ordering (data-gradient chain vs weight-gradient side stream) is part of the record:
// semseg_stream_wait_stream is an entry point like any other, and under capture its event pair becomes a graph edge.
//
// A plan holds no device memory and launches nothing by itself; destroying it frees its graphs.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"
#include "../../include/semseg_hip.h"

namespace {

inline float slot_f32(unsigned long long s) {
  const uint32_t u = (uint32_t)s;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}
inline double slot_f64(unsigned long long s) {
  double d;
  std::memcpy(&d, &s, 8);
  return d;
}

struct PlanThunk {
  const char* name;
  int (*call)(const unsigned long long*);
  int nargs;
};

#include "plan_thunks.inc"

constexpr int N_THUNKS = (int)(sizeof(PLAN_THUNKS) / sizeof(PLAN_THUNKS[0]));

struct Entry {
  int fn, nargs;
  size_t off;        // first slot in Plan::slots
};

struct Plan {
  uint32_t magic = 0x504c414e;   // "PLAN"
  std::vector<Entry> entries;
  std::vector<unsigned long long> slots;
  std::vector<hipGraphExec_t> graphs;
  std::vector<int> graph_nodes;
  int failed = -1;
};

inline Plan* as_plan(void* p) {
  Plan* pl = static_cast<Plan*>(p);
  return (pl && pl->magic == 0x504c414e) ? pl : nullptr;
}

// SEMSEG_PLAN_DEBUG=1: progress of a capture on stderr (which entry the runtime was in when something went wrong)
bool plan_debug() {
  static const bool on = [] { const char* v = std::getenv("SEMSEG_PLAN_DEBUG"); return v && v[0] == '1'; }();
  return on;
}

int replay(Plan* pl, int first, int last, bool trace = false) {
  for (int i = first; i < last; ++i) {
    const Entry& e = pl->entries[i];
    if (trace) {
      std::fprintf(stderr, "[plan] entry %d %s\n", i, PLAN_THUNKS[e.fn].name);
      std::fflush(stderr);
    }
    const int rc = PLAN_THUNKS[e.fn].call(pl->slots.data() + e.off);
    if (rc != SEMSEG_OK) {
      pl->failed = i;
      return rc;
    }
  }
  return SEMSEG_OK;
}

// Events of semseg_stream_wait_stream: a small per-thread ring.  hipStreamWaitEvent takes the event's state at the time of the
// call, so a ring slot may be re-recorded as soon as the wait has been enqueued; the ring only keeps the pool bounded.
struct EventRing {
  static constexpr int N = 32;
  hipEvent_t ev[N] = {};
  int next = 0;
  hipEvent_t get() {
    hipEvent_t& e = ev[next];
    next = (next + 1) % N;
    if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    return e;
  }
};
thread_local EventRing g_events;

// Under stream capture every wait gets an event of its own (never re-recorded inside one capture): the captured
// dependency is then a plain record-node -> wait edge, whatever the runtime does with re-recorded events.  They are kept
// for the life of the process (a capture makes a few hundred).
struct CaptureEvents {
  std::vector<hipEvent_t> ev;
  hipEvent_t fresh() {
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    ev.push_back(e);
    return e;
  }
};
thread_local CaptureEvents g_capture_events;

__global__ void step_state_kernel(float* lr2, float lr, float lr_head, unsigned long long* drop, unsigned long long off) {
  if (lr2) {
    lr2[0] = lr;
    lr2[1] = lr_head;
  }
  if (drop) drop[0] = off;
}

// One launch in front of a replayed step: the caller's input batch and targets into the buffers the record points at (any
// byte count; 16-byte body + byte tail) and the per-step values into device memory.  A source / destination pair that is not
// 16-byte aligned (a contiguous slice of a larger device batch: 3 * 473 * 473 * 4 bytes per image is 12 mod 16) is copied in
// dwords, one that is not even 4-byte aligned in bytes: the eager path has no alignment requirement either.
__device__ __forceinline__ void copy_any(unsigned char* d, const unsigned char* s, size_t nb, size_t tid, size_t nth) {
  const uintptr_t a = (uintptr_t)d | (uintptr_t)s;
  if ((a & 15) == 0) {
    const size_t n16 = nb >> 4;
    for (size_t i = tid; i < n16; i += nth) reinterpret_cast<uint4*>(d)[i] = reinterpret_cast<const uint4*>(s)[i];
    for (size_t i = (n16 << 4) + tid; i < nb; i += nth) d[i] = s[i];
  } else if ((a & 3) == 0) {
    const size_t n4 = nb >> 2;
    for (size_t i = tid; i < n4; i += nth) reinterpret_cast<uint32_t*>(d)[i] = reinterpret_cast<const uint32_t*>(s)[i];
    for (size_t i = (n4 << 2) + tid; i < nb; i += nth) d[i] = s[i];
  } else {
    for (size_t i = tid; i < nb; i += nth) d[i] = s[i];
  }
}
__global__ __launch_bounds__(256) void step_begin_kernel(unsigned char* xd, const unsigned char* xs, size_t xb, unsigned char* yd,
                                                         const unsigned char* ys, size_t yb, float* lr2, float lr, float lr_head,
                                                         unsigned long long* drop, unsigned long long off) {
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
  if (tid == 0) {
    if (lr2) {
      lr2[0] = lr;
      lr2[1] = lr_head;
    }
    if (drop) drop[0] = off;
  }
  copy_any(xd, xs, xb, tid, nth);
  copy_any(yd, ys, yb, tid, nth);
}

}  // namespace

extern "C" {

int semseg_plan_create(void** plan) {
  if (!plan) return SEMSEG_EINVAL;
  *plan = new Plan();
  return SEMSEG_OK;
}

int semseg_plan_destroy(void* plan) {
  Plan* pl = as_plan(plan);
  if (!pl) return SEMSEG_EINVAL;
  for (hipGraphExec_t g : pl->graphs)
    if (g) (void)hipGraphExecDestroy(g);
  pl->magic = 0;
  delete pl;
  return SEMSEG_OK;
}

int semseg_plan_fn_id(const char* name) {
  if (!name) return SEMSEG_EINVAL;
  for (int i = 0; i < N_THUNKS; ++i)
    if (std::strcmp(PLAN_THUNKS[i].name, name) == 0) return i;
  return SEMSEG_EINVAL;
}

int semseg_plan_fn_nargs(int fn_id) {
  if (fn_id < 0 || fn_id >= N_THUNKS) return SEMSEG_EINVAL;
  return PLAN_THUNKS[fn_id].nargs;
}

int semseg_plan_append(void* plan, int fn_id, int nargs, const unsigned long long* slots) {
  Plan* pl = as_plan(plan);
  if (!pl || fn_id < 0 || fn_id >= N_THUNKS || nargs != PLAN_THUNKS[fn_id].nargs || (nargs > 0 && !slots)) return SEMSEG_EINVAL;
  Entry e;
  e.fn = fn_id;
  e.nargs = nargs;
  e.off = pl->slots.size();
  pl->slots.insert(pl->slots.end(), slots, slots + nargs);
  pl->entries.push_back(e);
  return (int)pl->entries.size() - 1;
}

int semseg_plan_size(void* plan) {
  Plan* pl = as_plan(plan);
  return pl ? (int)pl->entries.size() : SEMSEG_EINVAL;
}

int semseg_plan_entry_fn(void* plan, int entry) {
  Plan* pl = as_plan(plan);
  if (!pl || entry < 0 || entry >= (int)pl->entries.size()) return SEMSEG_EINVAL;
  return pl->entries[entry].fn;
}

int semseg_plan_set_slot(void* plan, int entry, int arg, unsigned long long bits) {
  Plan* pl = as_plan(plan);
  if (!pl || entry < 0 || entry >= (int)pl->entries.size() || arg < 0 || arg >= pl->entries[entry].nargs) return SEMSEG_EINVAL;
  pl->slots[pl->entries[entry].off + arg] = bits;
  return SEMSEG_OK;
}

int semseg_plan_get_slot(void* plan, int entry, int arg, unsigned long long* bits) {
  Plan* pl = as_plan(plan);
  if (!pl || !bits || entry < 0 || entry >= (int)pl->entries.size() || arg < 0 || arg >= pl->entries[entry].nargs) return SEMSEG_EINVAL;
  *bits = pl->slots[pl->entries[entry].off + arg];
  return SEMSEG_OK;
}

// 0 when the two plans hold the same calls with the same argument slots, ignoring argument ignore_arg of entry point
// ignore_fn (< 0: nothing ignored); otherwise 1 and where[0] = first differing entry, where[1] = argument (-1: a different
// entry point or entry count).
int semseg_plan_compare(void* plan_a, void* plan_b, int ignore_fn, int ignore_arg, int* where) {
  Plan* a = as_plan(plan_a);
  Plan* b = as_plan(plan_b);
  if (!a || !b || !where) return SEMSEG_EINVAL;
  const size_t n = a->entries.size() < b->entries.size() ? a->entries.size() : b->entries.size();
  for (size_t i = 0; i < n; ++i) {
    const Entry& x = a->entries[i];
    const Entry& y = b->entries[i];
    if (x.fn != y.fn || x.nargs != y.nargs) {
      where[0] = (int)i; where[1] = -1;
      return 1;
    }
    for (int k = 0; k < x.nargs; ++k)
      if (a->slots[x.off + k] != b->slots[y.off + k] && !(x.fn == ignore_fn && k == ignore_arg)) {
        where[0] = (int)i; where[1] = k;
        return 1;
      }
  }
  if (a->entries.size() != b->entries.size()) {
    where[0] = (int)n; where[1] = -1;
    return 1;
  }
  return 0;
}

int semseg_plan_replay(void* plan, int first, int last) {
  Plan* pl = as_plan(plan);
  if (!pl || first < 0 || last < first || last > (int)pl->entries.size()) return SEMSEG_EINVAL;
  pl->failed = -1;
  return replay(pl, first, last);
}

int semseg_plan_failed_entry(void* plan) {
  Plan* pl = as_plan(plan);
  return pl ? pl->failed : SEMSEG_EINVAL;
}

// Captures the replay of entries [first, last) into one executable hipGraph owned by the plan.  `origin` must be a
// non-default stream every other stream of the range is forked from (and joined back into) through semseg_stream_wait_stream.
// Relaxed capture mode: other host threads (a data loader pinning memory, ...) stay free to call the runtime.
int semseg_plan_graph_capture(void* plan, int first, int last, hipStream_t origin) {
  Plan* pl = as_plan(plan);
  if (!pl || !origin || first < 0 || last < first || last > (int)pl->entries.size()) return SEMSEG_EINVAL;
  pl->failed = -1;
  const bool dbg = plan_debug();
  hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
  if (const char* m = std::getenv("SEMSEG_PLAN_CAPTURE_MODE")) {      // measurement / diagnosis only
    if (m[0] == '0') mode = hipStreamCaptureModeGlobal;
    if (m[0] == '1') mode = hipStreamCaptureModeThreadLocal;
  }
  if (dbg) std::fprintf(stderr, "[plan] begin capture of entries [%d, %d) mode %d\n", first, last, (int)mode);
  if (hipStreamBeginCapture(origin, mode) != hipSuccess) return SEMSEG_ELAUNCH;
  const int rc = replay(pl, first, last, dbg);
  hipGraph_t graph = nullptr;
  if (dbg) std::fprintf(stderr, "[plan] end capture (replay rc %d, failed entry %d)\n", rc, pl->failed);
  const hipError_t e = hipStreamEndCapture(origin, &graph);
  if (dbg) std::fprintf(stderr, "[plan] hipStreamEndCapture -> %d (%s)\n", (int)e, hipGetErrorString(e));
  if (rc != SEMSEG_OK || e != hipSuccess || !graph) {
    if (graph) (void)hipGraphDestroy(graph);
    (void)hipGetLastError();
    return rc != SEMSEG_OK ? rc : SEMSEG_ELAUNCH;
  }
  size_t nodes = 0;
  if (hipGraphGetNodes(graph, nullptr, &nodes) != hipSuccess) nodes = 0;
  if (dbg) std::fprintf(stderr, "[plan] graph of %zu nodes, instantiating\n", nodes);
  hipGraphExec_t exec = nullptr;
  const hipError_t ei = hipGraphInstantiateWithFlags(&exec, graph, 0);
  if (dbg) std::fprintf(stderr, "[plan] hipGraphInstantiateWithFlags -> %d (%s)\n", (int)ei, hipGetErrorString(ei));
  (void)hipGraphDestroy(graph);
  if (ei != hipSuccess || !exec) {
    (void)hipGetLastError();
    return SEMSEG_ELAUNCH;
  }
  pl->graphs.push_back(exec);
  pl->graph_nodes.push_back((int)nodes);
  return (int)pl->graphs.size() - 1;
}

int semseg_plan_graph_launch(void* plan, int graph, hipStream_t stream) {
  Plan* pl = as_plan(plan);
  if (!pl || graph < 0 || graph >= (int)pl->graphs.size() || !pl->graphs[graph]) return SEMSEG_EINVAL;
  return hipGraphLaunch(pl->graphs[graph], stream) == hipSuccess ? SEMSEG_OK : SEMSEG_ELAUNCH;
}

int semseg_plan_graph_nodes(void* plan, int graph) {
  Plan* pl = as_plan(plan);
  if (!pl || graph < 0 || graph >= (int)pl->graphs.size()) return SEMSEG_EINVAL;
  return pl->graph_nodes[graph];
}

// waiter: everything enqueued on it after this call runs after everything enqueued on `signaller` before this call.
int semseg_stream_wait_stream(hipStream_t waiter, hipStream_t signaller) {
  if (waiter == signaller) return SEMSEG_OK;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  const bool capturing = signaller && hipStreamIsCapturing(signaller, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive;
  hipEvent_t ev = capturing ? g_capture_events.fresh() : g_events.get();
  if (!ev) return SEMSEG_ELAUNCH;
  if (hipEventRecord(ev, signaller) != hipSuccess) return SEMSEG_ELAUNCH;
  if (hipStreamWaitEvent(waiter, ev, 0) != hipSuccess) return SEMSEG_ELAUNCH;
  return SEMSEG_OK;
}

int semseg_step_state_set(float* lr_dev2, float lr, float lr_head, unsigned long long* drop_dev,
                          unsigned long long drop_offset, hipStream_t stream) {
  if (!lr_dev2 && !drop_dev) return SEMSEG_EINVAL;
  step_state_kernel<<<1, 1, 0, stream>>>(lr_dev2, lr, lr_head, drop_dev, drop_offset);
  return semseg_launch_status();
}

int semseg_step_begin(void* x_dst, const void* x_src, size_t x_bytes, void* y_dst, const void* y_src, size_t y_bytes,
                      float* lr_dev2, float lr, float lr_head, unsigned long long* drop_dev, unsigned long long drop_offset,
                      hipStream_t stream) {
  if ((x_bytes && (!x_dst || !x_src)) || (y_bytes && (!y_dst || !y_src))) return SEMSEG_EINVAL;
  const size_t chunks = ((x_bytes > y_bytes ? x_bytes : y_bytes) >> 4) + 1;
  size_t grid = (chunks + 255) / 256;
  if (grid > 2048) grid = 2048;
  step_begin_kernel<<<(int)grid, 256, 0, stream>>>(static_cast<unsigned char*>(x_dst), static_cast<const unsigned char*>(x_src), x_bytes,
                                                   static_cast<unsigned char*>(y_dst), static_cast<const unsigned char*>(y_src), y_bytes,
                                                   lr_dev2, lr, lr_head, drop_dev, drop_offset);
  return semseg_launch_status();
}

// Host-only probe of the slot encoding (no device work): host_out[0..5] = the arguments as the callee saw them.  Lets the
// record -> append -> patch -> replay machinery be tested on a machine without a GPU.
int semseg_host_probe(unsigned long long* host_out, int a, long long b, size_t c, float d, double e, hipStream_t stream) {
  if (!host_out) return SEMSEG_EINVAL;
  host_out[0] = (unsigned long long)(long long)a;
  host_out[1] = (unsigned long long)b;
  host_out[2] = (unsigned long long)c;
  uint32_t u;
  std::memcpy(&u, &d, 4);
  host_out[3] = u;
  std::memcpy(&host_out[4], &e, 8);
  host_out[5] = (unsigned long long)(uintptr_t)stream;
  host_out[6] += 1;     // call counter
  return a == -12345 ? SEMSEG_EINVAL : SEMSEG_OK;      // a way to make a replay fail at a known entry
}

}  // extern "C"
