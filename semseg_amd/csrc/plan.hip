// Step plan: launch sequencing BELOW the C ABI (round 5).
//
// The train step of the reference (tool/train.py:269-276: forward, loss, backward, optimizer step) is ~1 200 launches of the
// entry points of include/semseg_hip.h on two or three HIP streams.  Their arguments — device pointers of buffers the engine
// owns for its whole life, shapes, tile codes, stream handles — are the same every step, so the sequence is RECORDED once
// (the host driver appends every call it makes: entry-point id + a row of 64-bit argument slots) and afterwards REPLAYED
// from here: one C loop over call thunks (plan_thunks.inc, generated from the header).  (Rounds 5-6 could also capture that loop
// into one hipGraph: measured twice, +5-10 % and +8-46 % on the device step — a graph's branches run on internal streams that
// know nothing of the high-priority data-gradient chain — and removed in round 6; profiles/r06_host_issue_b2.json.)
// What changes from step to step (learning rates, the dropout
// counter) lives in device memory written by semseg_step_state_set before the replay; inputs are copied into buffers the
// driver owns.  Cross-stream ordering (data-gradient chain vs weight-gradient side stream) is part of the record:
// semseg_stream_wait_stream is an entry point like any other.
//
// A plan holds no device memory and launches nothing by itself.
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
  int failed = -1;
};

inline Plan* as_plan(void* p) {
  Plan* pl = static_cast<Plan*>(p);
  return (pl && pl->magic == 0x504c414e) ? pl : nullptr;
}

int replay(Plan* pl, int first, int last) {
  for (int i = first; i < last; ++i) {
    const Entry& e = pl->entries[i];
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

// waiter: everything enqueued on it after this call runs after everything enqueued on `signaller` before this call.
int semseg_stream_wait_stream(hipStream_t waiter, hipStream_t signaller) {
  if (waiter == signaller) return SEMSEG_OK;
  hipEvent_t ev = g_events.get();
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
