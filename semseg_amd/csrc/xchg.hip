// SyncBN exchange over peer-mapped memory: the all-reduce of a small fp64 vector ([2C] statistics of one BatchNorm layer, or
// the staging vector of a SyncGroup) among the GPUs of one node in ONE kernel per rank, instead of a c10d / RCCL all-reduce
// (tool/train.py:142: nn.SyncBatchNorm does one exchange per layer and pass; 208 per PSPNet-101 step here).  The payload is
// <= 64 KB and the exchange sits on the critical path of every layer, so what matters is latency: an RCCL all-reduce is a
// host round trip through c10d plus a ring of W-1 hops per chunk; here every rank WRITES its vector straight into a slot
// of every peer's exchange buffer (one xGMI hop, all peers in parallel), raises a flag behind it, waits for the W flags of
// its own buffer and sums the W slots in rank order — the same order on every rank, so the result is bit-identical on all
// of them (the replicas must stay identical: nothing broadcasts parameters or running statistics afterwards).
//
// Memory: every rank owns one buffer in fine-grained (uncached) device memory — peer stores must become visible while the
// consumer's kernel is running, which ordinary (coarse-grained) hipMalloc memory only guarantees at kernel boundaries —
// laid out [2 parities][W slots][XCHG_MAX doubles] + [2][W] 64-bit flags, exported with hipIpcGetMemHandle and mapped by
// all peers.  Exchange number `seq` (1, 2, ...; the same call sequence on every rank) uses parity seq & 1: a rank can run at
// most one exchange ahead of its slowest peer (it needs that peer's flag of exchange s to finish s, and the peer only
// raises it after it has finished s - 1), so two slot sets are enough and nothing is ever reset.
// Ordering: data stores, then __threadfence_system() in every storing thread, then a workgroup barrier, then the flag
// stores (system-scope release); the consumer polls its flags with system-scope acquire loads (s_sleep between polls),
// then reads the slots with system-scope loads.  Every spin is bounded: after 20 s the kernel gives up, sets *err and
// finishes with whatever it has — the host raises (SyncExchange.check()).
//
// STATUS (round 5): exercised with two and four processes on ONE GPU (tests/test_dist_gpu.py, the ranks' buffers in the same
// HBM, IPC-mapped across processes); it has NOT run across xGMI.  The host side (semseg_amd/syncbn_xchg.py) therefore takes it
// only after a start-up self-test among the real peers — 64 exchanges of known vectors with a 2 s bound, every rank's result
// checked, the verdict agreed by all ranks — and falls back to RCCL otherwise (SEMSEG_SYNCBN_XCHG=auto, the default).
#include <cstring>

#include "common.h"
#include "../../include/semseg_hip.h"

namespace {

constexpr int XCHG_MAX = SEMSEG_XCHG_MAX_DOUBLES;   // doubles per slot
constexpr int XCHG_MAX_WORLD = 8;

struct XchgArgs {
  double* peer[XCHG_MAX_WORLD];   // base of peer p's exchange buffer as mapped in this process (peer[rank] = own buffer)
  const double* in;               // [nslot][n]
  double* out;                    // [n]
  int* err;
  unsigned long long seq;
  unsigned long long* seq_dev;      // non-null: the exchange number lives in device memory (seq = *seq_dev + 1, written back at the end)
  unsigned long long limit_ticks;   // bound of the flag wait in ticks of the 100 MHz constant clock
  int world, rank, nslot, n;
};

__device__ __forceinline__ double* slot_of(double* base, int world, int parity, int who) {
  return base + ((size_t)parity * world + who) * XCHG_MAX;
}
__device__ __forceinline__ unsigned long long* flags_of(double* base, int world, int parity) {
  return reinterpret_cast<unsigned long long*>(base + (size_t)2 * world * XCHG_MAX) + (size_t)parity * world;
}

__global__ __launch_bounds__(1024) void xchg_allreduce_kernel(const XchgArgs p) {
  // Exchange number from device memory (round 5): every rank's counter advances by one per exchange, in the same call
  // sequence, so the launch has no per-call host argument and a recorded step plan can replay it (csrc/plan.hip).  Every
  // thread reads the counter before the first barrier; thread 0 writes it back after the last one.  (The argument struct stays
  // const: a modified copy of it moved the dynamically indexed peer[] array into scratch memory — 7 us per exchange.)
  const unsigned long long seq = p.seq_dev ? p.seq_dev[0] + 1ull : p.seq;
  const int tid = threadIdx.x, W = p.world, par = (int)(seq & 1ull);
  // (a) fold the slot replicas of the local vector, (b) store it into slot [rank] of every peer's buffer
  for (int i = tid; i < p.n; i += blockDim.x) {
    double v = p.in[i];
    for (int s = 1; s < p.nslot; ++s) v += p.in[(size_t)s * p.n + i];
    for (int q = 0; q < W; ++q) {
      const int dst = (p.rank + q) % W;           // start with the own buffer, spread the peers over the links
      __hip_atomic_store(slot_of(p.peer[dst], W, par, p.rank) + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  __threadfence_system();
  __syncthreads();
  // (c) raise this rank's flag in every peer's buffer
  if (tid < W)
    __hip_atomic_store(flags_of(p.peer[(p.rank + tid) % W], W, par) + p.rank, seq, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  // (d) wait for every rank's flag in the own buffer (bounded)
  __shared__ int bad;
  if (tid == 0) bad = 0;
  __syncthreads();
  if (tid < W) {
    const unsigned long long* f = flags_of(p.peer[p.rank], W, par) + tid;
    // 100 MHz constant clock.  The bound has to cover honest skew between the ranks' HOSTS (a peer's exchange kernel is only
    // launched when its python thread gets there: first-use code-object loads, a rank whose process was scheduled late — a
    // 1 s bound fired once in the two-processes-on-one-GPU test), so it is 20 s; an exchange that already failed in this
    // process makes the following ones give up at once instead of waiting 20 s each.
    const unsigned long long t0 = wall_clock64(), limit = *p.err ? 0ull : p.limit_ticks;
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
      __builtin_amdgcn_s_sleep(16);
      if (wall_clock64() - t0 > limit) {      // a peer never arrived (crashed, different call sequence, not co-resident)
        bad = 1;
        break;
      }
    }
  }
  __syncthreads();
  if (bad && tid == 0) *p.err = 1;
  // (e) sum the W slots in rank order: identical on every rank
  double* mine = p.peer[p.rank];
  for (int i = tid; i < p.n; i += blockDim.x) {
    double v = __hip_atomic_load(slot_of(mine, W, par, 0) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    for (int q = 1; q < W; ++q)
      v += __hip_atomic_load(slot_of(mine, W, par, q) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    p.out[i] = v;
  }
  if (p.seq_dev && tid == 0) p.seq_dev[0] = seq;      // ordered after every thread's read by the barriers above
}

}  // namespace

extern "C" {

size_t semseg_xchg_buffer_bytes(int world) {
  if (world < 1 || world > XCHG_MAX_WORLD) return 0;
  return ((size_t)2 * world * XCHG_MAX + (size_t)2 * world) * 8;
}

int semseg_xchg_alloc(int world, void** ptr) {
  const size_t bytes = semseg_xchg_buffer_bytes(world);
  if (!ptr || bytes == 0) return SEMSEG_EINVAL;
  void* p = nullptr;
  // uncached fine-grained device memory (what RCCL uses for its own flag / staging buffers on this family)
  if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) != hipSuccess) {
    (void)hipGetLastError();
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
      (void)hipGetLastError();
      return SEMSEG_ELAUNCH;
    }
  }
  if (hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(p);
    return SEMSEG_ELAUNCH;
  }
  *ptr = p;
  return SEMSEG_OK;
}

int semseg_xchg_free(void* ptr) { return (!ptr || hipFree(ptr) == hipSuccess) ? SEMSEG_OK : SEMSEG_ELAUNCH; }

int semseg_xchg_ipc_export(void* ptr, void* handle64) {
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
  if (!ptr || !handle64) return SEMSEG_EINVAL;
  if (hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(handle64), ptr) != hipSuccess) {
    (void)hipGetLastError();
    return SEMSEG_ELAUNCH;
  }
  return SEMSEG_OK;
}

int semseg_xchg_ipc_import(const void* handle64, void** ptr) {
  if (!handle64 || !ptr) return SEMSEG_EINVAL;
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, 64);
  if (hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
    (void)hipGetLastError();
    return SEMSEG_ELAUNCH;
  }
  return SEMSEG_OK;
}

int semseg_xchg_ipc_close(void* ptr) { return (!ptr || hipIpcCloseMemHandle(ptr) == hipSuccess) ? SEMSEG_OK : SEMSEG_ELAUNCH; }

int semseg_xchg_allreduce_f64(const double* in, int nslot, int n, double* out, void* const* peer_bases, int world, int rank,
                              unsigned long long seq, unsigned long long* seq_dev, int* err_dev, int timeout_ms,
                              hipStream_t stream) {
  if (!in || !out || !peer_bases || !err_dev || world < 1 || world > XCHG_MAX_WORLD || rank < 0 || rank >= world ||
      nslot < 1 || n < 1 || n > XCHG_MAX || (seq == 0 && !seq_dev))
    return SEMSEG_EINVAL;
  XchgArgs a;
  for (int q = 0; q < XCHG_MAX_WORLD; ++q) a.peer[q] = q < world ? static_cast<double*>(peer_bases[q]) : nullptr;
  for (int q = 0; q < world; ++q)
    if (!a.peer[q]) return SEMSEG_EINVAL;
  a.in = in; a.out = out; a.err = err_dev; a.seq = seq; a.seq_dev = seq_dev;
  a.limit_ticks = (unsigned long long)(timeout_ms > 0 ? timeout_ms : 20000) * 100000ull; a.world = world; a.rank = rank; a.nslot = nslot; a.n = n;
  xchg_allreduce_kernel<<<1, 1024, 0, stream>>>(a);
  return semseg_launch_status();
}

}  // extern "C"
