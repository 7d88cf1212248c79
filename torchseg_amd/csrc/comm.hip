// Collectives of the hot path on RCCL, enqueued on the CALLER's stream.
//
// Replaces the exchange steps of the reference's SyncBN / DDP:
//   furnace/legacy/sync_bn/syncbn.py:75-78  (ReduceAddCoalesced + Broadcast of [sum, sum^2] across devices),
//   furnace/legacy/sync_bn/comm.py:57-132   (the master/slave queue pipes that carry them),
//   apex.parallel.SyncBatchNorm / DistributedDataParallel (train.py:24-25,98-99): torch.distributed all_reduce /
//   all_gather / broadcast on NCCL.
//
// Why not torch.distributed for the SyncBN statistics: a BiSeNet step issues 210 tiny all-reduces (2C+2 floats each).
// ProcessGroupNCCL runs every collective on its own stream, i.e. two event record/wait handshakes per collective and a
// Python -> C++ -> Work-object round trip; measured 1.9 ms per step on a 1-rank group where the wire time is zero
// (DESIGN.md section 6).  Here the collective is one ncclAllReduce call on the compute stream between the kernel that
// produces the message and the kernel that consumes it: no handshake, no allocation, capturable in a hipGraph (the
// mailbox path below as well: its call counter lives in device memory and is advanced by the kernel itself, so a
// replayed launch sees a fresh sequence number — tests/test_comm_gpu.py::test_mailbox_under_graph_replay).
//
// librccl is resolved at run time (dlopen) so that libtsg_hip.so has no link-time dependency on it and shares the
// copy that PyTorch already loaded (two RCCL copies in one process would each grab the xGMI topology).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "tsg_common.h"

namespace {

// the slice of rccl.h this file needs (ABI-stable since NCCL 2.x; /opt/rocm/include/rccl/rccl.h:40-43,187,220,260,448-468,591-678)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { kNcclSum = 0 };
enum { kNcclFloat32 = 7, kNcclBfloat16 = 9, kNcclUint8 = 1 };

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*ReduceScatter)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;                 // immutable once loaded
std::mutex g_load_mutex;

int load_rccl(const char* path) {
  std::lock_guard<std::mutex> lock(g_load_mutex);
  if (g_rccl.handle) return 0;
  void* h = nullptr;
  if (path && *path) h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
  if (!h) {
    // the copy already in the process (PyTorch's), then the system one
    static const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (h) break; }
    if (!h) for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  }
  if (!h) return TSG_E_COMM_LIB;
  Rccl r;
  r.handle = h;
#define SYM(field, name) *(void**)(&r.field) = dlsym(h, name); if (!r.field) return TSG_E_COMM_LIB
  SYM(GetUniqueId, "ncclGetUniqueId");
  SYM(CommInitRank, "ncclCommInitRank");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(AllReduce, "ncclAllReduce");
  SYM(AllGather, "ncclAllGather");
  SYM(Broadcast, "ncclBroadcast");
  SYM(ReduceScatter, "ncclReduceScatter");
  SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  g_rccl = r;
  return 0;
}

int nccl_type(int dtype) {
  switch (dtype) {
    case TSG_F32: return kNcclFloat32;
    case TSG_BF16: return kNcclBfloat16;
    default: return -1;
  }
}

// RCCL error -> our convention: positive codes are hipError_t, so RCCL failures get their own range
inline int rc(ncclResult_t r) { return r == 0 ? 0 : TSG_E_COMM_BASE - r; }

}  // namespace

// ---- one-shot small-message all-reduce over peer-mapped mailboxes ---------------------
// Every rank owns a mailbox in its HBM: [2 parities][world slots][cap floats] + [2][world] sequence flags.
// An all-reduce of n <= cap floats is ONE kernel per rank on the compute stream:
//   1. write my n floats into slot[parity][my_rank] of EVERY rank's mailbox (xGMI peer stores, own copy included),
//   2. release-store the call's sequence number into flag[parity][my_rank] of every mailbox,
//   3. acquire-spin until my own mailbox shows the sequence number from all ranks,
//   4. sum the slots in rank order (identical order on every rank => bit-identical replicas) into buf.
// One xGMI hop of latency, no RCCL kernel, no proxy thread.  The parity alternates per call: a rank can only be two
// calls ahead of a peer after that peer has passed step 2 of the call in between, i.e. finished reading the older
// parity, so two buffers suffice.  Mailboxes are uncached device memory (hipDeviceMallocUncached) so peer stores and
// the local spin bypass L2.
// The sequence number of a call is NOT a kernel argument: a hipGraph would bake one value in and every replay after the
// first would find its flags already satisfied (stale sums).  It is a counter in this rank's device memory that the
// kernel reads, advances and writes back; launches of one communicator are stream-ordered, and every rank issues the
// same sequence of calls, so the counters of all ranks stay in step whether the launches are eager or replayed.
constexpr int kXgmiMaxWorld = 16;

struct XgmiPeers {
  float* slots[kXgmiMaxWorld];        // base of every rank's mailbox (mapped into this process)
  unsigned long long* flags[kXgmiMaxWorld];
};

__global__ __launch_bounds__(256) void xgmi_allreduce_k(XgmiPeers peers, float* __restrict__ buf, int n, int cap,
                                                        int rank, int world, unsigned long long* __restrict__ seq_dev) {
  const int tid = threadIdx.x;
  __shared__ int timed_out;
  __shared__ unsigned long long seq_sh;
  if (tid == 0) {
    timed_out = 0;
    seq_sh = __hip_atomic_load(seq_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
  }
  __syncthreads();
  const unsigned long long seq = seq_sh;
  const int par = (int)(seq & 1ull);
  const size_t my_slot = ((size_t)par * world + rank) * cap;
  for (int p = 0; p < world; ++p) {
    float* dst = peers.slots[p] + my_slot;
    for (int i = tid; i < n; i += 256) __builtin_nontemporal_store(buf[i], dst + i);
  }
  __threadfence_system();
  __syncthreads();
  if (tid < world)
    __hip_atomic_store(peers.flags[tid] + (size_t)par * world + rank, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  if (tid < world) {
    const unsigned long long* f = peers.flags[rank] + (size_t)par * world + tid;
    // bounded spin (a few seconds): a rank that never arrives must not wedge the GPU; the result is then poisoned
    long long spins = 0;
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1ll << 25)) { timed_out = 1; break; }
    }
  }
  __syncthreads();
  // the next launch on this stream (eager or a replayed graph node) reads seq + 1
  if (tid == 0) __hip_atomic_store(seq_dev, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (timed_out) {
    for (int i = tid; i < n; i += 256) buf[i] = __int_as_float(0x7fc00000);
    return;
  }
  const float* mine = peers.slots[rank] + (size_t)par * world * cap;
  for (int i = tid; i < n; i += 256) {
    float s = 0.f;
    for (int r = 0; r < world; ++r) s += __builtin_nontemporal_load(mine + (size_t)r * cap + i);
    buf[i] = s;
  }
}

struct tsg_comm {
  ncclComm_t comm;                    // nullptr: created without RCCL (mailbox path only)
  int rank, world, device;
  // mailbox state (all zero until tsg_comm_xgmi_export / _attach)
  void* box;                          // this rank's mailbox allocation
  int cap;                            // floats per slot
  bool attached;
  void* peer_base[kXgmiMaxWorld];     // opened IPC mappings (own entry = box)
  XgmiPeers peers;
  unsigned long long* seq_dev;        // calls completed so far, in device memory (advanced by the kernel: graph-safe)
};

namespace {
size_t box_slot_bytes(int world, int cap) { return (size_t)2 * world * cap * sizeof(float); }
size_t box_bytes(int world, int cap) { return box_slot_bytes(world, cap) + (size_t)2 * world * sizeof(unsigned long long); }
}

extern "C" {

int tsg_comm_init_library(const char* librccl_path) { return load_rccl(librccl_path); }

int tsg_comm_unique_id_bytes(void) { return (int)sizeof(ncclUniqueId); }

int tsg_comm_get_unique_id(void* id_out) {
  if (!id_out) return TSG_E_NULL;
  int e = load_rccl(nullptr);
  if (e) return e;
  ncclUniqueId id;
  memset(&id, 0, sizeof id);
  e = rc(g_rccl.GetUniqueId(&id));
  if (e) return e;
  memcpy(id_out, &id, sizeof id);
  return 0;
}

int tsg_comm_create(const void* unique_id, int rank, int world, int device, tsg_comm** out) {
  if (!out) return TSG_E_NULL;
  if (world < 1 || rank < 0 || rank >= world || device < 0) return TSG_E_SHAPE;
  hipError_t he = hipSetDevice(device);
  if (he != hipSuccess) return (int)he;
  ncclComm_t c = nullptr;
  if (unique_id) {
    int e = load_rccl(nullptr);
    if (e) return e;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    e = rc(g_rccl.CommInitRank(&c, world, id, rank));
    if (e) return e;
  }
  tsg_comm* h = new tsg_comm();
  memset(h, 0, sizeof *h);
  h->comm = c; h->rank = rank; h->world = world; h->device = device;
  *out = h;
  return 0;
}

int tsg_comm_destroy(tsg_comm* c) {
  if (!c) return TSG_E_NULL;
  int e = 0;
  if (c->attached)
    for (int p = 0; p < c->world; ++p)
      if (p != c->rank && c->peer_base[p]) (void)hipIpcCloseMemHandle(c->peer_base[p]);
  if (c->box) (void)hipFree(c->box);
  if (c->seq_dev) (void)hipFree(c->seq_dev);
  if (c->comm) e = rc(g_rccl.CommDestroy(c->comm));
  delete c;
  return e;
}

size_t tsg_comm_xgmi_handle_bytes(void) { return sizeof(hipIpcMemHandle_t); }

int tsg_comm_xgmi_export(tsg_comm* c, int64_t max_floats, void* handle_out) {
  if (!c || !handle_out) return TSG_E_NULL;
  if (max_floats <= 0 || max_floats > (1 << 20) || c->world > kXgmiMaxWorld || c->box) return TSG_E_SHAPE;
  TSG_HIP(hipSetDevice(c->device));
  const int cap = (int)((max_floats + 3) / 4 * 4);
  void* p = nullptr;
  // uncached (MTYPE_UC) so that peer stores and the local spin never sit in a non-coherent L2 line;
  // TSG_XGMI_ALLOC=finegrained|default selects the other allocation kinds (bring-up knob)
  const char* kind = getenv("TSG_XGMI_ALLOC");
  if (kind && !strcmp(kind, "default")) TSG_HIP(hipMalloc(&p, box_bytes(c->world, cap)));
  else TSG_HIP(hipExtMallocWithFlags(&p, box_bytes(c->world, cap),
                                     kind && !strcmp(kind, "finegrained") ? hipDeviceMallocFinegrained : hipDeviceMallocUncached));
  TSG_HIP(hipMemset(p, 0, box_bytes(c->world, cap)));
  TSG_HIP(hipDeviceSynchronize());
  hipIpcMemHandle_t h;
  hipError_t e = hipIpcGetMemHandle(&h, p);
  if (e != hipSuccess) { (void)hipFree(p); return (int)e; }
  void* sq = nullptr;
  e = hipMalloc(&sq, sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMemset(sq, 0, sizeof(unsigned long long));
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) { (void)hipFree(p); if (sq) (void)hipFree(sq); return (int)e; }
  memcpy(handle_out, &h, sizeof h);
  c->box = p;
  c->cap = cap;
  c->seq_dev = (unsigned long long*)sq;
  return 0;
}

int tsg_comm_xgmi_attach(tsg_comm* c, const void* all_handles) {
  if (!c || !all_handles) return TSG_E_NULL;
  if (!c->box || c->attached) return TSG_E_SHAPE;
  TSG_HIP(hipSetDevice(c->device));
  const char* hs = (const char*)all_handles;
  for (int p = 0; p < c->world; ++p) {
    void* base = c->box;
    if (p != c->rank) {
      hipIpcMemHandle_t h;
      memcpy(&h, hs + (size_t)p * sizeof h, sizeof h);
      base = nullptr;
      TSG_HIP(hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess));
    }
    c->peer_base[p] = base;
    c->peers.slots[p] = (float*)base;
    c->peers.flags[p] = (unsigned long long*)((char*)base + box_slot_bytes(c->world, c->cap));
  }
  c->attached = true;
  return 0;
}

int tsg_xgmi_small_allreduce(tsg_comm* c, float* buf, int64_t count, void* stream) {
  if (!c || !buf) return TSG_E_NULL;
  if (count < 0) return TSG_E_SHAPE;
  if (count == 0) return 0;
  if (c->attached && count <= c->cap) {
    hipLaunchKernelGGL(xgmi_allreduce_k, dim3(1), dim3(256), 0, (hipStream_t)stream, c->peers, buf, (int)count, c->cap,
                       c->rank, c->world, c->seq_dev);
    TSG_CHECK_LAUNCH();
    return 0;
  }
  if (!c->comm) return TSG_E_SHAPE;       // no mailbox large enough and no RCCL communicator to fall back on
  return rc(g_rccl.AllReduce(buf, buf, (size_t)count, kNcclFloat32, kNcclSum, c->comm, (hipStream_t)stream));
}

int tsg_comm_rank(const tsg_comm* c) { return c ? c->rank : TSG_E_NULL; }
int tsg_comm_world(const tsg_comm* c) { return c ? c->world : TSG_E_NULL; }

int tsg_comm_allreduce(tsg_comm* c, void* buf, int64_t count, int dtype, void* stream) {
  if (!c || !buf || !c->comm) return TSG_E_NULL;
  if (count < 0) return TSG_E_SHAPE;
  const int t = nccl_type(dtype);
  if (t < 0) return TSG_E_DTYPE;
  if (count == 0) return 0;
  return rc(g_rccl.AllReduce(buf, buf, (size_t)count, t, kNcclSum, c->comm, (hipStream_t)stream));
}

int tsg_comm_allgather(tsg_comm* c, const void* send, void* recv, int64_t count_per_rank, int dtype, void* stream) {
  if (!c || !send || !recv || !c->comm) return TSG_E_NULL;
  if (count_per_rank < 0) return TSG_E_SHAPE;
  const int t = nccl_type(dtype);
  if (t < 0) return TSG_E_DTYPE;
  if (count_per_rank == 0) return 0;
  return rc(g_rccl.AllGather(send, recv, (size_t)count_per_rank, t, c->comm, (hipStream_t)stream));
}

int tsg_comm_reduce_scatter(tsg_comm* c, const void* send, void* recv, int64_t count_per_rank, int dtype, void* stream) {
  if (!c || !send || !recv || !c->comm) return TSG_E_NULL;
  if (count_per_rank < 0) return TSG_E_SHAPE;
  const int t = nccl_type(dtype);
  if (t < 0) return TSG_E_DTYPE;
  if (count_per_rank == 0) return 0;
  return rc(g_rccl.ReduceScatter(send, recv, (size_t)count_per_rank, t, kNcclSum, c->comm, (hipStream_t)stream));
}

int tsg_comm_broadcast(tsg_comm* c, void* buf, int64_t count, int dtype, int root, void* stream) {
  if (!c || !buf || !c->comm) return TSG_E_NULL;
  if (count < 0 || root < 0 || root >= c->world) return TSG_E_SHAPE;
  const int t = nccl_type(dtype);
  if (t < 0) return TSG_E_DTYPE;
  if (count == 0) return 0;
  return rc(g_rccl.Broadcast(buf, buf, (size_t)count, t, root, c->comm, (hipStream_t)stream));
}

const char* tsg_comm_error_string(int code) {
  if (code == TSG_E_COMM_LIB) return "librccl.so could not be loaded or lacks a required symbol";
  if (code <= TSG_E_COMM_BASE && g_rccl.GetErrorString) return g_rccl.GetErrorString(TSG_E_COMM_BASE - code);
  return "not a communicator error";
}

}  // extern "C"
