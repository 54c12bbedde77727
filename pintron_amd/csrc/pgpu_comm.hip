// The one exchange of the EST-sharded path: every rank's output bytes go to
// rank 0 over RCCL (xGMI inside a node).  est-fact itself is a single process in the reference
// (src/main-est-fact.c); what this replaces is the order-preserving concatenation a sharded run needs
// before rank 0 can write the files a single process would have written.
//
// RCCL is loaded on first use (dlopen), not linked: processes that never shard -- and Python
// processes that already carry torch's own RCCL -- do not get a second collective library mapped.
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <new>

#include "pgpu_index.h"

namespace {

// the handful of RCCL entry points used, with the types of rccl.h (ncclResult_t = int,
// ncclComm_t = opaque pointer, ncclUniqueId = 128 bytes, ncclUint8 = 1, ncclUint64 = 5)
struct Rccl {
  void* so = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, pgpu_comm_id, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;

bool load_rccl() {
  if (g_rccl.so) return true;
  const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
  void* so = nullptr;
  for (const char* n : names) if ((so = dlopen(n, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
  if (!so) return false;
  Rccl r;
  r.so = so;
#define SYM(field, name) *(void**)(&r.field) = dlsym(so, name); if (!r.field) { dlclose(so); return false; }
  SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
  SYM(AllGather, "ncclAllGather") SYM(Send, "ncclSend") SYM(Recv, "ncclRecv") SYM(GroupStart, "ncclGroupStart")
  SYM(GroupEnd, "ncclGroupEnd") SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
  g_rccl = r;
  return true;
}

constexpr int NCCL_UINT8 = 1, NCCL_UINT64 = 5;

}  // namespace

struct pgpu_comm {
  void* nccl = nullptr;
  int rank = 0, world = 1;
  unsigned long long* d_counts = nullptr;     // world + 1 entries: [0..world) gathered, [world] own
  uint8_t* d_send = nullptr; size_t send_cap = 0;
  uint8_t* d_recv = nullptr; size_t recv_cap = 0;
};

#define RCCL_TRY(call) do { const int r_ = (call); if (r_ != 0) { char m_[256]; snprintf(m_, sizeof m_, "%s failed: %s", #call, g_rccl.GetErrorString(r_)); return pgpu_ctx_fail(ctx, PGPU_EDEVICE, m_); } } while (0)
#define HIP_TRY2(call) do { const hipError_t e_ = (call); if (e_ != hipSuccess) return pgpu_ctx_fail(ctx, e_ == hipErrorOutOfMemory ? PGPU_ENOMEM : PGPU_EDEVICE, hipGetErrorString(e_)); } while (0)

extern "C" int pgpu_comm_unique_id(pgpu_ctx* ctx, pgpu_comm_id* id) {
  if (!ctx || !id) return PGPU_EINVAL;
  if (!load_rccl()) return pgpu_ctx_fail(ctx, PGPU_ENOSYS, "librccl could not be loaded");
  RCCL_TRY(g_rccl.GetUniqueId(id));
  return PGPU_OK;
}

extern "C" int pgpu_comm_init(pgpu_ctx* ctx, int rank, int world, const pgpu_comm_id* id, pgpu_comm** out) {
  if (!ctx || !out || !id || world < 1 || rank < 0 || rank >= world) return PGPU_EINVAL;
  *out = nullptr;
  if (!load_rccl()) return pgpu_ctx_fail(ctx, PGPU_ENOSYS, "librccl could not be loaded");
  if (pgpu_ctx_bind(ctx) != PGPU_OK) return PGPU_EDEVICE;
  pgpu_comm* c = new (std::nothrow) pgpu_comm();
  if (!c) return pgpu_ctx_fail(ctx, PGPU_ENOMEM, "out of host memory");
  c->rank = rank; c->world = world;
  const int rc = g_rccl.CommInitRank(&c->nccl, world, *id, rank);
  if (rc != 0) { delete c; return pgpu_ctx_fail(ctx, PGPU_EDEVICE, g_rccl.GetErrorString(rc)); }
  if (hipMalloc((void**)&c->d_counts, ((size_t)world + 1) * sizeof(unsigned long long)) != hipSuccess) {
    g_rccl.CommDestroy(c->nccl); delete c;
    return pgpu_ctx_fail(ctx, PGPU_ENOMEM, "out of device memory");
  }
  *out = c;
  return PGPU_OK;
}

extern "C" int pgpu_comm_destroy(pgpu_ctx* ctx, pgpu_comm* c) {
  if (!ctx || !c) return PGPU_EINVAL;
  if (pgpu_ctx_bind(ctx) != PGPU_OK) return PGPU_EDEVICE;
  hipStreamSynchronize(pgpu_ctx_stream(ctx));
  if (c->nccl) g_rccl.CommDestroy(c->nccl);
  hipFree(c->d_counts); hipFree(c->d_send); hipFree(c->d_recv);
  delete c;
  return PGPU_OK;
}

// every rank contributes `bytes` bytes, every rank receives world x bytes in rank order (the status /
// sizes vector of a sharded run: small, so it is staged through one device buffer per communicator)
extern "C" int pgpu_allgather(pgpu_ctx* ctx, pgpu_comm* c, const void* send, uint64_t bytes, void* recv) {
  if (!ctx || !c || (bytes && (!send || !recv))) return PGPU_EINVAL;
  if (bytes == 0) return PGPU_OK;
  if (pgpu_ctx_bind(ctx) != PGPU_OK) return PGPU_EDEVICE;
  hipStream_t st = pgpu_ctx_stream(ctx);
  const size_t W = (size_t)c->world, need = (W + 1) * bytes;
  if (need > c->send_cap) {
    hipFree(c->d_send); c->d_send = nullptr; c->send_cap = 0;
    HIP_TRY2(hipMalloc((void**)&c->d_send, need + 4096));
    c->send_cap = need + 4096;
  }
  HIP_TRY2(hipMemcpyAsync(c->d_send + W * bytes, send, bytes, hipMemcpyHostToDevice, st));
  RCCL_TRY(g_rccl.AllGather(c->d_send + W * bytes, c->d_send, bytes, NCCL_UINT8, c->nccl, st));
  HIP_TRY2(hipMemcpyAsync(recv, c->d_send, W * bytes, hipMemcpyDeviceToHost, st));
  HIP_TRY2(hipStreamSynchronize(st));
  return PGPU_OK;
}

// counts[r] = bytes of rank r (all ranks); on rank 0 recv holds the payloads in rank order.
// Once the sizes have been exchanged every rank is committed to its send / its receives: an error that
// only one side can see (rank 0's buffer too small, an allocation that fails) must not keep that side
// from posting its half, or the peers block for ever.  So rank 0 always receives into its own device
// buffer and reports PGPU_ENOSPC afterwards; a group that has been opened is always closed.
extern "C" int pgpu_gather(pgpu_ctx* ctx, pgpu_comm* c, const void* send, uint64_t send_bytes, void* recv,
                           uint64_t recv_cap, uint64_t* counts) {
  if (!ctx || !c || !counts || (send_bytes && !send)) return PGPU_EINVAL;
  if (pgpu_ctx_bind(ctx) != PGPU_OK) return PGPU_EDEVICE;
  hipStream_t st = pgpu_ctx_stream(ctx);
  const int W = c->world;
  // buffers first: a rank that cannot allocate says so in the size round (all ones) and nobody sends
  unsigned long long mine = send_bytes;
  if (c->rank != 0 && send_bytes > c->send_cap) {
    hipFree(c->d_send); c->d_send = nullptr; c->send_cap = 0;
    if (hipMalloc((void**)&c->d_send, send_bytes + send_bytes / 4 + 4096) == hipSuccess) c->send_cap = send_bytes + send_bytes / 4 + 4096;
    else mine = ~0ull;
  }
  // 1. everybody learns everybody's size
  HIP_TRY2(hipMemcpyAsync(c->d_counts + W, &mine, sizeof mine, hipMemcpyHostToDevice, st));
  RCCL_TRY(g_rccl.AllGather(c->d_counts + W, c->d_counts, 1, NCCL_UINT64, c->nccl, st));
  HIP_TRY2(hipMemcpyAsync(counts, c->d_counts, (size_t)W * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  HIP_TRY2(hipStreamSynchronize(st));
  uint64_t total = 0;
  for (int r = 0; r < W; ++r) {
    if (counts[r] == ~0ull) return pgpu_ctx_fail(ctx, PGPU_ENOMEM, "a rank could not allocate its send buffer");   // seen by every rank alike
    total += counts[r];
  }
  // 2. payloads: rank 0 posts one receive per sender, the senders one send each (a gatherv; the seven
  //    senders of an 8-GPU node use seven different xGMI links into rank 0)
  int later = PGPU_OK;                               // what this rank reports once its half has been posted
  if (c->rank == 0) {
    if (total > recv_cap || (total && !recv)) later = PGPU_ENOSPC;
    bool sink = false;
    if (total > c->recv_cap) {
      hipFree(c->d_recv); c->d_recv = nullptr; c->recv_cap = 0;
      if (hipMalloc((void**)&c->d_recv, total + total / 4 + 4096) == hipSuccess) c->recv_cap = total + total / 4 + 4096;
      else {
        // no room for the whole: receive every payload into one scratch block of the largest size (the
        // data is lost, the peers are released)
        uint64_t mx = 1;
        for (int r = 1; r < W; ++r) mx = counts[r] > mx ? counts[r] : mx;
        later = PGPU_ENOMEM; sink = true;
        if (hipMalloc((void**)&c->d_recv, mx) == hipSuccess) c->recv_cap = mx;
        else return pgpu_ctx_fail(ctx, PGPU_ENOMEM, "out of device memory in gather (the peers are left waiting)");
      }
    }
    // own part straight into place, the others over the wire
    if (send_bytes && !sink && hipMemcpyAsync(c->d_recv, send, send_bytes, hipMemcpyHostToDevice, st) != hipSuccess)
      later = PGPU_EDEVICE;                              // reported after the receives have been posted
    int grc = g_rccl.GroupStart();
    if (grc == 0) {
      uint64_t at = counts[0];
      int first_bad = 0;
      for (int r = 1; r < W; ++r) {
        if (counts[r]) { const int q = g_rccl.Recv(c->d_recv + (sink ? 0 : at), counts[r], NCCL_UINT8, r, c->nccl, st); if (q != 0 && !first_bad) first_bad = q; }
        at += counts[r];
      }
      const int end = g_rccl.GroupEnd();             // always: an open group would swallow every later call
      grc = first_bad ? first_bad : end;
    }
    if (grc != 0) { char m[256]; snprintf(m, sizeof m, "gather (receive side) failed: %s", g_rccl.GetErrorString(grc)); return pgpu_ctx_fail(ctx, PGPU_EDEVICE, m); }
    if (total && later == PGPU_OK) HIP_TRY2(hipMemcpyAsync(recv, c->d_recv, total, hipMemcpyDeviceToHost, st));
  } else if (send_bytes) {
    HIP_TRY2(hipMemcpyAsync(c->d_send, send, send_bytes, hipMemcpyHostToDevice, st));
    RCCL_TRY(g_rccl.Send(c->d_send, send_bytes, NCCL_UINT8, 0, c->nccl, st));
  }
  HIP_TRY2(hipStreamSynchronize(st));
  if (later == PGPU_ENOSPC) return pgpu_ctx_fail(ctx, PGPU_ENOSPC, "gather buffer too small");
  if (later == PGPU_EDEVICE) return pgpu_ctx_fail(ctx, PGPU_EDEVICE, "copy of rank 0's own payload failed");
  if (later != PGPU_OK) return pgpu_ctx_fail(ctx, later, "out of device memory in gather");
  return PGPU_OK;
}
