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

// counts[r] = bytes of rank r (all ranks); on rank 0 recv holds the payloads in rank order
extern "C" int pgpu_gather(pgpu_ctx* ctx, pgpu_comm* c, const void* send, uint64_t send_bytes, void* recv,
                           uint64_t recv_cap, uint64_t* counts) {
  if (!ctx || !c || !counts || (send_bytes && !send)) return PGPU_EINVAL;
  if (pgpu_ctx_bind(ctx) != PGPU_OK) return PGPU_EDEVICE;
  hipStream_t st = pgpu_ctx_stream(ctx);
  const int W = c->world;
  // 1. everybody learns everybody's size
  const unsigned long long mine = send_bytes;
  HIP_TRY2(hipMemcpyAsync(c->d_counts + W, &mine, sizeof mine, hipMemcpyHostToDevice, st));
  RCCL_TRY(g_rccl.AllGather(c->d_counts + W, c->d_counts, 1, NCCL_UINT64, c->nccl, st));
  HIP_TRY2(hipMemcpyAsync(counts, c->d_counts, (size_t)W * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  HIP_TRY2(hipStreamSynchronize(st));
  uint64_t total = 0;
  for (int r = 0; r < W; ++r) total += counts[r];
  if (c->rank == 0 && (total > recv_cap || (total && !recv))) return pgpu_ctx_fail(ctx, PGPU_ENOSPC, "gather buffer too small");
  // 2. payloads: rank 0 posts one receive per sender, the senders one send each (a gatherv; the seven
  //    senders of an 8-GPU node use seven different xGMI links into rank 0)
  if (send_bytes > c->send_cap) {
    hipFree(c->d_send); c->d_send = nullptr; c->send_cap = 0;
    HIP_TRY2(hipMalloc((void**)&c->d_send, send_bytes + send_bytes / 4 + 4096));
    c->send_cap = send_bytes + send_bytes / 4 + 4096;
  }
  if (c->rank == 0 && total > c->recv_cap) {
    hipFree(c->d_recv); c->d_recv = nullptr; c->recv_cap = 0;
    HIP_TRY2(hipMalloc((void**)&c->d_recv, total + total / 4 + 4096));
    c->recv_cap = total + total / 4 + 4096;
  }
  if (c->rank == 0) {
    // own part straight into place, the others over the wire
    if (send_bytes) HIP_TRY2(hipMemcpyAsync(c->d_recv, send, send_bytes, hipMemcpyHostToDevice, st));
    RCCL_TRY(g_rccl.GroupStart());
    uint64_t at = counts[0];
    for (int r = 1; r < W; ++r) { if (counts[r]) RCCL_TRY(g_rccl.Recv(c->d_recv + at, counts[r], NCCL_UINT8, r, c->nccl, st)); at += counts[r]; }
    RCCL_TRY(g_rccl.GroupEnd());
    if (total) HIP_TRY2(hipMemcpyAsync(recv, c->d_recv, total, hipMemcpyDeviceToHost, st));
  } else if (send_bytes) {
    HIP_TRY2(hipMemcpyAsync(c->d_send, send, send_bytes, hipMemcpyHostToDevice, st));
    RCCL_TRY(g_rccl.Send(c->d_send, send_bytes, NCCL_UINT8, 0, c->nccl, st));
  }
  HIP_TRY2(hipStreamSynchronize(st));
  return PGPU_OK;
}
