// Host side of libpintron_gpu.so: contexts, batched DP plans, C-ABI entry points.
// (C++ inside, extern "C" at the boundary; see include/pintron_gpu.h for the contract.)
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <time.h>
#include <new>
#include <string>
#include <mutex>
#include <vector>

#include "pgpu_internal.h"
#include "pgpu_index.h"

// Device scratch that outlives a plan: the batched host program creates and destroys a plan per
// round, and hipMalloc/hipFree (the latter synchronises the device) would dominate.  A context
// keeps one grow-only set of buffers per plan type; a plan borrows the set when it is free.
struct BufPool {
  static constexpr int SLOTS = 24;
  void* ptr[SLOTS] = {nullptr};
  size_t cap[SLOTS] = {0};
  bool busy = false;
  int sharers = 0;                // > 0: held by plans that use it one after the other (pgpu_pairing_plan_create_resident)
  const void* owner = nullptr;    // whose results the buffers hold right now
};

struct pgpu_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  char err[512] = {0};
  BufPool pools[2];          // 0: DP plans, 1: pairing plans
  bool timing = false;       // HIP events around every kernel group (bench / profiling)
  // the kernel groups of one DP plan are independent of each other: they are spread over a few
  // auxiliary streams so that a batch costs max(group) instead of sum(group) in latency
  static constexpr int NAUX = 8;
  hipStream_t aux[NAUX] = {nullptr};
  hipEvent_t ev_upload = nullptr;
  hipEvent_t ev_aux[NAUX] = {nullptr};
  bool fanout = true;        // spread groups over the auxiliary streams (PGPU_FANOUT=0 disables)
  int n_aux = NAUX;          // how many of them are used (PGPU_STREAMS=1..8)
  int merged = 2;            // 2: the latency-bound part of a batch (one-job-per-workgroup sweeps + every wave-per-job
                             //    family) in ONE launch beside the LCF launch; 1: the wave-per-job families in one
                             //    launch, the sweeps in theirs; 0: a launch per family (PGPU_MERGED)
  bool packed = true;        // four streams, kernel families packed by expected duration (PGPU_PACK=0: round-robin over n_aux)
  // waiting: the calling thread must not burn a host core that other EST fibres could use (the
  // default HIP wait spins).  It naps and polls the event: measured on C3, naps of 50-200 us beat
  // a blocking-sync event by 5-7 % whole-program (the interrupt path costs more host time than
  // the naps cost latency).  PGPU_WAIT=<us> sets the nap, 0 = blocking-sync event, -1 = spin.
  hipEvent_t ev_done = nullptr;
  hipEvent_t ev_wait = nullptr;      // pgpu_ctx_wait (pairing / MEG stages)
  long wait_poll_us = 20;
  bool align_coop = true;    // ALIGN with 65 .. 4096 rows on four waves (PGPU_ALIGN_COOP=0: one wave, as before)
  bool align_band = true;    // ALIGN above 64 rows: inside a band on one wave first (PGPU_ALIGN_BAND=0: always the whole matrix)
  bool lcf_sa_n = true;      // ... also for an EST prefix with one N (PGPU_LCF_SA_N=0: those take the DP kernel, as before)
  bool lcf_sa = true;        // longest common factors of genomic prefixes from the suffix array (PGPU_LCF_SA=0: always the DP kernel)
  int poison = -1;           // PGPU_POISON=<0..255>: fill strings + workspace of every DP plan with that byte first
  // pinned staging for the device->host result copies (pageable copies block and spin inside HIP)
  void* pin[2] = {nullptr, nullptr};
  size_t pin_cap[2] = {0, 0};
  // timing events are recycled (a plan with the DP pool borrows them; create/destroy per group
  // and batch is measurable at a few hundred batches per second)
  std::vector<hipEvent_t> ev_pool;
};

static int wait_stream(pgpu_ctx* ctx, hipStream_t st) {
  if (hipEventRecord(ctx->ev_done, st) != hipSuccess) return -1;
  if (ctx->wait_poll_us != 0) {
    for (;;) {
      const hipError_t q = hipEventQuery(ctx->ev_done);
      if (q == hipSuccess) return 0;
      if (q != hipErrorNotReady) return -1;
      if (ctx->wait_poll_us > 0) { struct timespec ts = {0, ctx->wait_poll_us * 1000L}; nanosleep(&ts, nullptr); }
    }
  }
  return hipEventSynchronize(ctx->ev_done) == hipSuccess ? 0 : -1;
}

static void* pinned(pgpu_ctx* ctx, int slot, size_t bytes) {
  if (ctx->pin_cap[slot] < bytes) {
    if (ctx->pin[slot]) (void)hipHostFree(ctx->pin[slot]);
    ctx->pin[slot] = nullptr; ctx->pin_cap[slot] = 0;
    // never less than 2 MB: the first batches of a run grow from a few hundred jobs to thousands, and every
    // regrowth is a free (which waits for the device) and an allocation
    const size_t want = std::max<size_t>(bytes + bytes / 2 + 4096, (size_t)2 << 20);
    if (hipHostMalloc(&ctx->pin[slot], want, hipHostMallocDefault) != hipSuccess) return nullptr;
    pgpu_trace_alloc(slot ? "ctx pinned down" : "ctx pinned up", ctx->pin[slot], want);
    ctx->pin_cap[slot] = want;
  }
  return ctx->pin[slot];
}

bool pgpu_ctx_pool_acquire(pgpu_ctx* ctx, int pool) {
  if (ctx->pools[pool].busy) return false;
  ctx->pools[pool].busy = true;
  return true;
}
void pgpu_ctx_pool_release(pgpu_ctx* ctx, int pool) { ctx->pools[pool].busy = false; ctx->pools[pool].owner = nullptr; }
// several plans, used strictly one after the other, on the same buffers
bool pgpu_ctx_pool_share(pgpu_ctx* ctx, int pool) {
  BufPool& p = ctx->pools[pool];
  if (p.busy && p.sharers == 0) return false;       // a plan holds it for itself
  p.busy = true; ++p.sharers;
  return true;
}
void pgpu_ctx_pool_unshare(pgpu_ctx* ctx, int pool, const void* who) {
  BufPool& p = ctx->pools[pool];
  if (p.owner == who) p.owner = nullptr;
  if (p.sharers > 0 && --p.sharers == 0) p.busy = false;
}
void pgpu_ctx_pool_set_owner(pgpu_ctx* ctx, int pool, const void* who) { ctx->pools[pool].owner = who; }
const void* pgpu_ctx_pool_owner(const pgpu_ctx* ctx, int pool) { return ctx->pools[pool].owner; }
// returns a buffer of at least `bytes` from the (acquired) pool, growing it when needed
void* pgpu_ctx_pool_get(pgpu_ctx* ctx, int pool, int slot, size_t bytes) {
  BufPool& p = ctx->pools[pool];
  if (bytes == 0) bytes = 16;
  if (p.cap[slot] < bytes) {
    if (p.ptr[slot]) (void)hipFree(p.ptr[slot]);
    p.ptr[slot] = nullptr; p.cap[slot] = 0;
    const size_t want = std::max<size_t>(bytes + bytes / 2 + 4096, pool == 0 ? (size_t)64 << 20 : 0);
    if (hipMalloc(&p.ptr[slot], want) != hipSuccess) {
      if (hipMalloc(&p.ptr[slot], bytes) != hipSuccess) return nullptr;
      p.cap[slot] = bytes;
    } else p.cap[slot] = want;
    pgpu_trace_alloc(pool ? "device pool 1" : "device pool 0", p.ptr[slot], p.cap[slot]);
  }
  return p.ptr[slot];
}
bool pgpu_ctx_timing(const pgpu_ctx* ctx) { return ctx->timing; }

extern "C" int pgpu_device_numa_node(pgpu_ctx* ctx) {
  if (!ctx) return -1;
  char bdf[64] = {0};
  if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf - 1, ctx->device) != hipSuccess) return -1;
  for (char* c = bdf; *c; ++c) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');   // sysfs spells it in lower case
  char path[160];
  snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
  FILE* f = fopen(path, "r");
  if (!f) return -1;
  int node = -1;
  if (fscanf(f, "%d", &node) != 1) node = -1;
  fclose(f);
  return node;
}

extern "C" int pgpu_set_timing(pgpu_ctx* ctx, int enabled) {
  if (!ctx) return PGPU_EINVAL;
  ctx->timing = enabled != 0;
  return PGPU_OK;
}

static int set_err(pgpu_ctx* ctx, int code, const char* fmt, ...) {
  if (ctx) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
    va_end(ap);
  }
  return code;
}

#define HIP_TRY(ctx, call)                                                                  \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return set_err(ctx, e_ == hipErrorOutOfMemory ? PGPU_ENOMEM : PGPU_EDEVICE,           \
                     "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

hipStream_t pgpu_ctx_stream(pgpu_ctx* ctx) { return ctx->stream; }
hipError_t pgpu_ctx_wait(pgpu_ctx* ctx) {
  if (ctx->wait_poll_us <= 0) return hipStreamSynchronize(ctx->stream);
  hipError_t e = hipEventRecord(ctx->ev_wait, ctx->stream);
  if (e != hipSuccess) return e;
  // the kernels behind these waits run for milliseconds: naps five times the DP plans' (100 us by default)
  const struct timespec ts = {0, ctx->wait_poll_us * 5000L};
  for (;;) {
    e = hipEventQuery(ctx->ev_wait);
    if (e != hipErrorNotReady) return e;
    nanosleep(&ts, nullptr);
  }
}
int pgpu_ctx_bind(pgpu_ctx* ctx) {
  const hipError_t e = hipSetDevice(ctx->device);
  return e == hipSuccess ? PGPU_OK : set_err(ctx, PGPU_EDEVICE, "hipSetDevice(%d) failed: %s", ctx->device, hipGetErrorString(e));
}
int pgpu_ctx_fail(pgpu_ctx* ctx, int code, const char* msg) { return set_err(ctx, code, "%s", msg); }

extern "C" int pgpu_abi_version(void) { return 1; }

// ---- profiler ranges --------------------------------------------------------------------------------
#include <dlfcn.h>
namespace {
struct Roctx { int state = 0; int (*push)(const char*) = nullptr; int (*pop)() = nullptr; };   // state: 0 untried, 1 on, -1 off
Roctx g_roctx;
bool roctx_ready() {
  if (g_roctx.state) return g_roctx.state > 0;
  bool want = false;
  { const char* m = getenv("PGPU_MARKERS"); if (m) want = m[0] != '0'; else {
      const char* t = getenv("ROCP_TOOL_LIBRARIES"); const char* pl = getenv("LD_PRELOAD");
      want = (t && t[0]) || (pl && strstr(pl, "rocprofiler")); } }
  if (want) {
    void* so = nullptr;
    for (const char* n : { "librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so" })
      if ((so = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
    if (so) {
      *(void**)(&g_roctx.push) = dlsym(so, "roctxRangePushA");
      *(void**)(&g_roctx.pop) = dlsym(so, "roctxRangePop");
    }
  }
  g_roctx.state = (g_roctx.push && g_roctx.pop) ? 1 : -1;
  return g_roctx.state > 0;
}
}  // namespace
extern "C" void pgpu_range_push(const char* name) { if (roctx_ready()) g_roctx.push(name ? name : ""); }
extern "C" void pgpu_range_pop(void) { if (roctx_ready()) g_roctx.pop(); }

extern "C" const char* pgpu_build_info(void) {
  static char info[256];
  if (!info[0]) {
    const char* stamp = __DATE__ " " __TIME__ " " __VERSION__;
    unsigned long long h = 1469598103934665603ull;
    for (const char* c = stamp; *c; ++c) { h ^= (unsigned char)*c; h *= 1099511628211ull; }
    snprintf(info, sizeof info, "libpintron_gpu abi %d, gfx950, built %s %s, hip %d.%d.%d, stamp %016llx", pgpu_abi_version(), __DATE__,
             __TIME__, HIP_VERSION_MAJOR, HIP_VERSION_MINOR, HIP_VERSION_PATCH, h);
  }
  return info;
}

static bool aux_stream(pgpu_ctx* ctx, int i) {
  if (!ctx->aux[i] && hipStreamCreateWithFlags(&ctx->aux[i], hipStreamNonBlocking) != hipSuccess) { ctx->aux[i] = nullptr; return false; }
  if (!ctx->ev_aux[i] && hipEventCreateWithFlags(&ctx->ev_aux[i], hipEventDisableTiming) != hipSuccess) { ctx->ev_aux[i] = nullptr; return false; }
  return true;
}

// One timing event per device, recorded once when the first context of the process comes up: the start of every
// kernel group is reported relative to it (pgpu_group_info.t0_ms), so that a caller with several contexts
// (the host program's service threads) can lay the launches of all of them on ONE time line and tell the time
// the device was busy from the sum of the launches' durations.
static hipEvent_t g_base_event[64];
static std::mutex g_base_mu;
static hipEvent_t base_event(int device, hipStream_t st) {
  if (device < 0 || device >= 64) return nullptr;
  std::lock_guard<std::mutex> lk(g_base_mu);
  if (!g_base_event[device]) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) == hipSuccess && hipEventRecord(e, st) == hipSuccess && hipEventSynchronize(e) == hipSuccess) g_base_event[device] = e;
    else if (e) hipEventDestroy(e);
  }
  return g_base_event[device];
}

extern "C" int pgpu_init(int device, pgpu_ctx** out) {
  if (!out) return PGPU_EINVAL;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return PGPU_EDEVICE;
  if (device < 0 || device >= n) return PGPU_EINVAL;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return PGPU_EDEVICE;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return PGPU_EDEVICE;   // gfx950 code objects only
  if (hipSetDevice(device) != hipSuccess) return PGPU_EDEVICE;
  pgpu_ctx* ctx = new (std::nothrow) pgpu_ctx();
  if (!ctx) return PGPU_ENOMEM;
  ctx->device = device;
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return PGPU_EDEVICE;
  }
  // the auxiliary streams are made when a plan first forks onto them (aux_stream): a stream costs ~5 ms and
  // ~17 MB of host memory in the runtime, and the default one-launch batch forks only for LCF jobs with an N
  if (hipEventCreateWithFlags(&ctx->ev_upload, hipEventDisableTiming) != hipSuccess) { delete ctx; return PGPU_EDEVICE; }
  if (hipEventCreateWithFlags(&ctx->ev_done, hipEventDisableTiming | hipEventBlockingSync) != hipSuccess) { delete ctx; return PGPU_EDEVICE; }
  if (hipEventCreateWithFlags(&ctx->ev_wait, hipEventDisableTiming) != hipSuccess) { delete ctx; return PGPU_EDEVICE; }
  { const char* f = getenv("PGPU_FANOUT"); ctx->fanout = !(f && f[0] == '0'); }
  { const char* f = getenv("PGPU_WAIT"); if (f) ctx->wait_poll_us = atol(f); }
  { const char* f = getenv("PGPU_ALIGN_COOP"); if (f && f[0] == '0') ctx->align_coop = false; }
  { const char* f = getenv("PGPU_LCF_SA"); if (f && f[0] == '0') ctx->lcf_sa = false; }
  { const char* f = getenv("PGPU_LCF_SA_N"); if (f && f[0] == '0') ctx->lcf_sa_n = false; }
  { const char* f = getenv("PGPU_ALIGN_BAND"); if (f && f[0] == '0') ctx->align_band = false; }
  { const char* f = getenv("PGPU_POISON"); if (f && f[0]) ctx->poison = atoi(f) & 255; }
  { const char* f = getenv("PGPU_PACK"); if (f && atoi(f) == 0) ctx->packed = false; }
  { const char* f = getenv("PGPU_MERGED"); if (f && atoi(f) >= 0 && atoi(f) <= 2) ctx->merged = atoi(f); }
  { const char* f = getenv("PGPU_STREAMS"); const int v = f ? atoi(f) : 0; if (v >= 1 && v <= pgpu_ctx::NAUX) { ctx->n_aux = v; ctx->packed = false; } }
  base_event(device, ctx->stream);
  *out = ctx;
  return PGPU_OK;
}

extern "C" int pgpu_destroy(pgpu_ctx* ctx) {
  if (!ctx) return PGPU_EINVAL;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  for (auto& pl : ctx->pools) for (auto& q : pl.ptr) if (q) hipFree(q);
  for (auto& a : ctx->aux) if (a) { hipStreamSynchronize(a); hipStreamDestroy(a); }
  if (ctx->ev_upload) hipEventDestroy(ctx->ev_upload);
  if (ctx->ev_done) hipEventDestroy(ctx->ev_done);
  if (ctx->ev_wait) hipEventDestroy(ctx->ev_wait);
  for (auto& e : ctx->ev_aux) if (e) hipEventDestroy(e);
  for (auto& e : ctx->ev_pool) if (e) hipEventDestroy(e);
  for (auto& q : ctx->pin) if (q) hipHostFree(q);
  hipStreamDestroy(ctx->stream);
  delete ctx;
  return PGPU_OK;
}

extern "C" const char* pgpu_last_error(const pgpu_ctx* ctx) { return ctx ? ctx->err : "null context"; }

// ---------------------------------------------------------------------------------------------
// DP plans
// ---------------------------------------------------------------------------------------------
namespace {

struct Group {
  int family;            // KernelFamily
  int kind;              // pgpu_dp_kind the jobs came from
  int R;                 // row class (0 when not applicable)
  size_t first, count;   // slice of the sorted device job table
  uint32_t max_chunks = 0, max_l2 = 0;
  uint32_t n_big = 0;          // leading jobs of the large row classes (ED, ALIGN, KBAND: above 16 rows per lane; GAP: above 4)
  uint32_t max_rows = 0;       // largest a_len of the group (LDS of the one-job-per-workgroup BORDERS kernel)
  bool traceback = false;      // this group is the traceback pass of (family)
  bool launched = false;       // its events were recorded by the last launch
  bool in_merged = false;      // its common row classes run inside the plan's merged launch; what is left here is the BIG part
  uint64_t cells = 0, algo_bytes = 0;
  uint64_t cells_big = 0, algo_big = 0;     // share of the first n_big jobs
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  float ms = 0.f;
  float t0_ms = -1.f;          // start of the last launch on the device's time line (base_event), -1: unknown
  std::string name;
};

uint32_t row_class(uint32_t rows) {         // smallest R in {1,2,4,...,64} with 64*R >= rows
  if (rows > 4096u) return ROW_CLASS_STRIPS;  // beyond that: strips of 4096 rows on the R = 64 kernels
  uint32_t R = 1;
  while (64u * R < rows) R <<= 1;
  return R;
}

}  // namespace

struct pgpu_dp_plan {
  size_t n_jobs = 0;
  std::vector<Group> groups;
  std::vector<pgpu_dp_result> prefill;       // status for jobs that never reach the device
  DevJob* d_jobs = nullptr;
  DevResult* d_results = nullptr;
  uint8_t* d_arena = nullptr;
  uint8_t* d_ws = nullptr;
  uint8_t* d_strs = nullptr;
  unsigned long long* d_keys = nullptr;
  size_t n_dev_jobs = 0, ws_bytes = 0, strs_bytes = 0, n_keys = 0;
  uint64_t cells[PGPU_DP_NKINDS] = {0}, algo_bytes[PGPU_DP_NKINDS] = {0};
  double ms[PGPU_DP_NKINDS] = {0};
  uint64_t launches[PGPU_DP_NKINDS] = {0};
  bool launched = false, synced = false;
  bool pooled = false;                      // device + pinned buffers and events borrowed from the context
  pgpu_ctx* owner = nullptr;
  // One device allocation: [arena | job table | LCF keys | results | strings | traceback workspace].
  // The first four are filled by ONE host->device copy from the pinned image `h_up`; results and
  // strings come back in ONE device->host copy into `h_down`, enqueued by launch after the kernels.
  uint8_t* d_base = nullptr;
  uint8_t* h_up = nullptr; uint8_t* h_down = nullptr;
  size_t up_bytes = 0, off_results = 0, down_bytes = 0;
  // merged launch (wave_jobs_kernel): segments = the common row classes of the wave-per-job families
  int n_segs = 0, seg_family[MAX_WAVE_SEGS] = {0}, seg_start[MAX_WAVE_SEGS] = {0}, seg_count[MAX_WAVE_SEGS] = {0};
  LcfIndexView lcf_ix{};       // the index view the suffix-array LCF jobs search (one index per plan)
  int merged_group = -1;       // index of the pseudo group that carries its timing and accounting
  // ... and, with the batch kernel, the one-job-per-workgroup BORDERS / AFFIX jobs
  bool batch = false;
  int bc_start = 0, bc_count = 0, ac_start = 0, ac_count = 0, lc_start = 0, lc_count = 0;   // BORDERS / AFFIX / ALIGN on several waves
  int ab_start = 0, ab_count = 0;              // banded ALIGN jobs of the merged launch (the follow-up launch looks at them)
  uint32_t bc_max_rows = 0;
  // LCF: the kernel leaves one 64-bit key per job directly in front of the results; they come back in the
  // same copy and sync turns them into results (lcf_out[k] = caller index of the job of key k)
  std::vector<uint32_t> lcf_out;
  size_t off_keys = 0;
  bool lcf_decoded = false;
};

static void plan_free(pgpu_dp_plan* p) {
  if (!p) return;
  if (p->pooled) pgpu_ctx_pool_release(p->owner, 0);
  else {
    for (auto& g : p->groups) {
      if (g.ev0) hipEventDestroy(g.ev0);
      if (g.ev1) hipEventDestroy(g.ev1);
    }
    if (p->d_base) hipFree(p->d_base);
    if (p->h_up) hipHostFree(p->h_up);
    if (p->h_down) hipHostFree(p->h_down);
  }
  delete p;
}

extern "C" int pgpu_dp_plan_create(pgpu_ctx* ctx, const pgpu_index* idx, const pgpu_dp_job* jobs,
                                   size_t n_jobs, const char* arena, size_t arena_len,
                                   pgpu_dp_plan** out) {
  const pgpu_dp_part one = { jobs, n_jobs, arena, arena_len };
  return pgpu_dp_plan_create_parts(ctx, idx, &one, 1, out);
}

extern "C" int pgpu_dp_plan_create_parts(pgpu_ctx* ctx, const pgpu_index* idx, const pgpu_dp_part* parts,
                                         size_t n_parts, pgpu_dp_plan** out) {
  if (!ctx || !out || (n_parts && !parts)) return set_err(ctx, PGPU_EINVAL, "bad argument");
  *out = nullptr;
  size_t n_jobs = 0, arena_len = 0;
  for (size_t q = 0; q < n_parts; ++q) {
    if ((parts[q].n_jobs && !parts[q].jobs) || (parts[q].arena_len && !parts[q].arena)) return set_err(ctx, PGPU_EINVAL, "bad argument");
    n_jobs += parts[q].n_jobs; arena_len += parts[q].arena_len;
  }
  if (n_jobs > 0x7fffffffu) return set_err(ctx, PGPU_EINVAL, "too many jobs");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  pgpu_dp_plan* p = new (std::nothrow) pgpu_dp_plan();
  if (!p) return set_err(ctx, PGPU_ENOMEM, "out of host memory");
  p->n_jobs = n_jobs;
  p->prefill.assign(n_jobs, pgpu_dp_result{});

  const uint8_t* d_gen = idx ? pgpu_index_genomic(idx) : nullptr;
  const size_t gen_len = idx ? pgpu_index_length(idx) : 0;
  p->lcf_ix = pgpu_index_lcf_view(idx);
  const bool lcf_sa = ctx->lcf_sa && idx && p->lcf_ix.focc && p->lcf_ix.rmq;
  const bool align_coop = ctx->align_coop;
  p->owner = ctx;
  p->pooled = pgpu_ctx_pool_acquire(ctx, 0);

  // pass 1: validate, classify; operands are addressed by OFFSET until the arena's place is known
  struct Keyed { DevJob j; uint64_t a_off, b_off; bool ag, bg; int family; int kind; uint32_t R; uint64_t size; };
  std::vector<Keyed> v;
  v.reserve(n_jobs);
  // jobs of part q address part q's arena; the arenas are laid out one after the other
  size_t i = 0, part_base = 0, part_i = 0, part_left = n_parts ? parts[0].n_jobs : 0;
  for (; i < n_jobs; ++i, --part_left) {
    while (part_left == 0) { part_base += parts[part_i].arena_len; ++part_i; part_left = parts[part_i].n_jobs; }
    const pgpu_dp_job& in = parts[part_i].jobs[parts[part_i].n_jobs - part_left];
    const size_t own_arena = parts[part_i].arena_len;
    pgpu_dp_result& pre = p->prefill[i];
    pre.status = PGPU_EINVAL;
    if (in.kind >= PGPU_DP_NKINDS) continue;
    const bool ag = in.flags & PGPU_JOB_A_GENOMIC, bg = in.flags & PGPU_JOB_B_GENOMIC;
    if ((ag || bg) && !d_gen) continue;
    const size_t a_space = ag ? gen_len : own_arena, b_space = bg ? gen_len : own_arena;
    if (in.a_off > a_space || in.a_len > a_space - in.a_off) continue;
    if (in.b_off > b_space || in.b_len > b_space - in.b_off) continue;
    Keyed k{};
    k.a_off = in.a_off + (ag ? 0 : part_base); k.b_off = in.b_off + (bg ? 0 : part_base); k.ag = ag; k.bg = bg;
    k.j.la = in.a_len; k.j.lb = in.b_len;
    k.j.p0 = in.p0; k.j.p1 = in.p1; k.j.p2 = in.p2; k.j.tail = in.tail;
    k.j.out_idx = (uint32_t)i;
    k.kind = (int)in.kind;
    pre.status = PGPU_ERANGE;
    const uint32_t la = in.a_len, lb = in.b_len;
    switch (in.kind) {
      case PGPU_DP_ALIGN:
        if (la > PGPU_MAX_ROWS_LEV || lb > PGPU_MAX_COLS) continue;
        k.family = KF_ALIGN; k.R = row_class(la); k.size = (uint64_t)la * lb;
        // exon against its stretch of the genomic sequence: the alignment hugs the diagonal.  Above 64 rows, lengths
        // within the band's half-width: inside the band on ONE wave among the wave-per-job jobs (four to a workgroup
        // instead of a workgroup each); the rare job the band cannot settle is finished by the follow-up launch
        if (ctx->align_band && align_coop && ctx->merged >= 1 && la > 64u && la <= 4096u && (la > lb ? la - lb : lb - la) <= ALIGN_BAND_HALF)
          k.family = KF_ALIGNB;
        k.j.tail = 0u;
        break;
      case PGPU_DP_GAP:
        if (lb > PGPU_MAX_GAP_SIDE || la > PGPU_MAX_GAP_SIDE || ((uint64_t)la + 1) * ((uint64_t)lb + 1) > PGPU_MAX_GAP_CELLS) continue;
        // beyond 2048 rows: the anti-diagonal kernel over HBM (slow, but the reference computes these too)
        k.family = KF_GAP; k.R = la > PGPU_MAX_ROWS_GAP ? ROW_CLASS_STRIPS : row_class(la); k.size = (uint64_t)la * lb; break;
      case PGPU_DP_ED:
        if (std::min(la, lb) > PGPU_MAX_ROWS_LEV || std::max(la, lb) > PGPU_MAX_COLS) continue;
        k.family = KF_ED; k.R = row_class(std::min(la, lb)); k.size = (uint64_t)la * lb; break;
      case PGPU_DP_KBAND: {
        const uint32_t n = std::max(la, lb), m = std::min(la, lb), ub = in.p0;
        if (n > PGPU_MAX_COLS || m > PGPU_MAX_ROWS_LEV) continue;
        k.family = KF_KBAND; k.R = row_class(m);
        k.size = (2ull * ub + 1 >= n) ? (uint64_t)la * lb : (uint64_t)m * (2ull * ub + 1);
        break;
      }
      case PGPU_DP_LCF: {
        if (lb > 65535u || la >= (1u << 28)) continue;
        k.family = KF_LCF; k.R = 0; k.size = (uint64_t)la * lb;
        // the small-exon search's question -- a prefix of the genomic sequence against a few dozen EST
        // characters -- is answered from the suffix array when exact matching is all there is to it: both
        // strings upper-case ACGT (no N wildcard can fire) and s2 short enough for one lane per start
        // ... or with ONE N in s2 (an EST with a stray N): the four strings with A, C, G, T in its place are searched
        // by the same wave (lcfsa_wave_body; job.p0 = the N's place + 1).  More Ns, or anything else: the DP kernel
        if (lcf_sa && ag && !bg && in.a_off == 0 && la <= p->lcf_ix.first_bad && lb <= 64u) {
          const unsigned char* b = (const unsigned char*)parts[part_i].arena + in.b_off;
          bool acgt = true;
          uint32_t n_wild = 0, wild_at = 0;
          for (uint32_t q = 0; q < lb && acgt; ++q) {
            if (b[q] == 'N' || b[q] == 'n') { ++n_wild; wild_at = q; }
            else acgt = b[q] == 'A' || b[q] == 'C' || b[q] == 'G' || b[q] == 'T';
          }
          if (acgt && n_wild <= (ctx->lcf_sa_n ? 1u : 0u)) { k.family = KF_LCFSA; k.j.p0 = n_wild ? wild_at + 1u : 0u; }
        }
        // two short strings: one wave (a lane per diagonal) instead of a workgroup and an atomic per job
        if (k.family == KF_LCF && ctx->lcf_sa && (uint64_t)la * lb <= 16384u && la + lb <= 4096u) k.family = KF_LCFW;
        break;
      }
      case PGPU_DP_BORDERS:
        // only the first and the last t_win = min(len_p + max_errs, len_t) characters of t are swept
        // (src/refine.c:117-121): t may be a whole intron of any length the index can hold
        if (la > PGPU_MAX_ROWS_LEV || std::min<uint64_t>((uint64_t)la + in.p2, lb) > PGPU_MAX_COLS) continue;
        if (in.p0 > in.p1 || in.p1 > la) { pre.status = PGPU_EINVAL; continue; }
        // cells as the reference bounds them: two matrices of len_p x t_win, t_win = min(len_p + max_errs, len_t)
        // (src/refine.c:117-121); the gap of check_gap_errors is a whole intron, the window a few dozen columns
        k.family = KF_BORDERS; k.R = row_class(la); k.size = (uint64_t)la * std::min<uint64_t>((uint64_t)la + in.p2, lb); break;
      case PGPU_DP_AFFIX:
        if (la > PGPU_MAX_ROWS_LEV || lb > PGPU_MAX_COLS) continue;
        k.family = KF_AFFIX; k.R = row_class(la); k.size = (uint64_t)la * lb; break;
      default: continue;
    }
    k.j.r_class = k.R;
    pre.status = PGPU_OK;
    v.push_back(k);
  }
  // order: family, row class, then largest first (long jobs start early; waves of a workgroup
  // and threads of a traceback wave get similar sizes)
  // A counting sort does it in three passes: 13-bit key = family, row class (descending) and the
  // size reduced to its binary order of magnitude (descending); jobs with the same key keep the
  // caller's order.
  {
    auto key_of = [](const Keyed& k) -> uint32_t {
      uint32_t rcls = 0;
      while ((1u << rcls) < k.R) ++rcls;                                  // R = 0,1,2,4,...,128 -> 0..7
      const uint32_t mag = 63u - (uint32_t)__builtin_clzll(k.size | 1ull);  // floor(log2(size)), 0..63
      return ((uint32_t)k.family << 10) | ((7u - rcls) << 6) | (63u - mag);   // long jobs first
    };
    constexpr uint32_t NKEYS = (uint32_t)KF_COUNT << 10;
    std::vector<uint32_t> start(NKEYS + 1, 0);
    std::vector<uint32_t> kk(v.size());
    for (size_t q = 0; q < v.size(); ++q) { kk[q] = key_of(v[q]); ++start[kk[q] + 1]; }
    for (uint32_t c = 0; c < NKEYS; ++c) start[c + 1] += start[c];
    std::vector<Keyed> sorted;
    sorted.resize(v.size());
    for (size_t q = 0; q < v.size(); ++q) sorted[start[kk[q]]++] = v[q];
    v.swap(sorted);
  }
  // workspaces and string slots
  size_t ws = 0, strs = 0, nkeys = 0;
  for (auto& k : v) {
    const uint32_t la = k.j.la, lb = k.j.lb;
    if (k.R == ROW_CLASS_STRIPS && k.family == KF_GAP) {          // gap_slow_kernel: 9 rolling diagonals + 1 B per cell
      k.j.ws_off = ws; ws += (size_t)9 * ((size_t)la + 1) * 4 + ((size_t)la + 1) * ((size_t)lb + 1);
      k.j.str_off = strs; strs += 2 * ((size_t)la + lb + 1);
    } else if (k.R == ROW_CLASS_STRIPS && k.family == KF_BORDERS) {   // borders_slow_kernel: 3 diagonals + 4 row-minima arrays
      k.j.ws_off = ws; ws += (size_t)7 * ((size_t)la + 1) * 4;
    } else if (k.R == ROW_CLASS_STRIPS) {     // two boundary rows, then (ALIGN) the directions of every strip
      const uint32_t nc = (k.family == KF_ED || k.family == KF_KBAND) ? std::max(la, lb) : lb;
      k.j.ws_off = ws; ws += 2 * strip_bnd_bytes(nc);
      if (k.family == KF_ALIGN) {
        ws += (size_t)((la + 4095u) / 4096u) * ((size_t)lb + 64) * 64 * align_entry_bytes(k.R);
        k.j.str_off = strs; strs += 2 * ((size_t)la + lb + 1);
      }
    } else if ((k.family == KF_ALIGN || k.family == KF_ALIGNB) && k.R >= 2 && ctx->align_coop) {   // four waves: [step][256 lanes] entries (the band's words fit in there)
      k.j.ws_off = ws; ws += ((size_t)lb + 256) * 256 * align_coop_entry_bytes(k.R);
      k.j.str_off = strs; strs += 2 * ((size_t)la + lb + 1);
    } else if (k.family == KF_ALIGN) {
      k.j.ws_off = ws; ws += ((size_t)lb + 64) * 64 * align_entry_bytes(k.R);
      k.j.str_off = strs; strs += 2 * ((size_t)la + lb + 1);
    } else if (k.family == KF_GAP) {
      k.j.ws_off = ws; ws += ((size_t)lb + 64) * 64 * gap_entry_bytes(k.R);
      k.j.str_off = strs; strs += 2 * ((size_t)la + lb + 1);
    } else if (k.family == KF_LCF) {
      ++nkeys;
      p->lcf_out.push_back(k.j.out_idx);
    }                                           // (KF_LCFSA: no workspace, the wave writes its result itself)
    ws = (ws + 15) & ~(size_t)15;
  }
  p->ws_bytes = ws; p->strs_bytes = strs; p->n_keys = nkeys; p->n_dev_jobs = v.size();

  // launch groups
  auto cells_of = [](const Keyed& k) -> uint64_t { return k.size; };
  i = 0;
  while (i < v.size()) {
    size_t j = i;
    // a launch group = a family; BORDERS and AFFIX split into (up to 64 rows: one wave per job),
    // (more rows: one job per workgroup) and, AFFIX only, (beyond 4096 rows: strips)
    auto variant = [align_coop](const Keyed& k) -> int {
      if (k.family == KF_GAP) return k.R == ROW_CLASS_STRIPS ? (int)ROW_CLASS_STRIPS : 0;
      if (k.family != KF_BORDERS && k.family != KF_AFFIX && !(k.family == KF_ALIGN && align_coop)) return 0;
      return k.R == 1 ? 1 : (k.R == ROW_CLASS_STRIPS ? (int)ROW_CLASS_STRIPS : 0);
    };
    Group g{};
    g.family = v[i].family; g.kind = v[i].kind; g.R = variant(v[i]); g.first = i;
    while (j < v.size() && v[j].family == g.family && variant(v[j]) == g.R &&
           (g.family != KF_LCF || j - i < 65535)) {
      const Keyed& k = v[j];
      const uint64_t la = k.j.la, lb = k.j.lb;
      // cells = what the kernels compute (the roofline's numerator).  The suffix-array search computes none of the
      // 46 x |prefix| cells the reference's loop bounds give for these jobs: they are left out rather than credited
      const uint64_t job_cells = k.family == KF_LCFSA ? 0 : (k.family == KF_GAP ? 3 : (k.family == KF_BORDERS ? 2 : 1)) * cells_of(k);
      // algorithmic HBM bytes: operands once; 1 B/cell of directions for
      // ALIGN, 3 B/cell for GAP; 8 B per row when row minima are produced
      // (BORDERS touches the first and the last t_win characters of t only)
      uint64_t job_bytes = la + (k.family == KF_BORDERS ? std::min<uint64_t>(lb, 2 * std::min<uint64_t>(la + k.j.p2, lb)) : lb);
      if (k.family == KF_ALIGN || k.family == KF_ALIGNB) job_bytes += la * lb + 3 * (la + lb);     // + directions read back, two strings
      if (k.family == KF_GAP) job_bytes += 3 * la * lb + 3 * (la + lb);
      if (k.family == KF_BORDERS) job_bytes += 2 * 8 * la;
      // suffix-array search: s2, and per lane about eight table probes, a few suffix-array / sequence probes of the
      // bisection and the range minima, and the characters of one direct comparison
      if (k.family == KF_LCFSA) job_bytes = lb + 64u * 96u;
      g.cells += job_cells; g.algo_bytes += job_bytes;
      g.max_rows = std::max(g.max_rows, (uint32_t)la);
      if (k.family != KF_ALIGNB && k.R >= (k.family == KF_GAP ? 8u : 32u)) { ++g.n_big; g.cells_big += job_cells; g.algo_big += job_bytes; }   // (the slow GAP / BORDERS groups ignore it)
      if (k.family == KF_LCF) {
        const uint32_t ch = (uint32_t)((la + lb + 254) / 256);   // ceil((la+lb-1)/256) diagonals
        g.max_chunks = std::max(g.max_chunks, ch);
        g.max_l2 = std::max(g.max_l2, (uint32_t)lb);
      }
      ++j;
    }
    g.count = j - i;
    char nm[64];
    static const char* fam[] = {"lev_wave<ALIGN>", "gap_wave", "lev_wave<ED>", "lev_wave<KBAND>", "lcf",
                                "borders_coop", "affix_coop", "lcf_sa", "lcf_small", "align_band"};
    if (g.family == KF_BORDERS && g.R == 1) snprintf(nm, sizeof nm, "lev_wave<BORDERS,R=1>");
    else if (g.family == KF_AFFIX && g.R == 1) snprintf(nm, sizeof nm, "lev_wave<AFFIX,R=1>");
    else if (g.family == KF_AFFIX && g.R == (int)ROW_CLASS_STRIPS) snprintf(nm, sizeof nm, "lev_wave<AFFIX,strips>");
    else if (g.family == KF_BORDERS && g.R == (int)ROW_CLASS_STRIPS) snprintf(nm, sizeof nm, "borders_slow");
    else if (g.family == KF_GAP && g.R == (int)ROW_CLASS_STRIPS) snprintf(nm, sizeof nm, "gap_slow");
    else if (g.family == KF_ALIGN && align_coop && g.R == 0) snprintf(nm, sizeof nm, "align_coop");
    else if (g.family == KF_ALIGN && align_coop && g.R == (int)ROW_CLASS_STRIPS) snprintf(nm, sizeof nm, "lev_wave<ALIGN,strips>");
    else snprintf(nm, sizeof nm, "%s", fam[g.family]);
    g.name = nm;
    p->groups.push_back(g);
    i = j;
  }
  for (auto& g : p->groups) {
    p->cells[g.kind] += g.cells;
    p->algo_bytes[g.kind] += g.algo_bytes;
  }
  if (ctx->merged) {
    // the common row classes of the wave-per-job families run in one launch (wave_jobs_kernel); the
    // groups keep their BIG part (first n_big jobs).  Long-running families first.
    static const int order_fam[9] = { KF_ALIGNB, KF_ALIGN, KF_GAP, KF_KBAND, KF_BORDERS, KF_AFFIX, KF_LCFSA, KF_LCFW, KF_ED };
    Group m{};
    m.family = KF_COUNT; m.kind = PGPU_DP_ALIGN; m.name = "wave_jobs";
    for (int f : order_fam)
      for (auto& g : p->groups) {
        if (g.family != f || g.traceback) continue;
        const bool split = f == KF_BORDERS || f == KF_AFFIX || (f == KF_ALIGN && align_coop);   // one wave up to 64 rows, several above
        const bool wave_family = split ? g.R == 1 : g.R == 0;
        if (!wave_family || p->n_segs >= MAX_WAVE_SEGS) continue;
        const size_t big = split ? 0 : g.n_big;
        if (g.count <= big) continue;
        p->seg_family[p->n_segs] = f; p->seg_start[p->n_segs] = (int)(g.first + big); p->seg_count[p->n_segs] = (int)(g.count - big);
        ++p->n_segs;
        if (f == KF_ALIGNB) { p->ab_start = (int)g.first; p->ab_count = (int)g.count; }
        m.count += g.count - big; m.cells += g.cells - (big ? g.cells_big : 0); m.algo_bytes += g.algo_bytes - (big ? g.algo_big : 0);
        g.in_merged = true;
        g.count = big; g.cells = big ? g.cells_big : 0; g.algo_bytes = big ? g.algo_big : 0;
      }
    if (ctx->merged >= 2) {
      // the one-job-per-workgroup sweeps join it (dp_batch_kernel) unless the largest BORDERS pattern
      // needs more LDS than the roles may share
      for (auto& g : p->groups) {
        if (g.traceback || g.R != 0 || g.count == 0 || g.count > 0x3fffffffu) continue;
        if (g.family == KF_BORDERS && dp_batch_lds_bytes(true, 1, g.max_rows, 1, 1) <= 64 * 1024) {
          p->bc_start = (int)g.first; p->bc_count = (int)g.count; p->bc_max_rows = g.max_rows;
        } else if (g.family == KF_AFFIX) {
          p->ac_start = (int)g.first; p->ac_count = (int)g.count;
        } else if (g.family == KF_ALIGN && align_coop) {
          p->lc_start = (int)g.first; p->lc_count = (int)g.count;
        } else continue;
        m.count += g.count; m.cells += g.cells; m.algo_bytes += g.algo_bytes;
        g.in_merged = true; g.count = 0; g.cells = 0; g.algo_bytes = 0;
        p->batch = true;
      }
      if (p->n_segs) p->batch = true;
      if (p->batch) m.name = "dp_batch";
    }
    if (p->n_segs || p->batch) { p->merged_group = (int)p->groups.size(); p->groups.push_back(m); }
  }
  if (ctx->timing) {
    const size_t need = 2 * p->groups.size();
    if (p->pooled) {
      while (ctx->ev_pool.size() < need) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) { plan_free(p); return set_err(ctx, PGPU_EDEVICE, "hipEventCreate failed"); }
        ctx->ev_pool.push_back(e);
      }
      for (size_t g = 0; g < p->groups.size(); ++g) { p->groups[g].ev0 = ctx->ev_pool[2 * g]; p->groups[g].ev1 = ctx->ev_pool[2 * g + 1]; }
    } else {
      for (auto& g : p->groups)
        if (hipEventCreate(&g.ev0) != hipSuccess || hipEventCreate(&g.ev1) != hipSuccess) {
          plan_free(p);
          return set_err(ctx, PGPU_EDEVICE, "hipEventCreate failed");
        }
    }
  }

  // one device allocation, one pinned upload image, one pinned download image
  auto up256 = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t off_arena = 0;
  const size_t off_jobs = up256(arena_len + 16);
  const size_t off_keys = up256(off_jobs + std::max<size_t>(v.size(), 1) * sizeof(DevJob));
  const size_t off_results = up256(off_keys + (nkeys + 1) * sizeof(unsigned long long));
  const size_t off_strs = off_results + std::max<size_t>(n_jobs, 1) * sizeof(DevResult);   // directly behind the results
  const size_t off_ws = up256(off_strs + strs + 16);
  const size_t total = off_ws + ws + 16;
  p->up_bytes = off_strs;                      // arena .. results (prefilled)
  p->off_results = off_results;
  p->off_keys = off_keys;
  p->down_bytes = (off_strs - off_keys) + strs;           // keys (+ padding), results, strings
  if (p->pooled) {
    p->d_base = (uint8_t*)pgpu_ctx_pool_get(ctx, 0, 0, total);
    p->h_up = (uint8_t*)pinned(ctx, 0, p->up_bytes);
    p->h_down = (uint8_t*)pinned(ctx, 1, p->down_bytes + 16);
  } else {
    void* q = nullptr;
    if (hipMalloc(&q, total) == hipSuccess) p->d_base = (uint8_t*)q;
    if (hipHostMalloc(&q, p->up_bytes, hipHostMallocDefault) == hipSuccess) p->h_up = (uint8_t*)q;
    if (hipHostMalloc(&q, p->down_bytes + 16, hipHostMallocDefault) == hipSuccess) p->h_down = (uint8_t*)q;
  }
  if (!p->d_base || !p->h_up || !p->h_down) {
    plan_free(p);
    return set_err(ctx, PGPU_ENOMEM, "batch buffers (%zu B device, %zu B pinned) could not be allocated", total, p->up_bytes + p->down_bytes);
  }
  p->d_arena = p->d_base + off_arena;
  p->d_jobs = (DevJob*)(p->d_base + off_jobs);
  p->d_keys = (unsigned long long*)(p->d_base + off_keys);
  p->d_results = (DevResult*)(p->d_base + off_results);
  p->d_strs = p->d_base + off_strs;
  p->d_ws = p->d_base + off_ws;

  // the upload image: operands, the sorted job table with device pointers, zeroed LCF keys,
  // prefilled results (status of the jobs that never reach the device)
  {
    size_t at = off_arena;
    for (size_t q = 0; q < n_parts; ++q) { if (parts[q].arena_len) memcpy(p->h_up + at, parts[q].arena, parts[q].arena_len); at += parts[q].arena_len; }
  }
  DevJob* hj = (DevJob*)(p->h_up + off_jobs);
  for (size_t q = 0; q < v.size(); ++q) {
    DevJob d = v[q].j;
    d.a = (v[q].ag ? d_gen : p->d_arena) + v[q].a_off;
    d.b = (v[q].bg ? d_gen : p->d_arena) + v[q].b_off;
    hj[q] = d;
  }
  memset(p->h_up + off_keys, 0, off_results - off_keys);
  if (n_jobs) memcpy(p->h_up + off_results, p->prefill.data(), n_jobs * sizeof(DevResult));
  const hipError_t e = hipMemcpyAsync(p->d_base, p->h_up, p->up_bytes, hipMemcpyHostToDevice, ctx->stream);
  if (e != hipSuccess) { plan_free(p); return set_err(ctx, PGPU_EDEVICE, "upload failed: %s", hipGetErrorString(e)); }
  if (ctx->poison >= 0) {
    // debugging aid (PGPU_POISON=<byte>): what the kernels are going to write -- strings, traceback workspace --
    // starts from a known pattern instead of whatever the previous batch left there; an answer that changes
    // with the pattern comes from a read of memory nothing wrote (tools/stress_parity.py --poison)
    if (hipMemsetAsync(p->d_base + off_strs, ctx->poison, total - off_strs, ctx->stream) != hipSuccess) { plan_free(p); return set_err(ctx, PGPU_EDEVICE, "poison fill failed"); }
  }
  *out = p;
  return PGPU_OK;
}

extern "C" int pgpu_dp_plan_launch(pgpu_ctx* ctx, pgpu_dp_plan* p) {
  if (!ctx || !p) return set_err(ctx, PGPU_EINVAL, "bad argument");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  p->synced = false; p->lcf_decoded = false;       // a relaunch downloads anew: the image holds raw keys again
  // Launch order: the groups are independent, and the host needs a few microseconds per launch, so
  // the long poles go first (one-job-per-workgroup sweeps with many rows, then the alignments with
  // their tracebacks, ...) and the thousands of tiny edit distances last.  A traceback group
  // directly follows its DP group in p->groups and stays behind it on the same stream.
  std::vector<size_t> order;
  std::vector<size_t> key_of(p->groups.size(), 0);
  {
    size_t kb = 0;
    for (size_t gi = 0; gi < p->groups.size(); ++gi) {
      const Group& g = p->groups[gi];
      if (g.traceback) continue;
      if (g.family == KF_LCF) { key_of[gi] = kb; kb += g.count; }
      if (g.in_merged && g.count == 0) continue;           // nothing left outside the merged launch
      order.push_back(gi);
    }
    auto weight = [&](size_t gi) -> long {
      const Group& g = p->groups[gi];
      switch (g.family) {
        case KF_COUNT: return p->batch ? 20000 : 9000;    // the merged launch: behind the one-job-per-workgroup poles it does not hold
        case KF_BORDERS: case KF_AFFIX: return g.R == 1 ? 150 : 10000 + (long)g.max_rows;
        case KF_ALIGN: if (g.name == "align_coop") return 9000 + (long)g.max_rows; return 5000;
        case KF_GAP: return 4000;
        case KF_LCF: return 3000;
        case KF_LCFSA: case KF_LCFW: return 2500;
        case KF_KBAND: return 2000;
        default: return 1000;
      }
    };
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return weight(a) > weight(b); });
  }
  // HIP spreads its streams over four hardware queues, and what shares a queue runs one after the
  // other.  Packed: four streams (one per queue), the families dealt out so that the four sums of
  // typical durations come out even; the batch launch stays on the main stream (lane -1), where the
  // upload already is, and only the streams a plan uses are forked and joined.  Otherwise
  // round-robin in launch order.
  std::vector<int> lane_of(order.size(), 0);
  unsigned used_mask = 0;
  for (size_t oi = 0; oi < order.size(); ++oi) {
    const Group& g = p->groups[order[oi]];
    int l = (int)(oi % (size_t)ctx->n_aux);
    if (ctx->packed) {
      const bool one_wave = g.R == 1;          // BORDERS / AFFIX with up to 64 rows
      switch (g.family) {
        case KF_AFFIX:   l = one_wave ? 1 : 0; break;
        case KF_BORDERS: l = one_wave ? 0 : 1; break;
        case KF_ED:      l = 1; break;
        case KF_COUNT:   l = p->batch ? -1 : 2; break;
        case KF_ALIGN: case KF_KBAND: l = 2; break;
        default: l = 3; break;           // GAP, LCF
      }
    }
    if (!ctx->fanout) l = -1;
    lane_of[oi] = l;
    if (l >= 0) used_mask |= 1u << l;
  }
  if (used_mask) {            // fork: the upload (plan_create) is on the main stream
    HIP_TRY(ctx, hipEventRecord(ctx->ev_upload, ctx->stream));
    for (int i = 0; i < pgpu_ctx::NAUX; ++i)
      if (used_mask & (1u << i)) {
        if (!aux_stream(ctx, i)) return set_err(ctx, PGPU_EDEVICE, "auxiliary stream %d could not be created", i);
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->aux[i], ctx->ev_upload, 0));
      }
  }
  for (size_t oi = 0; oi < order.size(); ++oi) {
    for (size_t gi = order[oi]; gi < p->groups.size() && (gi == order[oi] || p->groups[gi].traceback); ++gi) {
      Group& g = p->groups[gi];
      const DevJob* jobs = p->d_jobs + g.first;
      const int n = (int)g.count;
      hipStream_t st = lane_of[oi] >= 0 ? ctx->aux[lane_of[oi]] : ctx->stream;
      if (g.ev0) HIP_TRY(ctx, hipEventRecord(g.ev0, st));
      g.launched = true;
      pgpu_range_push(g.name.c_str());
      struct PopAtExit { ~PopAtExit() { pgpu_range_pop(); } } pop_at_exit;
      switch (g.family) {
        case KF_COUNT:
          if (p->batch) {
            if (!launch_dp_batch(p->d_jobs, p->n_segs, p->seg_family, p->seg_start, p->seg_count, p->bc_start, p->bc_count,
                                 p->bc_max_rows, p->ac_start, p->ac_count, p->lc_start, p->lc_count, p->d_results, p->d_ws, p->d_strs, p->lcf_ix, st))
              return set_err(ctx, PGPU_EDEVICE, "batch launch: LDS budget exceeded");
          } else {
            launch_wave_jobs(p->d_jobs, p->n_segs, p->seg_family, p->seg_start, p->seg_count, p->d_results, p->d_ws, p->d_strs, p->lcf_ix, st);
          }
          break;
        case KF_ALIGN: case KF_ED: case KF_BORDERS: case KF_AFFIX: case KF_KBAND:
          launch_lev(g.family, g.R, g.max_rows, jobs, n, (int)g.n_big, p->d_results, p->d_ws, p->d_strs, st); break;
        case KF_GAP:
          if (g.R == (int)ROW_CLASS_STRIPS) launch_gap_slow(jobs, n, p->d_results, p->d_ws, p->d_strs, st);
          else launch_gap(jobs, n, (int)g.n_big, p->d_results, p->d_ws, p->d_strs, st);
          break;
        case KF_LCF:
          launch_lcf(jobs, n, g.max_chunks, g.max_l2, p->d_keys + key_of[gi], st);
          break;
        case KF_ALIGNB: {                     // (only when the merged launch had no segment left for it)
          const int fam1 = g.family, start1 = (int)g.first, count1 = n;
          launch_wave_jobs(p->d_jobs, 1, &fam1, &start1, &count1, p->d_results, p->d_ws, p->d_strs, p->lcf_ix, st);
          launch_align_fallback(jobs, n, p->d_results, p->d_ws, p->d_strs, st);
          break;
        }
        case KF_LCFSA: case KF_LCFW: {       // a launch of their own only with PGPU_MERGED=0
          const int fam1 = g.family, start1 = (int)g.first, count1 = n;
          launch_wave_jobs(p->d_jobs, 1, &fam1, &start1, &count1, p->d_results, p->d_ws, p->d_strs, p->lcf_ix, st);
          break;
        }
        default: break;
      }
      if (g.ev1) HIP_TRY(ctx, hipEventRecord(g.ev1, st));
      // behind the merged launch (and outside its pair of events, which time the one kernel the profiler lists as
      // dp_batch_kernel): the whole-matrix sweep of the banded alignments it could not settle
      if (g.family == KF_COUNT) launch_align_fallback(p->d_jobs + p->ab_start, p->ab_count, p->d_results, p->d_ws, p->d_strs, st);
      HIP_TRY(ctx, hipGetLastError());
    }
  }
  for (int i = 0; i < pgpu_ctx::NAUX; ++i) {      // join: the main stream continues after every stream that was used
    if (!(used_mask & (1u << i))) continue;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_aux[i], ctx->aux[i]));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_aux[i], 0));
  }
  // results and alignment strings come back in one copy as soon as every group is done
  if (p->down_bytes)
    HIP_TRY(ctx, hipMemcpyAsync(p->h_down, p->d_base + p->off_keys, p->down_bytes, hipMemcpyDeviceToHost, ctx->stream));
  p->launched = true;
  return PGPU_OK;
}

// find_longest_common_factor_dp's answer from the key the kernel left (lcf_key in pgpu_dp_kernels.hip)
static void decode_lcf_keys(pgpu_dp_plan* p) {
  if (p->lcf_decoded) return;
  p->lcf_decoded = true;
  const unsigned long long* keys = (const unsigned long long*)p->h_down;
  DevResult* res = (DevResult*)(p->h_down + (p->off_results - p->off_keys));
  for (size_t k = 0; k < p->lcf_out.size(); ++k) {
    DevResult& r = res[p->lcf_out[k]];
    const unsigned long long key = keys[k];
    r.status = PGPU_OK;
    r.v[0] = (int32_t)(key >> 44);
    r.v[1] = key ? (int32_t)(0x0FFFFFFFu - (uint32_t)((key >> 16) & 0x0FFFFFFFu)) : 0;
    r.v[2] = key ? (int32_t)(0xFFFFu - (uint32_t)(key & 0xFFFFu)) : 0;
  }
}

extern "C" int pgpu_dp_plan_sync(pgpu_ctx* ctx, pgpu_dp_plan* p) {
  if (!ctx || !p) return set_err(ctx, PGPU_EINVAL, "bad argument");
  if (wait_stream(ctx, ctx->stream) != 0) return set_err(ctx, PGPU_EDEVICE, "waiting for the batch failed");
  p->synced = p->launched;
  if (p->launched) {
    decode_lcf_keys(p);
    for (int k = 0; k < PGPU_DP_NKINDS; ++k) { p->ms[k] = 0; p->launches[k] = 0; }
    for (auto& g : p->groups) {
      float ms = 0.f;
      if (!g.launched) continue;            // everything of this group ran inside the merged launch
      if (g.ev0 && g.ev1 && hipEventElapsedTime(&ms, g.ev0, g.ev1) == hipSuccess) g.ms = ms;
      if (g.ev0) { hipEvent_t b = base_event(ctx->device, ctx->stream); float t0 = 0.f; g.t0_ms = (b && hipEventElapsedTime(&t0, b, g.ev0) == hipSuccess) ? t0 : -1.f; }
      p->ms[g.kind] += g.ms;
      p->launches[g.kind] += 1;
    }
  }
  return PGPU_OK;
}

extern "C" size_t pgpu_dp_plan_string_bytes(const pgpu_dp_plan* p) { return p ? p->strs_bytes : 0; }

extern "C" int pgpu_dp_plan_fetch(pgpu_ctx* ctx, pgpu_dp_plan* p, pgpu_dp_result* results,
                                  char* strings, size_t cap) {
  if (!ctx || !p || (p->n_jobs && !results)) return set_err(ctx, PGPU_EINVAL, "bad argument");
  if (p->strs_bytes && strings && cap < p->strs_bytes) return set_err(ctx, PGPU_ENOSPC, "string buffer too small: need %zu", p->strs_bytes);
  if (!p->launched) return set_err(ctx, PGPU_EINVAL, "the plan has not been launched");
  // launch enqueued the download behind the kernels; after this wait the pinned image is complete
  if (!p->synced && wait_stream(ctx, ctx->stream) != 0) return set_err(ctx, PGPU_EDEVICE, "result download failed");
  p->synced = true;
  decode_lcf_keys(p);
  const size_t rb = p->n_jobs * sizeof(DevResult);
  if (rb) memcpy(results, p->h_down + (p->off_results - p->off_keys), rb);
  if (p->strs_bytes && strings) memcpy(strings, p->h_down + (p->down_bytes - p->strs_bytes), p->strs_bytes);
  return PGPU_OK;
}

extern "C" int pgpu_dp_plan_results_to_device(pgpu_ctx* ctx, pgpu_dp_plan* p, void* dst, size_t cap) {
  if (!ctx || !p || !dst) return set_err(ctx, PGPU_EINVAL, "bad argument");
  const size_t bytes = p->n_jobs * sizeof(DevResult);
  if (cap < bytes) return set_err(ctx, PGPU_ENOSPC, "device buffer too small: need %zu", bytes);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!p->launched) return set_err(ctx, PGPU_EINVAL, "the plan has not been launched");
  // the complete results are in the pinned image (the LCF answers are decoded there)
  if (!p->synced && wait_stream(ctx, ctx->stream) != 0) return set_err(ctx, PGPU_EDEVICE, "result download failed");
  p->synced = true;
  decode_lcf_keys(p);
  if (bytes) HIP_TRY(ctx, hipMemcpyAsync(dst, p->h_down + (p->off_results - p->off_keys), bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return PGPU_OK;
}

extern "C" int pgpu_dp_plan_destroy(pgpu_ctx* ctx, pgpu_dp_plan* p) {
  if (!ctx || !p) return PGPU_EINVAL;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  plan_free(p);
  return PGPU_OK;
}

extern "C" uint64_t pgpu_dp_plan_cells(const pgpu_dp_plan* p, int kind) {
  return (p && kind >= 0 && kind < PGPU_DP_NKINDS) ? p->cells[kind] : 0;
}
extern "C" uint64_t pgpu_dp_plan_algo_bytes(const pgpu_dp_plan* p, int kind) {
  return (p && kind >= 0 && kind < PGPU_DP_NKINDS) ? p->algo_bytes[kind] : 0;
}
extern "C" double pgpu_dp_plan_kernel_ms(const pgpu_dp_plan* p, int kind) {
  return (p && kind >= 0 && kind < PGPU_DP_NKINDS) ? p->ms[kind] : 0.0;
}
extern "C" uint64_t pgpu_dp_plan_launches(const pgpu_dp_plan* p, int kind) {
  return (p && kind >= 0 && kind < PGPU_DP_NKINDS) ? p->launches[kind] : 0;
}

extern "C" int pgpu_dp_plan_n_groups(const pgpu_dp_plan* p) { return p ? (int)p->groups.size() : 0; }

extern "C" int pgpu_dp_plan_group_info(const pgpu_dp_plan* p, int i, pgpu_group_info* out) {
  if (!p || !out || i < 0 || i >= (int)p->groups.size()) return PGPU_EINVAL;
  const Group& g = p->groups[i];
  memset(out, 0, sizeof(*out));
  snprintf(out->name, sizeof(out->name), "%s", g.name.c_str());
  out->kind = g.kind; out->jobs = g.count; out->cells = g.cells; out->algo_bytes = g.algo_bytes;
  out->ms = g.ms; out->t0_ms = g.t0_ms;
  return PGPU_OK;
}

extern "C" int pgpu_dp_batch(pgpu_ctx* ctx, const pgpu_index* idx, const pgpu_dp_job* jobs,
                             size_t n_jobs, const char* arena, size_t arena_len,
                             pgpu_dp_result* results, char* strings, size_t strings_cap,
                             size_t* strings_used) {
  pgpu_dp_plan* p = nullptr;
  int rc = pgpu_dp_plan_create(ctx, idx, jobs, n_jobs, arena, arena_len, &p);
  if (rc != PGPU_OK) return rc;
  if (strings_used) *strings_used = p->strs_bytes;
  rc = pgpu_dp_plan_launch(ctx, p);
  if (rc == PGPU_OK) rc = pgpu_dp_plan_sync(ctx, p);
  if (rc == PGPU_OK) rc = pgpu_dp_plan_fetch(ctx, p, results, strings, strings_cap);
  pgpu_dp_plan_destroy(ctx, p);
  return rc;
}
