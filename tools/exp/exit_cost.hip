// GPU box experiment: what the kernel's teardown of a process costs after _exit, as a function of what the
// process holds.  usage: exit_cost <vram MB> <vram chunks> <pinned MB> <anon MB> <streams> <stack mappings> <threads>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <unistd.h>
#include <sys/mman.h>
#include <pthread.h>
#include <vector>
struct Zap { std::vector<std::pair<char*, size_t>>* regions; int next; };
static void* zap_main(void* a) {
  Zap* z = (Zap*)a;
  for (;;) { const int k = __atomic_fetch_add(&z->next, 1, __ATOMIC_RELAXED); if (k >= (int)z->regions->size()) return nullptr;
             madvise((*z->regions)[k].first, (*z->regions)[k].second, MADV_DONTNEED); }
}
static double now() { timespec ts; clock_gettime(CLOCK_REALTIME, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
__global__ void touch(unsigned* p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = (unsigned)i; }
int main(int argc, char** argv) {
  const size_t vram = argc > 1 ? atol(argv[1]) : 0, chunks = argc > 2 ? atol(argv[2]) : 1, pinned = argc > 3 ? atol(argv[3]) : 0,
               anon = argc > 4 ? atol(argv[4]) : 0, streams = argc > 5 ? atol(argv[5]) : 1;
  const double t0 = now();
  hipFree(0);
  const double t1 = now();
  for (size_t s = 0; s < streams; ++s) { hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking); touch<<<1, 64, 0, st>>>(nullptr, 0); hipStreamSynchronize(st); }
  const double t2 = now();
  for (size_t c = 0; c < chunks && vram; ++c) {
    void* q = nullptr; const size_t bytes = (vram << 20) / chunks;
    if (hipMalloc(&q, bytes) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    touch<<<(unsigned)((bytes / 4 + 255) / 256), 256>>>((unsigned*)q, bytes / 4);
  }
  hipDeviceSynchronize();
  const double t3 = now();
  if (pinned) { void* q; hipHostMalloc(&q, pinned << 20, hipHostMallocDefault); memset(q, 1, pinned << 20); }
  const double t4 = now();
  const size_t stacks = argc > 6 ? atol(argv[6]) : 0;
  std::vector<std::pair<char*, size_t>> regions;
  const size_t zap_threads = argc > 7 ? atol(argv[7]) : 0;
  for (size_t mb = 0; mb < anon; mb += 16) {     // 16 MB mappings, as the record arena makes them
    char* q = (char*)mmap(NULL, 16 << 20, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    for (size_t i = 0; i < (16u << 20); i += 4096) ((volatile char*)q)[i] = 1;
    regions.push_back({q, (size_t)16 << 20});
  }
  for (size_t k = 0; k < stacks; ++k) {        // a fibre stack: 256 KB + guard page, 16 KB of it used
    char* m = (char*)mmap(NULL, 260 << 10, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_STACK, -1, 0);
    if (m == MAP_FAILED) { fprintf(stderr, "mmap failed at %zu\n", k); break; }
    mprotect(m, 4096, PROT_NONE);
    for (size_t i = (260 << 10) - (16 << 10); i < (260 << 10); i += 4096) ((volatile char*)m)[i] = 1;
  }
  const double t5 = now();
  if (zap_threads) {                               // give the pages back from several threads before leaving
    Zap z{&regions, 0}; pthread_t th[64];
    for (size_t t = 0; t < zap_threads && t < 64; ++t) pthread_create(&th[t], NULL, zap_main, &z);
    for (size_t t = 0; t < zap_threads && t < 64; ++t) pthread_join(th[t], NULL);
    fprintf(stderr, "zap %.3f ", now() - t5);
  }
  { FILE* f = fopen("/proc/self/status", "r"); char line[256];
    while (f && fgets(line, sizeof line, f)) if (!strncmp(line, "Rss", 3) || !strncmp(line, "VmRSS", 5)) { line[strcspn(line, "\n")] = 0; fprintf(stderr, "%s; ", line); }
    if (f) fclose(f); }
  fprintf(stderr, "init %.3f streams %.3f vram %.3f pinned %.3f anon %.3f leaves at %.6f\n", t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, now());
  _exit(0);
}
