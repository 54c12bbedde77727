/* est-fact for MI355X: drop-in replacement of PIntron's est-fact stage (src/main-est-fact.c).
 * Reads genomic.txt / ests.txt (and config.ini) in the current directory, writes
 * raw-multifasta-out.txt, processed-ests.txt and the MEG side files.  All pairings and dynamic
 * programs run on the GPU through libpintron_gpu.so; without a gfx950 device the program fails. */
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
#include <time.h>

#include "estfact.h"
#include "ef_gpu.h"
#include "ef_sched.h"

static void stamp(const char* what) {              /* PINTRON_VERBOSE: where the process's wall time outside the run goes */
  if (!getenv("PINTRON_VERBOSE")) return;
  struct timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  fprintf(stderr, "* main %s at %ld.%06ld\n", what, (long)ts.tv_sec, ts.tv_nsec / 1000);
}

/* PINTRON_VERBOSE=2: the resident memory of the process by mapping, largest first (what the kernel has to take
 * apart after main: the one-shot wall time counts it) */
static void resident_report(void) {
  const char* v = getenv("PINTRON_VERBOSE");
  if (!v || atoi(v) < 2) return;
  FILE* f = fopen("/proc/self/smaps", "r");
  if (!f) return;
  enum { TOP = 24 };
  struct { char name[96]; unsigned long kb, size_kb; } top[TOP] = {{{0}, 0, 0}}, cur = {{0}, 0, 0};
  unsigned long total = 0;
  char line[512];
  for (;;) {
    char* got = fgets(line, sizeof line, f);
    unsigned long a, b; char perms[8], path[256] = "";
    if (!got || sscanf(line, "%lx-%lx %7s %*s %*s %*s %255[^\n]", &a, &b, perms, path) >= 3) {
      if (cur.kb) {
        total += cur.kb;
        int at = -1;
        for (int k = 0; k < TOP; ++k) if (!strcmp(top[k].name, cur.name) && cur.name[0] != '[') { at = k; break; }
        if (at >= 0) { top[at].kb += cur.kb; top[at].size_kb += cur.size_kb; }
        else { int mn = 0; for (int k = 1; k < TOP; ++k) if (top[k].kb < top[mn].kb) mn = k; if (cur.kb > top[mn].kb) top[mn] = cur; }
      }
      if (!got) break;
      const char* p = path; while (*p == ' ') ++p;
      snprintf(cur.name, sizeof cur.name, "%s", *p ? p : "anonymous");
      if (!*p) snprintf(cur.name, sizeof cur.name, "[anon %lx %s]", a, perms);
      cur.kb = 0; cur.size_kb = (b - a) / 1024;
    } else if (!strncmp(line, "Rss:", 4)) cur.kb = strtoul(line + 4, NULL, 10);
  }
  fclose(f);
  fprintf(stderr, "* resident at the end of main: %lu MB", total / 1024);
  f = fopen("/proc/self/status", "r");
  while (f && fgets(line, sizeof line, f))
    if (!strncmp(line, "RssAnon", 7) || !strncmp(line, "RssFile", 7) || !strncmp(line, "RssShmem", 8) || !strncmp(line, "VmPTE", 5)) {
      line[strcspn(line, "\n")] = 0;
      char* v = strchr(line, ':'); *v++ = 0; while (*v == ' ' || *v == '\t') ++v;
      fprintf(stderr, "  %s %s", line, v);
    }
  if (f) fclose(f);
  fputc('\n', stderr);
  for (int n = 0; n < TOP; ++n) {
    int mx = 0; for (int k = 1; k < TOP; ++k) if (top[k].kb > top[mx].kb) mx = k;
    if (!top[mx].kb) break;
    fprintf(stderr, "*   %7lu MB of %7lu  %s\n", top[mx].kb / 1024, top[mx].size_kb / 1024, top[mx].name);
    top[mx].kb = 0;
  }
}

int main(int argc, char** argv) {
  stamp("entered");
  const char* mode = getenv("PINTRON_ESTFACT_MODE");
  if (mode && !strcmp(mode, "direct")) return ef_run(argc, argv, ef_gpu_open, ef_gpu_close);
  /* PINTRON_CLEAN_EXIT keeps the orderly teardown (profilers that flush at exit need it) */
  if (getenv("PINTRON_CLEAN_EXIT")) return ef_main_multi(argc, argv);
  ef_leave_without_cleanup = 1;
  const int rc = ef_main_multi(argc, argv);      /* --gpus / --genes, else the plain batched run */
  fflush(NULL);
  resident_report();
  stamp("leaves");
  _exit(rc);
}
