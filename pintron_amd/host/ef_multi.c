/* est-fact on several GPUs of one node, driven by the C program itself.
 *
 *   est-fact --gpus=N            one gene (cwd): N processes, one per GPU; rank r factorizes a contiguous
 *                                range of the ESTs (ef_load_ests).  What the downstream stages read --
 *                                the factorization records (raw-multifasta-out.txt) and the sequences of
 *                                the aligned ESTs (processed-ests.txt) -- goes to rank 0 in ONE gather
 *                                over RCCL (pgpu_gather) and is written there in rank order = input
 *                                order, so the files are those of a single process.  The MEG side files
 *                                (diagnostics the pipeline driver deletes, dist-scripts/pintron.py:975-983)
 *                                do not travel: the ranks share the directory, so every rank writes its
 *                                part of each file in place, at the offset that follows from the sizes
 *                                exchanged beforehand (PINTRON_SHARD_DIAGNOSTICS=0: not written at all).
 *   est-fact --genes=FILE        FILE lists directories (one per line), each holding genomic.txt and
 *                                ests.txt: gene g runs on rank g mod N and leaves its files in its own
 *                                directory; no exchange (C4 = 8 genes on 8 GPUs)
 * PINTRON_GPUS / PINTRON_GENES set the same from the environment, which is how an unmodified
 * pipeline driver (dist-scripts/pintron.py:878-884 calls est-fact without options) is pointed at
 * several GPUs; INTEGRATION.md section 7 shows the two-line driver patch.
 *
 * The parent starts the other ranks as fresh processes BEFORE anything touches the GPU and then
 * becomes rank 0 itself.
 *
 * Failure of one rank must end all of them (a collective with a missing peer never returns):
 *   - before the communicator exists the ranks agree on their health through marker files next to
 *     the communicator id ("<id>.ok.<r>" / "<id>.fail.<r>"): nobody enters ncclCommInitRank unless
 *     every rank opened its session;
 *   - the first exchange after the step is an all-gather of (status, sizes): a rank whose step
 *     failed is seen by all, and all leave with a non-zero status before any payload moves;
 *   - a rank that dies outright (abort(), a signal) is noticed by the parent's watchdog thread
 *     (waitpid), which ends the other children and the parent; the children ask the kernel to end them
 *     when the parent goes (PR_SET_PDEATHSIG). */
#define _GNU_SOURCE
#include <dirent.h>
#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <signal.h>
#include <spawn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/prctl.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include "estfact.h"
#include "ef_gpu.h"
#include "ef_sched.h"

extern char** environ;

static void nap_ms(long ms) { struct timespec ts = { ms / 1000, (ms % 1000) * 1000000L }; nanosleep(&ts, NULL); }

static int write_all(const char* path, const char* data, size_t len) {
  FILE* f = fopen(path, "wb");
  if (!f) { fprintf(stderr, "* FATAL cannot create %s\n", path); return 1; }
  const int bad = len && fwrite(data, 1, len, f) != len;
  return (fclose(f) != 0 || bad) ? 1 : 0;
}

/* PINTRON_FAULT_INJECT=<rank>:<open|step|abort>  (tests: a rank that fails in that phase) */
static int fault_here(int rank, const char* phase) {
  const char* f = getenv("PINTRON_FAULT_INJECT");
  if (!f) return 0;
  char* end = NULL;
  const long r = strtol(f, &end, 10);
  return end && *end == ':' && r == rank && !strcmp(end + 1, phase);
}

/* ---- health agreement before the communicator ------------------------------------------------------ */
static void health_post(const char* id_path, int rank, int ok) {
  char p[1200];
  snprintf(p, sizeof p, "%s.%s.%d", id_path, ok ? "ok" : "fail", rank);
  const int fd = open(p, O_CREAT | O_WRONLY | O_TRUNC, 0600);
  if (fd >= 0) close(fd);
}
/* 0 when every rank posted "ok"; 1 when one posted "fail" or `seconds` passed */
static int health_wait(const char* id_path, int world, double seconds) {
  struct timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
  for (;;) {
    int n_ok = 0;
    for (int r = 0; r < world; ++r) {
      char p[1200];
      snprintf(p, sizeof p, "%s.fail.%d", id_path, r);
      if (access(p, F_OK) == 0) return 1;
      snprintf(p, sizeof p, "%s.ok.%d", id_path, r);
      if (access(p, F_OK) == 0) ++n_ok;
    }
    if (n_ok == world) return 0;
    struct timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
    if ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec) > seconds) return 1;
    nap_ms(5);
  }
}
static void health_cleanup(const char* id_path, int world) {
  for (int r = 0; r < world; ++r) {
    char p[1200];
    snprintf(p, sizeof p, "%s.ok.%d", id_path, r); unlink(p);
    snprintf(p, sizeof p, "%s.fail.%d", id_path, r); unlink(p);
  }
}

/* ---- the parent's watchdog: a child that ends badly ends the run ------------------------------------ */
static struct { pid_t* kids; int world; volatile int done; const char* id_path; } wd;
static void kill_kids(void) { for (int r = 1; r < wd.world; ++r) if (wd.kids[r] > 0) kill(wd.kids[r], SIGKILL); }
static void* watchdog_main(void* arg) {
  (void)arg;
  while (!wd.done) {
    for (int r = 1; r < wd.world; ++r) {
      if (wd.kids[r] <= 0) continue;
      int status = 0;
      const pid_t q = waitpid(wd.kids[r], &status, WNOHANG);
      if (q == 0) continue;
      wd.kids[r] = -1;                                  /* reaped */
      if (q > 0 && WIFEXITED(status) && WEXITSTATUS(status) == 0) continue;
      /* the peers may be inside a collective that will never complete: end everything here */
      fprintf(stderr, "* FATAL rank %d ended abnormally (status 0x%x): stopping the other ranks\n", r, status);
      health_post(wd.id_path, r, 0);                    /* for peers still before the communicator */
      nap_ms(300);                                      /* ranks that saw the failure themselves leave with their own message */
      kill_kids();
      for (int k = 1; k < wd.world; ++k) if (wd.kids[k] > 0) waitpid(wd.kids[k], NULL, 0);
      health_cleanup(wd.id_path, wd.world); unlink(wd.id_path);
      fflush(NULL);
      _exit(1);
    }
    nap_ms(20);
  }
  return NULL;
}

/* what a rank tells the others after its step: status, the byte sizes of its six files (size[0], the text of
 * raw-multifasta-out, is not sent: rank 0 prints it from the records) and of its packed factorization records */
typedef struct { uint64_t failed; uint64_t size[6]; uint64_t rec_size; } shard_note;

/* rank 0: the raw-multifasta-out text of every rank's records, printed side by side (one thread per part) */
typedef struct { const ef_seq* gen; const char* rec; size_t rl; const char* pests; size_t pl; ef_sink out; int rc; } print_job;
static void* print_main(void* arg) {
  print_job* j = (print_job*)arg;
  j->rc = ef_raw_text_from_records(j->gen, j->rec, j->rl, j->pests, j->pl, &j->out);
  return NULL;
}

/* this rank's part of a side file, in place (the file was created and sized by rank 0) */
static int write_part(const char* path, uint64_t off, const char* data, size_t len) {
  const int fd = open(path, O_WRONLY);
  if (fd < 0) { fprintf(stderr, "* FATAL cannot open %s\n", path); return 1; }
  size_t done = 0;
  while (done < len) {
    const ssize_t w = pwrite(fd, data + done, len - done, (off_t)(off + done));
    if (w <= 0) { close(fd); fprintf(stderr, "* FATAL short write to %s\n", path); return 1; }
    done += (size_t)w;
  }
  return close(fd) != 0;
}

/* one gene, this rank's share, gather to rank 0 */
static int run_shard(int argc, char** argv, int rank, int world, const char* id_path) {
  ef_shard_rank = rank; ef_shard_world = world;
  ef_session* s = fault_here(rank, "open") ? NULL : ef_session_open(argc, argv);
  pgpu_ctx* ctx = s ? ef_session_context(s) : NULL;
  pgpu_comm_id id;
  memset(&id, 0, sizeof id);
  int rc = s ? 0 : 1;
  if (rank == 0 && !rc) {
    /* the id goes to the other ranks through a file: written under a temporary name and renamed,
     * so a reader never sees half of it */
    if (pgpu_comm_unique_id(ctx, &id) != PGPU_OK) { fprintf(stderr, "* FATAL %s\n", pgpu_last_error(ctx)); rc = 1; }
    char tmp[1100];
    snprintf(tmp, sizeof tmp, "%s.tmp", id_path);
    if (!rc && (write_all(tmp, (const char*)&id, sizeof id) != 0 || rename(tmp, id_path) != 0)) rc = 1;
  }
  /* nobody enters the communicator unless everybody can */
  health_post(id_path, rank, rc == 0);
  /* (a rank that died is reported by the parent's watchdog at once; the limit only bounds a rank that lives and
   * never gets through its start-up: PINTRON_RANK_TIMEOUT_S, 900 s by default -- a very large ests.txt on a cold
   * file system is parsed in well under that) */
  const char* lim = getenv("PINTRON_RANK_TIMEOUT_S");
  if (health_wait(id_path, world, lim && atof(lim) > 0 ? atof(lim) : 900.0) != 0) {
    if (rc == 0) fprintf(stderr, "* FATAL rank %d: another rank could not start (GPU missing or not gfx950, input unreadable): giving up\n", rank);
    if (s) ef_session_close(s);
    return 1;
  }
  if (rank != 0) {
    FILE* f = fopen(id_path, "rb");
    if (!f || fread(&id, 1, sizeof id, f) != sizeof id) { fprintf(stderr, "* FATAL rank %d: no communicator id from rank 0\n", rank); rc = 1; }
    if (f) fclose(f);
    if (rc) { ef_session_close(s); return 1; }     /* (the id file exists once rank 0 posted "ok": this cannot be a lone failure) */
  }
  pgpu_comm* comm = NULL;
  if (pgpu_comm_init(ctx, rank, world, &id, &comm) != PGPU_OK) { fprintf(stderr, "* FATAL rank %d: %s\n", rank, pgpu_last_error(ctx)); ef_session_close(s); return 1; }
  ef_sched_stats st;
  memset(&st, 0, sizeof st);
  if (fault_here(rank, "abort")) abort();
  rc = fault_here(rank, "step") ? 1 : ef_session_step(s, &st);

  static const char* names[6] = { "raw-multifasta-out.txt", "processed-ests.txt", "megs.txt", "processed-megs.txt",
                                  "processed-megs-info.txt", "meg-edges.txt" };
  const char* dsw = getenv("PINTRON_SHARD_DIAGNOSTICS");
  const int diagnostics = !(dsw && dsw[0] == '0' && dsw[1] == '\0');
  char* text[6] = { NULL, NULL, NULL, NULL, NULL, NULL };
  char* records = NULL;
  shard_note mine;
  memset(&mine, 0, sizeof mine);
  mine.failed = rc != 0;
  for (int k = 1; k < 6 && rc == 0; ++k) {              /* (0, the text of raw-multifasta-out, stays where it is) */
    if (k >= 2 && !diagnostics) continue;
    size_t len = 0;
    text[k] = ef_session_output(s, k, &len);
    mine.size[k] = len;
  }
  if (rc == 0) { size_t len = 0; records = ef_session_output(s, 6, &len); mine.rec_size = len; }
  /* 1. status and sizes, to every rank */
  shard_note* notes = (shard_note*)calloc((size_t)world, sizeof(shard_note));
  int grc = pgpu_allgather(ctx, comm, &mine, sizeof mine, notes);
  if (grc != PGPU_OK) { fprintf(stderr, "* FATAL rank %d: status exchange: %s\n", rank, pgpu_last_error(ctx)); rc = 1; }
  int any_failed = rc != 0;
  for (int r = 0; r < world && grc == PGPU_OK; ++r) if (notes[r].failed) { any_failed = 1; if (rc == 0) fprintf(stderr, "* FATAL rank %d: rank %d failed, giving up\n", rank, r); }
  if (!any_failed) {
    /* 2. ONE gather: per rank [packed factorization records | processed-ests text] -- what north_star calls "the
     * per-EST factorization records" and the sequences they refer to.  raw-multifasta-out.txt is printed on rank 0
     * from them (ef_records.c): the text itself, eight times the bytes, does not travel. */
    const uint64_t pay = mine.rec_size + mine.size[1];
    char* send = (char*)malloc(pay + 1);
    memcpy(send, records, mine.rec_size); memcpy(send + mine.rec_size, text[1], mine.size[1]);
    uint64_t total = 0;
    for (int r = 0; r < world; ++r) total += notes[r].rec_size + notes[r].size[1];
    char* all = rank == 0 ? (char*)malloc(total + 1) : NULL;
    uint64_t* counts = (uint64_t*)calloc((size_t)world, sizeof(uint64_t));
    grc = pgpu_gather(ctx, comm, send, pay, all, rank == 0 ? total : 0, counts);
    if (grc != PGPU_OK) { fprintf(stderr, "* FATAL rank %d: gather: %s\n", rank, pgpu_last_error(ctx)); rc = 1; }
    if (getenv("PINTRON_VERBOSE"))
      fprintf(stderr, "* rank %d/%d: gather payload %llu bytes (records %llu + processed ESTs %llu); rank 0 receives %llu\n", rank, world,
              (unsigned long long)pay, (unsigned long long)mine.rec_size, (unsigned long long)mine.size[1], (unsigned long long)total);
    if (rank == 0 && grc == PGPU_OK) {
      print_job* pj = (print_job*)calloc((size_t)world, sizeof(print_job));
      pthread_t* pth = (pthread_t*)calloc((size_t)world, sizeof(pthread_t));
      char* started = (char*)calloc((size_t)world, 1);
      uint64_t at = 0;
      for (int r = 0; r < world; ++r) {
        pj[r].gen = (const ef_seq*)ef_session_genomic(s); pj[r].rec = all + at; pj[r].rl = (size_t)notes[r].rec_size;
        pj[r].pests = all + at + notes[r].rec_size; pj[r].pl = (size_t)notes[r].size[1];
        at += notes[r].rec_size + notes[r].size[1];
        started[r] = r + 1 < world && pthread_create(&pth[r], NULL, print_main, &pj[r]) == 0;
        if (!started[r]) print_main(&pj[r]);
      }
      for (int r = 0; r < world; ++r) if (started[r]) pthread_join(pth[r], NULL);
      FILE* f = fopen(names[0], "wb");
      if (!f) { fprintf(stderr, "* FATAL cannot create %s\n", names[0]); rc = 1; }
      for (int r = 0; r < world && f; ++r) {
        if (pj[r].rc != 0) { fprintf(stderr, "* FATAL the records of rank %d do not fit its processed ESTs\n", r); rc = 1; }
        else if (pj[r].out.len && fwrite(pj[r].out.mem, 1, pj[r].out.len, f) != pj[r].out.len) rc = 1;
      }
      if (f && fclose(f) != 0) rc = 1;
      for (int r = 0; r < world; ++r) free(pj[r].out.mem);
      f = rc == 0 ? fopen(names[1], "wb") : NULL;
      if (rc == 0 && !f) { fprintf(stderr, "* FATAL cannot create %s\n", names[1]); rc = 1; }
      for (int r = 0; r < world && f; ++r)
        if (pj[r].pl && fwrite(pj[r].pests, 1, pj[r].pl, f) != pj[r].pl) rc = 1;
      if (f && fclose(f) != 0) rc = 1;
      free(pj); free(pth); free(started);
    }
    free(counts); free(all); free(send);
    /* 3. the side files, in place: rank 0 creates them at their final size, then every rank writes its part */
    if (diagnostics && rc == 0) {
      if (rank == 0) {
        for (int k = 2; k < 6; ++k) {
          uint64_t tot = 0;
          for (int r = 0; r < world; ++r) tot += notes[r].size[k];
          const int fd = open(names[k], O_CREAT | O_WRONLY | O_TRUNC, 0644);
          if (fd < 0 || ftruncate(fd, (off_t)tot) != 0) { fprintf(stderr, "* FATAL cannot create %s\n", names[k]); rc = 1; }
          if (fd >= 0) close(fd);
        }
      }
      /* "created" has to be known to all before anybody writes: one more (tiny) exchange, which also tells
       * everybody whether rank 0 managed */
      shard_note ready; memset(&ready, 0, sizeof ready); ready.failed = rc != 0;
      shard_note* ready_all = (shard_note*)calloc((size_t)world, sizeof(shard_note));
      if (pgpu_allgather(ctx, comm, &ready, sizeof ready, ready_all) != PGPU_OK) { fprintf(stderr, "* FATAL rank %d: %s\n", rank, pgpu_last_error(ctx)); rc = 1; }
      else if (ready_all[0].failed) rc = 1;
      free(ready_all);
      for (int k = 2; k < 6 && rc == 0; ++k) {
        uint64_t off = 0;
        for (int r = 0; r < rank; ++r) off += notes[r].size[k];
        if (mine.size[k] && write_part(names[k], off, text[k], (size_t)mine.size[k]) != 0) rc = 1;
      }
      /* (the parent returns only after every rank has ended, so the files are complete when est-fact is) */
    }
  }
  free(notes);
  if (getenv("PINTRON_VERBOSE"))
    fprintf(stderr, "* rank %d/%d: %zu ESTs (%zu aligned), %zu DP jobs\n", rank, world, st.units, st.aligned, st.dp_jobs);
  for (int k = 0; k < 6; ++k) free(text[k]);
  free(records);
  pgpu_comm_destroy(ctx, comm);
  ef_session_close(s);
  return (rc || any_failed) ? 1 : 0;
}

/* many genes, no exchange: gene g on rank g mod world */
static int run_genes(int argc, char** argv, const char* list_path, int rank, int world) {
  FILE* f = fopen(list_path, "r");
  if (!f) { fprintf(stderr, "* FATAL cannot read the gene list %s\n", list_path); return 1; }
  char line[4096];
  int g = 0, rc = 0;
  char start_dir[4096];
  if (!getcwd(start_dir, sizeof start_dir)) { fclose(f); return 1; }
  ef_leave_without_cleanup = 0;            /* several sessions in one process: each is taken apart */
  while (fgets(line, sizeof line, f)) {
    size_t n = strlen(line);
    while (n && (line[n - 1] == '\n' || line[n - 1] == '\r' || line[n - 1] == ' ')) line[--n] = '\0';
    if (n == 0 || line[0] == '#') continue;
    const int mine = g % world == rank;
    ++g;
    if (!mine) continue;
    if (chdir(start_dir) != 0 || chdir(line) != 0) { fprintf(stderr, "* FATAL cannot enter %s\n", line); rc = 1; continue; }
    if (ef_run_batched(argc, argv) != 0) { fprintf(stderr, "* FATAL est-fact failed in %s\n", line); rc = 1; }
  }
  fclose(f);
  return rc;
}

/* GPUs this process may use, counted WITHOUT touching the GPU runtime (the parent must not initialise it
 * before it starts the other ranks): the entries of a visibility mask, else the AMD render nodes.
 * -1: unknown. */
static int visible_gpu_count(void) {
  static const char* masks[] = { "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES" };
  for (int k = 0; k < 3; ++k) {
    const char* m = getenv(masks[k]);
    if (!m) continue;
    if (!m[0]) return -1;                 /* an empty mask: let the ranks find out (and agree to stop) */
    int n = 1;
    for (const char* c = m; *c; ++c) if (*c == ',') ++n;
    return n;
  }
  DIR* d = opendir("/dev/dri");
  if (!d) return -1;
  int n = 0;
  for (struct dirent* e; (e = readdir(d)) != NULL;) {
    if (strncmp(e->d_name, "renderD", 7) != 0) continue;
    char p[300], v[32] = { 0 };
    snprintf(p, sizeof p, "/sys/class/drm/%s/device/vendor", e->d_name);
    FILE* f = fopen(p, "r");
    if (!f) continue;
    if (fgets(v, sizeof v, f) && strtol(v, NULL, 16) == 0x1002) ++n;
    fclose(f);
  }
  closedir(d);
  return n > 0 ? n : -1;
}

int ef_main_multi(int argc, char** argv) {
  /* our two options are taken out of the argument list; the rest is est-fact's own */
  int world = getenv("PINTRON_GPUS") ? atoi(getenv("PINTRON_GPUS")) : 1;
  const char* genes = getenv("PINTRON_GENES");
  char** av = (char**)malloc((size_t)(argc + 1) * sizeof(char*));
  int ac = 0;
  for (int i = 0; i < argc; ++i) {
    if (!strncmp(argv[i], "--gpus=", 7)) world = atoi(argv[i] + 7);
    else if (!strncmp(argv[i], "--genes=", 8)) genes = argv[i] + 8;
    else av[ac++] = argv[i];
  }
  av[ac] = NULL;
  if (world < 1 || world > 64) { fprintf(stderr, "est-fact: invalid argument for option 'gpus'\n"); return 2; }
  int rank = 0;
  pid_t* kids = NULL;
  pthread_t wd_thread;
  int wd_started = 0;
  char id_path[1024];
  if (getenv("PINTRON_RANK")) {                       /* started by the parent below */
    rank = atoi(getenv("PINTRON_RANK"));
    snprintf(id_path, sizeof id_path, "%s", getenv("PINTRON_COMM_FILE") ? getenv("PINTRON_COMM_FILE") : ".pintron-comm-id");
    prctl(PR_SET_PDEATHSIG, SIGKILL);                 /* no orphans inside a collective */
    /* the parent was gone before the request took effect: it exported its own pid before it started us (a ppid
     * of 1 says nothing -- est-fact may BE pid 1 of a container, and an orphan under a subreaper gets another) */
    { const char* pp = getenv("PINTRON_PARENT_PID"); if (pp && atol(pp) > 0 && (long)getppid() != atol(pp)) return 1; }
  } else {
    snprintf(id_path, sizeof id_path, "%s/.pintron-comm-id-%ld", getenv("TMPDIR") ? getenv("TMPDIR") : "/tmp", (long)getpid());
    unlink(id_path);
    if (world > 1) {
      const int have = visible_gpu_count();
      if (have >= 0 && world > have) {
        fprintf(stderr, "* FATAL --gpus=%d but only %d GPU%s visible\n", world, have, have == 1 ? " is" : "s are");
        free(av);
        return 1;
      }
      health_cleanup(id_path, world);
      char exe[4096];
      const ssize_t n = readlink("/proc/self/exe", exe, sizeof exe - 1);
      if (n <= 0) { fprintf(stderr, "* FATAL cannot find the est-fact executable\n"); return 1; }
      exe[n] = '\0';
      char wv[32];
      snprintf(wv, sizeof wv, "%d", world);
      setenv("PINTRON_GPUS", wv, 1);
      setenv("PINTRON_COMM_FILE", id_path, 1);
      setenv("LOCAL_WORLD_SIZE", wv, 1);              /* the ranks share the host's cores (host_core_share) */
      { char pv[32]; snprintf(pv, sizeof pv, "%ld", (long)getpid()); setenv("PINTRON_PARENT_PID", pv, 1); }
      kids = (pid_t*)calloc((size_t)world, sizeof(pid_t));
      for (int r = 1; r < world; ++r) {
        char rv[32];
        snprintf(rv, sizeof rv, "%d", r);
        setenv("PINTRON_RANK", rv, 1);
        if (posix_spawn(&kids[r], exe, NULL, NULL, argv, environ) != 0) {
          fprintf(stderr, "* FATAL cannot start rank %d: %s\n", r, strerror(errno));
          kids[r] = 0;
          for (int k = 1; k < r; ++k) { kill(kids[k], SIGKILL); waitpid(kids[k], NULL, 0); }    /* none is left behind */
          free(kids); free(av);
          return 1;
        }
      }
      setenv("PINTRON_RANK", "0", 1);
      wd.kids = kids; wd.world = world; wd.done = 0; wd.id_path = id_path;
      wd_started = pthread_create(&wd_thread, NULL, watchdog_main, NULL) == 0;
    }
  }
  if (world > 1) {                                     /* rank r uses GPU r */
    char dv[32];
    snprintf(dv, sizeof dv, "%d", rank);
    if (!getenv("PINTRON_GPU_DEVICE") || kids || getenv("PINTRON_RANK")) setenv("PINTRON_GPU_DEVICE", dv, 1);
  }
  int rc;
  if (genes) rc = run_genes(ac, av, genes, rank, world);
  else if (world > 1) rc = run_shard(ac, av, rank, world, id_path);
  else rc = ef_run_batched(ac, av);
  if (kids) {
    if (rc != 0) {
      /* rank 0 failed: the others either saw it in an exchange and are leaving, or wait for rank 0 in
       * one -- give them a moment, then end them */
      nap_ms(200);
    }
    wd.done = 1;
    if (wd_started) pthread_join(wd_thread, NULL);
    for (int r = 1; r < world; ++r) {
      if (kids[r] <= 0) continue;                      /* reaped by the watchdog: it ended well */
      if (rc != 0) kill(kids[r], SIGKILL);
      int status = 0;
      if (waitpid(kids[r], &status, 0) < 0 || !WIFEXITED(status) || WEXITSTATUS(status) != 0) { if (rc == 0) fprintf(stderr, "* FATAL rank %d failed\n", r); rc = rc ? rc : 1; }
    }
    health_cleanup(id_path, world);
    unlink(id_path);
    free(kids);
  }
  free(av);
  return rc;
}
