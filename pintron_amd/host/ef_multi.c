/* est-fact on several GPUs of one node, driven by the C program itself.
 *
 *   est-fact --gpus=N            one gene (cwd): N processes, one per GPU; rank r factorizes a contiguous
 *                                range of the ESTs (ef_load_ests), the text of the six output files goes
 *                                to rank 0 over pgpu_gather (RCCL, xGMI) and is written there in rank
 *                                order = input order, so the files are those of a single process
 *   est-fact --genes=FILE        FILE lists directories (one per line), each holding genomic.txt and
 *                                ests.txt: gene g runs on rank g mod N and leaves its files in its own
 *                                directory; no exchange (C4 = 8 genes on 8 GPUs)
 * PINTRON_GPUS / PINTRON_GENES set the same from the environment, which is how an unmodified
 * pipeline driver (dist-scripts/pintron.py:878-884 calls est-fact without options) is pointed at
 * several GPUs; INTEGRATION.md section 7 shows the two-line driver patch.
 *
 * The parent starts the other ranks as fresh processes BEFORE anything touches the GPU and then
 * becomes rank 0 itself. */
#define _GNU_SOURCE
#include <errno.h>
#include <spawn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include "estfact.h"
#include "ef_gpu.h"
#include "ef_sched.h"

extern char** environ;

static int write_all(const char* path, const char* data, size_t len) {
  FILE* f = fopen(path, "wb");
  if (!f) { fprintf(stderr, "* FATAL cannot create %s\n", path); return 1; }
  const int bad = len && fwrite(data, 1, len, f) != len;
  return (fclose(f) != 0 || bad) ? 1 : 0;
}

/* one gene, this rank's share, gather to rank 0 */
static int run_shard(int argc, char** argv, int rank, int world, const char* id_path) {
  ef_shard_rank = rank; ef_shard_world = world;
  ef_session* s = ef_session_open(argc, argv);
  if (!s) return 1;
  pgpu_ctx* ctx = ef_session_context(s);
  pgpu_comm_id id;
  memset(&id, 0, sizeof id);
  int rc = 0;
  if (rank == 0) {
    /* the id goes to the other ranks through a file: written under a temporary name and renamed,
     * so a reader never sees half of it */
    if (pgpu_comm_unique_id(ctx, &id) != PGPU_OK) { fprintf(stderr, "* FATAL %s\n", pgpu_last_error(ctx)); rc = 1; }
    char tmp[1100];
    snprintf(tmp, sizeof tmp, "%s.tmp", id_path);
    if (!rc && (write_all(tmp, (const char*)&id, sizeof id) != 0 || rename(tmp, id_path) != 0)) rc = 1;
  } else {
    FILE* f = NULL;
    for (int tries = 0; tries < 6000 && !f; ++tries) {      /* up to a minute */
      f = fopen(id_path, "rb");
      if (!f) { struct timespec ts = { 0, 10 * 1000 * 1000 }; nanosleep(&ts, NULL); }
    }
    if (!f || fread(&id, 1, sizeof id, f) != sizeof id) { fprintf(stderr, "* FATAL rank %d: no communicator id from rank 0\n", rank); rc = 1; }
    if (f) fclose(f);
  }
  pgpu_comm* comm = NULL;
  if (!rc && pgpu_comm_init(ctx, rank, world, &id, &comm) != PGPU_OK) { fprintf(stderr, "* FATAL rank %d: %s\n", rank, pgpu_last_error(ctx)); rc = 1; }
  if (rc) { ef_session_close(s); return rc; }
  ef_sched_stats st;
  rc = ef_session_step(s, &st);
  /* a rank that failed still takes part in the gathers (with nothing), so nobody waits for ever */
  static const char* names[6] = { "raw-multifasta-out.txt", "processed-ests.txt", "megs.txt", "processed-megs.txt",
                                  "processed-megs-info.txt", "meg-edges.txt" };
  uint64_t* counts = (uint64_t*)calloc((size_t)world, sizeof(uint64_t));
  uint64_t* sizes = (uint64_t*)calloc((size_t)world, sizeof(uint64_t));
  for (int k = 0; k < 6; ++k) {
    size_t len = 0;
    char* text = rc == 0 ? ef_session_output(s, k, &len) : NULL;
    /* two rounds: the lengths (8 bytes per rank), so that rank 0 can size its buffer exactly, then
     * the text */
    const uint64_t mine = len;
    int grc = pgpu_gather(ctx, comm, &mine, sizeof mine, sizes, (uint64_t)world * sizeof(uint64_t), counts);
    char* all = NULL;
    uint64_t total = 0;
    if (grc == PGPU_OK && rank == 0) {
      for (int r = 0; r < world; ++r) total += sizes[r];
      all = (char*)malloc(total + 1);
    }
    if (grc == PGPU_OK) grc = pgpu_gather(ctx, comm, text, mine, all, total, counts);
    if (grc != PGPU_OK) { fprintf(stderr, "* FATAL rank %d: gather of %s: %s\n", rank, names[k], pgpu_last_error(ctx)); rc = 1; }
    if (rank == 0 && grc == PGPU_OK && write_all(names[k], all, (size_t)total) != 0) rc = 1;
    free(all); free(text);
    if (grc != PGPU_OK) break;
  }
  free(sizes);
  free(counts);
  if (getenv("PINTRON_VERBOSE"))
    fprintf(stderr, "* rank %d/%d: %zu ESTs (%zu aligned), %zu DP jobs\n", rank, world, st.units, st.aligned, st.dp_jobs);
  pgpu_comm_destroy(ctx, comm);
  ef_session_close(s);
  return rc;
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
  char id_path[1024];
  if (getenv("PINTRON_RANK")) {                       /* started by the parent below */
    rank = atoi(getenv("PINTRON_RANK"));
    snprintf(id_path, sizeof id_path, "%s", getenv("PINTRON_COMM_FILE") ? getenv("PINTRON_COMM_FILE") : ".pintron-comm-id");
  } else {
    snprintf(id_path, sizeof id_path, "%s/.pintron-comm-id-%ld", getenv("TMPDIR") ? getenv("TMPDIR") : "/tmp", (long)getpid());
    unlink(id_path);
    if (world > 1) {
      char exe[4096];
      const ssize_t n = readlink("/proc/self/exe", exe, sizeof exe - 1);
      if (n <= 0) { fprintf(stderr, "* FATAL cannot find the est-fact executable\n"); return 1; }
      exe[n] = '\0';
      char wv[32];
      snprintf(wv, sizeof wv, "%d", world);
      setenv("PINTRON_GPUS", wv, 1);
      setenv("PINTRON_COMM_FILE", id_path, 1);
      setenv("LOCAL_WORLD_SIZE", wv, 1);              /* the ranks share the host's cores (host_core_share) */
      kids = (pid_t*)calloc((size_t)world, sizeof(pid_t));
      for (int r = 1; r < world; ++r) {
        char rv[32];
        snprintf(rv, sizeof rv, "%d", r);
        setenv("PINTRON_RANK", rv, 1);
        if (posix_spawn(&kids[r], exe, NULL, NULL, argv, environ) != 0) { fprintf(stderr, "* FATAL cannot start rank %d: %s\n", r, strerror(errno)); return 1; }
      }
      setenv("PINTRON_RANK", "0", 1);
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
    for (int r = 1; r < world; ++r) {
      int status = 0;
      if (waitpid(kids[r], &status, 0) < 0 || !WIFEXITED(status) || WEXITSTATUS(status) != 0) { fprintf(stderr, "* FATAL rank %d failed\n", r); rc = rc ? rc : 1; }
    }
    unlink(id_path);
    free(kids);
  }
  free(av);
  return rc;
}
