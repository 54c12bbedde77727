/* est-fact for MI355X: drop-in replacement of PIntron's est-fact stage (src/main-est-fact.c).
 * Reads genomic.txt / ests.txt (and config.ini) in the current directory, writes
 * raw-multifasta-out.txt, processed-ests.txt and the MEG side files.  All pairings and dynamic
 * programs run on the GPU through libpintron_gpu.so; without a gfx950 device the program fails. */
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

#include "estfact.h"
#include "ef_gpu.h"
#include "ef_sched.h"

int main(int argc, char** argv) {
  const char* mode = getenv("PINTRON_ESTFACT_MODE");
  if (mode && !strcmp(mode, "direct")) return ef_run(argc, argv, ef_gpu_open, ef_gpu_close);
  /* PINTRON_CLEAN_EXIT keeps the orderly teardown (profilers that flush at exit need it) */
  if (getenv("PINTRON_CLEAN_EXIT")) return ef_main_multi(argc, argv);
  ef_leave_without_cleanup = 1;
  const int rc = ef_main_multi(argc, argv);      /* --gpus / --genes, else the plain batched run */
  fflush(NULL);
  _exit(rc);
}
