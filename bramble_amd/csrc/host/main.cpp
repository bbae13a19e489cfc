// bramble command line: see br_cli_main (include/bramble_amd.h)
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <unistd.h>
extern "C" int br_cli_main(int argc, char **argv);
extern "C" void br_cli_exit_at_end(int on);
// The output file is closed and renamed inside br_cli_main; what is left at this point is the HIP runtime's exit handlers
// walking tens of gigabytes of device and pinned allocations.  The process image goes away either way: leave at once.
static double epoch() { struct timespec t; clock_gettime(CLOCK_REALTIME, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
int main(int argc, char **argv) {
  const bool timing = getenv("BRAMBLE_AMD_TIMING") != nullptr;   // process start-up and teardown lie outside br_cli_main's own clock
  if (timing) fprintf(stderr, "[bramble] main entered at %.3f\n", epoch());
  br_cli_exit_at_end(1);
  const int rc = br_cli_main(argc, argv);
  if (timing) fprintf(stderr, "[bramble] main leaving at %.3f\n", epoch());
  fflush(stdout);
  fflush(stderr);
  if (getenv("BRAMBLE_AMD_CLI_CLEANUP")) return rc;   // everything was released inside: exit handlers (a profiler's among them) run
  _exit(rc);
}
