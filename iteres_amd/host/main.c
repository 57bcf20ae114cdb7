/* main.c — command dispatch of the drop-in `iteres` (iteres.c:3-29 of the reference). stat and filter are the
 * alignment commands rebuilt on the MI355X engine; cpgstat / cpgfilter (bedGraph input) use the engine's lookup and add
 * their floating-point sums up on the host in file order (cmd_cpg.c). */
#include "itx_host.h"

#include <omp.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

static int usage(void)
{
    fprintf(stderr, "\n");
    fprintf(stderr, "Program: iteres (repeat analysis utils from Wang lab)\n");
    fprintf(stderr, "Version: %s\n\n", ITERES_VERSION);
    fprintf(stderr, "Usage:   iteres <command> [options]\n\n");
    fprintf(stderr, "Command: stat        get repeat alignment statistics\n");
    fprintf(stderr, "         filter      filter alignment statistic on repName/repFamily/repClass\n");
    fprintf(stderr, "         cpgstat     generate CpG density from MRE-Seq data for repeats\n");
    fprintf(stderr, "         cpgfilter   filter CpG statistic on repName/repFamily/repClass\n");
    fprintf(stderr, "\n");
    return 1;
}

int main(int argc, char *argv[])
{
    if (argc < 2) return usage();
    numa_place();
    struct timespec ts_main0;
    clock_gettime(CLOCK_MONOTONIC, &ts_main0);
    /* The HIP runtime multiplexes its streams onto 4 hardware queues by default, and kernels that share a hardware queue run
     * one after the other. The decoder computes on four lanes of its own next to the engine's stream, and eight copy streams feed
     * them: with enough queues their kernels really do overlap (measured in round 2 with 8: scan of the 500 M-read BAM 1.95 s
     * instead of 2.3 s). Read by the runtime when it starts, so it has to be in the environment before the first HIP call; a
     * value the user set stays. */
    setenv("GPU_MAX_HW_QUEUES", "12", 0);
    multi_early(argc, argv);
    /* host threads (BGZF inflate, record parse, bigWig deflate): OMP_NUM_THREADS when given, else the processors this
     * process may run on, capped — the decode saturates long before a big host's core count and idle OpenMP workers
     * spinning on a shared box cost more than they give */
    if (!getenv("OMP_NUM_THREADS")) {
        int n = omp_get_num_procs();
        if (n > 16) n = 16;
        omp_set_num_threads(n > 0 ? n : 1);
    }
    int rc;
    if (strcmp(argv[1], "stat") == 0) rc = main_stat(argc - 1, argv + 1);
    else if (strcmp(argv[1], "filter") == 0) rc = main_filter(argc - 1, argv + 1);
    else if (strcmp(argv[1], "cpgstat") == 0) rc = main_cpgstat(argc - 1, argv + 1);
    else if (strcmp(argv[1], "cpgfilter") == 0) rc = main_cpgfilter(argc - 1, argv + 1);
    else {
        fprintf(stderr, "[iteres] unrecognized command '%s'\n", argv[1]);
        return 1;
    }
    multi_finish();
    /* Everything is written. The HIP runtime's own exit handlers would now unmap tens of gigabytes of device buffers and
     * page-locked chunks one by one (measured: about a second for a 500 M-read run); the process is going away anyway —
     * flush what stdio holds and leave. (stat / filter only: the CpG commands hold next to nothing on the device.) */
    const char *pre = getenv("LD_PRELOAD");                        /* a profiler rides along (rocprofv3): it writes its files from exit handlers */
    if ((strcmp(argv[1], "stat") == 0 || strcmp(argv[1], "filter") == 0) && !(pre && *pre) && !getenv("ITX_NO_FAST_EXIT")) {
        if (getenv("ITX_TIMING")) {
            struct timespec ts1;
            clock_gettime(CLOCK_MONOTONIC, &ts1);
            itx_timing_report();
            fprintf(stderr, "[itx timing] main() entered %.3f s ago\n", (double)(ts1.tv_sec - ts_main0.tv_sec) + 1e-9 * (double)(ts1.tv_nsec - ts_main0.tv_nsec));
        }
        fflush(NULL);
        _exit(rc);
    }
    return rc;
}
