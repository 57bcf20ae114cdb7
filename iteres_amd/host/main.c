/* main.c — command dispatch of the drop-in `iteres` (iteres.c:3-29 of the reference). stat and filter are the
 * alignment commands rebuilt on the MI355X engine; cpgstat / cpgfilter (bedGraph input) use the engine's lookup and add
 * their floating-point sums up on the host in file order (cmd_cpg.c). */
#include "itx_host.h"

#include <omp.h>
#include <stdlib.h>
#include <string.h>
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
    return rc;
}
