/* rmsk_time.c — development aid: wall time of the rmsk parse (tables.c) alone.  rmsk_time <chrom.sizes> <rep.sizes> <rmsk.txt> */
#define _GNU_SOURCE
#include "../itx_host.h"
#include <time.h>
int main(int argc, char **argv)
{
    if (argc != 4) return 2;
    struct timespec a, b;
    sizes_t cs, rs;
    rmsk_t rm;
    clock_gettime(CLOCK_MONOTONIC, &a);
    sizes_load(argv[1], &cs);
    sizes_load(argv[2], &rs);
    rmsk_load(argv[3], &cs, &rs, 0, "ALL", &rm);
    clock_gettime(CLOCK_MONOTONIC, &b);
    printf("rows %zu names %u fams %u clas %u: %.3f s\n", rm.n_rows, rm.reps.n, rm.fams.n, rm.clas.n,
           (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec));
    return 0;
}
