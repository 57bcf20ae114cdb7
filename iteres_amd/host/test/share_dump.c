/* share_dump.c — CPU-only check tool (tests/test_distributed_gloo.py): what every rank of a job of <world> ranks would take
 * of a list of BAM files — the product's own share planning (shares.c: plan_shares) and split points (bamio.c: find_split).
 * One line per (rank, file):   rank file lo hi | lo_found lo_block lo_off | hi_found hi_block hi_off
 * (found: 1 a record start was found at/after the offset, 0 none up to the end of the file, -1 given up; lo = 0 and
 * hi = SIZE_MAX need no search: found = 2). A first line "shared 0|1" says whether the job can be shared at all.
 * usage: share_dump <world> <min_share_bytes> <file.bam>[,<file.bam>...] */
#include "../itx_host.h"

#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    const int world = atoi(argv[1]);
    const size_t min_share = (size_t)strtoull(argv[2], NULL, 0);
    char *files[100];
    int n_files = 0;
    for (char *s = argv[3]; n_files < 100;) {
        files[n_files++] = s;
        char *c = strchr(s, ',');
        if (!c) break;
        *c = 0;
        s = c + 1;
    }
    for (int rank = 0; rank < world; rank++) {
        share_t sh[100];
        const int shared = plan_shares(files, n_files, 1, rank, world, min_share, sh);
        if (rank == 0) printf("shared %d\n", shared);
        for (int fi = 0; fi < n_files; fi++) {
            size_t b0 = 0, o0 = 0, b1 = 0, o1 = 0, cs = 0;
            int f0 = 2, f1 = 2;
            if (sh[fi].lo != sh[fi].hi) {
                if (sh[fi].lo > 0) f0 = aln_find_split(files[fi], sh[fi].lo, &b0, &o0, &cs);
                if (sh[fi].hi != SIZE_MAX) f1 = aln_find_split(files[fi], sh[fi].hi, &b1, &o1, &cs);
            }
            printf("%d %d %zu %zu | %d %zu %zu | %d %zu %zu\n", rank, fi, sh[fi].lo, sh[fi].hi, f0, b0, o0, f1, b1, o1);
        }
    }
    return 0;
}
