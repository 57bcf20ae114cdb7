/* shares.c — the shares of a multi-GPU job (the stream of generic.c:725-745 partitioned by record).
 * A file of its own so that the arithmetic can be exercised without a GPU (test/share_dump.c, tests/test_distributed_gloo.py). */
#include "itx_host.h"

#include <sys/stat.h>

/* ---- the shares of a multi-GPU job: the compressed bytes of all alignment files, laid end to end, are cut into
 * multi_world() equal ranges; a rank takes, of every file, what falls into its range (aln_open_range turns the byte
 * offsets into record boundaries). One rank: everything. */
/* sh[fi] for this rank; returns 0 when the job cannot be shared (then rank 0 has everything, the others nothing) */
int plan_shares(char **files, int n_files, int splittable, int rank, int world, size_t min_share, share_t *sh)
{
    for (int i = 0; i < n_files; i++) {
        sh[i].lo = 0;
        sh[i].hi = rank == 0 ? SIZE_MAX : 0;
    }
    if (world <= 1) return 1;
    if (!splittable) return 0;
    size_t size[100], total = 0;
    for (int i = 0; i < n_files; i++) {
        struct stat sb;
        if (stat(files[i], &sb) != 0 || !S_ISREG(sb.st_mode)) return 0;       /* a pipe, or a file that is not there (reported where the reference does) */
        size[i] = (size_t)sb.st_size;
        total += size[i];
    }
    if (total / (size_t)world < min_share) return 0;
    const size_t g0 = (size_t)((__uint128_t)total * (unsigned)rank / (unsigned)world), g1 = (size_t)((__uint128_t)total * ((unsigned)rank + 1) / (unsigned)world);
    size_t base = 0;
    for (int i = 0; i < n_files; i++) {
        const size_t a = g0 > base ? g0 - base : 0, b = g1 > base ? g1 - base : 0;
        sh[i].lo = a < size[i] ? a : size[i];
        sh[i].hi = b < size[i] ? b : SIZE_MAX;
        if (sh[i].hi != SIZE_MAX && sh[i].hi <= sh[i].lo) sh[i].lo = sh[i].hi = 0;
        if (sh[i].lo >= size[i]) sh[i].lo = sh[i].hi = 0;
        base += size[i];
    }
    return 1;
}

