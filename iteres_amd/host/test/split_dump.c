/* split_dump.c — CPU-only check tool (tests/test_host_reader.py): where would a share of this BAM begin for each of the
 * given compressed byte offsets? One line per offset: at found block off csize.
 * usage: split_dump <file.bam> <at> [<at> ...] */
#include "../itx_host.h"

#include <stdlib.h>

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    for (int i = 2; i < argc; i++) {
        size_t b = 0, o = 0, cs = 0;
        const size_t at = (size_t)strtoull(argv[i], NULL, 0);
        const int f = aln_find_split(argv[1], at, &b, &o, &cs);
        printf("%zu %d %zu %zu %zu\n", at, f, b, o, cs);
    }
    return 0;
}
