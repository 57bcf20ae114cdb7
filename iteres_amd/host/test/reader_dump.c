/* reader_dump.c — CPU-only check tool for the host decoder (tests/test_host_reader.py): decodes an alignment file
 * with the product's aln_reader into plain malloc'd staging arrays and prints one line per record:
 *   tid pos tmpend mapq flag5 mpos isize qname        (plus a header line per reference sequence)
 * usage: reader_dump <file> <is_sam 0|1> [batch=4096] */
#include "../itx_host.h"

#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    const int is_sam = atoi(argv[2]);
    const size_t cap = argc > 3 ? (size_t)atoll(argv[3]) : 4096;
    aln_reader *r = aln_open(argv[1], is_sam);
    if (!r) {
        fprintf(stderr, "open failed\n");
        return 1;
    }
    for (int t = 0; t < aln_n_targets(r); t++) printf("@%d\t%s\n", t, aln_target_name(r, t));
    itx_staging st;
    memset(&st, 0, sizeof st);
    st.tid = malloc(cap * 4); st.pos = malloc(cap * 4); st.tmpend = malloc(cap * 4); st.mapq = malloc(cap); st.flag5 = malloc(cap);
    st.mpos = malloc(cap * 4); st.isize = malloc(cap * 4); st.hit_row = malloc(cap * 4); st.capacity = cap;
    char **qn = calloc(cap, sizeof(char *));
    aln_side side = {.want_qnames = 1, .qname = qn};
    int any_paired = 0, xa = 0;
    size_t n, total = 0;
    while ((n = aln_read_batch(r, &st, cap, &side, &any_paired, &xa)) > 0) {
        for (size_t i = 0; i < n; i++) {
            printf("%d\t%d\t%d\t%u\t%u\t%d\t%d\t%s\n", st.tid[i], st.pos[i], st.tmpend[i], st.mapq[i], st.flag5[i], st.mpos[i], st.isize[i], qn[i]);
            free(qn[i]);
            qn[i] = NULL;
        }
        total += n;
    }
    printf("#records=%zu paired=%d xa=%d\n", total, any_paired, xa);
    aln_close(r);
    return 0;
}
