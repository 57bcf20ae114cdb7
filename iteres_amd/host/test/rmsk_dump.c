/* rmsk_dump.c — test tool: what the host makes of chrom.sizes / rep.sizes / rmsk.txt (tables.c), as text, so the
 * parallel parse can be compared with an independent model on a machine without a GPU.
 *   rmsk_dump <chrom.sizes> <rep.sizes> <rmsk.txt> [filter_field filter_name] */
#define _GNU_SOURCE
#include "../itx_host.h"

#include <stdlib.h>

int main(int argc, char **argv)
{
    if (argc != 4 && argc != 6) return 2;
    sizes_t cs, rs;
    rmsk_t rm;
    sizes_load(argv[1], &cs);
    sizes_load(argv[2], &rs);
    rmsk_load(argv[3], &cs, &rs, argc == 6 ? atoi(argv[4]) : 0, argc == 6 ? argv[5] : "ALL", &rm);
    printf("seen\t%d\n", rm.repeat_num);
    for (uint32_t i = 0; i < rm.chroms.n; i++) printf("chrom\t%s\t%lld\n", rm.chroms.name[i], (long long)rm.chrom_size[i]);
    for (uint32_t i = 0; i < rm.reps.n; i++)
        printf("rep\t%s\t%u\t%s\t%s\t%llu\t%llu\n", rm.reps.name[i], rm.rep_len[i], rm.fams.name[rm.rep_fam[i]], rm.clas.name[rm.rep_cla[i]],
               (unsigned long long)rm.rep_genome[i], (unsigned long long)rm.rep_total[i]);
    for (uint32_t i = 0; i < rm.fams.n; i++)
        printf("fam\t%s\t%s\t%llu\t%llu\n", rm.fams.name[i], rm.clas.name[rm.fam_cla[i]], (unsigned long long)rm.fam_genome[i],
               (unsigned long long)rm.fam_total[i]);
    for (uint32_t i = 0; i < rm.clas.n; i++)
        printf("cla\t%s\t%llu\t%llu\n", rm.clas.name[i], (unsigned long long)rm.cla_genome[i], (unsigned long long)rm.cla_total[i]);
    for (size_t i = 0; i < rm.n_rows; i++)
        printf("row\t%s\t%d\t%u\t%u\t%u\t%u\t%s\t%s\t%s\n", rm.chroms.name[rm.row_chrom_name[i]], rm.rows[i].chrom, rm.rows[i].start, rm.rows[i].end,
               rm.rows[i].cons_start, rm.rows[i].cons_end, rm.reps.name[rm.rows[i].rep], rm.fams.name[rm.rows[i].fam], rm.clas.name[rm.rows[i].cla]);
    return 0;
}
