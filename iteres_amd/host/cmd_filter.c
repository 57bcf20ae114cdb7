/* cmd_filter.c — `iteres filter`: same options, banners, output names and exit codes as filter.c:30-161 of the
 * reference; per-locus counting runs on the GPU engine (stream.c). */
#define _GNU_SOURCE
#include "itx_host.h"

#include <getopt.h>
#include <libgen.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static int filter_usage(void)
{
    fprintf(stderr, "\n");
    fprintf(stderr, "Obtain alignment statistics of individual loci of each repeat subfamily, family or class.\n\n");
    fprintf(stderr, "Usage:   iteres filter [options] <chromosome size file> <repeat size file> <rmsk.txt> <bam/sam alignment file>\n\n");
    fprintf(stderr, "Options: -S       input is SAM [off]\n");
    fprintf(stderr, "         -Q       mapping Quality threshold [10]\n");
    fprintf(stderr, "         -g       coverage threshold for overlapping [0.0001]\n");
    fprintf(stderr, "         -N       normalized by number of (0: non-redundant unique mapped reads, 1: unique reads, 2: mapped reads, 3: total reads) [0])\n");
    fprintf(stderr, "         -n       use repName (subfamily) as filter [null]\n");
    fprintf(stderr, "         -f       use repFamily as filter [null]\n");
    fprintf(stderr, "         -c       use repClass as filter [null]\n");
    fprintf(stderr, "         -t       only output repeats have more than [1] reads mapped\n");
    fprintf(stderr, "         -r       output the list of reads [off]\n");
    fprintf(stderr, "         -R       remove redundant reads [off]\n");
    fprintf(stderr, "         -T       treat 1 paired-end read as 2 single-end reads [off]\n");
    fprintf(stderr, "         -D       discard if only one end mapped in a paired end reads [off]\n");
    fprintf(stderr, "         -C       Add 'chr' string as prefix of reference sequence [off]\n");
    fprintf(stderr, "         -E       extend reads to represent fragment [150], specify 0 if want no extension\n");
    fprintf(stderr, "         -I       Insert length threshold [500]\n");
    fprintf(stderr, "         -o       output prefix [basename of input without extension]\n");
    fprintf(stderr, "         -h       help message\n");
    fprintf(stderr, "         -?       help message\n");
    fprintf(stderr, "\n");
    return 1;
}

int main_filter(int argc, char **argv)
{
    run_opts o;
    memset(&o, 0, sizeof o);
    o.mapq = 10;
    o.isize = 500;
    o.extension = 150;
    o.min_cov = 0.0001f;
    int optthreshold = 1, optreadlist = 0, optNorm = 0, c, filterField = 0;
    char *optoutput = NULL, *optname = NULL, *optclass = NULL, *optfamily = NULL;
    const time_t start_time = time(NULL);
    while ((c = getopt(argc, argv, "SQ:g:N:n:c:t:f:rRTDCE:I:o:h?")) >= 0) {
        switch (c) {
        case 'S': o.is_sam = 1; break;
        case 'Q': o.mapq = (unsigned)strtol(optarg, 0, 0); break;
        case 'g': o.min_cov = (float)atof(optarg); break;
        case 'N': optNorm = (int)(unsigned)strtol(optarg, 0, 0); break;
        case 't': optthreshold = (int)(unsigned)strtol(optarg, 0, 0); break;
        case 'r': optreadlist = 1; break;
        case 'R': o.dedup = 1; break;
        case 'T': o.treat = 1; break;
        case 'D': o.discard = 1; break;
        case 'C': o.add_chr = 1; break;
        case 'n': optname = strdup(optarg); break;
        case 'c': optclass = strdup(optarg); break;
        case 'f': optfamily = strdup(optarg); break;
        case 'E': o.extension = (unsigned)strtol(optarg, 0, 0); break;
        case 'I': o.isize = (unsigned)strtol(optarg, 0, 0); break;
        case 'o': optoutput = strdup(optarg); break;
        case 'h':
        case '?': return filter_usage();
        default: return 1;
        }
    }
    if (optind + 4 > argc) return filter_usage();
    o.chr_size_file = argv[optind];
    o.rep_size_file = argv[optind + 1];
    o.rmsk_file = argv[optind + 2];
    o.aln_arg = argv[optind + 3];
    if ((optname && optclass) || (optname && optfamily) || (optclass && optfamily))
        die("Please specify only one filter, either -n, -c or -f.");
    int nindex = 0;
    if (optNorm == 0) nindex = 7;
    else if (optNorm == 1) nindex = 8;
    else if (optNorm == 2) nindex = 6;
    else if (optNorm == 3) nindex = 4;
    else die("Wrong normalization method specified");
    const char *subfam = "ALL";
    if (optname) { subfam = optname; filterField = 10; }
    else if (optclass) { subfam = optclass; filterField = 11; }
    else if (optfamily) { subfam = optfamily; filterField = 12; }
    if (strcmp(subfam, "ALL") == 0) {
        fprintf(stderr, "* You didn't specify any filter, will output all repeats\n");
        filterField = 0;
    }
    char *output;
    if (optoutput) {
        output = optoutput;
    } else {
        char *copy = xstrdup(o.aln_arg);
        output = filename_without_ext(basename(copy));
        free(copy);
    }

    /* what is order-dependent stays with one rank: -R, the read-name lists (-r), SAM text */
    const int splittable = !o.is_sam && !o.dedup && !optreadlist;
    multi_begin(splittable, o.aln_arg, 0);
    stream_prefetch_allow(&o, 1, !o.is_sam && !o.dedup && !optreadlist);
    gpu_warmup_start(!o.is_sam, o.aln_arg, 0, splittable);
    sizes_t chr_sizes, rep_sizes;
    sizes_load(o.chr_size_file, &chr_sizes);
    stream_sizes_ready(&chr_sizes);
    sizes_load(o.rep_size_file, &rep_sizes);
    fprintf(stderr, "* Start to parse the rmsk file\n");
    rmsk_t rm;
    rmsk_load(o.rmsk_file, &chr_sizes, &rep_sizes, filterField, subfam, &rm);
    if (filterField == 0) {
        fprintf(stderr, "* Total %d repeats found.\n", rm.repeat_num);
    } else {
        if (rm.repeat_num <= 0) die("* No repeats found related to [%s], typo? or specify wrong repName/Class/Family filter?", subfam);
        fprintf(stderr, "* Total %d repeats for [%s].\n", rm.repeat_num, subfam);
    }
    fprintf(stderr, "* Start to parse the SAM/BAM file\n");
    itx_engine *eng = NULL;
    itx_table *tab = NULL;
    char **locus_names = NULL;
    host_counts hc = {0, 0};
    run_stream(&o, &rm, &chr_sizes, 1, 0, 10000, optreadlist, &eng, &tab, optreadlist ? &locus_names : NULL, &hc);

    fprintf(stderr, "* Preparing the output file\n");
    char *out = NULL, *outReport = NULL;
    if (asprintf(&out, "%s_%s.iteres.loci", output, subfam) < 0) die("Preparing output wrong");
    if (asprintf(&outReport, "%s_%s.iteres.reportloci", output, subfam) < 0) die("Preparing output wrong");
    uint64_t cnt[13];
    itx_result res;
    memset(&res, 0, sizeof res);
    res.cnt = cnt;
    res.locus_cnt = xcalloc(rm.n_rows + 1, sizeof(uint32_t));
    if (stream_finish(eng, &res) != ITX_OK) die("itx_engine_finish: %s", itx_last_error());
    cnt[11] -= hc.dup_unique;                     /* reads_nonredundant_unique: -R duplicates never reach it (generic.c:524-539) */
    write_filter_out(&rm, res.locus_cnt, locus_names, out, optreadlist, optthreshold, subfam, cnt[nindex]);
    fprintf(stderr, "* Preparing report file\n");
    write_report(outReport, cnt, o.mapq, subfam);
    itx_engine_destroy(eng);
    itx_table_destroy(tab);
    rmsk_free(&rm);
    sizes_free(&chr_sizes);
    sizes_free(&rep_sizes);
    fprintf(stderr, "* Done, time used %.0f seconds.\n", difftime(time(NULL), start_time));
    return 0;
}
