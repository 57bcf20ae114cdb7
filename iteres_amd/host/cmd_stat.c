/* cmd_stat.c — `iteres stat`: same options, banners, output names and exit codes as stat.c:30-186 of the
 * reference; the record loop runs on the GPU engine (stream.c). */
#define _GNU_SOURCE
#include "itx_host.h"

#include <getopt.h>
#include <libgen.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

static int stat_usage(void)
{
    fprintf(stderr, "\n");
    fprintf(stderr, "Obtain alignment statistics for each repeat subfamily, family and class.\n\n");
    fprintf(stderr, "Usage:   iteres stat [options] <chromosome size file> <repeat size file> <rmsk.txt> <bam/sam alignment file1,file2,file3...>\n\n");
    fprintf(stderr, "Options: -S       input is SAM [off]\n");
    fprintf(stderr, "         -Q       unique reads mapping Quality threshold [10]\n");
    fprintf(stderr, "         -c       coverage threshold for overlapping [0.0001]\n");
    fprintf(stderr, "         -x       discard multi-reads if mapped to different subfamily [on]\n");
    fprintf(stderr, "         -N       normalized by number of (0: reads in repeats, 1: non-redundant reads, 2: mapped reads, 3: total reads) [0])\n");
    fprintf(stderr, "         -U       unique reads normalized by number of (0: unique mapped reads in repeats, 1: unique mapped reads, 2: total reads) [0])\n");
    fprintf(stderr, "         -R       remove redundant reads [off]\n");
    fprintf(stderr, "         -T       treat 1 paired-end read as 2 single-end reads [off]\n");
    fprintf(stderr, "         -D       discard if only one end mapped in a paired end reads [off]\n");
    fprintf(stderr, "         -w       keep the wiggle file [off]\n");
    fprintf(stderr, "         -B       output bed file of mapped reads [off]\n");
    fprintf(stderr, "         -V       output bed file of unique mapped reads [off]\n");
    fprintf(stderr, "         -C       Add 'chr' string as prefix of reference sequence [off]\n");
    fprintf(stderr, "         -E       extend reads to represent fragment [150], specify 0 if want no extension\n");
    fprintf(stderr, "         -I       Insert length threshold [500]\n");
    fprintf(stderr, "         -o       output prefix [basename of input without extension]\n");
    fprintf(stderr, "         -h       help message\n");
    fprintf(stderr, "         -?       help message\n");
    fprintf(stderr, "\n");
    return 1;
}

/* ITX_TIMING=1: phase wall times on stderr (not part of the reference's output) */
static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static char *fmt_name(const char *prefix, const char *suffix)
{
    char *s = NULL;
    if (asprintf(&s, "%s%s", prefix, suffix) < 0) die("Mem Error.\n");
    return s;
}

int main_stat(int argc, char **argv)
{
    run_opts o;
    memset(&o, 0, sizeof o);
    o.xa_veto = 1;
    o.mapq = 10;
    o.isize = 500;
    o.extension = 150;
    o.min_cov = 0.0001f;
    unsigned optNorm = 0, optNorm2 = 0;
    int optBed = 0, optBedUniq = 0, c;
    char *optoutput = NULL;
    const time_t start_time = time(NULL);
    while ((c = getopt(argc, argv, "SQ:c:xN:U:RTDwBVCo:E:I:h?")) >= 0) {
        switch (c) {
        case 'S': o.is_sam = 1; break;
        case 'Q': o.mapq = (unsigned)strtol(optarg, 0, 0); break;
        case 'c': o.min_cov = (float)atof(optarg); break;
        case 'x': o.xa_veto = 0; break;
        case 'N': optNorm = (unsigned)strtol(optarg, 0, 0); break;
        case 'U': optNorm2 = (unsigned)strtol(optarg, 0, 0); break;
        case 'R': o.dedup = 1; break;
        case 'T': o.treat = 1; break;
        case 'D': o.discard = 1; break;
        case 'w': o.keep_wig = 1; break;
        case 'B': optBed = 1; break;
        case 'V': optBedUniq = 1; break;
        case 'C': o.add_chr = 1; break;
        case 'E': o.extension = (unsigned)strtol(optarg, 0, 0); break;
        case 'I': o.isize = (unsigned)strtol(optarg, 0, 0); break;
        case 'o': optoutput = strdup(optarg); break;
        case 'h':
        case '?': return stat_usage();
        default: return 1;
        }
    }
    if (optind + 4 > argc) return stat_usage();
    o.chr_size_file = argv[optind];
    o.rep_size_file = argv[optind + 1];
    o.rmsk_file = argv[optind + 2];
    o.aln_arg = argv[optind + 3];

    /* stat.c:77-79 */
    int numFields = 1;
    for (const char *s = o.aln_arg; *s; s++)
        if (*s == ',') numFields++;
    if (numFields > 100) numFields = 100;
    fprintf(stderr, "* Provided %i BAM/SAM file(s)\n", numFields);
    char *output;
    if (optoutput) {
        output = optoutput;
    } else {                                                              /* stat.c:84: basename of the first file, last extension cut */
        char *first = xstrdup(o.aln_arg);
        char *comma = strchr(first, ',');
        if (comma) *comma = 0;
        output = filename_without_ext(basename(first));
        free(first);
    }
    char *outWig = fmt_name(output, ".iteres.wig"), *outWigUniq = fmt_name(output, ".iteres.unique.wig");
    char *outBigWig = fmt_name(output, ".iteres.bigWig"), *outBigWigUniq = fmt_name(output, ".iteres.unique.bigWig");
    char *outReport = fmt_name(output, ".iteres.report"), *outStat = fmt_name(output, ".iteres.subfamily.stat");
    char *outFam = fmt_name(output, ".iteres.family.stat"), *outCla = fmt_name(output, ".iteres.class.stat");
    int nindex = 0, nindex2 = 0;
    if (optNorm == 0) nindex = 9;
    else if (optNorm == 1) nindex = 8;
    else if (optNorm == 2) nindex = 6;
    else if (optNorm == 3) nindex = 0;
    else die("Wrong normalization method specified");
    if (optNorm2 == 0) nindex2 = 10;
    else if (optNorm2 == 1) nindex2 = 7;
    else if (optNorm2 == 2) nindex2 = 0;
    else die("Wrong normalization method specified");
    if (optBed) o.bed_path = fmt_name(output, ".iteres.bed");                     /* stat.c:104-111 */
    if (optBedUniq) o.bed_uniq_path = fmt_name(output, ".iteres.unique.bed");

    const int timing = getenv("ITX_TIMING") != NULL;
    const double t_begin = now_s();
    /* what is order-dependent stays with one rank: -R, the bed files, SAM text; everything else shards by record */
    const int splittable = !o.is_sam && !o.dedup && !optBed && !optBedUniq;
    multi_begin(splittable, o.aln_arg, 1);
    stream_prefetch_allow(&o, 0, !o.is_sam && !o.dedup && !optBed && !optBedUniq);
    gpu_warmup_start(!o.is_sam, o.aln_arg, 1, splittable);
    sizes_t chr_sizes, rep_sizes;
    sizes_load(o.chr_size_file, &chr_sizes);
    stream_sizes_ready(&chr_sizes);
    sizes_load(o.rep_size_file, &rep_sizes);
    fprintf(stderr, "* Parsing the rmsk file\n");
    rmsk_t rm;
    rmsk_load(o.rmsk_file, &chr_sizes, &rep_sizes, 0, "ALL", &rm);
    fprintf(stderr, "* Total %d repeats found.\n", rm.repeat_num);

    const double t_loaded = now_s();
    fprintf(stderr, "* Parsing the SAM/BAM file\n");
    itx_engine *eng = NULL;
    itx_table *tab = NULL;
    host_counts hc = {0, 0};
    run_stream(&o, &rm, &chr_sizes, 0, 1, 100000, 0, &eng, &tab, NULL, &hc);

    const double t_streamed = now_s();
    fprintf(stderr, "* Writing stats and Wig file\n");
    itx_table_info info;
    if (itx_table_get_info(tab, &info) != ITX_OK) die("itx_table_get_info: %s", itx_last_error());
    uint64_t cnt[13];
    itx_result res;
    memset(&res, 0, sizeof res);
    res.cnt = cnt;
    res.rep_cnt = xcalloc(2 * (size_t)rm.reps.n + 1, sizeof(uint64_t));
    res.fam_cnt = xcalloc(2 * (size_t)rm.fams.n + 1, sizeof(uint64_t));
    res.cla_cnt = xcalloc(2 * (size_t)rm.clas.n + 1, sizeof(uint64_t));
    res.cov = xcalloc(info.cov_len + 1, sizeof(uint32_t));
    res.cov_uniq = xcalloc(info.cov_len + 1, sizeof(uint32_t));
    if (stream_finish(eng, &res) != ITX_OK) die("itx_engine_finish: %s", itx_last_error());
    const double t_finished = now_s();
    cnt[11] -= hc.dup_unique;                     /* reads_nonredundant_unique: -R duplicates never reach it (generic.c:907-922) */
    cnt[12] = hc.diff_subfam;                     /* reads_diff_subfam (generic.c:978) */
    uint64_t *cov_off = xcalloc((size_t)rm.reps.n + 1, sizeof(uint64_t));
    itx_table_cov_offsets(tab, cov_off);
    write_wig_and_stat(&rm, &res, cov_off, outStat, o.keep_wig ? outWig : NULL, outFam, outCla, o.keep_wig ? outWigUniq : NULL, cnt[nindex],
                       cnt[nindex2]);

    const double t_stats = now_s();
    /* stat.c:156-158: the two wigs as bigWig (written from the vectors, not by re-reading the text) */
    fprintf(stderr, "* Generating bigWig files\n");
    {
        const char **nm = xcalloc((size_t)rm.reps.n + 1, sizeof *nm);
        uint32_t *ln = xcalloc((size_t)rm.reps.n + 1, sizeof *ln);
        const float **va = xcalloc((size_t)rm.reps.n + 1, sizeof *va), **vu = xcalloc((size_t)rm.reps.n + 1, sizeof *vu);
        /* the converter reads "%u" back as a double and stores a float */
        float *fa = xmalloc(sizeof(float) * (info.cov_len + 1)), *fu = xmalloc(sizeof(float) * (info.cov_len + 1));
#pragma omp parallel for schedule(static)
        for (long i = 0; i < (long)info.cov_len; i++) {
            fa[i] = (float)(double)res.cov[i];
            fu[i] = (float)(double)res.cov_uniq[i];
        }
        size_t k = 0;
        for (uint32_t i = 0; i < rm.reps.n; i++)
            if (rm.rep_len[i]) {
                nm[k] = rm.reps.name[i];
                ln[k] = rm.rep_len[i];
                va[k] = fa + cov_off[i];
                vu[k] = fu + cov_off[i];
                k++;
            }
        /* The two files share nothing but their inputs, and ITX_BW_PAIR=1 writes them side by side, each with half of the
         * threads. Measured on the 16 cores a rank has (500 M reads, 5 runs each): 0.48 s against 0.43 s one after the other —
         * every stretch of the writer that matters is already spread over the threads, so the pair only adds a second team. Off. */
        if (!getenv("ITX_BW_PAIR") || omp_get_max_threads() < 2) {
            write_bigwig(outBigWig, outWig, nm, ln, va, k);
            write_bigwig(outBigWigUniq, outWigUniq, nm, ln, vu, k);
        } else {
            const int half = omp_get_max_threads() / 2;
            omp_set_max_active_levels(2);
#pragma omp parallel sections num_threads(2)
            {
#pragma omp section
                {
                    omp_set_num_threads(half);
                    write_bigwig(outBigWig, outWig, nm, ln, va, k);
                }
#pragma omp section
                {
                    omp_set_num_threads(half);
                    write_bigwig(outBigWigUniq, outWigUniq, nm, ln, vu, k);
                }
            }
        }
        free(fa); free(fu);
        free(nm); free(ln); free(va); free(vu);
    }

    fprintf(stderr, "* Preparing report file\n");
    write_report(outReport, cnt, o.mapq, "ALL");

    if (timing) {
        fprintf(stderr, "[itx timing] load %.3f s, table+scan %.3f s, finish+write %.3f s\n", t_loaded - t_begin, t_streamed - t_loaded,
                now_s() - t_streamed);
        fprintf(stderr, "[itx timing] finish %.3f s, stat + wig files %.3f s, bigWig files + report %.3f s\n", t_finished - t_streamed,
                t_stats - t_finished, now_s() - t_stats);
    }
    const double t_free0 = now_s();
    itx_engine_destroy(eng);
    itx_table_destroy(tab);
    rmsk_free(&rm);
    sizes_free(&chr_sizes);
    sizes_free(&rep_sizes);
    if (timing) fprintf(stderr, "[itx timing] engine, table and name tables released %.3f s\n", now_s() - t_free0);
    fprintf(stderr, "* Done, time used %.0f seconds.\n", difftime(time(NULL), start_time));
    return 0;
}
