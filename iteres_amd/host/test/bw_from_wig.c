/* bw_from_wig.c — test tool: feeds a text wig of "fixedStep chrom=<name> start=1 step=1 span=1" blocks to the
 * product's bigWig writer (bigwig.c), so the writer can be checked on a machine without a GPU.
 *   bw_from_wig <in.wig> <out.bigWig> */
#define _GNU_SOURCE
#include "../itx_host.h"

#include <stdlib.h>
#include <string.h>
#include <time.h>

int main(int argc, char **argv)
{
    if (argc != 3) {
        fprintf(stderr, "usage: bw_from_wig <in.wig> <out.bigWig>\n");
        return 2;
    }
    FILE *f = fopen(argv[1], "r");
    if (!f) return 3;
    char **names = NULL;
    uint32_t *len = NULL;
    float **val = NULL;
    size_t n = 0, cap = 0, vcap = 0;
    char *line = NULL;
    size_t lcap = 0;
    while (getline(&line, &lcap, f) >= 0) {
        if (strncmp(line, "fixedStep", 9) == 0) {
            if (n == cap) {
                cap = cap ? cap * 2 : 64;
                names = realloc(names, cap * sizeof *names);
                len = realloc(len, cap * sizeof *len);
                val = realloc(val, cap * sizeof *val);
            }
            const char *c = strstr(line, "chrom=") + 6;
            names[n] = strndup(c, strcspn(c, " \t\n"));
            len[n] = 0;
            val[n] = NULL;
            vcap = 0;
            n++;
        } else if (n && line[0] != '\n') {
            if (len[n - 1] == vcap) {
                vcap = vcap ? vcap * 2 : 1024;
                val[n - 1] = realloc(val[n - 1], vcap * sizeof(float));
            }
            val[n - 1][len[n - 1]++] = (float)strtod(line, NULL);          /* lineFileNeedDouble, then the packed float */
        }
    }
    fclose(f);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    write_bigwig(argv[2], argv[1], (const char *const *)names, len, (const float *const *)val, n);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (getenv("ITX_TIMING")) fprintf(stderr, "[itx timing] write_bigwig %.3f s\n", (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec));
    return 0;
}
