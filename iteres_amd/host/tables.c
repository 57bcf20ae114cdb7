/* tables.c — error helpers, name tables in kent-hash iteration order, the two-column size files and the
 * rmsk.txt loader. Restates what stat.c:137-141 / filter.c:121-125 set up before the record loop:
 * hashNameIntFile (cuskent/obscure.c:139-150) and rmsk2binKeeperHash (generic.c:1578-1707), minus the
 * binKeeper itself, which lives on the GPU (itx_table_create). */
#define _GNU_SOURCE
#include "itx_host.h"

#include <ctype.h>
#include <errno.h>
#include <stdarg.h>
#include <stdlib.h>
#include <omp.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

/* multi.c: a rank other than 0 keeps its banners to itself (stderr is /dev/null there) but not its errors */
FILE *itx_err_stream;
const char *itx_err_prefix;
void (*itx_die_hook)(void);
#define ERR_OUT (itx_err_stream ? itx_err_stream : stderr)

void die(const char *fmt, ...)
{
    va_list ap;
    fflush(stdout);
    if (itx_err_prefix) fputs(itx_err_prefix, ERR_OUT);
    va_start(ap, fmt);
    vfprintf(ERR_OUT, fmt, ap);
    va_end(ap);
    fputc('\n', ERR_OUT);
    if (itx_die_hook) itx_die_hook();
    /* errAbort ends with exit(-1) (errabort.c:166-172): status 255. Here other threads may be busy (OpenMP workers, the
     * loader, the HIP runtime starting up on the warm-up thread) and exit()'s teardown would race with them: flush what
     * stdio holds and leave at once with the same status. */
    fflush(NULL);
    _exit(255);
}

void warnf(const char *fmt, ...)
{
    va_list ap;
    if (itx_err_prefix) fputs(itx_err_prefix, ERR_OUT);
    va_start(ap, fmt);
    vfprintf(ERR_OUT, fmt, ap);
    va_end(ap);
    fputc('\n', ERR_OUT);
}

void *xmalloc(size_t n)
{
    void *p = malloc(n ? n : 1);
    if (!p) die("Out of memory needMem - request size %llu bytes", (unsigned long long)n);
    return p;
}
void *xcalloc(size_t n, size_t sz)
{
    void *p = calloc(n ? n : 1, sz ? sz : 1);
    if (!p) die("Out of memory needMem - request size %llu bytes", (unsigned long long)(n * sz));
    return p;
}
void *xrealloc(void *p, size_t n)
{
    void *q = realloc(p, n ? n : 1);
    if (!q) die("Out of memory needMoreMem - request size %llu bytes", (unsigned long long)n);
    return q;
}
char *xstrdup(const char *s)
{
    size_t n = strlen(s) + 1;
    char *p = xmalloc(n);
    memcpy(p, s, n);
    return p;
}

/* -------------------------------------------------------------------------------------------- names */
uint32_t kent_hash_string(const char *s)
{
    uint32_t r = 0;
    int c;
    while ((c = *s++) != '\0') r += (r << 3) + (uint32_t)c;     /* plain char: signed on x86-64, as in the reference */
    return r;
}

void names_init(names_t *t)
{
    memset(t, 0, sizeof *t);
    t->nbucket = 1024;
    t->bucket = xmalloc(sizeof(uint32_t) * t->nbucket);
    memset(t->bucket, 0xff, sizeof(uint32_t) * t->nbucket);
}

void names_free(names_t *t)
{
    for (uint32_t i = 0; i < t->n; i++) free(t->name[i]);
    free(t->name);
    free(t->bucket);
    free(t->next);
    memset(t, 0, sizeof *t);
}

int64_t names_find(const names_t *t, const char *s)
{
    if (!t->nbucket) return -1;
    for (uint32_t i = t->bucket[kent_hash_string(s) & (t->nbucket - 1)]; i != 0xffffffffu; i = t->next[i])
        if (strcmp(t->name[i], s) == 0) return i;
    return -1;
}

uint32_t names_intern(names_t *t, const char *s)
{
    int64_t f = names_find(t, s);
    if (f >= 0) return (uint32_t)f;
    if (t->n == t->cap) {
        t->cap = t->cap ? t->cap * 2 : 256;
        t->name = xrealloc(t->name, sizeof(char *) * t->cap);
        t->next = xrealloc(t->next, sizeof(uint32_t) * t->cap);
    }
    if (t->n >= t->nbucket) {                              /* keep chains short */
        t->nbucket *= 4;
        t->bucket = xrealloc(t->bucket, sizeof(uint32_t) * t->nbucket);
        memset(t->bucket, 0xff, sizeof(uint32_t) * t->nbucket);
        for (uint32_t i = 0; i < t->n; i++) {
            uint32_t b = kent_hash_string(t->name[i]) & (t->nbucket - 1);
            t->next[i] = t->bucket[b];
            t->bucket[b] = i;
        }
    }
    uint32_t id = t->n++;
    t->name[id] = xstrdup(s);
    uint32_t b = kent_hash_string(s) & (t->nbucket - 1);
    t->next[id] = t->bucket[b];
    t->bucket[b] = id;
    return id;
}

struct ko {
    uint32_t bucket, id;
};
static int ko_cmp(const void *a, const void *b)
{
    const struct ko *x = a, *y = b;
    if (x->bucket != y->bucket) return x->bucket < y->bucket ? -1 : 1;
    return x->id > y->id ? -1 : (x->id < y->id ? 1 : 0);     /* newest insertion first inside a bucket */
}

/* hashFirst/hashNext walk buckets 0..size-1, each bucket newest-first (hashAddN prepends, cuskent/hash.c:131-133);
 * the table doubles whenever elCount > size after an add (hash.c:136-140) and hashResize keeps the in-bucket
 * order (hash.c:389-409). The final order therefore depends only on the final size. */
void names_kent_order(const names_t *t, int start_pow2, uint32_t *order)
{
    uint64_t size = 1ull << start_pow2;
    for (uint64_t n = 1; n <= t->n; n++)
        if (n > size) size *= 2;
    struct ko *k = xmalloc(sizeof *k * (t->n ? t->n : 1));
    for (uint32_t i = 0; i < t->n; i++) {
        k[i].bucket = kent_hash_string(t->name[i]) & (uint32_t)(size - 1);
        k[i].id = i;
    }
    qsort(k, t->n, sizeof *k, ko_cmp);
    for (uint32_t i = 0; i < t->n; i++) order[i] = k[i].id;
    free(k);
}

/* ------------------------------------------------------------------------------------- line reading */
typedef struct {
    FILE *f;
    int is_pipe;
    pid_t child;              /* the decompressor behind a pipe */
    char *buf;
    size_t cap;
    int line_ix;
    const char *name;
} lines_t;

static int ends_with(const char *s, const char *suf)
{
    size_t a = strlen(s), b = strlen(suf);
    return a >= b && strcmp(s + a - b, suf) == 0;
}

/* generic.c:43-51 lineFileOpen2 + cuskent/linefile.c:34-60 (compressed inputs go through a decompressor pipe) */
static void lines_open(lines_t *l, const char *path)
{
    struct stat sb;
    memset(l, 0, sizeof *l);
    l->name = path;
    if (stat(path, &sb) == 0 && S_ISDIR(sb.st_mode)) die("Error: %s is a directory not a file", path);
    const char *prog = NULL, *flag = NULL;
    if (ends_with(path, ".gz") || ends_with(path, ".Z")) prog = "gzip", flag = "-dc";
    else if (ends_with(path, ".bz2")) prog = "bzip2", flag = "-dc";
    else if (ends_with(path, ".zip")) prog = "unzip", flag = "-p";
    if (prog) {
        /* the decompressor is exec'd with the path as an argument of its own, like the reference's pipeline
         * (cuskent/pipeline.c): no shell sees the file name */
        if (access(path, R_OK) != 0) die("Couldn't open %s , %s", path, strerror(errno));
        int fd[2];
        if (pipe(fd) != 0) die("Couldn't open %s , %s", path, strerror(errno));
        const pid_t pid = fork();
        if (pid < 0) die("Couldn't open %s , %s", path, strerror(errno));
        if (pid == 0) {
            dup2(fd[1], 1);
            close(fd[0]);
            close(fd[1]);
            char *arg = xmalloc(strlen(path) + 3);          /* a name that starts with '-' must not read as an option */
            sprintf(arg, "%s%s", path[0] == '-' ? "./" : "", path);
            execlp(prog, prog, flag, arg, (char *)NULL);
            _exit(127);
        }
        close(fd[1]);
        l->f = fdopen(fd[0], "r");
        l->child = pid;
        l->is_pipe = 1;
    } else {
        l->f = fopen(path, "r");
    }
    if (!l->f) die("Couldn't open %s , %s", path, strerror(errno));
}

/* next non-blank line that does not start with '#', chopped on white space (cuskent/linefile.c:855-870,
 * common.c:1915-1953): returns the number of words (at most max), 0 at end of file */
static int lines_next_words(lines_t *l, char **w, int max)
{
    ssize_t len;
    while ((len = getline(&l->buf, &l->cap, l->f)) >= 0) {
        l->line_ix++;
        char *p = l->buf;
        if (p[0] == '#') continue;
        int n = 0;
        for (;;) {
            if (n >= max) break;
            while (isspace((unsigned char)*p)) ++p;
            if (*p == 0) break;
            w[n++] = p;
            while (*p && !isspace((unsigned char)*p)) ++p;
            if (*p == 0) break;
            *p++ = 0;
        }
        if (n) return n;
    }
    return 0;
}

static void lines_close(lines_t *l)
{
    if (l->f) {
        fclose(l->f);
        if (l->is_pipe && l->child > 0) {
            int st;
            while (waitpid(l->child, &st, 0) < 0 && errno == EINTR) {}
        }
    }
    free(l->buf);
    memset(l, 0, sizeof *l);
}

/* -------------------------------------------------------------------------------------------- sizes */
void sizes_load(const char *path, sizes_t *out)
{
    lines_t l;
    char *w[2];
    int n;
    names_init(&out->names);
    out->value = NULL;
    size_t cap = 0;
    /* hashNameIntFile uses lineFileOpen (no directory check message of its own) */
    lines_open(&l, path);
    while ((n = lines_next_words(&l, w, 2)) != 0) {
        if (n < 2) die("Expecting %d words line %d of %s got %d", 2, l.line_ix, path, n);
        if (w[1][0] != '-' && !isdigit((unsigned char)w[1][0]))
            die("Expecting number field %d line %d of %s, got %s", 2, l.line_ix, path, w[1]);
        uint32_t id = names_intern(&out->names, w[0]);     /* a repeated name keeps its id; the later value wins */
        if (id >= cap) {
            cap = cap ? cap * 2 : 256;
            if (cap <= id) cap = id + 1;
            out->value = xrealloc(out->value, sizeof(int64_t) * cap);
        }
        out->value[id] = atoi(w[1]);
    }
    lines_close(&l);
}

void sizes_free(sizes_t *s)
{
    names_free(&s->names);
    free(s->value);
    s->value = NULL;
}

int64_t sizes_get(const sizes_t *s, const char *name, int64_t dflt)
{
    int64_t i = names_find(&s->names, name);
    return i < 0 ? dflt : s->value[i];
}

/* --------------------------------------------------------------------------------------------- rmsk */
#define GROW(ptr, n, cap, type)                                   \
    do {                                                          \
        if ((n) >= (cap)) {                                       \
            (cap) = (cap) ? (cap) * 2 : 1024;                     \
            (ptr) = xrealloc((ptr), sizeof(type) * (cap));        \
        }                                                         \
    } while (0)

/* the whole file in memory (also when it comes through a decompressor pipe), NUL-terminated */
char *slurp_text(const char *path, size_t *len)
{
    lines_t l;
    lines_open(&l, path);
    {
        /* a plain file of known size: one buffer, read as slices by the host threads (the doubling loop below copies a
         * half-gigabyte rmsk file three times over, on one thread) */
        struct stat sb;
        const int fd = fileno(l.f);
        if (!l.is_pipe && fd >= 0 && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > (off_t)(8u << 20) && ftello(l.f) == 0) {
            const size_t size = (size_t)sb.st_size;
            char *text = xmalloc(size + 1);
            int T = omp_get_max_threads();
            if (T < 1) T = 1;
            size_t bad = 0;
#pragma omp parallel for schedule(static, 1) num_threads(T) reduction(+ : bad)
            for (int t = 0; t < T; t++) {
                size_t at = size * (size_t)t / (size_t)T;
                const size_t hi = size * ((size_t)t + 1) / (size_t)T;
                while (at < hi) {
                    const ssize_t k = pread(fd, text + at, hi - at, (off_t)at);
                    if (k <= 0) {
                        bad++;
                        break;
                    }
                    at += (size_t)k;
                }
            }
            if (!bad) {
                text[size] = 0;
                lines_close(&l);
                *len = size;
                return text;
            }
            free(text);                                             /* the file changed under us: the plain way */
        }
    }
    size_t flen = 0, fcap = 1u << 24;
    char *text = xmalloc(fcap + 1);
    for (;;) {
        const size_t got = fread(text + flen, 1, fcap - flen, l.f);
        flen += got;
        if (got == 0) break;
        if (flen == fcap) {
            fcap *= 2;
            text = xrealloc(text, fcap + 1);
        }
    }
    text[flen] = 0;
    lines_close(&l);
    *len = flen;
    return text;
}

/* The rmsk file is parsed in pieces of whole lines, one per thread, each with the reference's row-by-row bookkeeping
 * (generic.c:1578-1707) on its own tables; the pieces are then joined in file order, which gives every name the id it
 * would have got from one pass (ids are first-appearance ranks, and the pieces are contiguous). */
typedef struct {
    rmsk_t r;
    size_t cap_rows, cap_chr, cap_rep, cap_fam, cap_cla;
    char *err;                 /* first fatal message of the piece, NULL when none */
} rmsk_part;

#define PART_DIE(...)                                   \
    do {                                                \
        if (asprintf(&pt->err, __VA_ARGS__) < 0) pt->err = NULL; \
        return;                                         \
    } while (0)

static void rmsk_parse_piece(char *text, size_t lo, size_t hi, size_t flen, const char *path, const sizes_t *chr_sizes, const sizes_t *rep_sizes,
                             int filter_field, const char *filter_name, rmsk_part *pt)
{
    rmsk_t *r = &pt->r;
    size_t cap_rows = 0, cap_chr = 0, cap_rep = 0, cap_fam = 0, cap_cla = 0;
    names_init(&r->chroms);
    names_init(&r->reps);
    names_init(&r->fams);
    names_init(&r->clas);
    /* lo and hi are line starts (rmsk_load cuts the file before any piece puts its NULs into the text) */
    size_t p = lo;
    while (p < hi) {
        char *line = text + p;
        const char *nl = memchr(line, '\n', flen - p);
        const size_t next = nl ? (size_t)(nl - text) + 1 : flen;
        char *end = text + (nl ? (size_t)(nl - text) : flen);
        const size_t line_off = p;
        p = next;
        if (*line == '#') continue;
        /* chopByWhite into at most 17 words (cuskent/linefile.c:855-870, common.c:1915-1953) */
        char *w[17];
        int n = 0;
        for (char *c = line;;) {
            if (n >= 17) break;
            while (c < end && isspace((unsigned char)*c)) ++c;
            if (c >= end || *c == 0) break;
            w[n++] = c;
            while (c < end && *c && !isspace((unsigned char)*c)) ++c;
            if (c >= end || *c == 0) break;
            *c++ = 0;
        }
        if (end < text + flen) *end = 0;
        if (n == 0) continue;
        if (n < 17) {
            int line_ix = 1;
            for (size_t k = 0; k < line_off; k++) line_ix += text[k] == '\n';
            PART_DIE("Expecting %d words line %d of %s got %d", 17, line_ix, path, n);
        }
        if (filter_field != 0 && strcmp(filter_name, w[filter_field]) != 0) continue;      /* generic.c:1588-1591 */
        r->repeat_num++;
        /* generic.c:1594-1607: (unsigned int)strtol(..., 0) */
        const char strand = w[9][0];
        itx_row row;
        row.cons_start = (uint32_t)strtol(strand == '+' ? w[13] : w[15], NULL, 0);
        row.cons_end = (uint32_t)strtol(w[14], NULL, 0);
        row.start = (uint32_t)strtol(w[6], NULL, 0);
        row.end = (uint32_t)strtol(w[7], NULL, 0);
        const uint32_t length = row.end - row.start;
        /* generic.c:1613-1626: first row of a chromosome creates its binKeeper if the size file knows it */
        int64_t ci = names_find(&r->chroms, w[5]);
        if (ci < 0) {
            const int size = (int)sizes_get(chr_sizes, w[5], 0);
            if (size == 0) continue;                                                        /* freermsk + continue */
            if (size < 0) PART_DIE("bad range %d,%d in binKeeperNew", 0, size);                  /* cuskent/binRange.c:145-146 */
            ci = names_intern(&r->chroms, w[5]);
            GROW(r->chrom_size, (size_t)ci, cap_chr, int64_t);
            r->chrom_size[ci] = size;
        }
        /* binKeeperAdd's range check (cuskent/binRange.c:176-178); itx_table_create repeats it for the whole table */
        {
            const int s = (int)row.start, e = (int)row.end, mx = (int)r->chrom_size[ci];
            if (s < 0 || e > mx || s > e) PART_DIE("(%d %d) out of range (%d %d) in binKeeperAdd", s, e, 0, mx);
        }
        row.chrom = (int32_t)names_find(&chr_sizes->names, w[5]);      /* the engine indexes chromosomes as the size file does */
        row.rep = row.fam = row.cla = 0;
        {   /* generic.c:1628-1693. With a name/class/family filter the reference leaves hashRep/hashFam/hashCla empty;
             * the rows still carry their own strings (printed by writeFilterOut), so the ids are kept either way. */
            const uint32_t before_rep = r->reps.n, before_fam = r->fams.n, before_cla = r->clas.n;
            const uint32_t rep = names_intern(&r->reps, w[10]);
            const uint32_t cla = names_intern(&r->clas, w[11]);
            const uint32_t fam = names_intern(&r->fams, w[12]);
            if (rep == before_rep) {
                if (rep >= cap_rep) {
                    cap_rep = cap_rep ? cap_rep * 2 : 1024;
                    r->rep_len = xrealloc(r->rep_len, sizeof(uint32_t) * cap_rep);
                    r->rep_fam = xrealloc(r->rep_fam, sizeof(uint32_t) * cap_rep);
                    r->rep_cla = xrealloc(r->rep_cla, sizeof(uint32_t) * cap_rep);
                    r->rep_genome = xrealloc(r->rep_genome, sizeof(uint64_t) * cap_rep);
                    r->rep_total = xrealloc(r->rep_total, sizeof(uint64_t) * cap_rep);
                }
                r->rep_len[rep] = (uint32_t)(int)sizes_get(rep_sizes, w[10], 0);
                r->rep_fam[rep] = fam;
                r->rep_cla[rep] = cla;
                r->rep_genome[rep] = 0;
                r->rep_total[rep] = 0;
            }
            if (fam == before_fam) {
                if (fam >= cap_fam) {
                    cap_fam = cap_fam ? cap_fam * 2 : 256;
                    r->fam_cla = xrealloc(r->fam_cla, sizeof(uint32_t) * cap_fam);
                    r->fam_genome = xrealloc(r->fam_genome, sizeof(uint64_t) * cap_fam);
                    r->fam_total = xrealloc(r->fam_total, sizeof(uint64_t) * cap_fam);
                }
                r->fam_cla[fam] = cla;
                r->fam_genome[fam] = 0;
                r->fam_total[fam] = 0;
            }
            if (cla == before_cla) {
                if (cla >= cap_cla) {
                    cap_cla = cap_cla ? cap_cla * 2 : 64;
                    r->cla_genome = xrealloc(r->cla_genome, sizeof(uint64_t) * cap_cla);
                    r->cla_total = xrealloc(r->cla_total, sizeof(uint64_t) * cap_cla);
                }
                r->cla_genome[cla] = 0;
                r->cla_total[cla] = 0;
            }
            r->rep_genome[rep]++;
            r->rep_total[rep] += length;
            r->fam_genome[fam]++;
            r->fam_total[fam] += length;
            r->cla_genome[cla]++;
            r->cla_total[cla] += length;
            row.rep = rep;
            row.fam = fam;
            row.cla = cla;
        }
        if (r->n_rows >= cap_rows) {
            cap_rows = cap_rows ? cap_rows * 2 : 4096;
            r->rows = xrealloc(r->rows, sizeof(itx_row) * cap_rows);
            r->row_chrom_name = xrealloc(r->row_chrom_name, sizeof(uint32_t) * cap_rows);
        }
        r->rows[r->n_rows] = row;
        r->row_chrom_name[r->n_rows] = (uint32_t)ci;
        r->n_rows++;
    }
    pt->cap_rows = cap_rows;
    (void)cap_chr; (void)cap_rep; (void)cap_fam; (void)cap_cla;
}

void rmsk_load(const char *path, const sizes_t *chr_sizes, const sizes_t *rep_sizes, int filter_field, const char *filter_name,
               rmsk_t *r)
{
    memset(r, 0, sizeof *r);
    size_t flen = 0;
    char *text = slurp_text(path, &flen);
    int T = omp_get_max_threads();
    if (T < 1) T = 1;
    if ((size_t)T > flen / 65536 + 1) T = (int)(flen / 65536 + 1);
    rmsk_part *parts = xcalloc((size_t)T, sizeof *parts);
    size_t *cut = xcalloc((size_t)T + 1, sizeof *cut);            /* piece t = lines starting in [cut[t], cut[t+1]) */
    for (int t = 1; t < T; t++) {
        const size_t at = flen * (size_t)t / (size_t)T;            /* a line belongs to the piece its first byte lies in */
        const char *nl = memchr(text + at - 1, '\n', flen - (at - 1));
        cut[t] = nl ? (size_t)(nl - text) + 1 : flen;
        if (cut[t] < cut[t - 1]) cut[t] = cut[t - 1];
    }
    cut[T] = flen;
#pragma omp parallel for schedule(static, 1) num_threads(T)
    for (int t = 0; t < T; t++)
        rmsk_parse_piece(text, cut[t], cut[t + 1], flen, path, chr_sizes, rep_sizes, filter_field, filter_name, &parts[t]);
    free(cut);
    for (int t = 0; t < T; t++)
        if (parts[t].err) die("%s", parts[t].err);             /* the first one in file order, like a single pass */
    /* ---- join in file order */
    names_init(&r->chroms);
    names_init(&r->reps);
    names_init(&r->fams);
    names_init(&r->clas);
    size_t total_rows = 0;
    for (int t = 0; t < T; t++) total_rows += parts[t].r.n_rows;
    r->rows = xmalloc(sizeof(itx_row) * (total_rows ? total_rows : 1));
    r->row_chrom_name = xmalloc(sizeof(uint32_t) * (total_rows ? total_rows : 1));
    size_t cap_chr = 0, cap_rep = 0, cap_fam = 0, cap_cla = 0;
    uint32_t *(*maps)[4] = xcalloc((size_t)T, sizeof *maps);
    size_t *row_base = xcalloc((size_t)T, sizeof *row_base);
    for (int t = 0; t < T; t++) {
        rmsk_t *q = &parts[t].r;
        r->repeat_num += q->repeat_num;
        uint32_t *mc = xmalloc(sizeof(uint32_t) * (q->chroms.n + 1)), *mf = xmalloc(sizeof(uint32_t) * (q->fams.n + 1));
        uint32_t *ml = xmalloc(sizeof(uint32_t) * (q->clas.n + 1)), *mr = xmalloc(sizeof(uint32_t) * (q->reps.n + 1));
        for (uint32_t i = 0; i < q->chroms.n; i++) {
            const uint32_t g = names_intern(&r->chroms, q->chroms.name[i]);
            GROW(r->chrom_size, (size_t)g, cap_chr, int64_t);
            r->chrom_size[g] = q->chrom_size[i];
            mc[i] = g;
        }
        for (uint32_t i = 0; i < q->clas.n; i++) {
            const uint32_t before = r->clas.n, g = names_intern(&r->clas, q->clas.name[i]);
            if (g == before) {
                if (g >= cap_cla) {
                    cap_cla = cap_cla ? cap_cla * 2 : 64;
                    r->cla_genome = xrealloc(r->cla_genome, sizeof(uint64_t) * cap_cla);
                    r->cla_total = xrealloc(r->cla_total, sizeof(uint64_t) * cap_cla);
                }
                r->cla_genome[g] = 0;
                r->cla_total[g] = 0;
            }
            r->cla_genome[g] += q->cla_genome[i];
            r->cla_total[g] += q->cla_total[i];
            ml[i] = g;
        }
        for (uint32_t i = 0; i < q->fams.n; i++) {
            const uint32_t before = r->fams.n, g = names_intern(&r->fams, q->fams.name[i]);
            if (g == before) {
                if (g >= cap_fam) {
                    cap_fam = cap_fam ? cap_fam * 2 : 256;
                    r->fam_cla = xrealloc(r->fam_cla, sizeof(uint32_t) * cap_fam);
                    r->fam_genome = xrealloc(r->fam_genome, sizeof(uint64_t) * cap_fam);
                    r->fam_total = xrealloc(r->fam_total, sizeof(uint64_t) * cap_fam);
                }
                r->fam_cla[g] = ml[q->fam_cla[i]];              /* class of the family's first row in the file */
                r->fam_genome[g] = 0;
                r->fam_total[g] = 0;
            }
            r->fam_genome[g] += q->fam_genome[i];
            r->fam_total[g] += q->fam_total[i];
            mf[i] = g;
        }
        for (uint32_t i = 0; i < q->reps.n; i++) {
            const uint32_t before = r->reps.n, g = names_intern(&r->reps, q->reps.name[i]);
            if (g == before) {
                if (g >= cap_rep) {
                    cap_rep = cap_rep ? cap_rep * 2 : 1024;
                    r->rep_len = xrealloc(r->rep_len, sizeof(uint32_t) * cap_rep);
                    r->rep_fam = xrealloc(r->rep_fam, sizeof(uint32_t) * cap_rep);
                    r->rep_cla = xrealloc(r->rep_cla, sizeof(uint32_t) * cap_rep);
                    r->rep_genome = xrealloc(r->rep_genome, sizeof(uint64_t) * cap_rep);
                    r->rep_total = xrealloc(r->rep_total, sizeof(uint64_t) * cap_rep);
                }
                r->rep_len[g] = q->rep_len[i];
                r->rep_fam[g] = mf[q->rep_fam[i]];              /* family / class of the name's first row in the file */
                r->rep_cla[g] = ml[q->rep_cla[i]];
                r->rep_genome[g] = 0;
                r->rep_total[g] = 0;
            }
            r->rep_genome[g] += q->rep_genome[i];
            r->rep_total[g] += q->rep_total[i];
            mr[i] = g;
        }
        /* the rows themselves move afterwards, all pieces at once */
        maps[t][0] = mc;
        maps[t][1] = mf;
        maps[t][2] = ml;
        maps[t][3] = mr;
        row_base[t] = r->n_rows;
        r->n_rows += q->n_rows;
    }
#pragma omp parallel for schedule(static, 1)
    for (int t = 0; t < T; t++) {
        rmsk_t *q = &parts[t].r;
        const uint32_t *mc = maps[t][0], *mf = maps[t][1], *ml = maps[t][2], *mr = maps[t][3];
        itx_row *dst = r->rows + row_base[t];
        uint32_t *dstc = r->row_chrom_name + row_base[t];
        for (size_t i = 0; i < q->n_rows; i++) {
            itx_row row = q->rows[i];
            row.rep = mr[row.rep];
            row.fam = mf[row.fam];
            row.cla = ml[row.cla];
            dst[i] = row;
            dstc[i] = mc[q->row_chrom_name[i]];
        }
        for (int k = 0; k < 4; k++) free(maps[t][k]);
        rmsk_free(q);
    }
    free(maps);
    free(row_base);
    free(parts);
    free(text);
}

void rmsk_free(rmsk_t *r)
{
    names_free(&r->chroms);
    names_free(&r->reps);
    names_free(&r->fams);
    names_free(&r->clas);
    free(r->rows);
    free(r->row_chrom_name);
    free(r->chrom_size);
    free(r->rep_len);
    free(r->rep_fam);
    free(r->rep_cla);
    free(r->fam_cla);
    free(r->rep_genome);
    free(r->rep_total);
    free(r->fam_genome);
    free(r->fam_total);
    free(r->cla_genome);
    free(r->cla_total);
    memset(r, 0, sizeof *r);
}

char *filename_without_ext(const char *path)
{
    char *s = xstrdup(path);
    char *dot = strrchr(s, '.');
    if (!dot || dot == s) return s;
    *dot = '\0';
    return s;
}
