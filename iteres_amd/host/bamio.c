/* bamio.c — alignment input for the host: BGZF/BAM and SAM text decoded straight into the engine's pinned
 * record SoA (tid, pos, tmpend, mapq, flag5, mpos, isize). Follows the file formats and, where behaviour is
 * observable, samtools 0.1.18 as vendored by the reference: bgzf.c:401-411,471-565 (block framing), bam.c:69-109
 * (header), bam.c:179-210 (record), bam.c:17-27 (bam_calend: only M, D, N advance), bam_aux.c:36-48 (aux walk),
 * bam_import.c:237-330 (SAM text; a mapped record without CIGAR is flagged unmapped). */
#define _GNU_SOURCE
#include "itx_host.h"

#include <ctype.h>
#include <dlfcn.h>
#include <omp.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include <sys/stat.h>
#include <unistd.h>

#define BGZF_MAX 0x10000

#include <time.h>
static double t_io, t_inflate, t_hop, t_parse;          /* ITX_TIMING: where the decoder's wall time goes */
static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static uint32_t rd_u32_at(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }

struct blk {
    size_t coff, csize, uoff, usize;
};

struct aln_reader {
    FILE *f;
    int is_sam;
    /* header */
    int n_targets;
    char **tname;
    names_t tnames;
    /* BGZF state: the file is consumed in chunks of many blocks; the blocks of a chunk are inflated in parallel
     * (they are independent gzip members), then records are located by one sequential hop over the block_len
     * fields and parsed in parallel straight into the staging SoA */
    uint8_t *cbuf;            /* compressed bytes of the current chunk (+ the incomplete block carried over):   */
    size_t clen;              /* a window into one of the two raw buffers below                                 */
    /* raw read-ahead (raw_next): a reader thread freads the next compressed chunk while this one is being inflated */
#define N_RAW_DEVICE (ITX_BAMWIN_LANES + 2)
#define N_RAW_DEVICE_FILE ALN_DEVICE_RAW_BUFFERS
    uint8_t *craw[N_RAW_DEVICE]; /* two for the host decoder. The device's, from a regular file: one being indexed, one whose pushes are
                                  * copying, the others read ahead (a buffer is free again once the pushes cut from it have COPIED their bytes: a
                                  * block the device declines is read from the file once more); from a pipe: one per push in flight + 2, held
                                  * until the push ends */
    uint32_t raw_lanes[N_RAW_DEVICE];   /* per buffer: the lanes whose pushes read their compressed bytes from it (bit s = lane s) */
    int n_raw, raw_cur;          /* raw_cur: the buffer cbuf points into */
    size_t io_got[N_RAW_DEVICE];      /* bytes the reader put into each buffer */
    long io_fill, io_take;       /* chunks read so far / taken by raw_next so far (chunk f lives in buffer f % n_raw) */
    int io_eof_seen;             /* the reader has met a short read: it reads no further */
    int io_fd;                /* >= 0: a regular file, read with pread at io_off (of io_size bytes)              */
    size_t io_off, io_size;
    size_t raw_step;          /* bytes per read step (aln_raw_step) */
    void (*win_hook)(void *, size_t);   /* aln_set_window_hook */
    void *win_hook_ctx;
    int io_on, io_stop, io_done; /* io_done: the file is read out and its last chunk taken */
    /* a share of the file (aln_open_range; one rank of a multi-GPU job): the stream starts at the BGZF block at byte
     * rg_lo_block of the file, rg_skip inflated bytes into it, and ends where the next share starts: rg_end_off bytes into
     * the block at rg_end_block (SIZE_MAX: at the end of the file) */
    int rg_on;
    size_t rg_lo_block, rg_skip, rg_end_block, rg_end_off, rg_end_csize;
    size_t io_abs_end;        /* file offset behind the last byte raw_next has delivered                           */
    size_t dstream_total;     /* inflated bytes of all blocks indexed before the current chunk (from rg_lo_block)  */
    size_t dskip_left;        /* inflated bytes still to be skipped at the front of the stream                     */
    int rg_stop_hit, rg_stop_missed, rg_verified;
    int rg_suspect;           /* something a false start (or damaged input) explains: the share is not to be trusted */
    void *hz;                 /* while the header is read by the plain host reader (a share that starts inside the file) */
    pthread_t io_thread;
    pthread_mutex_t io_mu;
    pthread_cond_t io_cv;
    uint8_t *ubuf;            /* inflated bytes: unconsumed tail of the previous chunk + this chunk            */
    size_t ulen, upos, ucap;
    size_t *rec_off;          /* start of every complete record in ubuf                                        */
    size_t n_rec, rec_next, rec_cap;
    size_t *spec;             /* locate_records: the pieces' starts before they are accepted                    */
    size_t spec_cap, hop_pieces, hop_redone;
    int eof;                  /* no more compressed input (end of file or a damaged block)                     */
    struct blk *blk;          /* block index of the chunk being inflated                                        */
    size_t blk_cap;
    /* device decoder (aln_use_device): two windows of inflated bytes live on the device, dw is the one being consumed */
    int dev, dw, dparsed, dlast, dflags, dseen_ok, dmore, dinput_done;
    long dk_begin, dk_ready, dk_cur;   /* chunks whose push was begun / has ended; the chunk whose window is being consumed (-1 none) */
    struct dev_job {
        struct blk *bl;        /* the chunk's blocks and where its compressed bytes lie, kept until its push has ended */
        size_t bl_cap, nb;
        const uint8_t *cbase;
        size_t cabs;           /* file offset of cbase (regular files: SIZE_MAX otherwise) */
        int w, last;
        size_t trunc;          /* SIZE_MAX, or: the window ends this many fresh bytes in (the share's end boundary) */
    } dj[ITX_BAMWIN_LANES];
    size_t dring_n[ITX_BAMWIN_WINDOWS];
    int dring_eof[ITX_BAMWIN_WINDOWS];
    uint8_t *dseen;           /* references with a mapped record in the current window                             */
    size_t dn_rec, drec_next, d_rewalked;
    itx_bgzf_block *dblk;     /* block index of a chunk for the device, and its per-block verdicts                */
    uint8_t *dstatus;
    size_t dblk_cap;
    uint8_t *hdr;             /* host copy of the window's front while the header is parsed                       */
    size_t hdr_len, hdr_pos;
    uint32_t *d_off;          /* side channels: offsets, XA marks and raw bytes of a batch's records              */
    uint8_t *d_xa, *d_raw;
    size_t d_side_cap, d_raw_cap;
    /* read-ahead (bgzf_load_chunk): spare buffer the loader thread inflates the next chunk into */
    uint8_t *nbuf;
    size_t ncap, nlen;
    int n_eof, pf_on, pf_state, pf_stop;       /* pf_state: 0 idle, 1 requested, 2 ready */
    pthread_t pf_thread;
    pthread_mutex_t pf_mu;
    pthread_cond_t pf_cv;
    /* SAM state */
    char *line;
    size_t line_cap;
    ssize_t pending_len;     /* first alignment line read while scanning the header, -1 none */
    long long n_lines;
};

/* ---- BGZF ------------------------------------------------------------------------------------------------ */
static aln_device_ops dev;                   /* .push_begin == NULL: the host decodes */
static int dev_nwin = ITX_BAMWIN_WINDOWS;   /* windows the pushes rotate through */
void aln_use_device(const aln_device_ops *ops)
{
    if (ops) dev = *ops;
    else memset(&dev, 0, sizeof dev);
    dev_nwin = dev.n_windows >= 2 && dev.n_windows <= ITX_BAMWIN_WINDOWS ? dev.n_windows : ITX_BAMWIN_WINDOWS;
}
#define DEV_CHK(call, what)                                                                                    \
    do {                                                                                                       \
        if ((call) != 0) die("device decoder: %s failed: %s", what, dev.last_error ? dev.last_error() : "?"); \
    } while (0)

/* the big byte buffers (compressed chunks, inflated chunks): page-locked when the device inflates */
static uint8_t *buf_alloc(size_t n)
{
    if (dev.alloc) {
        uint8_t *p = dev.alloc(n);
        if (!p) die("cannot get %zu bytes of page-locked memory", n);
        return p;
    }
    return xmalloc(n);
}
static void buf_free(uint8_t *p)
{
    if (!p) return;
    if (dev.release) dev.release(p);
    else free(p);
}
static uint8_t *buf_grow(uint8_t *old, size_t keep, size_t ncap)
{
    uint8_t *p = buf_alloc(ncap);
    if (old && keep) memcpy(p, old, keep);
    buf_free(old);
    return p;
}
#define CHUNK_COMPRESSED_DEFAULT (48u << 20)
#define CHUNK_COMPRESSED_DEVICE ALN_DEVICE_CHUNK  /* a lane per block on the device: it takes thousands of blocks to fill it */
/* compressed bytes read and inflated per step; ITX_BGZF_CHUNK overrides it (tests force many small steps) */
static size_t chunk_compressed(void)
{
    static size_t v;
    if (!v) {
        const char *e = getenv("ITX_BGZF_CHUNK");
        const long x = e ? atol(e) : 0;
        v = x >= 1 ? (size_t)x : dev.push_begin ? CHUNK_COMPRESSED_DEVICE : CHUNK_COMPRESSED_DEFAULT;
    }
    return v;
}
#define CHUNK_COMPRESSED chunk_compressed()

/* bgzf.c:401-411 check_header: gzip member with exactly one 6-byte extra field "BC" */
static int bgzf_header_ok(const uint8_t *h)
{
    return h[0] == 31 && h[1] == 139 && h[2] == 8 && (h[3] & 4) && (h[10] | h[11] << 8) == 6 && h[12] == 'B' && h[13] == 'C' &&
           (h[14] | h[15] << 8) == 2;
}

/* Raw DEFLATE of one block. libdeflate, when the system has it (dlopen, no build-time dependency), inflates two to three
 * times faster than zlib's inflate(); both are lossless decoders of the same stream, so the bytes are the same.
 * ITX_NO_LIBDEFLATE=1 keeps zlib. */
typedef struct libdeflate_decompressor ld_dec;
static ld_dec *(*ld_alloc)(void);
static int (*ld_inflate)(ld_dec *, const void *, size_t, void *, size_t, size_t *);
static int ld_state;                         /* 0 not probed, 1 in use, -1 unavailable */
static void ld_probe(void)
{
    if (ld_state) return;
    ld_state = -1;
    if (getenv("ITX_NO_LIBDEFLATE")) return;
    void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    ld_alloc = (ld_dec * (*)(void)) dlsym(h, "libdeflate_alloc_decompressor");
    ld_inflate = (int (*)(ld_dec *, const void *, size_t, void *, size_t, size_t *))dlsym(h, "libdeflate_deflate_decompress");
    if (ld_alloc && ld_inflate) ld_state = 1;
}

static int inflate_block(const uint8_t *src, size_t csize, uint8_t *dst, size_t usize)
{
    if (ld_state == 1) {
        static __thread ld_dec *d;
        if (!d) d = ld_alloc();
        size_t got = 0;
        if (d && ld_inflate(d, src + 18, csize - 18 - 8, dst, usize, &got) == 0 && got == usize) return 0;
        /* anything unexpected: let zlib have the last word */
    }
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    zs.next_in = (Bytef *)(src + 18);
    zs.avail_in = (uInt)(csize - 18 - 8);
    zs.next_out = dst;
    zs.avail_out = (uInt)usize;
    if (inflateInit2(&zs, -15) != Z_OK) return -1;
    const int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    return (rc == Z_STREAM_END && zs.total_out == usize) ? 0 : -1;
}

/* ---- a plain sequential BGZF reader on the host (pread + one block at a time): the header of a file whose records this
 * reader takes only a share of, and the search for the share's boundaries. Small amounts of data: no threads. */
typedef struct {
    int fd;
    size_t off, size;          /* the next block starts at file offset off; size of the file                         */
    uint8_t *u;                /* inflated bytes so far                                                               */
    size_t len, cap, pos;
    struct blk *b;             /* the blocks behind u: coff = file offset, csize, uoff, usize                         */
    size_t nb, bcap;
} hz_t;

/* inflates the next block behind u[len): 0 done, 1 end of file (or a block cut short by it), -1 not a block / damaged */
static int hz_block(hz_t *z)
{
    uint8_t h[18];
    if (z->off + 18 > z->size) return 1;
    if (pread(z->fd, h, 18, (off_t)z->off) != 18) return -1;
    if (!bgzf_header_ok(h)) return -1;
    const size_t bsize = (size_t)(h[16] | h[17] << 8) + 1;
    if (bsize < 26) return -1;
    if (z->off + bsize > z->size) return 1;
    uint8_t *c = xmalloc(bsize + 16);
    if (pread(z->fd, c, bsize, (off_t)z->off) != (ssize_t)bsize) {
        free(c);
        return -1;
    }
    const size_t usize = rd_u32_at(c + bsize - 4);
    if (usize > BGZF_MAX) {
        free(c);
        return -1;
    }
    if (z->len + usize + 64 > z->cap) {
        z->cap = (z->len + usize) * 2 + BGZF_MAX;
        z->u = xrealloc(z->u, z->cap);
    }
    if (usize && inflate_block(c, bsize, z->u + z->len, usize) != 0) {
        free(c);
        return -1;
    }
    free(c);
    if (z->nb == z->bcap) {
        z->bcap = z->bcap ? z->bcap * 2 : 64;
        z->b = xrealloc(z->b, sizeof *z->b * z->bcap);
    }
    z->b[z->nb].coff = z->off;
    z->b[z->nb].csize = bsize;
    z->b[z->nb].uoff = z->len;
    z->b[z->nb].usize = usize;
    z->nb++;
    z->len += usize;
    z->off += bsize;
    return 0;
}

static size_t hz_read(hz_t *z, void *dst, size_t n)
{
    while (z->len - z->pos < n)
        if (hz_block(z) != 0) break;
    size_t k = z->len - z->pos;
    if (k > n) k = n;
    memcpy(dst, z->u + z->pos, k);
    z->pos += k;
    return k;
}

static void hz_free(hz_t *z)
{
    free(z->u);
    free(z->b);
    memset(z, 0, sizeof *z);
}

static inline size_t looks_like_record(const uint8_t *u, size_t p, size_t L, int n_targets);

/* looks_like_record for a chain that may run out of inflated bytes: *step = the record's bytes when it is well-formed
 * and complete; returns 1 then, 0 when what is there rules a record out, -1 when only more data can tell (the fixed part
 * or the record's end lies beyond u[0, L)) */
static int32_t rd_i32(const uint8_t *p);
static uint32_t rd_u32(const uint8_t *p);
static int record_or_short(const uint8_t *u, size_t p, size_t L, int n_targets, size_t *step)
{
    if (p + 36 > L) return -1;
    const int32_t bl = rd_i32(u + p);
    if (bl < 32) return 0;
    const uint8_t *core = u + p + 4;
    const int32_t tid = rd_i32(core), pos = rd_i32(core + 4), l_qseq = rd_i32(core + 16), mtid = rd_i32(core + 20), mpos = rd_i32(core + 24);
    const uint32_t x1 = rd_u32(core + 8), x2 = rd_u32(core + 12);
    const size_t l_qname = x1 & 0xff, n_cigar = x2 & 0xffff;
    if (tid < -1 || tid >= n_targets || mtid < -1 || mtid >= n_targets || pos < -1 || mpos < -1 || l_qseq < 0 || l_qname == 0) return 0;
    const size_t need = l_qname + 4 * n_cigar + ((size_t)l_qseq + 1) / 2 + (size_t)l_qseq;
    if (need > (size_t)bl - 32) return 0;
    if (p + 36 + l_qname > L) return -1;
    if (u[p + 36 + l_qname - 1] != 0) return 0;
    if (p + 4 + (size_t)bl > L) return -1;
    *step = 4 + (size_t)bl;
    return 1;
}

/* Where a share of the file may begin: a point at or after compressed byte `at` where a BAM record starts, as (file offset
 * of a BGZF block, inflated bytes into that block — fewer than the block holds, size of that block). Two ranks that
 * call this with the same arguments get the same point, which is all that the split needs: any true record start will do.
 * It is a GUESS (a block header whose BSIZE chain leads to more headers; 8 well-formed records in a row): the rank whose
 * share ENDS there verifies it — its own chain of records, which starts at a known record start, has to arrive exactly
 * there — and the job falls back to one rank when a boundary does not hold.
 * Returns 1 with the point; 0 when the file ends without one (nothing starts behind `at`: the share before runs to the
 * end); -1 when the search had to be given up (no block header within reach, damaged blocks, or the chain of a candidate
 * still undecided after GIVE_UP inflated bytes — records of megabytes): the caller must not trust ANY share then. A chain
 * that runs out of inflated bytes is never a reason to reject a candidate: more blocks are inflated until it is decided
 * (with a fixed margin instead, true starts were rejected once eight records no longer fit into it — long reads — and a
 * boundary that fails on one side while a later one holds made two ranks count the same records). */
static int find_split(int fd, size_t fsize, size_t at, int n_targets, size_t *pB, size_t *pc, size_t *pcsize)
{
    if (at >= fsize) return 0;
    const size_t span = fsize - at < (4u << 16) + 64 ? fsize - at : (4u << 16) + 64;
    uint8_t *buf = xmalloc(span + 32);
    memset(buf, 0, span + 32);
    size_t have = 0;
    while (have < span) {
        const ssize_t k = pread(fd, buf + have, span - have, (off_t)(at + have));
        if (k <= 0) break;
        have += (size_t)k;
    }
    size_t B = SIZE_MAX;
    for (size_t p = 0; p + 18 <= have && p <= (1u << 16); p++) {
        if (!bgzf_header_ok(buf + p)) continue;
        size_t q = p;
        int ok = 1;
        for (int k = 0; k < 3; k++) {
            if (at + q == fsize) break;                              /* the chain ends with the file */
            if (q + 18 > have || !bgzf_header_ok(buf + q)) {
                ok = 0;
                break;
            }
            const size_t bsize = (size_t)(buf[q + 16] | buf[q + 17] << 8) + 1;
            if (bsize < 26 || at + q + bsize > fsize) {
                ok = 0;
                break;
            }
            q += bsize;
        }
        if (ok) {
            B = at + p;
            break;
        }
    }
    free(buf);
    if (B == SIZE_MAX) return at + have >= fsize && have <= (1u << 16) ? 0 : -1;     /* the last bytes of a file hold no block start: fine; no header in 64 KiB: not a BGZF file here */
    enum { K = 8, GIVE_UP = 64u << 20 };
    hz_t z;
    memset(&z, 0, sizeof z);
    z.fd = fd;
    z.off = B;
    z.size = fsize;
    size_t c = 0, found = SIZE_MAX;
    int at_end = 0, gave_up = 0;
    while (found == SIZE_MAX) {
        if (!at_end) {
            int st = 0;
            for (int i = 0; i < 4 && st == 0; i++) st = hz_block(&z);
            if (st < 0) {                                            /* a damaged block: nothing behind it can be vouched for */
                gave_up = 1;
                break;
            }
            at_end = st != 0;
        }
        int undecided = 0;
        for (; c < z.len && found == SIZE_MAX; c++) {
            size_t a = c;
            int k = 0, why = 1;
            while (k < K) {
                size_t step = 0;
                why = record_or_short(z.u, a, z.len, n_targets, &step);
                if (why != 1) break;
                a += step;
                k++;
            }
            if (k == K || (k > 0 && at_end && a == z.len)) {
                found = c;
            } else if (why < 0 && !at_end) {
                undecided = 1;                                       /* this candidate needs more bytes: come back to it */
                break;
            }
        }
        if (found != SIZE_MAX) break;
        if (at_end) break;                                           /* every byte looked at: no record starts here */
        if (undecided && z.len > GIVE_UP) {
            gave_up = 1;
            break;
        }
    }
    int ok = gave_up ? -1 : 0;
    if (found != SIZE_MAX) {
        /* as (block, offset inside it): the block that holds that byte */
        for (size_t i = 0; i < z.nb; i++)
            if (z.b[i].usize && found >= z.b[i].uoff && found < z.b[i].uoff + z.b[i].usize) {
                *pB = z.b[i].coff;
                *pc = found - z.b[i].uoff;
                *pcsize = z.b[i].csize;
                ok = 1;
                break;
            }
    }
    hz_free(&z);
    return ok;
}


/* Reads the next chunk of compressed blocks and inflates them in parallel into *pbuf at offset `at` (the buffer is
 * grown as needed, bytes before `at` are kept). Returns the number of bytes inflated; *peof is set when there is no
 * more input (end of file or a damaged block). Touches only the compressed-side state of the reader. */
/* ---- raw read-ahead: the file is read by its own thread, one chunk ahead, into the spare raw buffer behind RAW_HEAD
 * bytes of room for the incomplete block the indexer leaves over (less than one BGZF block). */
#define RAW_HEAD (BGZF_MAX + 64)
#define RAW_STEP (CHUNK_COMPRESSED < 256 ? (size_t)256 : CHUNK_COMPRESSED)      /* bytes per fread */
/* ... of a regular file with `left` bytes to go: no more than those (the buffers are page-locked when the device decodes, which
 * takes its time: a small file gets small ones) */
size_t aln_raw_step(size_t left)
{
    const size_t full = RAW_STEP, need = ((left + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1)) + ((size_t)1 << 20);
    return need < full ? need : full;
}
/* how many chunks the reader may be ahead of the one raw_next hands out next: from a regular file as many as there are
 * buffers beyond the one being indexed and the one whose pushes may still be copying (the reader used to wait for raw_next
 * after every chunk: while the producer sat in a push the file was not read, and then the producer waited for the file —
 * a quarter of the record loop of a BAM the device takes at the rate the page cache delivers it); else one */
static long io_ahead(const aln_reader *r) { return r->n_raw == N_RAW_DEVICE_FILE ? r->n_raw - 3 : 0; }

static void *io_main(void *arg)
{
    aln_reader *r = arg;
    pthread_mutex_lock(&r->io_mu);
    for (;;) {
        /* chunk io_fill goes into buffer io_fill % n_raw, last used by chunk io_fill - n_raw: raw_next has long moved on from it */
        while (!r->io_stop && (r->io_eof_seen || r->io_fill - r->io_take > io_ahead(r))) pthread_cond_wait(&r->io_cv, &r->io_mu);
        if (r->io_stop) break;
        const int b = (int)(r->io_fill % r->n_raw);
        const uint32_t lanes = r->raw_lanes[b];
        r->raw_lanes[b] = 0;
        pthread_mutex_unlock(&r->io_mu);
        /* ... and the pushes cut from it have copied their bytes (all of them were begun before the chunk after it was taken) */
        for (int sl = 0; sl < ITX_BAMWIN_LANES; sl++)
            if (lanes >> sl & 1u) DEV_CHK(dev.push_copied(dev.ctx, sl), "push_copied");
        if (!r->craw[b]) r->craw[b] = buf_alloc(RAW_HEAD + r->raw_step + 64);       /* page-locking a buffer takes a while: not under the lock */
        uint8_t *dst = r->craw[b] + RAW_HEAD;
        size_t got;
        if (r->io_fd >= 0) {
            /* a regular file: the step is read as eight slices at once (one thread copying out of the page cache delivers
             * about 7 GB/s; the device takes a BAM of real content at the 55 GB/s of its PCIe link) */
            const size_t step = r->raw_step;
            size_t want = r->io_size > r->io_off ? r->io_size - r->io_off : 0;
            if (want > step) want = step;
            static int max_parts;                                   /* ITX_READ_PARTS (1..16) overrides */
            if (!max_parts) {
                const char *e = getenv("ITX_READ_PARTS");
                const int v = e ? atoi(e) : 0;
                max_parts = v >= 1 && v <= 16 ? v : 8;
            }
            const int parts = want >= (4u << 20) ? max_parts : 1;
            const size_t per = (want + (size_t)parts - 1) / (size_t)parts;
            size_t done[16] = {0};
#pragma omp parallel for num_threads(parts) schedule(static, 1)
            for (int q = 0; q < parts; q++) {
                const size_t lo = (size_t)q * per, hi = lo + per < want ? lo + per : want;
                size_t at = lo;
                while (at < hi) {
                    const ssize_t k = pread(r->io_fd, dst + at, hi - at, (off_t)(r->io_off + at));
                    if (k <= 0) break;                             /* end of file or an error: what came before counts */
                    at += (size_t)k;
                }
                done[q] = at - lo;
            }
            got = 0;
            for (int q = 0; q < parts; q++) {
                got += done[q];
                if (done[q] < ((size_t)q * per + per < want ? per : want - (size_t)q * per)) break;     /* a short slice ends the stream there */
            }
            r->io_off += got;
        } else {
            got = fread(dst, 1, r->raw_step, r->f);
        }
        pthread_mutex_lock(&r->io_mu);
        r->io_got[b] = got;
        if (got < r->raw_step) r->io_eof_seen = 1;                 /* a short read: end of file (or an error, same thing here) */
        r->io_fill++;
        pthread_cond_broadcast(&r->io_cv);
    }
    pthread_mutex_unlock(&r->io_mu);
    return NULL;
}

/* Appends the next raw chunk behind the carried-over bytes; returns the bytes added (0 once the file is read out). */
static size_t raw_next(aln_reader *r)
{
    if (r->io_done) return 0;
    if (!r->io_on) {
        r->n_raw = r->dev ? N_RAW_DEVICE : 2;
        {
            struct stat sb;
            const int fd = fileno(r->f);
            r->io_fd = -1;
            if (fd >= 0 && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && ftello(r->f) == 0) {
                r->io_fd = fd;
                r->io_off = 0;
                r->io_size = (size_t)sb.st_size;
                if (r->rg_on) {
                    r->io_off = r->rg_lo_block;
                    /* the share ends inside the block at rg_end_block: nothing behind that block is this reader's */
                    if (r->rg_end_block != SIZE_MAX && r->rg_end_block + r->rg_end_csize < r->io_size) r->io_size = r->rg_end_block + r->rg_end_csize;
                }
            } else if (r->rg_on) {
                die("a share of %s was asked for, but it is not a regular file", "the alignment file");
            }
            r->io_abs_end = r->io_off;
            if (r->dev && r->io_fd >= 0 && dev.push_copied) r->n_raw = N_RAW_DEVICE_FILE;
        }
        r->raw_step = r->io_fd >= 0 ? aln_raw_step(r->io_size > r->io_off ? r->io_size - r->io_off : 0) : RAW_STEP;
        /* (Mapping the file and letting the device copy straight out of the page cache was tried twice and lost: through the
         * runtime's own path for unlocked memory the pipeline's copies go in staged pieces and hold the producer, 4.0 - 4.1 s per
         * run against 3.0 - 3.4 s; page-locking the mapping ahead with hipHostRegister costs 62 ms per GB — 3.5 s for the 56 GB
         * file, more when several threads do it — against the 11 core-seconds of pread it was meant to save:
         * profiles/r03_mmap_probe.txt, r03_cli_500M_hiseq_mmap*.json.) */
        pthread_mutex_init(&r->io_mu, NULL);
        pthread_cond_init(&r->io_cv, NULL);
        r->io_fill = r->io_take = 0;
        r->io_eof_seen = 0;
        r->io_stop = 0;
        if (pthread_create(&r->io_thread, NULL, io_main, r) != 0) die("cannot start the file read-ahead thread");
        r->io_on = 1;
    }
    pthread_mutex_lock(&r->io_mu);
    while (r->io_fill <= r->io_take) pthread_cond_wait(&r->io_cv, &r->io_mu);
    const int b = (int)(r->io_take % r->n_raw);
    const size_t got = r->io_got[b];
    uint8_t *nb = r->craw[b];
    if (r->clen > RAW_HEAD) die("BGZF: %zu bytes left over by the block indexer", r->clen);    /* cannot happen: < one block */
    if (r->clen) memcpy(nb + RAW_HEAD - r->clen, r->cbuf, r->clen);
    r->cbuf = nb + RAW_HEAD - r->clen;
    r->clen += got;
    r->io_abs_end += got;
    r->raw_cur = b;
    r->io_take++;
    if (got < r->raw_step) r->io_done = 1;                         /* the reader has stopped behind this chunk */
    pthread_cond_broadcast(&r->io_cv);                             /* a buffer further back has become the reader's */
    pthread_mutex_unlock(&r->io_mu);
    return got;
}

/* The complete BGZF blocks of the raw bytes at hand (bgzf.c:401-411 header check, BSIZE, the ISIZE trailer) into r->blk;
 * *poff = compressed bytes they take, *putot = bytes they inflate to. At most max_blocks blocks / max_utot inflated bytes
 * are taken (*pcapped: more complete blocks may follow in what is at hand). */
static size_t index_blocks(aln_reader *r, size_t max_blocks, size_t max_utot, size_t *poff, size_t *putot, int *pdamaged, int *pcapped)
{
    *pcapped = 0;
    struct blk *bl = r->blk;
    size_t nb = 0, off = 0, utot = 0;
    int damaged = 0;
    while (off + 18 <= r->clen) {
        const uint8_t *h = r->cbuf + off;
        if (!bgzf_header_ok(h)) {
            damaged = 1;
            break;
        }
        const size_t bsize = (size_t)(h[16] | h[17] << 8) + 1;
        if (bsize < 26) {
            damaged = 1;
            break;
        }
        if (off + bsize > r->clen) break;                      /* incomplete: wait for more input */
        const size_t usize = rd_u32_at(h + bsize - 4);
        if (usize > BGZF_MAX) {
            damaged = 1;
            break;
        }
        if (nb == max_blocks || utot + usize > max_utot) {      /* enough for one go: the rest stays for the next */
            *pcapped = 1;
            break;
        }
        if (nb == r->blk_cap) {
            r->blk_cap = r->blk_cap ? r->blk_cap * 2 : 4096;
            bl = r->blk = xrealloc(r->blk, sizeof *bl * r->blk_cap);
        }
        bl[nb].coff = off;
        bl[nb].csize = bsize;
        bl[nb].uoff = utot;
        bl[nb].usize = usize;
        nb++;
        utot += usize;
        off += bsize;
    }
    *poff = off;
    *putot = utot;
    *pdamaged = damaged;
    return nb;
}

static size_t bgzf_inflate_chunk(aln_reader *r, uint8_t **pbuf, size_t *pcap, size_t at, int *peof)
{
    double tq = now_s();
    const size_t got = raw_next(r);
    t_io += now_s() - tq;
    if (r->clen == 0) {
        *peof = 1;
        return 0;
    }
    size_t off = 0, utot = 0;
    int damaged = 0;
    int capped = 0;
    const size_t nb = index_blocks(r, SIZE_MAX, SIZE_MAX, &off, &utot, &damaged, &capped);
    struct blk *bl = r->blk;
    if (at + utot + 64 > *pcap) {
        const size_t old_cap = *pcap;
        *pcap = (at + utot) * 5 / 4 + BGZF_MAX + 64;
        *pbuf = buf_grow(*pbuf, at < old_cap ? at : old_cap, *pcap);      /* the bytes before `at` are the caller's */
    }
    tq = now_s();
    int bad = 0;
    uint8_t *dst0 = *pbuf + at;
    const struct blk *blocks = bl;
    const uint8_t *cbase = r->cbuf;
#pragma omp parallel for schedule(dynamic, 16) reduction(| : bad)
    for (long i = 0; i < (long)nb; i++)
        if (blocks[i].usize && inflate_block(cbase + blocks[i].coff, blocks[i].csize, dst0 + blocks[i].uoff, blocks[i].usize) != 0) bad |= 1;
    t_inflate += now_s() - tq;
    if (bad) {
        /* a block that does not inflate ends the stream there, like bgzf_read returning an error (bgzf.c:471-521) */
        size_t ok = 0;
        for (size_t i = 0; i < nb; i++) {
            if (bl[i].usize && inflate_block(r->cbuf + bl[i].coff, bl[i].csize, dst0 + bl[i].uoff, bl[i].usize) != 0) break;
            ok = bl[i].uoff + bl[i].usize;
        }
        utot = ok;
        damaged = 1;
    }
    /* the incomplete tail block stays where it is; raw_next carries it over */
    r->cbuf += off;
    r->clen -= off;
    if (damaged || (got == 0 && nb == 0)) *peof = 1;
    return utot;
}


static void pf_request(aln_reader *r);
static long pushes_in_flight(void);

/* ---- the device decoder (aln_use_device): this side only moves compressed bytes in ------------------------------------
 * Chunk k of the file is pushed into window k % ITX_BAMWIN_WINDOWS of the device on lane k % pushes_in_flight() (a lane = a
 * stream with its own scratch): that many pushes are kept in flight, so the Huffman pass of one chunk — a lane per block, latency-bound, most of the chip idle —
 * runs beside the token replay of the chunk before it. A block the device decoder flags is given to zlib here, whose verdict
 * is the reference's: inflated after all, its bytes are patched in; not inflatable, the stream ends in front of it
 * (bgzf.c:471-521). */
static void dev_begin(aln_reader *r)
{
    /* a window takes at most this much (extremely compressible input inflates a chunk to many gigabytes; the device side
     * counts in 32 bits and keeps a quarter megabyte of scratch per block) */
    enum { DEV_MAX_BLOCKS = 16384, DEV_MAX_BYTES = 1u << 30 };
    static size_t max_blocks;
    if (!max_blocks) {
        const char *e = getenv("ITX_DEV_WINDOW_BLOCKS");              /* tests: windows of a few blocks */
        const long x = e ? atol(e) : 0;
        max_blocks = x >= 1 ? (size_t)x : dev.max_blocks ? dev.max_blocks : DEV_MAX_BLOCKS;
        if (dev.max_blocks && max_blocks > dev.max_blocks) max_blocks = dev.max_blocks;
    }
    const size_t max_bytes = dev.max_bytes ? dev.max_bytes : DEV_MAX_BYTES;
    const long k = r->dk_begin;
    struct dev_job *j = &r->dj[k % pushes_in_flight()];
    double tq = now_s();
    size_t got = 1;
    if (!r->dmore) got = raw_next(r);                              /* complete blocks of the last raw chunk are still waiting */
    t_io += now_s() - tq;
    size_t off = 0, utot = 0;
    int damaged = 0, capped = 0;
    const size_t nb = r->clen ? index_blocks(r, max_blocks, max_bytes, &off, &utot, &damaged, &capped) : 0;
    r->dmore = capped;
    if (j->bl_cap < nb + 1) {
        j->bl_cap = nb + nb / 4 + 1;
        j->bl = xrealloc(j->bl, sizeof *j->bl * j->bl_cap);
    }
    if (r->dblk_cap < nb + 1) {
        r->dblk_cap = nb + nb / 4 + 1;
        r->dblk = xrealloc(r->dblk, sizeof *r->dblk * r->dblk_cap);
        r->dstatus = xrealloc(r->dstatus, r->dblk_cap);
    }
    size_t nb_use = nb;
    j->trunc = SIZE_MAX;
    int share_done = 0;
    if (r->rg_on && r->rg_end_block != SIZE_MAX && !r->rg_stop_hit) {
        /* where the share ends: rg_end_off inflated bytes into the block that starts at file offset rg_end_block. The
         * record before that point ends exactly there when the boundary is a true record start (checked once the last
         * window is parsed), so the window is cut there and nothing behind is decoded. */
        const size_t abs0 = r->io_abs_end - r->clen;
        for (size_t i = 0; i < nb; i++) {
            const size_t at = abs0 + r->blk[i].coff;
            if (at == r->rg_end_block) {
                r->rg_stop_hit = 1;
                j->trunc = r->blk[i].uoff + r->rg_end_off;
                if (j->trunc > r->blk[i].uoff + r->blk[i].usize) {      /* the boundary lies behind the block's bytes: not this stream's */
                    r->rg_stop_missed = 1;
                    j->trunc = r->blk[i].uoff;
                }
                nb_use = i + 1;
                share_done = 1;
                break;
            }
            if (at > r->rg_end_block) {                                 /* the chain of blocks steps over it: a false boundary */
                r->rg_stop_missed = 1;
                nb_use = i;
                j->trunc = r->blk[i].uoff;
                share_done = 1;
                break;
            }
        }
        if (share_done) {
            off = nb_use < nb ? r->blk[nb_use].coff : off;
            utot = nb_use ? r->blk[nb_use - 1].uoff + r->blk[nb_use - 1].usize : 0;
            r->dmore = 0;
        }
    }
    for (size_t i = 0; i < nb_use; i++) {
        j->bl[i] = r->blk[i];
        r->dblk[i].coff = (uint32_t)r->blk[i].coff;
        r->dblk[i].csize = (uint32_t)r->blk[i].csize;
        r->dblk[i].uoff = (uint32_t)r->blk[i].uoff;
        r->dblk[i].usize = (uint32_t)r->blk[i].usize;
    }
    r->dstream_total += utot;
    j->nb = nb_use;
    j->cbase = r->cbuf;
    j->cabs = r->io_fd >= 0 ? r->io_abs_end - r->clen : SIZE_MAX;
    if (r->n_raw == N_RAW_DEVICE_FILE && r->io_on) {
        pthread_mutex_lock(&r->io_mu);
        r->raw_lanes[r->raw_cur] |= 1u << (k % pushes_in_flight());
        pthread_mutex_unlock(&r->io_mu);
    }
    j->w = (int)(k % dev_nwin);
    j->last = damaged || share_done || (got == 0 && nb == 0);
    if (damaged && r->rg_on) r->rg_suspect = 1;
    static const uint8_t none[16];
    tq = now_s();
    DEV_CHK(dev.push_begin(dev.ctx, j->w, (int)(k % pushes_in_flight()), r->clen ? r->cbuf : none, off, r->dblk, nb_use), "push");
    t_inflate += now_s() - tq;
    r->cbuf += off;
    r->clen -= off;
    r->dk_begin = k + 1;
}

/* waits for the oldest push in flight and settles its flagged blocks; the caller publishes dk_ready */
static void dev_end(aln_reader *r)
{
    const long k = r->dk_ready;
    struct dev_job *j = &r->dj[k % pushes_in_flight()];
    size_t n_new = 0;
    const double tq = now_s();
    DEV_CHK(dev.push_end(dev.ctx, (int)(k % pushes_in_flight()), r->dstatus, &n_new), "push");
    int damaged = 0;
    for (size_t i = 0; i < j->nb; i++)
        if (r->dstatus[i]) {
            uint8_t tmp[BGZF_MAX + 8], again[BGZF_MAX + 64];
            const uint8_t *cb = j->cbase + j->bl[i].coff;
            if (r->n_raw == N_RAW_DEVICE_FILE) {
                /* the compressed bytes may have left their buffer by now: from the file once more */
                cb = again;
                if (j->bl[i].csize > sizeof again || pread(r->io_fd, again, j->bl[i].csize, (off_t)(j->cabs + j->bl[i].coff)) != (ssize_t)j->bl[i].csize)
                    die("cannot read the BGZF block at offset %zu again", j->cabs + j->bl[i].coff);
            }
            if (inflate_block(cb, j->bl[i].csize, tmp, j->bl[i].usize) == 0) {
                fprintf(stderr, "[iteres] note: BGZF block at chunk offset %zu declined by the device decoder (code %d), inflated by zlib\n", j->bl[i].coff,
                        r->dstatus[i]);
                DEV_CHK(dev.patch(dev.ctx, j->w, j->bl[i].uoff, tmp, j->bl[i].usize), "patch");
            } else {
                DEV_CHK(dev.truncate(dev.ctx, j->w, j->bl[i].uoff), "truncate");
                n_new = j->bl[i].uoff;
                damaged = 1;
                break;
            }
        }
    if (j->trunc != SIZE_MAX && !damaged && j->trunc <= n_new) {
        DEV_CHK(dev.truncate(dev.ctx, j->w, j->trunc), "truncate");
        n_new = j->trunc;
    }
    t_inflate += now_s() - tq;
    r->dring_n[j->w] = n_new;
    r->dring_eof[j->w] = j->last || damaged;
    if (damaged && r->rg_on) r->rg_suspect = 1;
}

/* pushes kept in flight: pass 1 of the device decoder is a lane per block and latency-bound (a chunk's 8 k blocks take the
 * same ~9 ms as 1 k would), so several chunks' pass 1 run side by side on a chip that one of them fills to a tenth;
 * ITX_PUSHES (1 .. ITX_BAMWIN_LANES) overrides */
static long pushes_in_flight(void)
{
    static long v;
    if (!v) {
        const char *e = getenv("ITX_PUSHES");
        const long x = e ? atol(e) : 0;
        v = x >= 1 && x <= ITX_BAMWIN_LANES ? x : ITX_BAMWIN_LANES_DEFAULT;
    }
    return v;
}

/* The producer (read-ahead thread): begin a push whenever a window and a lane are free and input is left, end the oldest
 * one otherwise. */
static void *dev_producer(void *arg)
{
    aln_reader *r = arg;
    pthread_mutex_lock(&r->pf_mu);
    for (;;) {
        int act = 0;                                               /* 1 begin, 2 end */
        while (!r->pf_stop) {
            const long begun = r->dk_begin, ready = r->dk_ready, cur = r->dk_cur;
            if (!r->dinput_done && begun - cur < dev_nwin && begun - ready < pushes_in_flight()) {
                act = 1;
                break;
            }
            if (begun > ready) {
                act = 2;
                break;
            }
            pthread_cond_wait(&r->pf_cv, &r->pf_mu);
        }
        if (r->pf_stop) break;
        pthread_mutex_unlock(&r->pf_mu);
        if (act == 1) {
            dev_begin(r);
            pthread_mutex_lock(&r->pf_mu);
            if (r->dj[(r->dk_begin - 1) % pushes_in_flight()].last) r->dinput_done = 1;
        } else {
            dev_end(r);
            pthread_mutex_lock(&r->pf_mu);
            if (r->dring_eof[r->dk_ready % dev_nwin]) r->dinput_done = 1;
            r->dk_ready++;
            pthread_cond_broadcast(&r->pf_cv);
        }
    }
    pthread_mutex_unlock(&r->pf_mu);
    return NULL;
}

/* The next window becomes the current one: the unconsumed tail (a partial record, or header bytes) moves in front of its
 * fresh bytes. Synchronous while the header is read, fed by the producer afterwards. Returns bytes added. */
static size_t dev_advance(aln_reader *r)
{
    if (r->eof) return 0;
    const long nxt = r->dk_cur + 1;
    if (!r->pf_on) {
        if (r->dk_begin <= nxt) dev_begin(r);
        while (r->dk_ready <= nxt) {
            dev_end(r);
            r->dk_ready++;
        }
    } else {
        pthread_mutex_lock(&r->pf_mu);
        while (r->dk_ready <= nxt) pthread_cond_wait(&r->pf_cv, &r->pf_mu);
        pthread_mutex_unlock(&r->pf_mu);
    }
    const int w = (int)(nxt % dev_nwin);
    if (r->dk_cur >= 0) {
        if (r->rg_on && r->rg_lo_block) {
            /* a share that starts at a guessed record start: a record "too long to carry" is what a false start looks
             * like — the share ends here, unverified, and the job is done again by one rank */
            if (dev.carry(dev.ctx, r->dw, w) != 0) {
                r->rg_suspect = 1;
                r->eof = 1;
                r->dlast = 1;
                if (r->pf_on) pthread_mutex_lock(&r->pf_mu);
                r->dk_cur = nxt;
                if (r->pf_on) {
                    pthread_cond_broadcast(&r->pf_cv);
                    pthread_mutex_unlock(&r->pf_mu);
                }
                r->dparsed = 1;
                r->dn_rec = r->drec_next = 0;
                return 0;
            }
        } else {
            DEV_CHK(dev.carry(dev.ctx, r->dw, w), "carry");
        }
    }
    if (r->pf_on) pthread_mutex_lock(&r->pf_mu);
    r->dk_cur = nxt;
    if (r->pf_on) {
        pthread_cond_broadcast(&r->pf_cv);                         /* the window left behind is free */
        pthread_mutex_unlock(&r->pf_mu);
    }
    r->dw = w;
    r->dparsed = 0;
    r->dn_rec = r->drec_next = 0;
    if (r->dring_eof[w]) r->eof = 1;
    if (r->dskip_left) {                                           /* a share starts rg_skip bytes into its first block */
        size_t av = 0;
        DEV_CHK(dev.avail(dev.ctx, w, &av), "avail");
        const size_t sk = av < r->dskip_left ? av : r->dskip_left;
        DEV_CHK(dev.skip(dev.ctx, w, sk), "skip");
        r->dskip_left -= sk;
    }
    return r->dring_n[w];
}

/* sequential read of n bytes from the device window's front (header parsing): through a host copy fetched 1 MiB at a time */
static size_t dev_read(aln_reader *r, void *dst, size_t n)
{
    for (;;) {
        if (r->hdr_len - r->hdr_pos >= n) {
            memcpy(dst, r->hdr + r->hdr_pos, n);
            r->hdr_pos += n;
            return n;
        }
        size_t avail = 0;
        DEV_CHK(dev.avail(dev.ctx, r->dw, &avail), "avail");
        if (avail > r->hdr_len) {                                  /* more of this window */
            size_t want = r->hdr_pos + n > r->hdr_len + (1u << 20) ? r->hdr_pos + n : r->hdr_len + (1u << 20);
            if (want > avail) want = avail;
            r->hdr = xrealloc(r->hdr, want);
            DEV_CHK(dev.peek(dev.ctx, r->dw, r->hdr_len, r->hdr + r->hdr_len, want - r->hdr_len), "peek");
            r->hdr_len = want;
            continue;
        }
        /* the window is used up: what was read is consumed, the rest travels to the next window */
        if (r->eof) {
            const size_t k = r->hdr_len - r->hdr_pos;
            memcpy(dst, r->hdr + r->hdr_pos, k);
            r->hdr_pos += k;
            return k;
        }
        DEV_CHK(dev.skip(dev.ctx, r->dw, r->hdr_pos), "skip");
        r->hdr_len = r->hdr_pos = 0;
        dev_advance(r);
    }
}

/* ---- read-ahead: while the caller hops over and parses one chunk, a loader thread reads and inflates the next one
 * into the spare buffer, behind PF_HEAD bytes of room for the caller's unconsumed tail (a partial record). */
#define PF_HEAD (4u << 20)
static void *pf_main(void *arg)
{
    aln_reader *r = arg;
    if (r->dev) return dev_producer(r);
    pthread_mutex_lock(&r->pf_mu);
    for (;;) {
        while (r->pf_state != 1 && !r->pf_stop) pthread_cond_wait(&r->pf_cv, &r->pf_mu);
        if (r->pf_stop) break;
        pthread_mutex_unlock(&r->pf_mu);
        int eof = 0;
        const size_t n = bgzf_inflate_chunk(r, &r->nbuf, &r->ncap, PF_HEAD, &eof);
        pthread_mutex_lock(&r->pf_mu);
        r->nlen = n;
        r->n_eof = eof;
        r->pf_state = 2;
        pthread_cond_broadcast(&r->pf_cv);
    }
    pthread_mutex_unlock(&r->pf_mu);
    return NULL;
}

static void pf_request(aln_reader *r)
{
    pthread_mutex_lock(&r->pf_mu);
    r->pf_state = 1;
    pthread_cond_broadcast(&r->pf_cv);
    pthread_mutex_unlock(&r->pf_mu);
}

static void pf_start(aln_reader *r)
{
    pthread_mutex_init(&r->pf_mu, NULL);
    pthread_cond_init(&r->pf_cv, NULL);
    r->pf_state = 0;
    r->pf_stop = 0;
    if (pthread_create(&r->pf_thread, NULL, pf_main, r) != 0) die("cannot start the BAM read-ahead thread");
    r->pf_on = 1;
    if (!r->eof && !r->dev) pf_request(r);                        /* the device producer drives itself */
}

/* Makes more inflated bytes available behind the unconsumed ones. Returns the number of bytes added (0 at end of
 * input). Synchronous while the header is read, from the read-ahead thread afterwards. */
static size_t bgzf_load_chunk(aln_reader *r)
{
    if (r->eof) return 0;
    if (!r->pf_on) {
        /* keep what the record parser has not consumed */
        if (r->upos) {
            memmove(r->ubuf, r->ubuf + r->upos, r->ulen - r->upos);
            r->ulen -= r->upos;
            r->upos = 0;
        }
        int eof = 0;
        const size_t n = bgzf_inflate_chunk(r, &r->ubuf, &r->ucap, r->ulen, &eof);
        r->ulen += n;
        if (eof) r->eof = 1;
        return n;
    }
    pthread_mutex_lock(&r->pf_mu);
    while (r->pf_state != 2) pthread_cond_wait(&r->pf_cv, &r->pf_mu);
    r->pf_state = 0;
    pthread_mutex_unlock(&r->pf_mu);
    const size_t tail = r->ulen - r->upos, n = r->nlen;
    if (tail > PF_HEAD) {                                      /* a record of more than 4 MB: make room the slow way */
        uint8_t *big = buf_alloc(tail + n + 64);
        memcpy(big, r->ubuf + r->upos, tail);
        memcpy(big + tail, r->nbuf + PF_HEAD, n);
        buf_free(r->ubuf);
        r->ubuf = big;
        r->ucap = tail + n + 64;
        r->upos = 0;
        r->ulen = tail + n;
    } else {
        memcpy(r->nbuf + PF_HEAD - tail, r->ubuf + r->upos, tail);
        uint8_t *tb = r->ubuf;
        r->ubuf = r->nbuf;
        r->nbuf = tb;
        const size_t tc = r->ucap;
        r->ucap = r->ncap;
        r->ncap = tc;
        r->upos = PF_HEAD - tail;
        r->ulen = PF_HEAD + n;
    }
    if (r->n_eof)
        r->eof = 1;
    else
        pf_request(r);
    return n;
}

/* sequential read of n bytes (header parsing) */
static size_t bgzf_read(aln_reader *r, void *dst, size_t n)
{
    if (r->hz) return hz_read((hz_t *)r->hz, dst, n);
    if (r->dev) return dev_read(r, dst, n);
    while (r->ulen - r->upos < n) {
        if (bgzf_load_chunk(r) == 0 && r->eof) break;
    }
    size_t k = r->ulen - r->upos;
    if (k > n) k = n;
    memcpy(dst, r->ubuf + r->upos, k);
    r->upos += k;
    return k;
}


static char *xstrndup_bound(const char *s, size_t max)
{
    const size_t k = strnlen(s, max);
    char *d = xmalloc(k + 1);
    memcpy(d, s, k);
    d[k] = 0;
    return d;
}

static int32_t rd_i32(const uint8_t *p) { return (int32_t)((uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24); }
static uint32_t rd_u32(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }

static void add_target(aln_reader *r, const char *name)
{
    r->tname = xrealloc(r->tname, sizeof(char *) * (size_t)(r->n_targets + 1));
    r->tname[r->n_targets++] = xstrdup(name);
}

static int bam_read_header(aln_reader *r)
{
    uint8_t b[8];
    if (bgzf_read(r, b, 4) != 4 || memcmp(b, "BAM\1", 4) != 0) {
        fprintf(stderr, "[bam_header_read] invalid BAM binary header (this is not a BAM file).\n");
        return -1;
    }
    if (bgzf_read(r, b, 4) != 4) return -1;
    int32_t l_text = rd_i32(b);
    while (l_text > 0) {                     /* the text header is not needed: tids come from the binary list */
        uint8_t skip[4096];
        size_t k = (size_t)l_text < sizeof skip ? (size_t)l_text : sizeof skip;
        if (bgzf_read(r, skip, k) != k) return -1;
        l_text -= (int32_t)k;
    }
    if (bgzf_read(r, b, 4) != 4) return -1;
    const int32_t n_ref = rd_i32(b);
    for (int32_t i = 0; i < n_ref; i++) {
        if (bgzf_read(r, b, 4) != 4) return -1;
        const int32_t l_name = rd_i32(b);
        if (l_name <= 0 || l_name > 1 << 20) return -1;
        char *nm = xmalloc((size_t)l_name + 1);
        if (bgzf_read(r, nm, (size_t)l_name) != (size_t)l_name) {
            free(nm);
            return -1;
        }
        nm[l_name] = 0;
        add_target(r, nm);
        free(nm);
        if (bgzf_read(r, b, 4) != 4) return -1;     /* l_ref */
    }
    return 0;
}

/* ---- SAM text header ------------------------------------------------------------------------------------- */
static int sam_read_header(aln_reader *r)
{
    r->pending_len = -1;
    ssize_t len;
    while ((len = getline(&r->line, &r->line_cap, r->f)) >= 0) {
        r->n_lines++;
        if (r->line[0] != '@') {
            r->pending_len = len;
            break;
        }
        if (strncmp(r->line, "@SQ", 3) == 0) {
            char *sn = strstr(r->line, "\tSN:");
            if (sn) {
                sn += 4;
                size_t k = strcspn(sn, "\t\r\n");
                char *nm = xmalloc(k + 1);
                memcpy(nm, sn, k);
                nm[k] = 0;
                add_target(r, nm);
                free(nm);
            }
        }
    }
    return 0;
}

aln_reader *aln_open(const char *path, int is_sam)
{
    FILE *f = fopen(path, is_sam ? "r" : "rb");
    if (!f) return NULL;
    ld_probe();
    aln_reader *r = xcalloc(1, sizeof *r);
    r->f = f;
    r->is_sam = is_sam;
    r->pending_len = -1;
    names_init(&r->tnames);
    int rc;
    if (is_sam) {
        rc = sam_read_header(r);
    } else {
        r->dev = dev.push_begin != NULL;
        r->dk_cur = -1;
        if (r->dev)
            for (int w = 0; w < ITX_BAMWIN_WINDOWS; w++) {         /* whatever an earlier file left unconsumed is not this file's */
                size_t left = 0;
                DEV_CHK(dev.avail(dev.ctx, w, &left), "avail");
                DEV_CHK(dev.skip(dev.ctx, w, left), "skip");
            }
        struct stat sb;
        if (r->dev && !getenv("ITX_HEADER_BY_DEVICE") && fstat(fileno(f), &sb) == 0 && S_ISREG(sb.st_mode)) {
            /* a regular file: the reference list by the plain host reader (a few blocks, milliseconds), and the device's stream
             * skips that many bytes when its first window arrives. Reading the header through the device meant one whole push —
             * a chunk read, copied and decoded while nothing else ran (0.19 s) — before the pipeline of pushes could start. */
            hz_t z;
            memset(&z, 0, sizeof z);
            z.fd = fileno(f);
            z.size = (size_t)sb.st_size;
            r->hz = &z;
            rc = bam_read_header(r);
            r->hz = NULL;
            r->dskip_left = z.pos;
            hz_free(&z);
            r->dparsed = 1;                                        /* nothing to parse before the first window is in */
        } else {
            rc = bam_read_header(r);
            if (r->dev && rc == 0) {
                DEV_CHK(dev.skip(dev.ctx, r->dw, r->hdr_pos), "skip");             /* the records start here */
                free(r->hdr);
                r->hdr = NULL;
                r->hdr_len = r->hdr_pos = 0;
            }
        }
    }
    if (rc != 0) {
        aln_close(r);
        return NULL;
    }
    for (int i = 0; i < r->n_targets; i++) names_intern(&r->tnames, r->tname[i]);    /* first occurrence wins a lookup */
    return r;
}

/* A share of a BAM file for one rank of a multi-GPU job: the records that START in the compressed byte range [lo, hi) of
 * the file — more exactly between the split points find_split gives for lo and for hi (hi = SIZE_MAX: up to the end).
 * Device decoder only. The header is read like any reader's (every rank needs the reference list). */
aln_reader *aln_open_range(const char *path, size_t lo, size_t hi)
{
    if (lo == 0 && hi == SIZE_MAX) return aln_open(path, 0);
    if (!dev.push_begin) die("a share of %s was asked for without the device decoder", path);
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    struct stat sb;
    if (fstat(fileno(f), &sb) != 0 || !S_ISREG(sb.st_mode)) {
        fclose(f);
        return NULL;
    }
    const size_t fsize = (size_t)sb.st_size;
    ld_probe();
    aln_reader *r = xcalloc(1, sizeof *r);
    r->f = f;
    r->pending_len = -1;
    names_init(&r->tnames);
    r->dev = 1;
    r->dk_cur = -1;
    r->rg_on = 1;
    r->rg_end_block = SIZE_MAX;
    for (int w = 0; w < ITX_BAMWIN_WINDOWS; w++) {
        size_t left = 0;
        DEV_CHK(dev.avail(dev.ctx, w, &left), "avail");
        DEV_CHK(dev.skip(dev.ctx, w, left), "skip");
    }
    /* the reference list, by the plain host reader (a few blocks) */
    hz_t z;
    memset(&z, 0, sizeof z);
    z.fd = fileno(f);
    z.size = fsize;
    r->hz = &z;
    int rc = bam_read_header(r);
    r->hz = NULL;
    hz_free(&z);
    if (rc != 0) {
        aln_close(r);
        return NULL;
    }
    int empty = 0;
    size_t B = 0, c = 0, cs = 0;
    if (hi != SIZE_MAX && hi < fsize) {
        const int fs = find_split(fileno(f), fsize, hi, r->n_targets, &B, &c, &cs);
        if (fs > 0) {
            r->rg_end_block = B;
            r->rg_end_off = c;
            r->rg_end_csize = cs;
        } else if (fs < 0) {
            r->rg_suspect = 1;                                     /* the search was given up: no share of this job can be trusted (whole-job fallback) */
        }
    }
    if (lo > 0) {
        const int fs = find_split(fileno(f), fsize, lo, r->n_targets, &B, &c, &cs);
        if (fs <= 0) {
            empty = 1;                                             /* 0: no record starts behind lo, the share before this one runs to the end */
            if (fs < 0) r->rg_suspect = 1;                         /* given up: see above */
        } else {
            if (r->rg_end_block != SIZE_MAX && (B > r->rg_end_block || (B == r->rg_end_block && c >= r->rg_end_off))) empty = 1;
            r->rg_lo_block = B;
            r->dskip_left = c;
            r->dparsed = 1;                                        /* nothing to parse before the first window is in */
        }
    } else {
        /* from the start of the file: the header goes by once more, in front of the records, through the device's windows */
        for (int i = 0; i < r->n_targets; i++) free(r->tname[i]);
        free(r->tname);
        r->tname = NULL;
        r->n_targets = 0;
        if (bam_read_header(r) != 0) {
            aln_close(r);
            return NULL;
        }
        DEV_CHK(dev.skip(dev.ctx, r->dw, r->hdr_pos), "skip");
        free(r->hdr);
        r->hdr = NULL;
        r->hdr_len = r->hdr_pos = 0;
    }
    if (empty) {
        r->eof = 1;
        r->dlast = 1;
        r->dparsed = 1;
        r->rg_verified = 1;
    }
    for (int i = 0; i < r->n_targets; i++) names_intern(&r->tnames, r->tname[i]);
    return r;
}

/* find_split for callers outside this file (the CPU test tool): the reference list is read first, like aln_open_range does */
int aln_find_split(const char *path, size_t at, size_t *block, size_t *off, size_t *csize)
{
    FILE *f = fopen(path, "rb");
    struct stat sb;
    if (!f || fstat(fileno(f), &sb) != 0) return -1;
    ld_probe();
    aln_reader *r = xcalloc(1, sizeof *r);
    r->f = f;
    r->pending_len = -1;
    names_init(&r->tnames);
    hz_t z;
    memset(&z, 0, sizeof z);
    z.fd = fileno(f);
    z.size = (size_t)sb.st_size;
    r->hz = &z;
    const int rc = bam_read_header(r);
    r->hz = NULL;
    hz_free(&z);
    int found = -1;
    if (rc == 0) found = find_split(fileno(f), (size_t)sb.st_size, at, r->n_targets, block, off, csize);     /* 1 found, 0 none up to the end, -1 given up */
    aln_close(r);
    return found;
}

/* after the last batch: 1 when the share's end boundary held (the record chain arrived exactly there) or it has none */
int aln_range_verified(const aln_reader *r)
{
    if (!r->rg_on) return 1;
    if (r->rg_suspect) return 0;
    if (r->rg_end_block == SIZE_MAX) return 1;
    return r->rg_verified == 1;
}

void aln_close(aln_reader *r)
{
    if (!r) return;
    if (getenv("ITX_TIMING") && !r->is_sam)
        fprintf(stderr, "[itx timing] BAM decode so far: file read %.3f s, inflate (%s) %.3f s, record hop %.3f s (%zu pieces, %zu walked again), parse %.3f s\n", t_io,
                r->dev ? "device" : ld_state == 1 ? "libdeflate" : "zlib", t_inflate, t_hop, r->hop_pieces, r->hop_redone + r->d_rewalked, t_parse);
    if (r->pf_on) {
        pthread_mutex_lock(&r->pf_mu);
        while (!r->dev && r->pf_state == 1) pthread_cond_wait(&r->pf_cv, &r->pf_mu);      /* let a chunk in flight land */
        r->pf_stop = 1;
        pthread_cond_broadcast(&r->pf_cv);
        pthread_mutex_unlock(&r->pf_mu);
        pthread_join(r->pf_thread, NULL);
    }
    if (r->dev)
        while (r->dk_ready < r->dk_begin) {                           /* pushes still in flight: the lanes must be idle for the next file */
            dev_end(r);
            r->dk_ready++;
        }
    if (r->io_on) {
        pthread_mutex_lock(&r->io_mu);
        r->io_stop = 1;                                                        /* (a read in flight lands first: the thread looks when it is back) */
        pthread_cond_broadcast(&r->io_cv);
        pthread_mutex_unlock(&r->io_mu);
        pthread_join(r->io_thread, NULL);
    }
    if (r->f) fclose(r->f);
    for (int k = 0; k < N_RAW_DEVICE; k++) buf_free(r->craw[k]);
    buf_free(r->nbuf);
    free(r->blk);
    free(r->dblk);
    free(r->dstatus);
    for (int k = 0; k < ITX_BAMWIN_LANES; k++) free(r->dj[k].bl);
    free(r->hdr);
    free(r->dseen);
    free(r->d_off);
    free(r->d_xa);
    free(r->d_raw);
    for (int i = 0; i < r->n_targets; i++) free(r->tname[i]);
    free(r->tname);
    names_free(&r->tnames);
    buf_free(r->ubuf);
    free(r->rec_off);
    free(r->spec);
    free(r->line);
    free(r);
}

int aln_n_targets(const aln_reader *r) { return r->n_targets; }
const char *aln_target_name(const aln_reader *r, int tid) { return r->tname[tid]; }

/* bam_aux.c:36-48 (the aux walk of bam_aux_get): pointer to the type byte of the first tag `t0 t1`, or NULL */
static const uint8_t *aux_find(const uint8_t *s, const uint8_t *end, char t0, char t1)
{
    while (s + 3 <= end) {
        const int hit = s[0] == (uint8_t)t0 && s[1] == (uint8_t)t1;
        const int type = toupper(s[2]);
        if (hit) return s + 2;
        s += 3;
        if (type == 'A' || type == 'C') s += 1;
        else if (type == 'S') s += 2;
        else if (type == 'I' || type == 'F') s += 4;
        else if (type == 'D') s += 8;
        else if (type == 'Z' || type == 'H') {
            while (s < end && *s) ++s;
            ++s;
        } else if (type == 'B') {
            if (s + 5 > end) return NULL;
            const int sub = toupper(s[0]);
            const uint32_t cnt = rd_u32(s + 1);
            const size_t esz = (sub == 'C' || sub == 'A') ? 1 : (sub == 'S') ? 2 : 4;
            s += 5 + (size_t)cnt * esz;
        } else
            return NULL;
    }
    return NULL;
}

/* bam_aux2i, bam_aux.c:159-170 */
static int32_t aux_to_int(const uint8_t *s, const uint8_t *end)
{
    if (!s) return 0;
    const int type = *s++;
    if (type == 'c' && s + 1 <= end) return (int32_t)(int8_t)s[0];
    if (type == 'C' && s + 1 <= end) return (int32_t)s[0];
    if (type == 's' && s + 2 <= end) return (int32_t)(int16_t)(s[0] | s[1] << 8);
    if (type == 'S' && s + 2 <= end) return (int32_t)(uint16_t)(s[0] | s[1] << 8);
    if ((type == 'i' || type == 'I') && s + 4 <= end) return rd_i32(s);
    return 0;
}

/* bam.c:179-210 for one record that is completely in memory */
static inline void bam_parse_one(const uint8_t *p, size_t n, itx_staging *st, aln_side *side, int *any_paired, int *aux_xa)
{
    const int32_t block_len = rd_i32(p);
    const uint8_t *core = p + 4, *data = p + 36;
    const size_t dlen = (size_t)block_len - 32;
    const int32_t tid = rd_i32(core), pos = rd_i32(core + 4);
    const uint32_t x1 = rd_u32(core + 8), x2 = rd_u32(core + 12);
    const uint32_t l_qname = x1 & 0xff, qual = (x1 >> 8) & 0xff, flag = x2 >> 16, n_cigar = x2 & 0xffff;
    const int32_t l_qseq = rd_i32(core + 16), mpos = rd_i32(core + 24), isize = rd_i32(core + 28);
    int32_t tmpend;
    if (n_cigar && (size_t)l_qname + 4 * (size_t)n_cigar <= dlen) {
        uint32_t e = (uint32_t)pos;                               /* bam.c:17-27 */
        const uint8_t *cg = data + l_qname;
        for (uint32_t k = 0; k < n_cigar; k++) {
            const uint32_t c = rd_u32(cg + 4 * k), op = c & 0xf;
            if (op == 0 || op == 2 || op == 3) e += c >> 4;       /* M, D, N */
        }
        tmpend = (int32_t)e;
    } else {
        tmpend = (int32_t)((uint32_t)pos + (uint32_t)l_qseq);     /* generic.c:820 */
    }
    st->tid[n] = tid;
    st->pos[n] = pos;
    st->tmpend[n] = tmpend;
    st->mapq[n] = (uint8_t)qual;
    st->flag5[n] = ITX_FLAG5(flag);
    st->mpos[n] = mpos;
    st->isize[n] = isize;
    if (flag & 1) *any_paired = 1;
    if (side && side->want_qnames) side->qname[n] = xstrdup(l_qname && l_qname <= dlen ? (const char *)data : "");
    const size_t ql = l_qseq > 0 ? (size_t)l_qseq : 0;
    const size_t off = (size_t)l_qname + 4 * (size_t)n_cigar + (ql + 1) / 2 + ql;
    const uint8_t *xa = off < dlen ? aux_find(data + off, data + dlen, 'X', 'A') : NULL;
    if (xa) *aux_xa = 1;
    if (side && side->want_aux) {
        side->xa[n] = NULL;
        side->nm[n] = 0;
        if (xa) {
            /* bam_aux2Z (bam_aux.c:193-199): the string of a Z/H tag; an XA tag of another type has no string the
             * reference could copy (it would crash there) and reads as empty here */
            const uint8_t *e = data + dlen;
            const char *z = (*xa == 'Z' || *xa == 'H') ? (const char *)(xa + 1) : "";
            side->xa[n] = xstrndup_bound(z, (*xa == 'Z' || *xa == 'H') ? (size_t)(e - (xa + 1)) : 0);
            side->nm[n] = aux_to_int(aux_find(data + off, e, 'N', 'M'), e);
        }
    }
}

/* ---- locating the records --------------------------------------------------------------------------------------
 * A BAM stream is a chain: a record's length field says where the next record starts (bam.c:179-210 reads them one by one).
 * Walking that chain is a pointer chase through hundreds of megabytes per chunk — one core, one cache miss per record. The
 * chunk is cut into pieces instead and every piece is walked by its own thread from a GUESSED start: the first offset in the
 * piece from which a few records in a row look like records. A guess proves nothing by itself; what makes the result exact
 * is the check afterwards, in stream order: piece 0 starts at a known record start, and a piece is accepted only if its
 * guess is precisely where the chain of the pieces before it arrives — otherwise that piece is walked again from there. */
typedef struct {
    size_t c, end, n, first;   /* guessed start (SIZE_MAX none), where the walk stopped, starts found, their place in spec */
    int why;                   /* 0 reached the piece's limit, 1 needs more input, 2 malformed length (bam.c:186-190) */
} hop_piece;

static size_t hop_run(const uint8_t *u, size_t p, size_t lim, size_t L, size_t *off, size_t *pn, int *why)
{
    size_t n = *pn;
    while (p < lim) {
        if (p + 4 > L) {
            *why = 1;
            *pn = n;
            return p;
        }
        __builtin_prefetch(u + p + 1024);
        __builtin_prefetch(u + p + 2048);
        const int32_t bl = rd_i32(u + p);
        if (bl < 32) {
            *why = 2;
            *pn = n;
            return p;
        }
        if (p + 4 + (size_t)bl > L) {
            *why = 1;
            *pn = n;
            return p;
        }
        off[n++] = p;
        p += 4 + (size_t)bl;
    }
    *why = 0;
    *pn = n;
    return p;
}

/* does a complete record that looks like one start at p? (only ever used to pick a guess) */
static inline size_t looks_like_record(const uint8_t *u, size_t p, size_t L, int n_targets)
{
    if (p + 36 > L) return 0;
    const int32_t bl = rd_i32(u + p);
    if (bl < 32 || p + 4 + (size_t)bl > L) return 0;
    const uint8_t *core = u + p + 4;
    const int32_t tid = rd_i32(core), pos = rd_i32(core + 4), l_qseq = rd_i32(core + 16), mtid = rd_i32(core + 20), mpos = rd_i32(core + 24);
    const uint32_t x1 = rd_u32(core + 8), x2 = rd_u32(core + 12);
    const size_t l_qname = x1 & 0xff, n_cigar = x2 & 0xffff;
    if (tid < -1 || tid >= n_targets || mtid < -1 || mtid >= n_targets || pos < -1 || mpos < -1 || l_qseq < 0 || l_qname == 0) return 0;
    const size_t need = l_qname + 4 * n_cigar + ((size_t)l_qseq + 1) / 2 + (size_t)l_qseq;
    if (need > (size_t)bl - 32) return 0;
    if (u[p + 36 + l_qname - 1] != 0) return 0;
    return 4 + (size_t)bl;
}

static size_t hop_piece_bytes(void)
{
    static size_t v;
    if (!v) {
        const char *e = getenv("ITX_HOP_PIECE");                  /* tests force tiny pieces: many guesses, many of them wrong */
        const long x = e ? atol(e) : 0;
        v = x >= 1 ? (size_t)x : (1u << 20);
    }
    return v;
}

/* Fills rec_off / n_rec with the starts of the complete records in ubuf[upos, ulen), exactly as the one-by-one walk would;
 * sets upos behind the last of them (when there is one), and eof / ulen at a malformed length. */
static void locate_records(aln_reader *r)
{
    const uint8_t *u = r->ubuf;
    const size_t p0 = r->upos, L = r->ulen;
    r->n_rec = r->rec_next = 0;
    if (p0 >= L) return;
    const size_t worst = (L - p0) / 36 + 2;                        /* a record is at least 36 bytes */
    const size_t piece = hop_piece_bytes();
    size_t T = (L - p0) / piece;
    const size_t tmax = (size_t)omp_get_max_threads() * 8;
    if (T > tmax) T = tmax;
    if (r->rec_cap < worst) {
        r->rec_cap = worst + worst / 4;
        r->rec_off = xrealloc(r->rec_off, sizeof(size_t) * r->rec_cap);
    }
    size_t p;
    int why = 0;
    if (T < 2) {
        p = hop_run(u, p0, L, L, r->rec_off, &r->n_rec, &why);
    } else {
        if (r->spec_cap < worst + T) {
            r->spec_cap = worst + worst / 4 + T;
            r->spec = xrealloc(r->spec, sizeof(size_t) * r->spec_cap);
        }
        hop_piece *pc = xcalloc(T, sizeof *pc);
        const size_t span = (L - p0) / T;
        const int n_targets = r->n_targets;
        size_t *spec = r->spec;
#pragma omp parallel for schedule(dynamic, 1)
        for (long t = 0; t < (long)T; t++) {
            const size_t b = p0 + (size_t)t * span, lim = (size_t)t + 1 < T ? p0 + ((size_t)t + 1) * span : L;
            hop_piece *q = &pc[t];
            q->first = (b - p0) / 36 + (size_t)t;                  /* room for every start this piece can hold */
            q->c = SIZE_MAX;
            q->n = 0;
            if (t == 0) {
                q->c = b;
            } else {
                for (size_t c = b; c < lim; c++) {
                    size_t a = c, k = 0;
                    while (k < 3 && a < L) {                       /* three in a row, or up to the end of what is inflated */
                        const size_t step = looks_like_record(u, a, L, n_targets);
                        if (!step) break;
                        a += step;
                        k++;
                    }
                    if (k == 3 || (k > 0 && a >= L)) {
                        q->c = c;
                        break;
                    }
                }
            }
            q->end = q->c;
            if (q->c != SIZE_MAX) q->end = hop_run(u, q->c, lim, L, spec + q->first, &q->n, &q->why);
        }
        /* in stream order: accept a piece whose guess is where the chain arrives, walk it again otherwise */
        size_t cur = p0;
        p = p0;
        for (size_t t = 0; t < T; t++) {
            const size_t lim = t + 1 < T ? p0 + (t + 1) * span : L;
            hop_piece *q = &pc[t];
            if (cur >= lim) {                                      /* a record reaches over the whole piece */
                q->n = 0;
                continue;
            }
            if (q->c != cur) {
                q->n = 0;
                q->end = hop_run(u, cur, lim, L, spec + q->first, &q->n, &q->why);
                r->hop_redone++;
            }
            p = cur = q->end;
            why = q->why;
            if (why) {
                for (size_t k = t + 1; k < T; k++) pc[k].n = 0;
                break;
            }
        }
        size_t tot = 0;
        for (size_t t = 0; t < T; t++) {
            const size_t n = pc[t].n;
            pc[t].c = tot;                                         /* reused: the piece's place in rec_off */
            tot += n;
        }
#pragma omp parallel for schedule(static)
        for (long t = 0; t < (long)T; t++)
            if (pc[t].n) memcpy(r->rec_off + pc[t].c, spec + pc[t].first, sizeof(size_t) * pc[t].n);
        r->n_rec = tot;
        r->hop_pieces += T;
        free(pc);
    }
    if (why == 2) {                                                /* bam.c:186-190: a malformed length ends the file */
        r->eof = 1;
        r->ulen = p;
    }
    if (r->n_rec) r->upos = p;                                     /* consumed up to here once these are parsed */
}

/* The device decoder's batch: the records of the current window are located and parsed on the device in one go; a batch
 * is a slice of them copied into the staging arrays. Read names and XA / NM strings, when somebody wants them, are cut
 * out of the records' raw bytes, fetched for just the records concerned. */
/* makes drec_next < dn_rec: parses the current window, moves on to the next one when it is used up; 0 at end of input */
static int dev_ensure_records(aln_reader *r)
{
    if (!r->pf_on && !r->eof) pf_start(r);
    while (r->drec_next == r->dn_rec) {
        if (!r->dparsed) {
            const double th = now_s();
            int malformed = 0, flags = 0;
            size_t redo = 0;
            DEV_CHK(dev.parse(dev.ctx, r->dw, r->n_targets, &r->dn_rec, &malformed, &flags, &redo), "parse");
            t_hop += now_s() - th;
            r->dparsed = 1;
            r->drec_next = 0;
            r->d_rewalked += redo;
            r->dflags = flags;
            r->dseen_ok = 0;
            if (malformed) {
                r->dlast = 1;                                  /* bam.c:186-190: nothing after this window counts */
                if (r->rg_on) r->rg_suspect = 1;               /* ... of the whole file: the other shares must not count either */
            }
            if (r->win_hook && r->dn_rec) r->win_hook(r->win_hook_ctx, r->dn_rec);
            continue;
        }
        if (r->dlast || r->eof) {                              /* end of input (a truncated tail record is dropped) */
            if (r->rg_on && r->rg_end_block != SIZE_MAX && !r->rg_verified) {
                /* the share's end is a true record start iff the chain of records arrives exactly there: nothing is left over */
                size_t left = 1;
                DEV_CHK(dev.avail(dev.ctx, r->dw, &left), "avail");
                r->rg_verified = (r->rg_stop_hit && !r->rg_stop_missed && !r->dlast && left == 0 && r->dskip_left == 0) ? 1 : -1;
            }
            return 0;
        }
        dev_advance(r);
    }
    return 1;
}

int aln_device_window(aln_reader *r, int *flags, const uint8_t **tid_seen)
{
    if (!r->dev || !dev_ensure_records(r) || r->drec_next != 0) return 0;
    if (!r->dseen_ok) {
        r->dseen = xrealloc(r->dseen, (size_t)r->n_targets + 1);
        DEV_CHK(dev.tids(dev.ctx, r->dseen, r->n_targets), "tids");
        r->dseen_ok = 1;
    }
    *flags = r->dflags;
    *tid_seen = r->dseen;
    return 1;
}

void aln_set_window_hook(aln_reader *r, void (*fn)(void *ctx, size_t n_rec), void *ctx)
{
    r->win_hook = fn;
    r->win_hook_ctx = ctx;
}

void aln_readahead(aln_reader *r)
{
    if (!r->is_sam && !r->pf_on && !r->eof) pf_start(r);
}

size_t aln_device_left(const aln_reader *r) { return r->dev ? r->dn_rec - r->drec_next : 0; }

int aln_device_xa_veto(aln_reader *r, itx_xaveto *x, size_t n, uint64_t *n_vetoed, uint64_t *n_hard)
{
    if (!r->dev || !dev.xa_veto || n > r->drec_next) return -1;
    DEV_CHK(dev.xa_veto(dev.ctx, x, r->drec_next - n, n, n_vetoed, n_hard), "xa_veto");
    return 0;
}

void aln_device_rewind(aln_reader *r, size_t n)
{
    if (r->dev && r->dparsed) r->drec_next = n <= r->drec_next ? r->drec_next - n : 0;
}

int aln_device_exhausted(aln_reader *r) { return r->dev && !dev_ensure_records(r); }

size_t aln_read_batch_device(aln_reader *r, size_t cap, itx_batch *b)
{
    if (!r->dev || !dev_ensure_records(r)) return 0;
    size_t m = r->dn_rec - r->drec_next;
    if (m > cap) m = cap;
    DEV_CHK(dev.device_batch(dev.ctx, r->drec_next, r->dflags & 1, b), "device_batch");
    r->drec_next += m;
    return m;
}

static size_t dev_read_batch(aln_reader *r, itx_staging *st, size_t cap, aln_side *side, int *any_paired, int *aux_xa)
{
    size_t n = 0;
    while (n < cap) {
        if (r->drec_next == r->dn_rec) {
            if (n) break;                                          /* a batch stays inside one window */
            if (!dev_ensure_records(r)) break;
        }
        size_t m = r->dn_rec - r->drec_next;
        if (m > cap - n) m = cap - n;
        const double tp = now_s();
        const int want_q = side && side->want_qnames, want_a = side && side->want_aux && (r->dflags & 2);
        if (want_q || want_a) {
            if (r->d_side_cap < m + 1) {
                r->d_side_cap = m + m / 4 + 1;
                r->d_off = xrealloc(r->d_off, sizeof(uint32_t) * r->d_side_cap);
                r->d_xa = xrealloc(r->d_xa, r->d_side_cap);
            }
            DEV_CHK(dev.fetch(dev.ctx, r->drec_next, m, st, n, r->d_off, r->d_xa), "fetch");
            /* the raw bytes from the first wanted record to the end of the last one */
            size_t i0 = 0, i1 = m;
            if (!want_q) {
                while (i0 < m && !r->d_xa[i0]) i0++;
                while (i1 > i0 && !r->d_xa[i1 - 1]) i1--;
            }
            if (i1 > i0) {
                uint8_t lenb[4];
                DEV_CHK(dev.bytes(dev.ctx, r->d_off[i1 - 1], lenb, 4), "bytes");
                const size_t lo = r->d_off[i0], hi = (size_t)r->d_off[i1 - 1] + 4 + (size_t)rd_u32(lenb);
                if (r->d_raw_cap < hi - lo) {
                    r->d_raw_cap = (hi - lo) + (hi - lo) / 4;
                    free(r->d_raw);
                    r->d_raw = xmalloc(r->d_raw_cap);
                }
                DEV_CHK(dev.bytes(dev.ctx, lo, r->d_raw, hi - lo), "bytes");
                const uint8_t *raw = r->d_raw;
                const uint32_t *ro = r->d_off;
                const uint8_t *xa = r->d_xa;
#pragma omp parallel for schedule(static)
                for (long i = (long)i0; i < (long)i1; i++)
                    if (want_q || xa[i]) {
                        int a1 = 0, x1 = 0;
                        const uint8_t marked = st->flag5[n + (size_t)i] & ITX_F5_NOLOOKUP;       /* set on the device (-R): the record's bytes do not hold it */
                        bam_parse_one(raw + (ro[i] - lo), n + (size_t)i, st, side, &a1, &x1);   /* the same field values again, plus the strings */
                        st->flag5[n + (size_t)i] |= marked;
                    }
            }
            side->has_strings = 1;                 /* entries outside [i0, i1) were never written: still NULL */
        } else {
            DEV_CHK(dev.fetch(dev.ctx, r->drec_next, m, st, n, NULL, NULL), "fetch");
        }
        t_parse += now_s() - tp;
        if (r->dflags & 1) *any_paired = 1;
        if (r->dflags & 2) *aux_xa = 1;
        r->drec_next += m;
        n += m;
    }
    return n;
}

static size_t bam_read_batch(aln_reader *r, itx_staging *st, size_t cap, aln_side *side, int *any_paired, int *aux_xa)
{
    if (r->dev) return dev_read_batch(r, st, cap, side, any_paired, aux_xa);
    if (side) side->has_strings = 1;
    size_t n = 0;
    if (!r->pf_on && !r->eof) pf_start(r);
    while (n < cap) {
        if (r->rec_next == r->n_rec) {
            /* locate the records of what is inflated; load more when none is complete */
            r->n_rec = r->rec_next = 0;
            double th = now_s();
            for (;;) {
                locate_records(r);
                if (r->n_rec) break;
                const double tl = now_s();
                const size_t more = bgzf_load_chunk(r);
                th += now_s() - tl;                               /* the load is accounted as io / inflate */
                if (more == 0 && r->eof) break;                   /* end of input (a truncated tail record is dropped) */
            }
            t_hop += now_s() - th;
            if (r->n_rec == 0) break;
        }
        size_t m = r->n_rec - r->rec_next;
        if (m > cap - n) m = cap - n;
        const size_t *ro = r->rec_off + r->rec_next;
        const double tp = now_s();
        int ap = 0, xa = 0;
#pragma omp parallel for schedule(static) reduction(| : ap, xa)
        for (long i = 0; i < (long)m; i++) {
            int a1 = 0, x1 = 0;
            bam_parse_one(r->ubuf + ro[i], n + (size_t)i, st, side, &a1, &x1);
            ap |= a1;
            xa |= x1;
        }
        t_parse += now_s() - tp;
        if (ap) *any_paired = 1;
        if (xa) *aux_xa = 1;
        r->rec_next += m;
        n += m;
    }
    return n;
}

/* bam_import.c: the textual flag letters of samtools 0.1.x ("pPuUrR12sfd") */
static unsigned flag_from_chars(const char *s)
{
    unsigned f = 0;
    for (; *s; ++s) {
        switch (*s) {
        case 'p': f |= 0x1; break;
        case 'P': f |= 0x2; break;
        case 'u': f |= 0x4; break;
        case 'U': f |= 0x8; break;
        case 'r': f |= 0x10; break;
        case 'R': f |= 0x20; break;
        case '1': f |= 0x40; break;
        case '2': f |= 0x80; break;
        case 's': f |= 0x100; break;
        case 'f': f |= 0x200; break;
        case 'd': f |= 0x400; break;
        default: break;
        }
    }
    return f;
}

static size_t sam_read_batch(aln_reader *r, itx_staging *st, size_t cap, aln_side *side, int *any_paired, int *aux_xa)
{
    if (side) side->has_strings = 1;
    size_t n = 0;
    while (n < cap) {
        ssize_t len;
        if (r->pending_len >= 0) {
            len = r->pending_len;
            r->pending_len = -1;
        } else {
            len = getline(&r->line, &r->line_cap, r->f);
            if (len < 0) break;
            r->n_lines++;
        }
        while (len > 0 && (r->line[len - 1] == '\n' || r->line[len - 1] == '\r')) r->line[--len] = 0;
        if (len == 0) continue;                                          /* empty lines are skipped */
        char *fld[12];
        int nf = 0;
        char *p = r->line;
        while (nf < 12) {
            fld[nf++] = p;
            if (nf == 12) break;                                         /* the 12th "field" keeps all optional fields */
            char *t = strchr(p, '\t');
            if (!t) break;
            *t = 0;
            p = t + 1;
        }
        if (nf < 11) break;                                              /* truncated line: the parser gives up */
        char *endp;
        long flag = strtol(fld[1], &endp, 0);
        if (*endp) flag = (long)flag_from_chars(fld[1]);
        int32_t tid = -1;
        if (strcmp(fld[2], "*") != 0) {
            const int64_t t = names_find(&r->tnames, fld[2]);
            if (t < 0) {
                if (r->n_targets == 0) {
                    fprintf(stderr, "[sam_read1] missing header? Abort!\n");
                    exit(1);
                }
                fprintf(stderr, "[sam_read1] reference '%s' is recognized as '*'.\n", fld[2]);
            }
            tid = (int32_t)t;
        }
        const int32_t pos = isdigit((unsigned char)fld[3][0]) ? atoi(fld[3]) - 1 : -1;
        const int qual = isdigit((unsigned char)fld[4][0]) ? atoi(fld[4]) : 0;
        uint32_t e = (uint32_t)pos;
        int n_cigar = 0;
        if (fld[5][0] != '*') {
            const char *s = fld[5];
            while (*s) {
                char *t;
                const long x = strtol(s, &t, 10);
                const int op = toupper((unsigned char)*t);
                if (!*t) break;
                if (op == 'M' || op == 'D' || op == 'N') e += (uint32_t)x;
                n_cigar++;
                s = t + 1;
            }
        } else if (!(flag & 0x4)) {
            fprintf(stderr, "Parse warning at line %lld: mapped sequence without CIGAR\n", r->n_lines);
            flag |= 0x4;
        }
        const int32_t mpos = isdigit((unsigned char)fld[7][0]) ? atoi(fld[7]) - 1 : -1;
        const int32_t isize = (fld[8][0] == '-' || isdigit((unsigned char)fld[8][0])) ? atoi(fld[8]) : 0;
        const int32_t l_qseq = strcmp(fld[9], "*") == 0 ? 0 : (int32_t)strlen(fld[9]);
        st->tid[n] = tid;
        st->pos[n] = pos;
        st->tmpend[n] = n_cigar ? (int32_t)e : (int32_t)((uint32_t)pos + (uint32_t)l_qseq);
        st->mapq[n] = (uint8_t)qual;
        st->flag5[n] = ITX_FLAG5((unsigned)flag);
        st->mpos[n] = mpos;
        st->isize[n] = isize;
        if (flag & 1) *any_paired = 1;
        if (side && side->want_qnames) side->qname[n] = xstrdup(fld[0]);
        if (side && side->want_aux) {
            side->xa[n] = NULL;
            side->nm[n] = 0;
        }
        if (nf == 12) {
            /* optional fields TAG:TYPE:VALUE (bam_import.c:402-470); the first XA and the first NM count (bam_aux_get) */
            const char *xa = NULL, *nm = NULL;
            for (char *a = fld[11]; a; a = strchr(a, '\t') ? strchr(a, '\t') + 1 : NULL) {
                if (!xa && strncmp(a, "XA:", 3) == 0) xa = a;
                if (!nm && strncmp(a, "NM:", 3) == 0) nm = a;
            }
            if (xa) *aux_xa = 1;
            if (xa && side && side->want_aux) {
                const int is_z = strlen(xa) >= 5 && (xa[3] == 'Z' || xa[3] == 'H') && xa[4] == ':';
                const char *v = is_z ? xa + 5 : "";
                side->xa[n] = xstrndup_bound(v, strcspn(v, "\t"));
                if (nm && strlen(nm) >= 5 && nm[3] == 'i' && nm[4] == ':') side->nm[n] = (int32_t)strtol(nm + 5, NULL, 10);
            }
        }
        n++;
    }
    return n;
}

size_t aln_read_batch(aln_reader *r, itx_staging *st, size_t cap, aln_side *side, int *any_paired, int *aux_xa)
{
    return r->is_sam ? sam_read_batch(r, st, cap, side, any_paired, aux_xa) : bam_read_batch(r, st, cap, side, any_paired, aux_xa);
}
