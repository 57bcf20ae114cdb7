// itx_xaveto.hip — the XA / NM multi-mapping veto on the device (generic.c:303-341 mapped2diffSubfam, gate generic.c:972-982;
// on by default in `iteres stat`, stat.c:34,51): a record that chose a row is dropped — reads_diff_subfam++ — when one of
// its alternative hits `chr,±pos,CIGAR,NM'` with NM' <= NM overlaps a row of ANOTHER subfamily (names compared without
// case, sameWord). On real bwa-aln BAMs a large share of the reads carries XA; with the strings walked on the host every
// such window had to leave HBM (measured: 6.1 s instead of 1.8 s for 100 M reads, 40 % of them with XA). Here the tag is
// read where the inflated record lies:
//
//   k_xa_verdict   one thread per record of the parsed window that carries XA and was classified (chosen row >= 0): walks
//                  the record's tags (bam_aux.c:36-48) for XA:Z and NM, re-derives the record's interval for qlen
//                  (generic.c:764-905), chops the string like chopByChar does (cuskent/common.c:2029-2053: at most 100
//                  alternatives, four fields each), finds the alternative's chromosome by name in a small hash table, and
//                  asks the table the one question the reference asks with binKeeperFind (binRange.c:196-227): is there an
//                  overlapping row whose name is another word? — the same downward scan over start-sorted rows as the
//                  classify path (ItxIv.pbelow), no bin lists.
//                  verdict[i] = 0 keep, 1 veto, 2 "the host has to look": a number that is not plain decimal (strtol with
//                  base 0 also takes blanks, 0x.. and octal), fewer than four fields (the reference asserts), anything
//                  this parser does not model. A window with a 2 in it takes the host route as before.
//   k_xa_apply     ORs ITX_F5_NOLOOKUP into flag5 of the vetoed records and counts them.
//
// Integer and byte work; every lane walks its own record (divergent by nature, a few hundred bytes each): bound by latency,
// not by anything a roofline prices — it has to beat a PCIe round trip of the window, and does by an order of magnitude.
#include "itx_device.h"

#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

struct XaChrom {              // what a lookup on one chromosome needs (cf. ItxTidRec)
    int32_t size;
    uint32_t iv_lo, iv_hi, bin_base;
};

struct XaName {               // one cell of the name table: FNV-1a of the name, where the name lies, which chromosome
    uint32_t hash, off, len;
    int32_t chrom;            // -1: empty cell
};

struct XaDev {
    const uint32_t *rep_word;     // [n_rep]   case-insensitive identity of every repName
    const uint32_t *row_rep;      // [n_rows]  repName id of the caller's row i
    const XaChrom *chrom;         // [n_chrom]
    const XaName *names;          // [name_mask + 1]
    const uint8_t *pool;          // the names' bytes
    uint32_t name_mask;
    const int2 *tid;              // [n_tid] (chromosome index or < 0, chromosome size)
    int32_t n_tid;
    uint32_t extension, isize_max;
    int32_t treat, discard;
};

struct itx_xaveto {
    int device;
    const itx_table *t;
    XaDev d;
    void *d_rep_word, *d_row_rep, *d_chrom, *d_names, *d_pool, *d_tid;
    int32_t *d_hit;               // chosen rows of the batch being judged
    uint8_t *d_verdict;
    uint32_t *d_count;            // [0] vetoed, [1] needs the host
    size_t cap;
    hipStream_t st;
};

static __device__ inline uint32_t xld32(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }

// strtol(s, 0, 0) on [s, e) as far as plain decimal goes: optional sign, digits; stops at the first other byte. *hard is set for
// what base 0 would read differently (leading blanks, a leading 0 followed by more, more than nine digits).
static __device__ inline int32_t xa_number(const uint8_t *s, const uint8_t *e, bool *hard)
{
    bool neg = false;
    if (s < e && (*s == ' ' || (*s >= 9 && *s <= 13))) *hard = true;
    if (s < e && (*s == '+' || *s == '-')) neg = *s++ == '-';
    if (s < e && *s == '0' && s + 1 < e && ((s[1] >= '0' && s[1] <= '9') || s[1] == 'x' || s[1] == 'X')) *hard = true;
    uint32_t v = 0, nd = 0;
    while (s < e && *s >= '0' && *s <= '9') {
        v = v * 10u + (uint32_t)(*s - '0');
        s++;
        if (++nd > 9) {
            *hard = true;
            break;
        }
    }
    return neg ? -(int32_t)v : (int32_t)v;
}

// is there a row overlapping [start, end) on chromosome c whose repName is another word? (binKeeperFind's clipping included)
static __device__ inline bool xa_any_other(const ItxDevTable &T, const XaDev &D, const XaChrom &c, int32_t start, int32_t end, uint32_t word)
{
    if (start < 0) start = 0;
    if (end > c.size) end = c.size;
    if (start >= end || c.iv_lo >= c.iv_hi) return false;
    const uint32_t top = T.bl[c.bin_base + ((uint32_t)end >> T.shift) + 1].x;      // every row with s < end lies below
    for (uint32_t k = top; k > c.iv_lo;) {
        --k;
        const uint4 v = *reinterpret_cast<const uint4 *>(&T.iv[k]);               // s, e, pbelow, rank
        if (clip_ov((int32_t)v.x, (int32_t)v.y, start, end) > 0) {
            const uint32_t rep = D.row_rep[T.orig[k]];
            if (D.rep_word[rep] != word) return true;
        }
        if ((int32_t)v.z <= start) break;                                          // nothing below ends past the start
    }
    return false;
}

__global__ __launch_bounds__(256) void k_xa_verdict(ItxDevTable T, XaDev D, const uint8_t *__restrict__ u, const uint32_t *__restrict__ rec_off,
                                                    const uint8_t *__restrict__ xa_mark, const int32_t *__restrict__ tid_a, const int32_t *__restrict__ pos_a,
                                                    const int32_t *__restrict__ end_a, const uint8_t *__restrict__ f5_a, const int32_t *__restrict__ mpos_a,
                                                    const int32_t *__restrict__ isize_a, const int32_t *__restrict__ hit, uint32_t n,
                                                    uint8_t *__restrict__ verdict, uint32_t *__restrict__ count)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    uint8_t out = 0;
    const int32_t h = hit[i];
    if (xa_mark[i] && h >= 0) {
        // ---- the record's interval as the loop derives it (generic.c:764-905; a classified record has passed every test there)
        const int32_t t = tid_a[i];
        const int2 tr = (t >= 0 && t < D.n_tid) ? D.tid[t] : make_int2(-1, 0);
        const uint32_t cend = (uint32_t)(tr.y - 1);
        const uint32_t f5 = f5_a[i];
        const int32_t pos = pos_a[i], tmpend = end_a[i], mpos = mpos_a ? mpos_a[i] : 0, isz = isize_a ? isize_a[i] : 0;
        const bool se = D.treat || !(f5 & F5_PAIRED) || (f5 & F5_MUNMAP);
        uint32_t st, en;
        if (se) {
            st = (uint32_t)pos;
            en = cend < (uint32_t)tmpend ? cend : (uint32_t)tmpend;
            if (D.extension) {
                if (!(f5 & F5_REVERSE)) {
                    const uint32_t e2 = st + D.extension;
                    en = e2 < cend ? e2 : cend;
                } else {
                    st = en < D.extension ? 0u : en - D.extension;
                }
            }
        } else if (isz > 0) {
            st = (uint32_t)pos;
            const uint32_t e2 = st + (uint32_t)isz;
            en = cend < e2 ? cend : e2;
        } else {
            st = (uint32_t)mpos;
            const uint32_t e2 = st - (uint32_t)isz;
            en = cend < e2 ? cend : e2;
        }
        const int32_t qlen = (int32_t)(en - st);
        const uint32_t word = D.rep_word[D.row_rep[h]];
        // ---- the tags: first XA (any type) and first NM, each by the walk of bam_aux_get
        const uint8_t *p = u + rec_off[i];
        const uint32_t block_len = xld32(p);
        const uint8_t *data = p + 36;
        const uint32_t dlen = block_len - 32u;
        const uint32_t x1 = xld32(p + 12), x2 = xld32(p + 16);
        const uint32_t l_qname = x1 & 0xffu, n_cigar = x2 & 0xffffu;
        const int32_t l_qseq = (int32_t)xld32(p + 20);
        const uint64_t ql = l_qseq > 0 ? (uint64_t)l_qseq : 0;
        const uint64_t off = (uint64_t)l_qname + 4ull * n_cigar + (ql + 1) / 2 + ql;
        const uint8_t *end = data + dlen;
        const uint8_t *xa = nullptr, *nmv = nullptr;                // the TYPE byte of the tag
        if (off < dlen) {
            const uint8_t *s = data + off;
            while (s + 3 <= end && (!xa || !nmv)) {
                if (!xa && s[0] == 'X' && s[1] == 'A') xa = s + 2;
                if (!nmv && s[0] == 'N' && s[1] == 'M') nmv = s + 2;
                uint32_t type = s[2];
                if (type >= 'a' && type <= 'z') type -= 32u;
                s += 3;
                if (type == 'A' || type == 'C') s += 1;
                else if (type == 'S') s += 2;
                else if (type == 'I' || type == 'F') s += 4;
                else if (type == 'D') s += 8;
                else if (type == 'Z' || type == 'H') {
                    while (s < end && *s) ++s;
                    ++s;
                } else if (type == 'B') {
                    if (s + 5 > end) break;
                    uint32_t sub = s[0];
                    if (sub >= 'a' && sub <= 'z') sub -= 32u;
                    const uint32_t cnt = xld32(s + 1);
                    const uint32_t esz = (sub == 'C' || sub == 'A') ? 1u : (sub == 'S') ? 2u : 4u;
                    if ((uint64_t)cnt * esz > (uint64_t)(end - s)) break;
                    s += 5u + cnt * esz;
                } else
                    break;
            }
        }
        int32_t nm = 0;                                              // bam_aux2i, bam_aux.c:159-170
        if (nmv) {
            const uint8_t ty = *nmv, *q = nmv + 1;
            if (ty == 'c' && q + 1 <= end) nm = (int32_t)(int8_t)q[0];
            else if (ty == 'C' && q + 1 <= end) nm = (int32_t)q[0];
            else if (ty == 's' && q + 2 <= end) nm = (int32_t)(int16_t)(q[0] | q[1] << 8);
            else if (ty == 'S' && q + 2 <= end) nm = (int32_t)(uint16_t)(q[0] | q[1] << 8);
            else if ((ty == 'i' || ty == 'I') && q + 4 <= end) nm = (int32_t)xld32(q);
        }
        if (xa && (*xa == 'Z' || *xa == 'H')) {
            const uint8_t *s = xa + 1, *z = s;
            while (z < end && *z) ++z;                               // the string: up to its NUL, or the end of the record
            bool hard = false, veto = false;
            uint32_t fields = 0;
            // chopByChar(ahstring, ';', row, 100): 100 fields at most, what follows the 100th ';' is not looked at
            while (s <= z && fields < 100u && !veto && !hard) {
                const uint8_t *fe = s;
                while (fe < z && *fe != ';') ++fe;
                fields++;
                if (fe > s) {                                        // strlen(row[i]) > 0
                    // chopByChar(row[i], ',', row2, 4): fields 0..2 end at commas, field 3 at the next comma or the end
                    const uint8_t *f[4], *g[4];
                    uint32_t nf = 0;
                    const uint8_t *c = s;
                    while (nf < 4u) {
                        f[nf] = c;
                        while (c < fe && *c != ',') ++c;
                        g[nf] = c;
                        nf++;
                        if (c >= fe) break;
                        ++c;                                         // behind the comma
                    }
                    if (nf != 4u) {
                        hard = true;                                 // the reference's assert(num2 == 4)
                    } else {
                        const int32_t nm2 = xa_number(f[3], g[3], &hard);
                        if (!hard && nm2 <= nm) {
                            int32_t start = xa_number(f[1], g[1], &hard);
                            if (start < 0) start = -start;
                            if (!hard) {
                                // hashLookup(hashRmsk, row2[0]): the chromosome by its exact name
                                const uint32_t len = (uint32_t)(g[0] - f[0]);
                                uint32_t hsh = 2166136261u;
                                for (uint32_t k = 0; k < len; k++) hsh = (hsh ^ f[0][k]) * 16777619u;
                                int32_t chrom = -1;
                                for (uint32_t slot = hsh & D.name_mask;; slot = (slot + 1u) & D.name_mask) {
                                    const XaName e = D.names[slot];
                                    if (e.chrom < 0) break;
                                    if (e.hash == hsh && e.len == len) {
                                        bool same = true;
                                        for (uint32_t k = 0; k < len && same; k++) same = D.pool[e.off + k] == f[0][k];
                                        if (same) {
                                            chrom = e.chrom;
                                            break;
                                        }
                                    }
                                }
                                if (chrom >= 0) veto = xa_any_other(T, D, D.chrom[chrom], start, start + qlen, word);
                            }
                        }
                    }
                }
                if (fe >= z) break;
                s = fe + 1;
            }
            out = hard ? 2 : veto ? 1 : 0;
        }
    }
    verdict[i] = out;
    const unsigned long long hardm = __ballot(out == 2);
    if ((threadIdx.x & 63u) == 0 && hardm) atomicAdd(&count[1], (uint32_t)__popcll(hardm));
}

__global__ __launch_bounds__(256) void k_xa_apply(const uint8_t *__restrict__ verdict, uint8_t *__restrict__ f5, uint32_t n, uint32_t *__restrict__ count)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const bool v = i < n && verdict[i] == 1;
    if (v) f5[i] |= (uint8_t)F5_NOLOOKUP;
    const unsigned long long m = __ballot(v);
    if ((threadIdx.x & 63u) == 0 && m) atomicAdd(&count[0], (uint32_t)__popcll(m));
}

#define XA_HIP(call)                                                                                      \
    do {                                                                                                  \
        hipError_t err__ = (call);                                                                        \
        if (err__ != hipSuccess) {                                                                        \
            itx_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(err__), __FILE__, __LINE__); \
            return ITX_E_NO_DEVICE;                                                                       \
        }                                                                                                 \
    } while (0)

extern "C" int itx_xaveto_create(const itx_table *t, const itx_params *p, const uint32_t *row_rep, const uint32_t *rep_word, const char *const *chrom_name, int n_chrom,
                                 size_t batch_capacity, itx_xaveto **out)
{
    if (!t || !p || !row_rep || !rep_word || !chrom_name || !out || n_chrom != t->n_chrom || batch_capacity == 0) {
        itx_set_error("itx_xaveto_create: bad argument");
        return ITX_E_ARG;
    }
    *out = nullptr;
    XA_HIP(hipSetDevice(t->device));
    itx_xaveto *x = new itx_xaveto();
    memset((void *)x, 0, sizeof *x);
    x->device = t->device;
    x->t = t;
    x->cap = batch_capacity;
    // chromosomes: what a lookup needs, and the table of their names
    std::vector<XaChrom> ch((size_t)n_chrom);
    std::string pool;
    uint32_t cells = 16;
    while (cells < 4u * (uint32_t)n_chrom) cells <<= 1;
    std::vector<XaName> names(cells);
    for (auto &e : names) e = XaName{0, 0, 0, -1};
    for (int c = 0; c < n_chrom; c++) {
        ch[(size_t)c] = XaChrom{t->h_chrom_size[c], t->h_chrom_off[c], t->h_chrom_off[c + 1], t->h_bin_off[c]};
        const char *nm = chrom_name[c] ? chrom_name[c] : "";
        const uint32_t len = (uint32_t)strlen(nm);
        uint32_t hsh = 2166136261u;
        for (uint32_t k = 0; k < len; k++) hsh = (hsh ^ (uint8_t)nm[k]) * 16777619u;
        uint32_t slot = hsh & (cells - 1);
        bool dup = false;
        while (names[slot].chrom >= 0) {                             // a name listed twice: the first entry answers, like a lookup in the size file's table
            if (names[slot].hash == hsh && names[slot].len == len && memcmp(pool.data() + names[slot].off, nm, len) == 0) {
                dup = true;
                break;
            }
            slot = (slot + 1) & (cells - 1);
        }
        if (dup) continue;
        names[slot] = XaName{hsh, (uint32_t)pool.size(), len, c};
        pool.append(nm, len);
    }
    pool.append(16, '\0');
    auto up = [&](void **dst, const void *src, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(dst, bytes ? bytes : 16);
        if (e != hipSuccess) return e;
        return bytes ? hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) : hipSuccess;
    };
    XA_HIP(up(&x->d_rep_word, rep_word, sizeof(uint32_t) * (size_t)t->n_rep));
    XA_HIP(up(&x->d_row_rep, row_rep, sizeof(uint32_t) * (size_t)t->n_rows));
    XA_HIP(up(&x->d_chrom, ch.data(), sizeof(XaChrom) * ch.size()));
    XA_HIP(up(&x->d_names, names.data(), sizeof(XaName) * names.size()));
    XA_HIP(up(&x->d_pool, pool.data(), pool.size()));
    XA_HIP(hipMalloc((void **)&x->d_hit, sizeof(int32_t) * (batch_capacity + 64)));
    XA_HIP(hipMalloc((void **)&x->d_verdict, batch_capacity + 64));
    XA_HIP(hipMalloc((void **)&x->d_count, 16));
    XA_HIP(hipStreamCreateWithFlags(&x->st, hipStreamNonBlocking));
    x->d.rep_word = (const uint32_t *)x->d_rep_word;
    x->d.row_rep = (const uint32_t *)x->d_row_rep;
    x->d.chrom = (const XaChrom *)x->d_chrom;
    x->d.names = (const XaName *)x->d_names;
    x->d.pool = (const uint8_t *)x->d_pool;
    x->d.name_mask = cells - 1;
    x->d.extension = p->extension;
    x->d.isize_max = p->isize_max;
    x->d.treat = p->treat_pe_as_se;
    x->d.discard = p->discard_half_mapped;
    *out = x;
    return ITX_OK;
}

extern "C" void itx_xaveto_destroy(itx_xaveto *x)
{
    if (!x) return;
    (void)hipSetDevice(x->device);
    if (x->st) {
        (void)hipStreamSynchronize(x->st);
        (void)hipStreamDestroy(x->st);
    }
    (void)hipFree(x->d_rep_word);
    (void)hipFree(x->d_row_rep);
    (void)hipFree(x->d_chrom);
    (void)hipFree(x->d_names);
    (void)hipFree(x->d_pool);
    (void)hipFree(x->d_tid);
    (void)hipFree(x->d_hit);
    (void)hipFree(x->d_verdict);
    (void)hipFree(x->d_count);
    delete x;
}

/* the BAM header in use: tid2chrom as for itx_engine_set_tidmap (index into chrom_size[], or < 0) */
extern "C" int itx_xaveto_set_tidmap(itx_xaveto *x, const int32_t *tid2chrom, int n_tid)
{
    if (!x || n_tid < 0 || (n_tid && !tid2chrom)) return ITX_E_ARG;
    XA_HIP(hipSetDevice(x->device));
    std::vector<int2> v((size_t)n_tid + 1);
    for (int k = 0; k < n_tid; k++) {
        const int32_t c = tid2chrom[k];
        v[(size_t)k] = make_int2(c, (c >= 0 && c < x->t->n_chrom) ? x->t->h_chrom_size[c] : 0);
    }
    XA_HIP(hipStreamSynchronize(x->st));
    if (x->d_tid) XA_HIP(hipFree(x->d_tid));
    x->d_tid = nullptr;
    XA_HIP(hipMalloc(&x->d_tid, sizeof(int2) * ((size_t)n_tid + 1)));
    XA_HIP(hipMemcpy(x->d_tid, v.data(), sizeof(int2) * ((size_t)n_tid + 1), hipMemcpyHostToDevice));
    x->d.tid = (const int2 *)x->d_tid;
    x->d.n_tid = n_tid;
    return ITX_OK;
}

extern "C" int32_t *itx_xaveto_hits(itx_xaveto *x) { return x ? x->d_hit : nullptr; }
extern "C" void *itx_xaveto_stream(itx_xaveto *x) { return x ? (void *)x->st : nullptr; }

// the judgement over n records whose window bytes / offsets / marks / SoA the caller (the inflater) holds; chosen rows in x->d_hit
int itx_xaveto_run(itx_xaveto *x, const uint8_t *u, const uint32_t *rec_off, const uint8_t *xa_mark, const int32_t *tid, const int32_t *pos, const int32_t *end,
                   uint8_t *f5, const int32_t *mpos, const int32_t *isize, size_t n, uint64_t *n_vetoed, uint64_t *n_hard)
{
    if (!x || n > x->cap || !x->d.tid) {
        itx_set_error("itx_xaveto_run: %s", !x ? "no object" : n > x->cap ? "batch exceeds the capacity" : "no tid map");
        return ITX_E_STATE;
    }
    *n_vetoed = *n_hard = 0;
    if (n == 0) return ITX_OK;
    XA_HIP(hipSetDevice(x->device));
    XA_HIP(hipMemsetAsync(x->d_count, 0, 8, x->st));
    hipLaunchKernelGGL(k_xa_verdict, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, x->st, x->t->dev, x->d, u, rec_off, xa_mark, tid, pos, end, f5, mpos, isize, x->d_hit,
                       (uint32_t)n, x->d_verdict, x->d_count);
    XA_HIP(hipGetLastError());
    uint32_t c[2] = {0, 0};
    XA_HIP(hipMemcpyAsync(c, x->d_count, 8, hipMemcpyDeviceToHost, x->st));
    XA_HIP(hipStreamSynchronize(x->st));
    *n_hard = c[1];
    if (c[1]) return ITX_OK;                                          // the host has to look: nothing is marked
    hipLaunchKernelGGL(k_xa_apply, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, x->st, x->d_verdict, f5, (uint32_t)n, x->d_count);
    XA_HIP(hipGetLastError());
    XA_HIP(hipMemcpyAsync(c, x->d_count, 8, hipMemcpyDeviceToHost, x->st));
    XA_HIP(hipStreamSynchronize(x->st));
    *n_vetoed = c[0];
    return ITX_OK;
}
