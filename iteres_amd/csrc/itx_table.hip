// itx_table.hip — builds the device-resident repeat table.
//
// Stands in for rmsk2binKeeperHash's per-chromosome binKeeper (generic.c:1613-1626,
// cuskent/binRange.c:140-186). The reference answers "which rows overlap [start,end)" by walking
// six levels of LIFO bin lists; the order in which it RETURNS the hits matters (generic.c:950-960
// picks "the last hit whose coverage beats the previous hit's"). Here the rows of a chromosome are
// laid out as one start-sorted array with
//   * a binned start index (bidx) for an O(1) upper bound on the candidate range,
//   * a prefix-maximum of the ends (pmax_e) as the scan-stop bound, and
//   * each row's RANK in binKeeperFind's return order (level coarse->fine, bin descending,
//     insertion ascending — cuskent/binRange.c:209-225) so the kernel can replay the best-hit rule
//     exactly without the bin lists.
#include "itx_common.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

static thread_local char g_err[512];
void itx_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
extern "C" const char *itx_last_error(void) { return g_err; }
extern "C" int itx_abi_version(void) { return 1000; }
extern "C" int itx_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        itx_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return ITX_E_NO_DEVICE;
    }
    return n;
}

// cuskent/binRange.c:20-21,119-138 (binFromRangeBinKeeperExtended): level 0 = finest (128 kb).
static const int kBinOffsets[6] = {4096 + 512 + 64 + 8 + 1, 512 + 64 + 8 + 1, 64 + 8 + 1, 8 + 1, 1, 0};
static bool bin_of_range(int start, int end, int *level, int *bin)
{
    int sb = start >> 17, eb = (end - 1) >> 17;
    for (int i = 0; i < 6; ++i) {
        if (sb == eb) {
            *level = i;
            *bin = kBinOffsets[i] + sb;
            return true;
        }
        sb >>= 3;
        eb >>= 3;
    }
    return false;
}

template <class T> static T *carve(char *base, size_t &off, size_t count)
{
    off = (off + 255) & ~size_t(255);
    T *p = reinterpret_cast<T *>(base + off);
    off += count * sizeof(T);
    return p;
}

extern "C" int itx_table_create(const itx_row *rows, size_t n_rows, const int64_t *chrom_size, int n_chrom,
                                const uint32_t *rep_len, uint32_t n_rep, uint32_t n_fam, uint32_t n_cla, int device,
                                itx_table **out, size_t *bad_row)
{
    if (!out || n_chrom < 0 || (n_rows && !rows) || (n_chrom && !chrom_size) || (n_rep && !rep_len)) {
        itx_set_error("itx_table_create: null argument");
        return ITX_E_ARG;
    }
    if (n_rows >= (1ull << 31)) {
        itx_set_error("itx_table_create: %zu rows exceed the 2^31 row limit", n_rows);
        return ITX_E_LIMIT;
    }
    if (n_fam > 65535 || n_cla > 65535) {
        itx_set_error("itx_table_create: more than 65535 families/classes (%u/%u)", n_fam, n_cla);
        return ITX_E_LIMIT;
    }
    for (int c = 0; c < n_chrom; c++)
        if (chrom_size[c] < 0 || chrom_size[c] > 0x7fffffffLL) {
            itx_set_error("itx_table_create: chromosome %d size %lld outside int range (binKeeperNew takes int)", c,
                          (long long)chrom_size[c]);
            return ITX_E_LIMIT;
        }
    // slot space: rep_len+1 slots per name
    std::vector<uint32_t> covslot(n_rep + 1);
    std::vector<uint64_t> covoff(n_rep + 1);
    uint64_t slots = 0, cov = 0;
    for (uint32_t r = 0; r < n_rep; r++) {
        covslot[r] = (uint32_t)slots;
        covoff[r] = cov;
        slots += (uint64_t)rep_len[r] + 1;
        cov += rep_len[r];
        if (slots >= (1ull << 30)) {
            itx_set_error("itx_table_create: consensus slot space exceeds 2^30");
            return ITX_E_LIMIT;
        }
    }
    covslot[n_rep] = (uint32_t)slots;
    covoff[n_rep] = cov;

    // validate rows as binKeeperAdd would, bucket by chromosome
    std::vector<uint32_t> chrom_cnt(n_chrom + 1, 0);
    std::vector<int> lvl(n_rows), bin(n_rows);
    for (size_t i = 0; i < n_rows; i++) {
        const itx_row &r = rows[i];
        bool ok = r.chrom >= 0 && r.chrom < n_chrom && (int)chrom_size[r.chrom] != 0 && r.rep < n_rep && r.fam < n_fam &&
                  r.cla < n_cla;
        if (ok) {
            int s = (int)r.start, e = (int)r.end, maxPos = (int)chrom_size[r.chrom];
            ok = !(s < 0 || e > maxPos || s > e) && bin_of_range(s, e, &lvl[i], &bin[i]);
        }
        if (!ok) {
            if (bad_row) *bad_row = i;
            itx_set_error("itx_table_create: row %zu (chrom %d, %u-%u) is outside its chromosome or has bad ids", i,
                          r.chrom, r.start, r.end);
            return ITX_E_RANGE;
        }
        chrom_cnt[r.chrom + 1]++;
    }
    std::vector<uint32_t> chrom_off(n_chrom + 1, 0);
    for (int c = 0; c < n_chrom; c++) chrom_off[c + 1] = chrom_off[c] + chrom_cnt[c + 1];

    // start-sorted order per chromosome (stable in file order)
    std::vector<uint32_t> order(n_rows);
    {
        std::vector<uint32_t> fill(chrom_off.begin(), chrom_off.end() - 1);
        for (size_t i = 0; i < n_rows; i++) order[fill[rows[i].chrom]++] = (uint32_t)i;
    }
    for (int c = 0; c < n_chrom; c++)
        std::stable_sort(order.begin() + chrom_off[c], order.begin() + chrom_off[c + 1],
                         [&](uint32_t a, uint32_t b) { return (int)rows[a].start < (int)rows[b].start; });
    // rank in binKeeperFind's return order: level coarse (5) -> fine (0), bin descending, file order ascending
    std::vector<uint32_t> rank_of_row(n_rows);
    {
        std::vector<uint32_t> canon(n_rows);
        for (int c = 0; c < n_chrom; c++) {
            uint32_t lo = chrom_off[c], hi = chrom_off[c + 1];
            std::copy(order.begin() + lo, order.begin() + hi, canon.begin() + lo);
            std::sort(canon.begin() + lo, canon.begin() + hi, [&](uint32_t a, uint32_t b) {
                if (lvl[a] != lvl[b]) return lvl[a] > lvl[b];
                if (bin[a] != bin[b]) return bin[a] > bin[b];
                return a < b;
            });
            for (uint32_t k = lo; k < hi; k++) rank_of_row[canon[k]] = k - lo;
        }
    }
    // bin shift: finest power of two >= 128 bp that keeps the index within ~4 entries per row
    uint64_t genome = 0;
    for (int c = 0; c < n_chrom; c++) genome += (uint64_t)chrom_size[c];
    int shift = 7;
    uint64_t budget = std::max<uint64_t>(4 * (uint64_t)n_rows, 1u << 16);
    while (shift < 17 && (genome >> shift) + 2 * (uint64_t)n_chrom > budget) shift++;
    std::vector<uint32_t> bin_off(n_chrom + 1, 0);
    for (int c = 0; c < n_chrom; c++) bin_off[c + 1] = bin_off[c] + (uint32_t)(((uint64_t)chrom_size[c] >> shift) + 2);
    std::vector<uint32_t> bidx(bin_off[n_chrom]);
    std::vector<ItxIv> iv(n_rows);
    std::vector<uint32_t> rnk(n_rows);
    std::vector<int32_t> orig(n_rows);
    std::vector<int32_t> csize(n_chrom);
    for (int c = 0; c < n_chrom; c++) {
        csize[c] = (int32_t)chrom_size[c];
        uint32_t lo = chrom_off[c], hi = chrom_off[c + 1];
        int32_t pm = INT32_MIN;
        for (uint32_t k = lo; k < hi; k++) {
            const itx_row &r = rows[order[k]];
            ItxIv &d = iv[k];
            d.s = (int32_t)r.start;
            d.e = (int32_t)r.end;
            pm = std::max(pm, d.e);
            d.pmax_e = pm;
            d.cs = r.cons_start;
            uint32_t len = rep_len[r.rep];
            d.jcap = std::min(r.cons_end, len);
            d.covslot = covslot[r.rep];
            d.zslot = covslot[r.rep] + len;
            d.famcla = (r.fam << 16) | r.cla;
            rnk[k] = rank_of_row[order[k]];
            orig[k] = (int32_t)order[k];
        }
        uint32_t nb = bin_off[c + 1] - bin_off[c];
        uint32_t k = lo;
        for (uint32_t b = 0; b < nb; b++) {
            int64_t bound = (int64_t)b << shift;
            while (k < hi && (int64_t)iv[k].s < bound) k++;
            bidx[bin_off[c] + b] = k;
        }
    }

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        itx_set_error("itx_table_create: HIP device %d not available (%d visible)", device, ndev);
        return ITX_E_NO_DEVICE;
    }
    ITX_HIP(hipSetDevice(device));
    itx_table *t = new itx_table();
    memset(t, 0, sizeof *t);
    t->device = device;
    t->n_chrom = n_chrom;
    t->shift = shift;
    t->n_rows = (uint32_t)n_rows;
    t->n_rep = n_rep;
    t->n_fam = n_fam;
    t->n_cla = n_cla;
    t->n_slots = (uint32_t)slots;
    t->cov_len = cov;
    // one allocation, carved
    size_t off = 0;
    char *nullbase = nullptr;
    ItxIv *o_iv = carve<ItxIv>(nullbase, off, n_rows + 1);
    uint32_t *o_rank = carve<uint32_t>(nullbase, off, n_rows + 1);
    int32_t *o_orig = carve<int32_t>(nullbase, off, n_rows + 1);
    uint32_t *o_coff = carve<uint32_t>(nullbase, off, n_chrom + 1);
    uint32_t *o_boff = carve<uint32_t>(nullbase, off, n_chrom + 1);
    uint32_t *o_bidx = carve<uint32_t>(nullbase, off, bidx.size() + 1);
    int32_t *o_csz = carve<int32_t>(nullbase, off, n_chrom + 1);
    uint32_t *o_rlen = carve<uint32_t>(nullbase, off, n_rep + 1);
    uint32_t *o_cslot = carve<uint32_t>(nullbase, off, n_rep + 1);
    uint64_t *o_covoff = carve<uint64_t>(nullbase, off, n_rep + 1);
    size_t total = (off + 255) & ~size_t(255);
    char *base = nullptr;
    hipError_t he = hipMalloc((void **)&base, total);
    if (he != hipSuccess) {
        itx_set_error("itx_table_create: hipMalloc(%zu) failed: %s", total, hipGetErrorString(he));
        delete t;
        return ITX_E_NOMEM;
    }
    t->d_all = base;
    t->table_bytes = total;
#define DEV(p) reinterpret_cast<decltype(p)>(base + reinterpret_cast<size_t>(p))
#define UP(dst, vec)                                                                                   \
    if (!(vec).empty()) {                                                                              \
        he = hipMemcpy(DEV(dst), (vec).data(), (vec).size() * sizeof((vec)[0]), hipMemcpyHostToDevice); \
        if (he != hipSuccess) {                                                                        \
            itx_set_error("itx_table_create: upload failed: %s", hipGetErrorString(he));               \
            (void)hipFree(base);                                                                           \
            delete t;                                                                                  \
            return ITX_E_NO_DEVICE;                                                                    \
        }                                                                                              \
    }
    std::vector<uint32_t> rlen(rep_len, rep_len + n_rep);
    UP(o_iv, iv);
    UP(o_rank, rnk);
    UP(o_orig, orig);
    UP(o_coff, chrom_off);
    UP(o_boff, bin_off);
    UP(o_bidx, bidx);
    UP(o_csz, csize);
    UP(o_rlen, rlen);
    UP(o_cslot, covslot);
    UP(o_covoff, covoff);
    t->dev.iv = DEV(o_iv);
    t->dev.rank = DEV(o_rank);
    t->dev.orig = DEV(o_orig);
    t->dev.chrom_off = DEV(o_coff);
    t->dev.bin_off = DEV(o_boff);
    t->dev.bidx = DEV(o_bidx);
    t->dev.chrom_size = DEV(o_csz);
    t->dev.n_chrom = n_chrom;
    t->dev.shift = shift;
    t->dev.n_rows = (uint32_t)n_rows;
    t->dev.n_rep = n_rep;
    t->dev.n_fam = n_fam;
    t->dev.n_cla = n_cla;
    t->dev.n_slots = (uint32_t)slots;
    t->d_rep_len = DEV(o_rlen);
    t->d_covslot = DEV(o_cslot);
    t->d_covoff = DEV(o_covoff);
#undef UP
#undef DEV
    t->h_rep_len = (uint32_t *)malloc(sizeof(uint32_t) * (n_rep + 1));
    t->h_covslot = (uint32_t *)malloc(sizeof(uint32_t) * (n_rep + 1));
    if (n_rep) memcpy(t->h_rep_len, rep_len, sizeof(uint32_t) * n_rep);
    memcpy(t->h_covslot, covslot.data(), sizeof(uint32_t) * (n_rep + 1));
    *out = t;
    return ITX_OK;
}

extern "C" void itx_table_destroy(itx_table *t)
{
    if (!t) return;
    if (t->d_all) {
        (void)hipSetDevice(t->device);
        (void)hipFree(t->d_all);
    }
    free(t->h_rep_len);
    free(t->h_covslot);
    delete t;
}

extern "C" int itx_table_get_info(const itx_table *t, itx_table_info *o)
{
    if (!t || !o) {
        itx_set_error("itx_table_get_info: null argument");
        return ITX_E_ARG;
    }
    ItxAccumLayout L = itx_accum_layout(t->n_rep, t->n_fam, t->n_cla, t->n_slots, t->n_rows);
    memset(o, 0, sizeof *o);
    o->n_rows = t->n_rows;
    o->n_rep = t->n_rep;
    o->n_fam = t->n_fam;
    o->n_cla = t->n_cla;
    o->cov_len = t->cov_len;
    o->n_u64 = L.n_u64;
    o->n_u32 = L.n_u32;
    o->table_bytes = t->table_bytes;
    o->n_chrom = t->n_chrom;
    o->bin_shift = t->shift;
    o->device = t->device;
    return ITX_OK;
}

extern "C" int itx_table_cov_offsets(const itx_table *t, uint64_t *off)
{
    if (!t || !off) {
        itx_set_error("itx_table_cov_offsets: null argument");
        return ITX_E_ARG;
    }
    uint64_t acc = 0;
    for (uint32_t r = 0; r < t->n_rep; r++) {
        off[r] = acc;
        acc += t->h_rep_len[r];
    }
    off[t->n_rep] = acc;
    return ITX_OK;
}
