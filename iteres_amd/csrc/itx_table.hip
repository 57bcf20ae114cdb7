// itx_table.hip — builds the device-resident repeat table.
//
// Stands in for rmsk2binKeeperHash's per-chromosome binKeeper (generic.c:1613-1626,
// cuskent/binRange.c:140-186) plus each row's links into hashRep/hashFam/hashCla (generic.c:1631-1693).
// The reference answers "which rows overlap [start,end)" by walking six levels of LIFO bin lists; the
// order in which it RETURNS the hits matters (generic.c:950-960 picks "the last hit whose coverage
// beats the previous hit's"). Here the rows of a chromosome are one start-sorted array with
//   * a binned index `bl` (per 2^shift bp: first row starting in/after the bin, first row whose
//     prefix-max end passes the bin start) bounding the candidate range from both sides in O(1),
//   * a prefix-maximum of the ends of the rows below (pbelow) as the scan-stop bound, and
//   * each row's RANK in binKeeperFind's return order (level coarse->fine, bin descending,
//     insertion ascending — cuskent/binRange.c:209-225) so the kernel can replay the best-hit rule
//     exactly without the bin lists.
// Accumulation is per UNIT = distinct (repName, repFamily, repClass) triple of a row: the reference bumps
// the row's own family/class entries (generic.c:1010-1024), and repName -> family is not functional in
// rmsk, so counting per triple and summing at finish gives all three stat tables from one accumulator.
#include "itx_common.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <array>
#include <atomic>
#include <numeric>
#include <thread>
#include <set>
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
extern "C" int itx_abi_version(void) { return 1005; }
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

// Runs fn(c) for every chromosome on a few host threads (chromosomes are independent in every sort below).
template <class F>
static void for_each_chrom(int n_chrom, F fn)
{
    unsigned nt = std::thread::hardware_concurrency();
    if (nt > 16) nt = 16;
    if (nt < 1) nt = 1;
    if ((int)nt > n_chrom) nt = n_chrom > 0 ? (unsigned)n_chrom : 1u;
    std::atomic<int> next{0};
    auto work = [&]() {
        for (int c = next.fetch_add(1); c < n_chrom; c = next.fetch_add(1)) fn(c);
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
}

// Runs fn(t, lo, hi) over n items cut into contiguous slices, one per host thread (at most 16, at least `grain` items each).
template <class F>
static size_t for_slices(size_t n, size_t grain, F fn)
{
    unsigned hw = std::thread::hardware_concurrency();
    size_t nt = hw ? hw : 1;
    if (nt > 16) nt = 16;
    if (nt > n / grain + 1) nt = n / grain + 1;
    std::vector<std::thread> th;
    for (size_t t = 1; t < nt; t++) th.emplace_back([&, t]() { fn(t, n * t / nt, n * (t + 1) / nt); });
    fn((size_t)0, (size_t)0, n / nt);
    for (auto &x : th) x.join();
    return nt;
}

struct Carver {
    size_t off = 0;
    size_t take(size_t bytes)
    {
        off = (off + 255) & ~size_t(255);
        size_t o = off;
        off += bytes;
        return o;
    }
};

#include <time.h>
static void tb_tick(const char *what, double *t0)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    const double t = (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
    if (getenv("ITX_TIMING_TABLE")) fprintf(stderr, "[itx timing] table: %s %.3f s\n", what, t - *t0);
    *t0 = t;
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
    for (int c = 0; c < n_chrom; c++)
        if (chrom_size[c] < 0 || chrom_size[c] > 0x7fffffffLL) {
            itx_set_error("itx_table_create: chromosome %d size %lld outside int range (binKeeperNew takes int)", c,
                          (long long)chrom_size[c]);
            return ITX_E_LIMIT;
        }
    double tb0 = 0;
    tb_tick("start", &tb0);
    // validate rows as binKeeperAdd would, bucket by chromosome (slices of the rows in parallel; the FIRST bad row is reported)
    std::vector<uint32_t> chrom_cnt(n_chrom + 1, 0);
    std::vector<int> lvl(n_rows), bin(n_rows);
    {
        std::vector<std::vector<uint32_t>> cnt_t(16, std::vector<uint32_t>((size_t)n_chrom + 1, 0));
        std::vector<size_t> bad_t(16, SIZE_MAX);
        for_slices(n_rows, 65536, [&](size_t t, size_t lo, size_t hi) {
            std::vector<uint32_t> &cc = cnt_t[t];
            for (size_t i = lo; i < hi; i++) {
                const itx_row &r = rows[i];
                bool ok = r.chrom >= 0 && r.chrom < n_chrom && (int)chrom_size[r.chrom] != 0 && r.rep < n_rep && r.fam < n_fam && r.cla < n_cla;
                if (ok) {
                    int s = (int)r.start, e = (int)r.end, maxPos = (int)chrom_size[r.chrom];
                    ok = !(s < 0 || e > maxPos || s > e) && bin_of_range(s, e, &lvl[i], &bin[i]);
                }
                if (!ok) {
                    bad_t[t] = i;
                    return;
                }
                cc[(size_t)r.chrom + 1]++;
            }
        });
        size_t bad = SIZE_MAX;
        for (size_t t = 0; t < 16; t++) bad = std::min(bad, bad_t[t]);
        if (bad != SIZE_MAX) {
            const itx_row &r = rows[bad];
            if (bad_row) *bad_row = bad;
            itx_set_error("itx_table_create: row %zu (chrom %d, %u-%u) is outside its chromosome or has bad ids", bad, r.chrom, r.start, r.end);
            return ITX_E_RANGE;
        }
        for (size_t t = 0; t < 16; t++)
            for (int c = 0; c <= n_chrom; c++) chrom_cnt[(size_t)c] += cnt_t[t][(size_t)c];
    }
    tb_tick("validate + bucket", &tb0);
    // units: distinct (rep, fam, cla) triples, ordered by (rep, fam, cla). Nearly every repName has ONE family and class: a
    // slice of the rows remembers the first (fam, cla) it meets per name in a plain array, and only a row whose name has
    // been seen with another pair goes through a (small) set — no hashing per row (the per-row hash lookups of the first
    // version were 0.17 s of a 0.45 s build). The units of a name are neighbours in sorted order, so a row's unit is its
    // name's first unit plus a scan over that name's handful of units.
    std::vector<uint32_t> unit_of_row(n_rows);
    std::vector<uint4> unit_ids;
    {
        typedef std::array<uint32_t, 3> Triple;
        std::vector<std::vector<Triple>> found(16);
        for_slices(n_rows, 65536, [&](size_t t, size_t lo, size_t hi) {
            std::vector<uint64_t> first_fc((size_t)n_rep, 0);                 // (fam << 32 | cla) + 1 of the name's first row in this slice
            std::set<Triple> extra;                                            // triples beyond a name's first pair (few)
            for (size_t i = lo; i < hi; i++) {
                const uint64_t fc = ((uint64_t)rows[i].fam << 32 | rows[i].cla) + 1;
                uint64_t &f = first_fc[rows[i].rep];
                if (f == fc) continue;
                const Triple k = {rows[i].rep, rows[i].fam, rows[i].cla};
                if (f == 0) {
                    f = fc;
                    found[t].push_back(k);
                } else {
                    extra.insert(k);
                }
            }
            found[t].insert(found[t].end(), extra.begin(), extra.end());
        });
        std::vector<Triple> triples;
        for (auto &f : found) triples.insert(triples.end(), f.begin(), f.end());
        std::sort(triples.begin(), triples.end());
        triples.erase(std::unique(triples.begin(), triples.end()), triples.end());
        unit_ids.resize(triples.size());
        std::vector<uint32_t> first_unit((size_t)n_rep + 1, 0);               // the first unit of every name (names without rows: unused)
        for (size_t k = triples.size(); k-- > 0;) {
            unit_ids[k] = make_uint4(triples[k][0], triples[k][1], triples[k][2], 0);
            first_unit[triples[k][0]] = (uint32_t)k;
        }
        for_slices(n_rows, 65536, [&](size_t, size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; i++) {
                uint32_t u = first_unit[rows[i].rep];
                while (unit_ids[u].y != rows[i].fam || unit_ids[u].z != rows[i].cla) u++;      // (the triple is there: it was collected above)
                unit_of_row[i] = u;
            }
        });
    }
    const uint32_t n_units = (uint32_t)unit_ids.size();
    std::vector<uint64_t> covoff(n_rep + 1);
    uint64_t cov = 0;
    for (uint32_t r = 0; r < n_rep; r++) {
        covoff[r] = cov;
        cov += rep_len[r];
    }
    covoff[n_rep] = cov;
    std::vector<uint32_t> unit_slot(n_units + 1);
    std::vector<uint64_t> unit_covoff(n_units);
    uint64_t slots = 0;
    for (uint32_t u = 0; u < n_units; u++) {
        unit_slot[u] = (uint32_t)slots;
        slots += (uint64_t)rep_len[unit_ids[u].x] + 1;
        if (slots >= (1ull << 30)) {
            itx_set_error("itx_table_create: consensus slot space exceeds 2^30");
            return ITX_E_LIMIT;
        }
        unit_covoff[u] = covoff[unit_ids[u].x];
        const bool prev_same = u > 0 && unit_ids[u - 1].x == unit_ids[u].x;
        const bool next_same = u + 1 < n_units && unit_ids[u + 1].x == unit_ids[u].x;
        unit_ids[u].w = (prev_same || next_same) ? 0u : 1u;     // solo: the name's only unit
    }
    unit_slot[n_units] = (uint32_t)slots;

    std::vector<uint32_t> chrom_off(n_chrom + 1, 0);
    for (int c = 0; c < n_chrom; c++) chrom_off[c + 1] = chrom_off[c] + chrom_cnt[c + 1];
    tb_tick("units", &tb0);
    // start-sorted order per chromosome (stable in file order)
    std::vector<uint32_t> order(n_rows);
    {
        std::vector<uint32_t> fill(chrom_off.begin(), chrom_off.end() - 1);
        for (size_t i = 0; i < n_rows; i++) order[fill[rows[i].chrom]++] = (uint32_t)i;
    }
    for_each_chrom(n_chrom, [&](int c) {
        std::stable_sort(order.begin() + chrom_off[c], order.begin() + chrom_off[c + 1],
                         [&](uint32_t a, uint32_t b) { return (int)rows[a].start < (int)rows[b].start; });
    });
    tb_tick("sort by start", &tb0);
    // rank in binKeeperFind's return order: level coarse (5) -> fine (0), bin descending, file order ascending
    std::vector<uint32_t> rank_of_row(n_rows);
    {
        std::vector<uint32_t> canon(n_rows);
        for_each_chrom(n_chrom, [&](int c) {
            uint32_t lo = chrom_off[c], hi = chrom_off[c + 1];
            std::copy(order.begin() + lo, order.begin() + hi, canon.begin() + lo);
            std::sort(canon.begin() + lo, canon.begin() + hi, [&](uint32_t a, uint32_t b) {
                if (lvl[a] != lvl[b]) return lvl[a] > lvl[b];
                if (bin[a] != bin[b]) return bin[a] > bin[b];
                return a < b;
            });
            for (uint32_t k = lo; k < hi; k++) rank_of_row[canon[k]] = k - lo;
        });
    }
    tb_tick("ranks", &tb0);
    // bin width: about one row per bin (128 bp .. 128 kb)
    uint64_t genome = 0;
    for (int c = 0; c < n_chrom; c++) genome += (uint64_t)chrom_size[c];
    int shift = 7;
    uint64_t budget = std::max<uint64_t>((uint64_t)n_rows + (uint64_t)n_rows / 2, 1u << 16);
    while (shift < 17 && (genome >> shift) + 2 * (uint64_t)n_chrom > budget) shift++;
    std::vector<uint32_t> bin_off(n_chrom + 1, 0);
    for (int c = 0; c < n_chrom; c++) bin_off[c + 1] = bin_off[c] + (uint32_t)(((uint64_t)chrom_size[c] >> shift) + 2);
    std::vector<uint2> bl(bin_off[n_chrom]);
    std::vector<ItxIv> iv(n_rows);
    std::vector<int32_t> orig(n_rows);
    std::vector<uint32_t> row_unit(n_rows);
    std::vector<int32_t> csize(n_chrom);
    for (int c = 0; c < n_chrom; c++) csize[c] = (int32_t)chrom_size[c];
    // rows: slices of the sorted order in parallel. pbelow (the running maximum of the ends of a chromosome's rows below) crosses
    // slice boundaries: a first pass takes every slice's maximum over the rows of its LAST chromosome, a short serial pass turns
    // those into what each slice starts with, the second pass fills the rows. (One thread per chromosome left chr1's 8 % of the
    // rows to one thread: 0.09 s.)
    {
        struct SliceSum {
            int32_t last_chrom, last_max;      // the slice's last chromosome and the maximum end over its rows of it
            bool one_chrom;                    // the whole slice lies in that chromosome
        };
        std::vector<SliceSum> ss(16, SliceSum{-1, INT32_MIN, false});
        const size_t nt = for_slices(n_rows, 32768, [&](size_t t, size_t lo, size_t hi) {
            if (lo >= hi) return;
            const int32_t lc = rows[order[hi - 1]].chrom;
            int32_t mx = INT32_MIN;
            size_t k = hi;
            while (k > lo && rows[order[k - 1]].chrom == lc) {
                mx = std::max(mx, (int32_t)rows[order[k - 1]].end);
                k--;
            }
            ss[t] = SliceSum{lc, mx, k == lo};
        });
        std::vector<int32_t> carry_in(16, INT32_MIN);      // maximum end over the rows of the slice's FIRST chromosome that lie before the slice
        {
            int32_t cur_chrom = -1, cur_max = INT32_MIN;
            for (size_t t = 0; t < nt; t++) {
                const size_t lo = n_rows * t / nt, hi = n_rows * (t + 1) / nt;
                if (lo >= hi) continue;
                const int32_t fc = rows[order[lo]].chrom;
                carry_in[t] = fc == cur_chrom ? cur_max : INT32_MIN;
                if (ss[t].one_chrom && ss[t].last_chrom == cur_chrom) cur_max = std::max(cur_max, ss[t].last_max);
                else {
                    cur_chrom = ss[t].last_chrom;
                    cur_max = ss[t].last_max;
                }
            }
        }
        for_slices(n_rows, 32768, [&](size_t t, size_t lo, size_t hi) {
            int32_t pm = carry_in[t], cur = lo < hi ? rows[order[lo]].chrom : -1;
            for (size_t k = lo; k < hi; k++) {
                const itx_row &r = rows[order[k]];
                if (r.chrom != cur) {
                    cur = r.chrom;
                    pm = INT32_MIN;
                }
                ItxIv &d = iv[k];
                d.s = (int32_t)r.start;
                d.e = (int32_t)r.end;
                d.pbelow = pm;
                pm = std::max(pm, d.e);
                d.cs = r.cons_start;
                const uint32_t len = rep_len[r.rep];
                const uint32_t u = unit_of_row[order[k]];
                d.jcap = std::min(r.cons_end, len);
                d.covslot = unit_slot[u];
                d.zslot = unit_slot[u] + len;
                d.rank = rank_of_row[order[k]];
                orig[k] = (int32_t)order[k];
                row_unit[k] = u;
            }
        });
    }
    // the binned index: slices of ALL bins in parallel; a slice finds its place in the rows by bisection (starts and prefix-max
    // ends both rise along a chromosome) and sweeps on from there
    {
        const size_t n_bins = bin_off[n_chrom];
        auto pmax = [&](uint32_t r) { return std::max(iv[r].pbelow, iv[r].e); };
        for_slices(n_bins, 65536, [&](size_t, size_t g0, size_t g1) {
            int c = (int)(std::upper_bound(bin_off.begin(), bin_off.end(), (uint32_t)g0) - bin_off.begin()) - 1;
            uint32_t k = 0, m = 0;
            bool fresh = true;
            for (size_t g = g0; g < g1; g++) {
                while (g >= bin_off[c + 1]) {
                    c++;
                    fresh = true;
                }
                const uint32_t lo = chrom_off[c], hi = chrom_off[c + 1];
                const int64_t bound = (int64_t)(g - bin_off[c]) << shift;
                if (fresh) {
                    // first row starting at or after the bin; first row whose prefix-max end passes the bin start
                    uint32_t a = lo, b = hi;
                    while (a < b) {
                        const uint32_t mid = a + (b - a) / 2;
                        if ((int64_t)iv[mid].s < bound) a = mid + 1;
                        else b = mid;
                    }
                    k = a;
                    a = lo, b = hi;
                    while (a < b) {
                        const uint32_t mid = a + (b - a) / 2;
                        if ((int64_t)pmax(mid) <= bound) a = mid + 1;
                        else b = mid;
                    }
                    m = a;
                    fresh = false;
                } else {
                    while (k < hi && (int64_t)iv[k].s < bound) k++;
                    while (m < hi && (int64_t)pmax(m) <= bound) m++;
                }
                bl[g] = make_uint2(k, m);
            }
        });
    }

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        itx_set_error("itx_table_create: HIP device %d not available (%d visible)", device, ndev);
        return ITX_E_NO_DEVICE;
    }
    tb_tick("rows + index", &tb0);
    ITX_HIP(hipSetDevice(device));
    Carver cv;
    const size_t o_iv = cv.take((n_rows + 1) * sizeof(ItxIv));
    const size_t o_orig = cv.take((n_rows + 1) * 4);
    const size_t o_bl = cv.take((bl.size() + 1) * sizeof(uint2));
    // first unit reaching into every window of 2^ITX_LOGW slots (k_hist sums the starts of a window per unit)
    const size_t n_win = ((size_t)slots >> ITX_LOGW) + 2;
    std::vector<uint32_t> part_unit(n_win, n_units);
    {
        uint32_t u = 0;
        for (size_t w = 0; w < n_win; w++) {
            while (u < n_units && (uint64_t)unit_slot[u + 1] <= ((uint64_t)w << ITX_LOGW)) u++;
            part_unit[w] = u;
        }
    }
    const size_t o_runit = cv.take((n_rows + 1) * 4);
    const size_t o_punit = cv.take(n_win * 4);
    const size_t o_uslot = cv.take(((size_t)n_units + 1) * 4);
    const size_t o_uids = cv.take(((size_t)n_units + 1) * sizeof(uint4));
    const size_t o_ucov = cv.take(((size_t)n_units + 1) * 8);
    const size_t total = cv.take(0) + 256;
    char *base = nullptr;
    hipError_t he = hipMalloc((void **)&base, total);
    if (he != hipSuccess) {
        itx_set_error("itx_table_create: hipMalloc(%zu) failed: %s", total, hipGetErrorString(he));
        return ITX_E_NOMEM;
    }
    auto up = [&](size_t off, const void *src, size_t bytes) -> bool {
        if (bytes == 0) return true;
        he = hipMemcpy(base + off, src, bytes, hipMemcpyHostToDevice);
        return he == hipSuccess;
    };
    bool ok = up(o_iv, iv.data(), iv.size() * sizeof(ItxIv)) &&
              up(o_orig, orig.data(), orig.size() * 4) && up(o_bl, bl.data(), bl.size() * sizeof(uint2)) &&
              up(o_runit, row_unit.data(), row_unit.size() * 4) && up(o_punit, part_unit.data(), part_unit.size() * 4) &&
              up(o_uslot, unit_slot.data(), unit_slot.size() * 4) && up(o_uids, unit_ids.data(), unit_ids.size() * sizeof(uint4)) &&
              up(o_ucov, unit_covoff.data(), unit_covoff.size() * 8);
    if (!ok) {
        itx_set_error("itx_table_create: upload failed: %s", hipGetErrorString(he));
        (void)hipFree(base);
        return ITX_E_NO_DEVICE;
    }
    itx_table *t = new itx_table();
    memset(t, 0, sizeof *t);
    t->device = device;
    t->n_chrom = n_chrom;
    t->shift = shift;
    t->n_rows = (uint32_t)n_rows;
    t->n_rep = n_rep;
    t->n_fam = n_fam;
    t->n_cla = n_cla;
    t->n_units = n_units;
    t->n_slots = (uint32_t)slots;
    t->cov_len = cov;
    t->d_all = base;
    t->table_bytes = total;
    t->dev.iv = (const ItxIv *)(base + o_iv);
    t->dev.orig = (const int32_t *)(base + o_orig);
    t->dev.bl = (const uint2 *)(base + o_bl);
    t->dev.row_unit = (const uint32_t *)(base + o_runit);
    t->dev.unit_slot = (const uint32_t *)(base + o_uslot);
    t->dev.part_unit = (const uint32_t *)(base + o_punit);
    t->dev.shift = shift;
    t->dev.n_rows = (uint32_t)n_rows;
    t->dev.n_units = n_units;
    t->dev.n_slots = (uint32_t)slots;
    t->d_unit_slot = (uint32_t *)(base + o_uslot);
    t->d_row_unit = (uint32_t *)(base + o_runit);
    t->d_part_unit = (uint32_t *)(base + o_punit);
    t->d_unit_ids = (uint4 *)(base + o_uids);
    t->d_unit_covoff = (uint64_t *)(base + o_ucov);
    t->h_rep_len = (uint32_t *)malloc(sizeof(uint32_t) * (n_rep + 1));
    if (n_rep) memcpy(t->h_rep_len, rep_len, sizeof(uint32_t) * n_rep);
    t->h_chrom_off = (uint32_t *)malloc(sizeof(uint32_t) * (n_chrom + 1));
    t->h_bin_off = (uint32_t *)malloc(sizeof(uint32_t) * (n_chrom + 1));
    t->h_chrom_size = (int32_t *)malloc(sizeof(int32_t) * (n_chrom + 1));
    memcpy(t->h_chrom_off, chrom_off.data(), sizeof(uint32_t) * (n_chrom + 1));
    memcpy(t->h_bin_off, bin_off.data(), sizeof(uint32_t) * (n_chrom + 1));
    if (n_chrom) memcpy(t->h_chrom_size, csize.data(), sizeof(int32_t) * n_chrom);
    tb_tick("upload", &tb0);
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
    free(t->h_chrom_off);
    free(t->h_bin_off);
    free(t->h_chrom_size);
    delete t;
}

extern "C" int itx_table_get_info(const itx_table *t, itx_table_info *o)
{
    if (!t || !o) {
        itx_set_error("itx_table_get_info: null argument");
        return ITX_E_ARG;
    }
    memset(o, 0, sizeof *o);
    o->n_rows = t->n_rows;
    o->n_rep = t->n_rep;
    o->n_fam = t->n_fam;
    o->n_cla = t->n_cla;
    o->cov_len = t->cov_len;
    o->n_units = t->n_units;
    o->n_slots = t->n_slots;
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
