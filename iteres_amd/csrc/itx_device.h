// itx_device.h — device functions shared by every kernel: per-record coordinate derivation
// (generic.c:748-905) and overlap classification (cuskent/binRange.c:196-227 + generic.c:950-970).
#pragma once
#include "itx_common.h"

// flag5 bits (include/iteres_amd.h)
#define F5_PAIRED 1u
#define F5_UNMAP 2u
#define F5_MUNMAP 4u
#define F5_REVERSE 8u
#define F5_READ1 16u

// What one record contributes to cnt[] (generic.c:1048-1060), as a bit set, plus its interval.
// bit k set => cnt[k] += 1.  (cnt[8] and cnt[12] are never touched on the device.)
struct ItxDerived {
    uint32_t cntbits;
    uint32_t start, end;   // the reference's unsigned start/end
    int32_t  chrom;        // >= 0 when the record goes on to the lookup
    bool     uniq;         // MAPQ >= -Q
};

__device__ __forceinline__ uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }

// generic.c:748-922 for one record. Returns with chrom < 0 when the record is dropped before the lookup.
__device__ __forceinline__ ItxDerived itx_derive(const ItxRunParams &P, const ItxDevTable &T, const ItxDevBatch &B, size_t i)
{
    ItxDerived d;
    d.chrom = -1;
    d.start = d.end = 0;
    const uint32_t fl = B.flag5[i];
    const uint32_t qual = B.mapq[i];
    d.uniq = qual >= P.mapq_min;
    // generic.c:748-759: which "read end" counter
    const bool end1 = !(fl & F5_PAIRED) || (fl & F5_READ1) || P.treat;
    d.cntbits = end1 ? 1u : 2u;
    if (fl & F5_UNMAP) return d;                                  // generic.c:764
    d.cntbits |= end1 ? (1u << 2) : (1u << 3);                   // generic.c:768-779
    const int32_t tid = B.tid[i];
    // generic.c:781-801; a tid outside the header crashes the reference, here it is "unknown chromosome"
    const int32_t chrom = (tid >= 0 && tid < P.n_tid) ? P.tid2chrom[tid] : -1;
    if (chrom < 0) return d;
    const uint32_t cend = (uint32_t)(T.chrom_size[chrom] - 1);   // generic.c:796
    if (cend == 1u) return d;                                    // generic.c:797
    d.cntbits |= end1 ? (1u << 4) : (1u << 5);                   // generic.c:802-813
    bool se_style;
    if (P.treat) {
        se_style = true;
    } else if (fl & F5_PAIRED) {
        if (!(fl & F5_MUNMAP)) {
            if (!(fl & F5_READ1)) return d;                      // generic.c:858-860
            const int32_t isz = B.isize[i];
            const uint32_t a = (uint32_t)(isz < 0 ? -isz : isz);
            if (a > P.isize_max || isz == 0) return d;           // generic.c:839-840
            se_style = false;
        } else {
            if (P.discard) return d;                             // generic.c:862-863
            se_style = true;
        }
    } else {
        se_style = true;
    }
    d.cntbits |= (1u << 6);                                       // reads_mapped
    if (d.uniq) d.cntbits |= (1u << 7) | (1u << 11);             // reads_mapped_unique, reads_nonredundant_unique (no -R here)
    uint32_t start, end;
    if (se_style) {                                               // generic.c:819-833
        start = (uint32_t)B.pos[i];
        end = umin32(cend, (uint32_t)B.tmpend[i]);
        if (P.extension) {
            if (!(fl & F5_REVERSE)) {
                end = umin32(start + P.extension, cend);
            } else {
                start = (end < P.extension) ? 0u : end - P.extension;
            }
        }
    } else {                                                      // generic.c:845-855
        const int32_t isz = B.isize[i];
        if (isz > 0) {
            start = (uint32_t)B.pos[i];
            end = umin32(cend, start + (uint32_t)isz);
        } else {
            start = (uint32_t)B.mpos[i];
            end = umin32(cend, start - (uint32_t)isz);
        }
    }
    d.start = start;
    d.end = end;
    d.chrom = chrom;
    return d;
}

// generic.c:296-301 getCov, with the interval already loaded.
__device__ __forceinline__ float itx_cov(uint32_t start, uint32_t end, int32_t s, int32_t e)
{
    const int32_t qs = (int32_t)start, qe = (int32_t)end;
    int32_t ov = (qe < e ? qe : e) - (qs > s ? qs : s);
    if (ov < 0) ov = 0;
    const float den = (float)(end - start);
    return den == 0.0f ? 0.0f : __fdiv_rn((float)ov, den);
}

// Returns the SORTED index of the row the reference would pick for [start,end) on `chrom`, or -1.
// Candidates: rows with s < end' are [chrom_lo, hi); hi comes from the binned start index, then the
// scan walks down while the prefix-max of the ends still exceeds start'. Hits are rows with positive
// clipped overlap (binRange.c:216). With one hit it is the answer; with several, the reference's rule
// "last hit, in list order, whose coverage exceeds the previous hit's" (generic.c:955-959) is replayed
// through the precomputed list-order ranks.
__device__ __forceinline__ int32_t itx_classify(const ItxDevTable &T, int32_t chrom, uint32_t ustart, uint32_t uend, float min_cov)
{
    int32_t qs = (int32_t)ustart, qe = (int32_t)uend;              // binKeeperFind(bk, int start, int end)
    const int32_t maxPos = T.chrom_size[chrom];
    if (qs < 0) qs = 0;                                            // binRange.c:204-206
    if (qe > maxPos) qe = maxPos;
    if (qs >= qe) return -1;
    const uint32_t lo = T.chrom_off[chrom];
    if (lo == T.chrom_off[chrom + 1]) return -1;
    const uint32_t *bi = T.bidx + T.bin_off[chrom] + ((uint32_t)qe >> T.shift);
    uint32_t hi = bi[0];
    uint32_t top = bi[1];
    if (top - hi > 8) {                                            // crowded bin: binary search for first s >= qe
        uint32_t a = hi, b = top;
        while (a < b) {
            uint32_t m = (a + b) >> 1;
            if (T.iv[m].s < qe) a = m + 1; else b = m;
        }
        hi = a;
    } else {
        while (hi < top && T.iv[hi].s < qe) hi++;
    }
    int32_t n = 0;
    uint32_t only = 0, low = hi;
    for (uint32_t k = hi; k > lo;) {
        --k;
        const int32_t pm = T.iv[k].pmax_e;
        if (pm <= qs) break;
        low = k;
        const int32_t s = T.iv[k].s, e = T.iv[k].e;
        const int32_t ov = (e < qe ? e : qe) - (s > qs ? s : qs);
        if (ov > 0) {
            n++;
            only = k;
        }
    }
    if (n == 0) return -1;
    uint32_t chosen = only;
    float tcov;
    if (n == 1) {
        tcov = itx_cov(ustart, uend, T.iv[only].s, T.iv[only].e);
    } else {
        // several hits: for each hit find its predecessor in list order among the hits
        int64_t best_rank = -1;
        tcov = 0.0f;
        for (uint32_t i = low; i < hi; i++) {
            const int32_t s = T.iv[i].s, e = T.iv[i].e;
            if (((e < qe ? e : qe) - (s > qs ? s : qs)) <= 0) continue;
            const uint32_t ri = T.rank[i];
            const float ci = itx_cov(ustart, uend, s, e);
            int64_t pr = -1;
            float pc = 0.0f;
            for (uint32_t k = low; k < hi; k++) {
                if (k == i) continue;
                const int32_t s2 = T.iv[k].s, e2 = T.iv[k].e;
                if (((e2 < qe ? e2 : qe) - (s2 > qs ? s2 : qs)) <= 0) continue;
                const uint32_t rk = T.rank[k];
                if (rk < ri && (int64_t)rk > pr) {
                    pr = rk;
                    pc = itx_cov(ustart, uend, s2, e2);
                }
            }
            if (ci > pc && (int64_t)ri > best_rank) {
                best_rank = ri;
                chosen = i;
                tcov = ci;
            }
        }
        if (best_rank < 0) return -1;   // tindex == 0 in the reference (cannot happen for positive overlaps)
    }
    if (tcov < min_cov) return -1;                                 // generic.c:961-962
    return (int32_t)chosen;
}

// Consensus range a classified read increments (generic.c:991-1007) in slot space:
// returns n (number of consensus positions) and sets *first to the first slot; n == 0 means the read
// is counted but adds no coverage.
__device__ __forceinline__ uint32_t itx_cov_range(const ItxIv &r, uint32_t start, uint32_t end, uint32_t *first)
{
    const uint32_t qlen = end - start;
    const uint32_t rstart = start - (uint32_t)r.s;                 // wraps when the read starts left of the repeat
    uint32_t rend = rstart + qlen;
    rend = rend < (uint32_t)r.e ? rend : (uint32_t)r.e;            // clamp against the GENOMIC end (reference quirk)
    const uint32_t j0 = rstart + r.cs;
    uint32_t n = 0;
    if (rstart < rend && j0 < r.jcap) {
        const uint32_t a = rend - rstart, b = r.jcap - j0;
        n = a < b ? a : b;
    }
    *first = r.covslot + j0;
    return n;
}
