// itx_device.h — device functions shared by the kernels: per-record coordinate derivation
// (generic.c:748-905) and overlap classification (cuskent/binRange.c:196-227 + generic.c:950-970).
#pragma once
#include "itx_common.h"

// flag5 bits (include/iteres_amd.h)
#define F5_PAIRED 1u
#define F5_UNMAP 2u
#define F5_MUNMAP 4u
#define F5_REVERSE 8u
#define F5_READ1 16u
#define F5_NOLOOKUP 32u

#define ITX_WIN 128          // intervals a wave stages in LDS for its tile of records (ItxIv each)

// One record's raw fields as the host decoder hands them over.
struct ItxRaw {
    int32_t tid, pos, tmpend;
    uint32_t mapq, fl;
};

__device__ __forceinline__ uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }

// generic.c:296-301 getCov, with the interval already loaded.
__device__ __forceinline__ float itx_cov(uint32_t start, uint32_t end, int32_t s, int32_t e)
{
    const int32_t qs = (int32_t)start, qe = (int32_t)end;
    int32_t ov = (qe < e ? qe : e) - (qs > s ? qs : s);
    if (ov < 0) ov = 0;
    const float den = (float)(end - start);
    return den == 0.0f ? 0.0f : __fdiv_rn((float)ov, den);
}

__device__ __forceinline__ int32_t clip_ov(int32_t s, int32_t e, int32_t qs, int32_t qe)
{
    return (e < qe ? e : qe) - (s > qs ? s : qs);                 // cuskent/common.c:2824-2831 on the clipped query
}

// Candidate accessors: the same algorithm runs over global memory (any input order) or over the
// wave's LDS window (coordinate-sorted input, the fast path).
struct IvGlobal {
    const ItxIv *iv;
    __device__ __forceinline__ int32_t s(uint32_t k) const { return iv[k].s; }
    __device__ __forceinline__ void sep(uint32_t k, int32_t &s_, int32_t &e_, int32_t &pb) const
    {
        const uint4 v = *reinterpret_cast<const uint4 *>(&iv[k]);
        s_ = (int32_t)v.x; e_ = (int32_t)v.y; pb = (int32_t)v.z;
    }
    __device__ __forceinline__ uint32_t rk(uint32_t k) const { return iv[k].rank; }
};
struct IvLds {
    const uint4 *w;            // entry j at w[2j], w[2j+1]
    __device__ __forceinline__ int32_t s(uint32_t j) const { return (int32_t)w[2 * j].x; }
    __device__ __forceinline__ void sep(uint32_t j, int32_t &s_, int32_t &e_, int32_t &pb) const
    {
        const uint4 v = w[2 * j];
        s_ = (int32_t)v.x; e_ = (int32_t)v.y; pb = (int32_t)v.z;
    }
    __device__ __forceinline__ uint32_t rk(uint32_t j) const { return w[2 * j].w; }
};

// Several hits among candidates [low, hi): replay generic.c:950-970 — walk the hits in the order
// binKeeperFind returns them (list-order rank) and keep the LAST one whose coverage exceeds its predecessor's.
// Returns the chosen candidate or -1 (also when its coverage is below min_cov, generic.c:961-962).
template <class ACC>
__device__ __forceinline__ int32_t itx_pick_multi(const ACC &A, uint32_t low, uint32_t hi, int32_t qs, int32_t qe, uint32_t ustart,
                                               uint32_t uend, float min_cov)
{
    int64_t best_rank = -1;
    float tcov = 0.0f;
    uint32_t chosen = 0;
    for (uint32_t i = low; i < hi; i++) {
        int32_t s, e, pm;
        A.sep(i, s, e, pm);
        if (clip_ov(s, e, qs, qe) <= 0) continue;
        const uint32_t ri = A.rk(i);
        const float ci = itx_cov(ustart, uend, s, e);
        int64_t pr = -1;
        float pc = 0.0f;
        for (uint32_t k = low; k < hi; k++) {
            if (k == i) continue;
            int32_t s2, e2, pm2;
            A.sep(k, s2, e2, pm2);
            if (clip_ov(s2, e2, qs, qe) <= 0) continue;
            const uint32_t rk = A.rk(k);
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
    if (best_rank < 0) return -1;       // tindex == 0 in the reference (cannot happen for positive overlaps)
    if (tcov < min_cov) return -1;
    return (int32_t)chosen;
}

// Picks, among candidates [lo, top) — top = any bound such that every row with s < qe lies below it; rows
// with s >= qe fail the overlap test on their own — the row the reference would pick, or -1. Hits are rows
// with positive clipped overlap (binRange.c:216); the scan walks down while the prefix-max of the ends still
// exceeds qs. One hit is the answer; several go through itx_pick_multi.
template <class ACC>
__device__ __forceinline__ int32_t itx_pick(const ACC &A, uint32_t lo, uint32_t top, int32_t qs, int32_t qe, uint32_t ustart, uint32_t uend,
                                            float min_cov)
{
    int32_t n = 0;
    uint32_t only = 0, low = top;
    int32_t os = 0, oe = 0;
    for (uint32_t k = top; k > lo;) {
        --k;
        int32_t s, e, pb;
        A.sep(k, s, e, pb);
        low = k;
        if (clip_ov(s, e, qs, qe) > 0) {
            n++;
            only = k;
            os = s;
            oe = e;
        }
        if (pb <= qs) break;                                       // nothing below ends past the query start
    }
    if (n == 0) return -1;
    if (n == 1) return itx_cov(ustart, uend, os, oe) < min_cov ? -1 : (int32_t)only;   // generic.c:961-962
    return itx_pick_multi(A, low, top, qs, qe, ustart, uend, min_cov);
}

// The FIRST hit in binKeeperFind's return order (what cpgBedGraphOverlapRepeat takes, generic.c:1084-1088): among the
// rows overlapping [qs, qe) the one with the smallest list-order rank, or -1.
template <class ACC>
__device__ __forceinline__ int32_t itx_pick_first(const ACC &A, uint32_t lo, uint32_t top, int32_t qs, int32_t qe)
{
    int32_t best = -1;
    uint32_t best_rk = 0xffffffffu;
    for (uint32_t k = top; k > lo;) {
        --k;
        int32_t s, e, pb;
        A.sep(k, s, e, pb);
        if (clip_ov(s, e, qs, qe) > 0) {
            const uint32_t rk = A.rk(k);
            if (rk < best_rk) {
                best_rk = rk;
                best = (int32_t)k;
            }
        }
        if (pb <= qs) break;
    }
    return best;
}

// Generic per-lane lookup straight from global memory (any record order). qs/qe already clipped.
__device__ __forceinline__ int32_t itx_classify_lane(const ItxDevTable &T, uint32_t iv_lo, uint32_t bin_base, int32_t qs, int32_t qe,
                                                     uint32_t ustart, uint32_t uend, float min_cov)
{
    // every row with s < qe starts before the end of qe's bin: that bin's upper index bounds the candidates
    const uint32_t top = T.bl[bin_base + ((uint32_t)qe >> T.shift) + 1].x;
    IvGlobal A{T.iv};
    return itx_pick(A, iv_lo, top, qs, qe, ustart, uend, min_cov);
}
__device__ __forceinline__ int32_t itx_first_lane(const ItxDevTable &T, uint32_t iv_lo, uint32_t bin_base, int32_t qs, int32_t qe)
{
    const uint32_t top = T.bl[bin_base + ((uint32_t)qe >> T.shift) + 1].x;
    IvGlobal A{T.iv};
    return itx_pick_first(A, iv_lo, top, qs, qe);
}

// Wave-wide min / max over all 64 lanes with DPP row shifts and row broadcasts (VALU only; a __shfl_xor
// butterfly lowers to six dependent ds_bpermute round trips). Every lane must be active. After the four
// row_shr steps lane 15 of each 16-lane row holds the row's result; row_bcast:15 folds rows 0->1 and 2->3,
// row_bcast:31 folds the lower half into the upper one; lane 63 then holds the wave's result.
#define ITX_DPP_STEP(op, v, id, ctrl, rowmask) v = op(v, __builtin_amdgcn_update_dpp((int)(id), (int)(v), ctrl, rowmask, 0xf, false))
__device__ __forceinline__ int32_t imin32(int32_t a, int32_t b) { return a < b ? a : b; }
__device__ __forceinline__ int32_t imax32(int32_t a, int32_t b) { return a > b ? a : b; }
__device__ __forceinline__ int32_t wave_min_i32(int32_t v)
{
    const int32_t id = 0x7fffffff;
    ITX_DPP_STEP(imin32, v, id, 0x111, 0xf);   // row_shr:1
    ITX_DPP_STEP(imin32, v, id, 0x112, 0xf);   // row_shr:2
    ITX_DPP_STEP(imin32, v, id, 0x114, 0xf);   // row_shr:4
    ITX_DPP_STEP(imin32, v, id, 0x118, 0xf);   // row_shr:8
    ITX_DPP_STEP(imin32, v, id, 0x142, 0xa);   // row_bcast:15 into rows 1 and 3
    ITX_DPP_STEP(imin32, v, id, 0x143, 0xc);   // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int32_t wave_max_i32(int32_t v)
{
    const int32_t id = (int32_t)0x80000000;
    ITX_DPP_STEP(imax32, v, id, 0x111, 0xf);
    ITX_DPP_STEP(imax32, v, id, 0x112, 0xf);
    ITX_DPP_STEP(imax32, v, id, 0x114, 0xf);
    ITX_DPP_STEP(imax32, v, id, 0x118, 0xf);
    ITX_DPP_STEP(imax32, v, id, 0x142, 0xa);
    ITX_DPP_STEP(imax32, v, id, 0x143, 0xc);
    return __builtin_amdgcn_readlane(v, 63);
}

// Runs of equal values over the lanes of a wave (every lane must call). Returns whether this lane starts a
// run; *len = length of the run starting here; *leader = first lane of the run this lane belongs to.
__device__ __forceinline__ bool wave_run(uint32_t v, bool has, uint32_t lane, uint32_t *len, uint32_t *leader)
{
    const uint32_t pv = (uint32_t)__shfl_up((int32_t)v, 1, 64);
    const unsigned long long m_has = __ballot(has);
    const bool prev_has = lane != 0 && ((m_has >> (lane - 1)) & 1ull);
    const bool st = has && (!prev_has || pv != v);            // a lane without a value always breaks the run
    const unsigned long long m_st = __ballot(st);
    const unsigned long long below_eq = (((1ull << lane) - 1ull) << 1) | 1ull;          // lanes <= this one
    const unsigned long long stop = (m_st | ~m_has) & ~below_eq;                         // next run start / gap above
    const uint32_t e = stop ? (uint32_t)__ffsll((long long)stop) - 1u : 64u;
    *len = e - lane;
    const unsigned long long mine = m_st & below_eq;                                     // highest start at or below
    *leader = mine ? 63u - (uint32_t)__clzll((long long)mine) : lane;
    return st;
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
