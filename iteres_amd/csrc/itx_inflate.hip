// itx_inflate.hip — BGZF blocks inflated on the device: one wavefront per block (itx_inflate_core.h), thousands of blocks
// per launch. Replaces the reference's sequential zlib inflate of every 64 KiB block (cussamtools/bgzf.c:367-397
// inflate_block, reached from bgzf_read -> bgzf_read_block, bgzf.c:425-521): only compressed bytes cross PCIe on the way in.
//
// HBM layout: the caller's compressed chunk is mirrored at the same offsets in d_comp (so a block's position keeps its
// alignment), the inflated bytes of all blocks are contiguous in d_out at the offsets the caller computed from the ISIZE
// trailers. A call is cut into groups of blocks that alternate between two streams: copy-in, kernel and copy-out of one
// group overlap the neighbours'.
#include "itx_common.h"

#define ITXI_WAVE 64u
#define ITXI_SIMPLE_IN          /* one word of input look-ahead: 10.96 ms per 24 k blocks against 11.27 with the 16-byte FIFO */
#define ITXI_FN static __device__ inline
#define ITXI_UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(x)))
#define ITXI_BCAST(v, j) ((uint32_t)__builtin_amdgcn_readlane((int32_t)(v), (int32_t)(j)))
static __device__ inline uint32_t itxi_scan_add(uint32_t v, uint32_t lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int32_t)v, o, 64);
        v += lane >= (uint32_t)o ? t : 0u;
    }
    return v;
}
#define ITXI_SCAN_ADD(v, lane) itxi_scan_add(v, lane)
#define ITXI_LANE_READ(v, j) ((uint32_t)__builtin_amdgcn_ds_bpermute((int32_t)((j) << 2), (int32_t)(v)))
#define ITXI_BALLOT(p) ((uint64_t)__ballot(p))
#define ITXI_MBCNT(m, lane) ((uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)((m) >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)(m), 0u)))
#define ITXI_LDS_OR(ptr, v) ((void)atomicOr((ptr), (v)))
#define ITXI_AT(p, i) (p)[(i) * 64u + ln]          /* a decoder's table element i: lane-interleaved (bank = lane) */
#define ITXI_BITREV32(x) __builtin_bitreverse32(x)
typedef uint16_t itxi_u16x2 __attribute__((ext_vector_type(2)));
static __device__ inline uint32_t itxi_pksign16(uint32_t a, uint32_t b)
{
    const itxi_u16x2 d = (__builtin_bit_cast(itxi_u16x2, a) - __builtin_bit_cast(itxi_u16x2, b)) >> (itxi_u16x2)15;     // v_pk_sub_u16, v_pk_lshrrev_b16
    return __builtin_bit_cast(uint32_t, d);
}
#define ITXI_PKSIGN16(a, b) itxi_pksign16(a, b)
#define ITXI_LOADW(w, i) ((w)[(i)])
#define ITXI_LOAD4(WORDS, AT, R0, R1, R2, R3)                                    \
    do {                                                                         \
        const uint4 v4__ = *reinterpret_cast<const uint4 *>((WORDS) + (AT));     \
        (R0) = v4__.x;                                                           \
        (R1) = v4__.y;                                                           \
        (R2) = v4__.z;                                                           \
        (R3) = v4__.w;                                                           \
    } while (0)
#define ITXI_LOADB(p, i) ((p)[(i)])
// A far match reads bytes this wave stored earlier through other lanes. Workgroup scope is all it takes — the wave's
// stores have to be acknowledged before its loads go out (s_waitcnt vmcnt(0)); the lines read are whole stripes written
// once, so no cached copy can be stale. (An agent-scope fence here writes the L2 back: measured 4 500 cycles per token.)
#define ITXI_FENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup")
#include "itx_inflate_core.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define BGZF_HEADER 18u      /* gzip header with the one "BC" extra field (bgzf.c:401-411) */
#define BGZF_TRAILER 8u      /* CRC32 + ISIZE */
#define SCR_STRIDE ITXI_REGION                    /* bytes of scratch per block: literals from its bottom, tokens from its top (itx_inflate_core.h) */

// pass 1: lane = block. meta[3b] = status, [3b+1] = literals, [3b+2] = tokens
__global__ __launch_bounds__(64) void k_tokens(const uint32_t *__restrict__ comp, const itx_bgzf_block *__restrict__ blk, uint32_t n, uint8_t *__restrict__ scr,
                                               uint32_t *__restrict__ meta)
{
#ifndef ITXI_SYM16
    __shared__ uint32_t s_lhi[9 * 64];                                   // 26 880 bytes in all: six of these workgroups to a CU
    __shared__ uint16_t s_loffs[16 * 64], s_doffs[16 * 64];
    __shared__ uint8_t s_lsym8[288 * 64], s_dsym[32 * 64];
#else
    __shared__ uint16_t s_lsym16[288 * 64], s_loffs[16 * 64], s_doffs[16 * 64];      // 43 008 bytes: three to a CU (experiment builds)
    __shared__ uint8_t s_dsym[32 * 64];
#endif
    const uint32_t ln = threadIdx.x, b = blockIdx.x * 64u + ln;
    if (b >= n) return;
    const uint32_t coff = blk[b].coff, csize = blk[b].csize, usize = blk[b].usize;
#ifndef ITXI_SYM16
    ItxiTab T{s_lsym8, s_lhi, s_dsym, s_loffs, s_doffs};
#else
    ItxiTab T{s_lsym16, s_dsym, s_loffs, s_doffs};
#endif
    uint8_t *region = scr + (size_t)b * SCR_STRIDE;
    ItxiTokens K{region, reinterpret_cast<uint32_t *>(region + SCR_STRIDE), 0, 0};
    int rc = ITXI_E_INPUT;
    if (csize >= BGZF_HEADER + BGZF_TRAILER + 2u && usize <= ITXI_MAX_BLOCK)
        rc = itxi_tokens(T, ln, comp, coff + BGZF_HEADER, coff + csize - BGZF_TRAILER, usize, K);
    meta[3 * b] = (uint32_t)rc;
    meta[3 * b + 1] = K.n_lit;
    meta[3 * b + 2] = K.n_tok;
}

// pass 2: wave = block
__global__ __launch_bounds__(64) void k_resolve(const itx_bgzf_block *__restrict__ blk, uint32_t first, uint32_t n, const uint8_t *__restrict__ scr,
                                                const uint32_t *__restrict__ meta, uint8_t *__restrict__ out, uint8_t *__restrict__ status)
{
    __shared__ uint32_t s_mem[(ITXI_RING + ITXI_LSTAGE + ITXI_BMAP / 8u + 8u) / 4];       // the ring, right behind it the literal stage, then the bitmap of token starts
    uint32_t *s_ring = s_mem, *s_stage = s_mem + ITXI_RING / 4, *s_bmap = s_mem + (ITXI_RING + ITXI_LSTAGE) / 4;
    if (blockIdx.x >= n) return;
    const uint32_t b = first + blockIdx.x;
    int rc = (int)meta[3 * b];
    if (rc == ITXI_OK) {
        const uint8_t *region = scr + (size_t)b * SCR_STRIDE;
        rc = itxi_resolve(s_ring, s_stage, s_bmap, region, reinterpret_cast<const uint32_t *>(region + SCR_STRIDE), meta[3 * b + 1], meta[3 * b + 2], out, blk[b].uoff,
                          blk[b].usize, threadIdx.x);
    }
    if (threadIdx.x == 0) status[b] = (uint8_t)rc;
}

// =====================================================================================================================
// The decoded stream stays on the device: a WINDOW of inflated bytes per chunk, the BAM records in it located and
// parsed there (cussamtools/bam.c:179-210 bam_read1 + the fields generic.c:745-905 takes from bam1_core_t, bam.h:169-177),
// and only the 26-byte-per-record SoA crosses PCIe on the way back.
//
// Locating the records is a chain (a record's length says where the next one starts). The window is cut into pieces; a
// thread per piece GUESSES the first record start in its piece (three well-formed records in a row) and walks on from
// there; the host then checks the pieces in stream order — a piece counts only if its guess is exactly where the chain
// of the pieces before it arrives, otherwise it is walked again from there (k_rewalk) — which makes the result exact
// whatever the guesses were (same scheme as the host reader's locate_records, iteres_amd/host/bamio.c).

#define WIN_HEAD (4u << 20)          /* room before a window's new bytes for the unconsumed tail of the previous one */
#define PIECE (16u << 10)            /* bytes per located piece */
#define PIECE_SLOTS (PIECE / 36u + 2u)

struct PieceSum {
    uint32_t c, end, n, why;         /* guessed start (0xffffffff none), where the walk stopped, starts found; why as the host reader */
};

static __device__ inline uint32_t ld32(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }

static __device__ inline uint32_t walk(const uint8_t *u, uint32_t p, uint32_t lim, uint32_t L, uint32_t *off, uint32_t *pn, uint32_t *why)
{
    uint32_t n = 0;
    while (p < lim) {
        if (p + 4u > L) {
            *why = 1;
            *pn = n;
            return p;
        }
        const int32_t bl = (int32_t)ld32(u + p);
        if (bl < 32) {                                            /* bam.c:186-190: a malformed length ends the file */
            *why = 2;
            *pn = n;
            return p;
        }
        if ((uint64_t)p + 4u + (uint32_t)bl > L) {
            *why = 1;
            *pn = n;
            return p;
        }
        off[n++] = p;
        p += 4u + (uint32_t)bl;
    }
    *why = 0;
    *pn = n;
    return p;
}

static __device__ inline uint32_t looks_like_record(const uint8_t *u, uint32_t p, uint32_t L, int32_t n_targets)
{
    if ((uint64_t)p + 36u > L) return 0;
    const int32_t bl = (int32_t)ld32(u + p);
    if (bl < 32 || (uint64_t)p + 4u + (uint32_t)bl > L) return 0;
    const uint8_t *core = u + p + 4;
    const int32_t tid = (int32_t)ld32(core), pos = (int32_t)ld32(core + 4), l_qseq = (int32_t)ld32(core + 16), mtid = (int32_t)ld32(core + 20),
                  mpos = (int32_t)ld32(core + 24);
    const uint32_t x1 = ld32(core + 8), x2 = ld32(core + 12);
    const uint32_t l_qname = x1 & 0xffu, n_cigar = x2 & 0xffffu;
    if (tid < -1 || tid >= n_targets || mtid < -1 || mtid >= n_targets || pos < -1 || mpos < -1 || l_qseq < 0 || l_qname == 0) return 0;
    const uint64_t need = (uint64_t)l_qname + 4ull * n_cigar + ((uint64_t)l_qseq + 1) / 2 + (uint64_t)l_qseq;
    if (need > (uint64_t)bl - 32u) return 0;
    if (u[p + 36u + l_qname - 1u] != 0) return 0;
    return 4u + (uint32_t)bl;
}

__global__ __launch_bounds__(64) void k_guess(const uint8_t *__restrict__ u, uint32_t p0, uint32_t L, uint32_t T, int32_t n_targets, PieceSum *__restrict__ sum,
                                              uint32_t *__restrict__ spec)
{
    const uint32_t t = blockIdx.x * 64u + threadIdx.x;
    if (t >= T) return;
    const uint32_t b = p0 + t * PIECE, lim = L - b > PIECE ? b + PIECE : L;       // T = ceil((L - p0) / PIECE): b < L
    uint32_t c = 0xffffffffu;
    if (t == 0) {
        c = b;
    } else {
        for (uint32_t x = b; x < lim; x++) {
            uint32_t a = x, k = 0;
            while (k < 3 && a < L) {
                const uint32_t step = looks_like_record(u, a, L, n_targets);
                if (!step) break;
                a += step;
                k++;
            }
            if (k == 3 || (k > 0 && a >= L)) {
                c = x;
                break;
            }
        }
    }
    PieceSum s{c, c, 0, 0};
    if (c != 0xffffffffu) s.end = walk(u, c, lim, L, spec + (size_t)t * PIECE_SLOTS, &s.n, &s.why);
    sum[t] = s;
}

// one piece walked again from where the chain really arrives (one thread: it happens when a guess was misled)
__global__ void k_rewalk(const uint8_t *__restrict__ u, uint32_t from, uint32_t lim, uint32_t L, uint32_t t, PieceSum *__restrict__ sum, uint32_t *__restrict__ spec)
{
    if (threadIdx.x || blockIdx.x) return;
    PieceSum s{from, from, 0, 0};
    s.end = walk(u, from, lim, L, spec + (size_t)t * PIECE_SLOTS, &s.n, &s.why);
    sum[t] = s;
}

// accepted pieces' starts, in stream order, into one array: base[t] = records before piece t, cnt[t] = its records
__global__ __launch_bounds__(64) void k_compact(const uint32_t *__restrict__ spec, const uint32_t *__restrict__ base, const uint32_t *__restrict__ cnt, uint32_t T,
                                                uint32_t *__restrict__ rec_off)
{
    const uint32_t t = blockIdx.x;
    if (t >= T) return;
    const uint32_t n = cnt[t], b = base[t];
    for (uint32_t j = threadIdx.x; j < n; j += 64u) rec_off[b + j] = spec[(size_t)t * PIECE_SLOTS + j];
}

// cussamtools/bam_aux.c:36-48 walk, as the host reader's aux_find: is there an XA tag?
static __device__ inline bool has_xa(const uint8_t *s, const uint8_t *end)
{
    while (s + 3 <= end) {
        if (s[0] == 'X' && s[1] == 'A') return true;
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
            if (s + 5 > end) return false;
            uint32_t sub = s[0];
            if (sub >= 'a' && sub <= 'z') sub -= 32u;
            const uint32_t cnt = ld32(s + 1);
            const uint32_t esz = (sub == 'C' || sub == 'A') ? 1u : (sub == 'S') ? 2u : 4u;
            if ((uint64_t)cnt * esz > (uint64_t)(end - s)) return false;
            s += 5u + cnt * esz;
        } else
            return false;
    }
    return false;
}

// one thread per record: the fields generic.c:745-905 reads, bam_calend (bam.c:17-27: M, D, N advance) or pos + l_qseq
__global__ __launch_bounds__(256) void k_parse(const uint8_t *__restrict__ u, const uint32_t *__restrict__ rec_off, uint32_t n, int32_t *__restrict__ tid_o,
                                               int32_t *__restrict__ pos_o, int32_t *__restrict__ end_o, uint8_t *__restrict__ mapq_o, uint8_t *__restrict__ f5_o,
                                               int32_t *__restrict__ mpos_o, int32_t *__restrict__ isize_o, uint8_t *__restrict__ xa_o, uint32_t *__restrict__ flags,
                                               uint8_t *__restrict__ seen, int32_t n_targets)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    bool paired = false, xa = false;
    if (i < n) {
        const uint8_t *p = u + rec_off[i];
        const uint32_t block_len = ld32(p);
        const uint8_t *core = p + 4, *data = p + 36;
        const uint32_t dlen = block_len - 32u;
        const int32_t tid = (int32_t)ld32(core), pos = (int32_t)ld32(core + 4);
        const uint32_t x1 = ld32(core + 8), x2 = ld32(core + 12);
        const uint32_t l_qname = x1 & 0xffu, qual = (x1 >> 8) & 0xffu, flag = x2 >> 16, n_cigar = x2 & 0xffffu;
        const int32_t l_qseq = (int32_t)ld32(core + 16), mpos = (int32_t)ld32(core + 24), isize = (int32_t)ld32(core + 28);
        int32_t tmpend;
        if (n_cigar && (uint64_t)l_qname + 4ull * n_cigar <= dlen) {
            uint32_t e = (uint32_t)pos;
            const uint8_t *cg = data + l_qname;
            for (uint32_t k = 0; k < n_cigar; k++) {
                const uint32_t c = ld32(cg + 4 * k), op = c & 0xfu;
                if (op == 0 || op == 2 || op == 3) e += c >> 4;
            }
            tmpend = (int32_t)e;
        } else {
            tmpend = (int32_t)((uint32_t)pos + (uint32_t)l_qseq);             /* generic.c:820 */
        }
        tid_o[i] = tid;
        pos_o[i] = pos;
        end_o[i] = tmpend;
        mapq_o[i] = (uint8_t)qual;
        f5_o[i] = ITX_FLAG5(flag);
        mpos_o[i] = mpos;
        isize_o[i] = isize;
        paired = flag & 1u;
        if (!(flag & 4u) && tid >= 0 && tid < n_targets) seen[tid] = 1;          /* a mapped record on this reference (generic.c:764,781) */
        const uint64_t ql = l_qseq > 0 ? (uint64_t)l_qseq : 0;
        const uint64_t off = (uint64_t)l_qname + 4ull * n_cigar + (ql + 1) / 2 + ql;
        xa = off < dlen && has_xa(data + off, data + dlen);
        xa_o[i] = xa ? 1 : 0;
    }
    const unsigned long long bp = __ballot(paired), bx = __ballot(xa);
    if ((threadIdx.x & 63u) == 0 && (bp | bx)) atomicOr(flags, (bp ? 1u : 0u) | (bx ? 2u : 0u));
}

// bytes [src, src + n) of a buffer moved UP by `shift` bytes (0 < shift; the ranges overlap): ONE workgroup walks the range from
// its end in chunks; a chunk is loaded whole, then — behind a barrier — stored: its destination lies above every source byte that is
// still to be read. For shifts too small to move the bytes in a few device-to-device copies (itx_bamwin_carry).
#define MOVE_THREADS 1024u
#define MOVE_PER_THREAD 64u                  /* bytes per thread and chunk */
__global__ __launch_bounds__(MOVE_THREADS) void k_move_up(uint8_t *__restrict__ buf, size_t src, size_t n, size_t shift)
{
    const size_t chunk = (size_t)MOVE_THREADS * MOVE_PER_THREAD;
    for (size_t left = n; left > 0;) {
        const size_t m = left < chunk ? left : chunk, c0 = src + left - m;
        uint8_t v[MOVE_PER_THREAD];
        const size_t a = (size_t)threadIdx.x * MOVE_PER_THREAD;
        uint32_t k_n = 0;
        if (a < m) k_n = (uint32_t)(m - a < MOVE_PER_THREAD ? m - a : MOVE_PER_THREAD);
#pragma unroll
        for (uint32_t k = 0; k < MOVE_PER_THREAD; k++) v[k] = k < k_n ? buf[c0 + a + k] : (uint8_t)0;
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < MOVE_PER_THREAD; k++)
            if (k < k_n) buf[c0 + a + k + shift] = v[k];
        __syncthreads();
        left -= m;
    }
}

// launched once when an inflater is made: loading the library's code objects costs a tenth of a second, and the helper
// thread that creates the inflater has it to spare
__global__ void k_warm(uint32_t *p)
{
    if (p) *p = 0;
}

static void report_at_exit();

// pushes an inflater can hold in flight (a SLOT each: copy stream, events, compressed bytes, block list, status): ITX_PUSHES,
// 1 .. ITX_BAMWIN_LANES
static int lanes_in_use()
{
    static int v;
    if (!v) {
        const char *e = getenv("ITX_PUSHES");
        const int x = e ? atoi(e) : 0;
        v = x >= 1 && x <= ITX_BAMWIN_LANES ? x : ITX_BAMWIN_LANES_DEFAULT;
    }
    return v;
}
// ... and the COMPUTE lanes their kernels run on (a stream and the 88 KB of token scratch per block each): ITX_LANES, at most
// the slots. Slot s computes on lane s % n. A push used to copy and compute on one stream: a lane was busy copy + pass 1 + pass 2
// = 46 ms per push of which the kernels are 34 — on average 2.6 of 4 pass-1 kernels were running (profiles/r03_cli_500M_trace_*).
// With more slots than lanes the next push's bytes cross PCIe while the lane still computes; more LANES instead cost a gigabyte
// of scratch each and their kernels crowd the engine's out of the CUs (k_hist 0.67 -> 3.1 ms with 8).
#define ITX_COMPUTE_LANES_DEFAULT 4
static int compute_lanes()
{
    static int v;
    if (!v) {
        const char *e = getenv("ITX_LANES");
        const int x = e ? atoi(e) : 0;
        v = x >= 1 && x <= ITX_BAMWIN_LANES ? x : ITX_COMPUTE_LANES_DEFAULT;
        if (v > lanes_in_use()) v = lanes_in_use();
    }
    return v;
}

struct itx_inflater {
    int device;
    hipStream_t st[2];
    uint8_t *arena;                        // itx_inflater_reserve: the one allocation the lanes' and the first n_reserved_win windows' buffers are cut from
    int n_reserved_win;
    int n_cu;
    uint8_t *d_comp, *d_out, *d_status, *d_lit;      // d_lit: the blocks' scratch regions (SCR_STRIDE bytes each)
    uint32_t *d_meta;
    itx_bgzf_block *d_blk;
    size_t comp_cap, out_cap, status_cap, blk_cap, lit_cap, meta_cap;
    hipEvent_t ev[4];
    float ms_tokens, ms_resolve, ms_resolve_all;
    hipEvent_t ev_res_end[2];
    // windows of inflated bytes that stay on the device (itx_bamwin_*)
    struct {
        uint8_t *buf;
        size_t cap;
        uint32_t start, len, consumed;     // unconsumed bytes are buf[start, len); consumed: end of the last parsed record
    } win[ITX_BAMWIN_WINDOWS];
    // a push in flight: its own stream and scratch, so that the Huffman pass of one chunk runs beside the replay of the last
    hipStream_t copy_st;
    struct {
        hipEvent_t copied;                 // the compressed bytes have left the caller's buffer
        hipEvent_t ev[3];                  // ITX_TIMING: before pass 1, between the passes, after pass 2
        hipEvent_t p1_done, done;          // pass 1 through (the shared pass-2 stream waits for it); the whole push through
        uint8_t *d_comp, *d_status, *h_status;
        itx_bgzf_block *d_blk, *h_blk;     // h_blk (page-locked): the block list shifted to the window's offsets
        size_t comp_cap, status_cap, blk_cap, h_cap;
        size_t n_blk, total;
        int busy;
    } lane[ITX_BAMWIN_LANES];
    // a compute lane: the stream the two passes of its slots' pushes run on, one after the other, and their scratch
    struct {
        hipStream_t cst;
        uint8_t *d_lit;
        uint32_t *d_meta;
        size_t lit_cap, meta_cap;
    } clane[ITX_BAMWIN_LANES];
    void *d_sum, *h_sum;                   // PieceSum per piece, and its host copy
    uint32_t *d_spec, *d_pb, *h_pb, *d_recoff, *d_flags;
    uint8_t *d_seen;
    size_t seen_cap;
    size_t sum_cap, spec_cap, pb_cap, recoff_cap, soa_cap;
    int32_t *d_tid, *d_pos, *d_end, *d_mpos, *d_isize;
    uint8_t *d_mapq, *d_f5, *d_xa;
    int parsed_w;
    size_t n_rec;
};

#define INF_HIP(call)                                                                                     \
    do {                                                                                                  \
        hipError_t err__ = (call);                                                                        \
        if (err__ != hipSuccess) {                                                                        \
            itx_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(err__), __FILE__, __LINE__); \
            return ITX_E_NO_DEVICE;                                                                       \
        }                                                                                                 \
    } while (0)

extern "C" int itx_inflater_create(int device, itx_inflater **out)
{
    if (!out) return ITX_E_ARG;
    *out = nullptr;
    const bool tim = getenv("ITX_TIMING_SETUP") != nullptr;
    struct timespec tsa, tsb;
#define SETUP_TICK(what)                                                                                                      \
    do {                                                                                                                      \
        if (tim) {                                                                                                            \
            clock_gettime(CLOCK_MONOTONIC, &tsb);                                                                             \
            fprintf(stderr, "[itx timing] inflater setup: %s %.3f s\n", what, (double)(tsb.tv_sec - tsa.tv_sec) + 1e-9 * (double)(tsb.tv_nsec - tsa.tv_nsec)); \
            tsa = tsb;                                                                                                        \
        }                                                                                                                     \
    } while (0)
    clock_gettime(CLOCK_MONOTONIC, &tsa);
    INF_HIP(hipSetDevice(device));
    SETUP_TICK("hipSetDevice");
    itx_inflater *h = (itx_inflater *)calloc(1, sizeof *h);
    if (!h) return ITX_E_NOMEM;
    h->device = device;
    for (int k = 0; k < 2; k++) INF_HIP(hipStreamCreateWithFlags(&h->st[k], hipStreamNonBlocking));
    SETUP_TICK("first two streams");
    for (int k = 0; k < 4; k++) INF_HIP(hipEventCreate(&h->ev[k]));
    for (int k = 0; k < 2; k++) INF_HIP(hipEventCreate(&h->ev_res_end[k]));
    for (int k = 0; k < lanes_in_use(); k++) {
        INF_HIP(hipEventCreateWithFlags(&h->lane[k].copied, hipEventDisableTiming));
        for (int q = 0; q < 3; q++) INF_HIP(hipEventCreate(&h->lane[k].ev[q]));
        INF_HIP(hipEventCreateWithFlags(&h->lane[k].p1_done, hipEventDisableTiming));
        INF_HIP(hipEventCreateWithFlags(&h->lane[k].done, hipEventDisableTiming));
    }
    // ONE stream carries every push's bytes across PCIe, in push order (the link is one: copies side by side only finish later,
    // all of them); a stream per slot also meant more streams than hardware queues, and streams that share a queue run one
    // after the other — two compute lanes on one queue halved pass 1's overlap (1.8 kernels in flight instead of 3)
    INF_HIP(hipStreamCreateWithFlags(&h->copy_st, hipStreamNonBlocking));
    for (int k = 0; k < compute_lanes(); k++) INF_HIP(hipStreamCreateWithFlags(&h->clane[k].cst, hipStreamNonBlocking));
    SETUP_TICK("lane streams, events, counters");
    h->n_cu = 256;
    if (hipDeviceGetAttribute(&h->n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || h->n_cu <= 0) h->n_cu = 256;
    if (getenv("ITX_TIMING")) {
        static bool once;
        if (!once) {
            once = true;
            atexit(report_at_exit);
        }
    }
    hipLaunchKernelGGL(k_warm, dim3(1), dim3(1), 0, h->st[0], (uint32_t *)nullptr);
    INF_HIP(hipGetLastError());
    INF_HIP(hipStreamSynchronize(h->st[0]));
    SETUP_TICK("first kernel (code object load)");
#undef SETUP_TICK
    *out = h;
    return ITX_OK;
}

extern "C" void itx_inflater_destroy(itx_inflater *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    for (int k = 0; k < 2; k++)
        if (h->st[k]) {
            (void)hipStreamSynchronize(h->st[k]);
            (void)hipStreamDestroy(h->st[k]);
        }
    (void)hipFree(h->d_comp);
    (void)hipFree(h->d_out);
    (void)hipFree(h->d_status);
    (void)hipFree(h->d_blk);
    (void)hipFree(h->d_lit);
    (void)hipFree(h->d_meta);
    for (int k = h->arena ? h->n_reserved_win : 0; k < ITX_BAMWIN_WINDOWS; k++) (void)hipFree(h->win[k].buf);
    for (int k = 0; k < ITX_BAMWIN_LANES; k++) {
        if (h->lane[k].copied) (void)hipEventDestroy(h->lane[k].copied);
        for (int q = 0; q < 3; q++)
            if (h->lane[k].ev[q]) (void)hipEventDestroy(h->lane[k].ev[q]);
        if (h->lane[k].p1_done) (void)hipEventDestroy(h->lane[k].p1_done);
        if (h->lane[k].done) (void)hipEventDestroy(h->lane[k].done);
        if (h->clane[k].cst) {
            (void)hipStreamSynchronize(h->clane[k].cst);
            (void)hipStreamDestroy(h->clane[k].cst);
        }
        if (!h->arena) {
            (void)hipFree(h->lane[k].d_comp);
            (void)hipFree(h->lane[k].d_status);
            (void)hipFree(h->clane[k].d_lit);
            (void)hipFree(h->clane[k].d_meta);
            (void)hipFree(h->lane[k].d_blk);
        }
        if (h->lane[k].h_blk) (void)hipHostFree(h->lane[k].h_blk);
        if (h->lane[k].h_status) (void)hipHostFree(h->lane[k].h_status);
    }
    if (h->copy_st) {
        (void)hipStreamSynchronize(h->copy_st);
        (void)hipStreamDestroy(h->copy_st);
    }
    (void)hipFree(h->arena);
    (void)hipFree(h->d_sum);
    (void)hipFree(h->d_spec);
    (void)hipFree(h->d_pb);
    (void)hipFree(h->d_recoff);
    (void)hipFree(h->d_flags);
    (void)hipFree(h->d_seen);
    (void)hipFree(h->d_tid);
    (void)hipFree(h->d_pos);
    (void)hipFree(h->d_end);
    (void)hipFree(h->d_mpos);
    (void)hipFree(h->d_isize);
    (void)hipFree(h->d_mapq);
    (void)hipFree(h->d_f5);
    (void)hipFree(h->d_xa);
    if (h->h_sum) (void)hipHostFree(h->h_sum);
    if (h->h_pb) (void)hipHostFree(h->h_pb);
    for (int k = 0; k < 4; k++)
        if (h->ev[k]) (void)hipEventDestroy(h->ev[k]);
    free(h);
}

// Page-locked host memory for the reader's chunk buffers. hipHostMalloc of 384 MB takes 62 ms — nearly all of it the kernel
// handing out and zeroing 98 k small pages, under a lock that other HIP calls of the process wait for; an anonymous mapping the
// kernel is asked to back with 2 MB pages (the box has transparent huge pages in "madvise" mode), touched once by a few threads
// and then registered, takes 25 ms single-threaded (touch 24 + hipHostRegister 1; profiles/r03_pin_probe.txt) and copies to
// the device at the same 57 GB/s. Falls back to hipHostMalloc where any step fails.
#include <sys/mman.h>
#include <mutex>
#include <thread>
#include <vector>
struct PinnedMap {
    void *user, *base;
    size_t len;
};
static std::mutex g_pin_mu;
static std::vector<PinnedMap> g_pinned;

extern "C" void *itx_pinned_alloc(size_t bytes)
{
    if (bytes == 0) bytes = 1;
    const size_t huge = (size_t)2 << 20;
    if (bytes >= 4 * huge && !getenv("ITX_PIN_PLAIN")) {
        const size_t len = ((bytes + huge - 1) & ~(huge - 1)) + huge;
        struct timespec t0;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        void *base = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (base != MAP_FAILED) {
            uint8_t *user = (uint8_t *)(((uintptr_t)base + huge - 1) & ~(uintptr_t)(huge - 1));
            const size_t ulen = (bytes + huge - 1) & ~(huge - 1);
            (void)madvise(user, ulen, MADV_HUGEPAGE);
            // first touch (one byte per 2 MB would do for huge pages; every 4 KB page when the kernel declines them)
            const unsigned nt = 4;
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; t++)
                th.emplace_back([=]() {
                    const size_t lo = ulen / nt * t, hi = t + 1 == nt ? ulen : ulen / nt * (t + 1);
                    for (size_t a = lo; a < hi; a += 4096) ((volatile uint8_t *)user)[a] = 0;
                });
            for (auto &x : th) x.join();
            struct timespec ta, tb;
            clock_gettime(CLOCK_MONOTONIC, &ta);
            const hipError_t re = hipHostRegister(user, ulen, hipHostRegisterDefault);
            clock_gettime(CLOCK_MONOTONIC, &tb);
            if (getenv("ITX_TIMING_ALLOC"))
                fprintf(stderr, "[itx alloc] page-locked %.0f MB: mapped + touched %.1f ms, hipHostRegister %.1f ms\n", (double)ulen / 1e6,
                        1e3 * ((double)(ta.tv_sec - t0.tv_sec) + 1e-9 * (double)(ta.tv_nsec - t0.tv_nsec)), 1e3 * ((double)(tb.tv_sec - ta.tv_sec) + 1e-9 * (double)(tb.tv_nsec - ta.tv_nsec)));
            if (re == hipSuccess) {
                std::lock_guard<std::mutex> g(g_pin_mu);
                g_pinned.push_back(PinnedMap{user, base, len});
                return user;
            }
            (void)hipGetLastError();
            munmap(base, len);
        }
    }
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

extern "C" void itx_pinned_free(void *p)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> g(g_pin_mu);
        for (size_t i = 0; i < g_pinned.size(); i++)
            if (g_pinned[i].user == p) {
                const PinnedMap m = g_pinned[i];
                g_pinned.erase(g_pinned.begin() + (long)i);
                (void)hipHostUnregister(m.user);
                munmap(m.base, m.len);
                return;
            }
    }
    (void)hipHostFree(p);
}

// ---------------------------------------------------------------------------------------------------------------------
// A backlog of parsed records in HBM (SoA, 14 B per record, 22 with mates): the caller parses windows while its table is
// still being built, keeps only what the engine will read — a ninth of the inflated bytes — and lets the windows go
// (host/stream.c). Offsets are multiples of 16 records (itx_engine_submit_device* wants batches to start there).
struct itx_backlog {
    int device;
    size_t cap, used;
    int32_t *tid, *pos, *end, *mpos, *isize;
    uint8_t *mapq, *f5;
    hipStream_t st;
};

extern "C" int itx_backlog_create(int device, size_t max_records, itx_backlog **out)
{
    if (!out || max_records == 0 || max_records >= ((size_t)1 << 31)) return ITX_E_ARG;
    *out = nullptr;
    INF_HIP(hipSetDevice(device));
    itx_backlog *b = (itx_backlog *)calloc(1, sizeof *b);
    if (!b) return ITX_E_NOMEM;
    b->device = device;
    b->cap = (max_records + 15) & ~(size_t)15;
    const size_t n = b->cap + 64;
    if (hipMalloc((void **)&b->tid, n * 4) != hipSuccess || hipMalloc((void **)&b->pos, n * 4) != hipSuccess || hipMalloc((void **)&b->end, n * 4) != hipSuccess ||
        hipMalloc((void **)&b->mapq, n) != hipSuccess || hipMalloc((void **)&b->f5, n) != hipSuccess || hipStreamCreateWithFlags(&b->st, hipStreamNonBlocking) != hipSuccess) {
        itx_set_error("itx_backlog_create: no device memory for %zu records", max_records);
        (void)hipGetLastError();
        (void)hipFree(b->tid); (void)hipFree(b->pos); (void)hipFree(b->end); (void)hipFree(b->mapq); (void)hipFree(b->f5);
        free(b);
        return ITX_E_NOMEM;
    }
    *out = b;
    return ITX_OK;
}

extern "C" void itx_backlog_destroy(itx_backlog *b)
{
    if (!b) return;
    (void)hipSetDevice(b->device);
    if (b->st) {
        (void)hipStreamSynchronize(b->st);
        (void)hipStreamDestroy(b->st);
    }
    (void)hipFree(b->tid); (void)hipFree(b->pos); (void)hipFree(b->end); (void)hipFree(b->mapq); (void)hipFree(b->f5);
    (void)hipFree(b->mpos); (void)hipFree(b->isize);
    free(b);
}

/* room left, in records */
extern "C" size_t itx_backlog_room(const itx_backlog *b) { return b ? b->cap - b->used : 0; }

/* appends n records (DEVICE arrays; mpos / isize NULL: no mates) — copied, the source may be reused when the call returns;
 * *at = where they lie (a multiple of 16). ITX_E_LIMIT: no room (nothing copied). */
extern "C" int itx_backlog_append(itx_backlog *b, const itx_batch *src, size_t n, size_t *at)
{
    if (!b || !src || !at || !src->tid || !src->pos || !src->tmpend || !src->mapq || !src->flag5) return ITX_E_ARG;
    if (n > b->cap - b->used) return ITX_E_LIMIT;
    INF_HIP(hipSetDevice(b->device));
    const size_t o = b->used;
    if (src->mpos && src->isize && !b->mpos) {
        const size_t m = b->cap + 64;
        INF_HIP(hipMalloc((void **)&b->mpos, m * 4));
        INF_HIP(hipMalloc((void **)&b->isize, m * 4));
    }
    INF_HIP(hipMemcpyAsync(b->tid + o, src->tid, n * 4, hipMemcpyDeviceToDevice, b->st));
    INF_HIP(hipMemcpyAsync(b->pos + o, src->pos, n * 4, hipMemcpyDeviceToDevice, b->st));
    INF_HIP(hipMemcpyAsync(b->end + o, src->tmpend, n * 4, hipMemcpyDeviceToDevice, b->st));
    INF_HIP(hipMemcpyAsync(b->mapq + o, src->mapq, n, hipMemcpyDeviceToDevice, b->st));
    INF_HIP(hipMemcpyAsync(b->f5 + o, src->flag5, n, hipMemcpyDeviceToDevice, b->st));
    if (src->mpos && src->isize) {
        INF_HIP(hipMemcpyAsync(b->mpos + o, src->mpos, n * 4, hipMemcpyDeviceToDevice, b->st));
        INF_HIP(hipMemcpyAsync(b->isize + o, src->isize, n * 4, hipMemcpyDeviceToDevice, b->st));
    }
    INF_HIP(hipStreamSynchronize(b->st));
    *at = o;
    b->used = (o + n + 15) & ~(size_t)15;
    return ITX_OK;
}

/* the records at `at` as a batch for itx_engine_submit_device* (with_mates: the batch was appended with mates) */
extern "C" int itx_backlog_batch(const itx_backlog *b, size_t at, int with_mates, itx_batch *out)
{
    if (!b || !out || at > b->used || (at & 15u) || (with_mates && !b->mpos)) return ITX_E_ARG;
    out->tid = b->tid + at;
    out->pos = b->pos + at;
    out->tmpend = b->end + at;
    out->mapq = b->mapq + at;
    out->flag5 = b->f5 + at;
    out->mpos = with_mates ? b->mpos + at : nullptr;
    out->isize = with_mates ? b->isize + at : nullptr;
    return ITX_OK;
}

// ITX_TIMING: where the decoder's device-side time goes (printed when the process ends)
static double g_alloc_s, g_tok_ms, g_res_ms;
static unsigned long g_allocs, g_pushes;
static double wall_now()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static bool g_reported;
static void report_at_exit()
{
    if (g_reported || !g_pushes) return;
    g_reported = true;
    fprintf(stderr, "[itx timing] device decoder: %lu pushes, pass 1 %.1f ms and pass 2 %.1f ms per push (HIP events, pushes overlap), %lu device allocations %.3f s\n", g_pushes,
            g_pushes ? g_tok_ms / (double)g_pushes : 0.0, g_pushes ? g_res_ms / (double)g_pushes : 0.0, g_allocs, g_alloc_s);
}

extern "C" void itx_timing_report(void) { report_at_exit(); }

template <typename T> static int grow(T **p, size_t *cap, size_t need, bool exact = false)
{
    if (need <= *cap) return ITX_OK;
    const double t_alloc0 = wall_now();
    struct Acc {
        double t0;
        ~Acc()
        {
            g_alloc_s += wall_now() - t0;
            g_allocs++;
        }
    } acc{t_alloc0};
    if (*p) INF_HIP(hipFree(*p));
    *p = nullptr;
    *cap = 0;
    const size_t want = exact ? need : need + need / 4;
    const double tm0 = wall_now();
    INF_HIP(hipMalloc((void **)p, want * sizeof(T)));
    if (getenv("ITX_TIMING_ALLOC")) fprintf(stderr, "[itx alloc] hipMalloc %.1f MB: %.2f ms\n", (double)(want * sizeof(T)) / 1e6, 1e3 * (wall_now() - tm0));
    *cap = want;
    return ITX_OK;
}

extern "C" int itx_inflate_bgzf(itx_inflater *h, const void *comp, size_t comp_len, const itx_bgzf_block *blk, size_t n_blk, void *out, size_t out_len,
                                uint8_t *status)
{
    if (!h || !comp || !blk || !status) return ITX_E_ARG;          // out == NULL: the bytes stay on the device (timing runs)
    if (n_blk == 0) return ITX_OK;
    if (comp_len > 0xfffffff0u || out_len > 0xfffffff0u || n_blk > 0x7fffffffu) return ITX_E_LIMIT;
    // what the kernel assumes about every block, checked here: inside the buffers, outputs disjoint and in order
    size_t uat = 0;
    for (size_t i = 0; i < n_blk; i++) {
        const itx_bgzf_block &b = blk[i];
        if ((size_t)b.coff + b.csize > comp_len || b.csize < BGZF_HEADER + BGZF_TRAILER || b.uoff != uat || (size_t)b.uoff + b.usize > out_len ||
            b.usize > 65536u) {
            itx_set_error("itx_inflate_bgzf: block %zu does not fit its buffers", i);
            return ITX_E_ARG;
        }
        uat += b.usize;
    }
    INF_HIP(hipSetDevice(h->device));
    int rc;
    if ((rc = grow(&h->d_comp, &h->comp_cap, comp_len + 64)) != ITX_OK) return rc;
    if ((rc = grow(&h->d_out, &h->out_cap, out_len + 64)) != ITX_OK) return rc;
    if ((rc = grow(&h->d_status, &h->status_cap, n_blk)) != ITX_OK) return rc;
    if ((rc = grow(&h->d_blk, &h->blk_cap, n_blk)) != ITX_OK) return rc;
    if ((rc = grow(&h->d_lit, &h->lit_cap, n_blk * (size_t)SCR_STRIDE)) != ITX_OK) return rc;
    if ((rc = grow(&h->d_meta, &h->meta_cap, 3 * n_blk)) != ITX_OK) return rc;
    // pass 1 over all blocks at once (a lane per block: it takes many blocks to fill the chip)
    INF_HIP(hipMemcpyAsync(h->d_blk, blk, n_blk * sizeof *blk, hipMemcpyHostToDevice, h->st[0]));
    INF_HIP(hipMemcpyAsync(h->d_comp, comp, comp_len, hipMemcpyHostToDevice, h->st[0]));
    INF_HIP(hipEventRecord(h->ev[0], h->st[0]));
    hipLaunchKernelGGL(k_tokens, dim3((unsigned)((n_blk + 63) / 64)), dim3(64), 0, h->st[0], (const uint32_t *)h->d_comp, h->d_blk, (uint32_t)n_blk, h->d_lit,
                       h->d_meta);
    INF_HIP(hipGetLastError());
    INF_HIP(hipEventRecord(h->ev[1], h->st[0]));
    INF_HIP(hipStreamSynchronize(h->st[0]));
    // pass 2 in groups that alternate between two streams: one group's copy-out overlaps the next one's kernel
    const size_t per = n_blk < 2048 ? n_blk : (n_blk + 3) / 4;
    int s = 0;
    INF_HIP(hipEventRecord(h->ev[2], h->st[0]));
    for (size_t b0 = 0; b0 < n_blk; b0 += per, s ^= 1) {
        const size_t b1 = b0 + per < n_blk ? b0 + per : n_blk;
        hipLaunchKernelGGL(k_resolve, dim3((unsigned)(b1 - b0)), dim3(64), 0, h->st[s], h->d_blk, (uint32_t)b0, (uint32_t)(b1 - b0), h->d_lit, h->d_meta,
                           h->d_out, h->d_status);
        INF_HIP(hipGetLastError());
        if (b0 == 0) INF_HIP(hipEventRecord(h->ev[3], h->st[0]));
        const size_t u0 = blk[b0].uoff, u1 = (size_t)blk[b1 - 1].uoff + blk[b1 - 1].usize;
        if (out && u1 > u0) INF_HIP(hipMemcpyAsync((uint8_t *)out + u0, h->d_out + u0, u1 - u0, hipMemcpyDeviceToHost, h->st[s]));
        INF_HIP(hipEventRecord(h->ev_res_end[s], h->st[s]));
        INF_HIP(hipMemcpyAsync(status + b0, h->d_status + b0, b1 - b0, hipMemcpyDeviceToHost, h->st[s]));
    }
    INF_HIP(hipStreamSynchronize(h->st[0]));
    INF_HIP(hipStreamSynchronize(h->st[1]));
    (void)hipEventElapsedTime(&h->ms_tokens, h->ev[0], h->ev[1]);
    (void)hipEventElapsedTime(&h->ms_resolve, h->ev[2], h->ev[3]);
    {
        float a = 0, b = 0;                                          // all groups of pass 2 (with the copies out, when there are any)
        (void)hipEventElapsedTime(&a, h->ev[2], h->ev_res_end[0]);
        if (per < n_blk) (void)hipEventElapsedTime(&b, h->ev[2], h->ev_res_end[1]);
        h->ms_resolve_all = a > b ? a : b;
    }
    return ITX_OK;
}

/* device time of the two passes of the last itx_inflate_bgzf call (pass 2: its first group), milliseconds */
extern "C" int itx_inflater_last_ms(const itx_inflater *h, float *tokens_ms, float *resolve_ms)
{
    if (!h) return ITX_E_ARG;
    if (tokens_ms) *tokens_ms = h->ms_tokens;
    if (resolve_ms) *resolve_ms = h->ms_resolve;
    return ITX_OK;
}

/* pass 2 over all groups of the last itx_inflate_bgzf call (kernels only when that call was given out == NULL) */
extern "C" int itx_inflater_last_resolve_all_ms(const itx_inflater *h, float *ms)
{
    if (!h || !ms) return ITX_E_ARG;
    *ms = h->ms_resolve_all;
    return ITX_OK;
}

// --------------------------------------------------------------------------------------------------------- windows

static int check_blocks(const itx_bgzf_block *blk, size_t n_blk, size_t comp_len, size_t *total)
{
    size_t uat = 0;
    for (size_t i = 0; i < n_blk; i++) {
        const itx_bgzf_block &b = blk[i];
        if ((size_t)b.coff + b.csize > comp_len || b.csize < BGZF_HEADER + BGZF_TRAILER || b.uoff != uat || b.usize > 65536u) {
            itx_set_error("BGZF block %zu does not fit its buffers", i);
            return ITX_E_ARG;
        }
        uat += b.usize;
    }
    *total = uat;
    return ITX_OK;
}

#define BAD_W(w) ((w) < 0 || (w) >= ITX_BAMWIN_WINDOWS)
#define BAD_S(s) ((s) < 0 || (s) >= lanes_in_use())

/* push, first half: everything is enqueued on lane s's stream and the call returns; the caller's buffers are in use until
 * itx_bamwin_push_copied (comp) / this call's return (blk) */
extern "C" int itx_bamwin_push_begin(itx_inflater *h, int w, int s, const void *comp, size_t comp_len, const itx_bgzf_block *blk, size_t n_blk)
{
    if (!h || BAD_W(w) || BAD_S(s) || !comp || !blk) return ITX_E_ARG;
    auto &Ln = h->lane[s];
    if (Ln.busy) return ITX_E_STATE;
    size_t total = 0;
    int rc = check_blocks(blk, n_blk, comp_len, &total);
    if (rc != ITX_OK) return rc;
    if (comp_len > 0xfffffff0u || total + WIN_HEAD > 0xfffffff0u || n_blk > 0x7fffffffu) return ITX_E_LIMIT;
    INF_HIP(hipSetDevice(h->device));
    if (h->arena && (w >= h->n_reserved_win || WIN_HEAD + total + 64 > h->win[w].cap || comp_len + 64 > Ln.comp_cap || n_blk > Ln.status_cap)) {
        itx_set_error("push of %zu blocks / %zu bytes into window %d exceeds what itx_inflater_reserve set up", n_blk, total, w);
        return ITX_E_LIMIT;
    }
    if ((rc = grow(&h->win[w].buf, &h->win[w].cap, WIN_HEAD + total + 64)) != ITX_OK) return rc;
    h->win[w].start = h->win[w].consumed = WIN_HEAD;
    h->win[w].len = WIN_HEAD + (uint32_t)total;
    Ln.n_blk = n_blk;
    Ln.total = total;
    Ln.busy = 1;
    if (n_blk == 0) return ITX_OK;
    if (Ln.h_cap < n_blk) {
        if (Ln.h_blk) (void)hipHostFree(Ln.h_blk);
        if (Ln.h_status) (void)hipHostFree(Ln.h_status);
        Ln.h_blk = nullptr;
        Ln.h_status = nullptr;
        Ln.h_cap = 0;
        const size_t want = n_blk + n_blk / 4 + 64;
        INF_HIP(hipHostMalloc((void **)&Ln.h_blk, want * sizeof(itx_bgzf_block), hipHostMallocDefault));
        INF_HIP(hipHostMalloc((void **)&Ln.h_status, want, hipHostMallocDefault));
        Ln.h_cap = want;
    }
    for (size_t i = 0; i < n_blk; i++) {
        Ln.h_blk[i] = blk[i];
        Ln.h_blk[i].uoff += WIN_HEAD;
    }
    auto &CL = h->clane[s % compute_lanes()];
    if ((rc = grow(&Ln.d_comp, &Ln.comp_cap, comp_len + 64)) != ITX_OK) return rc;
    if ((rc = grow(&Ln.d_status, &Ln.status_cap, n_blk)) != ITX_OK) return rc;
    if ((rc = grow(&Ln.d_blk, &Ln.blk_cap, n_blk)) != ITX_OK) return rc;
    if ((rc = grow(&CL.d_lit, &CL.lit_cap, n_blk * (size_t)SCR_STRIDE)) != ITX_OK) return rc;      // (growing frees: that waits for whatever still runs)
    if ((rc = grow(&CL.d_meta, &CL.meta_cap, 3 * n_blk)) != ITX_OK) return rc;
    // the bytes cross PCIe on the copy stream ...
    INF_HIP(hipMemcpyAsync(Ln.d_blk, Ln.h_blk, n_blk * sizeof *blk, hipMemcpyHostToDevice, h->copy_st));
    INF_HIP(hipMemcpyAsync(Ln.d_comp, comp, comp_len, hipMemcpyHostToDevice, h->copy_st));
    INF_HIP(hipEventRecord(Ln.copied, h->copy_st));
    // ... and both passes run on the compute lane, behind the push that had the lane (and its scratch) before
    hipStream_t st = CL.cst;
    INF_HIP(hipStreamWaitEvent(st, Ln.copied, 0));
    INF_HIP(hipEventRecord(Ln.ev[0], st));
    hipLaunchKernelGGL(k_tokens, dim3((unsigned)((n_blk + 63) / 64)), dim3(64), 0, st, (const uint32_t *)Ln.d_comp, Ln.d_blk, (uint32_t)n_blk, CL.d_lit, CL.d_meta);
    INF_HIP(hipGetLastError());
    // pass 2 on the same stream, one wave per block (every push's pass 2 on one shared stream, or a fixed set of waves
    // that take blocks in turn, were measured slower: DESIGN.md)
    hipStream_t sr = st;
    INF_HIP(hipEventRecord(Ln.ev[1], sr));
    hipLaunchKernelGGL(k_resolve, dim3((unsigned)n_blk), dim3(64), 0, sr, Ln.d_blk, 0u, (uint32_t)n_blk, CL.d_lit, CL.d_meta, h->win[w].buf, Ln.d_status);
    INF_HIP(hipGetLastError());
    INF_HIP(hipEventRecord(Ln.ev[2], sr));
    INF_HIP(hipMemcpyAsync(Ln.h_status, Ln.d_status, n_blk, hipMemcpyDeviceToHost, sr));
    INF_HIP(hipEventRecord(Ln.done, sr));
    return ITX_OK;
}

/* the compressed bytes of lane s's push have been copied: the caller may reuse that buffer */
extern "C" int itx_bamwin_push_copied(itx_inflater *h, int s)
{
    if (!h || BAD_S(s)) return ITX_E_ARG;
    if (!h->lane[s].busy || h->lane[s].n_blk == 0) return ITX_OK;
    INF_HIP(hipSetDevice(h->device));
    INF_HIP(hipEventSynchronize(h->lane[s].copied));
    return ITX_OK;
}

/* push, second half: waits for lane s's push; status[n_blk] as for itx_inflate_bgzf, *n_new = bytes the window gained */
extern "C" int itx_bamwin_push_end(itx_inflater *h, int s, uint8_t *status, size_t *n_new)
{
    if (!h || BAD_S(s) || !status || !n_new) return ITX_E_ARG;
    auto &Ln = h->lane[s];
    if (!Ln.busy) return ITX_E_STATE;
    INF_HIP(hipSetDevice(h->device));
    if (Ln.n_blk) INF_HIP(hipEventSynchronize(Ln.done));
    if (Ln.n_blk) {
        float a = 0, b = 0;
        if (hipEventElapsedTime(&a, Ln.ev[0], Ln.ev[1]) == hipSuccess && hipEventElapsedTime(&b, Ln.ev[1], Ln.ev[2]) == hipSuccess) {
            g_tok_ms += a;
            g_res_ms += b;
            g_pushes++;
        }
    }
    if (Ln.n_blk) memcpy(status, Ln.h_status, Ln.n_blk);
    *n_new = Ln.total;
    Ln.busy = 0;
    return ITX_OK;
}

/* Everything a stream of pushes will need, allocated once while the device is idle: growing a buffer in the middle of the
 * pipeline means hipFree, which waits for every kernel in flight. Per lane: the compressed chunk (comp_bytes), block list,
 * status, literal and token scratch for max_blocks blocks. Windows: as many of max_bytes (+ the carry-over head) as a share
 * of the free memory allows, at most ITX_BAMWIN_WINDOWS and at most *n_windows on entry when that is >= 1; on return
 * *n_windows says how many the caller may use (w < *n_windows). */
extern "C" int itx_inflater_reserve(itx_inflater *h, size_t comp_bytes, size_t max_blocks, size_t max_bytes, int *n_windows)
{
    if (!h || !n_windows || max_blocks == 0 || max_blocks > 0x7fffffffu || max_bytes + WIN_HEAD > 0xfffffff0u) return ITX_E_ARG;
    if (h->arena) return ITX_E_STATE;
    INF_HIP(hipSetDevice(h->device));
    const int n_lanes = lanes_in_use();
    for (int k = 0; k < n_lanes; k++)
        if (h->lane[k].busy || h->lane[k].d_comp) return ITX_E_STATE;
    size_t free_b = 0, total_b = 0;
    INF_HIP(hipMemGetInfo(&free_b, &total_b));
    const size_t per = (WIN_HEAD + max_bytes + 64 + 255) & ~(size_t)255;
    size_t n = (free_b / 5 * 2) / per;                              // two fifths of what is free now
    if (n > ITX_BAMWIN_WINDOWS) n = ITX_BAMWIN_WINDOWS;
    if (*n_windows >= 1 && (size_t)*n_windows < n) n = (size_t)*n_windows;       // the caller cannot use more
    if (const char *e = getenv("ITX_RESERVE_WINDOWS"))
        if (atol(e) >= 2 && (size_t)atol(e) < n) n = (size_t)atol(e);
    if (n < 4) n = 4;
    // ONE allocation for all of it: on some hosts every hipMalloc costs ~15 ms whatever its size (and holds up the other
    // threads' HIP calls meanwhile) — seventy of them were 1.2 s of a 4 s run
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t sz_comp = al(comp_bytes + 64), sz_status = al(max_blocks), sz_blk = al(max_blocks * sizeof(itx_bgzf_block)), sz_lit = al(max_blocks * (size_t)SCR_STRIDE),
                 sz_meta = al(3 * max_blocks * 4);
    const int n_comp = compute_lanes();
    const size_t total = (sz_comp + sz_status + sz_blk) * (size_t)n_lanes + (sz_lit + sz_meta) * (size_t)n_comp + per * n;
    uint8_t *base = nullptr;
    const double t0 = wall_now();
    hipError_t he = hipMalloc((void **)&base, total);
    g_alloc_s += wall_now() - t0;
    g_allocs++;
    if (he != hipSuccess) {
        itx_set_error("itx_inflater_reserve: hipMalloc(%zu) failed: %s", total, hipGetErrorString(he));
        return ITX_E_NOMEM;
    }
    h->arena = base;
    h->n_reserved_win = (int)n;
    uint8_t *p = base;
    for (int k = 0; k < n_lanes; k++) {
        auto &Ln = h->lane[k];
        Ln.d_comp = p; p += sz_comp; Ln.comp_cap = comp_bytes + 64;
        Ln.d_status = p; p += sz_status; Ln.status_cap = max_blocks;
        Ln.d_blk = (itx_bgzf_block *)p; p += sz_blk; Ln.blk_cap = max_blocks;
        if (Ln.h_cap < max_blocks) {
            if (Ln.h_blk) (void)hipHostFree(Ln.h_blk);
            if (Ln.h_status) (void)hipHostFree(Ln.h_status);
            Ln.h_blk = nullptr;
            Ln.h_status = nullptr;
            Ln.h_cap = 0;
            INF_HIP(hipHostMalloc((void **)&Ln.h_blk, (max_blocks + 64) * sizeof(itx_bgzf_block), hipHostMallocDefault));
            INF_HIP(hipHostMalloc((void **)&Ln.h_status, max_blocks + 64, hipHostMallocDefault));
            Ln.h_cap = max_blocks + 64;
        }
    }
    for (int k = 0; k < n_comp; k++) {
        auto &CL = h->clane[k];
        CL.d_lit = p; p += sz_lit; CL.lit_cap = max_blocks * (size_t)SCR_STRIDE;
        CL.d_meta = (uint32_t *)p; p += sz_meta; CL.meta_cap = 3 * max_blocks;
    }
    for (size_t w = 0; w < n; w++) {
        if (h->win[w].buf) (void)hipFree(h->win[w].buf);
        h->win[w].buf = p;
        h->win[w].cap = per;
        p += per;
    }
    *n_windows = (int)n;
    return ITX_OK;
}

extern "C" int itx_bamwin_push(itx_inflater *h, int w, const void *comp, size_t comp_len, const itx_bgzf_block *blk, size_t n_blk, uint8_t *status, size_t *n_new)
{
    if (!status || !n_new) return ITX_E_ARG;
    *n_new = 0;
    int rc = itx_bamwin_push_begin(h, w, 0, comp, comp_len, blk, n_blk);
    if (rc != ITX_OK) return rc;
    return itx_bamwin_push_end(h, 0, status, n_new);
}

extern "C" int itx_bamwin_patch(itx_inflater *h, int w, size_t uoff, const void *bytes, size_t len)
{
    if (!h || BAD_W(w) || !bytes || WIN_HEAD + uoff + len > h->win[w].len) return ITX_E_ARG;
    INF_HIP(hipSetDevice(h->device));
    INF_HIP(hipMemcpy(h->win[w].buf + WIN_HEAD + uoff, bytes, len, hipMemcpyHostToDevice));
    return ITX_OK;
}

extern "C" int itx_bamwin_truncate(itx_inflater *h, int w, size_t n_new)
{
    if (!h || BAD_W(w) || WIN_HEAD + n_new > h->win[w].len) return ITX_E_ARG;
    h->win[w].len = WIN_HEAD + (uint32_t)n_new;
    return ITX_OK;
}

extern "C" int itx_bamwin_carry(itx_inflater *h, int from, int to)
{
    if (!h || BAD_W(from) || BAD_W(to) || from == to) return ITX_E_ARG;
    const uint32_t tail = h->win[from].len - h->win[from].consumed;
    if (h->win[to].start != WIN_HEAD) return ITX_E_STATE;
    INF_HIP(hipSetDevice(h->device));
    if (tail > WIN_HEAD) {
        // A record of more than 4 MiB straddles the chunks (a very long read): the head room in front of the fresh bytes is too
        // small, so the fresh bytes move back instead — as long as the window's buffer holds both (the reference reads such
        // files, bam.c:179-210 just reallocs; so does the host decoder).
        const size_t fresh = h->win[to].len - WIN_HEAD;
        if ((size_t)tail + fresh + 64 > h->win[to].cap || (size_t)tail + fresh > 0xfffffff0u) {
            itx_set_error("a BAM record of more than %u bytes straddles two chunks and does not fit the window (%zu bytes): beyond the device decoder "
                          "(ITX_HOST_INFLATE=1 reads such files)",
                          tail, h->win[to].cap);
            return ITX_E_LIMIT;
        }
        // move [WIN_HEAD, WIN_HEAD + fresh) to [tail, tail + fresh): from the end, in pieces no longer than the shift (no piece
        // overlaps its source) — when the shift is a megabyte or more; a tail only just over the head room would mean millions of
        // tiny copies (a shift of 100 bytes over a 1 GiB window: 10 M calls), so small shifts are one kernel that walks backwards
        const size_t shift = (size_t)tail - WIN_HEAD;
        if (shift >= (1u << 20)) {
            for (size_t done = 0; done < fresh;) {
                const size_t n = fresh - done < shift ? fresh - done : shift;
                const size_t src = WIN_HEAD + fresh - done - n;
                INF_HIP(hipMemcpyAsync(h->win[to].buf + src + shift, h->win[to].buf + src, n, hipMemcpyDeviceToDevice, h->st[1]));
                done += n;
            }
        } else if (fresh) {
            hipLaunchKernelGGL(k_move_up, dim3(1), dim3(MOVE_THREADS), 0, h->st[1], h->win[to].buf, (size_t)WIN_HEAD, fresh, shift);
            INF_HIP(hipGetLastError());
        }
        INF_HIP(hipMemcpyAsync(h->win[to].buf, h->win[from].buf + h->win[from].consumed, tail, hipMemcpyDeviceToDevice, h->st[1]));
        INF_HIP(hipStreamSynchronize(h->st[1]));
        h->win[to].start = h->win[to].consumed = 0;
        h->win[to].len = tail + (uint32_t)fresh;
        h->win[from].consumed = h->win[from].len;
        return ITX_OK;
    }
    if (tail) {
        INF_HIP(hipMemcpyAsync(h->win[to].buf + WIN_HEAD - tail, h->win[from].buf + h->win[from].consumed, tail, hipMemcpyDeviceToDevice, h->st[1]));
        INF_HIP(hipStreamSynchronize(h->st[1]));
    }
    h->win[to].start = h->win[to].consumed = WIN_HEAD - tail;
    h->win[from].consumed = h->win[from].len;
    return ITX_OK;
}

extern "C" int itx_bamwin_avail(const itx_inflater *h, int w, size_t *bytes)
{
    if (!h || BAD_W(w) || !bytes) return ITX_E_ARG;
    *bytes = h->win[w].len - h->win[w].consumed;
    return ITX_OK;
}

/* bytes [off, off + len) of the window's unconsumed part, to the host */
extern "C" int itx_bamwin_peek(itx_inflater *h, int w, size_t off, void *dst, size_t len)
{
    if (!h || BAD_W(w) || !dst || (size_t)h->win[w].consumed + off + len > h->win[w].len) return ITX_E_ARG;
    INF_HIP(hipSetDevice(h->device));
    if (len) INF_HIP(hipMemcpy(dst, h->win[w].buf + h->win[w].consumed + off, len, hipMemcpyDeviceToHost));
    return ITX_OK;
}

extern "C" int itx_bamwin_skip(itx_inflater *h, int w, size_t n)
{
    if (!h || BAD_W(w) || (size_t)h->win[w].consumed + n > h->win[w].len) return ITX_E_ARG;
    h->win[w].consumed += (uint32_t)n;
    h->win[w].start = h->win[w].consumed;
    return ITX_OK;
}

extern "C" int itx_bamwin_parse(itx_inflater *h, int w, int n_targets, size_t *n_rec, int *malformed, int *flags, size_t *rewalked)
{
    if (!h || BAD_W(w) || !n_rec || !malformed || !flags) return ITX_E_ARG;
    *n_rec = 0;
    *malformed = 0;
    *flags = 0;
    h->n_rec = 0;
    h->parsed_w = w;
    const uint32_t p0 = h->win[w].consumed, L = h->win[w].len;
    if (p0 >= L) return ITX_OK;
    INF_HIP(hipSetDevice(h->device));
    hipStream_t st = h->st[1];
    const uint32_t T = (L - p0 + PIECE - 1) / PIECE;
    int rc;
    {
        size_t cap = h->sum_cap;
        if ((rc = grow((PieceSum **)&h->d_sum, &cap, (size_t)T)) != ITX_OK) return rc;
        if (cap != h->sum_cap) {
            if (h->h_sum) (void)hipHostFree(h->h_sum);
            h->h_sum = nullptr;
            INF_HIP(hipHostMalloc(&h->h_sum, cap * sizeof(PieceSum), hipHostMallocDefault));
            h->sum_cap = cap;
        }
        cap = h->pb_cap;
        if ((rc = grow(&h->d_pb, &cap, 2 * (size_t)T)) != ITX_OK) return rc;
        if (cap != h->pb_cap) {
            if (h->h_pb) (void)hipHostFree(h->h_pb);
            h->h_pb = nullptr;
            INF_HIP(hipHostMalloc((void **)&h->h_pb, cap * sizeof(uint32_t), hipHostMallocDefault));
            h->pb_cap = cap;
        }
    }
    if ((rc = grow(&h->d_spec, &h->spec_cap, (size_t)T * PIECE_SLOTS)) != ITX_OK) return rc;
    if (!h->d_flags) INF_HIP(hipMalloc((void **)&h->d_flags, 16));
    if ((rc = grow(&h->d_seen, &h->seen_cap, (size_t)(n_targets > 0 ? n_targets : 1))) != ITX_OK) return rc;
    INF_HIP(hipMemsetAsync(h->d_seen, 0, (size_t)(n_targets > 0 ? n_targets : 1), st));
    PieceSum *d_sum = (PieceSum *)h->d_sum, *sum = (PieceSum *)h->h_sum;
    const uint8_t *u = h->win[w].buf;
    hipLaunchKernelGGL(k_guess, dim3((T + 63) / 64), dim3(64), 0, st, u, p0, L, T, (int32_t)n_targets, d_sum, h->d_spec);
    INF_HIP(hipGetLastError());
    INF_HIP(hipMemcpyAsync(sum, d_sum, (size_t)T * sizeof(PieceSum), hipMemcpyDeviceToHost, st));
    INF_HIP(hipStreamSynchronize(st));
    // in stream order: a piece counts if its guess is where the chain arrives, else it is walked again from there
    uint32_t cur = p0, endp = p0, why = 0, tot = 0;
    uint32_t *base = h->h_pb, *cnt = h->h_pb + T;
    size_t redo = 0;
    for (uint32_t t = 0; t < T; t++) {
        const uint32_t b = p0 + t * PIECE, lim = L - b > PIECE ? b + PIECE : L;
        base[t] = tot;
        cnt[t] = 0;
        if (why || cur >= lim) continue;                        // the stream ended, or a record reaches over the whole piece
        if (sum[t].c != cur) {
            hipLaunchKernelGGL(k_rewalk, dim3(1), dim3(1), 0, st, u, cur, lim, L, t, d_sum, h->d_spec);
            INF_HIP(hipGetLastError());
            INF_HIP(hipMemcpyAsync(&sum[t], d_sum + t, sizeof(PieceSum), hipMemcpyDeviceToHost, st));
            INF_HIP(hipStreamSynchronize(st));
            redo++;
        }
        cnt[t] = sum[t].n;
        tot += sum[t].n;
        endp = cur = sum[t].end;
        why = sum[t].why;
    }
    if (rewalked) *rewalked = redo;
    if (why == 2) {                                               // bam.c:186-190: a malformed length ends the file
        *malformed = 1;
        h->win[w].len = endp;
    }
    h->win[w].consumed = tot ? endp : p0;
    if (why == 2 && tot == 0) h->win[w].consumed = endp;
    if (tot == 0) return ITX_OK;
    if ((rc = grow(&h->d_recoff, &h->recoff_cap, (size_t)tot)) != ITX_OK) return rc;
    if (h->soa_cap < tot) {
        const size_t want = (size_t)tot + tot / 4;
        int32_t **i32s[5] = {&h->d_tid, &h->d_pos, &h->d_end, &h->d_mpos, &h->d_isize};
        uint8_t **u8s[3] = {&h->d_mapq, &h->d_f5, &h->d_xa};
        for (auto pp : i32s) {
            if (*pp) INF_HIP(hipFree(*pp));
            *pp = nullptr;
            INF_HIP(hipMalloc((void **)pp, want * 4));
        }
        for (auto pp : u8s) {
            if (*pp) INF_HIP(hipFree(*pp));
            *pp = nullptr;
            INF_HIP(hipMalloc((void **)pp, want));
        }
        h->soa_cap = want;
    }
    INF_HIP(hipMemcpyAsync(h->d_pb, h->h_pb, 2 * (size_t)T * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    INF_HIP(hipMemsetAsync(h->d_flags, 0, 4, st));
    hipLaunchKernelGGL(k_compact, dim3(T), dim3(64), 0, st, h->d_spec, h->d_pb, h->d_pb + T, T, h->d_recoff);
    INF_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_parse, dim3((tot + 255) / 256), dim3(256), 0, st, u, h->d_recoff, tot, h->d_tid, h->d_pos, h->d_end, h->d_mapq, h->d_f5, h->d_mpos, h->d_isize,
                       h->d_xa, h->d_flags, h->d_seen, (int32_t)n_targets);
    INF_HIP(hipGetLastError());
    uint32_t fl = 0;
    INF_HIP(hipMemcpyAsync(&fl, h->d_flags, 4, hipMemcpyDeviceToHost, st));
    INF_HIP(hipStreamSynchronize(st));
    *flags = (int)fl;
    *n_rec = tot;
    h->n_rec = tot;
    return ITX_OK;
}

/* records [first, first + n) of the last parse: SoA into host arrays (any may be NULL), their offsets in the window and
 * XA marks on request */
extern "C" int itx_bamwin_fetch(itx_inflater *h, size_t first, size_t n, const itx_staging *dst, size_t dst_at, uint32_t *rec_off, uint8_t *xa)
{
    if (!h || first + n > h->n_rec) return ITX_E_ARG;
    if (n == 0) return ITX_OK;
    INF_HIP(hipSetDevice(h->device));
    hipStream_t st = h->st[1];
    if (dst) {
        if (dst_at + n > dst->capacity) return ITX_E_ARG;
        if (dst->tid) INF_HIP(hipMemcpyAsync(dst->tid + dst_at, h->d_tid + first, n * 4, hipMemcpyDeviceToHost, st));
        if (dst->pos) INF_HIP(hipMemcpyAsync(dst->pos + dst_at, h->d_pos + first, n * 4, hipMemcpyDeviceToHost, st));
        if (dst->tmpend) INF_HIP(hipMemcpyAsync(dst->tmpend + dst_at, h->d_end + first, n * 4, hipMemcpyDeviceToHost, st));
        if (dst->mapq) INF_HIP(hipMemcpyAsync(dst->mapq + dst_at, h->d_mapq + first, n, hipMemcpyDeviceToHost, st));
        if (dst->flag5) INF_HIP(hipMemcpyAsync(dst->flag5 + dst_at, h->d_f5 + first, n, hipMemcpyDeviceToHost, st));
        if (dst->mpos) INF_HIP(hipMemcpyAsync(dst->mpos + dst_at, h->d_mpos + first, n * 4, hipMemcpyDeviceToHost, st));
        if (dst->isize) INF_HIP(hipMemcpyAsync(dst->isize + dst_at, h->d_isize + first, n * 4, hipMemcpyDeviceToHost, st));
    }
    if (rec_off) INF_HIP(hipMemcpyAsync(rec_off, h->d_recoff + first, n * 4, hipMemcpyDeviceToHost, st));
    if (xa) INF_HIP(hipMemcpyAsync(xa, h->d_xa + first, n, hipMemcpyDeviceToHost, st));
    INF_HIP(hipStreamSynchronize(st));
    return ITX_OK;
}

/* raw bytes [off, off + len) of the parsed window (offsets as itx_bamwin_fetch's rec_off), to the host */
extern "C" int itx_bamwin_bytes(itx_inflater *h, size_t off, void *dst, size_t len)
{
    if (!h || !dst) return ITX_E_ARG;
    const int w = h->parsed_w;
    if (off + len > h->win[w].cap) return ITX_E_ARG;
    INF_HIP(hipSetDevice(h->device));
    if (len) INF_HIP(hipMemcpy(dst, h->win[w].buf + off, len, hipMemcpyDeviceToHost));
    return ITX_OK;
}

extern "C" int itx_bamwin_tids(itx_inflater *h, uint8_t *seen, int n_targets)
{
    if (!h || !seen || n_targets < 0 || (size_t)n_targets > h->seen_cap) return ITX_E_ARG;
    INF_HIP(hipSetDevice(h->device));
    if (n_targets) INF_HIP(hipMemcpy(seen, h->d_seen, (size_t)n_targets, hipMemcpyDeviceToHost));
    return ITX_OK;
}

int itx_xaveto_run(itx_xaveto *x, const uint8_t *u, const uint32_t *rec_off, const uint8_t *xa_mark, const int32_t *tid, const int32_t *pos, const int32_t *end,
                   uint8_t *f5, const int32_t *mpos, const int32_t *isize, size_t n, uint64_t *n_vetoed, uint64_t *n_hard);      // itx_xaveto.hip

/* the XA veto over records [first, first + n) of the last parsed window (chosen rows already in the veto object's buffer) */
extern "C" int itx_bamwin_xa_veto(itx_inflater *h, itx_xaveto *x, size_t first, size_t n, uint64_t *n_vetoed, uint64_t *n_hard)
{
    if (!h || !x || !n_vetoed || !n_hard || first + n > h->n_rec) return ITX_E_ARG;
    const int w = h->parsed_w;
    return itx_xaveto_run(x, h->win[w].buf, h->d_recoff + first, h->d_xa + first, h->d_tid + first, h->d_pos + first, h->d_end + first, h->d_f5 + first,
                          h->d_mpos + first, h->d_isize + first, n, n_vetoed, n_hard);
}

/* -R over all records of the last parsed window (csrc/itx_dedup.hip): the dropped ones get ITX_F5_NOLOOKUP where the window's flags lie */
extern "C" int itx_bamwin_dedup(itx_inflater *h, itx_dedup *d)
{
    if (!h || !d) return ITX_E_ARG;
    return itx_dedup_run(d, h->d_tid, h->d_pos, h->d_end, h->d_mapq, h->d_f5, h->d_mpos, h->d_isize, h->n_rec);
}

extern "C" int itx_bamwin_device_batch(itx_inflater *h, size_t first, int with_mates, itx_batch *out)
{
    if (!h || !out || first > h->n_rec || (first & 15u)) return ITX_E_ARG;
    out->tid = h->d_tid + first;
    out->pos = h->d_pos + first;
    out->tmpend = h->d_end + first;
    out->mapq = h->d_mapq + first;
    out->flag5 = h->d_f5 + first;
    out->mpos = with_mates ? h->d_mpos + first : nullptr;
    out->isize = with_mates ? h->d_isize + first : nullptr;
    return ITX_OK;
}
