// mc_device.hpp -- CDNA4 (gfx950, wave64) kernels of the classifier core.
//
// Replaces the three CUDA kernels of the reference (src/CuClarkDB.cu:999-1411):
//   queryKernel + queryElement  -> mc::query_kernel<LINE>  (fused with resultKernel
//                                  when only the final rows are wanted)
//   mergeKernel                 -> mc::merge_rows_kernel
//   resultKernel                -> mc::result_rows_kernel
// plus the load-time re-layout kernels that turn the on-disk bucket arrays
// (.sz/.ky/.lb, reference src/hashTable_hh.hh:473-546) into "bucket lines".
//
// Design:
//   * integer hash + random gather, HBM-transaction bound; no MFMA.
//   * one 64-lane wavefront walks one read: lane i owns k-mer i (and i+64, both probes
//     in flight together); waves are independent (no workgroup barrier anywhere).
//   * a wave stages the packed containers of a GROUP of reads into its private LDS
//     slice with 16-byte coalesced loads; lanes then cut their k-mers out of LDS.
//   * the table is re-laid out so that ONE probe touches ONE aligned line
//     (keys + labels + size of the bucket together), instead of the reference's three
//     dependent reads (bucketPointers[r], [r+1] -> keys[] -> labels[]).
//   * per-target hit counts live in registers: lane j of the wave holds the j-th
//     distinct target seen for the read (id, count); hits are folded in with
//     ballot/readlane, never through a T-sized shared array (reference :1017-1026).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace mc {

// ---------------------------------------------------------------------------
// division by the (run-time) table size: q = c / HTSIZE, r = c % HTSIZE
// (reference CuClarkDB.cu:1208-1209 divides by a compile-time constant)
// ---------------------------------------------------------------------------
struct DivU64 {
    uint64_t d;
    uint64_t magic;   // 0 => d is a power of two
    uint32_t shift;
    uint32_t add;
};

__device__ __forceinline__ uint64_t div_u64(uint64_t n, const DivU64 &dv)
{
    if (dv.magic == 0) return n >> dv.shift;
    uint64_t q = __umul64hi(dv.magic, n);
    if (dv.add) {
        uint64_t t = ((n - q) >> 1) + q;
        return t >> dv.shift;
    }
    return q >> dv.shift;
}

// n mod d for the shard filter when n / d < 2^32 and d > 1024 (MzArgs::inv_htsize != 0, set in launch_query):
// the quotient is estimated in double precision -- n loses at most 9 low bits in the conversion and the
// two roundings add 3 * 2^-53 relative, so the estimate is within one of the true quotient -- and the
// remainder is put right by at most one +-d.  About half the vector instructions of the 64-bit magic.
__device__ __forceinline__ uint64_t rem_u64_fp(uint64_t n, uint32_t d, double inv_d)
{
    const uint32_t q = (uint32_t)((double)n * inv_d);
    int64_t r = (int64_t)(n - (uint64_t)q * d);
    if (r < 0) r += d;
    else if (r >= (int64_t)d) r -= d;
    return (uint64_t)r;
}

// host side: libdivide-style magic for unsigned 64-bit division by a run-time constant
inline DivU64 make_div(uint64_t d)
{
    DivU64 r{};
    r.d = d;
    const uint32_t L = 63u - (uint32_t)__builtin_clzll(d);
    if ((d & (d - 1)) == 0) { r.magic = 0; r.shift = L; r.add = 0; return r; }
    const unsigned __int128 num = (unsigned __int128)1 << (64 + L);
    uint64_t m = (uint64_t)(num / d);
    const uint64_t rem = (uint64_t)(num % d);
    const uint64_t e = d - rem;
    if (e < ((uint64_t)1 << L)) {
        r.shift = L; r.add = 0;
    } else {
        m += m;
        const uint64_t twice = rem + rem;
        if (twice >= d || twice < rem) m += 1;
        r.shift = L; r.add = 1;
    }
    r.magic = m + 1;
    return r;
}

// ---------------------------------------------------------------------------
// bucket lines
//   LINE bytes per bucket.  Narrow keys (quotient < 2^32-1: every k <= 31 of the
//   reference's two table sizes), CAP = (LINE-4)/6 k-mers:
//     dword [0, CAP)            keys (quotients, ascending; unused = 0xFFFFFFFF)
//     dword [CAP, CAP+CAP/2)    labels, two u16 per dword
//   Wide keys (WIDE: quotients need 64 bits, the reference's T64 regime, k = 32,
//   src/main.cc:277-286), CAP = (LINE-4)/10 k-mers:
//     dword [0, 2*CAP)          keys as (lo, hi) pairs, unused = all ones
//     dword [2*CAP, 2*CAP+CAP/2) labels
//   Both:
//     dword LINE/4-1            header: low byte = number of keys, or 0xFF when the
//                               bucket did not fit: then dword0|dword1<<32 = offset
//                               into the overflow arrays and dword2 = its size.
// ---------------------------------------------------------------------------
static constexpr uint32_t KEY_SENTINEL = 0xFFFFFFFFu;
static constexpr uint32_t HDR_OVERFLOW = 0xFFu;

template <int LINE, bool WIDE = false> struct LineCfg {
    static constexpr int DW   = LINE / 4;
    static constexpr int CAP  = WIDE ? (LINE - 4) / 10 : (LINE - 4) / 6;
    static constexpr int KDW  = WIDE ? 2 : 1;          // dwords per key
    static constexpr int LAB0 = CAP * KDW;             // first label dword
    static constexpr int HDR  = DW - 1;
};
template <bool WIDE> struct KeyOf { typedef uint32_t type; };
template <> struct KeyOf<true> { typedef uint64_t type; };

struct QueryArgs {
    const uint32_t *reads_ptr;      // n_reads+1 container offsets (ref :1034-1035)
    const uint16_t *containers;     // [len][containers...] per part (ref :1044-1046)
    uint64_t n_reads;
    uint64_t n_containers;
    const uint32_t *n_dev;          // not NULL: [0] reads, [1] containers of the batch, counted on the device (mc_ingest.hip);
                                    // n_reads / n_containers are then upper bounds (the grid is sized from them)
    const uint8_t  *lines;          // bucket lines of this shard
    const void     *ovf_keys;       // u32 (narrow) or u64 (wide) quotients of oversized buckets
    const uint16_t *ovf_labels;
    uint64_t shard_begin;           // ref dbPartStart / dbPartEnd (:1212-1214)
    uint64_t shard_end;
    DivU64   div;
    uint32_t k;
    uint32_t maxhits;
    uint32_t flags;                 // MC_F_FINAL | MC_F_ROWS
    uint32_t stage_ok;              // containers pointer is 16-byte aligned
    uint16_t *final_rows;           // 5 u16 per read
    uint16_t *sparse_rows;          // 2*maxhits+2 u16 per read
    unsigned long long *over_maxhits;
};

static constexpr int WAVES_PER_BLOCK = 4;
static constexpr int BLOCK_THREADS   = 64 * WAVES_PER_BLOCK;
#ifndef MC_GROUP_READS
#define MC_GROUP_READS 16
#endif
static constexpr int GROUP_READS     = MC_GROUP_READS;     // reads staged per wave at a time (at most 32: one lane per read offset)
static_assert(GROUP_READS >= 1 && GROUP_READS <= 32, "group size");
#ifndef MC_STAGE_CON
#define MC_STAGE_CON 1024
#endif
static constexpr int STAGE_CON       = MC_STAGE_CON;   // u16 containers per wave LDS slice

// Per-target hit counts are u16 on the wire (reference RESULTS, dataType.hh:37-43).  The reference's
// packed 2 x u16 shared-memory atomics carry into the neighbouring target at 65 536 (CuClarkDB.cu:1104-1108):
// no defined behaviour to reproduce.  ONE rule here, in every output (sparse rows, merged rows, fused
// and unfused final rows): a count saturates at 65 535.  Saturating addition of non-negative numbers is
// associative, so the rule is shard-invariant; sumN stays the u16 sum of those counts (it wraps, :1370-1372).
__host__ __device__ __forceinline__ uint32_t sat_u16(uint32_t c) { return c > 0xFFFFu ? 0xFFFFu : c; }

// reverse complement: complement every 2-bit code, reverse the order of the codes,
// keep the low 2k bits.  Same function as reference CuClarkDB.cu:1196-1203, written
// with v_bfrev_b32 instead of five swap rounds.
__device__ __forceinline__ uint64_t revcomp(uint64_t x, uint32_t k)
{
    // bit reversal turns each 2-bit code around as well: swap the bits of every pair back; the complement is taken in
    // the same three-input operation as that swap (one v_bitop3 per half instead of v_bfi + v_not)
    const uint32_t lo = __brev((uint32_t)(x >> 32)), hi = __brev((uint32_t)x);     // halves trade places
    const uint32_t slo = ~(((lo >> 1) & 0x55555555u) | ((lo << 1) & 0xAAAAAAAAu));
    const uint32_t shi = ~(((hi >> 1) & 0x55555555u) | ((hi << 1) & 0xAAAAAAAAu));
    return (((uint64_t)shi << 32) | slo) >> (64u - 2u * k);
}

// Wave-wide reductions on DPP (no LDS crossbar round trips): xor butterflies inside each
// row of 16 lanes, then row_bcast15 / row_bcast31 carry the row results upward; lane 63
// ends with the total, which v_readlane broadcasts.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_zero_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ uint32_t lane_bcast(uint32_t v, uint32_t uniform_lane)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)__builtin_amdgcn_readfirstlane((int)uniform_lane));
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    uint32_t w;
    w = dpp_zero_u32<0xB1, 0xF>(v);  v = w > v ? w : v;      // quad_perm [1,0,3,2]
    w = dpp_zero_u32<0x4E, 0xF>(v);  v = w > v ? w : v;      // quad_perm [2,3,0,1]
    w = dpp_zero_u32<0x141, 0xF>(v); v = w > v ? w : v;      // row_half_mirror
    w = dpp_zero_u32<0x140, 0xF>(v); v = w > v ? w : v;      // row_mirror
    w = dpp_zero_u32<0x142, 0xA>(v); v = w > v ? w : v;      // row_bcast15 -> rows 1, 3
    w = dpp_zero_u32<0x143, 0xC>(v); v = w > v ? w : v;      // row_bcast31 -> rows 2, 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
    v += dpp_zero_u32<0xB1, 0xF>(v);
    v += dpp_zero_u32<0x4E, 0xF>(v);
    v += dpp_zero_u32<0x141, 0xF>(v);
    v += dpp_zero_u32<0x140, 0xF>(v);
    v += dpp_zero_u32<0x142, 0xA>(v);
    v += dpp_zero_u32<0x143, 0xC>(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// ---------------------------------------------------------------------------
// cooperative bucket probe
//
// A bucket line is fetched by LPP = LINE/16 adjacent lanes, 16 bytes each, so that one
// probe is ONE coalesced LINE-byte request (one L1-TLB lookup, one TCP->L2 request, one
// HBM fetch).  A wave step therefore takes LPP rounds: in round j the 64/LPP lane groups
// serve the probes of owner lanes j*64/LPP ... and group lane `part` holds dwords
// [4*part, 4*part+4) of the line.  (Measured on MI355X: a lane loading its whole line
// with four 16-byte loads costs four requests + four TLB lookups per probe and 1.37 HBM
// fetches per probe -- profiles/r01_v1_lane_per_probe_pmc.txt.)
// ---------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
static constexpr uint32_t LIDX_NONE = 0xFFFFFFFFu;     // no probe for this lane

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}
// minimum over the LPP lanes of a group, result in every lane of the group
template <int LPP>
__device__ __forceinline__ uint32_t group_min(uint32_t v)
{
    uint32_t w = dpp_u32<0xB1>(v);  v = w < v ? w : v;          // quad_perm [1,0,3,2]
    w = dpp_u32<0x4E>(v);           v = w < v ? w : v;          // quad_perm [2,3,0,1]
    if (LPP == 8) { w = dpp_u32<0x141>(v); v = w < v ? w : v; } // row_half_mirror
    return v;
}

template <int LINE>
__device__ __forceinline__ u32x4 part_load(const uint8_t *lines, uint32_t lidx, uint32_t part)
{
    const u32x4 *p = reinterpret_cast<const u32x4 *>(lines + (uint64_t)lidx * (uint64_t)LINE + part * 16u);
    // a line is used once per batch: nt keeps it from displacing the read stream in L2
    // (measured: plain loads 357 vs nt 416 Mreads/s, DESIGN.md "Tuning log")
    return __builtin_nontemporal_load(p);
}

// Evaluate one round: this lane holds dwords [4*part, 4*part+4) of the line of its
// group's probe, whose quotient is q.  Returns hit/label for the group's probe (the
// same value in every lane of the group).  Lowest matching index wins, like the
// ascending scan of reference CuClarkDB.cu:1236-1247.
template <int LINE, bool WIDE>
__device__ __forceinline__ bool part_find(const u32x4 v, uint64_t q, bool active, uint32_t lane,
                                          const void *ovf_keys_v, const uint16_t *ovf_labels,
                                          uint32_t &label)
{
    using C = LineCfg<LINE, WIDE>;
    typedef typename KeyOf<WIDE>::type key_t;
    constexpr int LPP = LINE / 16;
    const uint32_t part = lane & (LPP - 1);
    const uint32_t gbase = lane & ~(uint32_t)(LPP - 1);
    uint32_t midx = 0xFFu;
    if (!WIDE) {
#pragma unroll
        for (int d = 3; d >= 0; d--) {
            const uint32_t gd = part * 4u + (uint32_t)d;
            if (active && gd < (uint32_t)C::CAP && v[d] == (uint32_t)q) midx = gd;
        }
    } else {
#pragma unroll
        for (int d = 1; d >= 0; d--) {
            const uint32_t gd = part * 2u + (uint32_t)d;      // key index
            const uint64_t key = (uint64_t)v[2 * d] | ((uint64_t)v[2 * d + 1] << 32);
            if (active && gd < (uint32_t)C::CAP && key == q) midx = gd;
        }
    }
    midx = group_min<LPP>(midx);
    // label dword of key midx: global dword LAB0 + midx/2 -> lane (that/4), register (that%4)
    const uint32_t ld = (uint32_t)C::LAB0 + ((midx & 0x7Fu) >> 1);
    const uint32_t sel = ld & 3u;
    const uint32_t cand = sel == 0 ? v[0] : sel == 1 ? v[1] : sel == 2 ? v[2] : v[3];
    const uint32_t lw = (uint32_t)__shfl((int)cand, (int)(gbase + (ld >> 2)), 64);
    bool hit = midx != 0xFFu;
    label = (midx & 1u) ? (lw >> 16) : (lw & 0xFFFFu);

    // buckets that did not fit a line (header 0xFF): rare, handled by group lane 0
    const bool ovf_here = active && part == (uint32_t)(LPP - 1) && (v[3] & 0xFFu) == HDR_OVERFLOW;
    if (__ballot(ovf_here)) {
        const key_t *ovf_keys = static_cast<const key_t *>(ovf_keys_v);
        const bool ovf = __shfl((int)ovf_here, (int)(gbase + LPP - 1), 64) != 0;
        uint32_t res = 0;                               // bit 16 = hit, low 16 = label
        if (ovf && part == 0) {
            const uint64_t off = (uint64_t)v[0] | ((uint64_t)v[1] << 32);
            const uint32_t n = v[2];
            for (uint32_t i = 0; i < n; i++) {
                const uint64_t key = ovf_keys[off + i];
                if (key == q) { res = 0x10000u | ovf_labels[off + i]; break; }
                if (key > q) break;
            }
        }
        res = (uint32_t)__shfl((int)res, (int)gbase, 64);
        if (ovf) { hit = (res >> 16) != 0; label = res & 0xFFFFu; }
    }
    return hit;
}

// ---------------------------------------------------------------------------
// the query kernel
// ---------------------------------------------------------------------------
#ifndef MC_MIN_WAVES
#define MC_MIN_WAVES 1
#endif
#ifndef MC_NSLOT
#define MC_NSLOT 2      // k-mers per lane per step (both probes in flight together)
#endif
template <int LINE, bool WIDE>
__global__ __launch_bounds__(BLOCK_THREADS, MC_MIN_WAVES)
void query_kernel(const QueryArgs a)
{
    __shared__ __attribute__((aligned(16))) uint16_t s_con[WAVES_PER_BLOCK][STAGE_CON + 16];

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform: scalar registers
    uint16_t *slice = s_con[wave];

    const uint32_t k = a.k;
    const uint64_t kmask = k >= 32 ? ~0ull : ((1ull << (2u * k)) - 1ull);
    const uint64_t a_n_reads = a.n_dev ? (uint64_t)a.n_dev[0] : a.n_reads, a_n_con = a.n_dev ? (uint64_t)a.n_dev[1] : a.n_containers;
    const uint64_t n_groups = (a_n_reads + GROUP_READS - 1) / GROUP_READS;
    const uint64_t gstride = (uint64_t)gridDim.x * WAVES_PER_BLOCK;
    const uint32_t row_len = 2u * a.maxhits + 2u;

    for (uint64_t g = (uint64_t)blockIdx.x * WAVES_PER_BLOCK + wave; g < n_groups; g += gstride) {
        const uint64_t r0 = g * GROUP_READS;
        const uint32_t nr = (uint32_t)((a_n_reads - r0) < GROUP_READS ? (a_n_reads - r0) : GROUP_READS);

        // container offsets of the group's reads: lane i holds reads_ptr[r0 + i]
        uint32_t ptr_v = 0;
        if (lane <= nr) ptr_v = a.reads_ptr[r0 + lane];
        // The group is staged into the wave's LDS slice in as few pieces as fit: usually all
        // 16 reads at once; long reads (2 x 250 bp pairs, contigs) in smaller pieces; a single
        // read larger than the slice is read from global memory.
        for (uint32_t rs = 0; rs < nr;) {
        const uint32_t c0 = lane_bcast(ptr_v, rs);
        const uint32_t c0a = c0 & ~7u;
        const uint64_t fits = __ballot(lane > rs && lane <= nr && (ptr_v - c0a) <= (uint32_t)STAGE_CON);
        const bool staged = a.stage_ok && fits != 0;
        const uint32_t re = staged ? (uint32_t)(63 - __builtin_clzll((unsigned long long)fits)) : rs + 1u;
        const uint32_t c1 = lane_bcast(ptr_v, re);

        // stage [c0a, c1) into this wave's LDS slice (16-byte loads, coalesced)
        if (staged) {
            for (uint32_t j = lane * 8u; c0a + j < c1; j += 64u * 8u) {
                const uint64_t gi = (uint64_t)c0a + j;
                if (gi + 8u <= a_n_con) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(a.containers + gi);
                    *reinterpret_cast<uint4 *>(slice + j) = v;
                } else {
                    for (uint32_t t = 0; t < 8u; t++)
                        slice[j + t] = (gi + t < a_n_con) ? a.containers[gi + t] : (uint16_t)0;
                }
            }
            // LDS is in order per wave; only the compiler must not reorder
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }

        // container i of the batch (absolute index); every access is clamped so that
        // malformed input can only produce wrong counts, never an out-of-range access
        // (two instantiations of the per-group body, so that container reads compile to plain
        // LDS reads or plain global reads -- a run-time select would make them FLAT loads)
        auto run_group = [&](auto staged_c) {
        constexpr bool STAGED = decltype(staged_c)::value;
        auto con = [&](uint32_t i) -> uint32_t {
            if constexpr (STAGED) {
                const uint32_t li = i - c0a;
                return slice[li < (uint32_t)(STAGE_CON + 15) ? li : (uint32_t)(STAGE_CON + 15)];
            } else {
                const uint64_t ii = i < a_n_con ? i : a_n_con - 1;
                return a.containers[ii];
            }
        };

        for (uint32_t ri = rs; ri < re; ri++) {
            const uint32_t beg = lane_bcast(ptr_v, ri);
            uint32_t end = lane_bcast(ptr_v, ri + 1u);
            if ((uint64_t)end > a_n_con) end = (uint32_t)a_n_con;

            // accumulator: lane j = j-th distinct target of this read
            uint32_t acc_t = 0xFFFFFFFFu, acc_c = 0;
            uint32_t n_acc = 0;          // wave-uniform

            uint32_t pp = beg;
            while (pp < end) {                                  // parts (ref :1042-1117)
                const uint32_t plen = con(pp);
                const uint32_t first = pp + 1;
                pp = first + (plen ? (plen - 1u) / 8u + 1u : 0u);
                if (plen < k) continue;
                const uint32_t nk = plen - k + 1u;

                for (uint32_t base = 0; base < nk; base += 64u * MC_NSLOT) {
                    constexpr int LPP = LINE / 16;        // lanes per probe
                    constexpr int PPR = 64 / LPP;         // probes per round
                    constexpr int NSLOT = MC_NSLOT;
                    uint64_t qv[NSLOT];
                    uint32_t lidx[NSLOT];
#pragma unroll
                    for (int s = 0; s < NSLOT; s++) {
                        const uint32_t p = base + 64u * s + lane;
                        qv[s] = 0; lidx[s] = LIDX_NONE;
                        if (p < nk) {
                            // 80-bit window = containers j0..j0+4, first base in the top bits
                            const uint32_t j0 = first + (p >> 3);
                            const uint64_t hi = ((uint64_t)con(j0) << 48) | ((uint64_t)con(j0 + 1) << 32)
                                              | ((uint64_t)con(j0 + 2) << 16) | (uint64_t)con(j0 + 3);
                            const uint32_t lo = con(j0 + 4);
                            const uint32_t sh = 80u - 2u * (p & 7u) - 2u * k;   // >= 2
                            uint64_t x = sh >= 16u ? (hi >> (sh - 16u))
                                                   : ((hi << (16u - sh)) | (uint64_t)(lo >> sh));
                            x &= kmask;
                            const uint64_t rc = revcomp(x, k);
                            const uint64_t c  = x < rc ? x : rc;               // canonical (ref :1206)
                            const uint64_t q  = div_u64(c, a.div);
                            const uint64_t r  = c - q * a.div.d;
                            if ((r >= a.shard_begin) && (r < a.shard_end)) {   // ref :1212-1214
                                qv[s] = q;
                                lidx[s] = (uint32_t)(r - a.shard_begin);
                            }
                        }
                    }
                    // issue every line fetch of the step (2 slots x LPP rounds) before using any
                    const uint32_t part = lane & (LPP - 1);
                    uint64_t gq[NSLOT][LPP];
                    uint32_t gl[NSLOT][LPP];
                    u32x4 gv[NSLOT][LPP];
#pragma unroll
                    for (int s = 0; s < NSLOT; s++) {
#pragma unroll
                        for (int j = 0; j < LPP; j++) {
                            const int src = j * PPR + (int)(lane / LPP);
                            gq[s][j] = (uint32_t)__shfl((int)(uint32_t)qv[s], src, 64);
                            if (WIDE) gq[s][j] |= (uint64_t)(uint32_t)__shfl((int)(uint32_t)(qv[s] >> 32), src, 64) << 32;
                            gl[s][j] = (uint32_t)__shfl((int)lidx[s], src, 64);
                            gv[s][j] = u32x4{0u, 0u, 0u, 0u};
                            if (gl[s][j] != LIDX_NONE) gv[s][j] = part_load<LINE>(a.lines, gl[s][j], part);
                        }
                    }
                    // after the LPP rounds lane l holds the result of probe PPR*(l%LPP) + l/LPP:
                    // a permutation of the step's probes, which is all the counting needs
                    bool     hit[NSLOT];
                    uint32_t lab[NSLOT];
#pragma unroll
                    for (int s = 0; s < NSLOT; s++) {
                        hit[s] = false; lab[s] = 0;
#pragma unroll
                        for (int j = 0; j < LPP; j++) {
                            uint32_t l = 0;
                            const bool h = part_find<LINE, WIDE>(gv[s][j], gq[s][j], gl[s][j] != LIDX_NONE, lane,
                                                                 a.ovf_keys, a.ovf_labels, l);
                            if (part == (uint32_t)j) { hit[s] = h; lab[s] = l; }
                        }
                    }

                    // fold the hits of this step into the accumulator, one distinct
                    // target per iteration (wave-uniform control flow)
                    uint64_t m[NSLOT];
                    uint64_t many = 0;
#pragma unroll
                    for (int s = 0; s < NSLOT; s++) { m[s] = __ballot(hit[s]); many |= m[s]; }
                    while (many) {
                        uint32_t t = 0;
                        bool got = false;
#pragma unroll
                        for (int s = 0; s < NSLOT; s++) {
                            if (!got && m[s]) {
                                t = lane_bcast(lab[s], (uint32_t)(__ffsll((unsigned long long)m[s]) - 1));
                                got = true;
                            }
                        }
                        uint32_t cnt = 0;
                        many = 0;
#pragma unroll
                        for (int s = 0; s < NSLOT; s++) {
                            const uint64_t same = __ballot(hit[s] && lab[s] == t);
                            cnt += (uint32_t)__popcll(same);
                            m[s] &= ~same;
                            many |= m[s];
                            if (lab[s] == t) hit[s] = false;
                        }

                        const uint64_t ex = __ballot(acc_t == t);
                        if (ex) {
                            if (acc_t == t) acc_c += cnt;
                        } else if (n_acc < 64u) {
                            if (lane == n_acc) { acc_t = t; acc_c = cnt; }
                            n_acc++;
                        } else {
                            // 64 distinct targets already: keep the 64 smallest ids
                            const uint32_t mx = wave_max_u32(acc_t);
                            if (t < mx) {
                                const uint64_t who = __ballot(acc_t == mx);
                                if (lane == (uint32_t)(__ffsll((unsigned long long)who) - 1)) { acc_t = t; acc_c = cnt; }
                            }
                        }
                    }
                }
            }

            // ---- finalisation for this read ---------------------------------
            const uint64_t rd = r0 + ri;
            bool valid = lane < n_acc;
            uint32_t rank = 0;
            const bool need_rank = (a.flags & 2u) || (n_acc > a.maxhits);
            if (need_rank) {
                for (uint32_t j = 0; j < n_acc; j++) {
                    const uint32_t tj = lane_bcast(acc_t, j);
                    rank += (tj < acc_t) ? 1u : 0u;
                }
                if (n_acc > a.maxhits) {
                    valid = valid && rank < a.maxhits;
                    if (lane == 0) atomicAdd(a.over_maxhits, 1ull);
                }
            }
            const uint32_t n_keep = n_acc > a.maxhits ? a.maxhits : n_acc;

            if (a.flags & 2u) {                                   // sparse row (ref :1120-1182)
                // every element of the row is written exactly once
                uint16_t *row = a.sparse_rows + rd * row_len;
                if (lane == 0) row[0] = (uint16_t)n_keep;
                if (valid) {
                    row[1 + 2 * rank] = (uint16_t)acc_t;
                    row[2 + 2 * rank] = (uint16_t)sat_u16(acc_c);
                }
                for (uint32_t i = 1u + 2u * n_keep + lane; i < row_len; i += 64u) row[i] = 0;
            }
            if (a.flags & 1u) {                                   // fused top-2 (ref :1361-1411)
                // ascending-id scan with strict '>' == max count, ties to the smaller id
                const uint32_t cc  = sat_u16(acc_c);
                const uint32_t key = valid ? ((cc << 16) | (0xFFFFu - (acc_t & 0xFFFFu))) : 0u;
                uint32_t k1, k2, sum;
                if (n_acc <= 1u) {              // most reads hit no target or one: nothing to reduce
                    k1 = (uint32_t)__builtin_amdgcn_readlane((int)key, 0);
                    k2 = 0u;
                    sum = (uint32_t)__builtin_amdgcn_readlane((int)(valid ? cc : 0u), 0);
                } else {
                    k1  = wave_max_u32(key);
                    k2  = wave_max_u32(key == k1 ? 0u : key);
                    sum = wave_sum_u32(valid ? cc : 0u);
                }
                uint32_t out = 0;
                switch (lane) {
                case 0: out = sum & 0xFFFFu; break;
                case 1: out = k1 ? (0xFFFFu - (k1 & 0xFFFFu)) + 1u : 0u; break;
                case 2: out = k1 >> 16; break;
                case 3: out = k2 ? (0xFFFFu - (k2 & 0xFFFFu)) + 1u : 0u; break;
                case 4: out = k2 >> 16; break;
                default: break;
                }
                if (lane < 5u) a.final_rows[rd * 5u + lane] = (uint16_t)out;
            }
        }
        };   // run_group
        if (staged) run_group(std::true_type{}); else run_group(std::false_type{});
        rs = re;
        // the slice is rewritten by the next piece: keep the compiler from hoisting its stores
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        }   // pieces of the group
        // the slice is rewritten by the next group: keep the compiler from hoisting
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------
// merge / result (one thread per read), for the sharded path
// ---------------------------------------------------------------------------
// ref CuClarkDB.cu:1261-1355; keeps the maxhits smallest ids when the union is larger.
static __global__ void merge_rows_kernel(const uint16_t *A, const uint16_t *B, uint32_t row_len,
                                  uint64_t n_reads, uint16_t *out)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const uint16_t *ra = A + r * row_len, *rb = B + r * row_len;
    uint16_t *ro = out + r * row_len;
    const uint32_t maxhits = (row_len - 2u) / 2u;
    const uint32_t na = ra[0], nb = rb[0];
    uint16_t tmp[2 * 63 + 2];
    uint32_t ia = 0, ib = 0, n = 0;
    while ((ia < na || ib < nb) && n < maxhits) {
        uint16_t t, h;
        const uint16_t ta = ia < na ? ra[1 + 2 * ia] : (uint16_t)0xFFFF;
        const uint16_t tb = ib < nb ? rb[1 + 2 * ib] : (uint16_t)0xFFFF;
        if (ib >= nb || (ia < na && ta < tb))      { t = ta; h = ra[2 + 2 * ia]; ia++; }
        else if (ia >= na || tb < ta)              { t = tb; h = rb[2 + 2 * ib]; ib++; }
        else { t = ta; h = (uint16_t)sat_u16((uint32_t)ra[2 + 2 * ia] + rb[2 + 2 * ib]); ia++; ib++; }
        tmp[1 + 2 * n] = t; tmp[2 + 2 * n] = h; n++;
    }
    tmp[0] = (uint16_t)n;
    for (uint32_t i = 0; i < row_len; i++) ro[i] = i < 1u + 2u * n ? tmp[i] : (uint16_t)0;
}

// K-way form of the two kernels around it, for the multi-GPU combine: rows of up to MERGE_MAX_SRCS shards
// -> merged row (optional) and final row (optional) in one pass, one thread per read.  Same result as
// folding the sources pairwise through merge_rows_kernel and then result_rows_kernel: the union keeps the
// maxhits smallest ids, equal ids add (saturating), top-2 by ascending scan with strict '>'
// (reference CuClarkDB.cu:909-928 merges pairwise along a device tree, then :963-968).
static constexpr int MERGE_MAX_SRCS = 16;
struct RowSrcs {
    const uint16_t *p[MERGE_MAX_SRCS];
    uint32_t n;
};
static __global__ void merge_result_kernel(const RowSrcs S, uint32_t row_len, uint64_t n_reads,
                                           uint16_t *out_rows, uint16_t *out5)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const uint32_t maxhits = (row_len - 2u) / 2u;
    uint32_t idx[MERGE_MAX_SRCS], cnt[MERGE_MAX_SRCS];
#pragma unroll
    for (int w = 0; w < MERGE_MAX_SRCS; w++) {
        idx[w] = 0;
        cnt[w] = (uint32_t)w < S.n ? S.p[w][r * row_len] : 0u;
        if (cnt[w] > maxhits) cnt[w] = maxhits;                   // malformed input cannot run off the row
    }
    uint16_t best = 0, s_best = 0, ibest = 0, isbest = 0, sum = 0;
    uint16_t *ro = out_rows ? out_rows + r * row_len : nullptr;
    uint32_t n = 0;
    while (n < maxhits) {
        uint32_t t = 0x10000u;
#pragma unroll
        for (int w = 0; w < MERGE_MAX_SRCS; w++)
            if (idx[w] < cnt[w]) { const uint32_t tw = S.p[w][r * row_len + 1u + 2u * idx[w]]; t = tw < t ? tw : t; }
        if (t == 0x10000u) break;
        uint32_t h = 0;
#pragma unroll
        for (int w = 0; w < MERGE_MAX_SRCS; w++)
            if (idx[w] < cnt[w] && S.p[w][r * row_len + 1u + 2u * idx[w]] == t) { h += S.p[w][r * row_len + 2u + 2u * idx[w]]; idx[w]++; }
        const uint16_t sc = (uint16_t)sat_u16(h);
        if (ro) { ro[1u + 2u * n] = (uint16_t)t; ro[2u + 2u * n] = sc; }
        if (sc > best) { s_best = best; isbest = ibest; best = sc; ibest = (uint16_t)(t + 1u); }
        else if (sc > s_best) { s_best = sc; isbest = (uint16_t)(t + 1u); }
        sum = (uint16_t)(sum + sc);
        n++;
    }
    if (ro) {
        ro[0] = (uint16_t)n;
        for (uint32_t i = 1u + 2u * n; i < row_len; i++) ro[i] = 0;
    }
    if (out5) {
        uint16_t *o = out5 + r * 5;
        o[0] = sum; o[1] = ibest; o[2] = best; o[3] = isbest; o[4] = s_best;
    }
}

// ref CuClarkDB.cu:1361-1411
static __global__ void result_rows_kernel(const uint16_t *rows, uint32_t row_len, uint64_t n_reads,
                                   uint16_t *out5)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const uint16_t *row = rows + r * row_len;
    uint16_t best = 0, s_best = 0, ibest = 0, isbest = 0, sum = 0;
    const uint32_t count = row[0];
    for (uint32_t i = 0; i < count; i++) {
        const uint16_t sc = row[2 * i + 2];
        if (sc > best) { s_best = best; isbest = ibest; best = sc; ibest = (uint16_t)(row[2 * i + 1] + 1); }
        else if (sc > s_best) { s_best = sc; isbest = (uint16_t)(row[2 * i + 1] + 1); }
        sum = (uint16_t)(sum + sc);
    }
    uint16_t *o = out5 + r * 5;
    o[0] = sum; o[1] = ibest; o[2] = best; o[3] = isbest; o[4] = s_best;
}

// ---------------------------------------------------------------------------
// load-time re-layout: (sizes u8, keys u32, labels u16) -> bucket lines
// ---------------------------------------------------------------------------
static constexpr int RL_THREADS = 256;
static constexpr int RL_PER_THREAD = 4;
static constexpr int RL_BUCKETS = RL_THREADS * RL_PER_THREAD;   // buckets per workgroup

// histogram of bucket sizes (256 bins) -- picks the line size
static __global__ void size_hist_kernel(const uint8_t *sz, uint64_t n_buckets, unsigned long long *hist)
{
    __shared__ unsigned int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_buckets; i += stride)
        atomicAdd(&h[sz[i]], 1u);
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *s_wave, uint32_t &total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)inc, o, 64);
        if (lane >= (uint32_t)o) inc += t;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t pre = 0, tot = 0;
    for (uint32_t w = 0; w < RL_THREADS / 64; w++) { if (w < wave) pre += s_wave[w]; tot += s_wave[w]; }
    __syncthreads();
    total = tot;
    return pre + inc - v;
}

// per workgroup of RL_BUCKETS buckets: number of keys, and number of keys living in
// buckets larger than `cap`
static __global__ __launch_bounds__(RL_THREADS)
void block_sums_kernel(const uint8_t *sz, uint64_t n_buckets, uint32_t cap,
                       uint32_t *blk_keys, uint32_t *blk_ovf)
{
    __shared__ uint32_t s_a[RL_THREADS / 64], s_b[RL_THREADS / 64];
    const uint64_t b0 = (uint64_t)blockIdx.x * RL_BUCKETS + (uint64_t)threadIdx.x * RL_PER_THREAD;
    uint32_t keys = 0, ovf = 0;
    for (int i = 0; i < RL_PER_THREAD; i++) {
        const uint32_t c = (b0 + i < n_buckets) ? sz[b0 + i] : 0u;
        keys += c;
        if (c > cap) ovf += c;
    }
    uint32_t tk, to;
    block_exclusive_scan(keys, s_a, tk);
    block_exclusive_scan(ovf, s_b, to);
    if (threadIdx.x == 0) { blk_keys[blockIdx.x] = tk; blk_ovf[blockIdx.x] = to; }
}

template <int LINE, bool WIDE>
__global__ __launch_bounds__(RL_THREADS)
void fill_lines_kernel(const uint8_t *sz, const typename KeyOf<WIDE>::type *keys, const uint16_t *labels,
                       uint64_t n_buckets, const uint64_t *blk_key_off, const uint64_t *blk_ovf_off,
                       uint8_t *lines, typename KeyOf<WIDE>::type *ovf_keys, uint16_t *ovf_labels)
{
    using C = LineCfg<LINE, WIDE>;
    __shared__ uint32_t s_a[RL_THREADS / 64], s_b[RL_THREADS / 64];
    const uint64_t b0 = (uint64_t)blockIdx.x * RL_BUCKETS + (uint64_t)threadIdx.x * RL_PER_THREAD;
    uint32_t cnt[RL_PER_THREAD];
    uint32_t ksum = 0, osum = 0;
#pragma unroll
    for (int i = 0; i < RL_PER_THREAD; i++) {
        cnt[i] = (b0 + i < n_buckets) ? sz[b0 + i] : 0u;
        ksum += cnt[i];
        if (cnt[i] > (uint32_t)C::CAP) osum += cnt[i];
    }
    uint32_t tk, to;
    uint64_t koff = blk_key_off[blockIdx.x] + block_exclusive_scan(ksum, s_a, tk);
    uint64_t ooff = blk_ovf_off[blockIdx.x] + block_exclusive_scan(osum, s_b, to);

#pragma unroll
    for (int i = 0; i < RL_PER_THREAD; i++) {
        const uint64_t b = b0 + i;
        if (b >= n_buckets) break;
        const uint32_t c = cnt[i];
        uint32_t dw[C::DW];
#pragma unroll
        for (int j = 0; j < C::DW; j++) dw[j] = 0;
#pragma unroll
        for (int j = 0; j < C::CAP * C::KDW; j++) dw[j] = KEY_SENTINEL;
        if (c <= (uint32_t)C::CAP) {
#pragma unroll
            for (int j = 0; j < C::CAP; j++) {
                if ((uint32_t)j < c) {
                    const uint64_t key = keys[koff + j];
                    dw[j * C::KDW] = (uint32_t)key;
                    if (WIDE) dw[j * C::KDW + C::KDW - 1] = (uint32_t)(key >> 32);
                    const uint32_t lab = labels[koff + j];
                    dw[C::LAB0 + (j >> 1)] |= (j & 1) ? (lab << 16) : lab;
                }
            }
            dw[C::HDR] = c;
        } else {
            for (uint32_t j = 0; j < c; j++) {
                ovf_keys[ooff + j] = keys[koff + j];
                ovf_labels[ooff + j] = labels[koff + j];
            }
            dw[0] = (uint32_t)ooff; dw[1] = (uint32_t)(ooff >> 32); dw[2] = c;
            dw[C::HDR] = HDR_OVERFLOW;
            ooff += c;
        }
        uint4 *dst = reinterpret_cast<uint4 *>(lines + b * (uint64_t)LINE);
#pragma unroll
        for (int j = 0; j < LINE / 16; j++)
            dst[j] = make_uint4(dw[4 * j], dw[4 * j + 1], dw[4 * j + 2], dw[4 * j + 3]);
        koff += c;
    }
}

// widen file keys to the table's key width: 16-bit (k <= 23 full / k <= 20 light,
// reference main.cc:255-263) or 32-bit to 32 / 64 bits
template <typename IN, typename OUT>
__global__ void widen_keys_kernel(const IN *in, uint64_t n, OUT *out)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (OUT)in[i];
}

} // namespace mc
