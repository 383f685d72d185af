// mc_skm.hpp -- the super-k-mer index: the same database, stored RELATIVE to its minimizers.
//
// Why.  The minimizer index (mc_minimizer.hpp) stores every k-mer whole: 8 + 2 bytes in a 12-slot line.  The k-mers
// of a genome come in runs that share a minimizer (a super-k-mer: ~5 consecutive k-mers at 9 windows), related genomes
// put the runs of several targets behind ONE minimizer (16-20 k-mers against 12 slots), and the lines that overflow
// cost every read that crosses them a dependent fetch (1.1 ms of the genome-shaped table's 7.1).  A
// k-mer that contains the m-mer M at offset o is M plus W - 1 = 8 flank bases: o on the left, 8 - o on the right.
// All k-mers of one genome locus around M share ONE pair of flanks:
//
//   record (16 bytes) = hash(M) 46 bits | presence 9 bits (which offsets o are stored) | L 8 bases | R 8 bases | label
//                       up to 9 k-mers; 8 record slots per 128-byte line
//
// and a read's run of k-mers around one minimizer is matched against a record ONCE: the number of matching bases
// outward from M on either side bounds the offsets that can hit (lmatch >= o, rmatch >= 8 - o), the presence bits say
// which of them are stored.  The eight lanes that fetch a line hold one record each and match it in registers: a
// line is never parked in LDS, nothing is looked up per k-mer.
//
// Exactness.  hash() is a BIJECTION of the 2m-bit canonical m-mer (a Feistel step between two odd multiplications),
// so equal hashes are equal m-mers, and (M, o, flanks) is the k-mer: a hit is reported iff the canonical k-mer is
// stored, with its label -- the answer of reference src/CuClarkDB.cu:1216-1247, not its layout.  A k-mer whose
// smallest m-mer hash is reached by several of its windows is stored under every one of them (the lookup may come
// from either strand and picks whichever window its tie-break lands on); an m-mer that is its own reverse complement
// is stored in both orientations.
//
// The ordering key of a position is a double in [1, 2) as in mc_minimizer.hpp -- one v_min_f64 per window element --
// whose low five mantissa bits carry the position (mod 16) and the strand of the m-mer: the window minimum then
// brings WHERE the minimizer sits and HOW it is oriented along for free.
#pragma once

#include "mc_minimizer.hpp"

namespace mc {
namespace sk {

#define SK_HD __host__ __device__ __forceinline__

static constexpr int SK_W = mz::MZ_MAXW;            // windows per k-mer
static constexpr int SK_FL = SK_W - 1;              // flank bases of a k-mer around its minimizer
static_assert(SK_W == 9, "record layout: 8 + 8 flank bases in one dword, 9 presence bits");
static constexpr int SK_SLOTS = 8;                  // 16-byte slots per 128-byte line
static constexpr int SK_LINE = 128;
static constexpr uint32_t SK_HDR = 0x80000000u;     // dword 3 of slot 7: the slot is the line's header, not a record
static constexpr uint32_t SK_LIN_MAX = 3u;          // lines of a linear chain (7 + 3 * 8 = 31 records behind one line)
static constexpr uint32_t SK_WAVE_MAX = 64u;        // entries of a line one wave turns into records (more: hashed chains)
static constexpr uint32_t SK_PROBES = 4u;           // lines a k-mer of a hashed chain may sit behind its own
static constexpr uint32_t SK_CHAIN_LOAD_NUM = 5u, SK_CHAIN_LOAD_DEN = 2u;   // entries aimed at per hashed chain line: 2.5 of 8

// k-mer lengths the index serves: the m-mer hash needs 34 <= 2m <= 46 bits
__host__ __device__ inline bool sk_supported(uint32_t k) { const uint32_t m = mz::mmer_len(k); return k <= 31u && 2u * m >= 34u && 2u * m <= 46u; }

// ---------------------------------------------------------------------------
// host + device: hash, entries of a stored k-mer, records
// ---------------------------------------------------------------------------
SK_HD uint64_t sk_revcomp(uint64_t x, uint32_t len)
{
#ifdef __HIP_DEVICE_COMPILE__
    return mc::revcomp(x, len);
#endif
    uint64_t r = 0;
    x = ~x;
    for (uint32_t i = 0; i < len; i++) { r = (r << 2) | (x & 3ull); x >>= 2; }
    return r;
}

// the 2m-bit canonical m-mer -> 2m bits, a bijection: a = lo * C1 (odd), h = hi ^ (top bits of a), b = (a + h * C2) * C3.
// Returned as mantissa bits of the ordering key: b in bits 51..20, h in bits 19..(20 - B), zeros below (B = 2m - 32).
SK_HD uint64_t sk_key_bits(uint64_t cw, uint32_t m)
{
    const uint32_t B = 2u * m - 32u;
    const uint32_t lo = (uint32_t)cw, hi = (uint32_t)(cw >> 32);
    const uint32_t a = lo * 0x9E3779B1u;
    const uint32_t h = hi ^ (a >> (32u - B));
    const uint32_t b = (a + (h & 0xFFFFFFu) * 0x85EBCBu) * 0xC2B2AE35u;
    return ((uint64_t)b << 20) | ((uint64_t)h << (20u - B));
}
static constexpr uint64_t SK_ONE = 0x3FF0000000000000ull;         // exponent of a double in [1, 2)
static constexpr uint32_t SK_LOW = 31u;                            // position (bits 4..1) and strand (bit 0) of the m-mer

SK_HD uint32_t sk_line_of(uint64_t K, uint32_t n_lines)
{
    const uint32_t h = ((uint32_t)(K >> 32) * 0x9E3779B1u) ^ ((uint32_t)K >> 5);
#ifdef __HIP_DEVICE_COMPILE__
    return __umulhi(h, n_lines);
#else
    return (uint32_t)(((uint64_t)h * n_lines) >> 32);
#endif
}
SK_HD uint32_t sk_part_of(uint64_t K, uint32_t n_parts)
{
    const uint32_t h = ((uint32_t)K >> 5) * 0x85EBCA6Bu + (uint32_t)(K >> 32);
#ifdef __HIP_DEVICE_COMPILE__
    return __umulhi(h, n_parts);
#else
    return (uint32_t)(((uint64_t)h * n_parts) >> 32);
#endif
}

// A slot of a line: a record, or -- with one presence bit -- a single stored k-mer (an "entry": what the build
// passes scatter, and what a hashed chain holds).
//   d0  hash bits 31..5 of the key's low word (bits 4..0 = 0)
//   d1  bits 19..0: hash bits of the key's high word; bits 28..20: presence (bit 20 + o: the k-mer with M at offset o)
//   d2  bits 15..0: L, the 8 bases before M (the base next to M in bits 1..0); bits 31..16: R, the 8 bases behind M
//       (the base next to M in bits 31..30); bases no stored k-mer covers are 0
//   d3  label (bits 15..0)
struct SkSlot { uint32_t d0, d1, d2, d3; };

SK_HD uint32_t sk_ctz(uint32_t v)
{
#ifdef __HIP_DEVICE_COMPILE__
    return (uint32_t)__ffs((int)v) - 1u;
#else
    return (uint32_t)__builtin_ctz(v);
#endif
}
SK_HD uint32_t sk_clz(uint32_t v)
{
#ifdef __HIP_DEVICE_COMPILE__
    return (uint32_t)__clz((int)v);
#else
    return (uint32_t)__builtin_clz(v);
#endif
}

// the k-mer y (M forward at offset o) as an entry
SK_HD SkSlot sk_make_entry(uint64_t y, uint32_t k, uint32_t o, uint64_t key_bits, uint32_t label)
{
    const uint32_t L = (uint32_t)(y >> (2u * (k - o)));                                   // o bases
    const uint32_t R = (uint32_t)(y & ((1ull << (2u * ((uint32_t)SK_FL - o))) - 1ull));    // FL - o bases
    SkSlot e;
    e.d0 = (uint32_t)key_bits & ~SK_LOW;
    e.d1 = ((uint32_t)(key_bits >> 32) & 0xFFFFFu) | (1u << (20u + o));
    e.d2 = (((R << (2u * o)) & 0xFFFFu) << 16) | L;
    e.d3 = label;
    return e;
}

// every entry of the stored canonical k-mer c: one per window that reaches the smallest hash (both strands of an
// m-mer that is its own reverse complement), without repeats; at most 2 W
SK_HD int sk_entries(uint64_t c, uint32_t k, uint32_t m, uint32_t label, SkSlot *out, uint64_t *key_out)
{
    const uint64_t mmask = (1ull << (2u * m)) - 1ull;
    const uint64_t rcc = sk_revcomp(c, k);
    uint64_t kb[SK_W];
    uint32_t fw = 0, rv = 0;
    uint64_t best = ~0ull;
    for (uint32_t j = 0; j < (uint32_t)SK_W; j++) {
        const uint64_t w = (c >> (2u * (k - m - j))) & mmask, rw = sk_revcomp(w, m);
        kb[j] = sk_key_bits(w < rw ? w : rw, m);
        fw |= (w <= rw ? 1u : 0u) << j;
        rv |= (rw <= w ? 1u : 0u) << j;
        best = kb[j] < best ? kb[j] : best;
    }
    int n = 0;
    for (uint32_t j = 0; j < (uint32_t)SK_W; j++) {
        if (kb[j] != best) continue;
        for (int side = 0; side < 2; side++) {
            if (!((side ? rv : fw) >> j & 1u)) continue;
            const SkSlot e = side ? sk_make_entry(rcc, k, (uint32_t)SK_FL - j, best, label) : sk_make_entry(c, k, j, best, label);
            bool seen = false;
            for (int t = 0; t < n; t++) seen = seen || (out[t].d1 == e.d1 && out[t].d2 == e.d2);
            if (!seen) out[n++] = e;
        }
    }
    *key_out = SK_ONE | best;
    return n;
}

// can the single k-mer `e` join record `r` (same minimizer, same target, flanks that agree where both know them)?
SK_HD bool sk_consistent(const SkSlot &r, const SkSlot &e)
{
    if (r.d0 != e.d0 || ((r.d1 ^ e.d1) & 0xFFFFFu) != 0u || r.d3 != e.d3) return false;
    const uint32_t pr = (r.d1 >> 20) & 0x1FFu, o = sk_ctz(e.d1 >> 20);
    const uint32_t lcov = 31u - sk_clz(pr), rcov = (uint32_t)SK_FL - sk_ctz(pr);       // bases the record knows on either side
    const uint32_t cl = o < lcov ? o : lcov, cr = ((uint32_t)SK_FL - o) < rcov ? ((uint32_t)SK_FL - o) : rcov;
    const uint32_t x = r.d2 ^ e.d2;
    const bool l_ok = ((x & 0xFFFFu) & ((1u << (2u * cl)) - 1u)) == 0u;
    const bool r_ok = ((x >> 16) >> (16u - 2u * cr)) == 0u;
    return l_ok && r_ok;
}
SK_HD void sk_merge(SkSlot &r, const SkSlot &e)
{
    const uint32_t pr = (r.d1 >> 20) & 0x1FFu, o = sk_ctz(e.d1 >> 20);
    const uint32_t lcov = 31u - sk_clz(pr), rcov = (uint32_t)SK_FL - sk_ctz(pr);
    if (o > lcov) r.d2 = (r.d2 & 0xFFFF0000u) | (e.d2 & 0xFFFFu);
    if ((uint32_t)SK_FL - o > rcov) r.d2 = (r.d2 & 0xFFFFu) | (e.d2 & 0xFFFF0000u);
    r.d1 |= e.d1 & (0x1FFu << 20);
}

// What a run of a read asks a record: the run's k-mers hold M at offsets o_lo .. o_hi, `lr` holds the o_hi bases
// before M (low half, the base next to M in bits 1..0) and the FL - o_lo bases behind it (high half, the base next
// to M in bits 31..30).  Returns how many of the run's k-mers the record holds (their label: r.d3).
SK_HD uint32_t sk_match_flanks(uint32_t d1, uint32_t d2, uint32_t o_lo, uint32_t o_hi, uint32_t lr)
{
    const uint32_t x = d2 ^ lr;
    const uint32_t lmatch = sk_ctz((x & 0xFFFFu) | 0x10000u) >> 1;          // bases that agree, outward from M
    const uint32_t rmatch = sk_clz((x & 0xFFFF0000u) | 0x8000u) >> 1;
    const uint32_t lo = o_lo > (uint32_t)SK_FL - rmatch ? o_lo : (uint32_t)SK_FL - rmatch;
    const uint32_t hi = o_hi < lmatch ? o_hi : lmatch;
    if (hi < lo) return 0u;
    const uint32_t mask = ((2u << hi) - 1u) & ~((1u << lo) - 1u);
#ifdef __HIP_DEVICE_COMPILE__
    return (uint32_t)__popc((d1 >> 20) & mask & 0x1FFu);
#else
    return (uint32_t)__builtin_popcount((d1 >> 20) & mask & 0x1FFu);
#endif
}
SK_HD bool sk_same_minimizer(const SkSlot &r, uint32_t kd0, uint32_t kd1)       // a record (not a header) of the run's minimizer
{
    return ((r.d0 ^ kd0) | ((r.d1 ^ kd1) & 0xFFFFFu)) == 0u && !(r.d3 & SK_HDR);
}
SK_HD uint32_t sk_match(const SkSlot &r, uint32_t kd0, uint32_t kd1, uint32_t o_lo, uint32_t o_hi, uint32_t lr)
{
    return sk_same_minimizer(r, kd0, kd1) ? sk_match_flanks(r.d1, r.d2, o_lo, o_hi, lr) : 0u;
}

// The descriptor of a run from what the kernel's front half leaves: the run's key K (hash | position of the m-mer
// mod 16 | its strand), the part-relative positions of its first and last k-mer, and their "ends" (the first 8 bases
// in the high half, the last 8 in the low half, of the k-mer as the read spells it).
struct SkRun { uint32_t kd0, kd1, o_lo, o_hi, lr; };
SK_HD uint32_t sk_rev16(uint32_t e)          // reverse complement of 16 bases in a dword
{
#ifdef __HIP_DEVICE_COMPILE__
    const uint32_t r = __brev(e);                // bases in reverse order, the two bits of each swapped: swap them back, complement
    return ~(((r >> 1) & 0x55555555u) | ((r << 1) & 0xAAAAAAAAu));
#endif
    uint32_t v = ~e;
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
    return (v >> 16) | (v << 16);
}
SK_HD SkRun sk_run(uint64_t K, uint32_t p_a, uint32_t p_b, uint32_t ends_a, uint32_t ends_b)
{
    const uint32_t klo = (uint32_t)K, s = klo & 1u, pm = klo >> 1;
    const uint32_t ja = (pm - p_a) & 15u, jb = (pm - p_b) & 15u;             // offset of the m-mer in the first / last k-mer: ja >= jb
    SkRun r;
    r.kd0 = klo & ~SK_LOW;
    r.kd1 = (uint32_t)(K >> 32) & 0xFFFFFu;
    uint32_t el, er;
    if (s) { r.o_hi = ja; r.o_lo = jb; el = ends_a; er = ends_b; }            // M as the read spells it: L from the first k-mer
    else   { r.o_lo = (uint32_t)SK_FL - ja; r.o_hi = (uint32_t)SK_FL - jb; el = sk_rev16(ends_b); er = sk_rev16(ends_a); }
    const uint32_t L = (el >> 16) >> (16u - 2u * r.o_hi);
    const uint32_t R = ((er & 0xFFFFu) << (2u * r.o_lo)) & 0xFFFFu;
    r.lr = (R << 16) | L;
    return r;
}

// hashed chains (a line with more entries than a wave turns into records): which chain line a single k-mer sits in
SK_HD uint32_t sk_entry_hash(uint32_t kd0, uint32_t o, uint32_t lr_masked)
{
    return ((lr_masked * 0x9E3779B1u) ^ (kd0 >> 7) ^ (o * 0x85EBCA6Bu)) * 0xC2B2AE35u;
}
// the flanks of the k-mer at offset o alone, out of a run's (or a record's) flanks
SK_HD uint32_t sk_lr_of(uint32_t lr, uint32_t o)
{
    const uint32_t lmask = (1u << (2u * o)) - 1u;
    const uint32_t rmask = (0xFFFF0000u << (2u * o)) & 0xFFFF0000u;
    return lr & (lmask | rmask);
}
// Bloom word of an overflowing line over the minimizers of the records that are not in its first line
SK_HD void sk_bloom_bits(uint32_t kd0, uint32_t *w0, uint32_t *w1) { *w0 = 1u << ((kd0 >> 20) & 31u); *w1 = 1u << ((kd0 >> 25) & 31u); }
SK_HD bool sk_bloom_pass(uint32_t w0, uint32_t w1, uint32_t kd0) { return ((w0 >> ((kd0 >> 20) & 31u)) & (w1 >> ((kd0 >> 25) & 31u)) & 1u) != 0u; }

// extra lines of a line with n_rec records out of n_ent entries; header word (dword 3 of slot 7)
//   bits 7..0: lines of the linear chain | bits 15..8: s (2^s hashed chain lines, + SK_PROBES spare) | bit 31
SK_HD bool sk_is_crowded(uint32_t n_ent, uint32_t n_rec) { return n_ent > SK_WAVE_MAX || n_rec > (uint32_t)(SK_SLOTS - 1) + (uint32_t)SK_SLOTS * SK_LIN_MAX; }
SK_HD uint32_t sk_chain_log(uint32_t n_ent)
{
    const uint32_t want = (n_ent * SK_CHAIN_LOAD_DEN + SK_CHAIN_LOAD_NUM - 1u) / SK_CHAIN_LOAD_NUM;
    uint32_t s = 1u;
    while ((1u << s) < want) s++;
    return s;
}
SK_HD uint32_t sk_extras_of(uint32_t n_ent, uint32_t n_rec)
{
    if (n_rec <= (uint32_t)SK_SLOTS && n_ent <= SK_WAVE_MAX) return 0u;
    if (sk_is_crowded(n_ent, n_rec)) return (1u << sk_chain_log(n_ent)) + SK_PROBES;
    return (n_rec - (uint32_t)(SK_SLOTS - 1) + (uint32_t)SK_SLOTS - 1u) / (uint32_t)SK_SLOTS;
}


// ---------------------------------------------------------------------------
// build (device).  The table arrives in bucket-order chunks, twice (mc_index_begin .. mc_index_end):
//   pass 0  entries per FINE line (a line space d times finer than the final one; the records of a minimizer never
//           leave their fine line, and sk_line_of(K, n_fine) / d == sk_line_of(K, n_fine / d), so the final line count
//           is chosen AFTER the table has been seen: the records of d neighbouring fine lines simply add up)
//   scan    offsets of the fine lines into one array of entries
//   pass 1  the entries, scattered to their fine line's range
//   end     records per fine line (sk_records_kernel) -> d (sk_eval_kernel on the host's candidates) -> lines,
//           extra lines per line (sk_extras_kernel, scan) -> sk_encode_kernel writes the lines
// ---------------------------------------------------------------------------
template <int PASS, bool WIDE>
__global__ __launch_bounds__(RL_THREADS)
void sk_build_kernel(const uint8_t *sz, const typename KeyOf<WIDE>::type *keys, const uint16_t *labels,
                     uint64_t n_buckets, uint64_t n_keys, uint64_t bucket0, uint64_t htsize, const uint64_t *blk_key_off,
                     uint32_t k, uint32_t m, uint32_t part, uint32_t n_parts, uint32_t n_fine,
                     uint32_t *count, const uint32_t *count0, const uint32_t *off32, const uint64_t *blk_base,
                     SkSlot *entries, unsigned int *failed)
{
    __shared__ uint32_t s_a[RL_THREADS / 64];
    const uint64_t b0 = (uint64_t)blockIdx.x * RL_BUCKETS + (uint64_t)threadIdx.x * RL_PER_THREAD;
    uint32_t cnt[RL_PER_THREAD], ksum = 0;
#pragma unroll
    for (int i = 0; i < RL_PER_THREAD; i++) { cnt[i] = (b0 + i < n_buckets) ? sz[b0 + i] : 0u; ksum += cnt[i]; }
    uint32_t tk;
    uint64_t koff = blk_key_off[blockIdx.x] + block_exclusive_scan(ksum, s_a, tk);
    for (int i = 0; i < RL_PER_THREAD; i++) {
        const uint64_t b = b0 + i;
        if (b >= n_buckets) break;
        for (uint32_t j = 0; j < cnt[i]; j++) {
            if (koff + j >= n_keys) break;                    // bucket sizes that announce more k-mers than were passed (flagged elsewhere)
            const uint64_t c = (uint64_t)keys[koff + j] * htsize + (bucket0 + b);     // the canonical k-mer
            SkSlot e[2 * SK_W];
            uint64_t K;
            const int n = sk_entries(c, k, m, labels[koff + j], e, &K);
            if (n_parts > 1u && sk_part_of(K, n_parts) != part) continue;
            const uint32_t f = sk_line_of(K, n_fine);
            const uint32_t slot = atomicAdd(&count[f], (uint32_t)n);
            if (PASS == 1) {
                if (slot + (uint32_t)n > count0[f]) { atomicOr(failed, 2u); continue; }      // a second pass that is not the first again
                SkSlot *dst = entries + blk_base[f >> 10] + off32[f] + slot;
                for (int t = 0; t < n; t++) *reinterpret_cast<uint4 *>(dst + t) = make_uint4(e[t].d0, e[t].d1, e[t].d2, e[t].d3);
            }
        }
        koff += cnt[i];
    }
}
static_assert(RL_BUCKETS == 1024, "blk_base is indexed by line >> 10");

// exclusive offsets inside a workgroup's RL_BUCKETS counters + the workgroup's total
static __global__ __launch_bounds__(RL_THREADS)
void sk_scan_kernel(const uint32_t *cnt, uint64_t n, uint32_t *off32, uint32_t *blk_tot)
{
    __shared__ uint32_t s_a[RL_THREADS / 64];
    const uint64_t b0 = (uint64_t)blockIdx.x * RL_BUCKETS + (uint64_t)threadIdx.x * RL_PER_THREAD;
    uint32_t c[RL_PER_THREAD], sum = 0;
#pragma unroll
    for (int i = 0; i < RL_PER_THREAD; i++) { c[i] = (b0 + i < n) ? cnt[b0 + i] : 0u; sum += c[i]; }
    uint32_t tot;
    uint32_t o = block_exclusive_scan(sum, s_a, tot);
#pragma unroll
    for (int i = 0; i < RL_PER_THREAD; i++) { if (b0 + i < n) off32[b0 + i] = o; o += c[i]; }
    if (threadIdx.x == 0) blk_tot[blockIdx.x] = tot;
}
// exclusive u64 offsets of the workgroup totals (one workgroup), the grand total behind the last
static __global__ __launch_bounds__(256)
void sk_scan_blocks_kernel(const uint32_t *blk, uint32_t n, uint64_t *off)
{
    __shared__ uint64_t s_sum[256];
    const uint32_t per = (n + 255u) / 256u;
    const uint32_t lo = threadIdx.x * per < n ? threadIdx.x * per : n, hi = lo + per < n ? lo + per : n;
    uint64_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += blk[i];
    s_sum[threadIdx.x] = sum;
    __syncthreads();
    uint64_t pre = 0, tot = 0;
    for (uint32_t t = 0; t < 256u; t++) { if (t < threadIdx.x) pre += s_sum[t]; tot += s_sum[t]; }
    for (uint32_t i = lo; i < hi; i++) { off[i] = pre; pre += blk[i]; }
    if (threadIdx.x == 0) off[n] = tot;
}

// sum and sum of squares of the counters (MC_INDEX=auto: how the k-mers clump)
static __global__ __launch_bounds__(256)
void sk_moments_kernel(const uint32_t *cnt, uint64_t n, unsigned long long *out)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long s1 = 0, s2 = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { const unsigned long long v = cnt[i]; s1 += v; s2 += v * v; }
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    if ((threadIdx.x & 63) == 0 && s1) { atomicAdd(&out[0], s1); atomicAdd(&out[1], s2); }
}

// One wave, up to 64 entries (lane i holds entry i): first fit into records; lane r < n_rec ends with record r.
__device__ __forceinline__ uint32_t sk_form_records(SkSlot &e, uint32_t n, uint32_t lane)
{
    SkSlot rec{0u, 0u, 0u, 0u};
    uint32_t n_rec = 0;
    for (uint32_t i = 0; i < n; i++) {
        const SkSlot E{lane_bcast(e.d0, i), lane_bcast(e.d1, i), lane_bcast(e.d2, i), lane_bcast(e.d3, i)};
        bool ok = false;
        if (lane < n_rec) ok = sk_consistent(rec, E);
        const uint64_t mm = __ballot(ok);
        if (mm) {
            if (lane == (uint32_t)__ffsll((unsigned long long)mm) - 1u) sk_merge(rec, E);
        } else {
            if (lane == n_rec) rec = E;
            n_rec++;
        }
    }
    e = rec;
    return n_rec;
}

// records per FINE line (0xFFFF: more entries than a wave handles)
static __global__ __launch_bounds__(256)
void sk_records_kernel(const uint32_t *count0, const uint32_t *off32, const uint64_t *blk_base, const SkSlot *entries,
                       uint32_t n_fine, uint32_t *nrec)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t base = wave * 64u; base < n_fine; base += n_waves * 64u) {
        const uint64_t mine = base + lane;
        const uint32_t n_mine = mine < n_fine ? count0[mine] : 0u;
        if (mine < n_fine) nrec[mine] = n_mine <= 1u ? n_mine : (n_mine > SK_WAVE_MAX ? 0xFFFFu : 0u);
        uint64_t todo = __ballot(n_mine > 1u && n_mine <= SK_WAVE_MAX);
        while (todo) {
            const uint32_t l = (uint32_t)__ffsll((unsigned long long)todo) - 1u;
            todo &= todo - 1u;
            const uint64_t f = base + l;
            const uint32_t n = lane_bcast(n_mine, l);
            const SkSlot *src = entries + blk_base[f >> 10] + off32[f];
            SkSlot e{0u, 0u, 0u, 0u};
            if (lane < n) { const uint4 v = *reinterpret_cast<const uint4 *>(src + lane); e = SkSlot{v.x, v.y, v.z, v.w}; }
            const uint32_t n_rec = sk_form_records(e, n, lane);
            if (lane == 0) nrec[f] = n_rec;
        }
    }
}

// entries and records of the final line `line` = fine lines [line * d, (line + 1) * d)
__device__ __forceinline__ void sk_line_totals(const uint32_t *count0, const uint32_t *nrec, uint64_t line, uint32_t d, uint32_t &n_ent, uint32_t &n_rec)
{
    n_ent = 0; n_rec = 0;
    bool big = false;
    for (uint32_t t = 0; t < d; t++) {
        n_ent += count0[line * d + t];
        const uint32_t r = nrec[line * d + t];
        big = big || r == 0xFFFFu;
        n_rec += r == 0xFFFFu ? 0u : r;
    }
    if (big || n_ent > SK_WAVE_MAX) n_rec = 0xFFFFu;
}

// what a merge factor d would give: out[0] lines that overflow, [1] extra lines, [2] lines with hashed chains,
// [3] largest line (entries), [4] records in all, [5] lines that hold anything
static __global__ __launch_bounds__(256)
void sk_eval_kernel(const uint32_t *count0, const uint32_t *nrec, uint64_t n_lines, uint32_t d, unsigned long long *out)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long ov = 0, ex = 0, cr = 0, mx = 0, rc = 0, ne = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_lines; i += stride) {
        uint32_t n_ent, n_rec;
        sk_line_totals(count0, nrec, i, d, n_ent, n_rec);
        const uint32_t x = sk_extras_of(n_ent, n_rec);
        ov += x ? 1u : 0u; ex += x; cr += (x && sk_is_crowded(n_ent, n_rec)) ? 1u : 0u;
        mx = n_ent > mx ? n_ent : mx; rc += n_rec == 0xFFFFu ? n_ent : n_rec; ne += n_ent ? 1u : 0u;
    }
    for (int o = 32; o > 0; o >>= 1) {
        ov += __shfl_xor(ov, o, 64); ex += __shfl_xor(ex, o, 64); cr += __shfl_xor(cr, o, 64); rc += __shfl_xor(rc, o, 64); ne += __shfl_xor(ne, o, 64);
        const unsigned long long t = __shfl_xor(mx, o, 64); mx = t > mx ? t : mx;
    }
    if ((threadIdx.x & 63) == 0) {
        if (ov) atomicAdd(&out[0], ov);
        if (ex) atomicAdd(&out[1], ex);
        if (cr) atomicAdd(&out[2], cr);
        atomicMax(&out[3], mx);
        if (rc) atomicAdd(&out[4], rc);
        if (ne) atomicAdd(&out[5], ne);
    }
}

static __global__ __launch_bounds__(256)
void sk_extras_kernel(const uint32_t *count0, const uint32_t *nrec, uint64_t n_lines, uint32_t d, uint32_t *xcnt)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_lines; i += stride) {
        uint32_t n_ent, n_rec;
        sk_line_totals(count0, nrec, i, d, n_ent, n_rec);
        xcnt[i] = sk_extras_of(n_ent, n_rec);
    }
}

// the lines: one wave per final line that holds anything
static __global__ __launch_bounds__(256)
void sk_encode_kernel(const uint32_t *count0, const uint32_t *off32, const uint64_t *blk_base, const SkSlot *entries,
                      const uint32_t *nrec, uint32_t n_lines, uint32_t d, const uint32_t *xoff32, const uint64_t *xblk_base,
                      uint8_t *lines, uint8_t *extra, unsigned int *failed)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t base = wave * 64u; base < n_lines; base += n_waves * 64u) {
        const uint64_t mine = base + lane;
        uint32_t n_mine = 0, r_mine = 0;
        if (mine < n_lines) sk_line_totals(count0, nrec, mine, d, n_mine, r_mine);
        uint64_t todo = __ballot(n_mine != 0u);
        while (todo) {
            const uint32_t l = (uint32_t)__ffsll((unsigned long long)todo) - 1u;
            todo &= todo - 1u;
            const uint64_t line = base + l;
            const uint32_t n = lane_bcast(n_mine, l), r_tot = lane_bcast(r_mine, l);
            const uint64_t f0 = line * d;
            const SkSlot *src = entries + blk_base[f0 >> 10] + off32[f0];
            uint4 *first = reinterpret_cast<uint4 *>(lines + line * SK_LINE);
            const uint32_t n_x = sk_extras_of(n, r_tot);
            const uint64_t xb = n_x ? xblk_base[line >> 10] + xoff32[line] : 0ull;
            uint4 *more = reinterpret_cast<uint4 *>(extra + xb * SK_LINE);
            if (!sk_is_crowded(n, r_tot) && n <= SK_WAVE_MAX) {
                SkSlot e{0u, 0u, 0u, 0u};
                if (lane < n) { const uint4 v = *reinterpret_cast<const uint4 *>(src + lane); e = SkSlot{v.x, v.y, v.z, v.w}; }
                const uint32_t n_rec = sk_form_records(e, n, lane);
                if (n_rec != r_tot) { if (lane == 0) atomicOr(failed, 8u); continue; }        // the counting pass saw other records
                // The records of one minimizer go into neighbouring slots: the query kernel's lanes hold slots q and q + 4 of a
                // line, so up to four records of a minimizer are matched side by side (a genus of four behind one m-mer), and
                // the layout no longer depends on the order the build's atomics ran in.  `at` = rank by (minimizer, formation order).
                uint32_t at = 0;
                {
                    const uint64_t mine = ((uint64_t)(e.d1 & 0xFFFFFu) << 32) | e.d0;
                    for (uint32_t j = 0; j < n_rec; j++) {
                        const uint64_t other = ((uint64_t)(lane_bcast(e.d1, j) & 0xFFFFFu) << 32) | lane_bcast(e.d0, j);
                        at += (other < mine || (other == mine && j < lane)) ? 1u : 0u;
                    }
                }
                if (n_rec <= (uint32_t)SK_SLOTS) {
                    if (lane < n_rec) first[at] = make_uint4(e.d0, e.d1, e.d2, e.d3);
                } else {
                    // 7 records + header in the first line, the others in a linear chain; the header's Bloom word covers
                    // the minimizers of the chain's records
                    uint32_t w0 = 0, w1 = 0;
                    if (lane < n_rec && at < (uint32_t)(SK_SLOTS - 1)) first[at] = make_uint4(e.d0, e.d1, e.d2, e.d3);
                    else if (lane < n_rec) {
                        more[at - (uint32_t)(SK_SLOTS - 1)] = make_uint4(e.d0, e.d1, e.d2, e.d3);
                        sk_bloom_bits(e.d0, &w0, &w1);
                    }
                    for (int o = 32; o > 0; o >>= 1) { w0 |= (uint32_t)__shfl_xor((int)w0, o, 64); w1 |= (uint32_t)__shfl_xor((int)w1, o, 64); }
                    if (lane == 0) first[SK_SLOTS - 1] = make_uint4(w0, w1, (uint32_t)xb, SK_HDR | n_x);
                }
            } else {
                // hashed chains: every entry on its own, in the chain line its hash picks or one of the SK_PROBES - 1 behind it
                const uint32_t s = sk_chain_log(n);
                uint32_t far = 0;
                for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
                    if (i0 + lane < n) {
                        const uint4 v = *reinterpret_cast<const uint4 *>(src + i0 + lane);
                        const uint32_t o = sk_ctz(v.y >> 20);
                        const uint32_t h = sk_entry_hash(v.x, o, v.z) >> (32u - s);
                        bool placed = false;
                        for (uint32_t pr = 0; pr < SK_PROBES && !placed; pr++) {
                            uint4 *L = more + (uint64_t)(h + pr) * SK_SLOTS;
                            for (uint32_t sl = 0; sl < (uint32_t)SK_SLOTS && !placed; sl++) {
                                unsigned long long *key = reinterpret_cast<unsigned long long *>(L + sl);
                                if (*reinterpret_cast<volatile unsigned long long *>(key) != 0ull) continue;
                                if (atomicCAS(key, 0ull, ((unsigned long long)v.y << 32) | v.x) == 0ull) {
                                    reinterpret_cast<uint32_t *>(L + sl)[2] = v.z;
                                    reinterpret_cast<uint32_t *>(L + sl)[3] = v.w;
                                    placed = true;
                                    far = pr > far ? pr : far;
                                }
                            }
                        }
                        if (!placed) atomicOr(failed, 1u);
                    }
                }
                for (int o = 32; o > 0; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)far, o, 64); far = t > far ? t : far; }
                if (lane == 0) first[SK_SLOTS - 1] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, (uint32_t)xb, SK_HDR | (s << 8) | ((far + 1u) << 16));
            }
        }
    }
}

// ---------------------------------------------------------------------------
// query
// ---------------------------------------------------------------------------
struct SkArgs {
    QueryArgs q;               // reads, outputs, k, maxhits, flags (lines / shard range unused)
    const uint8_t *lines;      // the lines this context owns
    const uint8_t *extra;      // extra lines (linear and hashed chains)
    uint32_t n_lines;          // lines of THIS context (every part of a table has the same number)
    uint32_t part, n_parts;    // MZ_LINES: this context answers for the minimizers with sk_part_of(K) == part
    uint32_t m;
};

static constexpr int SK_RUNS = 32;            // runs matched per batch: 4 rounds of 8 lines, 8 lanes a line
#ifdef MC_SK_MARK          // measurement builds only: markers in the instruction stream (tools/isa_sections.py counts between them)
#define SK_MARK(n) asm volatile("s_nop " #n)
#else
#define SK_MARK(n) do { } while (0)
#endif
#ifndef MC_SK_MIN_WAVES
#define MC_SK_MIN_WAVES 7
#endif
// per-wave LDS: staged containers | m-mer keys of a step | ends of the step's k-mers | run descriptors | read offsets
static constexpr int SK_LDS_KEYS = mz::MZ_LDS_SLICE;
static constexpr int SK_LDS_ENDS = SK_LDS_KEYS + ((64 * mz::MZ_NS + SK_W + 3) * 8 + 15) / 16 * 16;
static constexpr int SK_LDS_DESC = SK_LDS_ENDS + 64 * mz::MZ_NS * 4;
static constexpr int SK_LDS_RDPTR = SK_LDS_DESC + SK_RUNS * 16;
static constexpr int SK_LDS_WAVE = SK_LDS_RDPTR + ((GROUP_READS + 1) * 4 + 15) / 16 * 16;
static_assert(SK_LDS_KEYS % 16 == 0 && SK_LDS_DESC % 16 == 0 && SK_LDS_WAVE % 16 == 0, "16-byte aligned LDS regions");

// the key of the m-mer w (rcw its reverse complement) at a position whose (position mod 16) << 1 is pos2
__device__ __forceinline__ uint64_t sk_pos_key(uint64_t w, uint64_t rcw, uint32_t m, uint32_t pos2)
{
    const uint64_t cw = mz::min_below_2_62(w, rcw);
    // sk_key_bits, packed by hand: the compiler builds the two halves with 64-bit shifts (8 operations for 3)
    const uint32_t B = 2u * m - 32u;
    const uint32_t a = (uint32_t)cw * 0x9E3779B1u;
    const uint32_t h = (uint32_t)(cw >> 32) ^ (a >> (32u - B));
    const uint32_t b = (a + __umul24(h, 0x85EBCBu)) * 0xC2B2AE35u;
    uint32_t low;          // pos2 | (w <= rcw): the compare's carry added in
    asm("v_cmp_le_u64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %3, vcc" : "=v"(low) : "v"(w), "v"(rcw), "v"(pos2) : "vcc");
    const uint32_t khi = __builtin_amdgcn_alignbit(0x3FFu, b, 12);                 // 0x3FF00000 | b >> 12
    const uint32_t klo = (b << 20) | ((h << (20u - B)) | low);
    return ((uint64_t)khi << 32) | klo;
}

// SHARD: mz::MZ_ALL (the whole table) or mz::MZ_LINES (the k-mers of 1/G of the minimizers); KC: k compiled in (31, 27) or 0
template <int SHARD, int KC>
__global__ __launch_bounds__(BLOCK_THREADS, MC_SK_MIN_WAVES)
void sk_query_kernel(const SkArgs A)
{
    using namespace mz;
    static_assert(SHARD == MZ_ALL || SHARD == MZ_LINES, "bucket-range shards are served by the other indexes");
    static_assert(MZ_NS == 2, "two consecutive positions per lane");
    const QueryArgs &a = A.q;
    __shared__ __attribute__((aligned(16))) uint8_t s_mem[WAVES_PER_BLOCK][SK_LDS_WAVE];

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint16_t *slice = reinterpret_cast<uint16_t *>(s_mem[wave]);
    uint64_t *keyv = reinterpret_cast<uint64_t *>(s_mem[wave] + SK_LDS_KEYS);
    uint32_t *endv = reinterpret_cast<uint32_t *>(s_mem[wave] + SK_LDS_ENDS);
    u32x4 *desc = reinterpret_cast<u32x4 *>(s_mem[wave] + SK_LDS_DESC);
    uint32_t *rdptr = reinterpret_cast<uint32_t *>(s_mem[wave] + SK_LDS_RDPTR);

    const uint32_t k = KC ? (uint32_t)KC : a.k, m = KC ? mmer_len((uint32_t)KC) : A.m;
    constexpr uint32_t W = SK_W;
    const uint64_t kmask = (1ull << (2u * k)) - 1ull;          // k <= 31
    const uint64_t mmask = (1ull << (2u * m)) - 1ull;
    const uint32_t n_reads = a.n_dev ? a.n_dev[0] : (uint32_t)a.n_reads, n_con = a.n_dev ? a.n_dev[1] : (uint32_t)a.n_containers;
    const uint32_t n_groups = (n_reads + (GROUP_READS - 1)) / GROUP_READS;
    const uint32_t gstride = gridDim.x * WAVES_PER_BLOCK;
    auto flags_now = [&]() -> uint32_t { uint32_t f = a.flags; asm volatile("" : "+s"(f)); return f; };
    auto karg = [&](size_t off, auto type_c) {
        typedef decltype(type_c) T;
        const __attribute__((address_space(4))) char *p = (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(off));
        return *reinterpret_cast<const __attribute__((address_space(4))) T *>(p + off);
    };
    auto arg_maxhits = [&]() { return karg(offsetof(SkArgs, q) + offsetof(QueryArgs, maxhits), uint32_t()); };
    auto arg_sparse_rows = [&]() { return karg(offsetof(SkArgs, q) + offsetof(QueryArgs, sparse_rows), (uint16_t *)nullptr); };
    auto arg_final_rows = [&]() { return karg(offsetof(SkArgs, q) + offsetof(QueryArgs, final_rows), (uint16_t *)nullptr); };
    auto arg_over_maxhits = [&]() { return karg(offsetof(SkArgs, q) + offsetof(QueryArgs, over_maxhits), (unsigned long long *)nullptr); };
    auto arg_extra = [&]() { return karg(offsetof(SkArgs, extra), (const uint8_t *)nullptr); };
    auto arg_reads_ptr = [&]() { return karg(offsetof(SkArgs, q) + offsetof(QueryArgs, reads_ptr), (const uint32_t *)nullptr); };
    auto arg_containers = [&]() { return karg(offsetof(SkArgs, q) + offsetof(QueryArgs, containers), (const uint16_t *)nullptr); };

    for (uint32_t g = blockIdx.x * WAVES_PER_BLOCK + wave; g < n_groups; g += gstride) {
        const uint32_t r0 = g * GROUP_READS;
        const uint32_t nr = (n_reads - r0) < (uint32_t)GROUP_READS ? (n_reads - r0) : (uint32_t)GROUP_READS;
        uint32_t ptr_v = 0;
        {
            const uint32_t lg = opaque(lane);
            if (lg <= nr) { ptr_v = arg_reads_ptr()[r0 + lg]; rdptr[lg] = ptr_v; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t rs = 0; rs < nr;) {
        const uint32_t c0 = lane_bcast(ptr_v, rs);
        const uint32_t c0a = c0 & ~7u;
        const uint64_t fits = __ballot(lane > rs && lane <= nr && (ptr_v - c0a) <= (uint32_t)MZ_STAGE_CON);
        const bool staged = a.stage_ok && fits != 0;
        const uint32_t re = staged ? (uint32_t)(63 - __builtin_clzll((unsigned long long)fits)) : rs + 1u;
        const uint32_t c1 = lane_bcast(ptr_v, re);
        if (staged) {
            const uint16_t *containers = arg_containers();
            for (uint32_t j = opaque(lane) * 8u; c0a + j < c1; j += 64u * 8u) {
                const uint64_t gi = (uint64_t)c0a + j;
                if (gi + 8u <= (uint64_t)n_con) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(containers + gi);
                    *reinterpret_cast<uint4 *>(slice + j) =
                        make_uint4(__builtin_rotateright32(v.y, 16), __builtin_rotateright32(v.x, 16),
                                   __builtin_rotateright32(v.w, 16), __builtin_rotateright32(v.z, 16));
                } else {
                    for (uint32_t t = 0; t < 8u; t++)
                        slice[(j + t) ^ 3u] = (gi + t < (uint64_t)n_con) ? containers[gi + t] : (uint16_t)0;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        auto run_group = [&](auto staged_c) {
        constexpr bool STAGED = decltype(staged_c)::value;
        auto con = [&](uint32_t i) -> uint32_t {
            if constexpr (STAGED) {
                const uint32_t li = i - c0a;
                return slice[(li < (uint32_t)(MZ_STAGE_CON + 12) ? li : (uint32_t)(MZ_STAGE_CON + 12)) ^ 3u];
            } else {
                const uint32_t ii = i < n_con ? i : n_con - 1u;
                return arg_containers()[ii];
            }
        };
        auto bases_at = [&](uint32_t first, uint32_t p, uint32_t len, uint64_t mask) -> uint64_t {
            if constexpr (STAGED) {
                uint32_t j0 = first - c0a + (p >> 3);
                if (j0 > (uint32_t)(MZ_STAGE_CON + 4)) j0 = (uint32_t)(MZ_STAGE_CON + 4);
                const uint64_t *w = reinterpret_cast<const uint64_t *>(slice) + (j0 >> 2);
                const uint64_t wa = w[0], wb = w[1];
                const uint32_t b = 16u * (j0 & 3u) + 2u * (p & 7u);
                const uint64_t top = (wa << b) | ((wb >> 1) >> (63u - b));
                return top >> (64u - 2u * len);
            }
            p = opaque(p);
            const uint32_t j0 = first + (p >> 3);
            const uint64_t hi = ((uint64_t)con(j0) << 48) | ((uint64_t)con(j0 + 1) << 32)
                              | ((uint64_t)con(j0 + 2) << 16) | (uint64_t)con(j0 + 3);
            const uint32_t lo = con(j0 + 4);
            const uint32_t sh = 80u - 2u * (p & 7u) - 2u * len;
            const uint64_t x = sh >= 16u ? (hi >> (sh - 16u)) : ((hi << (16u - sh)) | (uint64_t)(lo >> sh));
            return x & mask;
        };

        for (uint32_t ri = rs; ri < re; ri++) {
            const uint32_t beg = (uint32_t)__builtin_amdgcn_readfirstlane((int)rdptr[ri]);
            uint32_t end = (uint32_t)__builtin_amdgcn_readfirstlane((int)rdptr[ri + 1u]);
            if (end > n_con) end = n_con;
            uint32_t acc_t = 0xFFFFFFFFu, acc_c = 0, n_acc = 0;

            // fold what the lanes found -- `cnt` k-mers of target `lab` each -- into the read's accumulator: lane j holds the
            // j-th distinct target.  Counts are at most 9: four lane masks, one per bit, and popcounts on the scalar side.
            auto fold = [&](uint32_t lab, uint32_t cnt, auto wide_c) {
                constexpr bool WIDE = decltype(wide_c)::value;            // counts up to 63 (the rounds of a batch added up), else up to 15
                uint64_t many = mask_ne(cnt, 0u);
                if (many == 0) return;
                const uint64_t b1 = mask_ne(cnt & 1u, 0u), b2 = mask_ne(cnt & 2u, 0u), b4 = mask_ne(cnt & 4u, 0u), b8 = mask_ne(cnt & 8u, 0u);
                uint64_t b16 = 0, b32 = 0;
                if constexpr (WIDE) { b16 = mask_ne(cnt & 16u, 0u); b32 = mask_ne(cnt & 32u, 0u); }
                while (many) {
                    const uint32_t t = lane_bcast(lab, (uint32_t)(__ffsll((unsigned long long)many) - 1));
                    const uint64_t same = mask_eq_s(lab, t) & many;
                    many &= ~same;
                    uint32_t c = (uint32_t)__popcll(same & b1) + 2u * (uint32_t)__popcll(same & b2)
                               + 4u * (uint32_t)__popcll(same & b4) + 8u * (uint32_t)__popcll(same & b8);
                    if constexpr (WIDE) c += 16u * (uint32_t)__popcll(same & b16) + 32u * (uint32_t)__popcll(same & b32);
                    const uint64_t ex = __ballot(acc_t == t);
                    if (ex) {
                        if (acc_t == t) acc_c += c;
                    } else if (n_acc < 64u) {
                        if (lane == n_acc) { acc_t = t; acc_c = c; }
                        n_acc++;
                    } else {
                        const uint32_t mx = wave_max_u32(acc_t);
                        if (t < mx) {
                            const uint64_t who = __ballot(acc_t == mx);
                            if (lane == (uint32_t)(__ffsll((unsigned long long)who) - 1)) { acc_t = t; acc_c = c; }
                        }
                    }
                }
            };

            uint32_t pp = beg;
            while (pp < end) {
                const uint32_t plen = (uint32_t)__builtin_amdgcn_readfirstlane((int)con(pp));
                const uint32_t first = pp + 1;
                pp = first + (plen ? (plen - 1u) / 8u + 1u : 0u);
                if (plen < k) continue;
                const uint32_t nk = plen - k + 1u;

                for (uint32_t base = 0; base < nk; base += 64u * MZ_NS) {
                    SK_MARK(9);
                    // (1) lane l: the k-mers at positions base + 2l and base + 2l + 1, the keys of the m-mers that start there
                    //     (hash | position mod 16 | strand), the ends of the k-mers (first 8 bases | last 8 bases)
                    const bool last_step = base + 64u * MZ_NS >= nk;
                    const uint32_t nm = nk + (W - 1u);
                    const bool tail_in_step = nm - base <= 64u * MZ_NS;
                    const uint32_t p0 = base + 2u * lane;
                    const uint64_t in0 = mask_lt_s(p0, nk), in1 = mask_lt_s(p0, nk - 1u);
                    uint64_t x0 = 0, x1 = 0, rc0 = 0, rc1 = 0;
                    uint64_t key0 = MZ_KEY_NONE, key1 = MZ_KEY_NONE;
                    const uint32_t pos2 = (opaque(lane) & 7u) << 2;                 // ((base + 2 lane) mod 16) << 1: base is a multiple of 128
                    if (STAGED || p0 < (tail_in_step ? nm : nk)) {
                        if constexpr (STAGED) {
                            const uint32_t j0 = first - c0a + (p0 >> 3);
                            const uint64_t *w = reinterpret_cast<const uint64_t *>(slice) + (j0 >> 2);
                            const uint64_t wa = w[0], wb = w[1];
                            const uint32_t b = 16u * (j0 & 3u) + 2u * (p0 & 7u);
                            const uint64_t top = (wa << b) | ((wb >> 1) >> (63u - b));
                            x0 = top >> (64u - 2u * k);
                            x1 = (top >> (62u - 2u * k)) & kmask;
                        } else {
                            x0 = bases_at(first, p0, k, kmask);
                            x1 = bases_at(first, p0 + 1u, k, kmask);
                        }
                        rc0 = revcomp(x0, k);
                        if (k >= 17u) {
                            const uint64_t r2 = rc0 >> 2;
                            rc1 = ((uint64_t)((uint32_t)(r2 >> 32) | ((~(uint32_t)x1 & 3u) << (2u * k - 34u))) << 32) | (uint32_t)r2;
                        } else {
                            rc1 = (rc0 >> 2) | ((uint64_t)(3u - ((uint32_t)x1 & 3u)) << (2u * k - 2u));
                        }
                        key0 = sk_pos_key(x0 >> (2u * (k - m)), rc0 & mmask, m, pos2);
                        key1 = sk_pos_key(x1 >> (2u * (k - m)), rc1 & mmask, m, pos2 | 2u);
                    }
                    {
                        typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
                        *reinterpret_cast<u64x2 *>(keyv + 2u * lane) = u64x2{key0, key1};
                        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                        *reinterpret_cast<u32x2 *>(endv + 2u * lane) =
                            u32x2{((uint32_t)(x0 >> (2u * k - 16u)) << 16) | ((uint32_t)x0 & 0xFFFFu),
                                  ((uint32_t)(x1 >> (2u * k - 16u)) << 16) | ((uint32_t)x1 & 0xFFFFu)};
                    }
                    if (!tail_in_step) {
                        // the W-1 m-mers behind the part's last k-mer, or -- not the last step -- behind this step's last position
                        const uint32_t q = nk - 1u - base;
                        uint64_t x_last = 0, rc_last = 0;
                        if (last_step) {
                            x_last = lane_bcast64((q & 1u) ? x1 : x0, q >> 1);
                            rc_last = lane_bcast64((q & 1u) ? rc1 : rc0, q >> 1);
                        }
                        if (lane < W - 1u) {
                            const uint32_t ln = opaque(lane);
                            if (last_step) {
                                const uint32_t i = ln + 1u;
                                keyv[64 * MZ_NS + ln] = (uint64_t)opaque((uint32_t)(MZ_KEY_NONE >> 32)) << 32;
                                keyv[q + i] = sk_pos_key((x_last >> (2u * (k - m - i))) & mmask, (rc_last >> (2u * i)) & mmask, m, ((q + i) & 15u) << 1);
                            } else {
                                const uint64_t w = bases_at(first, base + 64u * MZ_NS + ln, m, mmask);
                                keyv[64 * MZ_NS + ln] = sk_pos_key(w, revcomp(w, m), m, (ln & 15u) << 1);
                            }
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    SK_MARK(10);
                    // (2) window minimum = the k-mer's key, (3) runs of equal keys, numbered
                    uint32_t n_runs;
                    uint64_t K[MZ_NS];
                    uint32_t run[MZ_NS];
                    uint64_t lead[MZ_NS], tail[MZ_NS];
                    {
                        uint64_t v[SK_W + 1];
#pragma unroll
                        for (int i = 0; i < SK_W + 1; i++) v[i] = keyv[2u * lane + i];
                        uint64_t mid = v[1];
#pragma unroll
                        for (int i = 2; i < SK_W; i++) mid = key_min(mid, v[i]);
                        K[0] = key_min(v[0], mid); K[1] = key_min(mid, v[SK_W]);
                        uint64_t own0 = in0, own1 = in1;
                        if constexpr (SHARD == MZ_LINES) {
                            own0 &= mask_eq_s(sk_part_of(K[0], A.n_parts), A.part);
                            own1 &= mask_eq_s(sk_part_of(K[1], A.n_parts), A.part);
                        }
                        // the low words differ whenever the minimizer is another occurrence (its position is in them)
                        const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)(uint32_t)K[1], 0x138, 0xf, 0xf, false);
                        uint64_t b0 = mask_ne((uint32_t)K[0], prev), b1 = mask_ne((uint32_t)K[1], (uint32_t)K[0]);
                        if constexpr (SHARD == MZ_LINES) { b0 |= ~(own1 << 1); b1 |= ~own0; }
                        b0 &= own0; b1 &= own1;
                        lead[0] = b0; lead[1] = b1;
                        // a position ends its run when the next one starts one or holds no k-mer of ours
                        tail[0] = own0 & (b1 | ~own1);
                        tail[1] = own1 & ((b0 >> 1) | ~(own0 >> 1));
                        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b0, 0u))
                                             + __builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0u));
                        run[0] = below + (__builtin_amdgcn_inverse_ballot_w64(b0) ? 1u : 0u) - 1u;
                        run[1] = run[0] + (__builtin_amdgcn_inverse_ballot_w64(b1) ? 1u : 0u);
                        n_runs = (uint32_t)__popcll(b0) + (uint32_t)__popcll(b1);
                    }

                    for (uint32_t rb = 0; rb < n_runs; rb += SK_RUNS) {
                        const uint32_t nb = n_runs - rb < (uint32_t)SK_RUNS ? n_runs - rb : (uint32_t)SK_RUNS;
                        SK_MARK(11);
                        // (4) the runs of this batch publish key and first / last position ...
#pragma unroll
                        for (int s = 0; s < MZ_NS; s++) {
                            const uint32_t at = (run[s] - rb) & (uint32_t)(SK_RUNS - 1);
                            const uint64_t here = mask_lt_s(run[s] - rb, (uint32_t)SK_RUNS);
                            uint32_t *d = reinterpret_cast<uint32_t *>(desc + at);
                            if (__builtin_amdgcn_inverse_ballot_w64(lead[s] & here)) {
                                typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
                                *reinterpret_cast<u32x3 *>(d) = u32x3{(uint32_t)K[s], (uint32_t)(K[s] >> 32), 2u * lane + (uint32_t)s};
                            }
                            if (__builtin_amdgcn_inverse_ballot_w64(tail[s] & here)) d[3] = 2u * lane + (uint32_t)s;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        SK_MARK(13);
                        // (5) FOUR lanes fetch the line of a run, two slots each (lane q of the four: slots q and q + 4 -- the build
                        //     puts the records of one minimizer into neighbouring slots, so they sit in different lanes); 16 runs a
                        //     round, every round in flight before any is used.  The line comes straight from the run's key: the loads
                        //     are on their way while (6) works out what the records will be asked.
                        constexpr int ROUNDS = SK_RUNS / 16;
                        u32x4 va[ROUNDS], vb[ROUNDS];
                        const uint32_t lf = opaque(lane);
                        const uint32_t last = nb - 1u;
                        const uint64_t lane_base = (uint64_t)(uintptr_t)A.lines + (lf & 3u) * 16u;
#pragma unroll
                        for (int rd = 0; rd < ROUNDS; rd++) {
                            if (16u * rd < nb) {
                                const uint32_t j = 16u * rd + (lf >> 2);
                                const uint64_t Kj = *reinterpret_cast<const uint64_t *>(desc + (j < last ? j : last));      // (lane groups past the last run: its line again)
#ifdef MC_SK_DEBUG_WINDOW      // measurement builds only (WRONG results): every fetch inside a cache-resident window of lines
                                const uint32_t ln = sk_line_of(Kj, A.n_lines) & (uint32_t)(MC_SK_DEBUG_WINDOW - 1);
#else
                                const uint32_t ln = sk_line_of(Kj, A.n_lines);
#endif
                                uint64_t addr;              // lane_base + ln * 128 in one operation
                                asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(addr) : "v"(ln), "s"(128u), "v"(lane_base) : "vcc");
                                va[rd] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>((uintptr_t)addr));
                                vb[rd] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>((uintptr_t)addr + 64u));
                            }
                        }
                        SK_MARK(12);
                        // (6) lane r < nb turns run r into what the records are asked: key hash, offsets, flanks
                        {
                            const uint32_t r = opaque(lane) & (uint32_t)(SK_RUNS - 1);
                            const u32x4 raw = desc[r];
                            const uint32_t ea = endv[raw[2] & 127u], eb = endv[raw[3] & 127u];
                            const uint64_t Kr = ((uint64_t)raw[1] << 32) | raw[0];
                            const SkRun R = sk_run(Kr, base + raw[2], base + raw[3], ea, eb);
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            // (entries behind the batch's last run: an empty range of offsets -- the lane groups past the last run
                            // match nothing, without a test)
                            const bool real = __builtin_amdgcn_inverse_ballot_w64(mask_lt_s(r, nb));
                            if (lane < (uint32_t)SK_RUNS)
                                desc[r] = u32x4{R.kd0, real ? (R.kd1 | (R.o_lo << 20) | (R.o_hi << 24)) : (15u << 20), R.lr, 0u};
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        SK_MARK(14);
                        // (7) a lane looks for the run's minimizer in its two slots and works the flanks out for the one that has it
                        //     (both: a second turn, rare); what a lane finds over the rounds is added up as long as it is one target's
                        //     and folded into the read's accumulator once per batch
                        uint32_t lab_a = 0u, cnt_a = 0u, hdr_any = 0u;
                        auto gather = [&](uint32_t lab, uint32_t c) {
                            if ((mask_ne(c, 0u) & mask_ne(cnt_a, 0u) & mask_ne(lab, lab_a)) != 0) { fold(lab_a, cnt_a, std::true_type{}); cnt_a = 0u; }
                            lab_a = c ? lab : lab_a;
                            cnt_a += c;
                        };
#pragma unroll
                        for (int rd = 0; rd < ROUNDS; rd++) {
                            if (16u * rd < nb) {
                                const uint32_t j = 16u * rd + (lf >> 2);
                                const u32x4 dq = desc[j];
                                const uint32_t kd1 = dq[1] & 0xFFFFFu, o_lo = (dq[1] >> 20) & 15u, o_hi = (dq[1] >> 24) & 15u;
                                const uint64_t ma = mask_eq_s((va[rd][0] ^ dq[0]) | ((va[rd][1] ^ kd1) & 0xFFFFFu), 0u);
                                const uint64_t mb = mask_eq_s((vb[rd][0] ^ dq[0]) | ((vb[rd][1] ^ kd1) & 0xFFFFFu) | (vb[rd][3] & SK_HDR), 0u);
                                if ((ma | mb) != 0) {
                                    const bool in_a = __builtin_amdgcn_inverse_ballot_w64(ma);
                                    const uint32_t e1 = in_a ? va[rd][1] : vb[rd][1], e2 = in_a ? va[rd][2] : vb[rd][2], e3 = in_a ? va[rd][3] : vb[rd][3];
                                    uint32_t c = sk_match_flanks(e1, e2, o_lo, o_hi, dq[2]);
                                    c = __builtin_amdgcn_inverse_ballot_w64(ma | mb) ? c : 0u;
                                    gather(e3 & 0xFFFFu, c);
                                    if ((ma & mb) != 0) {
                                        uint32_t c2 = sk_match_flanks(vb[rd][1], vb[rd][2], o_lo, o_hi, dq[2]);
                                        c2 = __builtin_amdgcn_inverse_ballot_w64(ma & mb) ? c2 : 0u;
                                        gather(vb[rd][3] & 0xFFFFu, c2);
                                    }
                                }
                                hdr_any |= vb[rd][3];
                            }
                        }
                        SK_MARK(15);
                        fold(lab_a, cnt_a, std::true_type{});
                        SK_MARK(8);
                        if ((mask_ne(hdr_any & SK_HDR, 0u) & 0x8888888888888888ull) != 0) {
                            // Rare: some line of the batch has extra lines (its slot 7 -- the second slot of a group's lane 3 -- is a header)
#pragma unroll
                            for (int rd = 0; rd < ROUNDS; rd++) {
                                if (16u * rd >= nb) continue;
                                const uint32_t j = 16u * rd + (lf >> 2);
                                const u32x4 dq = desc[j < last ? j : last];
                                const uint64_t valid = mask_lt_s(j, nb);
                                if ((mask_ne(vb[rd][3] & SK_HDR, 0u) & valid & 0x8888888888888888ull) == 0) continue;
                                // the header goes to the 4 lanes of its group
                                const int src = (int)((lf | 3u) << 2);
                                const uint32_t h0 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)vb[rd][0]), h1 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)vb[rd][1]);
                                const uint32_t h2 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)vb[rd][2]), h3 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)vb[rd][3]);
                                const bool has = (h3 & SK_HDR) != 0u && __builtin_amdgcn_inverse_ballot_w64(valid);
                                const uint32_t kd1 = dq[1] & 0xFFFFFu, o_lo = (dq[1] >> 20) & 15u, o_hi = (dq[1] >> 24) & 15u;
                                const uint8_t *xl = arg_extra() + (lf & 3u) * 16u;
                                // one line of a chain, two slots a lane, matched against offsets lo .. hi of the run
                                auto chain_line = [&](bool on, uint32_t line, uint32_t lo, uint32_t hi) {
                                    uint32_t ca = 0u, cb = 0u, la = 0u, lb = 0u;
                                    if (on) {
                                        const u32x4 xa = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(xl + ((uint64_t)line << 7)));
                                        const u32x4 xb = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(xl + ((uint64_t)line << 7) + 64u));
                                        ca = sk_match(SkSlot{xa[0], xa[1], xa[2], xa[3] & 0xFFFFu}, dq[0], kd1, lo, hi, dq[2]);
                                        cb = sk_match(SkSlot{xb[0], xb[1], xb[2], xb[3] & 0xFFFFu}, dq[0], kd1, lo, hi, dq[2]);
                                        la = xa[3] & 0xFFFFu; lb = xb[3] & 0xFFFFu;
                                    }
                                    fold(la, ca, std::false_type{});
                                    fold(lb, cb, std::false_type{});
                                };
                                // a linear chain: every line of it, when the Bloom word knows the run's minimizer
                                const uint32_t n_lin = has && sk_bloom_pass(h0, h1, dq[0]) ? (h3 & 0xFFu) : 0u;
                                for (uint32_t i = 0; __ballot(i < n_lin) != 0; i++) chain_line(i < n_lin, h2 + i, o_lo, o_hi);
                                // hashed chains: k-mer by k-mer, the chain line the k-mer's own hash picks and the ones behind it
                                const uint32_t sl = has ? (h3 >> 8) & 0xFFu : 0u, probes = (h3 >> 16) & 15u;
                                for (uint32_t t = 0; __ballot(sl != 0u && o_lo + t <= o_hi) != 0; t++) {
                                    const uint32_t o = o_lo + t;
                                    const bool on = sl != 0u && o <= o_hi;
                                    const uint32_t hh = on ? sk_entry_hash(dq[0], o, sk_lr_of(dq[2], o)) >> (32u - sl) : 0u;
                                    for (uint32_t pr = 0; __ballot(on && pr < probes) != 0; pr++) chain_line(on && pr < probes, h2 + hh + pr, o, o);
                                }
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }

            // ---- finalisation (identical to mz_query_kernel) ---------------------
            const uint64_t rd = (uint64_t)(r0 + ri);
            bool valid = lane < n_acc;
            uint32_t rank = 0;
            const uint32_t maxhits = arg_maxhits(), row_len = 2u * maxhits + 2u;
            const bool need_rank = (flags_now() & 2u) || (n_acc > maxhits);
            if (need_rank) {
                for (uint32_t j = 0; j < n_acc; j++) {
                    const uint32_t tj = lane_bcast(acc_t, j);
                    rank += (tj < acc_t) ? 1u : 0u;
                }
                if (n_acc > maxhits) {
                    valid = valid && rank < maxhits;
                    if (__builtin_amdgcn_inverse_ballot_w64(1ull)) atomicAdd(arg_over_maxhits(), 1ull);
                }
            }
            const uint32_t n_keep = n_acc > maxhits ? maxhits : n_acc;
            if (flags_now() & 2u) {
                uint16_t *row = arg_sparse_rows() + rd * row_len;
                if (__builtin_amdgcn_inverse_ballot_w64(1ull)) row[0] = (uint16_t)n_keep;
                if (valid) { row[1 + 2 * rank] = (uint16_t)acc_t; row[2 + 2 * rank] = (uint16_t)sat_u16(acc_c); }
                for (uint32_t i = 1u + 2u * n_keep + lane; i < row_len; i += 64u) row[i] = 0;
            }
            if (flags_now() & 1u) {
                uint32_t o0, o1, o2, o3 = 0u, o4 = 0u;
                if (n_acc <= 1u) {
                    const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)acc_t, 0);
                    const uint32_t c0v = (uint32_t)__builtin_amdgcn_readlane((int)acc_c, 0);
                    const bool has = n_acc == 1u;
                    o0 = has ? sat_u16(c0v) : 0u;
                    o1 = has ? (t0 & 0xFFFFu) + 1u : 0u;
                    o2 = o0;
                } else {
                    const uint32_t cc  = sat_u16(acc_c);
                    const uint32_t key = valid ? ((cc << 16) | (0xFFFFu - (acc_t & 0xFFFFu))) : 0u;
                    const uint32_t k1  = wave_max_u32(key);
                    const uint32_t k2  = wave_max_u32(key == k1 ? 0u : key);
                    const uint32_t sum = wave_sum_u32(valid ? cc : 0u);
                    o0 = sum & 0xFFFFu;
                    o1 = k1 ? (0xFFFFu - (k1 & 0xFFFFu)) + 1u : 0u;
                    o2 = k1 >> 16;
                    o3 = k2 ? (0xFFFFu - (k2 & 0xFFFFu)) + 1u : 0u;
                    o4 = k2 >> 16;
                }
                if (__builtin_amdgcn_inverse_ballot_w64(1ull)) {
                    struct __attribute__((packed, aligned(2))) Row5 { uint32_t w0, w1; uint16_t h; };
                    *reinterpret_cast<Row5 *>(arg_final_rows() + rd * 5u) =
                        Row5{(o0 & 0xFFFFu) | (o1 << 16), (o2 & 0xFFFFu) | (o3 << 16), (uint16_t)o4};
                }
            }
        }
        };   // run_group
        if (staged) run_group(std::true_type{}); else run_group(std::false_type{});
        rs = re;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        }   // pieces of the group
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}
} // namespace sk
} // namespace mc
