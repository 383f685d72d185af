// mc_minimizer.hpp -- locality-aware index: the same database, bucketed by MINIMIZER.
//
// Why.  With the bucket-line table (mc_device.hpp) every k-mer of a read is one random
// HBM request, and the kernel sits on the memory system's random-request ceiling
// (~50 G requests/s).  Consecutive k-mers of a read overlap in k-1 bases;
// k-mers that share their minimizer (the smallest, under a fixed hash, of the canonical
// m-mers inside the k-mer) can share a bucket.  A 150 bp read has 120 k-mers but only
// ~2*120/(k-m+2) distinct minimizers, so it needs ~21 line fetches instead of 120.
//
// What stays the same.  The on-disk database, the batch format, and the answer: a k-mer
// hits iff its canonical value is a stored k-mer, with that k-mer's label.  Only the
// in-HBM arrangement differs, built once at load from the same arrays:
//
//   K(c)    = min over the k-m+1 = MZ_MAXW (9) windows w of canonical k-mer c of key(min(w, rc(w)))
//             (orientation-free: x and rc(x) have the same set of canonical m-mers)
//   line(c) = mulhi(mix32(K(c)), lines per part)      part(c) = mulhi(mix32'(K(c)), n_parts)
//             (two hashes of the 52-bit key: a table spread over G contexts addresses G x 2^32 lines)
//   a line  = 128 bytes: 12 keys (full canonical k-mers, u64, ascending, unused = all ones last),
//             12 labels (u16), dword30 = 0, or for a line that overflows: lines per chain (bits 0-1) | "also the
//             chain behind" (bit 2) | log2 of the number of chains (bits 3-7) | bit 16; dword31 = first extra line.
//             A line that overflows keeps 11 keys: slot 11 is a 62-bit Bloom word over the k-mers that are not in
//             the first line
//   extra lines (same shape, contiguous per line) hold what does not fit: a chain of at most MZ_EMAX
//   lines.  A CROWDED line (one minimizer shared by thousands of k-mers: a conserved m-mer in many
//   genomes) gets 2^s such chains ("segments") and a k-mer's segment is picked by a hash of the k-mer
//   itself, so that one crowded minimizer costs a lookup at most 2 * MZ_EMAX extra line reads however
//   many k-mers share it, and never costs the rest of the table its index.
//
// Lookup is exact: the stored key is the whole canonical k-mer.
//
// The query kernel (mz_query_kernel below) is bound by instruction issue and by its register
// and LDS budget, not by DRAM: see the notes at opaque(), at the launch bounds and DESIGN.md 3-4.
#pragma once

#include "mc_device.hpp"

namespace mc {
namespace mz {

static constexpr int MZ_LINE = 128;
static constexpr int MZ_CAP = 12;
static constexpr int MZ_EMAX = 3;                                        // extra lines chained to a primary line
// A line that overflows keeps MZ_CAP1 k-mers in its first line: slot 11 holds a 62-bit Bloom word over the k-mers that
// live in its extra lines (round 2: 16 bits in the header -- with 5-8 k-mers behind a line a quarter of all misses
// went on to the chain for nothing, and so did 6 reads in 10 that hit nothing at all, tools/stats_build.sh)
static constexpr int MZ_CAP1 = MZ_CAP - 1;
static constexpr uint32_t MZ_CHAIN_CAP = (uint32_t)MZ_CAP1 + (uint32_t)MZ_CAP * MZ_EMAX;   // k-mers a line keeps (first + extra lines)
// dword 30 of a primary line: bits 0-1 = lines per chain (0..MZ_EMAX), bit 2 = a lookup also scans the chain behind
// its own (some chain was full when the table was built), bits 3-7 = s: the line has 2^s chains and a k-mer's chain
// is picked by a hash of the k-mer (0: one chain), bit 16 = the line has extra lines (the header of a primary line that
// does not overflow is 0; EXTRA lines carry no header -- they are memset to 0xFF and their dwords 30-31 are never read)
static constexpr uint32_t MZ_HDR_LEN = 3u;
static constexpr uint32_t MZ_HDR_CHAIN = 0x10000u;
static constexpr uint32_t MZ_HDR_TWO = 4u;
static constexpr uint32_t MZ_HDR_SEG_SHIFT = 3u, MZ_HDR_SEG_MASK = 31u;
static constexpr uint32_t MZ_SEG_LOAD = 9u;          // k-mers aimed at per segment of MZ_EMAX * MZ_CAP = 36 slots (P(overflow) ~ 1e-12)
static constexpr uint64_t MZ_EMPTY = ~0ull;
#ifndef MC_MZ_MAXW
#define MC_MZ_MAXW 9
#endif
// windows per k-mer: w = k - m + 1, odd (the window is read in 16-byte pairs).  Measured on the headline
// workload (k = 31) in round 1: w = 7 / 9 / 11 / 13 / 15 / 17 -> 1141 / 1229 / 1231 / 1167 / 748 / 101 Mreads/s: fewer
// windows mean more lines per read (density 2/(w+1)) but smaller minimizer groups and a larger minimizer
// space (4^m): past w = 13 unrelated minimizers crowd the lines.  Round 2: with w = 9 the 128 positions of a
// step hold the 120 k-mers of a 150 bp read AND the 8 m-mers behind the last one, so the tail code of the
// kernel never runs for reads up to 150 bp: 9 / 11 windows -> 1450 / 1360 Mreads/s (genome-shaped table 1165 / 1117).
static constexpr int MZ_MAXW = MC_MZ_MAXW;
static_assert(MZ_MAXW % 2 == 1 && MZ_MAXW >= 3 && MZ_MAXW <= 17, "window count");
#ifndef MC_MZ_NS
#define MC_MZ_NS 2
#endif
static constexpr int MZ_NS = MC_MZ_NS;        // k-mer positions per lane and step (a 150 bp read = one step of 128)
#ifndef MC_MZ_RUNS
#define MC_MZ_RUNS (16 * MC_MZ_NS)
#endif
static constexpr int MZ_RUNS = MC_MZ_RUNS;    // runs (distinct lines) fetched per batch (multiple of 8)
static constexpr int MZ_LSTRIDE = MZ_LINE + 16; // LDS stride of a staged line: keeps equal offsets of different
                                                // runs on different banks (a 128-byte stride is a 9-way conflict)

// minimizer length for a k-mer length: always w = k - m + 1 = MZ_MAXW windows (k >= MZ_MAXW)
__host__ __device__ __forceinline__ uint32_t mmer_len(uint32_t k) { return k > (uint32_t)MZ_MAXW - 1u ? k - ((uint32_t)MZ_MAXW - 1u) : 1u; }

// Ordering key of a canonical m-mer (any 64-bit value works, the high word is just stirred in): t = low word of lo*C1, stirred with the
// high byte -- a bijection of lo for every hi, its top bits a multiplicative hash of all of lo -- then
// 20 more bits of the product.  The 52 bits are the mantissa of a double in [1, 2): for such doubles
// numeric order = integer order of the bit pattern, so a window minimum is ONE v_min_f64 per element
// instead of a 64-bit compare and two selects.  6 VALU ops (v_mad_u64_u32, v_mul, v_xor, two
// v_alignbit).  The minimizer KEY of a k-mer is the smallest key over its windows -- the same for x
// and rc(x), which contain the same canonical m-mers; two m-mers that share a key merely share lines.
// (32-bit keys are too few: the minima of 6.4e9 k-mers crowd into a few 1e8 values and pile unrelated
// minimizers onto a line.)
static constexpr uint64_t MZ_KEY_NONE = 0x4000000000000000ull;    // 2.0: above every key

__device__ __forceinline__ uint64_t key_of_canonical(uint64_t cw)
{
    const uint32_t lo = (uint32_t)cw, hi = (uint32_t)(cw >> 32);
    const uint64_t p = (uint64_t)lo * 0x9E3779B1u;
    const uint32_t t = (uint32_t)p ^ (hi * 0x85EBCA6Bu);
    const uint32_t khi = __builtin_amdgcn_alignbit(0x3FFu, t, 12);                    // 0x3FF00000 | t >> 12
    const uint32_t klo = __builtin_amdgcn_alignbit(t, (uint32_t)(p >> 32), 12);       // t[11:0] : p[63:44]
    return ((uint64_t)khi << 32) | klo;
}

__device__ __forceinline__ uint64_t mmer_key(uint64_t w, uint32_t m)
{
    const uint64_t rc = revcomp(w, m);
    return key_of_canonical(w < rc ? w : rc);
}

__device__ __forceinline__ uint64_t key_min(uint64_t a, uint64_t b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(__builtin_bit_cast(double, a)), "v"(__builtin_bit_cast(double, b)));
    return __builtin_bit_cast(uint64_t, r);
}

// Smaller of two values below 2^62 in ONE instruction: such bit patterns are non-negative doubles that are never
// NaN or infinite (exponent field at most 0x3FF), and for those numeric order = integer order (f64 denormals are
// kept in HIP's default float mode).  A 64-bit compare and two selects otherwise.
#ifndef MC_MZ_F64_MIN
#define MC_MZ_F64_MIN 1
#endif
__device__ __forceinline__ uint64_t min_below_2_62(uint64_t a, uint64_t b)
{
#if MC_MZ_F64_MIN
    return key_min(a, b);
#else
    return a < b ? a : b;
#endif
}

// the same from the m-mer and its reverse complement when both are at hand (m-mers are below 2^48)
__device__ __forceinline__ uint64_t mmer_key2(uint64_t w, uint64_t rcw)
{
    return key_of_canonical(min_below_2_62(w, rcw));
}

// A copy of a per-lane value the compiler cannot trace back: what is derived from it is computed
// where it is used instead of being hoisted out of every loop into long-lived registers
// (this kernel is bound by its register budget: 5 vs 6 waves per SIMD).
__device__ __forceinline__ uint32_t opaque(uint32_t v)
{
    asm volatile("" : "+v"(v));
    return v;
}

// Lane masks straight from one compare.  A ballot of a condition that is also used per lane (or of
// anything but a single compare) is compiled to a select of 0/1 and a second compare; these are one
// v_cmp into a scalar pair, combined with scalar ANDs, and turned back into a per-lane condition for
// free with __builtin_amdgcn_inverse_ballot_w64.  (Lanes that are switched off read as 0.)
__device__ __forceinline__ uint64_t mask_ne(uint32_t a, uint32_t b)
{
    uint64_t m;
    asm("v_cmp_ne_u32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
    return m;
}
__device__ __forceinline__ uint64_t mask_lt_s(uint32_t a, uint32_t uniform_b)       // a < b, b wave-uniform
{
    uint64_t m;
    asm("v_cmp_gt_u32_e64 %0, %1, %2" : "=s"(m) : "s"(uniform_b), "v"(a));
    return m;
}
__device__ __forceinline__ uint64_t mask_gt_s(uint32_t a, uint32_t uniform_b)       // a > b, b wave-uniform
{
    uint64_t m;
    asm("v_cmp_lt_u32_e64 %0, %1, %2" : "=s"(m) : "s"(uniform_b), "v"(a));
    return m;
}
__device__ __forceinline__ uint64_t mask_eq_s(uint32_t a, uint32_t uniform_b)
{
    uint64_t m;
    asm("v_cmp_eq_u32_e64 %0, %1, %2" : "=s"(m) : "s"(uniform_b), "v"(a));
    return m;
}

__device__ __forceinline__ uint64_t lane_bcast64(uint64_t v, uint32_t uniform_lane)
{
    return (uint64_t)lane_bcast((uint32_t)v, uniform_lane) | ((uint64_t)lane_bcast((uint32_t)(v >> 32), uniform_lane) << 32);
}

// K(c) for a stored canonical k-mer (index build)
__device__ __forceinline__ uint64_t kmer_min_key(uint64_t c, uint32_t k, uint32_t m)
{
    const uint64_t mmask = (1ull << (2 * m)) - 1ull;
    uint64_t best = MZ_KEY_NONE;
    for (uint32_t i = 0; i + m <= k; i++) {
        const uint64_t key = mmer_key((c >> (2 * (k - m - i))) & mmask, m);
        best = key < best ? key : best;          // same order as v_min_f64 on these patterns
    }
    return best;
}

// The key is a hash already (mmer_key); being a window MINIMUM skews its top bits only, so the line
// is drawn from the low word, stirred with the high one: 3 VALU ops.
__device__ __forceinline__ uint32_t line_of(uint64_t K, uint32_t n_lines)
{
    const uint32_t h = ((uint32_t)(K >> 32) * 0x9E3779B1u) ^ (uint32_t)K;
    return __umulhi(h, n_lines);
}

// A table that is spread over several contexts (GPUs): WHICH of them owns the k-mers of a minimizer is drawn
// from a second hash of the key, the line inside that context's share from line_of() -- every part has the same
// number of lines.  Round 2 cut ONE 32-bit line space into ranges: n_keys / fill lines in all, so a table of
// 32e9 k-mers (8 cards x 4e9) could not be loaded at the sparse fills the cards had room for.  Two hashes of
// the 52-bit key address n_parts x 2^32 lines; the 32-bit limit is now per context (550 GB of lines: never).
// The second hash must bring bits the first does not have (the part drawn from line_of()'s own hash would leave
// 7 of 8 lines of a part without any minimizer): low word times a constant plus the high word, 2 VALU ops.
__device__ __forceinline__ uint32_t part_of(uint64_t K, uint32_t n_parts)
{
    const uint32_t h = (uint32_t)K * 0x85EBCA6Bu + (uint32_t)(K >> 32);
    return __umulhi(h, n_parts);
}

// The Bloom word of an overflowing line (key slot 11 of its first line) over the k-mers that live in its extra lines:
// two bits per k-mer, one in each half of the word (31 + 29 usable bits; an overflowing line spills 2-8 k-mers as a
// rule: 0.4-5 % false positives; the 16-bit word in the header that round 2 had: 20-40 %).  A k-mer that is not in the
// first line follows the chain only if both its bits are set, so nearly every miss ends at the first line.  The word
// must never equal a k-mer (the match compares all 12 slots): its top two bits are ones -- above every canonical k-mer
// of k < 32 -- and for k = 32 its top 32 bits are ones and only the low half carries the filter (a 32-mer that starts
// with sixteen A is canonical only if it ends with sixteen T, i.e. only with all filter bits zero, and an overflowing
// line has some set; the test below needs no switch: the upper half always passes).  Never all ones (= empty slot)
// in the first case either: bit 61 stays clear.
static constexpr uint64_t MZ_BLOOM_BASE = 0xC000000000000000ull, MZ_BLOOM_BASE32 = 0xFFFFFFFF00000000ull;
__device__ __forceinline__ uint32_t bloom_hash(uint64_t c) { return ((uint32_t)c ^ (uint32_t)(c >> 32)) * 0x9E3779B1u; }
__device__ __forceinline__ uint32_t bloom_hi_bit(uint32_t h) { const uint32_t b = (h >> 22) & 31u; return b < 28u ? b : 28u; }
__device__ __forceinline__ uint64_t bloom_bits(uint64_t c, bool k32)       // build side
{
    const uint32_t h = bloom_hash(c);
    const uint64_t lo = 1u << (h >> 27);
    return k32 ? lo : (lo | ((uint64_t)(1u << bloom_hi_bit(h)) << 32));
}
__device__ __forceinline__ bool bloom_pass(uint64_t word, uint64_t c)       // lookup side: 9 VALU operations
{
    const uint32_t h = bloom_hash(c);
    return (((uint32_t)word >> (h >> 27)) & ((uint32_t)(word >> 32) >> bloom_hi_bit(h)) & 1u) != 0u;
}
__host__ __device__ __forceinline__ uint64_t bloom_base(uint32_t k) { return k >= 32u ? MZ_BLOOM_BASE32 : MZ_BLOOM_BASE; }

// ---------------------------------------------------------------------------
// index build from the raw bucket arrays (sizes u8, quotients, labels)
// ---------------------------------------------------------------------------
// PASS 0: count k-mers per line.  PASS 1: place them.  The table arrives in bucket-order CHUNKS
// (sizes, quotients, labels of buckets [bucket0, bucket0 + n_buckets)), each chunk once per pass, so
// neither the raw arrays of the whole table nor the file have to be resident next to the lines.
// A context may own only the lines [line0, line0 + n_local) of the n_lines_total the whole table
// is spread over (a line-range shard): k-mers of other lines are skipped.

// crowded lines: which of the 2^s chains a k-mer belongs to (the top bits of a multiplicative hash; chain 0
// when s = 0): 5 VALU operations with the header field extraction
__device__ __forceinline__ uint32_t seg_of(uint64_t c, uint32_t seg_log)
{
    // both halves go through a multiplication: k-mers that share a minimizer differ in BOTH flanks, and an
    // xor of the halves lets the flanks cancel
    const uint32_t h = ((uint32_t)(c >> 32) * 0x85EBCA6Bu + (uint32_t)c) * 0x9E3779B1u;
    return __umulhi(h, 1u << seg_log);
}
// number of chains (log2) of a line with `c` k-mers: 0 unless it is crowded
__host__ __device__ __forceinline__ uint32_t seg_log_of(uint32_t c)
{
    if (c <= MZ_CHAIN_CAP) return 0u;
    const uint32_t want = (c - (uint32_t)MZ_CAP1 + MZ_SEG_LOAD - 1u) / MZ_SEG_LOAD;      // segments at the aimed load
    uint32_t s = 1u;
    while ((1u << s) < want) s++;
    return s;
}

template <int PASS, bool WIDE>
__global__ __launch_bounds__(RL_THREADS)
void mz_build_kernel(const uint8_t *sz, const typename KeyOf<WIDE>::type *keys, const uint16_t *labels,
                     uint64_t n_buckets, uint64_t n_keys, uint64_t bucket0, uint64_t htsize, const uint64_t *blk_key_off,
                     uint32_t k, uint32_t m, uint32_t part, uint32_t n_parts, uint32_t n_local,
                     uint32_t *count, uint8_t *lines, uint8_t *extra_lines, unsigned int *failed)
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
            // a chunk whose bucket sizes announce more k-mers than the caller passed (mz_scan_blocks_kernel flags it, the
            // build ends with MC_EINVAL at the next pass boundary): nothing is read past the arrays meanwhile
            if (koff + j >= n_keys) break;
            const uint64_t c = (uint64_t)keys[koff + j] * htsize + (bucket0 + b);     // the canonical k-mer
            const uint64_t K = kmer_min_key(c, k, m);
            if (n_parts > 1u && part_of(K, n_parts) != part) continue;               // another part's k-mer
            const uint32_t l = line_of(K, n_local);
            const uint32_t slot = atomicAdd(&count[l], 1u);       // PASS 1: the counters were reset; they end equal to PASS 0's
            if (PASS == 1) {
                uint8_t *first = lines + (uint64_t)l * MZ_LINE;
                uint32_t *hdr = reinterpret_cast<uint32_t *>(first) + 30;
                const uint32_t h0 = *reinterpret_cast<volatile uint32_t *>(hdr);       // low bits: written before this pass
                const uint32_t cap1 = (h0 & MZ_HDR_CHAIN) ? (uint32_t)MZ_CAP1 : (uint32_t)MZ_CAP;   // an overflowing line: slot 11 is its Bloom word
                if (slot < cap1) {
                    reinterpret_cast<uint64_t *>(first)[slot] = c;
                    reinterpret_cast<uint16_t *>(first + 8 * MZ_CAP)[slot] = labels[koff + j];
                } else {
                    if (!(h0 & MZ_HDR_CHAIN)) { atomicOr(failed, 2u); continue; }   // the first pass counted at most MZ_CAP k-mers here
                    atomicOr(reinterpret_cast<unsigned long long *>(first) + MZ_CAP1, (unsigned long long)bloom_bits(c, k >= 32u));
                    const uint32_t seg_log = (h0 >> MZ_HDR_SEG_SHIFT) & MZ_HDR_SEG_MASK;
                    uint8_t *chain0 = extra_lines + (uint64_t)hdr[1] * MZ_LINE;
                    if (seg_log == 0u) {                               // one chain, filled in arrival order
                        // The chain was sized from PASS 0's count of this line.  A second pass that is not the
                        // first one again (another table, a file that changed) must not write past it.
                        if (slot >= (uint32_t)MZ_CAP1 + (uint32_t)MZ_CAP * (h0 & MZ_HDR_LEN)) { atomicOr(failed, 2u); continue; }
                        const uint32_t e = slot - MZ_CAP1;
                        uint8_t *base = chain0 + (uint64_t)(e / MZ_CAP) * MZ_LINE;
                        reinterpret_cast<uint64_t *>(base)[e % MZ_CAP] = c;
                        reinterpret_cast<uint16_t *>(base + 8 * MZ_CAP)[e % MZ_CAP] = labels[koff + j];
                    } else {
                        // crowded: the k-mer's own chain, or -- it is full -- the one behind it (one spare chain
                        // follows the last); slots are claimed with compare-and-swap on the key
                        const uint32_t seg = seg_of(c, seg_log);
                        bool placed = false;
                        for (uint32_t e = 0; e < 2u * MZ_EMAX * MZ_CAP && !placed; e++) {
                            uint8_t *base = chain0 + ((uint64_t)seg * MZ_EMAX + e / MZ_CAP) * MZ_LINE;
                            unsigned long long *key = reinterpret_cast<unsigned long long *>(base) + e % MZ_CAP;
                            if (*key != MZ_EMPTY) continue;
                            if (atomicCAS(key, (unsigned long long)MZ_EMPTY, (unsigned long long)c) == MZ_EMPTY) {
                                reinterpret_cast<uint16_t *>(base + 8 * MZ_CAP)[e % MZ_CAP] = labels[koff + j];
                                if (e >= MZ_EMAX * MZ_CAP) atomicOr(hdr, MZ_HDR_TWO);
                                placed = true;
                            }
                        }
                        if (!placed) atomicOr(failed, 1u);             // two chains full at a third of the aimed load: ~1e-24
                    }
                }
            }
        }
        koff += cnt[i];
    }
}

// exclusive offsets (u64) of the per-workgroup k-mer counts of a chunk, on the device (one workgroup: a chunk has
// 16 K workgroups of buckets, the whole table 1.6 M), so that feeding a chunk needs no round trip to the host;
// a chunk whose bucket sizes do not add up to the k-mers the caller announced sets bit 2 of `failed`
static __global__ __launch_bounds__(256)
void mz_scan_blocks_kernel(const uint32_t *blk, uint32_t n, uint64_t *off, uint64_t expect, unsigned int *failed)
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
    if (threadIdx.x == 0 && tot != expect) atomicOr(failed, 4u);
}

// after PASS 0: extra lines of a line with `c` k-mers (crowded: 2^s chains of MZ_EMAX lines + one spare)
__host__ __device__ __forceinline__ uint32_t chain_len_of(uint32_t c)
{
    if (c <= (uint32_t)MZ_CAP) return 0u;
    const uint32_t ex = c / (uint32_t)MZ_CAP;                 // ceil((c - CAP1) / CAP): the first line keeps CAP1 = CAP - 1
    return ex > (uint32_t)MZ_EMAX ? (uint32_t)MZ_EMAX : ex;
}
// number of one-line chains (log2) of an overflowing, not crowded line: its k-mers beyond the first line go to the
// chain their hash picks, so a lookup reads ONE extra line instead of walking up to three
__host__ __device__ __forceinline__ uint32_t small_seg_log_of(uint32_t c)
{
    return c <= (uint32_t)MZ_CAP1 + (uint32_t)MZ_CAP ? 0u : c <= (uint32_t)MZ_CAP1 + 2u * (uint32_t)MZ_CAP ? 1u : 2u;
}
__host__ __device__ __forceinline__ uint32_t extras_of(uint32_t c)
{
    if (c <= (uint32_t)MZ_CAP) return 0u;
    const uint32_t s = seg_log_of(c);
    if (s) return ((1u << s) + 1u) * (uint32_t)MZ_EMAX;             // crowded: 2^s chains of MZ_EMAX lines + one spare chain
    const uint32_t t = small_seg_log_of(c);
    return t ? (1u << t) + 1u : 1u;                                   // 2^t one-line chains + one spare line (>= chain_len_of(c))
}
__host__ __device__ __forceinline__ uint32_t spilled_of(uint32_t c) { return c > MZ_CHAIN_CAP ? c - (uint32_t)MZ_CAP1 : 0u; }

// per workgroup of RL_BUCKETS lines: extra lines; overall: k-mers in segmented chains, overflowing lines, largest line, crowded lines
static __global__ __launch_bounds__(RL_THREADS)
void mz_extras_blocksum_kernel(const uint32_t *count, uint64_t n, uint32_t *blk_extra, unsigned long long *totals)
{
    __shared__ uint32_t s_a[RL_THREADS / 64];
    const uint64_t b0 = (uint64_t)blockIdx.x * RL_BUCKETS + (uint64_t)threadIdx.x * RL_PER_THREAD;
    uint32_t ex = 0, mx = 0;
    unsigned long long sp = 0, ov = 0, cr = 0;
    for (int i = 0; i < RL_PER_THREAD; i++) {
        const uint32_t c = (b0 + i < n) ? count[b0 + i] : 0u;
        ex += extras_of(c); sp += spilled_of(c); ov += c > (uint32_t)MZ_CAP ? 1u : 0u; cr += c > MZ_CHAIN_CAP ? 1u : 0u;
        mx = c > mx ? c : mx;
    }
    uint32_t te;
    block_exclusive_scan(ex, s_a, te);
    if (threadIdx.x == 0) blk_extra[blockIdx.x] = te;
    for (int o = 32; o > 0; o >>= 1) { sp += __shfl_xor(sp, o, 64); ov += __shfl_xor(ov, o, 64); cr += __shfl_xor(cr, o, 64); const uint32_t t = (uint32_t)__shfl_xor((int)mx, o, 64); mx = t > mx ? t : mx; }
    if ((threadIdx.x & 63) == 0) {
        if (sp) atomicAdd(&totals[0], sp);
        if (ov) atomicAdd(&totals[1], ov);
        atomicMax(&totals[2], (unsigned long long)mx);
        if (cr) atomicAdd(&totals[3], cr);
    }
}

static __global__ void mz_sum_u32_kernel(const uint32_t *v, uint64_t n, unsigned long long *out)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long s = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) s += v[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(out, s);
}

// headers: dword 30 = extra lines | spill flag (the placing pass ORs the Bloom bits in), dword 31 = first extra line
static __global__ __launch_bounds__(RL_THREADS)
void mz_header_kernel(const uint32_t *count, uint64_t n, const uint64_t *blk_extra_off, uint8_t *lines, uint32_t k)
{
    __shared__ uint32_t s_a[RL_THREADS / 64];
    const uint64_t b0 = (uint64_t)blockIdx.x * RL_BUCKETS + (uint64_t)threadIdx.x * RL_PER_THREAD;
    uint32_t c[RL_PER_THREAD], ex = 0;
    for (int i = 0; i < RL_PER_THREAD; i++) { c[i] = (b0 + i < n) ? count[b0 + i] : 0u; ex += extras_of(c[i]); }
    uint32_t tot;
    uint64_t o = blk_extra_off[blockIdx.x] + block_exclusive_scan(ex, s_a, tot);
    for (int i = 0; i < RL_PER_THREAD; i++) {
        if (b0 + i >= n) break;
        uint32_t *hdr = reinterpret_cast<uint32_t *>(lines + (b0 + i) * MZ_LINE) + 30;
        // (a line that does not overflow keeps header 0: "nothing behind this line" is one compare in the lookup)
        hdr[0] = c[i] > (uint32_t)MZ_CAP ? (chain_len_of(c[i]) | (seg_log_of(c[i]) << MZ_HDR_SEG_SHIFT) | MZ_HDR_CHAIN) : 0u;
        hdr[1] = (uint32_t)o;
        if (c[i] > (uint32_t)MZ_CAP) reinterpret_cast<uint64_t *>(lines + (b0 + i) * MZ_LINE)[MZ_CAP1] = bloom_base(k);
        o += extras_of(c[i]);
    }
}

// Lines that overflow: decide WHICH k-mers stay in the first line.  The placing pass fills slots in
// arrival order, which splits the k-mers of one minimizer (a super-k-mer of a genome: up to 11
// consecutive k-mers) between the first line and its extra lines, so a read crossing that region pays
// the dependent extra-line fetch for every such run.  Here the chain is rewritten with whole groups
// first: entries sorted by (size of their minimizer group, descending; minimizer key; k-mer).  One
// WAVE per overflowing line, one lane per k-mer of its chain (at most MZ_CHAIN_CAP = 48; crowded lines keep
// their arrival order): group size and rank are counted against every other entry by broadcast, so
// nothing is sorted in memory.  The result no longer depends on the order the atomics happened to run in.
static_assert(MZ_CHAIN_CAP <= 64, "one lane per chain entry");
static_assert((MZ_RUNS & (MZ_RUNS - 1)) == 0, "runs per batch: a power of two (index mask in the match)");
static __global__ __launch_bounds__(256)
void mz_regroup_kernel(const uint32_t *count, uint32_t n_lines, uint32_t k, uint32_t m,
                       uint8_t *lines, uint8_t *extra_lines)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t base = wave * 64u; base < n_lines; base += n_waves * 64u) {
        const uint64_t mine = base + lane;
        const uint32_t n_mine = mine < n_lines ? count[mine] : 0u;
        uint64_t todo = __ballot(n_mine > (uint32_t)MZ_CAP && n_mine <= MZ_CHAIN_CAP);
        while (todo) {
            const uint32_t l = (uint32_t)__ffsll((unsigned long long)todo) - 1u;
            todo &= todo - 1u;
            const uint64_t i = base + l;
            const uint32_t n = lane_bcast(n_mine, l);
            uint8_t *first = lines + i * MZ_LINE;
            uint8_t *more = extra_lines + (uint64_t)reinterpret_cast<const uint32_t *>(first)[31] * MZ_LINE;
            // entry e as the placing pass left it: the first MZ_CAP1 in the first line, the rest in the chain, in order
            const bool have = lane < n;
            uint64_t key = ~0ull, mk = ~0ull;
            uint32_t lab = 0;
            if (have) {
                const uint32_t e = lane < (uint32_t)MZ_CAP1 ? lane : lane - (uint32_t)MZ_CAP1;
                const uint8_t *L = lane < (uint32_t)MZ_CAP1 ? first : more + (uint64_t)(e / MZ_CAP) * MZ_LINE;
                key = reinterpret_cast<const uint64_t *>(L)[e % MZ_CAP];
                lab = reinterpret_cast<const uint16_t *>(L + 8 * MZ_CAP)[e % MZ_CAP];
                mk = kmer_min_key(key, k, m);
            }
            // a group = the k-mers of ONE target that share the minimizer: what one read's run asks for (related
            // genomes share minimizers; their k-mers around it differ and belong to different targets)
            uint32_t gsz = 0;
            for (uint32_t f = 0; f < n; f++) gsz += (lane_bcast64(mk, f) == mk && lane_bcast(lab, f) == lab) ? 1u : 0u;
            uint32_t rank = 0;                       // entries that come before this one (keys are distinct)
            for (uint32_t f = 0; f < n; f++) {
                const uint32_t gf = lane_bcast(gsz, f), lf = lane_bcast(lab, f);
                const uint64_t mf = lane_bcast64(mk, f), kf = lane_bcast64(key, f);
                const bool before = gf != gsz ? gf > gsz : (mf != mk ? mf < mk : (lf != lab ? lf < lab : kf < key));
                rank += before ? 1u : 0u;
            }
            // Where the entries beyond the first line go.  Up to 12 of them: one extra line.  More: 2 or 4 one-line
            // chains, an entry in the chain its hash picks (seg_of) or -- that one is full -- in the line behind it,
            // so that a lookup reads one extra line, rarely two.  If some entry would land further away the line
            // falls back to one linear chain (read front to back, as all chains were before).
            const bool over = have && rank >= (uint32_t)MZ_CAP1;
            const uint32_t n_over = n - (uint32_t)MZ_CAP1;
            uint32_t seg_log = small_seg_log_of(n);
            uint32_t dst_line = over ? (rank - MZ_CAP1) / MZ_CAP : 0u, dst_slot = over ? (rank - MZ_CAP1) % MZ_CAP : 0u;   // linear
            uint32_t len = (n_over + MZ_CAP - 1u) / MZ_CAP, two = 0u;
            uint32_t fill[6] = {0u, 0u, 0u, 0u, 0u, 0u};          // used slots of each extra line (wave-uniform)
            if (seg_log) {
                const uint32_t n_seg = 1u << seg_log;
                const uint32_t home = seg_of(key, seg_log);
                uint32_t cur = home, hl = 0u, hs = 0u;
                bool placed = false;
                for (uint32_t sg = 0; sg <= n_seg; sg++) {
                    const uint64_t want = __ballot(over && !placed && cur == sg);
                    const uint32_t pos = (uint32_t)__popcll(want & ((1ull << lane) - 1ull));
                    if (over && !placed && cur == sg) {
                        if (pos < (uint32_t)MZ_CAP) { placed = true; hl = sg; hs = pos; }
                        else cur = sg + 1u;
                    }
                    const uint32_t cnt = (uint32_t)__popcll(want);
                    fill[sg] = cnt < (uint32_t)MZ_CAP ? cnt : (uint32_t)MZ_CAP;
                }
                const bool bad = __ballot(over && (!placed || hl > home + 1u)) != 0;
                if (!bad) {
                    dst_line = hl; dst_slot = hs;
                    len = 1u;
                    two = __ballot(over && hl == home + 1u) ? MZ_HDR_TWO : 0u;
                } else {
                    seg_log = 0u;
                }
            }
            if (!seg_log)
                for (uint32_t e = 0; e < 6u; e++) {
                    const uint32_t used = n_over > e * MZ_CAP ? n_over - e * MZ_CAP : 0u;
                    fill[e] = used < (uint32_t)MZ_CAP ? used : (uint32_t)MZ_CAP;
                }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // every entry is in registers before any is rewritten
            __builtin_amdgcn_wave_barrier();
            uint64_t bloom = 0;
            if (have) {
                uint8_t *L = over ? more + (uint64_t)dst_line * MZ_LINE : first;
                const uint32_t sl = over ? dst_slot : rank;
                reinterpret_cast<uint64_t *>(L)[sl] = key;
                reinterpret_cast<uint16_t *>(L + 8 * MZ_CAP)[sl] = (uint16_t)lab;
                if (over) bloom = bloom_bits(key, k >= 32u);
            }
            // slots of the extra lines that hold nothing (the placing pass wrote entries in arrival order there)
            const uint32_t n_alloc = extras_of(n);
            for (uint32_t idx = lane; idx < n_alloc * (uint32_t)MZ_CAP; idx += 64u) {
                const uint32_t e = idx / MZ_CAP, sl = idx % MZ_CAP;
                const uint32_t used = e == 0 ? fill[0] : e == 1 ? fill[1] : e == 2 ? fill[2] : e == 3 ? fill[3] : e == 4 ? fill[4] : fill[5];
                if (sl >= used) reinterpret_cast<uint64_t *>(more + (uint64_t)e * MZ_LINE)[sl] = MZ_EMPTY;
            }
            for (int o = 32; o > 0; o >>= 1) bloom |= __shfl_xor(bloom, o, 64);
            if (lane == 0) {
                uint32_t *hdr = reinterpret_cast<uint32_t *>(first) + 30;
                hdr[0] = len | two | (seg_log << MZ_HDR_SEG_SHIFT) | MZ_HDR_CHAIN;
                reinterpret_cast<uint64_t *>(first)[MZ_CAP1] = bloom_base(k) | bloom;        // key slot 11: the Bloom word
            }
        }
    }
}

// Last build pass: the keys of every line (first lines and extra lines, each on its own) in ascending order (unused
// slots = all ones come last), labels moved along.  A lookup then compares its k-mer with the 7th key and scans only the half of the line that can
// hold it (mz_match_line): 6 compares instead of 12.  It also makes the first lines independent of the order
// the placing pass's atomics ran in.  One lane per line, odd-even transposition in registers (build time only).
static __global__ __launch_bounds__(256)
void mz_sort_lines_kernel(uint8_t *lines, uint32_t n_lines)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_lines; i += stride) {
        uint8_t *L = lines + i * MZ_LINE;
        uint64_t key[MZ_CAP];
        uint32_t lab[MZ_CAP];
        {
            const u32x4 *L4 = reinterpret_cast<const u32x4 *>(L);
#pragma unroll
            for (int t = 0; t < MZ_CAP / 2; t++) {
                const u32x4 v = L4[t];
                key[2 * t] = (uint64_t)v[0] | ((uint64_t)v[1] << 32);
                key[2 * t + 1] = (uint64_t)v[2] | ((uint64_t)v[3] << 32);
            }
            const u32x4 a = L4[6];
            const uint2 b = *reinterpret_cast<const uint2 *>(L + 112);
            const uint32_t w[6] = {a[0], a[1], a[2], a[3], b.x, b.y};
#pragma unroll
            for (int t = 0; t < MZ_CAP; t++) lab[t] = (w[t >> 1] >> (16 * (t & 1))) & 0xFFFFu;
        }
        bool moved = false;
#pragma unroll
        for (int r = 0; r < MZ_CAP; r++) {
#pragma unroll
            for (int e = r & 1; e + 1 < MZ_CAP; e += 2) {
                const bool sw = key[e] > key[e + 1];
                const uint64_t ka = sw ? key[e + 1] : key[e], kb = sw ? key[e] : key[e + 1];
                const uint32_t la = sw ? lab[e + 1] : lab[e], lb = sw ? lab[e] : lab[e + 1];
                key[e] = ka; key[e + 1] = kb; lab[e] = la; lab[e + 1] = lb;
                moved = moved || sw;
            }
        }
        if (moved) {
            u32x4 *L4 = reinterpret_cast<u32x4 *>(L);
#pragma unroll
            for (int t = 0; t < MZ_CAP / 2; t++)
                L4[t] = u32x4{(uint32_t)key[2 * t], (uint32_t)(key[2 * t] >> 32), (uint32_t)key[2 * t + 1], (uint32_t)(key[2 * t + 1] >> 32)};
            L4[6] = u32x4{lab[0] | (lab[1] << 16), lab[2] | (lab[3] << 16), lab[4] | (lab[5] << 16), lab[6] | (lab[7] << 16)};
            *reinterpret_cast<uint2 *>(L + 112) = make_uint2(lab[8] | (lab[9] << 16), lab[10] | (lab[11] << 16));
        }
    }
}

// ---------------------------------------------------------------------------
// query
// ---------------------------------------------------------------------------
#ifdef MC_MZ_STATS      // measurement builds only (tools/stats_build.sh): where the lookups of a workload end
__device__ unsigned long long g_mz_stats[16];
#define MZ_STAT(i, v) do { if (__builtin_amdgcn_inverse_ballot_w64(1ull)) atomicAdd(&g_mz_stats[i], (unsigned long long)(v)); } while (0)
#else
#define MZ_STAT(i, v) do { } while (0)
#endif

struct MzArgs {
    QueryArgs q;               // reads, outputs, shard range, div, k, maxhits, flags (lines unused)
    const uint8_t *lines;      // the primary lines this context owns
    const uint8_t *extra;      // extra lines
    uint32_t n_lines;          // primary lines of THIS context (every part of a table has the same number)
    uint32_t part, n_parts;    // MZ_LINES: this context answers for the minimizers with part_of(K) == part
    uint32_t m;
    double inv_htsize;         // 1/HTSIZE when the bucket-range filter may use rem_u64_fp, else 0
};

// which k-mers a context answers for
enum { MZ_ALL = 0,             // the whole table
       MZ_BUCKETS = 1,         // k-mers of a bucket range [shard_begin, shard_end) of the reference's hash
                               // (CuClarkDB.cu:552-559, :1212-1214): every shard fetches nearly every line
       MZ_LINES = 2 };         // the k-mers of 1/G of the MINIMIZERS (part_of): a part fetches, matches and scores 1/G of the runs

// One lane against one 128-byte line parked in LDS.  The keys of a first line are in ascending order
// (mz_sort_lines_kernel; unused slots = all ones last), so the 7th key says which half of the line can hold
// the k-mer: 6 compares instead of 12, 3 wide LDS reads instead of 6, for one more LDS round trip -- the 7th
// key and the header of BOTH positions of a lane are read before either half (mz_line_head).  The label is
// read only by lanes that matched.
struct MzHead { uint64_t pivot; uint32_t hdr, extra_base; };
__device__ __forceinline__ MzHead mz_line_head(const uint8_t *line)
{
    MzHead h;
    h.pivot = *reinterpret_cast<const uint64_t *>(line + 8 * (MZ_CAP / 2));
    const uint2 tail = *reinterpret_cast<const uint2 *>(line + MZ_LINE - 8);      // header, extra base
    h.hdr = tail.x;
    h.extra_base = tail.y;
    return h;
}
// slot of the line that holds c, MZ_CAP when none does
__device__ __forceinline__ uint32_t mz_match_line(const uint8_t *line, uint64_t pivot, uint64_t c)
{
    const uint32_t half = c >= pivot ? (uint32_t)(MZ_CAP / 2) : 0u;               // first slot of the half
    const u32x4 *L4 = reinterpret_cast<const u32x4 *>(line + 8u * half);
    u32x4 kv[MZ_CAP / 4];
#pragma unroll
    for (int t = 0; t < MZ_CAP / 4; t++) kv[t] = L4[t];
    uint32_t at = (uint32_t)MZ_CAP - half;              // "none" once the half is added
#pragma unroll
    for (int e = 0; e < MZ_CAP / 2; e++) {
        const uint64_t key = (uint64_t)kv[e >> 1][2 * (e & 1)] | ((uint64_t)kv[e >> 1][2 * (e & 1) + 1] << 32);
        at = key == c ? (uint32_t)e : at;
    }
    return at + half;
}
__device__ __forceinline__ uint32_t mz_line_label(const uint8_t *line, uint32_t slot)
{
    return reinterpret_cast<const uint16_t *>(line + 8 * MZ_CAP)[slot];
}

#ifndef MC_MZ_SORTED_EXTRA
#define MC_MZ_SORTED_EXTRA 1      // extra lines are sorted like first lines (mc_api.hip index_end): half-line scan there too
#endif
// an extra line parked in LDS, all 12 slots (what a lookup did before extra lines were sorted as well)
__device__ __forceinline__ uint32_t mz_match_full(const uint8_t *line, uint64_t c)
{
    const u32x4 *L4 = reinterpret_cast<const u32x4 *>(line);
    u32x4 kv[MZ_CAP / 2];
#pragma unroll
    for (int t = 0; t < MZ_CAP / 2; t++) kv[t] = L4[t];
    uint32_t at = MZ_CAP;
#pragma unroll
    for (int e = 0; e < MZ_CAP; e++) {
        const uint64_t key = (uint64_t)kv[e >> 1][2 * (e & 1)] | ((uint64_t)kv[e >> 1][2 * (e & 1) + 1] << 32);
        at = key == c ? (uint32_t)e : at;
    }
    return at;
}

// Occupancy is what this kernel lives on (measured on the headline workload, after the register
// diet: 5 waves/SIMD 934, 6 waves 1022, 7 waves 1047-1056, 8 waves (24 runs per batch to fit the LDS,
// no scratch) 1032 Mreads/s): <= 72 VGPRs and <= 5760 bytes of LDS per wave give 7 workgroups per CU
// (5792 bytes already drop one: 776 Mreads/s).
#ifndef MC_MZ_MIN_WAVES
#define MC_MZ_MIN_WAVES 7
#endif
#ifndef MC_MZ_STAGE_CON
#define MC_MZ_STAGE_CON 448    // u16 containers of a read group staged per wave (16 reads x 150 bp = 320)
#endif
static constexpr int MZ_STAGE_CON = MC_MZ_STAGE_CON;
// per-wave LDS: staged containers | line index of each run | parked lines; the m-mer keys of a
// step (dead once the window minima are taken) share the space of the parked lines
static constexpr int MZ_LDS_SLICE = (MZ_STAGE_CON + 16) * 2;
static constexpr int MZ_LDS_RUNLINE = MZ_LDS_SLICE;
static constexpr int MZ_LDS_LINES = MZ_LDS_RUNLINE + MZ_RUNS * 4;
static constexpr int MZ_LDS_KEYS_BYTES = (64 * MZ_NS + MZ_MAXW + 3) * 8;
static constexpr int MZ_LDS_RDPTR = MZ_LDS_LINES + (MZ_RUNS * MZ_LSTRIDE > MZ_LDS_KEYS_BYTES ? MZ_RUNS * MZ_LSTRIDE : MZ_LDS_KEYS_BYTES);
// container offsets of the group's reads (GROUP_READS + 1 of them): read from here inside the read loop, so that the
// register they are loaded into does not have to live through it
static constexpr int MZ_LDS_WAVE = MZ_LDS_RDPTR + ((GROUP_READS + 1) * 4 + 15) / 16 * 16;
static_assert(MZ_LDS_LINES % 16 == 0 && MZ_LDS_WAVE % 16 == 0, "16-byte aligned LDS regions");
// SHARD: MZ_ALL / MZ_BUCKETS / MZ_LINES (separate instantiations keep the divider and the ranges
// out of the unsharded kernel's scalar registers).
// KC: the k-mer length as a compile-time constant (31: cuCLARK's default, 27: cuCLARK-l), 0 = read from the
// arguments.  With a run-time k the shift amounts, masks and "k < 32" / "k >= 17" switches derived from it were kept
// in scalar registers the kernel does not have: a dozen of them were reloaded from spill lanes (v_readlane) at the
// start of every step.  As constants they are immediates and the switches disappear.
template <int SHARD, int KC>
__global__ __launch_bounds__(BLOCK_THREADS, MC_MZ_MIN_WAVES)
void mz_query_kernel(const MzArgs A)
{
    constexpr bool SHARDED = SHARD == MZ_BUCKETS;      // per-k-mer filter; MZ_LINES filters whole runs
    const QueryArgs &a = A.q;
    __shared__ __attribute__((aligned(16))) uint8_t s_mem[WAVES_PER_BLOCK][MZ_LDS_WAVE];

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform: scalar registers
    uint16_t *slice = reinterpret_cast<uint16_t *>(s_mem[wave]);
    uint32_t *runline = reinterpret_cast<uint32_t *>(s_mem[wave] + MZ_LDS_RUNLINE);
    uint8_t *linebuf = s_mem[wave] + MZ_LDS_LINES;
    uint64_t *keyv = reinterpret_cast<uint64_t *>(linebuf);     // aliases the parked lines (see above)
    uint32_t *rdptr = reinterpret_cast<uint32_t *>(s_mem[wave] + MZ_LDS_RDPTR);

    const uint32_t k = KC ? (uint32_t)KC : a.k, m = KC ? mmer_len((uint32_t)KC) : A.m;
    constexpr uint32_t W = MZ_MAXW;            // k - m + 1 windows: m = k - (MZ_MAXW - 1) (mmer_len)
    const uint64_t kmask = k >= 32 ? ~0ull : ((1ull << (2u * k)) - 1ull);
    const uint64_t mmask = (1ull << (2u * m)) - 1ull;
    // a batch holds fewer than 2^32 reads and containers (reads_ptr is u32; launch_query checks): 32-bit
    // counters, half the scalar registers
    const uint32_t n_reads = a.n_dev ? a.n_dev[0] : (uint32_t)a.n_reads, n_con = a.n_dev ? a.n_dev[1] : (uint32_t)a.n_containers;
    const uint32_t n_groups = (n_reads + (GROUP_READS - 1)) / GROUP_READS;
    const uint32_t gstride = gridDim.x * WAVES_PER_BLOCK;
    // wave-uniform switches are tested where they are used, from ONE scalar register (hoisted out of the
    // loops they become lane masks that live in -- and are spilled from -- two registers each)
    auto flags_now = [&]() -> uint32_t { uint32_t f = a.flags; asm volatile("" : "+s"(f)); return f; };
    // Arguments that are needed once per read or less are read from the kernel-argument segment where they are used
    // (a scalar load from constant memory) instead of living in scalar registers through every loop: the kernel has
    // more wave-uniform values than registers for them, and what does not fit is parked in lanes of a vector
    // register and fetched back with v_readlane in the hot loops (loop counters among them).
    auto karg = [&](size_t off, auto type_c) {
        typedef decltype(type_c) T;
        const __attribute__((address_space(4))) char *p = (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(off));           // not hoisted, not merged with the loads at the kernel's start
        return *reinterpret_cast<const __attribute__((address_space(4))) T *>(p + off);
    };
    auto arg_maxhits = [&]() { return karg(offsetof(MzArgs, q) + offsetof(QueryArgs, maxhits), uint32_t()); };
    auto arg_sparse_rows = [&]() { return karg(offsetof(MzArgs, q) + offsetof(QueryArgs, sparse_rows), (uint16_t *)nullptr); };
    auto arg_final_rows = [&]() { return karg(offsetof(MzArgs, q) + offsetof(QueryArgs, final_rows), (uint16_t *)nullptr); };
    auto arg_over_maxhits = [&]() { return karg(offsetof(MzArgs, q) + offsetof(QueryArgs, over_maxhits), (unsigned long long *)nullptr); };
    auto arg_extra = [&]() { return karg(offsetof(MzArgs, extra), (const uint8_t *)nullptr); };
    auto arg_reads_ptr = [&]() { return karg(offsetof(MzArgs, q) + offsetof(QueryArgs, reads_ptr), (const uint32_t *)nullptr); };
    auto arg_containers = [&]() { return karg(offsetof(MzArgs, q) + offsetof(QueryArgs, containers), (const uint16_t *)nullptr); };

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
        // The group is staged into the wave's LDS slice in as few pieces as fit: usually all
        // 16 reads at once; long reads (2 x 250 bp pairs, contigs) in smaller pieces; a single
        // read larger than the slice is read from global memory.
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
                // the slice holds the containers as big-endian 64-bit words (4 containers each,
                // first base in the top bits): container i lives at u16 index i ^ 3, and any
                // k-mer is cut from two consecutive aligned words
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
        // `len` bases starting at base position p of the part whose containers start at `first`
        auto bases_at = [&](uint32_t first, uint32_t p, uint32_t len, uint64_t mask) -> uint64_t {
            if constexpr (STAGED) {
                uint32_t j0 = first - c0a + (p >> 3);                         // container index in the slice
                if (j0 > (uint32_t)(MZ_STAGE_CON + 4)) j0 = (uint32_t)(MZ_STAGE_CON + 4);
                const uint64_t *w = reinterpret_cast<const uint64_t *>(slice) + (j0 >> 2);
                const uint64_t wa = w[0], wb = w[1];
                const uint32_t b = 16u * (j0 & 3u) + 2u * (p & 7u);           // bit offset of the first base
                const uint64_t top = (wa << b) | ((wb >> 1) >> (63u - b));   // b in [0, 62]
                return top >> (64u - 2u * len);                               // len bases: nothing left to mask
            }
            p = opaque(p);          // rare path: keep its shift amounts out of the long-lived registers
            const uint32_t j0 = first + (p >> 3);
            const uint64_t hi = ((uint64_t)con(j0) << 48) | ((uint64_t)con(j0 + 1) << 32)
                              | ((uint64_t)con(j0 + 2) << 16) | (uint64_t)con(j0 + 3);
            const uint32_t lo = con(j0 + 4);
            const uint32_t sh = 80u - 2u * (p & 7u) - 2u * len;
            const uint64_t x = sh >= 16u ? (hi >> (sh - 16u)) : ((hi << (16u - sh)) | (uint64_t)(lo >> sh));
            return x & mask;
        };

        for (uint32_t ri = rs; ri < re; ri++) {
            const uint32_t beg = (uint32_t)__builtin_amdgcn_readfirstlane((int)rdptr[ri]);      // same address in every lane
            uint32_t end = (uint32_t)__builtin_amdgcn_readfirstlane((int)rdptr[ri + 1u]);
            if (end > n_con) end = n_con;
            uint32_t acc_t = 0xFFFFFFFFu, acc_c = 0, n_acc = 0;

            uint32_t pp = beg;
            while (pp < end) {
                const uint32_t plen = (uint32_t)__builtin_amdgcn_readfirstlane((int)con(pp));   // same address in every lane
                const uint32_t first = pp + 1;
                pp = first + (plen ? (plen - 1u) / 8u + 1u : 0u);
                if (plen < k) continue;
                const uint32_t nk = plen - k + 1u;

                for (uint32_t base = 0; base < nk; base += 64u * MZ_NS) {
                    // (1) lane l holds the two k-mers at positions base + 2l and base + 2l + 1: the second
                    //     is the first shifted by one base (no second cut from LDS, no second reverse
                    //     complement), and their two windows share W - 1 of W + 1 m-mer keys.  Keys of the
                    //     m-mers at base .. base + 64*NS + W - 2 go to LDS (m-mer p = first m bases of k-mer p).
                    static_assert(MZ_NS == 2, "two consecutive positions per lane");
                    bool     active[MZ_NS], leader[MZ_NS];
                    uint64_t c[MZ_NS];
                    uint32_t line[MZ_NS], run[MZ_NS];
                    uint64_t own[MZ_NS];           // lanes whose position holds a k-mer this context answers for
                    const bool last_step = base + 64u * MZ_NS >= nk;                // wave-uniform
                    // The m-mers of the part sit at positions 0 .. nm-1; the W-1 behind its last k-mer start
                    // no k-mer.  When they all fall inside this step's 64*NS positions (every read of up to
                    // 64*NS - W + k bases: 148 at k = 31 with 11 windows, 150 with 9) the lanes that own those
                    // positions cut them in the same instructions as everybody else -- the first m bases of
                    // the 32 bases they read are inside the part -- and the tail code below is skipped.
                    const uint32_t nm = nk + (W - 1u);
                    const bool tail_in_step = nm - base <= 64u * MZ_NS;             // wave-uniform
                    const uint32_t p0 = base + 2u * lane;
                    const uint64_t in0 = mask_lt_s(p0, nk), in1 = mask_lt_s(p0, nk - 1u);     // positions p0, p0 + 1 hold a k-mer
                    active[0] = __builtin_amdgcn_inverse_ballot_w64(in0);
                    active[1] = __builtin_amdgcn_inverse_ballot_w64(in1);
                    uint64_t x0 = 0, x1 = 0, rc0 = 0, rc1 = 0;
                    uint64_t key0 = MZ_KEY_NONE, key1 = MZ_KEY_NONE;
                    c[0] = 0; c[1] = 0;
                    // Staged: no predicate at all.  The LDS reads are clamped, and what a lane past the last
                    // m-mer computes is read by no k-mer of the part (a window ends at position nm - 1).
                    if (STAGED || p0 < (tail_in_step ? nm : nk)) {
                        if constexpr (STAGED) {
                            // container index in the slice.  No clamp: a position past the part's last m-mer (whose
                            // value nobody reads) stays below 2 * MZ_STAGE_CON + 32 containers, inside this wave's LDS
                            const uint32_t j0 = first - c0a + (p0 >> 3);
                            const uint64_t *w = reinterpret_cast<const uint64_t *>(slice) + (j0 >> 2);
                            const uint64_t wa = w[0], wb = w[1];
                            const uint32_t b = 16u * (j0 & 3u) + 2u * (p0 & 7u);      // bit offset of base p0: <= 60
                            const uint64_t top = (wa << b) | ((wb >> 1) >> (63u - b)); // 32 bases from p0
                            x0 = top >> (64u - 2u * k);
                            if (k < 32u) {
                                x1 = (top >> (62u - 2u * k)) & kmask;                 // the k bases from p0 + 1 are inside `top`
                            } else {
                                const uint64_t top1 = (top << 2) | ((wb >> (62u - b)) & 3ull);   // 32 bases from p0 + 1
                                x1 = top1;
                            }
                        } else {
                            x0 = bases_at(first, p0, k, kmask);
                            x1 = bases_at(first, p0 + 1u, k, kmask);
                        }
                        rc0 = revcomp(x0, k);
                        if (k >= 17u) {       // the complement of x1's last base lands in the high word: one shift-or
                            const uint64_t r2 = rc0 >> 2;
                            rc1 = ((uint64_t)((uint32_t)(r2 >> 32) | ((~(uint32_t)x1 & 3u) << (2u * k - 34u))) << 32) | (uint32_t)r2;
                        } else {
                            rc1 = (rc0 >> 2) | ((uint64_t)(3u - ((uint32_t)x1 & 3u)) << (2u * k - 2u));
                        }
                        key0 = mmer_key2(x0 >> (2u * (k - m)), rc0 & mmask);        // first m bases, both strands
                        // (canonical form: k = 32 values reach 2^64, everything shorter stays below 2^62)
                        c[0] = k < 32u ? min_below_2_62(x0, rc0) : (x0 < rc0 ? x0 : rc0);
                        // also right for the lane whose second position is nk (no k-mer there): the m-mer at nk
                        // lies inside the part
                        key1 = mmer_key2(x1 >> (2u * (k - m)), rc1 & mmask);
                        c[1] = k < 32u ? min_below_2_62(x1, rc1) : (x1 < rc1 ? x1 : rc1);
                        if constexpr (SHARDED) {
                            // runs are formed over ALL k-mers of the part (inpart); only the k-mers of this
                            // shard are looked up (active), and only runs with such a k-mer are fetched
#pragma unroll
                            for (int s = 0; s < MZ_NS; s++) {
                                uint64_t r;
                                if (A.inv_htsize != 0.0) {              // 4^k <= HTSIZE * 2^32: every real table
                                    r = rem_u64_fp(c[s], (uint32_t)a.div.d, A.inv_htsize);
                                } else {
                                    const uint64_t q = div_u64(c[s], a.div);
                                    r = c[s] - q * a.div.d;
                                }
                                active[s] = active[s] && (r >= a.shard_begin) && (r < a.shard_end);
                            }
                        }
                    }
                    {
                        typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
                        *reinterpret_cast<u64x2 *>(keyv + 2u * lane) = u64x2{key0, key1};
                    }
                    if (!tail_in_step) {
                        // the W-1 m-mers that start behind the part's last k-mer (positions nk .. nk+W-2) all
                        // lie inside that k-mer: cut from its value, stored on top of the stores above
                        const uint32_t q = nk - 1u - base;                               // last k-mer: lane q/2, half q%2
                        uint64_t x_last = 0, rc_last = 0;
                        if (last_step) {                                                 // every lane takes part in the select
                            x_last = lane_bcast64((q & 1u) ? x1 : x0, q >> 1);
                            rc_last = lane_bcast64((q & 1u) ? rc1 : rc0, q >> 1);
                        }
                        if (lane < W - 1u) {
                            const uint32_t ln = opaque(lane);
                            if (last_step) {
                                const uint32_t i = ln + 1u;
                                keyv[64 * MZ_NS + ln] = (uint64_t)opaque((uint32_t)(MZ_KEY_NONE >> 32)) << 32;     // (made here, not kept in registers)
                                keyv[q + i] = mmer_key2((x_last >> (2u * (k - m - i))) & mmask, (rc_last >> (2u * i)) & mmask);
                            } else {
                                keyv[64 * MZ_NS + ln] = mmer_key(bases_at(first, base + 64u * MZ_NS + ln, m, mmask), m);
                            }
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    // (2) minimizer key = window minimum, (3) line, (4) runs of equal lines
                    uint32_t n_runs;
                    {
                        uint64_t v[MZ_MAXW + 1];
#pragma unroll
                        for (int i = 0; i < MZ_MAXW + 1; i++) v[i] = keyv[2u * lane + i];
                        uint64_t mid = v[1];
#pragma unroll
                        for (int i = 2; i < MZ_MAXW; i++) mid = key_min(mid, v[i]);
                        const uint64_t K0 = key_min(v[0], mid), K1 = key_min(mid, v[MZ_MAXW]);
                        // (what a lane without a k-mer computes here is never looked at: a position with a
                        // k-mer has one before it, and only leaders and active lanes use their line)
                        line[0] = line_of(K0, A.n_lines);
                        line[1] = line_of(K1, A.n_lines);
                        uint64_t own0 = in0, own1 = in1;
                        if constexpr (SHARD == MZ_LINES) {
                            // k-mers whose minimizer belongs to another part drop out here, and only owned
                            // runs are numbered (fetched, matched, scored).  Two runs of one read that differ in
                            // part but collide in the line index are told apart by ownership: a run is a
                            // maximal stretch of OWNED positions with one line.
                            own0 &= mask_eq_s(part_of(K0, A.n_parts), A.part);
                            own1 &= mask_eq_s(part_of(K1, A.n_parts), A.part);
                            active[0] = __builtin_amdgcn_inverse_ballot_w64(own0);
                            active[1] = __builtin_amdgcn_inverse_ballot_w64(own1);
                        }
                        // second line of the lane before (DPP wave_shr:1); nothing before lane 0
                        const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)line[1], 0x138, 0xf, 0xf, false);
                        uint64_t b0 = mask_ne(line[0], prev), b1 = mask_ne(line[1], line[0]);
                        if constexpr (SHARD == MZ_LINES) {
                            // line indexes are per part: the position before may show the same index and belong
                            // to another part's line -- an owned position behind one that is not owned leads a run
                            b0 |= ~(own1 << 1);
                            b1 |= ~own0;
                        }
                        b0 &= own0; b1 &= own1;
                        own[0] = own0; own[1] = own1;
                        leader[0] = __builtin_amdgcn_inverse_ballot_w64(b0);
                        leader[1] = __builtin_amdgcn_inverse_ballot_w64(b1);
                        // leaders in lower lanes (v_mbcnt) = index of this lane's first run
                        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b0, 0u))
                                             + __builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0u));
                        run[0] = below + (leader[0] ? 1u : 0u) - 1u;
                        run[1] = run[0] + (leader[1] ? 1u : 0u);
                        n_runs = (uint32_t)__popcll(b0) + (uint32_t)__popcll(b1);
                    }

                    uint64_t hitm[MZ_NS] = {0ull, 0ull};      // lanes whose k-mer was found (lane masks: scalar registers)
                    uint32_t lab[MZ_NS] = {0u, 0u};
                    // one batch of up to MZ_RUNS runs; ALL = the step has no more than that (nearly always):
                    // no per-lane range tests
                    // runline[0, nb) lists lines of `base`: 8 lanes fetch one 128-byte line, MZ_RUNS/8 rounds, all
                    // issued before use.  Rounds are skipped as a whole (scalar branch); inside a round the lane
                    // groups past the last line re-read it (same request, no predication).  Parked in linebuf.
                    auto fetch_lines = [&](const uint8_t *base, const uint32_t nb, auto holes_c) {
                        constexpr bool HOLES = decltype(holes_c)::value;       // entries may be ~0: nothing to fetch
                        u32x4 v[MZ_RUNS / 8];
                        const uint32_t lf = opaque(lane);
                        const uint8_t *lane_base = base + (lf & 7u) * 16u;      // this lane's 16 bytes of a line
                        const uint32_t last = nb - 1u;
#pragma unroll
                        for (int rd = 0; rd < MZ_RUNS / 8; rd++) {
                            if (8u * rd < nb) {
                                const uint32_t j = 8u * rd + (lf >> 3);
#ifdef MC_MZ_DEBUG_WINDOW      // measurement builds only (WRONG results): every fetch inside a cache-resident window of lines
                                const uint32_t rl = runline[j < last ? j : last] & (uint32_t)(MC_MZ_DEBUG_WINDOW - 1);
#else
                                const uint32_t rl = runline[j < last ? j : last];
#endif
                                const u32x4 *src = reinterpret_cast<const u32x4 *>(lane_base + ((uint64_t)rl << 7));
                                if constexpr (HOLES) {
                                    v[rd] = u32x4{0u, 0u, 0u, 0u};
                                    if (rl != 0xFFFFFFFFu) v[rd] = __builtin_nontemporal_load(src);
                                } else {
                                    v[rd] = __builtin_nontemporal_load(src);
                                }
                            }
                        }
#pragma unroll
                        for (int rd = 0; rd < MZ_RUNS / 8; rd++) {
                            if (8u * rd < nb) {
                                const uint32_t j = 8u * rd + (lf >> 3);
                                *reinterpret_cast<u32x4 *>(linebuf + j * MZ_LSTRIDE + (lf & 7u) * 16u) = v[rd];
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                    };
                    auto batch = [&](const uint32_t rb, auto all_c) {
                        constexpr bool ALL = decltype(all_c)::value;
                        auto in_batch = [&](int s) -> bool { return ALL || (run[s] >= rb && run[s] < rb + MZ_RUNS); };
                        if constexpr (SHARDED) {
                            // any k-mer of this shard publishes its run's line; runs without one keep ~0
                            if (lane < (uint32_t)MZ_RUNS) runline[lane] = 0xFFFFFFFFu;
#pragma unroll
                            for (int s = 0; s < MZ_NS; s++)
                                if (active[s] && in_batch(s)) runline[run[s] - rb] = line[s];
                        } else {
#pragma unroll
                            for (int s = 0; s < MZ_NS; s++)
                                if (leader[s] && in_batch(s)) runline[run[s] - rb] = line[s];
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        const uint32_t nb = n_runs - rb < (uint32_t)MZ_RUNS ? n_runs - rb : (uint32_t)MZ_RUNS;
                        fetch_lines(A.lines, nb, std::integral_constant<bool, SHARDED>{});
                        // 7th key and header of both positions first, then the halves.  EVERY lane runs the match (a
                        // lane without a k-mer reads some parked line and is masked out afterwards): no exec-mask
                        // juggling, and who found what stays in lane masks.
                        const uint8_t *Lm[MZ_NS];
                        MzHead hd[MZ_NS];
#pragma unroll
                        for (int s = 0; s < MZ_NS; s++) {
                            Lm[s] = linebuf + ((run[s] - rb) & (uint32_t)(MZ_RUNS - 1)) * MZ_LSTRIDE;
                            hd[s] = mz_line_head(Lm[s]);
                        }
                        uint64_t pend[MZ_NS];                   // k-mers that missed on a line that has extra lines
#pragma unroll
                        for (int s = 0; s < MZ_NS; s++) {
                            uint64_t act = SHARDED ? (uint64_t)__builtin_amdgcn_ballot_w64(active[s]) : own[s];
                            if (!ALL) act &= mask_lt_s(run[s] - rb, (uint32_t)MZ_RUNS);
                            const uint32_t at = mz_match_line(Lm[s], hd[s].pivot, c[s]);
                            const uint64_t found = mask_ne(at, (uint32_t)MZ_CAP) & act;
                            if (__builtin_amdgcn_inverse_ballot_w64(found)) lab[s] = mz_line_label(Lm[s], at);
                            hitm[s] |= found;
#ifdef MC_MZ_DEBUG_NOCHAIN      // measurement builds only (WRONG results): what the lookups behind the first line cost
                            pend[s] = 0;
#else
                            pend[s] = act & ~found & mask_ne(hd[s].hdr, 0u);               // header 0: nothing behind this line
#endif
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
#ifdef MC_MZ_STATS
                        {
                            MZ_STAT(0, 1);                                                        // batches (steps)
                            MZ_STAT(1, nb);                                                       // runs
                            MZ_STAT(2, __popcll(own[0]) + __popcll(own[1]));                      // k-mers looked up
                            MZ_STAT(3, __popcll(hitm[0]) + __popcll(hitm[1]));                    // found in a first line
                            MZ_STAT(4, __popcll(pend[0]) + __popcll(pend[1]));                    // missed on a line with a chain (before Bloom)
                            const uint64_t ovl0 = __builtin_amdgcn_ballot_w64(leader[0] && (hd[0].hdr & MZ_HDR_LEN) != 0u);
                            const uint64_t ovl1 = __builtin_amdgcn_ballot_w64(leader[1] && (hd[1].hdr & MZ_HDR_LEN) != 0u);
                            MZ_STAT(5, __popcll(ovl0) + __popcll(ovl1));                          // runs on overflowing lines
                        }
#endif
                        if ((pend[0] | pend[1]) != 0) {
                            // Rare on a clean table: lines beyond the first.  A k-mer goes on only if both its Bloom
                            // bits are set; its chain is the one of the line's 2^s chains it hashes to (s = 0: the
                            // only one), plus the one behind it when header bit 2 says that some chain was full.
                            // first the filter alone -- most steps with a miss on such a line end here (9 reads in 10 of
                            // a genome-shaped table get this far, 6 in 10 further) --, then, for what passed, the chain
#pragma unroll
                            for (int s = 0; s < MZ_NS; s++) {
                                bool go = false;
                                if (__builtin_amdgcn_inverse_ballot_w64(pend[s]))
                                    go = bloom_pass(*reinterpret_cast<const uint64_t *>(Lm[s] + 8 * MZ_CAP1), c[s]);      // the line's Bloom word
                                pend[s] = __builtin_amdgcn_ballot_w64(go);
                            }
                            uint32_t xb[MZ_NS] = {0u, 0u}, xn[MZ_NS] = {0u, 0u};          // first line of the k-mer's chain, lines to look at
                            if ((pend[0] | pend[1]) != 0) {
                                bool seg_any = false;
#pragma unroll
                                for (int s = 0; s < MZ_NS; s++) {
                                    if (__builtin_amdgcn_inverse_ballot_w64(pend[s])) {
                                        const uint32_t hdr = hd[s].hdr, len = hdr & MZ_HDR_LEN;
                                        xn[s] = len << ((hdr >> 2) & 1u);
                                        xb[s] = hd[s].extra_base;
                                        seg_any = seg_any || (hdr & (MZ_HDR_SEG_MASK << MZ_HDR_SEG_SHIFT)) != 0u;
                                    }
                                }
                                if (__builtin_amdgcn_ballot_w64(seg_any) != 0) {          // hashed chains: rare
#pragma unroll
                                    for (int s = 0; s < MZ_NS; s++)
                                        if (__builtin_amdgcn_inverse_ballot_w64(pend[s]))
                                            xb[s] += seg_of(c[s], (hd[s].hdr >> MZ_HDR_SEG_SHIFT) & MZ_HDR_SEG_MASK) * (hd[s].hdr & MZ_HDR_LEN);
                                }
                            }
#ifdef MC_MZ_STATS
                            MZ_STAT(6, 1);                                                        // steps with a miss on a chained line
                            if ((pend[0] | pend[1]) != 0) MZ_STAT(7, 1);                          // steps that go to the chains
                            MZ_STAT(8, __popcll(pend[0]) + __popcll(pend[1]));                    // k-mers that go to the chains (after Bloom)
                            {   // distinct chain lines among neighbours (what a per-line job numbering would fetch)
                                const uint32_t pv = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)xb[1], 0x138, 0xf, 0xf, false);
                                const uint64_t d0 = mask_ne(xb[0], pv) & pend[0], d1 = (mask_ne(xb[1], xb[0]) | ~pend[0]) & pend[1];
                                MZ_STAT(9, __popcll(d0) + __popcll(d1));
                            }
#endif
                            // Round r looks at line r of each pending k-mer's chain, for both positions of every lane
                            // at once.  The lines are fetched exactly like first lines -- the pending lookups are
                            // numbered, publish their line, 8 lanes fetch one line, the lines are parked in LDS -- so a
                            // step pays ONE more memory round trip however many of its k-mers go on (six 16-byte loads
                            // per lane and position, one position after the other, before).
                            for (uint32_t r = 0; (pend[0] | pend[1]) != 0; r++) {
                                // One job per chain LINE, not per k-mer: the pending k-mers of a run sit side by side and
                                // mostly want the same line (round 2 fetched it once for each of them: 6.6 lines per read
                                // on the genome-shaped table where 3.8 are distinct).  A pending position leads a job when
                                // the position before it is not pending or wants another line -- as runs are numbered.
                                const uint32_t pxb = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)xb[1], 0x138, 0xf, 0xf, false);
                                const uint64_t d0 = pend[0] & (mask_ne(xb[0], pxb) | ~(pend[1] << 1));
                                const uint64_t d1 = pend[1] & (mask_ne(xb[1], xb[0]) | ~pend[0]);
                                const uint32_t n = (uint32_t)__popcll(d0) + (uint32_t)__popcll(d1);
                                uint32_t job[MZ_NS];
                                {
                                    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(d0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)d0, 0u))
                                                         + __builtin_amdgcn_mbcnt_hi((uint32_t)(d1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)d1, 0u));
                                    job[0] = below + (__builtin_amdgcn_inverse_ballot_w64(d0) ? 1u : 0u) - 1u;
                                    job[1] = job[0] + (__builtin_amdgcn_inverse_ballot_w64(d1) ? 1u : 0u);
                                }
                                uint64_t found[MZ_NS] = {0ull, 0ull};
                                for (uint32_t ch = 0; ch < n; ch += MZ_RUNS) {
                                    MZ_STAT(10, 1);                                               // chain fetch rounds (of up to 32 lines)
                                    MZ_STAT(11, n - ch < (uint32_t)MZ_RUNS ? n - ch : (uint32_t)MZ_RUNS);   // chain lines fetched
                                    uint64_t here[MZ_NS];
#pragma unroll
                                    for (int s = 0; s < MZ_NS; s++) {
                                        here[s] = pend[s] & mask_lt_s(job[s] - ch, (uint32_t)MZ_RUNS);
                                        if (__builtin_amdgcn_inverse_ballot_w64(here[s] & (s == 0 ? d0 : d1))) runline[job[s] - ch] = xb[s] + r;
                                    }
                                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                                    __builtin_amdgcn_wave_barrier();
                                    fetch_lines(arg_extra(), n - ch < (uint32_t)MZ_RUNS ? n - ch : (uint32_t)MZ_RUNS, std::false_type{});
#pragma unroll
                                    for (int s = 0; s < MZ_NS; s++) {
                                        if (here[s] == 0) continue;
                                        const uint8_t *X = linebuf + ((job[s] - ch) & (uint32_t)(MZ_RUNS - 1)) * MZ_LSTRIDE;
#if MC_MZ_SORTED_EXTRA
                                        const uint32_t at = mz_match_line(X, *reinterpret_cast<const uint64_t *>(X + 8 * (MZ_CAP / 2)), c[s]);
#else
                                        const uint32_t at = mz_match_full(X, c[s]);
#endif
                                        const uint64_t f = mask_ne(at, (uint32_t)MZ_CAP) & here[s];
                                        if (__builtin_amdgcn_inverse_ballot_w64(f)) lab[s] = mz_line_label(X, at);
                                        found[s] |= f;
                                    }
                                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                                    __builtin_amdgcn_wave_barrier();
                                }
                                // still pending: not found yet and another line in the chain
                                MZ_STAT(12, __popcll(found[0]) + __popcll(found[1]));            // found in a chain
#pragma unroll
                                for (int s = 0; s < MZ_NS; s++) {
                                    hitm[s] |= found[s];
                                    pend[s] &= ~found[s] & mask_gt_s(xn[s], r + 1u);
                                }
                            }
                        }
                    };
                    if (n_runs <= (uint32_t)MZ_RUNS) {
                        if (n_runs) batch(0u, std::true_type{});
                    } else {
                        for (uint32_t rb = 0; rb < n_runs; rb += MZ_RUNS) batch(rb, std::false_type{});
                    }

                    // fold the hits into the accumulator (as in query_kernel)
                    uint64_t mm[MZ_NS];
                    uint64_t many = 0;
#pragma unroll
                    for (int s = 0; s < MZ_NS; s++) { mm[s] = hitm[s]; many |= mm[s]; }
                    while (many) {
                        uint32_t t = 0;
                        bool got = false;
#pragma unroll
                        for (int s = 0; s < MZ_NS; s++) {
                            if (!got && mm[s]) {
                                t = lane_bcast(lab[s], (uint32_t)(__ffsll((unsigned long long)mm[s]) - 1));
                                got = true;
                            }
                        }
                        uint32_t cnt = 0;
                        many = 0;
#pragma unroll
                        for (int s = 0; s < MZ_NS; s++) {
                            const uint64_t same = mask_eq_s(lab[s], t) & mm[s];      // mm: hits not counted yet
                            cnt += (uint32_t)__popcll(same);
                            mm[s] &= ~same;
                            many |= mm[s];
                        }
                        const uint64_t ex = __ballot(acc_t == t);
                        if (ex) {
                            if (acc_t == t) acc_c += cnt;
                        } else if (n_acc < 64u) {
                            if (lane == n_acc) { acc_t = t; acc_c = cnt; }
                            n_acc++;
                        } else {
                            const uint32_t mx = wave_max_u32(acc_t);
                            if (t < mx) {
                                const uint64_t who = __ballot(acc_t == mx);
                                if (lane == (uint32_t)(__ffsll((unsigned long long)who) - 1)) { acc_t = t; acc_c = cnt; }
                            }
                        }
                    }
                }
            }

            // ---- finalisation (identical to query_kernel) ---------------------
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
                    if (__builtin_amdgcn_inverse_ballot_w64(1ull)) atomicAdd(arg_over_maxhits(), 1ull);      // lane 0
                }
            }
            const uint32_t n_keep = n_acc > maxhits ? maxhits : n_acc;
            if (flags_now() & 2u) {
                uint16_t *row = arg_sparse_rows() + rd * row_len;
                if (__builtin_amdgcn_inverse_ballot_w64(1ull)) row[0] = (uint16_t)n_keep;           // lane 0
                if (valid) { row[1 + 2 * rank] = (uint16_t)acc_t; row[2 + 2 * rank] = (uint16_t)sat_u16(acc_c); }
                for (uint32_t i = 1u + 2u * n_keep + lane; i < row_len; i += 64u) row[i] = 0;
            }
            if (flags_now() & 1u) {
                // the row [sumN, idxBest+1, best, idxSecond+1, second] as five wave-uniform values
                uint32_t o0, o1, o2, o3 = 0u, o4 = 0u;
                if (n_acc <= 1u) {              // most reads hit no target or one: lane 0 has it all
                    const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)acc_t, 0);
                    const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)acc_c, 0);
                    const bool has = n_acc == 1u;
                    o0 = has ? sat_u16(c0) : 0u;
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
                // the five values are wave-uniform: packed on the scalar side and stored by lane 0 (8 + 2 bytes at a
                // 2-byte aligned address; global memory takes unaligned stores) instead of one select chain per lane
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
        // the slice is rewritten by the next piece: keep the compiler from hoisting its stores
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        }   // pieces of the group
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

} // namespace mz
} // namespace mc
