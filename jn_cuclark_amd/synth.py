"""Deterministic synthetic workloads: genomes, reads, databases.

Everything is derived from splitmix64(seed + counter) so tests, bench.py and the CPU
baseline see the same bytes.  numpy only; the large bench workload has a torch
(on-GPU) twin in ``synth_gpu``.  No dependency on ``oracle/``.

2-bit codes follow the reference packer: A=3 C=2 G=1 T=0, first base most significant
(reference src/CuCLARK_hh.hh:294-297, :1660-1661).
"""
import numpy as np

CODE2BASE = np.frombuffer(b"TGCA", dtype=np.uint8)   # code -> ASCII
M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    """splitmix64 finaliser on a uint64 array (wraps modulo 2^64)."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def rand_u64(seed, n, stream=0):
    with np.errstate(over="ignore"):
        base = splitmix64(np.uint64(seed) * np.uint64(0x632BE59BD9B4E019) + np.uint64(stream))
        return splitmix64(base + np.arange(n, dtype=np.uint64))


def random_codes(seed, n, stream=0):
    """n uniform 2-bit codes (uint8)."""
    words = rand_u64(seed, (n + 31) // 32, stream)
    sh = (np.arange(32, dtype=np.uint64) * np.uint64(2))[None, :]
    return ((words[:, None] >> sh) & np.uint64(3)).astype(np.uint8).reshape(-1)[:n]


def codes_to_ascii(codes):
    return CODE2BASE[codes].tobytes()


def kmers_of(codes, k):
    """forward k-mer value at every start position of a code array (uint64)."""
    codes = np.asarray(codes, dtype=np.uint64)
    n = codes.size - k + 1
    if n <= 0:
        return np.zeros(0, dtype=np.uint64)
    v = np.zeros(n, dtype=np.uint64)
    for j in range(k):
        v = (v << np.uint64(2)) | codes[j:j + n]
    return v


def revcomp(x, k):
    x = np.asarray(x, dtype=np.uint64)
    r = x.copy()
    for sh, m in ((2, 0x3333333333333333), (4, 0x0F0F0F0F0F0F0F0F), (8, 0x00FF00FF00FF00FF),
                  (16, 0x0000FFFF0000FFFF)):
        m = np.uint64(m)
        r = ((r >> np.uint64(sh)) & m) | ((r & m) << np.uint64(sh))
    r = (r >> np.uint64(32)) | (r << np.uint64(32))
    return (~r) >> np.uint64(64 - 2 * k)


def canonical(x, k):
    x = np.asarray(x, dtype=np.uint64)
    return np.minimum(x, revcomp(x, k))


def pack_uniform(codes2d):
    """Pack reads that are plain ACGT, all of one length L >= k, into the batch format:
    one part per read = [L][ceil(L/8) containers].  Returns (reads_ptr u32, containers u16)."""
    codes2d = np.asarray(codes2d, dtype=np.uint16)
    n, L = codes2d.shape
    nc = (L + 7) // 8
    pad = np.zeros((n, nc * 8), dtype=np.uint16)
    pad[:, :L] = codes2d
    sh = (np.uint16(14) - np.arange(8, dtype=np.uint16) * np.uint16(2))[None, None, :]
    con = (pad.reshape(n, nc, 8) << sh).sum(axis=2, dtype=np.uint32).astype(np.uint16)
    out = np.empty((n, nc + 1), dtype=np.uint16)
    out[:, 0] = L
    out[:, 1:] = con
    ptr = (np.arange(n + 1, dtype=np.uint64) * np.uint64(nc + 1)).astype(np.uint32)
    return ptr, out.reshape(-1)


# --------------------------------------------------------------------------------
# databases
# --------------------------------------------------------------------------------
def db_from_kmers(canon, labels, htsize, wide=False):
    """(canonical k-mers, labels) -> the three on-disk arrays (.sz u8, .ky u32, .lb u16),
    buckets ascending, quotients ascending inside a bucket
    (reference src/hashTable_hh.hh:473-546)."""
    canon = np.asarray(canon, dtype=np.uint64)
    labels = np.asarray(labels, dtype=np.uint16)
    r = canon % np.uint64(htsize)
    q = canon // np.uint64(htsize)
    order = np.lexsort((q, r))
    r, q, labels = r[order], q[order], labels[order]
    cnt = np.bincount(r.astype(np.int64), minlength=htsize)
    if cnt.max(initial=0) > 255:
        raise ValueError("bucket larger than 255")
    kmax = int(q.max()) if q.size else 0
    if kmax >= 0xFFFFFFFF or wide:       # the reference's 8-byte-key regime (main.cc:277-286)
        return cnt.astype(np.uint8), q.astype(np.uint64), labels
    return cnt.astype(np.uint8), q.astype(np.uint32), labels


def discriminative(kmers_fwd, targets, k):
    """canonical k-mers that occur in exactly one target (reference RemoveCommon,
    src/HashTableStorage_hh.hh:229-280), as (canon, label) sorted by value."""
    c = canonical(kmers_fwd, k)
    t = np.asarray(targets, dtype=np.uint16)
    order = np.lexsort((t, c))
    c, t = c[order], t[order]
    first = np.ones(c.size, dtype=bool)
    first[1:] = c[1:] != c[:-1]
    starts = np.flatnonzero(first)
    ends = np.append(starts[1:], c.size)
    same = t[ends - 1] == t[starts]          # sorted by target inside a run
    return c[starts][same], t[starts][same]


def toy_genomes(n_targets, length, seed, shared=0):
    """n_targets random genomes; the first `shared` bases of genome 1 are copied from
    genome 0 (a non-discriminative region)."""
    g = [random_codes(seed + i, length) for i in range(n_targets)]
    if shared and n_targets > 1:
        g[1] = g[1].copy()
        g[1][:shared] = g[0][:shared]
    return g


def genome_db(genomes, k, htsize, wide=False):
    km = np.concatenate([kmers_of(g, k) for g in genomes])
    tg = np.concatenate([np.full(max(g.size - k + 1, 0), i, dtype=np.uint16) for i, g in enumerate(genomes)])
    canon, lab = discriminative(km, tg, k)
    return db_from_kmers(canon, lab, htsize, wide=wide)


def random_db(seed, htsize, n_keys, n_targets, k):
    """n_keys distinct uniform canonical-or-not k-mers with uniform labels."""
    kmax = np.uint64((1 << (2 * k)) - 1)
    x = rand_u64(seed, int(n_keys * 1.05) + 16) & kmax
    x = np.unique(x)[:n_keys]
    lab = (rand_u64(seed, x.size, stream=1) % np.uint64(n_targets)).astype(np.uint16)
    return db_from_kmers(x, lab, htsize)


# --------------------------------------------------------------------------------
# reads
# --------------------------------------------------------------------------------
def sample_reads(genomes, n, length, seed, sub_rate=0.01):
    """n reads of `length` bases drawn uniformly from the genomes, with substitutions."""
    rnd = rand_u64(seed, n, stream=0)
    gi = (rnd % np.uint64(len(genomes))).astype(np.int64)
    out = np.empty((n, length), dtype=np.uint8)
    pos_r = rand_u64(seed, n, stream=1)
    for i in range(n):
        g = genomes[gi[i]]
        p = int(pos_r[i] % np.uint64(g.size - length + 1))
        out[i] = g[p:p + length]
    if sub_rate > 0:
        u = rand_u64(seed, n * length, stream=2)
        mut = (u % np.uint64(1000000)) < np.uint64(int(sub_rate * 1000000))
        delta = ((u >> np.uint64(32)) % np.uint64(3) + np.uint64(1)).astype(np.uint8)
        flat = out.reshape(-1)
        flat[mut] = (flat[mut] + delta[mut]) & 3
    return out, gi


def fasta_text(names, seqs, width=0):
    parts = []
    for nm, s in zip(names, seqs):
        parts.append(b">" + nm + b"\n")
        if width and len(s) > width:
            for i in range(0, len(s), width):
                parts.append(s[i:i + width] + b"\n")
        else:
            parts.append(s + b"\n")
    return b"".join(parts)


def fastq_text(names, seqs):
    return b"".join(b"@" + nm + b"\n" + s + b"\n+\n" + b"I" * len(s) + b"\n" for nm, s in zip(names, seqs))
