"""Shared builders for the parity tests (small synthetic databases + read batches)."""
import numpy as np

from jn_cuclark_amd import synth


def small_db(seed=5, k=21, htsize=1000003, n_targets=6, glen=4000, shared=300):
    genomes = synth.toy_genomes(n_targets, glen, seed, shared=shared)
    sz, ky, lb = synth.genome_db(genomes, k, htsize)
    return genomes, sz, ky, lb


def mixed_fasta(genomes, k, seed=9, n=300, length=150):
    """FASTA text exercising the packer: sampled reads with substitutions, random
    reads, reads with N (two parts), lower case, U, reads shorter than k, a part
    shorter than k between two long ones, multi-line records, an all-N read."""
    codes, _ = synth.sample_reads(genomes, n, length, seed)
    names, seqs = [], []
    rnd = synth.rand_u64(seed, n, stream=9)
    for i in range(n):
        s = bytearray(synth.codes_to_ascii(codes[i]))
        kind = int(rnd[i] % np.uint64(12))
        if kind == 0:                       # one N in the middle -> two parts
            s[int(rnd[i] >> np.uint64(8)) % length] = ord("N")
        elif kind == 1:                     # lower case
            s = bytearray(bytes(s).lower())
        elif kind == 2:                     # RNA
            s = bytearray(bytes(s).replace(b"T", b"U"))
        elif kind == 3:                     # shorter than k
            s = s[: k - 1 - (i % 5)]
        elif kind == 4:                     # short part between two long parts
            s[40] = ord("N"); s[40 + 1 + (i % (k - 1))] = ord("N")
        elif kind == 5:                     # short part at the end
            s[length - 1 - (i % (k - 1))] = ord("N")
        elif kind == 6:                     # short part at the start
            s[i % (k - 1)] = ord("N")
        elif kind == 7:                     # uniform random read (no hits expected)
            s = bytearray(synth.codes_to_ascii(synth.random_codes(seed * 1000 + i, length)))
        elif kind == 8 and i % 3 == 0:      # all N
            s = bytearray(b"N" * length)
        elif kind == 9:                     # other IUPAC symbols end parts too
            s[75] = ord("R"); s[76] = ord("-")
        names.append(("read%d_%d some description" % (i, kind)).encode())
        seqs.append(bytes(s))
    return names, seqs


def pack_with_oracle(oracle, text, k):
    ns, ne, sp, ep, ln = oracle.index_reads(text)
    rp, con = oracle.pack_reads(text, sp, ep, ln, k)
    return (ns, ne, sp, ep, ln), rp, con


def bgzf_compress(data, block=65280, eof_block=True):
    """BGZF as bgzip / htslib write it (SAM spec 4.1): gzip members of at most 64 KB of text, each with a "BC" extra
    field that holds the member's length minus one, and an empty member at the end"""
    import struct
    import zlib
    out = bytearray()
    chunks = [data[i:i + block] for i in range(0, len(data), block)] + ([b""] if eof_block else [])
    for c in chunks:
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        d = co.compress(c) + co.flush()
        out += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(d) + 25) + d
        out += struct.pack("<II", zlib.crc32(c), len(c))
    return bytes(out)
