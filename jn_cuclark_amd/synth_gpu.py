"""On-GPU synthetic workloads for bench.py.

Plumbing only (torch for device memory and bulk integer ops, libmcsynth.so for the
per-bucket generator): a bacteria-scale table is built in HBM in the array form of the
on-disk format and handed to ``CuClarkDB.read_device`` -- the same re-layout path the
file loader uses.  Reads are sampled from the synthetic genomes (with substitutions), so
they hit runs of overlapping k-mers like real reads, mixed with uniform random reads.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SYN = None


def _syn():
    global _SYN
    if _SYN is None:
        path = os.path.join(_HERE, "libmcsynth.so")
        if not os.path.exists(path):
            raise ImportError("%s missing: run `make -C jn_cuclark_amd/csrc`" % path)
        lib = C.CDLL(path)
        vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
        lib.mcs_last_error.restype = C.c_char_p
        lib.mcs_counts_device.restype = C.c_int
        lib.mcs_counts_device.argtypes = [u64, u32, u64, u32, C.c_double, u64, u64, vp, u64, vp, C.POINTER(u64), vp]
        lib.mcs_fill_device.restype = C.c_int
        lib.mcs_fill_device.argtypes = [u64, u32, u64, u32, C.c_double, u64, u64, vp, vp, vp, vp, u64, vp, vp, vp, vp, vp]
        lib.mcs_bucket_host.restype = C.c_int
        lib.mcs_bucket_host.argtypes = [u64, u32, u64, u32, C.c_double, u64, C.POINTER(u32), vp, vp]
        _SYN = lib
    return _SYN


def _chk(rc):
    if rc != 0:
        raise RuntimeError("libmcsynth: %s" % _syn().mcs_last_error().decode())


def _s64(x):
    """python int (u64 bit pattern) -> signed int64 value"""
    return x - (1 << 64) if x >= (1 << 63) else x


def revcomp_t(x, k):
    """torch int64 twin of the reverse complement (reference src/CuClarkDB.cu:1196-1203)."""
    r = x
    for sh, m in ((2, 0x3333333333333333), (4, 0x0F0F0F0F0F0F0F0F), (8, 0x00FF00FF00FF00FF),
                  (16, 0x0000FFFF0000FFFF)):
        m = _s64(m)
        r = ((r >> sh) & m) | ((r & m) << sh)
    r = ((r >> 32) & 0xFFFFFFFF) | (r << 32)
    s = 64 - 2 * k
    return ((~r) >> s) & ((1 << (64 - s)) - 1) if s > 0 else ~r


def kmers_t(codes, k):
    """forward k-mers of every window of a [G, L] uint8 code tensor -> int64 [G, L-k+1] (k <= 31)"""
    n = codes.shape[1] - k + 1
    v = torch.zeros((codes.shape[0], n), dtype=torch.int64, device=codes.device)
    for j in range(k):
        v = (v << 2) | codes[:, j:j + n].to(torch.int64)
    return v


def make_genomes(n_targets, length, seed, device):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return torch.randint(0, 4, (n_targets, length), dtype=torch.uint8, device=device, generator=g)


def genome_kmers_by_bucket(genomes, k, htsize):
    """(r, q, label) of every k-mer of the genomes, sorted by bucket r: what build_db appends to the background,
    computed once (a caller that generates the table range by range, pass after pass, hands it back as `appended`)"""
    device = genomes.device
    km = kmers_t(genomes, k)
    lab = torch.arange(genomes.shape[0], device=device, dtype=torch.int16)[:, None].expand_as(km).reshape(-1)
    km = km.reshape(-1)
    c = torch.minimum(km, revcomp_t(km, k))
    del km
    r = c % htsize
    r, order = torch.sort(r)
    q = (c // htsize)[order]
    del c
    lab = lab[order]
    del order
    return r.contiguous(), q.contiguous(), lab.contiguous()


def build_db(device, seed, k, htsize, n_targets, lam, genomes=None, shard=None, appended=None, count_only=False):
    """Returns (d_sz uint8[nb], d_keys int32[n], d_labels int16[n]) for buckets `shard`
    (count_only: just the number of k-mers)."""
    lib = _syn()
    b0, b1 = shard if shard else (0, htsize)
    nb = b1 - b0
    stream = torch.cuda.current_stream(device).cuda_stream
    app_r = app_q = app_l = None
    n_app = 0
    if appended is not None:
        r, q, lab = appended
        lo, hi = (int(v) for v in torch.searchsorted(r, torch.tensor([b0, b1], device=device, dtype=r.dtype)).tolist())
        app_r, app_q, app_l = r[lo:hi], q[lo:hi], lab[lo:hi]          # contiguous slices
        n_app = hi - lo
    elif genomes is not None:
        km = kmers_t(genomes, k)
        lab = torch.arange(genomes.shape[0], device=device, dtype=torch.int16)[:, None].expand_as(km).reshape(-1)
        km = km.reshape(-1)
        c = torch.minimum(km, revcomp_t(km, k))
        del km
        r = c % htsize
        q = c // htsize
        del c
        if shard:
            keep = (r >= b0) & (r < b1)
            r, q, lab = r[keep], q[keep], lab[keep]
        app_r, app_q, app_l = r.contiguous(), q.contiguous(), lab.contiguous()
        n_app = app_r.numel()
    d_sz = torch.empty((nb + 3) // 4 * 4, dtype=torch.uint8, device=device)
    n_keys = C.c_uint64()
    _chk(lib.mcs_counts_device(seed, k, htsize, n_targets, lam, b0, nb,
                               app_r.data_ptr() if n_app else None, n_app,
                               d_sz.data_ptr(), C.byref(n_keys), stream))
    n = n_keys.value
    if count_only:
        return n
    d_keys = torch.empty(max(n, 1), dtype=torch.int32, device=device)
    d_labels = torch.empty(max(n, 1), dtype=torch.int16, device=device)
    d_off = torch.empty(nb, dtype=torch.int64, device=device)
    d_cur = torch.empty((nb + 3) // 4 * 4, dtype=torch.uint8, device=device)
    _chk(lib.mcs_fill_device(seed, k, htsize, n_targets, lam, b0, nb, d_sz.data_ptr(),
                             app_r.data_ptr() if n_app else None, app_q.data_ptr() if n_app else None,
                             app_l.data_ptr() if n_app else None, n_app,
                             d_off.data_ptr(), d_cur.data_ptr(), d_keys.data_ptr(), d_labels.data_ptr(), stream))
    torch.cuda.synchronize(device)
    del d_off, d_cur, app_r, app_q, app_l
    return d_sz[:nb], d_keys[:n], d_labels[:n]


def bucket_host(seed, k, htsize, n_targets, lam, bucket):
    """CPU twin of the background generator: (keys u32[], labels u16[]) of one bucket."""
    import numpy as np
    cnt = C.c_uint32()
    keys = np.zeros(64, dtype=np.uint32)
    labs = np.zeros(64, dtype=np.uint16)
    _chk(_syn().mcs_bucket_host(seed, k, htsize, n_targets, lam, bucket, C.byref(cnt),
                                keys.ctypes.data, labs.ctypes.data))
    return keys[:cnt.value].copy(), labs[:cnt.value].copy()


def make_reads(genomes, n_reads, length, seed, planted_frac=0.5, sub_rate=0.01, chunk=1 << 20,
               return_truth=False):
    """n_reads packed reads of `length` bases: the first planted_frac are windows of the
    genomes with substitutions, the rest uniform random.  Returns (reads_ptr int32[n+1],
    containers int16[n*(1+ceil(L/8))]) in the reference batch format
    (src/CuCLARK_hh.hh:1615-1715): one part per read = [L][containers]."""
    device = genomes.device
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    T, G = genomes.shape
    nc = (length + 7) // 8
    out = torch.empty((n_reads, nc + 1), dtype=torch.int16, device=device)
    flat = genomes.reshape(-1)
    n_planted = int(n_reads * planted_frac)
    ar = torch.arange(length, device=device, dtype=torch.int64)[None, :]
    sh = (14 - 2 * torch.arange(8, device=device, dtype=torch.int32))[None, None, :]
    truth = []
    for s in range(0, n_reads, chunk):
        e = min(n_reads, s + chunk)
        m = e - s
        n_pl = max(0, min(e, n_planted) - s)
        codes = torch.randint(0, 4, (m, length), dtype=torch.uint8, device=device, generator=g)
        if n_pl:
            gi = torch.randint(0, T, (n_pl,), device=device, generator=g)
            truth.append(gi)
            pos = torch.randint(0, G - length + 1, (n_pl,), device=device, generator=g)
            win = flat[(gi * G + pos)[:, None] + ar]
            mut = torch.rand((n_pl, length), device=device, generator=g) < sub_rate
            delta = torch.randint(1, 4, (n_pl, length), dtype=torch.uint8, device=device, generator=g)
            codes[:n_pl] = torch.where(mut, (win + delta) & 3, win)
        pad = torch.zeros((m, nc * 8), dtype=torch.int32, device=device)
        pad[:, :length] = codes
        con = (pad.reshape(m, nc, 8) << sh).sum(dim=2)
        out[s:e, 0] = length
        out[s:e, 1:] = con.to(torch.int16)       # wraps to the u16 bit pattern
    if n_reads * (nc + 1) >= 2 ** 32:
        raise ValueError("batch exceeds the 32-bit container offsets of the batch format")
    ptr = (torch.arange(n_reads + 1, device=device, dtype=torch.int64) * (nc + 1)).to(torch.int32)
    if return_truth:      # source genome of every planted read (the first n_planted reads)
        return ptr, out.reshape(-1), (torch.cat(truth) if truth else torch.zeros(0, dtype=torch.int64, device=device))
    return ptr, out.reshape(-1)


# --------------------------------------------------------------------------------------------------
# genome-shaped table: EVERY stored k-mer comes from a genome, next to its neighbours
# --------------------------------------------------------------------------------------------------
def make_structured_genomes(n_targets, length, seed, device, genus=4, divergence=0.05,
                            conserved_len=1500, conserved_div=0.03, tandem_len=2000, polya_len=500):
    """Genomes with the structure real ones have and uniform random ones lack:
      * species of one "genus" (`genus` consecutive targets) descend from a common ancestor with `divergence`
        substitutions: a fifth of their 31-mers is shared inside the genus (not discriminative);
      * one conserved block (16S-like) sits in EVERY genome with `conserved_div` substitutions per genome:
        conserved m-mers in thousands of different contexts (crowded minimizers);
      * a tandem repeat (a random 7-mer unit, `tandem_len` bases) in every genome: few distinct k-mers, each
        many times in ONE target;
      * a poly-A run in every tenth genome: one k-mer in many targets.
    Returns uint8 codes [n_targets, length]."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    n_gen = (n_targets + genus - 1) // genus
    anc = torch.randint(0, 4, (n_gen, length), dtype=torch.uint8, device=device, generator=g)
    out = anc.repeat_interleave(genus, dim=0)[:n_targets].contiguous()
    del anc
    rows = 256
    for s in range(0, n_targets, rows):
        e = min(n_targets, s + rows)
        mut = torch.rand((e - s, length), device=device, generator=g) < divergence
        delta = torch.randint(1, 4, (e - s, length), dtype=torch.uint8, device=device, generator=g)
        out[s:e] = torch.where(mut, (out[s:e] + delta) & 3, out[s:e])
    cons = torch.randint(0, 4, (conserved_len,), dtype=torch.uint8, device=device, generator=g)
    pos = torch.randint(0, length - conserved_len - tandem_len - polya_len - 64, (n_targets,), device=device, generator=g)
    mut = torch.rand((n_targets, conserved_len), device=device, generator=g) < conserved_div
    delta = torch.randint(1, 4, (n_targets, conserved_len), dtype=torch.uint8, device=device, generator=g)
    block = torch.where(mut, (cons[None, :] + delta) & 3, cons[None, :].expand(n_targets, -1))
    ar = torch.arange(conserved_len, device=device)[None, :]
    out.scatter_(1, pos[:, None] + ar, block)
    unit = torch.randint(0, 4, (n_targets, 7), dtype=torch.uint8, device=device, generator=g)
    tr = unit.repeat(1, tandem_len // 7 + 1)[:, :tandem_len]
    out.scatter_(1, (pos + conserved_len + 32)[:, None] + torch.arange(tandem_len, device=device)[None, :], tr)
    pa = torch.arange(0, n_targets, 10, device=device)
    out[pa[:, None], (pos[pa] + conserved_len + tandem_len + 64)[:, None] + torch.arange(polya_len, device=device)[None, :]] = 3
    return out


def build_genome_db(genomes, k, htsize, n_ranges=16, slab=256):
    """The discriminative k-mers of `genomes` (canonical k-mers that occur in exactly one target, once per
    target: reference RemoveCommon, src/HashTableStorage_hh.hh:229-280) as bucket-order chunks
    [(d_sz uint8[nb], d_keys int32[n], d_labels int16[n], b0, b1), ...] -- the array form of the on-disk
    format, one chunk per bucket range, ready for CuClarkDB.read_chunks(device=True).  k <= 31."""
    dev = genomes.device
    T, L = genomes.shape
    canon = []                                   # canonical k-mers of every genome, slab by slab (8 bytes per base)
    for s in range(0, T, slab):
        km = kmers_t(genomes[s:min(T, s + slab)], k)
        canon.append(torch.minimum(km, revcomp_t(km, k)))
        del km
    chunks = []
    total = 0
    for j in range(n_ranges):
        b0, b1 = htsize * j // n_ranges, htsize * (j + 1) // n_ranges
        cs, ls = [], []
        for i, c in enumerate(canon):
            r = c % htsize
            keep = (r >= b0) & (r < b1)
            del r
            lab = torch.arange(i * slab, i * slab + c.shape[0], device=dev, dtype=torch.int16)[:, None].expand_as(c)
            cs.append(c[keep])
            ls.append(lab[keep])
            del keep, lab
        c = torch.cat(cs)
        lab = torch.cat(ls)
        del cs, ls
        c, order = torch.sort(c)
        lab = lab[order]
        del order
        first = torch.ones(c.numel(), dtype=torch.bool, device=dev)
        first[1:] = c[1:] != c[:-1]
        run = torch.cumsum(first.to(torch.int32), 0) - 1
        start = torch.nonzero(first).squeeze(1)
        lab0 = lab[start]
        bad = torch.zeros(start.numel(), dtype=torch.int32, device=dev)
        bad.index_add_(0, run, (lab != lab0[run]).to(torch.int32))
        good = bad == 0
        c, lab = c[start][good], lab0[good]
        del first, run, start, lab0, bad, good
        comp, order = torch.sort(((c % htsize) << 32) | (c // htsize))       # bucket, then quotient: the file order
        lab = lab[order]
        del c, order
        sz = torch.bincount((comp >> 32) - b0, minlength=b1 - b0)
        if int(sz.max().item()) > 255:
            raise ValueError("a bucket holds more than 255 k-mers (the on-disk format has one byte per bucket)")
        chunks.append((sz.to(torch.uint8), (comp & 0xFFFFFFFF).to(torch.int32), lab.contiguous(), b0, b1))
        total += int(lab.numel())
        del comp, sz, lab
    return chunks, total


# --------------------------------------------------------------------------------------------------
# files: the generated table as <base>.sz/.ky/.lb, reads as FASTQ text (tools/file_e2e.py, tests/test_gpu_filesize.py)
# --------------------------------------------------------------------------------------------------
def write_db_files(base, chunks):
    """Append bucket-order chunks (d_sz, d_keys int32, d_labels int16, b0, b1) -- consecutive ranges that cover the
    table -- to <base>.sz/.ky/.lb: the reference's on-disk format (src/hashTable_hh.hh:473-546: one size byte per
    bucket, 4-byte quotients, 2-byte labels, bucket order).  Returns (n_keys, bytes written)."""
    n_keys = total = 0
    with open(base + ".sz", "wb") as fs, open(base + ".ky", "wb") as fk, open(base + ".lb", "wb") as fl:
        for d_sz, d_keys, d_labels, _b0, _b1 in chunks:
            for f, t in ((fs, d_sz), (fk, d_keys), (fl, d_labels)):
                a = t.cpu().numpy()
                a.tofile(f)
                total += a.nbytes
            n_keys += int(d_keys.numel())
    return n_keys, total


FASTQ_RECORD = 2 + 10 + 1 + 150 + 3 + 150 + 1          # "@r" + 10 digits, 150 bases, "+", 150 qualities


def write_fastq(path, genomes, n_reads, seed, planted_frac=0.5, sub_rate=0.01, slab=2_000_000):
    """n_reads x 150 bp as FASTQ text (317 bytes per record, names @r0000000000 ...), the mix of make_reads: the
    first planted_frac are windows of the genomes with substitutions, the rest uniform random.  Generated slab by
    slab on the GPU, written as it goes.  Returns the source genome of every planted read (int64 tensor, host)."""
    dev = genomes.device
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    T, G = genomes.shape
    L = 150
    flat = genomes.reshape(-1)
    n_planted = int(n_reads * planted_frac)
    ar = torch.arange(L, device=dev, dtype=torch.int64)[None, :]
    base_of = torch.tensor([ord(c) for c in "TGCA"], dtype=torch.uint8, device=dev)      # codes A=3 C=2 G=1 T=0
    pow10 = torch.tensor([10 ** (9 - j) for j in range(10)], dtype=torch.int64, device=dev)[None, :]
    truth = []
    with open(path, "wb") as f:
        for s in range(0, n_reads, slab):
            e = min(n_reads, s + slab)
            m = e - s
            n_pl = max(0, min(e, n_planted) - s)
            codes = torch.randint(0, 4, (m, L), dtype=torch.uint8, device=dev, generator=g)
            if n_pl:
                gi = torch.randint(0, T, (n_pl,), device=dev, generator=g)
                truth.append(gi.cpu())
                pos = torch.randint(0, G - L + 1, (n_pl,), device=dev, generator=g)
                win = flat[(gi * G + pos)[:, None] + ar]
                mut = torch.rand((n_pl, L), device=dev, generator=g) < sub_rate
                delta = torch.randint(1, 4, (n_pl, L), dtype=torch.uint8, device=dev, generator=g)
                codes[:n_pl] = torch.where(mut, (win + delta) & 3, win)
            rec = torch.empty((m, FASTQ_RECORD), dtype=torch.uint8, device=dev)
            rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
            idx = torch.arange(s, e, device=dev, dtype=torch.int64)[:, None]
            rec[:, 2:12] = ((idx // pow10) % 10 + 48).to(torch.uint8)
            rec[:, 12] = 10
            rec[:, 13:163] = base_of[codes.to(torch.int64)]
            rec[:, 163] = 10; rec[:, 164] = ord("+"); rec[:, 165] = 10
            rec[:, 166:316] = ord("I")
            rec[:, 316] = 10
            rec.cpu().numpy().tofile(f)
            del rec, codes
    return torch.cat(truth) if truth else torch.zeros(0, dtype=torch.int64)


def pick_dir(need_bytes, want=None):
    """a directory with `need_bytes` (+ 8 GB) free: `want`, /dev/shm, $TMPDIR, /tmp; None when there is none"""
    import shutil
    for d in ([want] if want else []) + ["/dev/shm", os.environ.get("TMPDIR") or "/tmp", "/tmp"]:
        try:
            if d and os.path.isdir(d) and shutil.disk_usage(d).free > need_bytes + (8 << 30):
                return d
        except OSError:
            pass
    return None


def host_driver_run(exe, work, k, n_targets, fastq, n_reads, threads=16, batches=32, truth=None, check=200_000, fastq2=None, timeout=600):
    """`exe -k K -T targets -D work -O fastq -R work/res` (the database files must be in `work` under the reference's
    name, src/CuCLARK_hh.hh:586-590): returns the program's own rate ("Done in Xs (N reads/min, M reads)", the
    reference's timer, src/CuCLARK_hh.hh:552-563, :1931-1939: file -> CSV, the database load is outside it), the wall
    clock of the whole process and -- with `truth` (source genome of the leading reads) -- how many of the first `check`
    CSV lines name their genome."""
    import subprocess
    import time as _t
    with open(os.path.join(work, "targets.txt"), "w") as f:
        f.write("".join("%s/g%04d.fa\tT%04d\n" % (work, i, i) for i in range(n_targets)))
    for i in range(n_targets):          # the driver checks that the target files exist (it builds from them when the database is missing)
        p = "%s/g%04d.fa" % (work, i)
        if not os.path.exists(p):
            with open(p, "w") as f:
                f.write(">g%04d\n" % i)
    for old in (os.path.join(work, "res.csv"),):          # (a result file left by an earlier run is not this run's to truncate: 1.7 GB of page cache)
        if os.path.exists(old):
            os.remove(old)
    t0 = _t.time()
    inputs = ["-P", fastq, fastq2] if fastq2 else ["-O", fastq]          # -P: paired-end mates in two files (src/main.cc:43-69)
    r = subprocess.run([exe, "-k", str(k), "-T", os.path.join(work, "targets.txt"), "-D", work] + inputs + ["-R", os.path.join(work, "res"),
                        "-n", str(threads), "-b", str(batches), "--verbose"], capture_output=True, text=True, timeout=timeout)      # TimeoutExpired: the caller's to report
    wall = _t.time() - t0
    if r.returncode != 0:
        raise RuntimeError("host driver failed: " + r.stderr[-500:])
    done = [ln for ln in r.stderr.split("\n") if "Done in" in ln]
    timing = [ln.strip() for ln in r.stderr.split("\n") if "timing" in ln]
    # "Done in 0.6s (3870967741 reads/min, 40000000 reads)": the seconds have one decimal, the rate is exact
    rpm = float(done[0].split("(")[1].split(" reads/min")[0]) if done else float("nan")
    out = {"reads": n_reads, "Mreads_per_s": round(rpm / 60e6, 2), "seconds": round(n_reads / (rpm / 60.0), 4) if rpm == rpm and rpm > 0 else None,
           "wall_s": round(wall, 2), "line": done[0].strip() if done else "", "timing": timing, "threads": threads, "batches": batches}
    n_lines = good = bad = 0
    with open(os.path.join(work, "res.csv"), "rb") as f:
        f.readline()
        for i, ln in enumerate(f):
            n_lines += 1
            if truth is not None and i < min(check, len(truth)):
                c = ln.split(b",")
                if c[0] != b"r%010d" % i:
                    raise RuntimeError("CSV line %d names %r" % (i, c[0]))
                good += c[-3] == b"T%04d" % truth[i]
                bad += c[-3] != b"T%04d" % truth[i] and c[-3] != b"NA"
    out["csv_lines"] = n_lines
    if truth is not None:
        out["checked"] = min(check, len(truth))
        out["assigned_to_their_genome"] = int(good)
        out["assigned_elsewhere"] = int(bad)
    return out
