"""configs[3]/[4] on ONE card: a table that does not fit one GPU (default ~32e9 k-mers, 8192 targets), cut into N
parts by minimizer exactly as the GPUs of a sharded job hold it (mc_index_begin(part, N): every part streams the
WHOLE generated table through the build kernels, twice, and keeps its share at the fill mc_index_plan gives a card
of this size), the parts played in turn against the same reads, ONE k-way merge + top-2 over the N row sets.
Checked two ways, neither of which needs the table in one piece:
  * ground truth: a read sampled from genome g is assigned to target g (as tests/test_gpu_fullsize.py);
  * exactly, on a sample of reads: every k-mer of the read is looked up in a numpy model of the table -- the
    generator's CPU twin regenerates the k-mer's bucket (mcs_bucket_host), the genome k-mers are searched in their
    sorted list -- and the per-target sums and top-2 must equal the merged result row for row.
Reference behaviour: src/CuClarkDB.cu:516-559 (parts sized by memory), :842-851 (every part sees every read),
:909-928 + :963-968 (merge, then top-2).
    python tools/shard_rehearsal.py [--lam 19.9] [--parts 8] [--reads 200000] [--sample 2000]      (GPU box)"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from jn_cuclark_amd import CuClarkDB, synth, synth_gpu, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--lam", type=float, default=19.9)
ap.add_argument("--parts", type=int, default=8)
ap.add_argument("--reads", type=int, default=200_000)
ap.add_argument("--rate-reads", type=int, default=2_000_000)
ap.add_argument("--sample", type=int, default=2000)
ap.add_argument("--bg-reads", type=int, default=500, help="extra reads planted around background k-mers of the table")
ap.add_argument("--targets", type=int, default=8192)
ap.add_argument("--ranges", type=int, default=32)
ap.add_argument("--pairs", action="store_true",
                help="configs[4]: 2 x 250 bp pairs (one read of two parts, src/file.cc:250) instead of 150 bp reads: mate 1 a window of "
                     "a genome, mate 2 the reverse complement of the window 400 bases on, both with 1 %% substitutions")
ap.add_argument("--only-parts", type=str, default="", help="comma list: build and time only these parts (no merge check)")
a = ap.parse_args()

K, HT, GLEN, MAXHITS, SEED = 31, 1610612741, 100_000, 15, 41
dev = torch.device("cuda", 0)
t_all = time.time()
genomes = synth_gpu.make_genomes(a.targets, GLEN, seed=SEED, device=dev)
app = synth_gpu.genome_kmers_by_bucket(genomes, K, HT)
ranges = [(HT * j // a.ranges, HT * (j + 1) // a.ranges) for j in range(a.ranges)]
n_keys = sum(synth_gpu.build_db(dev, SEED, K, HT, a.targets, a.lam, appended=app, shard=r, count_only=True) for r in ranges)
total_hbm = torch.cuda.get_device_properties(0).total_memory
plan = _lib.index_plan(n_keys, a.parts, total_hbm)
print("table: %.2fe9 k-mers of %d targets (HTSIZE %d, k %d); one card holds at most ~12e9; plan for %d parts on %.0f GB cards: "
      "%.1f k-mers per line, %.3fe9 lines and %.1f GB per part, smallest part count %d"
      % (n_keys / 1e9, a.targets, HT, K, a.parts, total_hbm / 1e9, plan["fill"], plan["lines_per_part"] / 1e9,
         plan["bytes_per_part"] / 1e9, plan["min_parts"]), flush=True)
assert plan["fits"] == 1


def chunks():
    for b0, b1 in ranges:
        d_sz, d_keys, d_labels = synth_gpu.build_db(dev, SEED, K, HT, a.targets, a.lam, appended=app, shard=(b0, b1))
        yield d_sz, d_keys, d_labels, b0, b1
        del d_sz, d_keys, d_labels


n_mix = a.reads
PER_READ = 20           # u16 words of a packed read: [150][19 containers]
KMERS = 120
if a.pairs:
    a.bg_reads = 0
    PER_READ, KMERS = 2 * 33, 2 * 220
    g = torch.Generator(device=dev)
    g.manual_seed(42)
    npl = n_mix // 2
    gi = torch.randint(0, a.targets, (npl,), device=dev, generator=g)
    pos = torch.randint(0, GLEN - 700, (npl,), device=dev, generator=g)
    ar = torch.arange(250, device=dev, dtype=torch.int64)[None, :]
    flat = genomes.reshape(-1)
    codes = torch.randint(0, 4, (n_mix, 2, 250), dtype=torch.uint8, device=dev, generator=g)
    m1 = flat[(gi * GLEN + pos)[:, None] + ar]
    m2 = 3 - flat[(gi * GLEN + pos + 400)[:, None] + ar].flip(1)                   # the other strand
    for j, w in enumerate((m1, m2)):
        mut = torch.rand((npl, 250), device=dev, generator=g) < 0.01
        delta = torch.randint(1, 4, (npl, 250), dtype=torch.uint8, device=dev, generator=g)
        codes[:npl, j] = torch.where(mut, (w + delta) & 3, w)
    pad = torch.zeros((n_mix, 2, 256), dtype=torch.int32, device=dev)
    pad[:, :, :250] = codes
    sh = (14 - 2 * torch.arange(8, device=dev, dtype=torch.int32))[None, None, None, :]
    cw = (pad.reshape(n_mix, 2, 32, 8) << sh).sum(dim=3)
    out = torch.empty((n_mix, 2, 33), dtype=torch.int16, device=dev)
    out[:, :, 0] = 250
    out[:, :, 1:] = cw.to(torch.int16)
    rp, con, truth = None, out.reshape(-1), gi
    codes_all = codes
else:
    rp, con, truth = synth_gpu.make_reads(genomes, n_mix, 150, seed=42, return_truth=True)
# + reads around BACKGROUND k-mers of the table (a uniform random read never hits one of 32e9 out of 4^31): a stored
# k-mer from the generator's CPU twin, where its stored value is the canonical form, inside 150 random bases
rng = np.random.default_rng(7)
bg_codes = rng.integers(0, 4, size=(a.bg_reads, 150), dtype=np.uint8)
n_planted_bg = 0
for i in range(a.bg_reads):
    while True:
        b = int(rng.integers(0, HT))
        bk, _ = synth_gpu.bucket_host(SEED, K, HT, a.targets, a.lam, b)
        if bk.size == 0:
            continue
        c = np.uint64(int(bk[int(rng.integers(0, bk.size))]) * HT + b)
        if int(c) < (1 << (2 * K)) and synth.canonical(np.array([c]), K)[0] == c:
            break
    at = int(rng.integers(0, 150 - K + 1))
    kc = np.array([(int(c) >> (2 * (K - 1 - j))) & 3 for j in range(K)], dtype=np.uint8)
    bg_codes[i, at:at + K] = kc if rng.integers(0, 2) else (3 - kc[::-1])       # either strand
    n_planted_bg += 1
if a.bg_reads:
    _, con_bg = synth.pack_uniform(bg_codes)
    con = torch.cat([con, torch.from_numpy(con_bg.view(np.int16)).to(dev)])
n = n_mix + a.bg_reads
rp = (torch.arange(n + 1, device=dev, dtype=torch.int64) * PER_READ).to(torch.int32)
rp2, con2 = synth_gpu.make_reads(genomes, a.rate_reads, 150, seed=43)
st = torch.cuda.current_stream().cuda_stream
row_len = 2 * MAXHITS + 2
parts, owned = [], 0
which = [int(x) for x in a.only_parts.split(",")] if a.only_parts else list(range(a.parts))
for p in which:
    torch.cuda.empty_cache()
    db = CuClarkDB(k=K, numBatches=1, numTargets=a.targets, device=0, htsize=HT, maxhits=MAXHITS)
    t0 = time.time()
    db.read_chunks(chunks, n_keys, part=p, n_parts=a.parts, device=True)
    build_s = time.time() - t0
    info = db.db_info()
    assert info["part"] == p and info["n_parts"] == a.parts and info["n_keys"] == n_keys
    assert info["line_end"] - info["line_begin"] == plan["lines_per_part"]
    owned += info["n_keys_owned"]
    rows = torch.zeros((n, row_len), dtype=torch.int16, device=dev)
    db.query_device(rp, con, rows_t=rows, stream=st)
    rows2 = torch.zeros((a.rate_reads, row_len), dtype=torch.int16, device=dev)
    db.query_device(rp2, con2, rows_t=rows2, stream=st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        db.query_device(rp2, con2, rows_t=rows2, stream=st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("part %d/%d: %.3fe9 k-mers owned (%.2f %% of the table), %.3fe9 lines + %.3fe9 extra = %.1f GB, %.2f %% of the lines overflow, "
          "largest line %d; streamed + built in %.1f s; %.3f ms per %d reads = %.0f Mreads/s (rows out)"
          % (p, a.parts, info["n_keys_owned"] / 1e9, 100.0 * info["n_keys_owned"] / n_keys, (info["line_end"] - info["line_begin"]) / 1e9,
             info["n_extra_lines"] / 1e9, info["device_bytes"] / 1e9,
             100.0 * info["n_lines_overflowing"] / max(1, info["line_end"] - info["line_begin"]), info["largest_line"],
             build_s, ms, a.rate_reads, a.rate_reads / ms / 1e3), flush=True)
    parts.append(rows)
    del rows2
    db.close()
    del db

if a.only_parts:
    sys.exit(0)
assert owned == n_keys, (owned, n_keys)
torch.cuda.empty_cache()
fin_t = torch.zeros((n, 5), dtype=torch.int16, device=dev)
with CuClarkDB(k=K, numBatches=1, numTargets=a.targets, device=0, htsize=HT, maxhits=MAXHITS) as db:
    db.merge_result_device(parts, n, final_t=fin_t, stream=st)
    torch.cuda.synchronize()
fin = fin_t.cpu().numpy().view(np.uint16)
assert all(int((p_[:, 0] != 0).sum()) > n // 40 for p_ in parts), "a part that contributes nothing"

# ---- ground truth ---------------------------------------------------------------------------------------------
npl = truth.numel()
tr = truth.cpu().numpy()
ok = fin[:npl, 1] == tr + 1
assert ok.mean() > 0.995, ok.mean()
assert (fin[:npl, 1][~ok] != 0).mean() < 0.01 if (~ok).any() else True
assert 0.60 < fin[:npl, 0].mean() / KMERS < 0.85
assert np.all(fin[:, 0] <= KMERS) and np.all(fin[:, 2] <= fin[:, 0]) and np.all(fin[:, 4] <= fin[:, 2])
rnd_hits = int(fin[npl:n_mix, 0].astype(np.int64).sum())
bg_hit = int((fin[n_mix:, 0] >= 1).sum())
assert bg_hit == a.bg_reads, "a read planted around a stored background k-mer found nothing"
print("ground truth: %d genome-sampled reads, %.4f assigned to their genome, mean hits %.1f of %d; %d random reads with %d hits in all; "
      "%d of %d reads around a stored background k-mer hit"
      % (npl, ok.mean(), fin[:npl, 0].mean(), KMERS, n_mix - npl, rnd_hits, bg_hit, a.bg_reads), flush=True)

# ---- exact, on a sample: a numpy model of the table ------------------------------------------------------------
m = a.sample // 2
idx = np.concatenate([np.arange(m), np.arange(n_mix - m, n_mix), np.arange(n_mix, n)])
if a.pairs:
    parts_of = [[codes_all[int(i), 0].cpu().numpy(), codes_all[int(i), 1].cpu().numpy()] for i in idx]
else:
    con_h = con.cpu().numpy().view(np.uint16).reshape(n, -1)[idx]
    codes = np.zeros((idx.size, 152), dtype=np.uint8)
    for j in range(8):
        codes[:, j::8][:, :19] = ((con_h[:, 1:] >> (14 - 2 * j)) & 3).astype(np.uint8)
    codes = codes[:, :150]
    parts_of = [[codes[i]] for i in range(idx.size)]
g_r, g_q, g_l = app
expected = np.zeros((idx.size, 5), dtype=np.uint16)
n_bg = n_gen = 0
for i in range(idx.size):
    c = np.concatenate([synth.canonical(synth.kmers_of(pc, K), K) for pc in parts_of[i]])
    r, q = c % np.uint64(HT), c // np.uint64(HT)
    labs = []
    # genome k-mers: the sorted (r, q) list on the device
    rt = torch.from_numpy(r.astype(np.int64)).to(dev)
    lo, hi = torch.searchsorted(g_r, rt).cpu().numpy(), torch.searchsorted(g_r, rt, right=True).cpu().numpy()
    for j in range(c.size):
        hit = None
        bk, bl = synth_gpu.bucket_host(SEED, K, HT, a.targets, a.lam, int(r[j]))
        w = np.flatnonzero(bk == np.uint32(q[j]))
        if w.size:
            hit = int(bl[w[0]]); n_bg += 1
        if hi[j] > lo[j]:
            qq = g_q[int(lo[j]):int(hi[j])].cpu().numpy()
            w = np.flatnonzero(qq == int(q[j]))
            if w.size:
                assert hit is None
                hit = int(g_l[int(lo[j]) + int(w[0])].item()); n_gen += 1
        if hit is not None:
            labs.append(hit)
    if labs:
        t, cnt = np.unique(np.array(labs), return_counts=True)          # ascending target ids
        b = int(np.argmax(cnt))                                         # first maximum = smallest id (strict '>')
        expected[i, 0], expected[i, 1], expected[i, 2] = cnt.sum(), t[b] + 1, cnt[b]
        rest = cnt.copy(); rest[b] = 0
        if rest.max() > 0:
            s2 = int(np.argmax(rest))
            expected[i, 3], expected[i, 4] = t[s2] + 1, rest[s2]
assert np.array_equal(expected, fin[idx]), "merged result differs from the numpy model of the table"
print("exact: %d reads (%d k-mers looked up in a numpy model of the table: %d background hits, %d genome hits) == merged result of the %d parts, row for row"
      % (idx.size, idx.size * KMERS, n_bg, n_gen, a.parts), flush=True)
print("done in %.0f s" % (time.time() - t_all), flush=True)
