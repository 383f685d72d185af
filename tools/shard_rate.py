"""Rate of the query kernel on ONE part of a table that is spread over N GPUs, measured on one GPU: the
headline table, part 0 of N loaded, every read of the batch looked up against it (sparse rows out), as each
GPU of an N-GPU sharded job does.  Line-range parts (mc_load_db_part / mc_index_*: the default of mc_group
and of bench.py --mode shard) and, for comparison, the reference's bucket ranges.
`genomes`: the genome-shaped table (bench.py --db genomes) instead, as line-range parts -- with MC_INDEX=skm the parts hold
super-k-mer records and a part matches only the runs it owns (lanes are runs there), with MC_INDEX=minimizer the 12-slot lines.
    [MC_INDEX=skm|minimizer] python tools/shard_rate.py [lines|buckets|genomes] [N ...]      (run on the GPU box)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from jn_cuclark_amd import CuClarkDB, synth_gpu
from jn_cuclark_amd.dist import shard_range

K, HT, T, LAM, GLEN, MAXHITS = 31, 1610612741, 4096, 3.75, 100_000, 15
args = sys.argv[1:]
kind = "lines"
if args and args[0] in ("lines", "buckets", "genomes"):
    kind = args.pop(0)
dev = torch.device("cuda", 0)
host = None
if kind == "genomes":
    genomes = synth_gpu.make_structured_genomes(T, 1_500_000, seed=31, device=dev)
    chunks, n_keys_g = synth_gpu.build_genome_db(genomes, K, HT)
    host = [(c[0].cpu().numpy(), c[1].cpu().numpy(), c[2].cpu().numpy(), c[3], c[4]) for c in chunks]
    del chunks
    torch.cuda.empty_cache()
else:
    genomes = synth_gpu.make_genomes(T, GLEN, seed=31, device=dev)
n_reads = 10_000_000
rp, con = synth_gpu.make_reads(genomes, n_reads, 150, seed=32)
rows = torch.zeros((n_reads, 2 * MAXHITS + 2), dtype=torch.int16, device=dev)
stream = torch.cuda.current_stream().cuda_stream
raw = synth_gpu.build_db(dev, 31, K, HT, T, LAM, genomes=genomes) if kind == "lines" else None
for n in [int(a) for a in args] or [1, 2, 8]:
    db = CuClarkDB(k=K, numBatches=1, numTargets=T, device=0, htsize=HT, maxhits=MAXHITS)
    if kind == "genomes":
        t0 = time.time()
        db.read_chunks(lambda: host, n_keys_g, part=0, n_parts=n, device=False)
        build_s = time.time() - t0
    elif kind == "lines":
        t0 = time.time()
        db.read_chunks(lambda: [(raw[0], raw[1], raw[2], 0, HT)], int(raw[1].numel()), part=0, n_parts=n, device=True)
        build_s = time.time() - t0
    else:
        shard = shard_range(HT, 0, n)
        d_sz, d_keys, d_labels = synth_gpu.build_db(dev, 31, K, HT, T, LAM, genomes=genomes, shard=shard)
        t0 = time.time()
        db.read_device(d_sz, d_keys, d_labels, shard=shard)
        build_s = time.time() - t0
        del d_sz, d_keys, d_labels
    for _ in range(2):
        db.query_device(rp, con, None, rows, stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        db.query_device(rp, con, None, rows, stream)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    info = db.db_info()
    print("%s [%s] part 1/%d of the table (%.1f GB index, %.2fe9 k-mers owned, built in %.1f s): %.2f ms per 10 M reads = %.0f Mreads/s per GPU, rows out"
          % (kind, {1: "minimizer lines", 2: "super-k-mer records"}.get(info["index_kind"], "bucket lines"), n, info["device_bytes"] / 1e9, info["n_keys_owned"] / 1e9, build_s, ms, n_reads / ms / 1e3), flush=True)
    db.close()
    torch.cuda.empty_cache()
