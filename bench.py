#!/usr/bin/env python3
"""bench.py -- Mreads/s classified on the headline configuration of BASELINE.json.

    python bench.py --gpus 1 --steps K --warmup W          (driver: torchrun for N > 1)

Workload (config.workload, SURVEY.md section 8d config 3 = BASELINE configs[2]):
cuCLARK full-size table (HTSIZE 1610612741 buckets), k = 31, ~6.4e9 k-mers of 4096
targets resident in HBM (synthetic: background k-mers + the k-mers of 4096 synthetic
genomes), 10 M x 150 bp reads per step (half sampled from the genomes with 1 %
substitutions, half uniform random).  A "step" is one pass of the hot path over that
batch: packed reads (already in HBM) -> k-mers -> canonical -> hash -> bucket probe ->
per-target hit counts -> (best, second) per read, through the C ABI (mc_query_device).

One JSON line on stdout (rank 0).  `value` is the whole-job rate with inputs resident
in HBM.  `roofline` is for the query kernel (HBM-transaction bound, no MFMA);
`cpu_baseline` is the oracle's restatement timed on the host cores (rank 0, N=1 only),
on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# before the HIP runtime starts (as bin/cuCLARK does): the copy-in / compute / copy-out queues of the streamed path must
# not share one of the runtime's default 4 hardware queues (csrc/mc_api.hip, mc_open); no effect on the kernel rate
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s HBM3E (spec)
READ_LEN = 150
K = 31
HTSIZE = 1610612741            # reference src/parameters.hh:37
MAXHITS = 15                   # reference src/parameters.hh:44


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per step per GPU")
    ap.add_argument("--lam", type=float, default=3.75, help="background k-mers per bucket (Poisson mean)")
    ap.add_argument("--targets", type=int, default=4096)
    ap.add_argument("--genome-len", type=int, default=100_000)
    ap.add_argument("--htsize", type=int, default=HTSIZE)
    ap.add_argument("--k", type=int, default=K)
    ap.add_argument("--mode", choices=["replica", "shard"], default="replica",
                    help="N>1: replica = DB on every GPU, reads split; shard = DB split by bucket "
                         "range, every GPU sees every read, sparse rows exchanged over RCCL")
    ap.add_argument("--shards", type=int, default=0,
                    help="--mode shard: parts the table is cut into (0 = one group of all ranks); the ranks form "
                         "world // shards groups that each hold the whole table and classify their own batches "
                         "(jn_cuclark_amd.dist.shard_groups), rows are exchanged inside a group")
    ap.add_argument("--config", type=int, default=3, choices=[2, 3],
                    help="BASELINE.json configs[]: 3 = full table, k=31, 10M reads (default, the metric's "
                         "configuration); 2 = cuCLARK-l light table (~4 GB on disk), k=27, 1M reads")
    ap.add_argument("--db", choices=["synthetic", "genomes"], default="synthetic",
                    help="synthetic (default, the headline table): 94 %% isolated random k-mers + the k-mers of 4096 x 100 kb "
                         "genomes; genomes: EVERY k-mer from structured genomes (genera with shared sequence, a conserved "
                         "16S-like block, tandem repeats, poly-A): the shape of a real database, --genome-len bases each")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the host-to-host (PCIe-inclusive) measurement")
    ap.add_argument("--read-len", type=int, default=READ_LEN,
                    help="read length in bases (the BASELINE metric is quoted at 150; other lengths are side measurements)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-shard-check", action="store_true",
                    help="N>1 replica runs also exercise the sharded RCCL path once, untimed; skip that")
    ap.add_argument("--cpu-sample", type=int, default=0, help="reads in the CPU baseline sample (0 = auto)")
    ap.add_argument("--verify", type=int, default=20000, help="reads checked against the oracle (0 = none)")
    ap.add_argument("--no-extras", action="store_true",
                    help="N = 1 runs of the headline workload also report, as extra keys of the JSON line (never the value): "
                         "`genomes` -- the same kernel on the genome-shaped table of --db genomes -- and `e2e_host` -- the "
                         "table written as .sz/.ky/.lb files, loaded by bin/cuCLARK, a FASTQ file classified to CSV; skip them")
    ap.add_argument("--e2e-reads", type=int, default=40_000_000, help="reads in the FASTQ file of `e2e_host`")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start one rank per GPU as CHILD processes (torch.distributed.run)
        # before anything here has touched the GPU or imported torch, and leave with the launcher's status -- never an
        # exec, never a 1-GPU number under an n_gpus: N label
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        log("bench: --gpus %d without WORLD_SIZE: launching %s" % (args.gpus, " ".join(cmd[2:9])))
        raise SystemExit(subprocess.call(cmd))
    if args.config == 2:
        args.htsize, args.k, args.lam, args.targets, args.genome_len = 57777779, 27, 10.4, 2048, 14000
        if args.reads == 10_000_000:
            args.reads = 1_000_000
    global MAXHITS
    MAXHITS = 23 if args.htsize == 57777779 else 15        # reference parameters_light_hh:45 / parameters.hh:44

    import numpy as np
    import torch
    import torch.distributed as dist
    if os.environ.get("BENCH_TRACE_AFTER"):          # where does a run sit after N seconds (a rehearsal that seems to hang)
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["BENCH_TRACE_AFTER"]), exit=False)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    # one rank per GPU; BENCH_BACKEND=gloo + fewer GPUs than ranks is only for rehearsing the
    # multi-rank control flow on a one-GPU box (ranks then share the card)
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from jn_cuclark_amd import CuClarkDB, synth_gpu

    k, ht = args.k, args.htsize
    shard_mode = world > 1 and args.mode == "shard"
    # Ranks that share a card (a rehearsal of N ranks on fewer GPUs) share its HBM.  Every rank sizes its index from the
    # memory it finds free -- four ranks that each take "the whole card" for a 264 GB index is what stalled round 3's
    # 4-rank rehearsal inside the generator (gpurun_out/r3n/rep4.err) -- so the budget is divided before anything is built.
    ranks_on_dev = (world + torch.cuda.device_count() - 1) // torch.cuda.device_count()
    # sharded: S parts x G groups; rank r holds part r % S for group r // S (ranks past S * G idle)
    S, G, group, gi, part = 1, world, None, rank, 0
    if shard_mode:
        from jn_cuclark_amd.dist import shard_groups
        S = args.shards if 0 < args.shards <= world else world
        group, gi, part, G = shard_groups(S)
    active = (not shard_mode) or group is not None

    # ---- database in HBM ---------------------------------------------------------
    t0 = time.time()
    db = CuClarkDB(k=k, numBatches=1, numTargets=args.targets, device=dev_index, htsize=ht, maxhits=MAXHITS)
    raw_host = None          # the table as host arrays, for the oracle (rank 0, N = 1)
    if args.db == "genomes":
        if args.genome_len == 100_000:
            args.genome_len = 1_500_000
        genomes = synth_gpu.make_structured_genomes(args.targets, args.genome_len, seed=31, device=dev)
        chunks, n_keys = synth_gpu.build_genome_db(genomes, k, ht)
        nonempty = float(sum(int((c[0] != 0).sum().item()) for c in chunks)) / ht if rank == 0 else 0.0
        # The table goes to the host and every temporary of the generator back to the driver BEFORE the index is
        # allocated: 140 GB of lines carved out of a heap that torch has fragmented end up on small pages, and the
        # query kernel (one TLB miss per line) then runs anywhere between 630 and 910 Mreads/s on the same table.
        # A loader that starts from files (mc_load_db) allocates from a fresh heap and does not have this problem.
        host = [(c[0].cpu().numpy(), c[1].cpu().numpy(), c[2].cpu().numpy(), c[3], c[4]) for c in chunks]
        del chunks
        genomes_h = genomes.cpu()
        del genomes
        torch.cuda.empty_cache()
        if rank == 0 and world == 1 and not (args.no_cpu_baseline and args.verify == 0):
            raw_host = tuple(np.concatenate([c[i] for c in host]) for i in range(3))
        # line-range parts when sharded: every rank streams the whole table and keeps its lines
        fill_fixed = share_card(db, n_keys, S if shard_mode else 1, ranks_on_dev, torch, dev_index)
        db.read_chunks(lambda: host, n_keys, part=part if shard_mode else 0, n_parts=S if shard_mode else 1, device=False)
        del host
        genomes = genomes_h.to(dev)
        del genomes_h
    else:
        # The table is generated in 16 bucket-range chunks and fed to the index build chunk by chunk, once per build
        # pass (the generator is deterministic), exactly as a loader streams the files: neither the raw arrays of the
        # whole table (39 GB) nor anything else sits in HBM next to the index while it is built, so the index takes the
        # sparsest fill the card allows.  A first round of generation counts the k-mers (and keeps a host copy of the
        # table for the oracle when one is wanted).
        genomes = synth_gpu.make_genomes(args.targets, args.genome_len, seed=31, device=dev)
        want_host = rank == 0 and world == 1 and not (args.no_cpu_baseline and args.verify == 0)
        n_ranges = 16
        ranges = [(ht * j // n_ranges, ht * (j + 1) // n_ranges) for j in range(n_ranges)]
        host_parts, n_keys, nonzero = [], 0, 0

        def synth_chunks():
            for b0, b1 in ranges:
                d_sz, d_keys, d_labels = synth_gpu.build_db(dev, 31, k, ht, args.targets, args.lam, genomes=genomes, shard=(b0, b1))
                yield d_sz, d_keys, d_labels, b0, b1
                del d_sz, d_keys, d_labels

        for d_sz, d_keys, d_labels, b0, b1 in synth_chunks():
            n_keys += int(d_keys.numel())
            if rank == 0:
                nonzero += int((d_sz != 0).sum().item())
            if want_host:
                host_parts.append((d_sz.cpu().numpy(), d_keys.cpu().numpy(), d_labels.cpu().numpy()))
        nonempty = nonzero / ht
        torch.cuda.empty_cache()
        fill_fixed = share_card(db, n_keys, S if shard_mode else 1, ranks_on_dev, torch, dev_index)
        db.read_chunks(synth_chunks, n_keys, part=part if shard_mode else 0, n_parts=S if shard_mode else 1, device=True)
        if want_host:
            raw_host = tuple(np.concatenate([p_[i] for p_ in host_parts]) for i in range(3))
        del host_parts
    torch.cuda.empty_cache()
    if fill_fixed:
        os.environ.pop("MC_MZ_FILL", None)
    info = db.db_info()
    torch.cuda.synchronize()
    index = {1: "minimizer", 2: "skm"}.get(info["index_kind"], "lines")
    if rank == 0:
        log("db: %.2fe9 k-mers, %s index, %.1f GB in HBM, %.2f %% of the %s overflow, largest line %d k-mers, built in %.1fs"
            % (n_keys / 1e9, index, info["device_bytes"] / 1e9,
               100.0 * (info["n_lines_overflowing"] / max(1, info["line_end"] - info["line_begin"]) if index != "lines"
                        else info["n_overflow_buckets"] / ht),
               "lines" if index != "lines" else "buckets", info["largest_line"], time.time() - t0))

    # ---- reads in HBM ------------------------------------------------------------
    n_reads = args.reads
    read_seed = 32 + max(gi, 0) if shard_mode else 32 + rank      # the shards of a group see the SAME batch
    rp_t, con_t = synth_gpu.make_reads(genomes, n_reads, args.read_len, seed=read_seed)
    fin_t = torch.zeros((n_reads, 5), dtype=torch.int16, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    sharded = None
    if shard_mode:
        from jn_cuclark_amd.dist import ShardedClassifier, HipBackend
        sharded = ShardedClassifier(HipBackend(db, dev), group=group) if active else None

    def step():
        if not shard_mode:
            db.query_device(rp_t, con_t, final_t=fin_t, stream=stream)
            return
        if not active:
            return
        # every GPU: partial sparse rows of ALL reads for its line range of the index, chunk by chunk; per chunk
        # a reduce-scatter by read range over xGMI (all_to_all) + k-way merge + top-2 on the owner, on a
        # second stream under the next chunk's query
        fin, _ = sharded.classify(rp_t, con_t, n_reads)
        fin_t[: fin.shape[0]] = fin

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()
        step()
        ev[i][1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = sorted(a.elapsed_time(b) for a, b in ev)
    kern_ms_avg = sum(kern_ms) / len(kern_ms)
    if rank == 0:
        log("step times (ms, sorted): " + " ".join("%.3f" % t for t in kern_ms))
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0 and hasattr(db._lib, "mc_debug_stats"):
        # a measurement build (tools/stats_build.sh, MC_LIB_PATH): where the lookups of one step end
        import ctypes as C
        arr = (C.c_ulonglong * 16)()
        db._lib.mc_debug_stats(arr)
        step()
        torch.cuda.synchronize()
        db._lib.mc_debug_stats(arr)
        names = ["steps", "runs", "kmers_looked_up", "found_in_first_line", "missed_on_chained_line", "runs_on_overflowing_lines",
                 "steps_with_such_a_miss", "steps_that_go_to_chains", "kmers_to_chains_after_bloom", "distinct_chain_lines_by_neighbours",
                 "chain_fetch_rounds", "chain_lines_fetched", "found_in_chain"]
        log("lookup stats per read: " + ", ".join("%s %.3f" % (nm, arr[i] / n_reads) for i, nm in enumerate(names)))

    total_reads = n_reads * args.steps * (G if shard_mode else world)
    value = total_reads / elapsed / 1e6

    out = None
    if rank == 0:
        fin = fin_t.cpu().numpy().view(np.uint16)
        st = db.stats()
        kmers_per_read = args.read_len - k + 1
        hit_rate = float(fin[:, 0].astype(np.float64).mean()) / kmers_per_read if not shard_mode else float("nan")
        assigned = float((fin[:, 1] > 0).mean())
        # algorithmic bytes per read on the reference layout: 44 B in + 10 B out
        # + per k-mer 8 B of bucket offsets + 4 B per non-empty bucket + 2 B label per hit
        hr = 0.0 if hit_rate != hit_rate else hit_rate
        bytes_per_read = 54.0 + kmers_per_read * (8.0 + 4.0 * nonempty + 2.0 * hr)
        achieved = bytes_per_read * n_reads / (kern_ms_avg * 1e-3) / 1e9
        # (the k-mer length is compiled in for k = 31 and 27, csrc/mc_minimizer.hpp)
        kernel_name = ("mc::mz::mz_query_kernel<%d, %d>" % (2 if shard_mode else 0, k if k in (31, 27) else 0)) if index == "minimizer" \
            else ("mc::sk::sk_query_kernel<%d, %d>" % (2 if shard_mode else 0, k if k in (31, 27) else 0)) if index == "skm" \
            else "mc::query_kernel<%d, false>" % info["line_bytes"]
        # HBM traffic per launch comes from a rocprofv3 --pmc run of this same command (tools/prof_pmc.sh writes
        # profiles/traffic.json): counters cannot be read from inside the process.  The figure is used only if it
        # was taken with the SAME kernel sources on the same workload; otherwise null.
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if (tj.get("reads_per_launch") == n_reads and tj.get("kernel") == kernel_name and tj.get("htsize") == ht
                        and tj.get("db", "synthetic") == args.db and tj.get("read_len", 150) == args.read_len
                        and tj.get("kmers_per_line") == round(n_keys / max(1, info["n_lines"]), 2)
                        and tj.get("source_sha") == source_sha()):
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mreads/s classified, %dbp k=%d %s" % (args.read_len, k, "RefSeq-bacteria-scale DB" if ht == HTSIZE else "DB"),
            "value": round(value, 3), "unit": "Mreads/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong" if (shard_mode and G == 1) else "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {
                "workload": "cuCLARK full table HTSIZE=%d k=%d, %.2fe9 k-mers of %d targets in HBM (%s; "
                            "%s index, %d-byte lines), %d x %dbp reads per step per GPU, inputs resident in HBM"
                            % (ht, k, n_keys / 1e9, args.targets,
                               "all from structured genomes" if args.db == "genomes" else "background + genome k-mers",
                               index, info["line_bytes"], n_reads, args.read_len),
                "db": args.db,
                "index": {"kind": index, "fallback": bool(info["index_fallback"]),
                          "lines": info["line_end"] - info["line_begin"], "extra_lines": info["n_extra_lines"],
                          "kmers_per_line": round(n_keys / max(1, info["n_lines"]), 2),
                          "lines_overflowing_frac": round(info["n_lines_overflowing"] / max(1, info["line_end"] - info["line_begin"]), 5),
                          "lines_crowded": info["n_lines_crowded"], "kmers_in_hashed_chains": info["n_spilled_keys"],
                          "largest_line_kmers": info["largest_line"], "hbm_bytes": info["device_bytes"]},
                "reads_per_step": n_reads * (G if shard_mode else world), "k": k, "htsize": ht,
                "n_kmers_db": n_keys, "targets": args.targets, "maxhits": MAXHITS,
                "parallelism": ("%d parts x %d groups, all_to_all inside a group" % (S, G)) if shard_mode else ("replica%d" % world),
                "kmer_hit_rate": None if hit_rate != hit_rate else round(hit_rate, 4),
                "reads_assigned": round(assigned, 4), "reads_over_maxhits": st["reads_over_maxhits"],
            },
            "roofline": {
                "bound": "hbm", "kernel": kernel_name, "index": index,
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "kernel_ms": round(kern_ms_avg, 4), "algorithmic_bytes_per_read": round(bytes_per_read, 1),
                "kmers_per_s": round(kmers_per_read * n_reads / (kern_ms_avg * 1e-3), 1),
                "traffic_source": None if traffic is None else "profiles/traffic.json (rocprofv3 --pmc, same sources: %s)" % source_sha(),
                "hbm_traffic_GBs": None if traffic is None else round(traffic / (kern_ms_avg * 1e-3) / 1e9, 1),
            },
        }

        # ---- host-to-host: the streaming interface of the boundary (pinned buffers, three queues) ----
        if world == 1 and not args.no_pipelined:
            out["pipelined"] = pipelined_rate(db, torch, np, rp_t, con_t, n_reads, fin)

        # ---- oracle: spot check + CPU baseline (rank 0, N = 1) ----------------------
        if world == 1 and not (args.no_cpu_baseline and args.verify == 0):
            from oracle import pyoracle
            t1 = time.time()
            odb = pyoracle.OracleDB.from_arrays(ht, raw_host[0], raw_host[1].view(np.uint32), raw_host[2].view(np.uint16))
            raw_host = None
            log("oracle database on host in %.1fs" % (time.time() - t1))
            rp_h = rp_t.cpu().numpy().view(np.uint32)
            con_h = con_t.cpu().numpy().view(np.uint16)
            per_read = con_h.size // n_reads

            def sample(idx0, m):
                p = (np.arange(m + 1, dtype=np.uint64) * np.uint64(per_read)).astype(np.uint32)
                return p, con_h[idx0 * per_read:(idx0 + m) * per_read]

            if args.verify:
                for idx0 in (0, n_reads - args.verify):      # planted half and random half
                    p, c = sample(idx0, args.verify)
                    want, _ = odb.classify(k, p, c, MAXHITS)
                    if not np.array_equal(want, fin[idx0:idx0 + args.verify]):
                        raise SystemExit("bench: HIP results differ from the oracle")
                out["config"]["verified_reads_vs_oracle"] = 2 * args.verify
            if not args.no_cpu_baseline:
                cores = pyoracle.host_cores()
                pyoracle.set_num_threads(cores)
                m = args.cpu_sample or 500000 * cores
                m = min(m, n_reads // 2)
                # half planted + half random, like the full batch
                p, c1 = sample(0, m // 2)
                _, c2 = sample(n_reads - m // 2, m // 2)
                cc = np.concatenate([c1, c2])
                pp = (np.arange(2 * (m // 2) + 1, dtype=np.uint64) * np.uint64(per_read)).astype(np.uint32)
                odb.classify(k, pp[:2001], cc[:2000 * per_read], MAXHITS)     # warm
                t1 = time.perf_counter()
                want_cpu, _ = odb.classify(k, pp, cc, MAXHITS)
                dt = time.perf_counter() - t1
                # the sample's rows are parity evidence that is already paid for: the first m/2 and the last m/2 reads of the batch
                h = m // 2
                if not (np.array_equal(want_cpu[:h], fin[:h]) and np.array_equal(want_cpu[h:], fin[n_reads - h:])):
                    raise SystemExit("bench: HIP results differ from the oracle (cpu_baseline sample)")
                out["config"]["verified_reads_vs_oracle"] = max(out["config"].get("verified_reads_vs_oracle", 0), 2 * h)
                del want_cpu
                out["cpu_baseline"] = {
                    "value": round(2 * (m // 2) / dt / 1e6, 4), "unit": "Mreads/s", "cores": cores,
                    "kind": "port",
                    "sample": "%d reads of the same batch (half genome-sampled, half random) against the "
                              "same table on the host, OpenMP schedule(dynamic), %.1f s" % (2 * (m // 2), dt),
                }
                # the same on ONE core, a twentieth of the sample
                pyoracle.set_num_threads(1)
                m1 = max(2000, (m // 40) * 2)
                pp1 = pp[:m1 + 1]
                cc1 = np.concatenate([c1[:(m1 // 2) * per_read], c2[:(m1 // 2) * per_read]])
                t1 = time.perf_counter()
                odb.classify(k, pp1, cc1, MAXHITS)
                dt1 = time.perf_counter() - t1
                out["cpu_baseline_1core"] = {"value": round(m1 / dt1 / 1e6, 5), "unit": "Mreads/s", "cores": 1, "kind": "port",
                                             "sample": "%d reads of the same mix, %.1f s" % (m1, dt1)}
                pyoracle.set_num_threads(cores)
            odb.close()
        elif world == 1:
            out["cpu_baseline"] = None

    # ---- secondary measurements in the same line (N = 1, headline workload; never the value) ------------------
    if rank == 0 and world == 1 and not args.no_extras and args.db == "synthetic" and args.config == 3 and args.read_len == READ_LEN:
        db.close()
        del rp_t, con_t, fin_t
        torch.cuda.empty_cache()
        for name, fn in (("genomes", extra_genomes), ("e2e_host", extra_e2e_host)):
            t1 = time.time()
            try:
                out[name] = fn(args, torch, np, dev, dev_index, genomes)
            except (Exception, SystemExit) as e:          # the headline measurement must survive a failure here (a failed check too)
                out[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            log("%s: %.1f s" % (name, time.time() - t1))
            torch.cuda.empty_cache()

    # the ONE JSON line (the extras above run under deadlines: a child process that hangs costs its own key, not the line)
    if rank == 0:
        print(json.dumps(out), flush=True)

    watchdog = None
    shard_failed = False
    if world > 1:
        # nothing below may keep the job alive: a rank stuck in a collective (a peer died) ends the job --
        # with a NON-ZERO status (the JSON line is already out and flushed)
        import threading
        watchdog = threading.Timer(300.0, lambda: (log("bench: post-measurement phase timed out"), os._exit(3)))
        watchdog.daemon = True
        watchdog.start()
    if world > 1 and not shard_mode and not args.no_shard_check:
        db.close()
        del rp_t, con_t, fin_t
        torch.cuda.empty_cache()
        chk = shard_path_check(args, dist, torch, np, dev, dev_index, rank, world, backend)
        if rank == 0:
            print("shard_path " + json.dumps(chk), file=sys.stderr, flush=True)
        shard_failed = bool(chk.get("error")) or not chk.get("verified_equal_to_unsharded", False)

    db.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if watchdog is not None:
        watchdog.cancel()
    if shard_failed:
        raise SystemExit("bench: the sharded RCCL path failed its check (see shard_path on stderr)")


def share_card(db, n_keys, n_parts, ranks_on_dev, torch, dev_index):
    """Ranks that share a card: fix the fill (MC_MZ_FILL) to what 1/ranks of the card's HBM affords, or refuse before
    anything is built.  One rank per card (the driver's runs): nothing to do, the library sizes the index itself."""
    if ranks_on_dev <= 1 or os.environ.get("MC_MZ_FILL"):
        return False
    import ctypes as C
    from jn_cuclark_amd._lib import McIndexPlan
    total = torch.cuda.get_device_properties(dev_index).total_memory
    plan = McIndexPlan()
    if db._lib.mc_index_plan(C.c_uint64(n_keys), C.c_uint32(n_parts), C.c_uint64(total // ranks_on_dev), C.byref(plan)) != 0 or not plan.fits:
        raise SystemExit("bench: %d ranks share one card: a %.2fe9-k-mer table in %d part(s) does not fit 1/%d of its %.0f GB; "
                         "use a smaller table (--lam, --htsize) for a rehearsal" % (ranks_on_dev, n_keys / 1e9, n_parts, ranks_on_dev, total / 1e9))
    os.environ["MC_MZ_FILL"] = "%g" % plan.fill
    log("bench: %d ranks on this card: index at %g k-mers per line (%.1f GB per rank)" % (ranks_on_dev, plan.fill, plan.bytes_per_part / 1e9))
    return True


_SHA = None


def source_sha():
    """hash of the sources libmcclark.so is built from: binds profiles/traffic.json to the kernels it was measured on"""
    global _SHA
    if _SHA is None:
        import glob
        import hashlib
        h = hashlib.sha256()
        for f in sorted(glob.glob(os.path.join(ROOT, "jn_cuclark_amd", "csrc", "*.h*")) + glob.glob(os.path.join(ROOT, "include", "*.h"))):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
        _SHA = h.hexdigest()[:16]
    return _SHA


def extra_genomes(args, torch, np, dev, dev_index, _genomes):
    """The query kernel on the genome-shaped table (`--db genomes`: every stored k-mer from structured genomes -- genera with
    shared sequence, a conserved block, tandem repeats, poly-A), 10 M reads per launch as the headline, checked against
    the oracle.  What a real RefSeq table looks like to the index; the headline table is kinder to it."""
    from jn_cuclark_amd import CuClarkDB, synth_gpu
    k, ht, T = args.k, args.htsize, args.targets
    genomes = synth_gpu.make_structured_genomes(T, 1_500_000, seed=31, device=dev)
    chunks, n_keys = synth_gpu.build_genome_db(genomes, k, ht)
    host = [(c[0].cpu().numpy(), c[1].cpu().numpy(), c[2].cpu().numpy(), c[3], c[4]) for c in chunks]
    del chunks
    genomes_h = genomes.cpu()
    del genomes
    torch.cuda.empty_cache()
    # this table gets the index MC_INDEX=auto picks for it (super-k-mer records: its k-mers come in runs around shared
    # minimizers); the headline table keeps whatever the run was started with
    saved = os.environ.get("MC_INDEX")
    if saved is None:
        os.environ["MC_INDEX"] = "auto"
    db = CuClarkDB(k=k, numBatches=1, numTargets=T, device=dev_index, htsize=ht, maxhits=MAXHITS)
    if saved is None:
        os.environ.pop("MC_INDEX", None)
    try:
        db.read_chunks(lambda: host, n_keys, device=False)
        info = db.db_info()
        genomes = genomes_h.to(dev)
        n_reads = args.reads
        rp_t, con_t = synth_gpu.make_reads(genomes, n_reads, READ_LEN, seed=32)
        fin_t = torch.zeros((n_reads, 5), dtype=torch.int16, device=dev)
        stream = torch.cuda.current_stream().cuda_stream
        for _ in range(2):
            db.query_device(rp_t, con_t, final_t=fin_t, stream=stream)
        torch.cuda.synchronize()
        steps = 5
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        t0 = time.perf_counter()
        for i in range(steps):
            ev[i][0].record()
            db.query_device(rp_t, con_t, final_t=fin_t, stream=stream)
            ev[i][1].record()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        kern = sum(a.elapsed_time(b) for a, b in ev) / steps
        res = {"value": round(n_reads * steps / dt / 1e6, 2), "unit": "Mreads/s", "kernel_ms": round(kern, 4), "steps": steps,
               "reads_per_step": n_reads, "n_kmers_db": n_keys,
               "kernel": ("mc::sk::sk_query_kernel<0, %d>" if info["index_kind"] == 2 else "mc::mz::mz_query_kernel<0, %d>") % (k if k in (31, 27) else 0),
               "index": {"kind": {1: "minimizer", 2: "skm"}.get(info["index_kind"], "lines"),
                         "hbm_bytes_per_kmer": round(info["device_bytes"] / max(1, n_keys), 2),
                         "lines": info["line_end"] - info["line_begin"], "extra_lines": info["n_extra_lines"],
                         "kmers_per_line": round(n_keys / max(1, info["n_lines"]), 2),
                         "lines_overflowing_frac": round(info["n_lines_overflowing"] / max(1, info["line_end"] - info["line_begin"]), 5),
                         "lines_crowded": info["n_lines_crowded"], "largest_line_kmers": info["largest_line"], "hbm_bytes": info["device_bytes"]},
               "what": "10 M x 150 bp per launch on a table whose EVERY k-mer comes from structured genomes (4096 x 1.5 Mb: genera of 4 "
                       "with shared sequence, a conserved block, tandem repeats, poly-A), on the index MC_INDEX=auto picks for it"}
        # HBM traffic per launch of THIS table, from its own rocprofv3 --pmc passes, if they were taken with these sources
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_genomes.json")))
            if (tj.get("source_sha") == source_sha() and tj.get("reads_per_launch") == n_reads and tj.get("htsize") == ht
                    and tj.get("kmers_per_line") == res["index"]["kmers_per_line"]):
                res["traffic"] = tj.get("hbm_bytes_per_launch")
                res["hbm_traffic_GBs"] = round(tj["hbm_bytes_per_launch"] / (kern * 1e-3) / 1e9, 1)
        except Exception:
            pass
        if args.verify:
            from oracle import pyoracle
            fin = fin_t.cpu().numpy().view(np.uint16)
            raw = tuple(np.concatenate([c[i] for c in host]) for i in range(3))
            odb = pyoracle.OracleDB.from_arrays(ht, raw[0], raw[1].view(np.uint32), raw[2].view(np.uint16))
            del raw
            rp_h = rp_t.cpu().numpy().view(np.uint32)
            con_h = con_t.cpu().numpy().view(np.uint16)
            per_read = con_h.size // n_reads
            for idx0 in (0, n_reads - args.verify):
                p = (np.arange(args.verify + 1, dtype=np.uint64) * np.uint64(per_read)).astype(np.uint32)
                want, _ = odb.classify(k, p, con_h[idx0 * per_read:(idx0 + args.verify) * per_read], MAXHITS)
                if not np.array_equal(want, fin[idx0:idx0 + args.verify]):
                    raise SystemExit("bench: HIP results on the genome-shaped table differ from the oracle")
            odb.close()
            res["verified_reads_vs_oracle"] = 2 * args.verify
        return res
    finally:
        db.close()


def extra_e2e_host(args, torch, np, dev, dev_index, genomes):
    """File to CSV through the host driver at the metric's size: the headline table written in the reference's on-disk
    format (.sz/.ky/.lb, 41 GB), a FASTQ file of --e2e-reads reads, `bin/cuCLARK -k 31 -O reads.fq -R res`: the
    program's own timer (the reference's, src/CuCLARK_hh.hh:552-563: FASTQ text -> index -> pack -> GPU -> CSV text, the
    database load outside it), the load it did before (wall clock of the whole process minus that), the CSV checked
    against the ground truth.  Skipped, with the reason, when no directory has room for the files."""
    import shutil
    from jn_cuclark_amd import synth_gpu
    k, ht, T = args.k, args.htsize, args.targets
    need = int(6.5e9 * 6.3) + ht + args.e2e_reads * (synth_gpu.FASTQ_RECORD + 40)
    d = synth_gpu.pick_dir(need)
    if d is None:
        return {"skipped": "no directory with %.0f GB free for the database files and the FASTQ file" % (need / 1e9)}
    work = os.path.join(d, "mc_bench_e2e_%d" % os.getpid())
    os.makedirs(work, exist_ok=True)
    try:
        base = os.path.join(work, "db_central_k%d_t%d_s%d_m0.tsk" % (k, T, ht))
        ranges = [(ht * j // 16, ht * (j + 1) // 16) for j in range(16)]

        def chunks():
            for b0, b1 in ranges:
                d_sz, d_keys, d_labels = synth_gpu.build_db(dev, 31, k, ht, T, args.lam, genomes=genomes, shard=(b0, b1))
                yield d_sz, d_keys, d_labels, b0, b1

        t0 = time.time()
        n_keys, nbytes = synth_gpu.write_db_files(base, chunks())
        t_db = time.time() - t0
        fq = os.path.join(work, "reads.fq")
        truth = synth_gpu.write_fastq(fq, genomes, args.e2e_reads, seed=91).numpy()
        torch.cuda.empty_cache()
        runs = []
        for _ in range(2):          # the hosts are shared: two runs, both reported, their mean is the value
            r = synth_gpu.host_driver_run(os.path.join(ROOT, "bin", "cuCLARK"), work, k, T, fq, args.e2e_reads, threads=16, batches=32, truth=truth)
            ok = r["csv_lines"] == args.e2e_reads and r["assigned_to_their_genome"] > 0.995 * r["checked"] and r["assigned_elsewhere"] < 200
            if not ok:
                raise SystemExit("bench: the host driver's CSV fails the ground-truth check: %r" % (r,))
            runs.append(r)
        r = max(runs, key=lambda x: x["Mreads_per_s"])
        return {"value": round(sum(x["Mreads_per_s"] for x in runs) / len(runs), 2), "unit": "Mreads/s", "runs_Mreads_per_s": [x["Mreads_per_s"] for x in runs],
                "reads": args.e2e_reads, "seconds": r["seconds"],
                "process_wall_s": r["wall_s"], "fastq_GB": round(os.path.getsize(fq) / 1e9, 2), "db_files_GB": round(nbytes / 1e9, 1),
                "db_files_in": d, "db_files_written_s": round(t_db, 1), "threads": 16, "batches": 32, "line": r["line"],
                "csv_lines": r["csv_lines"], "first_reads_checked": r["checked"], "assigned_to_their_genome": r["assigned_to_their_genome"],
                "what": "bin/cuCLARK -k 31 on a FASTQ file against the headline table loaded from its .sz/.ky/.lb files: the program's own "
                        "timer (FASTQ text -> CSV text; the reference's, src/CuCLARK_hh.hh:552-563), process_wall_s = with process start and "
                        "the database load (mc_group_load_db of 41 GB of files)"}
    finally:
        shutil.rmtree(work, ignore_errors=True)


def pipelined_rate(db, torch, np, rp_t, con_t, n_reads, fin_expected, nb=4, rounds=5):
    """PCIe-inclusive rate through mc_alloc_batches / mc_submit / mc_wait (reference malloc / queryBatch /
    waitForBatch, CuClarkDB.cu:321-421, :835-987): the step's batch cut into `nb` batches whose packed reads start
    in pinned HOST memory and whose results end there; all nb in flight, a batch resubmitted as soon as its previous
    results were waited for; copy-in, kernel and copy-out run on three queues.  Never the headline value."""
    per = n_reads // nb
    rp_h = rp_t.cpu().numpy().view(np.uint32)
    con_h = con_t.cpu().numpy().view(np.uint16)
    saved = db.numBatches
    db.numBatches = nb
    try:
        max_con = int(rp_h[per * nb] - rp_h[per * (nb - 1)]) if nb > 1 else int(rp_h[per])
        max_con = max(max_con, int(max(rp_h[per * (b + 1)] - rp_h[per * b] for b in range(nb))))
        rp_l, con_l, fin_l, _ = db.malloc(per, max_con)
        h2d = d2h = 0
        for b in range(nb):
            lo, hi = int(rp_h[per * b]), int(rp_h[per * (b + 1)])
            rp_l[b][: per + 1] = rp_h[per * b: per * (b + 1) + 1] - rp_h[per * b]
            con_l[b][: hi - lo] = con_h[lo:hi]
            db.readyBatch(b, per, hi - lo)
            h2d += (per + 1) * 4 + (hi - lo) * 2
            d2h += per * 10
        for b in range(nb):         # warm
            db.queryBatch(b)
        for b in range(nb):
            db.waitForBatch(b)
        # steady state: a batch is resubmitted as soon as its previous results were waited for (no drain per round)
        t0 = time.perf_counter()
        for i in range(rounds * nb):
            b = i % nb
            if i >= nb:
                db.waitForBatch(b)
            db.queryBatch(b)
        for b in range(nb):
            db.waitForBatch(b)
        dt = time.perf_counter() - t0
        ok = all(np.array_equal(fin_l[b][: per * 5].reshape(per, 5), fin_expected[per * b: per * (b + 1)]) for b in range(nb))
        db.freeBatchMemory()
    finally:
        db.numBatches = saved
    if not ok:
        raise SystemExit("bench: the streamed (pinned-buffer) results differ from the device-resident ones")
    return {"value": round(rounds * nb * per / dt / 1e6, 2), "unit": "Mreads/s",
            "h2d_GBs": round(rounds * h2d / dt / 1e9, 2), "d2h_GBs": round(rounds * d2h / dt / 1e9, 2),
            "batches_in_flight": nb, "reads_per_batch": per,
            "what": "packed reads in pinned host memory -> mc_submit (H2D, kernel, D2H on three queues chained by events) -> mc_wait "
                    "-> final rows in pinned host memory; equal to the device-resident results"}


def shard_path_check(args, dist, torch, np, dev, dev_index, rank, world, backend):
    """N > 1, outside the timed region: run the SHARDED path (rank r holds line-range part r of the index,
    every rank sees every read, sparse rows reduce-scattered by read range over RCCL chunk by chunk on a
    second stream, k-way merge + top-2 on the owner) on a 0.75e9-k-mer table and check it, rank by rank, against
    the unsharded result.  Returns a small dict; never raises (the caller turns a failure into a non-zero exit)."""
    import time as _t
    try:
        from jn_cuclark_amd import CuClarkDB, synth_gpu
        from jn_cuclark_amd.dist import ShardedClassifier, HipBackend, shard_groups
        k, ht, T, lam, n = 29, 200000033, 512, 3.75, 2_000_000
        # 4 ranks and more: 2 parts x world/2 groups (every group holds the whole table, rows are exchanged inside it);
        # fewer: one group of `world` parts
        S = 2 if world >= 4 and world % 2 == 0 else world
        group, gi, part, G = shard_groups(S)
        genomes = synth_gpu.make_genomes(T, 50_000, seed=41, device=dev)
        rp, con = synth_gpu.make_reads(genomes, n, READ_LEN, seed=42)
        full = synth_gpu.build_db(dev, 41, k, ht, T, lam, genomes=genomes)
        with CuClarkDB(k=k, numBatches=1, numTargets=T, device=dev_index, htsize=ht, maxhits=15) as dbf:
            dbf.read_device(*full)
            want = torch.zeros((n, 5), dtype=torch.int16, device=dev)
            dbf.query_device(rp, con, final_t=want, stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
        with CuClarkDB(k=k, numBatches=1, numTargets=T, device=dev_index, htsize=ht, maxhits=15) as dbs:
            dbs.read_chunks(lambda: [(full[0], full[1], full[2], 0, ht)], int(full[1].numel()), part=part, n_parts=S, device=True)
            del full
            sc = ShardedClassifier(HipBackend(dbs, dev), group=group)
            fin, ranges = sc.classify(rp, con, n)
            torch.cuda.synchronize()
            ok = bool(torch.equal(fin, torch.cat([want[lo:hi] for lo, hi in ranges])))
            dist.barrier()
            t0 = _t.perf_counter()
            steps = 5
            for _ in range(steps):
                sc.classify(rp, con, n)
            torch.cuda.synchronize()
            dist.barrier()
            dt = _t.perf_counter() - t0
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return {"verified_equal_to_unsharded": bool(flag.item()), "Mreads_per_s": round(n * steps * G / dt / 1e6, 2),
                "table": "HTSIZE=%d k=%d, %d parts x %d groups, %d reads/step and group in 4 chunks, all_to_all of %d-byte rows inside a group"
                         % (ht, k, S, G, n, 2 * dbs.row_len)}
    except Exception as e:      # the headline measurement must survive a failure here
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}


if __name__ == "__main__":
    main()
