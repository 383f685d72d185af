"""The file loader and the host driver at the metric's size (run on the GPU box):
  1. the bench's 6.45e9-k-mer table (HTSIZE 1610612741, k = 31, 4096 targets) written as .sz/.ky/.lb -- the
     reference's on-disk format, 41 GB -- to a directory with room for it (/dev/shm, $TMPDIR, /tmp);
  2. mc_load_db of those files, timed (what a `kent -c` user waits for; reference: CuClarkDB::read,
     src/CuClarkDB.cu:463-770, its timer -DTIME_DBLOADING at src/CuCLARK_hh.hh:617-628), mc_db_info compared with the
     chunk-fed build of the same table, and the classification of 1 M reads compared between the two (bit-identical)
     and with the ground truth (a read sampled from genome g is assigned to target g);
  3. a FASTQ file of N reads (default 40 M x 150 bp, 12.7 GB), `bin/cuCLARK -k 31 -O reads.fq` against that
     database, twice: the program's own "Done in Xs (N reads/min, M reads)" (the reference's timer,
     src/CuCLARK_hh.hh:552-563, :1931-1939: file -> CSV, the database load is outside it) and the wall clock of the
     whole process; the CSV is checked against the ground truth.
    python tools/file_e2e.py [--reads N] [--threads 16] [--dir D] [--json out.json]
The oracle comparison of the file-loaded table lives in tests/test_gpu_filesize.py (tools do not touch oracle/)."""
import argparse
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
from jn_cuclark_amd import CuClarkDB, synth_gpu

K, HT, T, LAM, GLEN, MAXHITS = 31, 1610612741, 4096, 3.75, 100_000, 15


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=40_000_000)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--batches", type=int, default=32)
    ap.add_argument("--dir", default=None)
    ap.add_argument("--lam", type=float, default=LAM)
    ap.add_argument("--json", default=None)
    ap.add_argument("--skip-load-compare", action="store_true")
    ap.add_argument("--paired", action="store_true",
                    help="also: the same reads as mates of two files (-P r_1.fq r_2.fq; mate 2 = the same records again), "
                         "the driver's direct two-file ingest (no joined copy of the files)")
    ap.add_argument("--sweep", default="", help="threads:batches pairs to run after the two standard runs, e.g. 16:64,16:128,32:64")
    ap.add_argument("--card-shares", default="",
                    help="percentages, e.g. 0,40,60,100: after the standard runs, the same file with that share of its ranges going to the "
                         "card as text (MC_GPU_INGEST / MC_CARD_SHARE: 0 = the host indexes and packs everything, 100 = the card does)")
    ap.add_argument("--env-sweep", default="",
                    help="A/B of switches of the host driver: settings separated by ';', each NAME=VALUE[,NAME=VALUE...]; every setting gets "
                         "--env-reps runs after the standard ones, interleaved (e.g. MC_FMT_PREFETCH=0;MC_FMT_PREFETCH=12)")
    ap.add_argument("--env-reps", type=int, default=3)
    ap.add_argument("--load-times", action="store_true",
                    help="also time tools/load_time.py on the files: 1 member, 2 and 3 members (parts) on this card, with this "
                         "build and -- when build/libmcclark_r02.so is there -- with round 2's loader")
    a = ap.parse_args()
    out = {}
    dev = torch.device("cuda", 0)
    need = int(6.5e9 * 6.3 * max(a.lam, 0.1) / 3.75 + 3e9) + HT + a.reads * (synth_gpu.FASTQ_RECORD + 40)
    d = synth_gpu.pick_dir(need, a.dir)
    if d is None:
        print("no directory with %.0f GB free: nothing measured" % (need / 1e9))
        return 2
    work = os.path.join(d, "mc_file_e2e_%d" % os.getpid())
    os.makedirs(work, exist_ok=True)
    try:
        return run(a, out, dev, work)
    finally:
        shutil.rmtree(work, ignore_errors=True)


def run(a, out, dev, work):
    base = os.path.join(work, "db_central_k%d_t%d_s%d_m0.tsk" % (K, T, HT))
    genomes = synth_gpu.make_genomes(T, GLEN, seed=31, device=dev)
    n_ranges = 16
    ranges = [(HT * j // n_ranges, HT * (j + 1) // n_ranges) for j in range(n_ranges)]

    def chunks():
        for b0, b1 in ranges:
            d_sz, d_keys, d_labels = synth_gpu.build_db(dev, 31, K, HT, T, a.lam, genomes=genomes, shard=(b0, b1))
            yield d_sz, d_keys, d_labels, b0, b1
            del d_sz, d_keys, d_labels

    t0 = time.time()
    n_keys, nbytes = synth_gpu.write_db_files(base, chunks())
    out["db_files"] = {"dir": os.path.dirname(work), "n_kmers": n_keys, "bytes": nbytes, "write_s": round(time.time() - t0, 1)}
    print("database files: %.2fe9 k-mers, %.1f GB in %s, generated + written in %.1f s" % (n_keys / 1e9, nbytes / 1e9, work, time.time() - t0), flush=True)
    torch.cuda.empty_cache()

    n1 = 1_000_000
    rp, con, truth = synth_gpu.make_reads(genomes, n1, 150, seed=77, return_truth=True)
    st = torch.cuda.current_stream().cuda_stream
    # ---- 2. load from the files --------------------------------------------------------------------------------
    loads = []
    fin_file = info_file = None
    for rep in range(2):                     # second time: the files come from the page cache
        with CuClarkDB(k=K, numBatches=1, numTargets=T, device=0, htsize=HT, maxhits=MAXHITS) as db:
            t0 = time.time()
            assert db.read(base) is True
            loads.append(round(time.time() - t0, 2))
            info_file = db.db_info()
            fin_file = torch.zeros((n1, 5), dtype=torch.int16, device=dev)
            db.query_device(rp, con, final_t=fin_file, stream=st)
            torch.cuda.synchronize()
        torch.cuda.empty_cache()
    out["load"] = {"seconds": loads, "GB_per_s": [round(2 * (nbytes - HT) / 1e9 / s, 2) for s in loads],
                   "what": "mc_load_db of the .sz/.ky/.lb files: .sz once, .ky/.lb streamed twice (count pass, place pass) "
                           "through two pinned chunk buffers into the index build; first = as written, second = again"}
    print("mc_load_db: %s s (%.1f GB of index, %.2f k-mers per line)" % (loads, info_file["device_bytes"] / 1e9,
                                                                          n_keys / max(1, info_file["n_lines"])), flush=True)
    fin = fin_file.cpu().numpy().view(np.uint16)
    tr = truth.cpu().numpy()
    ok = fin[: tr.size, 1] == tr + 1
    assert ok.mean() > 0.995 and int(fin[tr.size:, 0].astype(np.int64).sum()) < 100
    out["load"]["ground_truth_ok"] = round(float(ok.mean()), 5)
    if not a.skip_load_compare:
        with CuClarkDB(k=K, numBatches=1, numTargets=T, device=0, htsize=HT, maxhits=MAXHITS) as db:
            t0 = time.time()
            os.environ["MC_MZ_FILL"] = "%g" % (round(2.0 * n_keys / info_file["n_lines"]) / 2.0)      # the file loader keeps its chunks in HBM and takes a denser fill for it
            db.read_chunks(chunks, n_keys, device=True)
            os.environ.pop("MC_MZ_FILL", None)
            fed_s = time.time() - t0
            info_fed = db.db_info()
            fin_fed = torch.zeros((n1, 5), dtype=torch.int16, device=dev)
            db.query_device(rp, con, final_t=fin_fed, stream=st)
            torch.cuda.synchronize()
        same = {k_: info_file[k_] == info_fed[k_] for k_ in info_file}
        assert torch.equal(fin_file, fin_fed), "file-loaded and chunk-fed tables classify differently"
        assert all(same.values()), ("mc_db_info differs", {k_: (info_file[k_], info_fed[k_]) for k_, v in same.items() if not v})
        out["load"]["equal_to_chunk_fed_build"] = True
        out["load"]["chunk_fed_build_s"] = round(fed_s, 2)
        print("file-loaded index == chunk-fed index (mc_db_info field by field, 1 M reads bit-identical); chunk-fed build %.1f s" % fed_s, flush=True)
        torch.cuda.empty_cache()
    del rp, con, fin_file

    if a.load_times:
        libs = [os.path.join(ROOT, "jn_cuclark_amd", "libmcclark.so")]
        if os.path.exists(os.path.join(ROOT, "build", "libmcclark_r02.so")):
            libs.append(os.path.join(ROOT, "build", "libmcclark_r02.so"))
        lt = []
        for members, mode in ((1, "auto"), (2, "shards"), (3, "shards")):
            for lib in libs:
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "load_time.py"), base, "--lib", lib,
                                    "--members", str(members), "--mode", mode], capture_output=True, text=True)
                line = (r.stdout.strip().split("\n") or [""])[-1] if r.returncode == 0 else "FAILED: " + (r.stdout + r.stderr)[-300:]
                print(line, flush=True)
                lt.append(line)
        out["load_times"] = lt

    # ---- 3. FASTQ -> CSV through the host driver -----------------------------------------------------------------
    fq = os.path.join(work, "reads.fq")
    t0 = time.time()
    truth = synth_gpu.write_fastq(fq, genomes, a.reads, seed=91).numpy()
    print("FASTQ: %d reads, %.2f GB, generated + written in %.1f s" % (a.reads, os.path.getsize(fq) / 1e9, time.time() - t0), flush=True)
    del genomes
    torch.cuda.empty_cache()
    exe = os.path.join(ROOT, "bin", "cuCLARK")
    runs = []
    for rep in range(2):
        r = synth_gpu.host_driver_run(exe, work, K, T, fq, a.reads, threads=a.threads, batches=a.batches, truth=truth)
        assert r["csv_lines"] == a.reads and r["assigned_to_their_genome"] > 0.995 * r["checked"] and r["assigned_elsewhere"] < 200, r
        runs.append(r)
        print("run %d: wall %.1f s | %s | %s" % (rep, r["wall_s"], r["line"], " | ".join(r["timing"])), flush=True)
    for tb in [x for x in a.sweep.split(",") if x]:
        t_, b_ = (int(v) for v in tb.split(":"))
        r = synth_gpu.host_driver_run(exe, work, K, T, fq, a.reads, threads=t_, batches=b_, truth=truth)
        print("sweep -n %d -b %d: %.2f Mreads/s, wall %.1f s | %s" % (t_, b_, r["Mreads_per_s"], r["wall_s"], " | ".join(r["timing"])), flush=True)
        out.setdefault("sweep", []).append({"threads": t_, "batches": b_, "Mreads_per_s": r["Mreads_per_s"], "wall_s": r["wall_s"]})
    for sh in [int(x) for x in a.card_shares.split(",") if x]:
        os.environ["MC_GPU_INGEST"] = "1" if sh > 0 else "0"
        os.environ["MC_CARD_SHARE"] = str(sh)
        for rep in range(2):
            r = synth_gpu.host_driver_run(exe, work, K, T, fq, a.reads, threads=a.threads, batches=a.batches, truth=truth)
            assert r["csv_lines"] == a.reads and r["assigned_to_their_genome"] > 0.995 * r["checked"] and r["assigned_elsewhere"] < 200, r
            size = os.path.getsize(os.path.join(work, "res.csv"))
            print("card share %3d %% run %d: %.2f Mreads/s, wall %.1f s, csv %d bytes | %s" % (sh, rep, r["Mreads_per_s"], r["wall_s"], size, " | ".join(r["timing"])[:420]), flush=True)
            out.setdefault("card_shares", []).append({"share": sh, "Mreads_per_s": r["Mreads_per_s"], "wall_s": r["wall_s"], "csv_bytes": size})
    os.environ.pop("MC_GPU_INGEST", None)
    os.environ.pop("MC_CARD_SHARE", None)
    settings = [x for x in a.env_sweep.split(";") if x]
    for rep in range(a.env_reps if settings else 0):
        for st in settings:
            kv = dict(x.split("=", 1) for x in st.split(","))
            os.environ.update(kv)
            r = synth_gpu.host_driver_run(exe, work, K, T, fq, a.reads, threads=a.threads, batches=a.batches, truth=truth)
            for k_ in kv:
                os.environ.pop(k_, None)
            assert r["csv_lines"] == a.reads and r["assigned_to_their_genome"] > 0.995 * r["checked"], r
            print("env %-40s rep %d: %.2f Mreads/s, wall %.1f s | %s" % (st, rep, r["Mreads_per_s"], r["wall_s"], " | ".join(r["timing"])[:200]), flush=True)
            out.setdefault("env_sweep", []).append({"env": st, "Mreads_per_s": r["Mreads_per_s"], "wall_s": r["wall_s"]})
    if a.paired:
        # mate files: the same records with /1 and /2 behind the id (the ids match after the cut at '/', src/file.cc:205-268)
        f1, f2 = os.path.join(work, "r_1.fq"), os.path.join(work, "r_2.fq")
        rec = np.fromfile(fq, dtype=np.uint8).reshape(-1, synth_gpu.FASTQ_RECORD)
        n_pairs = min(a.reads, 20_000_000)
        for path, tag in ((f1, b"/1"), (f2, b"/2")):
            m = rec[:n_pairs].copy()
            m[:, 10:12] = np.frombuffer(tag, dtype=np.uint8)
            m.tofile(path)
            del m
        del rec
        r = synth_gpu.host_driver_run(exe, work, K, T, f1, n_pairs, threads=a.threads, batches=a.batches, fastq2=f2)
        assert r["csv_lines"] == n_pairs, r
        print("paired: %d pairs of 2 x 150 bp from two files: %.2f M pairs/s, wall %.1f s | %s" % (n_pairs, r["Mreads_per_s"], r["wall_s"], " | ".join(r["timing"])), flush=True)
        out["e2e_host_paired"] = {"pairs": n_pairs, "Mpairs_per_s": r["Mreads_per_s"], "wall_s": r["wall_s"], "timing": r["timing"]}
        os.remove(f1); os.remove(f2)
    out["e2e_host"] = {"reads": a.reads, "fastq_GB": round(os.path.getsize(fq) / 1e9, 2), "runs": runs,
                       "Mreads_per_s": max(x["Mreads_per_s"] for x in runs),
                       "what": "bin/cuCLARK -k 31 -O reads.fq: FASTQ text -> index -> pack -> GPU -> CSV text, the program's own timer "
                               "(the reference's, src/CuCLARK_hh.hh:552-563: the database load is outside it); wall_s = the whole process"}
    print(json.dumps(out), flush=True)
    if a.json:
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
