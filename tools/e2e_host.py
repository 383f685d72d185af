#!/usr/bin/env python3
"""End-to-end timing of the host driver on a FASTQ file (run on the GPU box):
generate n reads of 150 bp (half from toy genomes), build a light database through the
driver itself, then time `bin/cuCLARK-l -O reads.fq` as the reference reports it
(`Done in Xs (N reads/min, M reads)`, src/CuCLARK_hh.hh:1931-1939).
e2e_host.py [reads] [threads] [workdir] [paired | gz | bgzf]"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jn_cuclark_amd import synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    threads = sys.argv[2] if len(sys.argv) > 2 else "16"
    work = sys.argv[3] if len(sys.argv) > 3 else "/tmp/e2e"
    os.makedirs(work + "/db", exist_ok=True)
    genomes = synth.toy_genomes(8, 200_000, seed=71)
    tl = []
    for i, g in enumerate(genomes):
        p = "%s/g%d.fa" % (work, i)
        open(p, "wb").write(synth.fasta_text([b"g%d" % i], [synth.codes_to_ascii(g)], width=80))
        tl.append("%s\tT%d\n" % (p, i))
    open(work + "/targets.txt", "w").write("".join(tl))
    t0 = time.time()
    rng = np.random.default_rng(5)
    half = n // 2
    gi = rng.integers(0, 8, half)
    pos = rng.integers(0, 200_000 - 150, half)
    allg = np.stack(genomes)
    idx = pos[:, None] + np.arange(150)[None, :]
    planted = allg[gi[:, None], idx]
    mut = rng.random((half, 150)) < 0.01
    planted = np.where(mut, (planted + rng.integers(1, 4, (half, 150))) & 3, planted).astype(np.uint8)
    codes = np.concatenate([planted, rng.integers(0, 4, (n - half, 150), dtype=np.uint8)])
    seq = synth.CODE2BASE[codes]                       # [n, 150] ASCII
    rec = np.empty((n, 12 + 1 + 150 + 3 + 150 + 1), dtype=np.uint8)
    names = np.char.zfill(np.arange(n).astype(str), 10)
    rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
    rec[:, 2:12] = np.frombuffer("".join(names).encode(), dtype=np.uint8).reshape(n, 10)
    rec[:, 12] = 10
    rec[:, 13:163] = seq
    rec[:, 163] = 10; rec[:, 164] = ord("+"); rec[:, 165] = 10
    rec[:, 166:316] = ord("I")
    rec[:, 316] = 10
    fq = work + "/reads.fq"
    rec.tofile(fq)
    print("generated %d reads (%.2f GB) in %.1fs" % (n, os.path.getsize(fq) / 1e9, time.time() - t0), flush=True)
    exe = os.environ.get("MC_E2E_EXE") or os.path.join(ROOT, "bin", "cuCLARK-l")          # (MC_E2E_EXE: another build of the driver, e.g. one with -fsanitize=thread)
    inputs = ["-O", fq]
    if len(sys.argv) > 4 and sys.argv[4] == "paired":
        # the same reads as mates: file 2 = the same records with "/2" names (ids match after the cut at '/')
        rec2 = rec.copy()
        rec[:, 10:12] = np.frombuffer(b"/1", dtype=np.uint8)
        rec2[:, 10:12] = np.frombuffer(b"/2", dtype=np.uint8)
        f1, f2 = work + "/reads_1.fq", work + "/reads_2.fq"
        rec.tofile(f1)
        rec2.tofile(f2)
        inputs = ["-P", f1, f2]
        print("paired: 2 x %.2f GB" % (os.path.getsize(f1) / 1e9), flush=True)
    want_sha = None
    if len(sys.argv) > 4 and sys.argv[4] in ("gz", "bgzf"):
        # the same file gzipped (one member, zlib level 1) or as BGZF blocks (bgzip's layout): classified segment by segment
        # (host/gzstream.hpp); the CSV must be the plain file's
        import hashlib
        import zlib
        r = subprocess.run([exe, "-T", work + "/targets.txt", "-D", work + "/db", "-O", fq, "-R", work + "/res", "-n", threads, "-b", "32"],
                           capture_output=True, text=True)
        want_sha = hashlib.sha256(open(work + "/res.csv", "rb").read()).hexdigest()[:16]
        os.remove(work + "/res.csv")
        t0 = time.time()
        raw = open(fq, "rb").read()
        if sys.argv[4] == "gz":
            co = zlib.compressobj(1, zlib.DEFLATED, 31)
            z = co.compress(raw) + co.flush()
        else:
            import struct
            out = bytearray()
            for i in list(range(0, len(raw), 65280)) + [len(raw)]:
                c = raw[i:i + 65280]
                co = zlib.compressobj(1, zlib.DEFLATED, -15)
                d = co.compress(c) + co.flush()
                out += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(d) + 25) + d
                out += struct.pack("<II", zlib.crc32(c), len(c))
            z = bytes(out)
        fq = work + "/reads.fq.gz"
        open(fq, "wb").write(z)
        inputs = ["-O", fq]
        print("%s: %.2f GB in %.1fs (plain run: rc=%d, csv %s)" % (sys.argv[4], len(z) / 1e9, time.time() - t0, r.returncode, want_sha), flush=True)
        del raw, z
    for run in range(2):
        t0 = time.time()
        r = subprocess.run([exe, "-T", work + "/targets.txt", "-D", work + "/db"] + inputs + ["-R", work + "/res",
                            "-n", threads, "-b", "32", "--verbose"], capture_output=True, text=True)
        dt = time.time() - t0
        tail = [l for l in r.stderr.split("\n") if "Done in" in l or "timing" in l]
        print("run %d: rc=%d wall %.2fs  %s" % (run, r.returncode, dt, " | ".join(tail)), flush=True)
    lines = sum(1 for _ in open(work + "/res.csv"))
    print("csv lines:", lines)
    if want_sha:
        import hashlib
        got = hashlib.sha256(open(work + "/res.csv", "rb").read()).hexdigest()[:16]
        print("csv of the gzip run %s the plain run's (%s)" % ("EQUALS" if got == want_sha else "DIFFERS FROM", got))
        if got != want_sha:
            raise SystemExit(1)


if __name__ == "__main__":
    main()
