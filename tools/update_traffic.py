#!/usr/bin/env python3
"""tools/update_traffic.py <pmc_query.txt> <bench_trace.json> -- turn the FETCH_SIZE / WRITE_SIZE rows of a
tools/prof_pmc.sh run into profiles/traffic.json, keyed by the workload and by the hash of the kernel sources
(bench.py uses the figure only for exactly that build and workload)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

vals = {}
for ln in open(sys.argv[1]):
    p = ln.split("\t")
    if len(p) >= 3 and p[2].startswith("avg_per_launch="):
        vals[p[0]] = float(p[2].split("=")[1])
line = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
fetch_kb, write_kb = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
line_bytes = 128 if line["roofline"]["index"] in ("minimizer", "skm") else 64
# MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE tallies a 128-byte request as 64 bytes.  (The super-k-mer kernel asks for its
# 128-byte lines in two 64-byte halves: the second half hits in L2, the miss brings in the whole line -- TCC_EA0_RDREQ counts one
# request per line and FETCH_SIZE 64 bytes for it: the same factor, cross-checked by RDREQ x 128 B.)
hbm = (2.0 if line_bytes == 128 else 1.0) * fetch_kb * 1024 + write_kb * 1024
out = {
    "source": "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, average per launch of %s)" % (
        os.path.relpath(sys.argv[1], ROOT), line["roofline"]["kernel"]),
    "kernel": line["roofline"]["kernel"], "htsize": line["config"]["htsize"], "db": line["config"].get("db", "synthetic"),
    "reads_per_launch": line["config"]["reads_per_step"], "read_len": 150, "line_bytes": line_bytes,
    "kmers_per_line": line["config"]["index"]["kmers_per_line"],
    "fetch_size_kb": fetch_kb, "write_size_kb": write_kb, "hbm_bytes_per_launch": int(hbm),
    "rdreq": vals.get("TCC_EA0_RDREQ_sum"),
    "source_sha": bench.source_sha(),
    "correction": "FETCH_SIZE x 2 for 128-byte requests (MI355X_MICROARCH.md, HBM section: gfx950 tallies a 128-byte request as "
                  "64 bytes), cross-checked by TCC_EA0_RDREQ x 128 B; 64-byte requests (bucket-line kernel) are exact.",
}
# (a profile of the genome-shaped table -- bench.py --db genomes -- goes to its own file: the `genomes` key of the bench line reads it)
name = "traffic_genomes.json" if out["db"] == "genomes" else "traffic.json"
json.dump(out, open(os.path.join(ROOT, "profiles", name), "w"), indent=1)
print(json.dumps(out))
