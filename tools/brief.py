#!/usr/bin/env python3
"""print the interesting fields of a bench.py JSON line read from stdin"""
import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r = d["roofline"]
    print(tag, "Mreads/s=%.1f" % d["value"], "kernel_ms=%.3f" % r["kernel_ms"], "index=%s" % r.get("index"),
          "alg_GB/s=%.0f" % r["achieved"], "hit=%s" % d["config"].get("kmer_hit_rate"), flush=True)
