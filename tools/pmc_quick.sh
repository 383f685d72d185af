#!/bin/bash
# tools/pmc_quick.sh "<bench args>" lib1.so lib2.so ... -- instruction counts of the query kernel for several builds on one
# box: one rocprofv3 --pmc pass per build (SQ_INSTS_VALU / SALU / LDS / VMEM_RD, SQ_BUSY_CYCLES, SQ_WAIT_INST_ANY), average per launch
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
ARGS="$1"; shift
for lib in "$@"; do
  O=gpurun_out/pmcq_$(basename $lib .so); rm -rf $O; mkdir -p $O
  MC_LIB_PATH=$PWD/$lib rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O -- python3 bench.py --no-cpu-baseline --no-pipelined --no-extras --verify 0 --steps 3 --warmup 1 $ARGS > $O/bench.json 2> $O/err.txt
  f=$(find $O -name "*counter_collection.csv" | head -1)
  echo "== $lib $(python3 -c "import json;j=json.loads(open('$O/bench.json').readlines()[-1]);print(j['value'], j['roofline']['kernel_ms'])" 2>/dev/null)"
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "query_kernel" in r.get("Kernel_Name", ""):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("  " + "  ".join("%s %.4g" % (k, sum(v) / len(v)) for k, v in sorted(acc.items())))
PY
  rm -rf $O/*/
done
