#!/bin/bash
# tools/prof_counters.sh <tag> "<counters of pass 1>" ["<counters of pass 2>" ...] -- ad-hoc PMC passes over
# bench.py's query kernel (run inside gpurun; one rocprofv3 --pmc pass per argument)
set -u
TAG=$1; shift
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-pipelined --verify 0 ${BENCH_ARGS:-}"
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc$i -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc$i.err
  f=$(find $OUT/pmc$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" >> $OUT/pmc_query.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if "query_kernel" in r.get("Kernel_Name", ""):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("%s\tlaunches=%d\tavg_per_launch=%.6g" % (k, len(v), sum(v) / len(v)))
PY
  else echo "group $i: no counter file" >> $OUT/pmc_query.txt; tail -3 $OUT/pmc$i.err >> $OUT/pmc_query.txt; fi
  rm -rf $OUT/pmc$i
done
cat $OUT/pmc_query.txt
