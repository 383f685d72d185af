#!/bin/bash
# tools/prof_lds.sh <tag> -- LDS / issue counters of the query kernel (run inside gpurun)
set -u
TAG=${1:-lds}; shift || true
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --verify 0 $*"
i=0
for grp in \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" \
  "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_ADDR_CONFLICT" ; do
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
