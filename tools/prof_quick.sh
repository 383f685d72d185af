#!/bin/bash
# tools/prof_quick.sh <kernel-substring> [bench args]: a few PMC groups for one kernel (GPU box)
KN=$1; shift
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pq; rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --verify 0 $*"
for grp in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
  "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_HIT_sum" \
  "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" ; do
  rm -rf $OUT/p
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p -- python3 bench.py $ARGS > /dev/null 2> $OUT/err.txt
  f=$(find $OUT/p -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$KN" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if sys.argv[2] in r.get("Kernel_Name", ""):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("%s\t%.6g" % (k, sum(v) / len(v)))
PY
done
