#!/bin/bash
# tools/prof_pmc.sh <tag> [bench args...] -- run on the GPU box (inside gpurun).
# Separate rocprofv3 passes (kernel-trace/stats only, and one --pmc group per pass, as the
# MI355X guide prescribes); extracts the mc::query_kernel rows into gpurun_out/<tag>/.
set -u
TAG=${1:-pmc}; shift || true
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p $OUT
ARGS="--steps ${PROF_STEPS:-3} --warmup ${PROF_WARMUP:-1} --no-cpu-baseline --no-pipelined --no-extras --verify 0 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
{ head -1 $f; grep "query_kernel" $f; } > $OUT/kernel_stats_query.csv
cp $f $OUT/all_kernel_stats.csv
i=0
# PMC_SHORT=1: only the passes the bench line and DESIGN's per-read figures need (traffic, instruction mix, clock, DRAM requests)
if [ -n "${PMC_SHORT:-}" ]; then
  GROUPS_LIST=("FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" \
    "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
    "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
    "GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum")
else
  GROUPS_LIST=()
fi
for grp in "${GROUPS_LIST[@]}" \
  "FETCH_SIZE TCC_HIT_sum" \
  "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
  "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
  "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_sum" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
  "GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY" \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum" \
  "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_BUSY_sum" ; do
  i=$((i+1))
  if [ -n "${PMC_SHORT:-}" ] && [ $i -gt ${#GROUPS_LIST[@]} ]; then break; fi
  if [ -n "${PMC_MAX:-}" ] && [ $i -gt ${PMC_MAX} ]; then break; fi
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
  else echo "group $i: no counter file (see $OUT/pmc$i.err)" >> $OUT/pmc_query.txt; fi
  rm -rf $OUT/pmc$i
done
rm -rf $OUT/trace/*/*kernel_trace.csv
cat $OUT/kernel_stats_query.csv; cat $OUT/pmc_query.txt
# (NO_TRAFFIC=1: a profile of another workload must not replace the headline's traffic figure)
if [ -z "${NO_TRAFFIC:-}" ]; then
python3 tools/update_traffic.py $OUT/pmc_query.txt $OUT/bench_trace.json > $OUT/traffic.json
fi
