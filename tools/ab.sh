#!/bin/bash
# tools/ab.sh lib1.so lib2.so ... -- bench the same workload with several builds of libmcclark.so on one box
# (VERIFY=n reads checked against the oracle, default 0; ARGS="--read-len 100 --reads 4000000" for other workloads; REPS)
cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  for rep in $(seq 1 ${REPS:-2}); do
    MC_LIB_PATH=$PWD/$lib python3 bench.py --no-cpu-baseline --no-pipelined --no-extras --steps 5 --warmup 2 --verify ${VERIFY:-0} $ARGS 2>/tmp/ab.err | python3 -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); print('$lib', j['value'], j['roofline']['kernel_ms'], 'verified', j['config'].get('verified_reads_vs_oracle'))" || tail -5 /tmp/ab.err
  done
done
