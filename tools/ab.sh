#!/bin/bash
# tools/ab.sh lib1.so lib2.so ... -- bench the same workload with several builds of libmcclark.so on one box
cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  for rep in 1 2; do
    MC_LIB_PATH=$PWD/$lib python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --verify 0 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); print('$lib', j['value'], j['roofline']['kernel_ms'])"
  done
done
