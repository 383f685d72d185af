#!/bin/bash
# tools/ab_headline.sh lib1.so lib2.so ... -- the headline table with several builds on one box (VERIFY=0 for measurement builds)
cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  MC_LIB_PATH=$PWD/$lib python3 bench.py --no-cpu-baseline --no-pipelined --no-extras --steps 5 --warmup 2 --verify ${VERIFY:-5000} 2>/tmp/ab_h.err | python3 -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); print('$lib', j['value'], j['roofline']['kernel_ms'], 'verified', j['config'].get('verified_reads_vs_oracle'))" || tail -5 /tmp/ab_h.err
done
