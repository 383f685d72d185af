#!/bin/bash
# pipelined (PCIe-inclusive) rate of bench.py under a few settings: does the copy of one batch overlap the kernel of the other?
cd "$GRAFT_REPO_ROOT"
run() { echo "== $*"; env "$@" python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --verify 0 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); p = j['pipelined']; print(j['value'], 'pipelined', p['value'], 'h2d', p['h2d_GBs'], 'd2h', p['d2h_GBs'])"; }
for v in "$@"; do run $v; done
