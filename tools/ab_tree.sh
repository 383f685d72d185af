#!/bin/bash
# tools/ab_tree.sh dirA dirB ... -- bench.py of several source trees (each with its own built library) on ONE box,
# interleaved twice; "." is the working tree.  For A/B against an older commit: git archive <rev> bench.py jn_cuclark_amd
# profiles/traffic.json | tar -x -C build/ab/<name>, build its library there.
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
  for d in "$@"; do
    ( cd "$d" && python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 --verify 0 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); print('$d', j['value'], j['roofline']['kernel_ms'])" )
  done
done
