#!/bin/bash
# tools/ab_raw.sh "<bench args>" lib1.so lib2.so ... -- kernel time of several builds on one box, nothing verified (for
# measurement builds that give wrong results on purpose); REPS rounds, the builds interleaved
cd "$GRAFT_REPO_ROOT"
ARGS="$1"; shift
for rep in $(seq 1 ${REPS:-2}); do
for lib in "$@"; do
    MC_LIB_PATH=$PWD/$lib python3 bench.py --no-cpu-baseline --no-pipelined --no-extras --steps 5 --warmup 2 --verify 0 $ARGS 2>/tmp/ab.err | python3 -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); print('$lib', j['value'], j['roofline']['kernel_ms'])" || tail -5 /tmp/ab.err
done
done
