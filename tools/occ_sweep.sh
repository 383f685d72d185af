cd "$GRAFT_REPO_ROOT"
for occ in 8 7 8 7; do
  MC_GRID_OCC=$occ MC_INDEX=skm python3 bench.py --db genomes --no-cpu-baseline --no-pipelined --no-extras --steps 5 --warmup 2 --verify 5000 2>/tmp/occ.err | python3 -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); print('occ $occ', j['value'], j['roofline']['kernel_ms'])" || tail -5 /tmp/occ.err
done
