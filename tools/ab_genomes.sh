#!/bin/bash
# tools/ab_genomes.sh lib1.so lib2.so ... -- the genome-shaped table (bench.py --db genomes) with several builds on one box
cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  MC_LIB_PATH=$PWD/$lib python3 bench.py --db genomes --no-cpu-baseline --no-pipelined --no-extras --steps 5 --warmup 2 --verify ${VERIFY:-5000} 2>/tmp/ab_g.err | python3 -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); i = j['config']['index']; print('$lib', j['value'], j['roofline']['kernel_ms'], 'kmers/line', i['kmers_per_line'], 'overflow', i['lines_overflowing_frac'], 'GB', round(i['hbm_bytes']/1e9,1), 'verified', j['config'].get('verified_reads_vs_oracle'))" || tail -5 /tmp/ab_g.err
done
