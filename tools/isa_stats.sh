#!/bin/bash
# tools/isa_stats.sh [-DFLAG ...] -- compile mc_api.hip for gfx950 (device only), disassemble the unsharded minimizer
# query kernel and count what matters for it: instructions by class, lane spills of scalars (v_writelane/v_readlane),
# scratch.  Runs in the build container (no GPU needed).
set -e
cd "$(dirname "$0")/../jn_cuclark_amd/csrc"
T=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-function --offload-device-only "$@" -c mc_api.hip -o $T/dev.o
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/dev.o --targets=hip-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.elf
K=${KERNEL:-_ZN2mc2mz15mz_query_kernelILi0ELi31EEEvNS0_6MzArgsE}
/opt/rocm/lib/llvm/bin/llvm-objdump -d --disassemble-symbols=$K $T/dev.elf > $T/k.s
python3 - $T/k.s <<'PY'
import re, sys, collections
c = collections.Counter()
n = 0
for ln in open(sys.argv[1]):
    m = re.match(r"\s+([a-z_0-9]+)\s", ln)
    if not m: continue
    op = m.group(1); n += 1
    cls = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") and not op.startswith(("s_load", "s_waitcnt", "s_cbranch", "s_branch", "s_nop")) else
           "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else
           "branch" if op.startswith(("s_cbranch", "s_branch")) else "other")
    c[cls] += 1
    if op in ("v_writelane_b32", "v_readlane_b32", "v_readfirstlane_b32"): c[op] += 1
    if op.startswith("scratch_"): c["scratch"] += 1
print("instructions", n, dict(c))
PY
rm -rf $T
