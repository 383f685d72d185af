#!/bin/bash
# tools/stats_build.sh -- build/libmcclark_stats.so: the library with the lookup counters of mc_minimizer.hpp compiled in
# (-DMC_MZ_STATS; slower, for measurement only).  tools/kernel_stats.py runs a workload with it and prints where the
# lookups end: first line / chain, Bloom false positives, chain lines fetched.
set -e
cd "$(dirname "$0")/../jn_cuclark_amd/csrc"
mkdir -p ../../build
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DMC_MZ_STATS "$@" -shared -o ../../build/libmcclark_stats.so mc_api.hip mc_group.hip mc_build.hip
