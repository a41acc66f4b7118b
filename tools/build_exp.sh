#!/bin/bash
# Cross-compiled variants of libtwl_align.so for the geometry experiments (tools/exp_thr.py, tools/lone_pair_probe.py):
#   bash tools/build_exp.sh <name> -DTWL_EXP_THR_W=4 -DTWL_EXP_THR_RPL=2 -DTWL_EXP_THR_MINW=5     ->  build_exp/<name>.so  (+ <name>.resources: registers, scratch, LDS per kernel)
set -e
cd "$(dirname "$0")/.."
mkdir -p build_exp
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -std=c++17 '-DTWL_SOURCE_HASH="experiment"' "$@" \
    -Rpass-analysis=kernel-resource-usage -o build_exp/$name.so twilight_amd/csrc/twl_align.hip 2> build_exp/$name.resources
grep -A12 "talco_lean_kernelILi6ELi${W:-4}" build_exp/$name.resources | head -0
