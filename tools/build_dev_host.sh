#!/bin/bash
# A TWL_DEV build of the host library + CLI into <dir> (development dumps: TWL_DUMP_BATCH, TWL_DUMP_SCHEDULE); the product build is __graft_entry__.build().
#   tools/build_dev_host.sh <dir>
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
D=$1; mkdir -p $D
cp $R/twilight_amd/libtwl_align.so $D/
cd $R/twilight_amd/csrc/host
g++ -O2 -std=c++17 -fopenmp -ffp-contract=off -DTWL_DEV -fPIC -shared -o $D/libtwl_host.so phylo.cpp seqdb_io.cpp helpers.cpp progressive.cpp driver.cpp align_gpu.cpp align_resident.cpp align_owned.cpp capi.cpp $D/libtwl_align.so -Wl,-rpath,'$ORIGIN' -lz
g++ -O2 -std=c++17 -fopenmp -ffp-contract=off -o $D/twilight-mi355x main.cpp $D/libtwl_host.so $D/libtwl_align.so -Wl,-rpath,'$ORIGIN' -lz
