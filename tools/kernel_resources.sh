#!/bin/bash
# Register / scratch / LDS use of every talco_lean_kernel instantiation (hipcc remarks; cross-compiles without a GPU).
#   tools/kernel_resources.sh [grep pattern on the demangled template arguments]
cd "$(dirname "$0")/../twilight_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -std=c++17 -Rpass-analysis=kernel-resource-usage -o /tmp/twl_res.so twl_align.hip 2>&1 |
  grep -E "Function Name|SGPRs:|VGPRs:|ScratchSize|Occupancy|LDS Size" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' |
  awk '/Function Name/{if (l) print l; l=$0; next} {l=l" | "$0} END{print l}' | c++filt | grep -E "${1:-talco_lean}"
