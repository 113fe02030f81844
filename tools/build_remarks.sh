#!/bin/bash
# compile the library with resource-usage remarks and print the per-kernel table (no GPU needed).  usage: tools/build_remarks.sh [extra -D flags]
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -ffp-contract=off -DMFAS_SRC_DIGEST=\"remarks\" "$@" -Iinclude -shared -fPIC \
  mfas_amd/csrc/mfas_hip.hip -o /tmp/mfas_remarks.so -Rpass-analysis=kernel-resource-usage 2> /tmp/mfas_remarks.txt || { grep -E "error" -A5 /tmp/mfas_remarks.txt | head -60; exit 1; }
python tools/resource_usage.py /tmp/mfas_remarks.txt
