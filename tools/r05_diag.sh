#!/bin/bash
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -x 2>&1 | tail -6
for lib in tools/_prev.so mfas_amd/csrc/libmfas_hip.so; do
echo "## $lib"
MFAS_LIB=$PWD/$lib timeout 600 python tools/popsweep.py 16 20 0 10 6,16,28
done
bash tools/r04_chain_phases.sh 2>&1 | grep -A3 "R=16"
} > gpurun_out/r05_diag.log 2>&1
cat gpurun_out/r05_diag.log | cut -c1-400
