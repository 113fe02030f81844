#!/bin/bash
# round 6: chain_split — parity first, then the A/B timing and the stamped trace.  usage (GPU box): bash tools/r06_split.sh
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "chain_split or same_group_launch_fuzz or full_size_properties" > gpurun_out/r06_split_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r06_split_tests.log
tail -3 gpurun_out/r06_split_tests.log
timeout 600 python tools/split_ab.py 1,2,3,6,8,10 2 2>&1 | grep -E "^#|K=" > gpurun_out/r06_split_ab.log
cat gpurun_out/r06_split_ab.log
export MFAS_LIB=$PWD/mfas_amd/csrc/libmfas_hip_timing.so
for k in 1 6; do timeout 200 python tools/split_ab.py $k 1 2>&1 | grep -E "chain timing|K="; done > gpurun_out/r06_split_stamps.log
grep split gpurun_out/r06_split_stamps.log | sed "s/.*| split/| split/"
