#!/bin/bash
# The GPU test suite file by file, each under its own hard timeout (a device hang inside a ctypes call cannot be interrupted
# by pytest-timeout), log under gpurun_out/suite.log.  usage (GPU box): bash tools/run_gpu_suite.sh [extra pytest args]
mkdir -p gpurun_out; : > gpurun_out/suite.log
for f in tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_mirror.py tests/test_gpu_mmimdb.py tests/test_gpu_sharing.py tests/test_avmnist.py tests/test_gpu_bench.py tests/test_fullsize.py; do
  echo "=== $f" >> gpurun_out/suite.log
  timeout 600 python -m pytest $f -m gpu -x -q "$@" >> gpurun_out/suite.log 2>&1
  echo "rc=$?" >> gpurun_out/suite.log
done
grep -E "===|rc=|passed|failed|FAILED|Error" gpurun_out/suite.log | tail -40
