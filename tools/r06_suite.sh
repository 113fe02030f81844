#!/bin/bash
# whole GPU suite file by file (hard timeouts), then the chain_split A/B.  usage (GPU box): bash tools/r06_suite.sh [files...]
mkdir -p gpurun_out; : > gpurun_out/suite.log
files="$@"; [ -z "$files" ] && files="tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_mirror.py tests/test_gpu_mmimdb.py tests/test_gpu_sharing.py tests/test_avmnist.py tests/test_gpu_bench.py tests/test_fullsize.py"
for f in $files; do
  echo "=== $f" >> gpurun_out/suite.log
  timeout 900 python -m pytest $f -m gpu -x -q >> gpurun_out/suite.log 2>&1
  echo "rc=$?" >> gpurun_out/suite.log
done
grep -E "===|rc=|passed|failed|FAILED|Error" gpurun_out/suite.log | tail -40
