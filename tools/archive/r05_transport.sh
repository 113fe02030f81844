#!/bin/bash
# round 5: granule transport of the resident schedule — bit-identity against launch-per-phase, then the step times
mkdir -p gpurun_out
{
for cfg in "16 20 0 6 2 2000 800" "16 20 1 6 2 2000 800 alphas" "16 20 0 16 2 2000 800 mixed" "16 16 1 28 2 1000 400 mixed" "16 20 0 3 2 2000 800"; do
  echo "## persist_check $cfg"
  timeout 300 python tools/persist_check.py $cfg 2>&1 | tail -4
done
echo "## popsweep R=16"
timeout 900 python tools/popsweep.py 16 20 0 10 4,6,8,16,24,28
timeout 600 python tools/popsweep.py 16 20 0 10 6,16,28 mixed
} > gpurun_out/r05_transport.log 2>&1
tail -40 gpurun_out/r05_transport.log
