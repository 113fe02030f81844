#!/bin/bash
# chain phase stamps (-DMFAS_CHAIN_TIMING build): shader cycles since the chain function was entered; slots 0 entry done, 1-4 forward cell i starts,
# 5 forward done, 6 head done, 7 softmax done, 8-11 backward cell L-1..0 starts, 12 end; candidate 0, train step 3
export MFAS_LIB=$PWD/mfas_amd/csrc/libmfas_hip_timing.so
for cfg in "128 16 1 1" "128 16 1 6" "128 16 1 16" "16 20 0 6" "16 20 0 28"; do set -- $cfg
  echo "## R=$1 B=$2 bn=$3, $4 candidates (default schedule)"
  timeout 300 python tools/popsweep.py $1 $2 $3 2 $4 2000 800 2>&1 | grep -E "chain timing|K=" | tail -3
done
echo "## R=128 B=16 bn=1, 6 candidates, two launches per step (MFAS_SAME_GROUP=0: the standalone prefetching k_chain)"
MFAS_SAME_GROUP=0 timeout 300 python tools/popsweep.py 128 16 1 2 6 2000 800 2>&1 | grep -E "chain timing|K=" | tail -3
