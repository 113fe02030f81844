#!/bin/bash
# final tree: sweeps, chain stamps and the step trace again (after the ticket publish, the XCD-aware placement and the BatchNorm divisions)
out=gpurun_out/r05g; mkdir -p $out
python -c "import __graft_entry__ as g; g.build(); print(g.build_variant('timing', ['-DMFAS_CHAIN_TIMING']))" > $out/build.log 2>&1
{ timeout 400 python tools/popsweep.py 16 20 0 10 1,4,6,8,12,16,24,28
  timeout 400 python tools/popsweep.py 16 20 0 10 6,16,28 mixed
  timeout 600 python tools/popsweep.py 128 16 1 10 1,3,6,8,16; } 2>&1 | grep -v amdgpu > $out/popsweep.log
{ export MFAS_LIB=$PWD/mfas_amd/csrc/libmfas_hip_timing.so
  for cfg in "16 20 0 1" "16 20 0 6" "16 20 0 16" "16 20 0 28" "16 20 1 6" "128 16 1 1" "128 16 1 6"; do set -- $cfg
    echo "## R=$1 B=$2 bn=$3, $4 candidates (default schedule)"
    timeout 300 python tools/popsweep.py $1 $2 $3 2 $4 2000 800 2>&1 | grep -E "chain timing|K=" | tail -3
  done; unset MFAS_LIB; } > $out/chain_phases.log 2>&1
bash tools/r05_trace.sh > /dev/null 2>&1; cp gpurun_out/r05t/trace.log $out/persist_trace.log
cat $out/popsweep.log; grep -E "##|chain timing" $out/chain_phases.log | head -30; grep -E "step 1[0-2]:|ready" $out/persist_trace.log
