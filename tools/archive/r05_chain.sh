#!/bin/bash
# round 5: the register-resident lean chain — parity suites, schedule bit-identity, step times, phase stamps
out=gpurun_out/r05c; mkdir -p $out
python -c "import __graft_entry__ as g; g.build(); print(g.build_variant('timing', ['-DMFAS_CHAIN_TIMING']))" > $out/build.log 2>&1
{
for cfg in "16 20 0 6 2 2000 800" "16 20 1 6 2 2000 800 alphas" "16 20 0 16 2 2000 800 mixed" "16 16 1 28 2 1000 400 mixed" "16 20 0 3 2 2000 800"; do
  echo "## persist_check $cfg"
  timeout 300 python tools/persist_check.py $cfg cc=256 2>&1 | grep -v amdgpu | tail -3
done
} > $out/persist_check.log 2>&1
for f in tests/test_gpu_parity.py tests/test_gpu_fuzz.py; do
  timeout 900 python -m pytest $f -m gpu -x -q 2>&1 | tail -15 > $out/$(basename $f .py).log
done
{ timeout 400 python tools/popsweep.py 16 20 0 10 1,6,16,28
  timeout 400 python tools/popsweep.py 16 20 0 10 16,28 mixed; } 2>&1 | grep -v amdgpu > $out/popsweep.log
export MFAS_LIB=$PWD/mfas_amd/csrc/libmfas_hip_timing.so
for cfg in "16 20 0 6" "16 20 1 6" "16 20 0 28"; do set -- $cfg
  echo "## R=$1 B=$2 bn=$3, $4 candidates (default schedule)"
  timeout 300 python tools/popsweep.py $1 $2 $3 2 $4 2000 800 2>&1 | grep -E "chain timing|K=" | tail -3
done > $out/chain_phases.log 2>&1
unset MFAS_LIB
cat $out/persist_check.log $out/test_gpu_parity.log $out/test_gpu_fuzz.log $out/popsweep.log $out/chain_phases.log
