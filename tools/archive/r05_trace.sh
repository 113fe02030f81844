#!/bin/bash
# persistent-loop trace (10 ns ticks) of the resident schedule at R=16: where a step's time goes between chain and units
out=gpurun_out/r05t; mkdir -p $out
for cfg in "16 20 0 6 2 2000 800" "16 20 0 16 2 2000 800"; do
  echo "## $cfg"
  MFAS_PERSIST_TRACE=1 timeout 300 python tools/persist_check.py $cfg cc=256 2>&1 | grep -v amdgpu | grep -E "trace|step |ready|persist=1|IDENT|MISM" | tail -14
done > $out/trace.log 2>&1
cat $out/trace.log
