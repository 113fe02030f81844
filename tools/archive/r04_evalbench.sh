#!/bin/bash
# round 4: the dev pass alone (tools/evalbench.py), f32 products (MFAS_EVAL_NO_B3=1) against the bf16 x 3 build; then the eval tests
out=gpurun_out/eval_b3; mkdir -p $out
for env in "MFAS_EVAL_NO_B3=1" "MFAS_X=0"; do
  for rk in "128 128" "128 8" "96 64"; do
    env $env timeout 300 python tools/evalbench.py $rk 2>&1 | tail -2
  done
done > $out/evalbench.log 2>&1
cat $out/evalbench.log
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_mirror.py -x -q -m gpu -k "random_population or natural or forward or eval or baseline or golden or digest or built_from" 2>&1 | tail -5
