#!/bin/bash
# round 4, GPU run 6: the whole GPU suite on the current tree + small-K chunk sizes at R=128 + sample-order cost on large R=16 populations
out=gpurun_out/r6; mkdir -p $out
bash tools/run_gpu_suite.sh > $out/suite_summary.log 2>&1; cp gpurun_out/suite.log $out/suite.log
{ for cc in 0 64 128; do echo "# cc=$cc"; timeout 600 python tools/popsweep.py 128 16 1 10 1,3,6,10 cc=$cc 2>&1 | grep -v amdgpu; done; } > $out/popsweep_r128_cc.log
for ord in per_candidate shared; do
  timeout 600 python bench.py --workload c5 --steps 2 --warmup 1 --engine-order $ord > $out/bench_c5_$ord.log 2>&1
  timeout 600 python bench.py --R 16 --no-bn --batch 20 --pop 128 --steps 2 --warmup 1 --no-cpu-baseline --no-small-pop --engine-order $ord > $out/bench_r16_pop128_$ord.log 2>&1
  timeout 600 python bench.py --R 16 --no-bn --batch 20 --pop 512 --steps 1 --warmup 1 --no-cpu-baseline --no-small-pop --engine-order $ord > $out/bench_r16_pop512_$ord.log 2>&1
done
cat $out/suite_summary.log | tail -30
