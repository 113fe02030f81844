#!/bin/bash
# round 4, GPU run 1: new tests + the default bench line + shuffle / init comparisons
out=gpurun_out/r1; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_bench.py -x -q -m gpu > $out/t_bench.log 2>&1; echo "rc=$?" >> $out/t_bench.log
timeout 900 python -m pytest tests/test_gpu_mirror.py -x -q -m gpu -k "sharding or plan_query or large_share or differentiable or loader or signature" > $out/t_mirror.log 2>&1; echo "rc=$?" >> $out/t_mirror.log
timeout 900 python bench.py > $out/bench_default.log 2> $out/bench_default.err
for wl in c2 c3; do for ord in shared per_candidate; do
  timeout 300 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --engine-order $ord > $out/bench_${wl}_${ord}.log 2>&1
done; done
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-small-pop --engine-order per_candidate > $out/bench_c1_per_candidate.log 2>&1
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-small-pop --engine-init device > $out/bench_c1_device_init.log 2>&1
timeout 600 python bench.py --workload c5 --steps 2 --warmup 1 > $out/bench_c5.log 2>&1
tail -3 $out/t_bench.log $out/t_mirror.log
