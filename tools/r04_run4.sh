#!/bin/bash
# round 4, GPU run 4: device-side torch streams, mixed group factors, ASAN variant, remaining new tests
out=gpurun_out/r4; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_bench.py -x -q -m gpu > $out/t_bench.log 2>&1; echo "rc=$?" >> $out/t_bench.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "loaded_library or full_size_properties or natural" > $out/t_parity.log 2>&1; echo "rc=$?" >> $out/t_parity.log
for cfg in "1 0" "4 4" "4 2" "2 2" "4 3" "1 0"; do set -- $cfg
  MFAS_SUBCHUNKS=$1 MFAS_SUBCHUNK_SKIP=$2 timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-small-pop --engine-init device > $out/bench_c1_sub$1_skip$2_$RANDOM.log 2>&1
done
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_default.log 2>&1
for wl in c2 c3; do for ord in shared per_candidate; do
  timeout 300 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --engine-order $ord > $out/bench_${wl}_${ord}.log 2>&1
done; done
MFAS_HOST_INIT=1 timeout 300 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_c3_hostinit.log 2>&1
tail -n 3 $out/t_*.log
