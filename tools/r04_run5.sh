#!/bin/bash
# round 4, GPU run 5: first-batch preload ahead of the staging barrier
out=gpurun_out/r5; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "loaded_library or multi_chunk or full_size_properties or same_group_launch or schedule_fuzz or tap_major" > $out/t_parity.log 2>&1; echo "rc=$?" >> $out/t_parity.log
timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu > $out/t_fuzz.log 2>&1; echo "rc=$?" >> $out/t_fuzz.log
timeout 900 python -m pytest tests/test_gpu_bench.py -x -q -m gpu -k "asan or streams" > $out/t_bench.log 2>&1; echo "rc=$?" >> $out/t_bench.log
for cfg in "1 shared" "1 per_candidate" "4 shared" "1 shared" "1 per_candidate" "2 shared"; do set -- $cfg
  MFAS_SUBCHUNKS=$1 timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-small-pop --engine-order $2 > $out/bench_c1_sub$1_$2_$RANDOM.log 2>&1
done
timeout 300 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_c3.log 2>&1
timeout 300 python bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_c2.log 2>&1
timeout 600 python bench.py --workload c5 --steps 2 --warmup 1 > $out/bench_c5.log 2>&1
{ timeout 600 python tools/popsweep.py 128 16 1 10 1,6,16,32,64 2>&1 | grep -v amdgpu; timeout 600 python tools/popsweep.py 16 20 0 10 6,28,50,128,512 2>&1 | grep -v amdgpu; } > $out/popsweep.log
tail -n 3 $out/t_*.log
