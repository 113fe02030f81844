#!/bin/bash
# round 4, GPU run 2: multi-chunk sweep units (tests + timing), staged init upload, batched per-candidate orders
out=gpurun_out/r2; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "multi_chunk or full_size_properties or same_group_launch or loaded_library" > $out/t_parity.log 2>&1; echo "rc=$?" >> $out/t_parity.log
timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "natural" > $out/t_fuzz.log 2>&1; echo "rc=$?" >> $out/t_fuzz.log
timeout 900 python -m pytest tests/test_gpu_mirror.py -x -q -m gpu -k "per_candidate or signature or init_from_module or sharding" > $out/t_mirror.log 2>&1; echo "rc=$?" >> $out/t_mirror.log
timeout 900 python -m pytest tests/test_gpu_bench.py -x -q -m gpu -k "asan or backward or holds" > $out/t_bench.log 2>&1; echo "rc=$?" >> $out/t_bench.log
for sub in 4 1 2 8; do
  MFAS_SUBCHUNKS=$sub timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-small-pop --engine-init device > $out/bench_c1_sub$sub.log 2>&1
done
MFAS_SUBCHUNKS=4 timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-small-pop --engine-order per_candidate > $out/bench_c1_sub4_percand.log 2>&1
MFAS_SUBCHUNKS=1 timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-small-pop --engine-order per_candidate > $out/bench_c1_sub1_percand.log 2>&1
for wl in c2 c3; do for ord in shared per_candidate; do
  timeout 300 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --engine-order $ord > $out/bench_${wl}_${ord}.log 2>&1
done; done
timeout 300 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline --engine-init device > $out/bench_c3_devinit.log 2>&1
tail -n 3 $out/t_*.log
