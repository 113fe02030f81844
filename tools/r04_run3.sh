#!/bin/bash
# round 4, GPU run 3: multi-chunk units without spills (tests + group-factor sweep), chain MFMA-priority experiment, ASAN variant
out=gpurun_out/r3; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "multi_chunk or full_size_properties or same_group_launch or loaded_library" > $out/t_parity.log 2>&1; echo "rc=$?" >> $out/t_parity.log
timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "natural" > $out/t_fuzz.log 2>&1; echo "rc=$?" >> $out/t_fuzz.log
timeout 900 python -m pytest tests/test_gpu_bench.py -x -q -m gpu -k "asan" > $out/t_asan.log 2>&1; echo "rc=$?" >> $out/t_asan.log
for sub in 1 4 8 2 16 1 4; do
  MFAS_SUBCHUNKS=$sub timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-small-pop --engine-init device > $out/bench_c1_sub${sub}_$RANDOM.log 2>&1
done
MFAS_SUBCHUNKS=4 timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-small-pop --engine-order per_candidate --engine-init device > $out/bench_c1_sub4_percand.log 2>&1
MFAS_SUBCHUNKS=8 timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-small-pop --engine-order per_candidate --engine-init device > $out/bench_c1_sub8_percand.log 2>&1
{ for prio in 0 1 0 1; do echo "# MFAS_CHAIN_PRIO=$prio"; MFAS_CHAIN_PRIO=$prio timeout 600 python tools/popsweep.py 128 16 1 10 1,3,6,12 2>&1 | grep -v amdgpu; done; } > $out/popsweep_prio.log
tail -n 3 $out/t_*.log
