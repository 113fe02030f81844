#!/bin/bash
# round 4 evidence run (GPU box): the default bench line, rocprofv3 kernel statistics of the same command, the search-sized
# workloads, the MM-IMDB-shaped workload, PMC traffic / MFMA passes, the self-spawned 2-rank run, population sweeps.
out=gpurun_out/r04; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python bench.py > $out/bench_pop128.log 2> $out/bench_pop128.err
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rp_bench -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-small-pop > $GRAFT_REPO_ROOT/$out/rp_bench.log 2>&1)
timeout 600 python bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_c2_1gpu.log 2>&1
timeout 600 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_c3_1gpu.log 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rp_c3 -o c3 -- python $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/rp_c3.log 2>&1)
timeout 600 python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_c5_1gpu.log 2>&1
timeout 600 python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --engine-order shared > $out/bench_c5_1gpu_shared_order.log 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rp_c5 -o c5 -- python $GRAFT_REPO_ROOT/bench.py --workload c5 --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/rp_c5.log 2>&1)
timeout 900 python bench.py --gpus 2 --backend gloo --steps 1 --warmup 1 --pop 64 --no-cpu-baseline > $out/bench_2ranks_gloo_1gpu.log 2> $out/bench_2ranks_gloo_1gpu.err
timeout 1200 python tools/pmc_traffic.py $out/pmc > $out/pmc_traffic.log 2>&1
timeout 1200 python tools/pmc_mfma.py $out/pmc_mfma > $out/pmc_mfma.log 2>&1
{ timeout 400 python tools/popsweep.py 16 20 0 10 4,6,8,12,16,24,28,32,50
  timeout 400 python tools/popsweep.py 16 20 0 10 6,16,28,50 mixed
  timeout 600 python tools/popsweep.py 128 16 1 10 1,3,6,8,12,16,24,32; } 2>&1 | grep -v amdgpu > $out/popsweep.log
timeout 900 python main_searchable_ntu.py --synthetic 10000 5600 --num_samples 50 --search_iterations 5 --max_fusions 4 --epochs 10 --no-verbose --timing > $out/search_config4.log 2>&1
find $out -name "*kernel_stats.csv" | head; ls -la $out
