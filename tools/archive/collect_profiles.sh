#!/bin/bash
# Round profile collection on a GPU box: every step bounded by its own timeout, outputs under gpurun_out/prof (copy the
# summaries to profiles/ afterwards).  usage: bash tools/collect_profiles.sh [part]   part = sweep | bench | misc | all
part=${1:-all}
out=gpurun_out/prof
mkdir -p $out
export TMPDIR=/tmp
if [ "$part" = sweep ] || [ "$part" = all ]; then
  { timeout 400 python tools/popsweep.py 16 20 0 10 4,6,8,12,16,24,28,32,50
    timeout 400 python tools/popsweep.py 16 20 0 10 6,16,28,50 mixed
    timeout 400 python tools/popsweep.py 16 16 1 10 6,16,28
    timeout 400 python tools/popsweep.py 64 16 1 10 6,16,32
    timeout 600 python tools/popsweep.py 128 16 1 10 3,6,8,12,16,24,32; } 2>&1 | grep -v amdgpu > $out/popsweep.log
  { timeout 600 python tools/popsweep.py 128 16 1 10 1,3,6,8,12,16,24,32
    timeout 400 python tools/popsweep.py 64 16 1 10 6,16,32
    timeout 400 python tools/popsweep.py 32 20 0 10 6,16,32
    MFAS_SAME_GROUP=0 timeout 400 python tools/popsweep.py 128 16 1 10 1,3,6,8; } 2>&1 | grep -v amdgpu > $out/popsweep_general.log
  { for cfg in "16 20 0 6 2 2000 800 cc=256" "16 16 1 9 2 2000 800 cc=128" "16 20 0 7 2 2000 800 mixed alphas cc=512"; do
      timeout 300 python tools/persist_check.py $cfg 2>&1 | grep -E "cand/s|IDENT|MISMATCH|differ"; done; } > $out/persist_check.log
  MFAS_PERSIST_TRACE=1 timeout 300 python tools/persist_check.py 16 20 0 6 2 2000 800 cc=256 2>&1 | grep -v amdgpu > $out/persist_trace_k6_r16.log
fi
if [ "$part" = bench ] || [ "$part" = all ]; then
  timeout 900 python bench.py > $out/bench_pop128.log 2>&1
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rp_bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/rp_bench.log 2>&1)
  timeout 600 python bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_c2_1gpu.log 2>&1
  timeout 600 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_c3_1gpu.log 2>&1
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rp_c2 -- python $GRAFT_REPO_ROOT/bench.py --workload c2 --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/rp_c2.log 2>&1)
  find $out -name "*kernel_stats.csv" | head
fi
if [ "$part" = misc ] || [ "$part" = all ]; then
  { timeout 300 python tests/devtools/call_breakdown.py 128 16 1 6,16,50,128; timeout 300 python tests/devtools/call_breakdown.py 16 20 0 6,16,50,128; } > $out/call_breakdown.log 2>&1
  timeout 900 python main_searchable_ntu.py --synthetic 10000 5600 --num_samples 50 --search_iterations 5 --max_fusions 4 --epochs 10 --no-verbose --timing > $out/search_config4.log 2>&1
fi
ls -la $out
