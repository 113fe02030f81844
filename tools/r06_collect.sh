#!/bin/bash
# round 6 evidence run (GPU box): the default bench line, rocprofv3 kernel statistics of the same command, the search-sized workloads, the
# MM-IMDB-shaped workload, PMC traffic / MFMA passes, population sweeps (R = 16 with / without BatchNorm, R = 128 with / without chain_split),
# chain phase stamps.  Summaries are copied to profiles/r06_* by the builder.
out=gpurun_out/r06; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python bench.py > $out/bench_pop128.log 2> $out/bench_pop128.err
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rp_bench -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-small-pop --no-search > $GRAFT_REPO_ROOT/$out/rp_bench.log 2>&1)
timeout 600 python bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_c2_1gpu.log 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rp_c2 -o c2 -- python $GRAFT_REPO_ROOT/bench.py --workload c2 --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/rp_c2.log 2>&1)
timeout 600 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_c3_1gpu.log 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rp_c3 -o c3 -- python $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/rp_c3.log 2>&1)
timeout 600 python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_c5_1gpu.log 2>&1
timeout 600 python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --engine-order shared > $out/bench_c5_1gpu_shared_order.log 2>&1
# the R = 128 small populations under rocprofv3 (k_step_same<1, *, 4>: chain_split)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rp_split -o split -- python $GRAFT_REPO_ROOT/tools/split_ab.py 1,6 2 > $GRAFT_REPO_ROOT/$out/rp_split.log 2>&1)
timeout 1200 python tools/pmc_traffic.py $out/pmc > $out/pmc_traffic.log 2>&1
timeout 1200 python tools/pmc_mfma.py $out/pmc_mfma > $out/pmc_mfma.log 2>&1
{ timeout 400 python tools/popsweep.py 16 20 0 10 1,4,6,8,12,16,24,28
  timeout 400 python tools/popsweep.py 16 20 0 10 6,16,28 mixed
  timeout 400 python tools/popsweep.py 16 20 1 10 1,6,16,28
  timeout 900 python tools/split_ab.py 1,2,3,6,8,10,12,16,24 10; } 2>&1 | grep -v amdgpu > $out/popsweep.log
{ export MFAS_LIB=$PWD/mfas_amd/csrc/libmfas_hip_timing.so
  for cfg in "16 20 0 6" "16 20 1 6"; do set -- $cfg
    echo "## R=$1 B=$2 bn=$3, $4 candidates (default schedule)"
    timeout 300 python tools/popsweep.py $1 $2 $3 2 $4 2000 800 2>&1 | grep -E "chain timing|K=" | tail -3
  done
  for k in 1 6; do echo "## R=128 B=16 bn=1, $k candidates: one-CU chain (first three stamp lines), chain_split (next three)"; timeout 200 python tools/split_ab.py $k 1 2>&1 | grep -E "chain timing|K="; done
  unset MFAS_LIB; } > $out/chain_phases.log 2>&1
find $out -name "*kernel_stats.csv" | head; ls $out | head -40
python - <<'PY'
import json
for f in ("bench_pop128","bench_c2_1gpu","bench_c3_1gpu","bench_c5_1gpu","bench_c5_1gpu_shared_order"):
    try:
        l=json.loads([x for x in open(f"gpurun_out/r06/{f}.log").read().strip().splitlines() if x.startswith("{")][-1])
        print(f, round(l["value"],1), "ms/step", round(l["ms_per_step"],1), {k:l["roofline"].get(k) for k in ("wall_us_per_launch","avg_launch_us","frac","achieved_hip_events")})
        if f=="bench_pop128":
            for k,v in l["config"]["small_pop"].items(): print("  ",k, round(v["cand_per_s"],1), v.get("us_per_train_step_incl_dev_eval"), v.get("kernel_us_per_train_step"))
            print("  search_c3", {k:l["config"]["search_c3"].get(k) for k in ("total_s","train_s","controller_s","cand_per_s","decision_digest")})
    except Exception as e: print(f, "ERR", e)
PY
