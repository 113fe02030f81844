#!/bin/bash
# round 5 evidence run (GPU box): the default bench line, rocprofv3 kernel statistics of the same command, the search-sized workloads, the
# MM-IMDB-shaped workload, PMC traffic / MFMA passes, population sweeps, chain phase stamps, the resident schedule's step trace.
out=gpurun_out/r05; mkdir -p $out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); print(g.build_variant('timing', ['-DMFAS_CHAIN_TIMING']))" > $out/build.log 2>&1
timeout 900 python bench.py > $out/bench_pop128.log 2> $out/bench_pop128.err
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rp_bench -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-small-pop > $GRAFT_REPO_ROOT/$out/rp_bench.log 2>&1)
timeout 600 python bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_c2_1gpu.log 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rp_c2 -o c2 -- python $GRAFT_REPO_ROOT/bench.py --workload c2 --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/rp_c2.log 2>&1)
timeout 600 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_c3_1gpu.log 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rp_c3 -o c3 -- python $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/rp_c3.log 2>&1)
timeout 600 python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_c5_1gpu.log 2>&1
timeout 600 python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --engine-order shared > $out/bench_c5_1gpu_shared_order.log 2>&1
timeout 1200 python tools/pmc_traffic.py $out/pmc > $out/pmc_traffic.log 2>&1
timeout 1200 python tools/pmc_mfma.py $out/pmc_mfma > $out/pmc_mfma.log 2>&1
{ timeout 400 python tools/popsweep.py 16 20 0 10 1,4,6,8,12,16,24,28
  timeout 400 python tools/popsweep.py 16 20 0 10 6,16,28 mixed
  timeout 600 python tools/popsweep.py 128 16 1 10 1,3,6,8,16; } 2>&1 | grep -v amdgpu > $out/popsweep.log
{ export MFAS_LIB=$PWD/mfas_amd/csrc/libmfas_hip_timing.so
  for cfg in "16 20 0 1" "16 20 0 6" "16 20 0 16" "16 20 0 28" "16 20 1 6" "128 16 1 1" "128 16 1 6"; do set -- $cfg
    echo "## R=$1 B=$2 bn=$3, $4 candidates (default schedule)"
    timeout 300 python tools/popsweep.py $1 $2 $3 2 $4 2000 800 2>&1 | grep -E "chain timing|K=" | tail -3
  done; unset MFAS_LIB; } > $out/chain_phases.log 2>&1
bash tools/r05_trace.sh > /dev/null 2>&1; cp gpurun_out/r05t/trace.log $out/persist_trace.log
(rocprofv3 -L 2>/dev/null || rocprofv3 --list-avail 2>/dev/null) | grep -iE "mall|dram|hbm|TCC_EA|TCC_HIT|TCC_MISS|TCC_REQ" | head -60 > $out/counters_avail.log 2>&1
find $out -name "*kernel_stats.csv" | head; ls $out | head -40
python - <<'PY'
import json
for f in ("bench_pop128","bench_c2_1gpu","bench_c3_1gpu","bench_c5_1gpu","bench_c5_1gpu_shared_order"):
    try:
        l=json.loads([x for x in open(f"gpurun_out/r05/{f}.log").read().strip().splitlines() if x.startswith("{")][-1])
        print(f, round(l["value"],1), "ms/step", round(l["ms_per_step"],1), {k:l["roofline"].get(k) for k in ("avg_launch_us","frac","launches")})
        if f=="bench_pop128":
            for k,v in l["config"]["small_pop"].items(): print("  ",k, round(v["cand_per_s"],1), v.get("us_per_train_step_incl_dev_eval"), v.get("kernel_us_per_train_step"))
            print("  search_c3", {k:l["config"]["search_c3"].get(k) for k in ("total_s","train_s","controller_s","cand_per_s","decision_digest")})
    except Exception as e: print(f, "ERR", e)
PY
