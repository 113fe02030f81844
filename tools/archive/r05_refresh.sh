#!/bin/bash
# final tree: the default bench line + the search-sized workloads again (the scheduler-table memo moved configs[2] / [3])
out=gpurun_out/r05f; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python bench.py > $out/bench_pop128.log 2> $out/bench_pop128.err
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rp_bench -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-small-pop > $GRAFT_REPO_ROOT/$out/rp_bench.log 2>&1)
timeout 600 python bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_c2_1gpu.log 2>&1
timeout 600 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_c3_1gpu.log 2>&1
python - <<'PY'
import json
for f in ("bench_pop128","bench_c2_1gpu","bench_c3_1gpu"):
    l=json.loads([x for x in open(f"gpurun_out/r05f/{f}.log").read().strip().splitlines() if x.startswith("{")][-1])
    print(f, round(l["value"],1), "ms/step", round(l["ms_per_step"],1), {k:l["roofline"].get(k) for k in ("avg_launch_us","frac","profile_box_avg_us")}, l["roofline"].get("stream_probe"))
    if f=="bench_pop128":
        for k,v in l["config"]["small_pop"].items(): print("  ",k, round(v["cand_per_s"],1), v.get("us_per_train_step_incl_dev_eval"), v.get("kernel_us_per_train_step"))
        print("  search_c3", {k:l["config"]["search_c3"].get(k) for k in ("total_s","train_s","controller_s","cand_per_s","decision_digest")})
        print("  cpu", l["cpu_baseline"]["value"], "other_both", l["config"]["other_both"]["cand_per_s"])
PY
