#!/bin/bash
out=gpurun_out/r05c5; mkdir -p $out
timeout 600 python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_c5_1gpu.log 2>$out/bench_c5.err
timeout 600 python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --engine-order shared > $out/bench_c5_1gpu_shared_order.log 2>/dev/null
timeout 600 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_c3_1gpu.log 2>/dev/null
python - <<'PY'
import json
for f in ("bench_c5_1gpu","bench_c5_1gpu_shared_order","bench_c3_1gpu"):
    try:
        l=json.loads(open(f"gpurun_out/r05c5/{f}.log").read().strip().splitlines()[-1])
        print(f, round(l["value"],1), l["unit"], "ms/step", round(l["ms_per_step"],1), {k:l["roofline"].get(k) for k in ("avg_launch_us","frac","launches")}, l["roofline"].get("schedule"))
    except Exception as e: print(f, "ERR", e)
PY
