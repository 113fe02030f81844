#!/bin/bash
# configs[3] / configs[4] bench lines with and without the deferred unit hand-off, same box; then the fuzz suite (persistent schedules bit-identical)
out=gpurun_out/s5; mkdir -p $out
for w in c3 c5; do
  for sw in on off; do
    if [ $sw = off ]; then export MFAS_RES_NO_DEFER=1; else unset MFAS_RES_NO_DEFER; fi
    timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_${w}_defer_${sw}.log 2>&1
    python - <<PY
import json
l=[x for x in open("$out/bench_${w}_defer_${sw}.log") if x.startswith("{")]
d=json.loads(l[-1]); r=d["roofline"]
print("$w defer $sw:", round(d["value"],1), "cand/s", round(d["ms_per_step"],1), "ms/step; launch us", round(r["avg_launch_us"] or 0,1), "schedule", r.get("schedule"))
PY
  done
done 2>&1 | tee $out/defer_bench.log
unset MFAS_RES_NO_DEFER
timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3 | tee -a $out/defer_bench.log
