#!/bin/bash
# round 6, VERDICT item 2: the headline (128 x conf 4, R=128) with (a) reduce-in-sweep forced, (b) 128-column chunks (one slab per 128 columns), against the default
mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" python bench.py --gpus 1 --steps 3 --warmup 1 --no-small-pop --no-search --no-cpu-baseline $EXTRA > gpurun_out/r06_hl_$tag.json 2> gpurun_out/r06_hl_$tag.err
  python - <<PY
import json
d=[json.loads(l) for l in open("gpurun_out/r06_hl_$tag.json") if l.startswith("{")][-1]
r=d["roofline"]
print("$tag", "cand/s %.2f  ms/step %.1f  wall us/launch %.1f  hip-event us/launch %.1f  frac(wall) %.3f" % (d["value"], d["ms_per_step"], r["wall_us_per_launch"], r["avg_launch_us"], r["frac"]))
PY
}
EXTRA="" run default A=1
EXTRA="" run red_in_sweep MFAS_FORCE_RED_IN_SWEEP=1
EXTRA="--chunk-cols 128" run cc128 A=1
EXTRA="" run default_again A=1
