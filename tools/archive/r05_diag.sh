#!/bin/bash
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests/test_fullsize.py -q -x -m gpu -k "search_default" -s 2>&1 | tail -30
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r05_bench_quick.json 2> gpurun_out/r05_bench_quick.err; echo "bench rc=$?"
python - <<'PY'
import json
l=json.load(open('gpurun_out/r05_bench_quick.json'))
print('value',l['value'],'frac',l['roofline']['frac'],'avg_us',l['roofline']['avg_launch_us'], 'prof', l['roofline'].get('profile_box_avg_us'))
sp=l['config']['small_pop']
for k,v in sp.items(): print(k, round(v['cand_per_s'],1), v.get('us_per_train_step_incl_dev_eval'), v.get('kernel_us_per_train_step'), v.get('mean_best_dev_acc'))
print(l['config']['search_c3'])
PY
} > gpurun_out/r05_diag.log 2>&1
cat gpurun_out/r05_diag.log | cut -c1-600
