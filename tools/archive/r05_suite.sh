#!/bin/bash
# whole GPU suite + the default bench line (+ rocprofv3 kernel statistics of the same command)
out=gpurun_out/r05s; mkdir -p $out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/smoke.log
bash tools/run_gpu_suite.sh > $out/suite_summary.log 2>&1; cp gpurun_out/suite.log $out/suite.log
timeout 900 python bench.py > $out/bench_default.log 2> $out/bench_default.err; echo "bench rc=$?" >> $out/suite_summary.log
tail -3 $out/smoke.log; tail -30 $out/suite_summary.log
python - <<'PY'
import json
l=json.loads(open('gpurun_out/r05s/bench_default.log').read().strip().splitlines()[-1])
print('value',l['value'],'frac',l['roofline']['frac'],'avg_us',l['roofline']['avg_launch_us'])
sp=l['config']['small_pop']
for k,v in sp.items(): print(k, round(v['cand_per_s'],1), v.get('us_per_train_step_incl_dev_eval'), v.get('kernel_us_per_train_step'), v.get('mean_best_dev_acc'))
print(l['config'].get('search_c3'))
PY
