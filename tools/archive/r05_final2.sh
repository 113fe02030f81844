#!/bin/bash
# final tree after the deferred unit hand-off: A/B log, whole GPU suite + default bench, c3 / c5 lines with kernel statistics, population sweep
bash tools/r05_defer_ab.sh > /dev/null 2>&1
bash tools/r05_suite.sh
out=gpurun_out/r05s; export TMPDIR=/tmp
for w in c3 c5; do
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/${w}prof -o $w -- python $GRAFT_REPO_ROOT/bench.py --workload $w --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/bench_$w.log 2>&1)
f=$(find $out/${w}prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $out/${w}_kernel_stats.csv && head -4 $out/${w}_kernel_stats.csv | cut -c1-160
tail -1 $out/bench_$w.log | cut -c1-200
done
{ timeout 400 python tools/popsweep.py 16 20 0 10 1,6,16,20,24,28
  timeout 400 python tools/popsweep.py 16 20 0 10 16,22,28 mixed; } 2>&1 | grep -v amdgpu > $out/popsweep.log
cat $out/popsweep.log
