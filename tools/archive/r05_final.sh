#!/bin/bash
# final-tree confirmation: smoke, whole GPU suite, default bench, small-population PMC traffic, c5 kernel statistics
bash tools/r05_suite.sh
out=gpurun_out/r05s; export TMPDIR=/tmp
timeout 600 python tools/pmc_traffic.py $out/pmc small > $out/pmc_small.log 2>&1; tail -25 $out/pmc_small.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/c5prof -o c5 -- python $GRAFT_REPO_ROOT/bench.py --workload c5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/bench_c5.log 2>&1)
f=$(find $out/c5prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $out/c5_kernel_stats.csv && head -8 $out/c5_kernel_stats.csv
tail -1 $out/bench_c5.log | cut -c1-400
