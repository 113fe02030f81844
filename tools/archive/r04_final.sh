#!/bin/bash
# final check of the round's tree on a GPU box: smoke(), the whole GPU suite, the default bench line
out=gpurun_out/final; mkdir -p $out
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/smoke.log
bash tools/run_gpu_suite.sh > $out/suite_summary.log 2>&1; cp gpurun_out/suite.log $out/suite.log
timeout 900 python bench.py > $out/bench_default.log 2> $out/bench_default.err
tail -3 $out/smoke.log; cat $out/suite_summary.log | tail -24; head -c 600 $out/bench_default.log
