#!/bin/bash
# round 4: k_eval with exact bf16 x 3 feature products (R = 128, bf16 tables) against the f32-MFMA build of the same kernel
out=gpurun_out/eval_b3; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -x -q -m gpu -k "random_population or natural or forward or eval or baseline or golden" > $out/tests.log 2>&1; echo "tests rc=$?" >> $out/tests.log
for v in b3 f32; do
  if [ $v = f32 ]; then export MFAS_EVAL_NO_B3=1; else unset MFAS_EVAL_NO_B3; fi
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_$v -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-small-pop > $GRAFT_REPO_ROOT/$out/bench_$v.log 2> $GRAFT_REPO_ROOT/$out/bench_$v.err)
  f=$(find $out/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "## $v"; head -c 400 $out/bench_$v.log; echo; grep -i "k_eval\|k_step" $f | head -4
done
tail -5 $out/tests.log
