#!/bin/bash
# dev pass beside the next epoch's resident launch (second plane set, mfas_hip.hip "eval overlap") A/B on one box: MFAS_NO_EVAL_OVERLAP=1 = before
out=gpurun_out/s5; mkdir -p $out
{
for cfg in "16 20 0 22 10 10000 5600" "16 20 0 25 10 10000 5600 mixed" "16 20 0 7 10 10000 5600 mixed" "16 20 0 13 10 10000 5600 mixed" "16 20 0 16 10 10000 5600"; do
    echo "## $cfg  (persist=0 lines: MFAS_NO_EVAL_OVERLAP=1, persist=1 lines: overlap)"
    timeout 300 python tools/persist_check.py $cfg toggle=MFAS_NO_EVAL_OVERLAP persist 2>&1 | grep -E "persist=|IDENT|MISM" | sed 's/ status \[.*//'
done
echo "## resident vs launch-per-phase, same units (cc=256), overlap on: statistics and parameters bit-identical"
timeout 300 python tools/persist_check.py 16 20 0 6 4 4010 800 cc=256 2>&1 | grep -E "persist=|IDENT|MISM" | sed 's/ status \[.*//'
timeout 300 python tools/persist_check.py 16 20 0 12 3 4000 800 mixed cc=256 2>&1 | grep -E "persist=|IDENT|MISM" | sed 's/ status \[.*//'
} > $out/overlap_ab.log 2>&1
cat $out/overlap_ab.log
