#!/bin/bash
# deferred unit hand-off (persist.hip.h, two-unit workgroups) A/B on one box: MFAS_RES_NO_DEFER=1 = the previous behaviour
out=gpurun_out/s5; mkdir -p $out
{
for cfg in "16 20 0 28 10" "16 20 0 22 10" "16 20 0 28 10 10000 5600 mixed" "16 20 0 16 10"; do
  for rep in 1 2; do
    echo "## $cfg  defer on";  timeout 300 python tools/persist_check.py $cfg toggle=MFAS_UNUSED persist 2>&1 | grep -E "persist=1|IDENT|MISM" | tail -2
    echo "## $cfg  defer OFF"; MFAS_RES_NO_DEFER=1 timeout 300 python tools/persist_check.py $cfg toggle=MFAS_UNUSED persist 2>&1 | grep -E "persist=1|IDENT|MISM" | tail -2
  done
done
echo "## bit-identity: MFAS_RES_NO_DEFER toggled under the resident schedule"
timeout 300 python tools/persist_check.py 16 20 0 28 3 4000 800 toggle=MFAS_RES_NO_DEFER persist 2>&1 | grep -E "persist=|IDENT|MISM"
timeout 300 python tools/persist_check.py 16 20 0 24 3 4000 800 mixed toggle=MFAS_RES_NO_DEFER persist 2>&1 | grep -E "persist=|IDENT|MISM"
timeout 300 python tools/persist_check.py 16 20 0 22 2 4010 800 cc=256 2>&1 | grep -E "persist=|IDENT|MISM"
} > $out/defer_ab.log 2>&1
cat $out/defer_ab.log
