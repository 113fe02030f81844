#!/bin/bash
# deferred unit hand-off (persist.hip.h, two-unit workgroups) A/B on one box: MFAS_RES_NO_DEFER=1 = the previous behaviour
out=gpurun_out/s5; mkdir -p $out
{
for cfg in "16 20 0 28 10 10000 5600" "16 20 0 22 10 10000 5600" "16 20 0 28 10 10000 5600 mixed" "16 20 0 16 10 10000 5600"; do
    [ "${cfg:8:2}" = "16" ] && export MFAS_RES_DEFER=1 || unset MFAS_RES_DEFER   # (16 candidates: two 256-column units, off by default: forced on for the A/B)
    echo "## $cfg  (persist=0 lines: MFAS_RES_NO_DEFER=1, persist=1 lines: deferred hand-off)"
    timeout 300 python tools/persist_check.py $cfg toggle=MFAS_RES_NO_DEFER persist 2>&1 | grep -E "persist=|IDENT|MISM"
done
unset MFAS_RES_DEFER
echo "## short last batch (N = 4010: 10 rows), 3 epochs"
timeout 300 python tools/persist_check.py 16 20 0 28 3 4010 800 toggle=MFAS_RES_NO_DEFER persist 2>&1 | grep -E "persist=|IDENT|MISM"
timeout 300 python tools/persist_check.py 16 20 0 24 3 4010 800 mixed 2>&1 | grep -E "persist=|IDENT|MISM"
echo "## trace"
for cfg in "16 20 0 28 2 2000 800"; do MFAS_PERSIST_TRACE=1 timeout 200 python tools/persist_check.py $cfg 2>&1 | grep -v amdgpu | grep -E "trace|step |ready|persist=" | tail -13; done
} > $out/defer_ab.log 2>&1
cat $out/defer_ab.log
