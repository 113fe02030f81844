#!/bin/bash
# deferred unit hand-off (persist.hip.h, two-unit workgroups; opt-in) A/B on one box: persist=0 lines = MFAS_RES_DEFER=1 (k_president<..., DEFER>), persist=1 lines = the default kernels
out=gpurun_out/s5; mkdir -p $out
{
for cfg in "16 20 0 28 10 10000 5600" "16 20 0 22 10 10000 5600" "16 20 0 28 10 10000 5600 mixed" "16 20 0 16 10 10000 5600"; do
    echo "## $cfg  (persist=0 lines: MFAS_RES_DEFER=1 = deferred hand-off, persist=1 lines: default)"
    timeout 300 python tools/persist_check.py $cfg toggle=MFAS_RES_DEFER persist 2>&1 | grep -E "persist=|IDENT|MISM"
done
echo "## short last batch (N = 4010: 10 rows), 3 epochs"
timeout 300 python tools/persist_check.py 16 20 0 28 3 4010 800 toggle=MFAS_RES_DEFER persist 2>&1 | grep -E "persist=|IDENT|MISM"
timeout 300 python tools/persist_check.py 16 20 0 24 3 4010 800 mixed 2>&1 | grep -E "persist=|IDENT|MISM"
echo "## trace"
for cfg in "16 20 0 28 2 2000 800"; do MFAS_PERSIST_TRACE=1 timeout 200 python tools/persist_check.py $cfg 2>&1 | grep -v amdgpu | grep -E "trace|step |ready|persist=" | tail -13; done
} > $out/defer_ab.log 2>&1
cat $out/defer_ab.log
