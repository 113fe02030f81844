#!/bin/bash
# k_eval<4,1,1,true> (R <= 16, bf16 rows) at 4 (default) / 5 / 6 workgroups per CU (launch bounds; 112 / 96 / 80 VGPRs, 0 / 7 / 19 spilled) + final-tree checks
out=gpurun_out/s5; mkdir -p $out
{ for lib in "" tools/tmp_occ/libmfas_occ5.so tools/tmp_occ/libmfas_occ6.so; do
    echo "## ${lib:-default library}"
    for K in 16 28 7; do MFAS_LIB=${lib:+$PWD/$lib} timeout 200 python tools/evalbench.py 16 $K 2>&1 | grep -v amdgpu; done
  done; } > $out/eval_occ.log 2>&1
cat $out/eval_occ.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "loaded_library or deferred_handoff or persistent_resident or eval_forward" 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
