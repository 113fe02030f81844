#!/bin/bash
# SQ counters of the dev-evaluation kernel in the c3 workload (separate --pmc passes, no tracing): where do k_eval's cycles go?
# usage (GPU box): bash tools/pmc_eval.sh > gpurun_out/pmc_eval.log
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  rm -rf /tmp/pmc_eval_$i
  timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_eval_$i -- python $R/bench.py --workload c3 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
  python - "$i" <<'PY'
import csv, glob, sys, collections
i = sys.argv[1]
tot = collections.defaultdict(float); n = 0
for f in glob.glob(f"/tmp/pmc_eval_{i}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_eval" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n += 1
print({k: v for k, v in tot.items()}, "rows", n)
PY
done
