#!/bin/bash
# do the dev pass (own stream) and the next epoch's resident launch really overlap?  kernel trace: start / end of k_president and k_eval
out=gpurun_out/s5; mkdir -p $out; export TMPDIR=/tmp
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/ovl -o ovl -- python $GRAFT_REPO_ROOT/tools/persist_check.py 16 20 0 22 5 10000 5600 toggle=MFAS_UNUSED persist > $GRAFT_REPO_ROOT/$out/ovl_run.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $out/ovl -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY' | tee $out/overlap_trace.log
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith(("void k_president", "void k_eval"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[-16:]:
    print(f'{r["Kernel_Name"][5:22]:18s} queue {r.get("Queue_Id","?"):>3s} stream {r.get("Stream_Id","?"):>3s} start {(int(r["Start_Timestamp"])-t0)/1e3:10.1f} us  end {(int(r["End_Timestamp"])-t0)/1e3:10.1f} us  dur {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:8.1f}')
PY
