for cfg in "--R 16 --batch 20 --no-bn" "--R 16 --batch 16 --no-bn" "--R 128 --batch 16"; do
echo "== $cfg"
MFAS_LIB=$PWD/mfas_amd/csrc/libmfas_ct.so python bench.py --no-cpu-baseline --steps 1 --warmup 0 --pop 6 --epochs 1 $cfg 2>&1 | grep "chain timing\|metric" | cut -c1-330
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r16/prof -o s -- python bench.py --no-cpu-baseline --steps 1 --warmup 0 --R 16 --batch 20 --no-bn --pop 6 --epochs 2 > gpurun_out/r16.log 2>&1
head -4 gpurun_out/r16/prof/*kernel_stats.csv | cut -c1-150
find gpurun_out/r16 -name "*kernel_trace.csv" -delete
