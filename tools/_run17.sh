python -m pytest tests -x -q -m gpu 2>&1 | tail -4
for cfg in "--R 16 --batch 20 --no-bn" "--R 16 --batch 16 --no-bn"; do
echo "== $cfg"
MFAS_LIB=$PWD/mfas_amd/csrc/libmfas_ct.so python bench.py --no-cpu-baseline --steps 1 --warmup 0 --pop 6 --epochs 1 $cfg 2>&1 | grep "chain timing" | cut -c1-330
for K in 6 24 50 100; do for lean in 0 1; do
  if [ $lean = 0 ]; then export MFAS_NO_LEAN_CHAIN=1; else unset MFAS_NO_LEAN_CHAIN; fi
  python bench.py --no-cpu-baseline --steps 1 --warmup 1 --pop $K --epochs 4 $cfg 2>&1 | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('K',$K,'lean',$lean, round(d['value'],1), round(d['roofline']['avg_launch_us'],1), round(d['config']['mean_best_dev_acc'],4))"
done; done
unset MFAS_NO_LEAN_CHAIN
done
