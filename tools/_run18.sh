python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for cfg in "--R 16 --batch 20 --no-bn" "--R 128 --batch 16"; do
MFAS_LIB=$PWD/mfas_amd/csrc/libmfas_ct.so python bench.py --no-cpu-baseline --steps 1 --warmup 0 --pop 6 --epochs 1 $cfg 2>&1 | grep "chain timing" | cut -c1-330
done
python tools/popsweep.py 6,16,50,128,512 --R 16 --batch 20 --no-bn
python tools/popsweep.py 6,16,32,64,128
python bench.py | tail -1 | cut -c1-600
