out=gpurun_out/rounds; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_mirror.py -x -q -m gpu -k "large_share or per_candidate or signature" > $out/t_mirror.log 2>&1; echo "rc=$?" >> $out/t_mirror.log
timeout 600 python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_c5.log 2>&1
timeout 600 python bench.py --R 16 --no-bn --batch 20 --pop 128 --steps 2 --warmup 1 --no-cpu-baseline --no-small-pop > $out/bench_r16_pop128.log 2>&1
timeout 600 python bench.py --R 16 --no-bn --batch 20 --pop 512 --steps 1 --warmup 1 --no-cpu-baseline --no-small-pop > $out/bench_r16_pop512.log 2>&1
timeout 600 python bench.py --R 16 --no-bn --batch 20 --pop 64 --steps 2 --warmup 1 --no-cpu-baseline --no-small-pop > $out/bench_r16_pop64.log 2>&1
MFAS_NO_ROUNDS=1 timeout 600 python bench.py --R 16 --no-bn --batch 20 --pop 64 --steps 2 --warmup 1 --no-cpu-baseline --no-small-pop > $out/bench_r16_pop64_norounds.log 2>&1
tail -n 3 $out/t_mirror.log; for f in $out/bench_*.log; do echo "$f $(grep -o '"value": [0-9.]*' $f | head -1) $(grep -o '"other_order": {[^}]*}' $f | head -1)"; done
