out=gpurun_out/dma; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "persistent or resident or loaded_library or two_processes" > $out/t_parity.log 2>&1; echo "rc=$?" >> $out/t_parity.log
timeout 600 python -m pytest tests/test_gpu_mirror.py -x -q -m gpu -k "large_share or per_candidate or small_r_eval" > $out/t_mirror.log 2>&1; echo "rc=$?" >> $out/t_mirror.log
for rep in 1 2; do
  echo "# register staging (default build)"; timeout 600 python tools/popsweep.py 16 20 0 10 6,12,16,24,28 2>&1 | grep -v amdgpu
  echo "# LDS-DMA staging (-DMFAS_RES_DMA=1 variant)"; MFAS_LIB=$PWD/mfas_amd/csrc/libmfas_hip_dma.so timeout 600 python tools/popsweep.py 16 20 0 10 6,12,16,24,28 2>&1 | grep -v amdgpu
done > $out/popsweep_ab.log
timeout 300 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline --no-small-pop > $out/bench_c3_dma.log 2>&1
MFAS_LIB=$PWD/mfas_amd/csrc/libmfas_hip_dma.so timeout 300 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline --no-small-pop > $out/bench_c3_nodma.log 2>&1
timeout 300 python bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline --no-small-pop > $out/bench_c2_dma.log 2>&1
MFAS_LIB=$PWD/mfas_amd/csrc/libmfas_hip_dma.so timeout 300 python bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline --no-small-pop > $out/bench_c2_nodma.log 2>&1
tail -3 $out/t_parity.log $out/t_mirror.log; cat $out/popsweep_ab.log; for f in $out/bench_*.log; do echo $f; grep -o '"value": [0-9.]*' $f | head -1; done
