#!/bin/bash
# Round 6, VERDICT item 7: more reference seeds for G19b (configs[2], 64 -> 128 calls) and G15 (configs[1], 256 -> 1024 seeds) so
# that 3 s.e. + 0.1 % <= 0.2 % top-1.  Build container only (imports /root/reference).  NPROC worker processes, 1 BLAS thread each.
# Afterwards: python tests/golden/make_golden.py g19bmerge; python tests/golden/make_golden.py g15merge
cd "$(dirname "$0")/.." || exit 1
NPROC=${NPROC:-6}
mkdir -p gpurun_out/golden_logs
{
  for a in $(seq 64 8 120); do echo "G19_SEED0=$a G19_NS=8 python tests/golden/make_golden.py g19b > gpurun_out/golden_logs/g19b_$a.log 2>&1"; done
  for a in $(seq 256 16 1008); do echo "G15_SEED0=$a G15_NS=16 G15_THREADS=1 python tests/golden/make_golden.py g15 >> gpurun_out/golden_logs/g15_$a.log 2>&1"; done
} | PYTHONDONTWRITEBYTECODE=1 OMP_NUM_THREADS=1 MKL_NUM_THREADS=1 nice -n 19 xargs -P "$NPROC" -I{} bash -c "{}"
