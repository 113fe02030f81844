#!/bin/bash
# round 5, first GPU pass of the session: default bench line, rocprofv3 kernel statistics of the same command, R=16 / R=128 small-population
# sweeps, chain phase stamps (timing build), search-default full-size tests.
out=gpurun_out/r05; mkdir -p $out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); print(g.build_variant('timing', ['-DMFAS_CHAIN_TIMING']))" > $out/build.log 2>&1
timeout 900 python bench.py > $out/bench_pop128.log 2> $out/bench_pop128.err; echo "bench rc=$?"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rp_bench -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-small-pop > $GRAFT_REPO_ROOT/$out/rp_bench.log 2>&1)
{ timeout 400 python tools/popsweep.py 16 20 0 10 1,4,6,8,12,16,24,28
  timeout 400 python tools/popsweep.py 16 20 0 10 6,16,28 mixed
  timeout 600 python tools/popsweep.py 128 16 1 10 1,3,6,8,16; } 2>&1 | grep -v amdgpu > $out/popsweep.log
bash tools/archive/r04_chain_phases.sh > $out/chain_phases.log 2>&1
timeout 1500 python -m pytest tests/test_fullsize.py -q -x -m gpu -k "search_default" 2>&1 | tail -5 > $out/search_default_tests.log
find $out -name "*kernel_stats.csv" | head -3
cat $out/popsweep.log; cat $out/chain_phases.log; cat $out/search_default_tests.log; head -c 1500 $out/bench_pop128.log
