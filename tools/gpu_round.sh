#!/bin/bash
# One GPU-box visit: parity suite, default bench, kernel-trace stats, two PMC passes.  Outputs under gpurun_out/$1.
set -e
tag=${1:-run}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
timeout -k 10 200 python bench.py --layers-out $out/layers_full.json > $out/bench_full.json 2> $out/bench_full.err
cat $out/bench_full.json
timeout -k 10 200 python bench.py --workload detector --cpu-seconds 0 --layers-out $out/layers_det.json > $out/bench_det.json 2> $out/bench_det.err
cat $out/bench_det.json
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python3 bench.py --cpu-seconds 0 --no-profile > $out/stats.log 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o run -- python3 bench.py --cpu-seconds 0 --no-profile --steps 4 --warmup 1 > $out/pmc_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o run -- python3 bench.py --cpu-seconds 0 --no-profile --steps 4 --warmup 1 > $out/pmc_write.log 2>&1
ls $out/stats
