#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r2c
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_trocr.py -x -q -m gpu -s > $out/pytest_trocr.log 2>&1 || { tail -30 $out/pytest_trocr.log; exit 1; }
tail -4 $out/pytest_trocr.log
timeout -k 10 200 python __graft_entry__.py --smoke > $out/smoke.log 2>&1 || { tail -20 $out/smoke.log; exit 1; }
tail -4 $out/smoke.log
timeout -k 10 300 python bench.py --recognizer trocr --steps 3 --warmup 1 --cpu-seconds 0 --no-profile > $out/bench_trocr.json 2> $out/bench_trocr.err || tail -20 $out/bench_trocr.err
cut -c1-200 $out/bench_trocr.json
timeout -k 10 300 python bench.py --backbone resnet50 --recognizer trocr --mixed --steps 3 --warmup 1 --cpu-seconds 0 --no-profile > $out/bench_cfg4.json 2> $out/bench_cfg4.err || tail -20 $out/bench_cfg4.err
cut -c1-200 $out/bench_cfg4.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_trocr -o run -- python3 bench.py --recognizer trocr --steps 2 --warmup 1 --cpu-seconds 0 --no-profile > $out/stats_trocr.log 2>&1
head -8 $out/stats_trocr/run_kernel_stats.csv | cut -c1-150
