#!/bin/bash
# GPU-box visit: whole parity suite, configs[4] bench (R50 + TrOCR, mixed 720p/1080p), TrOCR on the configs[2] frames, kernel stats.
cd $GRAFT_REPO_ROOT
out=gpurun_out/r2a
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 800 python -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
timeout -k 10 300 python bench.py --backbone resnet50 --recognizer trocr --mixed --steps 3 --warmup 1 --cpu-seconds 0 --no-profile > $out/bench_cfg4.json 2> $out/bench_cfg4.err || tail -20 $out/bench_cfg4.err
cat $out/bench_cfg4.json
timeout -k 10 300 python bench.py --recognizer trocr --steps 3 --warmup 1 --cpu-seconds 0 --no-profile > $out/bench_trocr.json 2> $out/bench_trocr.err || tail -20 $out/bench_trocr.err
cat $out/bench_trocr.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_trocr -o run -- python3 bench.py --recognizer trocr --steps 2 --warmup 1 --cpu-seconds 0 --no-profile > $out/stats_trocr.log 2>&1
ls $out/stats_trocr/* | head
