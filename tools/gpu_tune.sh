#!/bin/bash
# GPU-box visit: build the kernel-selection table, install it in the package copy on the box, run the tuning tests and a bench.
set -e
cd $GRAFT_REPO_ROOT
out=gpurun_out/tune
mkdir -p $out
timeout -k 10 700 python tools/tune_table.py --out $out/gfx950.txt > $out/tune.log 2>&1 || { tail -30 $out/tune.log; exit 1; }
tail -12 $out/tune.log
mkdir -p video-text-detection-system_amd/vtd_amd/tuning
cp $out/gfx950.txt video-text-detection-system_amd/vtd_amd/tuning/gfx950.txt
timeout -k 10 400 python -m pytest tests/test_gpu_tuning.py -x -q -m gpu > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
timeout -k 10 300 python bench.py --layers-out $out/layers_full.json > $out/bench_full.json 2> $out/bench_full.err || { tail -20 $out/bench_full.err; exit 1; }
cat $out/bench_full.json
VTD_HALO_STAMPS=1 timeout -k 10 200 python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-profile > $out/stamps.json 2> $out/stamps.err || true
grep stamps $out/stamps.err | tail -4
