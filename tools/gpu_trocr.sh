#!/bin/bash
# TrOCR benches: ResNet-18 + TrOCR on the configs[2] frames, and BASELINE configs[4] (ResNet-50 + TrOCR, mixed 720p / 1080p)
cd $GRAFT_REPO_ROOT
out=gpurun_out/trocr
mkdir -p $out
export TMPDIR=/tmp
VTD_AUTOTUNE_VERBOSE=1 timeout -k 10 400 python bench.py --recognizer trocr --steps 3 --warmup 1 --cpu-seconds 0 --no-profile > $out/bench_trocr.json 2> $out/bench_trocr.err || { tail -20 $out/bench_trocr.err; exit 1; }
cut -c1-200 $out/bench_trocr.json
grep -c "cfg 15\|cfg 13\|cfg 12\|cfg 14" $out/bench_trocr.err
timeout -k 10 400 python bench.py --backbone resnet50 --recognizer trocr --mixed --steps 3 --warmup 1 --cpu-seconds 0 --no-profile > $out/bench_cfg4.json 2> $out/bench_cfg4.err || { tail -20 $out/bench_cfg4.err; exit 1; }
cut -c1-200 $out/bench_cfg4.json
