#!/bin/bash
# GPU-box visit: add the launch slots the shipped kernel-selection table lacks (tools/tune_table.py --keep-existing), then the whole GPU
# gate with the new table installed in the box's copy: tuning tests, every -m gpu test, smoke, default + detector bench lines.
cd $GRAFT_REPO_ROOT
out=gpurun_out/tune
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python tools/tune_table.py --keep-existing --insitu-steps 0 --out $out/gfx950.txt > $out/tune.log 2>&1 || { tail -30 $out/tune.log; exit 1; }
tail -4 $out/tune.log
cp $out/gfx950.txt video-text-detection-system_amd/vtd_amd/tuning/gfx950.txt
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1 || { tail -60 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
timeout -k 10 200 python __graft_entry__.py --smoke > $out/smoke.log 2>&1 || { tail -20 $out/smoke.log; exit 1; }
tail -2 $out/smoke.log
timeout -k 10 300 python bench.py --layers-out $out/layers_full.json > $out/bench_full.json 2> $out/bench_full.err || { tail -20 $out/bench_full.err; exit 1; }
cut -c1-300 $out/bench_full.json
timeout -k 10 300 python bench.py --workload detector --cpu-seconds 0 --layers-out $out/layers_det.json > $out/bench_det.json 2> $out/bench_det.err || { tail -5 $out/bench_det.err; exit 1; }
cut -c1-200 $out/bench_det.json
