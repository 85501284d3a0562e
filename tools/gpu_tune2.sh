#!/bin/bash
set -e
cd $GRAFT_REPO_ROOT
out=gpurun_out/tune2
mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_gpu_tuning.py -q -m gpu > $out/pytest.log 2>&1 || tail -40 $out/pytest.log
tail -3 $out/pytest.log
timeout -k 10 300 python bench.py --layers-out $out/layers_full.json > $out/bench_full.json 2> $out/bench_full.err || { tail -20 $out/bench_full.err; exit 1; }
cat $out/bench_full.json
VTD_BENCH_CRNN=default timeout -k 10 300 python bench.py --cpu-seconds 0 --no-profile > $out/bench_defcrnn.json 2> $out/bench_defcrnn.err || true
cat $out/bench_defcrnn.json
timeout -k 10 300 python bench.py --workload detector --cpu-seconds 0 --no-profile > $out/bench_det.json 2> $out/bench_det.err || true
cat $out/bench_det.json
VTD_TUNING=0 VTD_AUTOTUNE_VERBOSE=1 timeout -k 10 300 python bench.py --cpu-seconds 0 --no-profile > $out/bench_notable.json 2> $out/bench_notable.err || true
cat $out/bench_notable.json
VTD_HALO_STAMPS=1 VTD_FORCE_CLASSED_CFG=103 VTD_TUNING=0 timeout -k 10 200 python bench.py --workload detector --steps 2 --warmup 1 --cpu-seconds 0 --no-profile > $out/stamps.json 2> $out/stamps.err || true
grep stamps $out/stamps.err | tail -4
