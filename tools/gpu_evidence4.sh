#!/bin/bash
# Round-4 evidence visit (one box): the whole -m gpu suite, smoke, the bench lines (default, detector, upload, 1080p, ResNet-50 detector),
# recogniser per-launch table, kernel-trace stats, PMC traffic passes of the default workload.  The Transformer lines: tools/gpu_evidence4b.sh.
# Outputs under gpurun_out/ev4; tools/collect_evidence4.py copies the summaries into profiles/r04_*.
cd $GRAFT_REPO_ROOT
out=gpurun_out/ev4
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
step() { echo "[ev4] $1" ; }
if [ "${EV4_SKIP_TESTS:-0}" != "1" ]; then
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1 || { tail -60 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
timeout -k 10 200 python __graft_entry__.py --smoke > $out/smoke.log 2>&1 || { tail -20 $out/smoke.log; exit 1; }
tail -3 $out/smoke.log
fi
timeout -k 10 400 python bench.py --layers-out $out/layers_full.json > $out/bench_full.json 2> $out/bench_full.err || { tail -5 $out/bench_full.err; exit 1; }
cut -c1-260 $out/bench_full.json
timeout -k 10 300 python bench.py --workload detector --cpu-seconds 0 --layers-out $out/layers_det.json > $out/bench_det.json 2> $out/bench_det.err || { tail -5 $out/bench_det.err; exit 1; }
cut -c1-200 $out/bench_det.json
timeout -k 10 300 python bench.py --upload --cpu-seconds 0 > $out/bench_full_upload.json 2> $out/bench_full_upload.err || { tail -5 $out/bench_full_upload.err; exit 1; }
cut -c1-160 $out/bench_full_upload.json
timeout -k 10 300 python bench.py --height 1080 --width 1920 --cpu-seconds 0 > $out/bench_full_1080p.json 2> $out/bench_full_1080p.err || { tail -5 $out/bench_full_1080p.err; exit 1; }
cut -c1-160 $out/bench_full_1080p.json
step "fusions off (same box)"
VTD_DETECTOR_OPTIONS=fuse_downsample=0 VTD_RECOGNIZER_OPTIONS=fuse_pools=0 VTD_TILE_HEIGHT_MODEL=0 timeout -k 10 300 python bench.py --cpu-seconds 0 --no-profile > $out/bench_full_unfused.json 2> $out/bench_full_unfused.err || { tail -5 $out/bench_full_unfused.err; exit 1; }
cut -c1-160 $out/bench_full_unfused.json
timeout -k 10 300 python bench.py --backbone resnet50 --workload detector --cpu-seconds 0 --sustain-seconds 0 --layers-out $out/layers_r50.json > $out/bench_r50_det.json 2> $out/bench_r50_det.err || { tail -5 $out/bench_r50_det.err; exit 1; }
cut -c1-160 $out/bench_r50_det.json
step "recogniser alone per launch"
REC_CROPS=272 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/rec272 -o run -- python3 tools/lstm_bench.py > $out/rec272.log 2>&1 || { tail -5 $out/rec272.log; exit 1; }
python tools/rec_layers.py $(ls $out/rec272/*kernel_trace.csv | head -1) 23 > $out/recognizer_launch_table.txt
tail -1 $out/recognizer_launch_table.txt
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats -o run -- python3 $R/bench.py --cpu-seconds 0 --sustain-seconds 0 --no-profile > $R/$out/stats.log 2>&1 || { tail -5 $R/$out/stats.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$out/pmc_fetch -o run -- python3 $R/bench.py --cpu-seconds 0 --sustain-seconds 0 --no-profile --steps 4 --warmup 1 > $R/$out/pmc_fetch.log 2>&1 || { tail -5 $R/$out/pmc_fetch.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$out/pmc_write -o run -- python3 $R/bench.py --cpu-seconds 0 --sustain-seconds 0 --no-profile --steps 4 --warmup 1 > $R/$out/pmc_write.log 2>&1 || { tail -5 $R/$out/pmc_write.log; exit 1; }
cd $R
python tools/pmc_summary.py $out/pmc_fetch $out/pmc_write $out/pmc_traffic_per_launch.json
python tools/hbm_table.py $out/pmc_traffic_per_launch.json $(ls $out/stats/*kernel_trace.csv | head -1) $out/layers_det.json $out/hbm_bound_kernels.json || true
ls $out | head -60
