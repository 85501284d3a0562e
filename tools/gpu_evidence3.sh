#!/bin/bash
# Round-3 evidence visit (one box): the whole -m gpu suite, smoke, the bench lines (default, detector, Transformer lines at B = 32 / 64),
# the stage-isolated Transformer timings, kernel-trace stats and the PMC traffic passes of the default and the Transformer workloads.
# Outputs under gpurun_out/ev3; the summaries are copied into profiles/r03_* by hand.
cd $GRAFT_REPO_ROOT
out=gpurun_out/ev3
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1 || { tail -60 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
timeout -k 10 200 python __graft_entry__.py --smoke > $out/smoke.log 2>&1 || { tail -20 $out/smoke.log; exit 1; }
tail -3 $out/smoke.log
timeout -k 10 400 python bench.py --layers-out $out/layers_full.json > $out/bench_full.json 2> $out/bench_full.err || { tail -5 $out/bench_full.err; exit 1; }
cut -c1-260 $out/bench_full.json
timeout -k 10 300 python bench.py --workload detector --cpu-seconds 0 --layers-out $out/layers_det.json > $out/bench_det.json 2> $out/bench_det.err || { tail -5 $out/bench_det.err; exit 1; }
cut -c1-200 $out/bench_det.json
timeout -k 10 300 python tools/trocr_stage_bench.py > $out/trocr_stages.log 2>&1 || { tail -5 $out/trocr_stages.log; exit 1; }
tail -3 $out/trocr_stages.log
timeout -k 10 500 python bench.py --recognizer trocr --steps 8 --warmup 2 --cpu-seconds 0 > $out/bench_r18_trocr_b32.json 2> $out/bench_r18_trocr_b32.err || { tail -5 $out/bench_r18_trocr_b32.err; exit 1; }
cut -c1-200 $out/bench_r18_trocr_b32.json
timeout -k 10 500 python bench.py --recognizer trocr --batch 64 --steps 6 --warmup 2 --cpu-seconds 0 --sustain-seconds 0 > $out/bench_r18_trocr_b64.json 2> $out/bench_r18_trocr_b64.err || { tail -5 $out/bench_r18_trocr_b64.err; exit 1; }
cut -c1-200 $out/bench_r18_trocr_b64.json
timeout -k 10 900 python bench.py --backbone resnet50 --recognizer trocr --mixed --steps 8 --warmup 2 --cpu-seconds 12 > $out/bench_cfg4_b32.json 2> $out/bench_cfg4_b32.err || { tail -5 $out/bench_cfg4_b32.err; exit 1; }
cut -c1-200 $out/bench_cfg4_b32.json
timeout -k 10 600 python bench.py --backbone resnet50 --recognizer trocr --mixed --batch 64 --steps 6 --warmup 2 --cpu-seconds 0 --sustain-seconds 0 > $out/bench_cfg4_b64.json 2> $out/bench_cfg4_b64.err || { tail -5 $out/bench_cfg4_b64.err; exit 1; }
cut -c1-200 $out/bench_cfg4_b64.json
timeout -k 10 300 python bench.py --backbone resnet50 --workload detector --cpu-seconds 0 --sustain-seconds 0 --layers-out $out/layers_r50.json > $out/bench_r50_det.json 2> $out/bench_r50_det.err || { tail -5 $out/bench_r50_det.err; exit 1; }
cut -c1-160 $out/bench_r50_det.json
timeout -k 10 300 python bench.py --upload --cpu-seconds 0 > $out/bench_full_upload.json 2> $out/bench_full_upload.err || { tail -5 $out/bench_full_upload.err; exit 1; }
cut -c1-160 $out/bench_full_upload.json
VTD_TROCR_MERGE=0 timeout -k 10 500 python bench.py --recognizer trocr --steps 8 --warmup 2 --cpu-seconds 0 --sustain-seconds 0 > $out/bench_r18_trocr_b32_unmerged.json 2> $out/bench_r18_trocr_b32_unmerged.err || { tail -5 $out/bench_r18_trocr_b32_unmerged.err; exit 1; }
cut -c1-160 $out/bench_r18_trocr_b32_unmerged.json
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats -o run -- python3 $R/bench.py --cpu-seconds 0 --sustain-seconds 0 --no-profile > $R/$out/stats.log 2>&1 || { tail -5 $R/$out/stats.log; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats_trocr -o run -- python3 $R/bench.py --recognizer trocr --steps 4 --warmup 1 --cpu-seconds 0 --sustain-seconds 0 --no-profile > $R/$out/stats_trocr.log 2>&1 || { tail -5 $R/$out/stats_trocr.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$out/pmc_fetch -o run -- python3 $R/bench.py --cpu-seconds 0 --sustain-seconds 0 --no-profile --steps 4 --warmup 1 > $R/$out/pmc_fetch.log 2>&1 || { tail -5 $R/$out/pmc_fetch.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$out/pmc_write -o run -- python3 $R/bench.py --cpu-seconds 0 --sustain-seconds 0 --no-profile --steps 4 --warmup 1 > $R/$out/pmc_write.log 2>&1 || { tail -5 $R/$out/pmc_write.log; exit 1; }
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$out/pmc_fetch_trocr -o run -- python3 $R/bench.py --recognizer trocr --cpu-seconds 0 --sustain-seconds 0 --no-profile --steps 2 --warmup 0 > $R/$out/pmc_fetch_trocr.log 2>&1 || { tail -5 $R/$out/pmc_fetch_trocr.log; exit 1; }
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$out/pmc_write_trocr -o run -- python3 $R/bench.py --recognizer trocr --cpu-seconds 0 --sustain-seconds 0 --no-profile --steps 2 --warmup 0 > $R/$out/pmc_write_trocr.log 2>&1 || { tail -5 $R/$out/pmc_write_trocr.log; exit 1; }
cd $R
python tools/pmc_summary.py $out/pmc_fetch $out/pmc_write $out/pmc_traffic_per_launch.json
python tools/pmc_summary.py $out/pmc_fetch_trocr $out/pmc_write_trocr $out/pmc_traffic_per_launch_trocr.json
python tools/hbm_table.py $out/pmc_traffic_per_launch.json $(ls $out/stats/*kernel_trace.csv | head -1) $out/layers_det.json $out/hbm_bound_kernels.json || true
ls $out | head -40
