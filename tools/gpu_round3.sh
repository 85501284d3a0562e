#!/bin/bash
# Round-2 evidence visit: default bench (both workloads), sustained 4000-step run with the shader clock sampled beside it, kernel-trace
# stats and the two PMC traffic passes of the same command.  Outputs under gpurun_out/r2b.
cd $GRAFT_REPO_ROOT
out=gpurun_out/r2b
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python bench.py --layers-out $out/layers_full.json > $out/bench_full.json 2> $out/bench_full.err || tail -5 $out/bench_full.err
cut -c1-400 $out/bench_full.json
timeout -k 10 300 python bench.py --workload detector --cpu-seconds 0 --layers-out $out/layers_det.json > $out/bench_det.json 2> $out/bench_det.err
cut -c1-200 $out/bench_det.json
# sustained: ~12 s of back-to-back steps; clock sampled twice a second from sysfs / rocm-smi
( for i in $(seq 1 60); do date +%s.%N; cat /sys/class/drm/card*/device/pp_dpm_sclk 2>/dev/null | grep '\*'; rocm-smi --showclocks 2>/dev/null | grep -i sclk | head -1; sleep 0.5; done ) > $out/clock_samples.txt 2>&1 &
sampler=$!
timeout -k 10 400 python bench.py --steps 4000 --warmup 20 --cpu-seconds 0 --no-profile > $out/bench_sustained.json 2> $out/bench_sustained.err
kill $sampler 2>/dev/null
cut -c1-300 $out/bench_sustained.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python3 bench.py --cpu-seconds 0 --no-profile > $out/stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o run -- python3 bench.py --cpu-seconds 0 --no-profile --steps 4 --warmup 1 > $out/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o run -- python3 bench.py --cpu-seconds 0 --no-profile --steps 4 --warmup 1 > $out/pmc_write.log 2>&1
python tools/pmc_summary.py $out/pmc_fetch $out/pmc_write $out/pmc_traffic_per_launch.json
ls $out/stats | head -5
