#!/bin/bash
# Round-4 evidence, second visit: the Transformer recogniser after the cross-attention moved onto the encoder states (trocr_xattn.hip).
# Stage times, the two Transformer lines (12 tickets per pass; 4; the reference's key / value form), kernel stats, PMC traffic passes and the
# ROCm library's GEMM on the encoder's shapes as a yardstick.  tools/collect_evidence4b.py copies the summaries into profiles/r04_*.
cd $GRAFT_REPO_ROOT
out=gpurun_out/ev4b
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 python tools/trocr_stage_bench.py > $out/trocr_stages.log 2>&1 || { tail -5 $out/trocr_stages.log; exit 1; }
tail -3 $out/trocr_stages.log
VTD_TROCR_XATTN=0 timeout -k 10 300 python tools/trocr_stage_bench.py > $out/trocr_stages_kv.log 2>&1 || { tail -5 $out/trocr_stages_kv.log; exit 1; }
tail -1 $out/trocr_stages_kv.log
timeout -k 10 200 python tools/lib_gemm_probe.py 272 > $out/lib_gemm.txt 2>&1 || { tail -5 $out/lib_gemm.txt; exit 1; }
grep TFLOP $out/lib_gemm.txt
VTD_BENCH_STAMPS=1 timeout -k 10 700 python bench.py --recognizer trocr --steps 36 --warmup 24 --cpu-seconds 0 > $out/bench_r18_trocr_b32.json 2> $out/bench_r18_trocr_b32.err || { tail -5 $out/bench_r18_trocr_b32.err; exit 1; }
cut -c1-200 $out/bench_r18_trocr_b32.json
VTD_TROCR_PASS_TICKETS=4 timeout -k 10 600 python bench.py --recognizer trocr --steps 16 --warmup 8 --cpu-seconds 0 --sustain-seconds 0 > $out/bench_r18_trocr_b32_t4.json 2> $out/bench_r18_trocr_b32_t4.err || { tail -5 $out/bench_r18_trocr_b32_t4.err; exit 1; }
cut -c1-200 $out/bench_r18_trocr_b32_t4.json
VTD_TROCR_XATTN=0 VTD_TROCR_PASS_TICKETS=4 VTD_TROCR_MAX_CROPS=1280 timeout -k 10 600 python bench.py --recognizer trocr --steps 16 --warmup 8 --cpu-seconds 0 --sustain-seconds 0 > $out/bench_r18_trocr_b32_kv.json 2> $out/bench_r18_trocr_b32_kv.err || { tail -5 $out/bench_r18_trocr_b32_kv.err; exit 1; }
cut -c1-200 $out/bench_r18_trocr_b32_kv.json
timeout -k 10 900 python bench.py --backbone resnet50 --recognizer trocr --mixed --steps 36 --warmup 24 --cpu-seconds 12 > $out/bench_cfg4_b32.json 2> $out/bench_cfg4_b32.err || { tail -5 $out/bench_cfg4_b32.err; exit 1; }
cut -c1-200 $out/bench_cfg4_b32.json
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats_trocr -o run -- python3 $R/bench.py --recognizer trocr --steps 24 --warmup 12 --cpu-seconds 0 --sustain-seconds 0 --no-profile > $R/$out/stats_trocr.log 2>&1 || { tail -5 $R/$out/stats_trocr.log; exit 1; }
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$out/pmc_fetch_trocr -o run -- python3 $R/bench.py --recognizer trocr --cpu-seconds 0 --sustain-seconds 0 --no-profile --steps 12 --warmup 0 > $R/$out/pmc_fetch_trocr.log 2>&1 || { tail -5 $R/$out/pmc_fetch_trocr.log; exit 1; }
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$out/pmc_write_trocr -o run -- python3 $R/bench.py --recognizer trocr --cpu-seconds 0 --sustain-seconds 0 --no-profile --steps 12 --warmup 0 > $R/$out/pmc_write_trocr.log 2>&1 || { tail -5 $R/$out/pmc_write_trocr.log; exit 1; }
cd $R
python tools/pmc_summary.py $out/pmc_fetch_trocr $out/pmc_write_trocr $out/pmc_traffic_per_launch_trocr.json
find $out -name "*kernel_trace.csv" -delete
rm -rf $out/pmc_fetch_trocr $out/pmc_write_trocr
ls $out | head -40
