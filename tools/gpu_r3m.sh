#!/bin/bash
# cross-attention with ten rounds of loads in flight: TrOCR parity tests, stage times, the two Transformer lines
cd $GRAFT_REPO_ROOT
out=gpurun_out/r3m
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_trocr.py -x -q -m gpu > $out/pytest.log 2>&1 || { tail -60 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
timeout -k 10 300 python tools/trocr_stage_bench.py > $out/stage.txt 2>&1 || { tail -20 $out/stage.txt; exit 1; }
tail -3 $out/stage.txt
line() {
python -c "
import json; b=json.load(open('$1')); r=b['roofline']
print('$2: %.1f frames/s  %.1f ms/step  crops/step %.0f | cross-attn %.1f us avg, %.0f GB/s, rows/launch %.1f' % (b['value'], b['ms_per_step'], b['config']['crops_recognized_per_step_rank0'], r['avg_launch_us'], r['achieved'], r['avg_live_rows_per_launch']))"
}
timeout -k 10 500 python bench.py --recognizer trocr --steps 8 --warmup 2 --cpu-seconds 0 --sustain-seconds 0 > $out/b32.json 2> $out/b32.err || { tail -20 $out/b32.err; exit 1; }
line $out/b32.json "r18+trocr B=32"
timeout -k 10 500 python bench.py --backbone resnet50 --recognizer trocr --mixed --steps 8 --warmup 2 --cpu-seconds 0 --sustain-seconds 0 > $out/cfg4.json 2> $out/cfg4.err || { tail -20 $out/cfg4.err; exit 1; }
line $out/cfg4.json "cfg4 r50+trocr mixed B=32"
