#!/bin/bash
# round 3: merged recogniser passes: TrOCR + configs + pipeline tests, Transformer bench lines, PCIe-inclusive default line
cd $GRAFT_REPO_ROOT
out=gpurun_out/r3j
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_trocr.py tests/test_gpu_configs.py tests/test_gpu_pipeline.py -x -q -m gpu > $out/pytest.log 2>&1 || { tail -60 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
line() {
python -c "
import json; b=json.load(open('$1')); r=b['roofline']
print('$2: %.1f frames/s  %.1f ms/step  crops/step %.0f | cross-attn %.1f us avg, %.0f GB/s, rows/launch %.1f' % (b['value'], b['ms_per_step'], b['config']['crops_recognized_per_step_rank0'], r['avg_launch_us'], r['achieved'], r['avg_live_rows_per_launch']))"
}
for m in 1 0; do
VTD_TROCR_MERGE=$m VTD_TROCR_MAX_CROPS=1024 timeout -k 10 500 python bench.py --recognizer trocr --steps 8 --warmup 2 --cpu-seconds 0 --sustain-seconds 0 > $out/b32_$m.json 2> $out/b32_$m.err || { tail -20 $out/b32_$m.err; exit 1; }
line $out/b32_$m.json "r18+trocr B=32 merge=$m"
done
VTD_TROCR_MAX_CROPS=1024 timeout -k 10 500 python bench.py --backbone resnet50 --recognizer trocr --mixed --steps 8 --warmup 2 --cpu-seconds 0 --sustain-seconds 0 > $out/cfg4.json 2> $out/cfg4.err || { tail -20 $out/cfg4.err; exit 1; }
line $out/cfg4.json "cfg4 r50+trocr mixed B=32 merged"
VTD_TROCR_MAX_CROPS=2048 timeout -k 10 600 python bench.py --recognizer trocr --batch 64 --steps 6 --warmup 2 --cpu-seconds 0 --sustain-seconds 0 > $out/b64.json 2> $out/b64.err || { tail -20 $out/b64.err; exit 1; }
line $out/b64.json "r18+trocr B=64 merged"
timeout -k 10 300 python bench.py --upload --cpu-seconds 0 > $out/upload.json 2> $out/upload.err || { tail -5 $out/upload.err; exit 1; }
python -c "
import json; b=json.load(open('$out/upload.json')); print('default line with per-step PCIe upload: %.0f frames/s, sustained %.0f' % (b['value'], b['sustained']['value']))"
timeout -k 10 300 python bench.py --upload --workload detector --cpu-seconds 0 > $out/upload_det.json 2> $out/upload_det.err || { tail -5 $out/upload_det.err; exit 1; }
python -c "
import json; b=json.load(open('$out/upload_det.json')); print('detector line with per-step PCIe upload: %.0f frames/s, sustained %.0f' % (b['value'], b['sustained']['value']))"
