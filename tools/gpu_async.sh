#!/bin/bash
# Recogniser passes on a worker thread (VTD_TROCR_ASYNC=1) against passes in the submitting thread: parity test, then the two Transformer
# lines on one box, alternating
cd $GRAFT_REPO_ROOT
out=gpurun_out/async
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_trocr.py -x -q -m gpu -k "worker_thread or queued_tickets or encoder_pass_beside" > $out/pytest.log 2>&1 || { tail -40 $out/pytest.log; exit 1; }
tail -1 $out/pytest.log
for rep in 1 2; do
for a in 0 1; do
  VTD_TROCR_ASYNC=$a timeout -k 10 600 python bench.py --recognizer trocr --steps 48 --warmup 36 --cpu-seconds 0 --sustain-seconds 0 --no-profile > $out/r18_$a$rep.json 2> $out/r18_$a$rep.err || { tail -20 $out/r18_$a$rep.err; exit 1; }
  python - <<PY
import json
b=json.load(open("$out/r18_$a$rep.json")); print("ResNet-18 + Transformer, async=$a rep $rep: %.1f frames/s %.1f ms/step" % (b["value"], b["ms_per_step"]))
PY
done
done
for a in 0 1; do
  VTD_TROCR_ASYNC=$a timeout -k 10 900 python bench.py --backbone resnet50 --recognizer trocr --mixed --steps 48 --warmup 36 --cpu-seconds 0 --sustain-seconds 0 --no-profile > $out/cfg4_$a.json 2> $out/cfg4_$a.err || { tail -20 $out/cfg4_$a.err; exit 1; }
  python - <<PY
import json
b=json.load(open("$out/cfg4_$a.json")); print("configs[4], async=$a: %.1f frames/s %.1f ms/step" % (b["value"], b["ms_per_step"]))
PY
done
