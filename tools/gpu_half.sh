#!/bin/bash
# head_entry_half (cfg 107) against head_entry_halo256 (cfg 103): parity on every head-entry candidate, stamps, bench lines (one box)
cd $GRAFT_REPO_ROOT
out=gpurun_out/half
mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_detector.py -x -q -m gpu -k "composed_head_entry or fused_fpn or r50" > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
for cfg in 107 103; do
VTD_HALO_STAMPS=1 VTD_FORCE_CLASSED_CFG=$cfg timeout -k 10 200 python bench.py --workload detector --steps 2 --warmup 1 --cpu-seconds 0 --no-profile > $out/st$cfg.json 2> $out/st$cfg.err || { tail -5 $out/st$cfg.err; exit 1; }
grep "head_entry.*stamps" $out/st$cfg.err | tail -1
done
for cfg in 107 103 107 103; do
  VTD_FORCE_CLASSED_CFG=$cfg timeout -k 10 200 python bench.py --cpu-seconds 0 > $out/bench_$cfg.json 2> $out/bench_$cfg.err || { tail -5 $out/bench_$cfg.err; exit 1; }
  python - <<PY
import json
b=json.load(open("$out/bench_$cfg.json")); r=b["roofline"]
print("cfg $cfg: %.0f frames/s | %s | in situ %.1f us frac %.3f | alone %.1f us frac %.3f" % (b["value"], r["kernel"][:24], r["avg_launch_us"], r["frac"], r["alone_on_gpu"]["avg_launch_us"], r["alone_on_gpu"]["frac"]))
PY
done
