#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/two
mkdir -p $out
for tw in 0 1 0 1; do
  VTD_BENCH_TWO_ENGINES=$tw timeout -k 10 200 python bench.py --workload detector --cpu-seconds 0 --no-profile --steps 40 --warmup 4 > $out/b$tw.json 2> $out/b$tw.err || { tail -5 $out/b$tw.err; exit 1; }
  python - <<PY
import json
b=json.load(open("$out/b$tw.json"))
print("two engines $tw: %.0f frames/s %.3f ms" % (b["value"], b["ms_per_step"]))
PY
done
