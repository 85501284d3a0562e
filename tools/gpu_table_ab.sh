#!/bin/bash
# A/B of two kernel-selection tables on one box: the shipped one against gpurun_out/tune/new_table.txt (default line and detector line)
cd $GRAFT_REPO_ROOT
out=gpurun_out/table_ab
mkdir -p $out
for rep in 1 2; do
for t in shipped new; do
  if [ $t = shipped ]; then unset VTD_TUNING_FILE; else export VTD_TUNING_FILE=$GRAFT_REPO_ROOT/gpurun_out/tune/new_table.txt; fi
  timeout -k 10 200 python bench.py --cpu-seconds 0 --no-profile > $out/f_$t$rep.json 2> $out/f_$t$rep.err || { tail -5 $out/f_$t$rep.err; exit 1; }
  timeout -k 10 200 python bench.py --workload detector --cpu-seconds 0 --no-profile --sustain-seconds 3 > $out/d_$t$rep.json 2> $out/d_$t$rep.err || { tail -5 $out/d_$t$rep.err; exit 1; }
  python - <<PY
import json
f=json.loads(open("$out/f_$t$rep.json").read().strip().splitlines()[-1]); d=json.loads(open("$out/d_$t$rep.json").read().strip().splitlines()[-1])
print("$t rep $rep: full %.0f (sustained %.0f) | detector %.0f (sustained %.0f)" % (f["value"], f["sustained"]["value"], d["value"], d["sustained"]["value"]))
PY
done
done
