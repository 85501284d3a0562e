#!/bin/bash
# A/B of an instrumented library variant (VARIANT=<tag>, built with build_native.py under VTD_LIB_VARIANT) against the product library: one box
cd $GRAFT_REPO_ROOT
out=gpurun_out/variant
mkdir -p $out
for rep in 1 2; do
for v in product $VARIANT; do
  if [ $v = product ]; then unset VTD_LIB_VARIANT; else export VTD_LIB_VARIANT=$v; fi
  timeout -k 10 200 python bench.py --workload detector --steps 10 --warmup 2 --cpu-seconds 0 --layers-out $out/layers_$v$rep.json > $out/b_$v$rep.json 2> $out/b_$v$rep.err || { tail -5 $out/b_$v$rep.err; exit 1; }
  timeout -k 10 200 python bench.py --cpu-seconds 0 --no-profile > $out/f_$v$rep.json 2> $out/f_$v$rep.err || { tail -5 $out/f_$v$rep.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$out/b_$v$rep.json")); f=json.load(open("$out/f_$v$rep.json"))
rows=[r for r in json.load(open("$out/layers_$v$rep.json")) if r["calls"]]
ig=sum(1e3*r["ms_total"]/r["calls"] for r in rows if "igemm" in r["launch"])
print("$v rep $rep: detector %.0f frames/s | full %.0f frames/s | igemm launches %.0f us | all launches %.0f us" % (d["value"], f["value"], ig, sum(1e3*r["ms_total"]/r["calls"] for r in rows)))
PY
done
done
