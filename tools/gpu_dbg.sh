#!/bin/bash
# conv_igemm timing experiments (results are wrong on purpose): what the launches cost without their loads
cd $GRAFT_REPO_ROOT; export VTD_LIB_VARIANT=convexp
out=gpurun_out/dbg
mkdir -p $out
for d in 0 9 7; do
  VTD_CONV_DEBUG=$d timeout -k 10 200 python bench.py --workload detector --steps 4 --warmup 1 --cpu-seconds 0 --layers-out $out/layers$d.json > $out/b$d.json 2> $out/b$d.err || { tail -5 $out/b$d.err; exit 1; }
done
python - <<PY
import json
t=[json.load(open("$out/layers%d.json"%d)) for d in (0,9,7)]
for i,r in enumerate(t[0]):
    if r["calls"] and "igemm" in r["launch"]:
        print("%-62s" % r["launch"][:62], " ".join("%7.1f" % (1e3*t[d][i]["ms_total"]/max(t[d][i]["calls"],1)) for d in range(3)))
PY
