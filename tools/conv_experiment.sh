#!/bin/bash
# timing-only experiment: how much of each convolution's time is the A-operand gather / the loads at all
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp
for d in 0 1 2 3; do
  VTD_CONV_DEBUG=$d timeout -k 10 200 python bench.py --workload detector --cpu-seconds 0 --layers-out gpurun_out/exp/layers_dbg$d.json > gpurun_out/exp/bench_dbg$d.json 2> gpurun_out/exp/err$d.log || true
done
python - <<'PY'
import json
t=[json.load(open(f'gpurun_out/exp/layers_dbg{d}.json')) for d in range(4)]
for i,r in enumerate(t[0]):
    if not r['calls']: continue
    print(f"{r['launch'][:66]:66s}", *[f"{x[i]['ms_total']/x[i]['calls']*1e3:8.1f}" for x in t])
PY
