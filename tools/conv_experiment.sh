#!/bin/bash
# timing-only experiment: how much of each convolution's time is the A-operand gather / the loads at all
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp
# the hook is compiled in only for this experiment (it costs branches in the hot loop): it goes into its own library
# (libvtd_hip_convexp.so, own object directory); the product library is never rebuilt with these flags
export VTD_LIB_VARIANT=convexp
VTD_EXTRA_HIPCC_FLAGS=-DVTD_CONV_EXPERIMENT python video-text-detection-system_amd/build_native.py > gpurun_out/exp/build.log 2>&1 || exit 1
export VTD_HALO_CONV=0
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
