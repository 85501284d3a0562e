#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/stamps
mkdir -p $out
VTD_HALO_STAMPS=1 timeout -k 10 200 python bench.py --workload detector --steps 2 --warmup 1 --cpu-seconds 0 --no-profile > $out/st.json 2> $out/st.err || { tail -5 $out/st.err; exit 1; }
grep "stamps\]" $out/st.err | sort | uniq -c | sort -rn | head -12
