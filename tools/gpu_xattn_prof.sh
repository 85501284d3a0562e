#!/bin/bash
# per-kernel averages of the Transformer recogniser's stages (tools/trocr_stage_bench.py) with the cross-attention on the encoder states
cd $GRAFT_REPO_ROOT
out=gpurun_out/xattn_prof
mkdir -p $out
export TMPDIR=/tmp
for x in ${XS:-1}; do
VTD_TROCR_XATTN=$x REPS=2 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats$x -o run -- python3 tools/trocr_stage_bench.py > $out/stats$x.log 2>&1 || { tail -5 $out/stats$x.log; exit 1; }
tail -2 $out/stats$x.log
python - <<PY
import csv, glob
rows=list(csv.DictReader(open(glob.glob("$out/stats$x/**/run_kernel_stats.csv", recursive=True)[0])))
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:22]:
    print("%7d calls %8.1f us avg %9.1f ms total  %s" % (int(r['Calls']), float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, r['Name'][:100]))
PY
python - <<PY
import csv, glob, collections
# the cross-attention's launches by grid height (the host's lagged row bound): duration and bytes / duration
by = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob("$out/stats$x/**/run_kernel_trace.csv", recursive=True)[0])):
    n = r["Kernel_Name"]
    if "dec_xattn" in n or "dec_attn_kernel<false" in n or "dec_xv" in n or "dec_xq" in n:
        g = int(r.get("Grid_Size", r.get("Grid_Size_X", 0))) // int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 1)))
        by[(n.split("(")[1 if n.startswith("void (") else 0][:28] if False else ("xattn" if "dec_xattn" in n else "xv" if "dec_xv" in n else "xq" if "dec_xq" in n else "attn_kv"), g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (k, g), d in sorted(by.items()):
    if len(d) >= 12: print("%-8s grid %6d: %5d launches %8.1f us avg %8.1f min" % (k, g, len(d), sum(d) / len(d), min(d)))
PY
find $out -name "*kernel_trace.csv" -delete
done
