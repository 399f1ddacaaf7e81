#!/bin/bash
# round 5: where the ragged ADX host call spends the time after its upload -- the pipeline's own timeline (events per chunk) for
# both bucket orders, and the kernels of one call under rocprofv3
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $O
VGA_HIP_PIPELINE_TIMELINE=1 timeout 600 python tools/time_ragged_host.py --codecs adx --reps 1 > $O/ragged_adx_timeline.log 2>&1; echo "timeline rc=$?"
grep -v amdgpu $O/ragged_adx_timeline.log | tail -120
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ragged_adx -o ragged -- python $GRAFT_REPO_ROOT/tools/time_ragged_host.py --codecs adx --reps 1 > $O/prof_ragged_adx.log 2>&1; echo "rocprof rc=$?"
f=$(find $O/prof_ragged_adx -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_ragged_adx.csv && head -12 $f | cut -c1-200
t=$(find $O/prof_ragged_adx -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python3 - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
last = rows[-600:] if len(rows) > 600 else rows
# the last call's kernels: name, start, duration (ms), grid
for r in rows[-200:]:
    n = r["Kernel_Name"].split("(")[0][-50:]
    print("%-52s start %9.2f dur %8.3f grid %s" % (n, (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Grid_Size", "")))
PY
rm -rf $O/prof_ragged_adx
