#!/bin/bash
# round 6, call z: kernel timeline of the ragged coefficient search (five-wave channels on a side stream) inside bench.py's mixed-lengths leg
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --codec gc --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-other-configs --no-signals > /dev/null 2>&1
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee $O/r06_z_ragged_coefs_kernel_timeline.log
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
for r in rows:
    n = r["Kernel_Name"]
    if "gc_coefs" in n or "gc_encode_persistent" in n:
        print("%-40s start %10.3f ms  end %10.3f ms  dur %8.3f ms  grid %s wg %s queue %s" % (n.split("(")[0][-40:], (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6,
              (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), r.get("Queue_Id", "?")))
PY
