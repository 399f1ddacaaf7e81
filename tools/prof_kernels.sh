#!/bin/bash
# Per-kernel average durations (rocprofv3 --kernel-trace --stats) of any GPU command, e.g.
#   tools/prof_kernels.sh python tools/time_tile_variants.py        (GC decode, ADX encode / decode at 4096 x 60 s)
# Prints the kernels of this library (vga::) with calls and average microseconds.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
D=$(mktemp -d /tmp/profk.XXXXXX)
args=(); for a in "$@"; do if [ -e "$R/$a" ] && [ "${a:0:1}" != "/" ]; then args+=("$R/$a"); else args+=("$a"); fi; done   # rocprofv3 runs from /tmp
( cd /tmp && TMPDIR=/tmp timeout ${PROF_TIMEOUT:-600} rocprofv3 --kernel-trace --stats --output-format csv -d $D -o p -- "${args[@]}" > $D/log.txt 2>&1 )
f=$(find $D -name "*kernel_stats*.csv" | head -1)
[ -z "$f" ] && { tail -5 $D/log.txt; exit 1; }
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "vga::" in r["Name"]:
        print("%-56s calls %4s  avg %10.1f us" % (r["Name"].split("(")[0].replace("void ", "")[-56:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf $D
