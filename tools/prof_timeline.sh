#!/bin/bash
# start offsets and durations of the last kernels of `bench.py --codec adx` (rocprofv3 --kernel-trace): where a launch's gaps are
D=$(mktemp -d /tmp/tl.XXXX)
cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $D -o t -- python $GRAFT_REPO_ROOT/bench.py --codec adx --steps 3 --warmup 1 --no-cpu-baseline > $D/log 2>&1
python3 - $D <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
rows = [r for r in rows if 'synth' not in r['Kernel_Name'] and 'copyBuffer' not in r['Kernel_Name']]
t0 = None
for r in rows[-int(__import__('os').environ.get('TL_ROWS', '24')):]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if t0 is None: t0 = s
    print("%9.3f ms  +%8.3f ms  %s" % ((s - t0) / 1e6, (e - s) / 1e6, r['Kernel_Name'][:70]))
PY
