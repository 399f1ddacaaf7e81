#!/bin/bash
# PMC passes (HBM traffic) for the headline kernels: separate --pmc runs, kernel-trace only.
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc/$c -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc/$c.log 2>&1; echo "$c rc=$?"
done
cd $GRAFT_REPO_ROOT; find gpurun_out/pmc -name "*.csv" | head; for f in $(find gpurun_out/pmc -name "*counter_collection*.csv"); do echo $f; head -3 $f; done
