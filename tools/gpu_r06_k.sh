#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_hca -o hca -- python $GRAFT_REPO_ROOT/bench.py --codec hca --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $O/prof_hca.log 2>&1
cd $GRAFT_REPO_ROOT
head -12 $O/prof_hca/hca_kernel_stats.csv | cut -c1-200
