#!/bin/bash
# round 5, evidence visit 1: the suite, kernel stats, PMC traffic, SQ counters on the frozen kernels; per-kernel split of the two
# signal classes whose seams do not close
cd $GRAFT_REPO_ROOT
STAGES="smoke tests prof pmc sq ranks8" bash tools/gpu_round5.sh > gpurun_out/r05_round5.log 2>&1
tail -5 gpurun_out/r05_round5.log | cut -c1-200
grep -v amdgpu gpurun_out/r05/pytest_gpu.log | tail -2
O=$GRAFT_REPO_ROOT/gpurun_out/r05
cd /tmp && export TMPDIR=/tmp
for cls in clipped_square sine440; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$cls -o $cls -- python $GRAFT_REPO_ROOT/tools/time_signal_class.py $cls > $O/signal_$cls.log 2>&1
  grep -v amdgpu.ids $O/signal_$cls.log | grep " ms"
  f=$(find $O/prof_$cls -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_signal_$cls.csv
done
# the ragged ADX / HCA host calls as the product runs them (two upload streams) with the buckets in both orders
cd $GRAFT_REPO_ROOT
timeout 600 python tools/time_ragged_host.py > $O/ragged_host_final.log 2>&1; grep -v amdgpu $O/ragged_host_final.log | tail -9
