#!/bin/bash
# round 5: the ragged ADX call's kernels under rocprofv3 once more, on the final sources (compare profiles/r05_q_ragged_host_orders.log)
O=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ragged_adx -o ragged -- python $GRAFT_REPO_ROOT/tools/time_ragged_host.py --codecs adx --orders 1 --reps 1 > $O/prof_ragged_adx_final.log 2>&1; echo "rocprof rc=$?"
f=$(find $O/prof_ragged_adx -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_ragged_adx_final.csv && head -8 $f | cut -c1-200
rm -rf $O/prof_ragged_adx
