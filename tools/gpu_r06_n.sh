#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O; rm -rf $O/prof_small
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_small -o small -- python $GRAFT_REPO_ROOT/tools/time_encode_channels.py 96 > $O/prof_small.log 2>&1
cd $GRAFT_REPO_ROOT
head -12 $O/prof_small/small_kernel_stats.csv | cut -c1-160
