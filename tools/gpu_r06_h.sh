#!/bin/bash
# round 6, call h: CU-driven gather of registered host rows against one hipMemcpyAsync per row
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
timeout 300 tools/variants/bench_h2d_gather 4 > $O/r06_h_h2d_gather.log 2>&1
cat $O/r06_h_h2d_gather.log
