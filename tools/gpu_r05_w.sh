#!/bin/bash
# round 5, the evidence pass in one visit: tools/gpu_r05_j.sh (suite, kernel stats, PMC traffic, SQ counters, 8-rank plumbing,
# signal-class kernel split, ragged host calls) and then the bench lines
cd $GRAFT_REPO_ROOT
bash tools/gpu_r05_j.sh
cd $GRAFT_REPO_ROOT
STAGES="bench default" bash tools/gpu_round5.sh > gpurun_out/r05_round5_bench.log 2>&1
cat gpurun_out/r05/bench_default.time
