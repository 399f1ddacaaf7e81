#!/bin/bash
# one GPU-box visit: smoke, GPU tests, bench, rocprof kernel stats (all under timeouts)
mkdir -p gpurun_out
python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
timeout 900 python -m pytest tests -m gpu -x -q --timeout=300 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
timeout 600 python bench.py ${BENCH_ARGS:---no-cpu-baseline --steps 3} > gpurun_out/bench_full.log 2>&1; echo "bench full rc=$?"; tail -1 gpurun_out/bench_full.log
if [ -n "$WITH_PROF" ]; then
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT; for f in $(find gpurun_out/prof -name "*kernel_stats*.csv" | head -1); do head -6 $f; done
fi
