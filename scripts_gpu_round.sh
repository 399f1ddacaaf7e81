#!/bin/bash
# one GPU-box visit: smoke, GPU tests, small + full bench (all under timeouts)
mkdir -p gpurun_out
python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
timeout 600 python -m pytest tests -m gpu -x -q --timeout=200 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
timeout 200 python bench.py --channels 512 --seconds 10 --steps 3 --no-cpu-baseline > gpurun_out/bench_small.log 2>&1; echo "bench small rc=$?"; tail -2 gpurun_out/bench_small.log
timeout 400 python bench.py --steps 3 > gpurun_out/bench_full.log 2>&1; echo "bench full rc=$?"; tail -2 gpurun_out/bench_full.log
