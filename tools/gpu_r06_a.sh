#!/bin/bash
# round 6, call a: the wave-per-frame HCA encoder: parity tests, then configs[3] timing
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_hca.py -x -q -m gpu > gpurun_out/r06_a_pytest_hca.log 2>&1
tail -15 gpurun_out/r06_a_pytest_hca.log
timeout 300 python tools/time_hca_decode.py > gpurun_out/r06_a_time_hca.log 2>&1
cat gpurun_out/r06_a_time_hca.log
