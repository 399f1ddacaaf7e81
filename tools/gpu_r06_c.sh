#!/bin/bash
# round 6, call c: wave-per-frame HCA encoder after the emission rewrite + register work: parity, timing of variants
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_hca.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python tools/time_hca_decode.py > $O/r06_c_hca_encode_variants.log 2>&1
grep -v amdgpu.ids $O/r06_c_hca_encode_variants.log | cut -c1-200
