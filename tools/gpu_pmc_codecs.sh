#!/bin/bash
# SQ counters for the secondary codec kernels (tools/bench_codecs.py shapes), one pass of 8 SQ slots.
mkdir -p gpurun_out/pmc_codecs
cd /tmp && export TMPDIR=/tmp
timeout 800 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_codecs/a -o pmc -- python $GRAFT_REPO_ROOT/tools/bench_codecs.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_codecs/a.log 2>&1; echo "rc=$?"
