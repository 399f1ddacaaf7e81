#!/bin/bash
# SQ issue/stall counters for the GC-ADPCM kernels (one pass: 8 SQ slots), kernel-trace only.
mkdir -p gpurun_out/pmc_sq
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sq/a -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_sq/a.log 2>&1; echo "a rc=$?"
if [ -n "$SECOND_PASS" ]; then
timeout 500 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sq/b -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_sq/b.log 2>&1; echo "b rc=$?"
fi
cd $GRAFT_REPO_ROOT; for f in $(find gpurun_out/pmc_sq -name "*counter_collection*.csv"); do echo $f; head -2 $f | cut -c1-300; done
