#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O; rm -rf $O/pmc_sq_hca
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_sq_hca/a -o pmc -- python $GRAFT_REPO_ROOT/bench.py --codec hca --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $O/pmc_sq_hca_a.log 2>&1; echo "a rc=$?"
timeout 500 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc_sq_hca/b -o pmc -- python $GRAFT_REPO_ROOT/bench.py --codec hca --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $O/pmc_sq_hca_b.log 2>&1; echo "b rc=$?"
timeout 500 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_LDS_ADDR_CONFLICT --kernel-trace --output-format csv -d $O/pmc_sq_hca/c -o pmc -- python $GRAFT_REPO_ROOT/bench.py --codec hca --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $O/pmc_sq_hca_c.log 2>&1; echo "c rc=$?"
cd $GRAFT_REPO_ROOT
python tools/summarize_pmc.py sq $O/r06_l_sq_counters_hca.json $O/pmc_sq_hca/a $O/pmc_sq_hca/b $O/pmc_sq_hca/c | grep -B1 -A30 '"hca_frames_wave_kernel"'
